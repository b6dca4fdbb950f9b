// gemm.hip — C = act(A . W^T + bias) + residual for gfx950, bf16 (v_mfma_f32_16x16x32_bf16) and exact
// fp32 (v_mfma_f32_16x16x4_f32).  Two kernels:
//   * gemm_tiled_kernel : 128x128 block tile, 4 waves (2x2, 64x64 each), 128-byte K slabs staged
//     through XOR-swizzled LDS (conflict-free ds_read_b128), register double-buffering of the global
//     loads.  MFMA-bound shapes: HuBERT conv-as-GEMM, encoder layers, Llama prefill.
//   * gemm_skinny_kernel: M <= 64 rows (KV-cached decode, M = batch).  HBM-bound weight streaming:
//     one block owns 16 (or 32 gate/up) weight rows, its waves interleave 64-byte K steps, weight
//     fragments go global -> VGPR -> MFMA with no LDS round trip, partial sums meet in LDS once.
// A rows may overlap (lda < K): that is how the strided convolutions run without im2col.
#include <stdlib.h>

#include "common.h"
#include "gemm_internal.h"

static int g_disable_glds = 0;  // tuning switch (SL_DISABLE_GLDS=1): A/B the two staging paths in one process

// packed-weight GEMMs with more rows than this run the streaming kernel; SL_STREAM_MIN_M overrides (tuning)
static int stream_min_m() { return sl_env().stream_min_m; }

#include "gemm_epilogue.h"

template <typename T, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_tiled_kernel(GemmP p) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int BK = TROWB / (int)sizeof(T);  // 64 bf16 / 32 f32
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][TBM * TROWB];  // [buf][A|W]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, q = lane >> 4;

  // XCD-aware tile order: consecutive blocks of one XCD (blockIdx % 8 equal) walk tiles that share
  // the same W panel, so the panel stays in that XCD's L2 (bijective remap, guide §5 T1).
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int qn = nt >> 3, rn = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  int bm, bn;   // 8 x 8 patches of tiles per XCD at a time (see gemm_tiled256_kernel)
  {
    constexpr int GM = 8;
    const int per = GM * p.tiles_n, grp = bid / per, first = grp * GM;
    const int gsz = (p.tiles_m - first) < GM ? (p.tiles_m - first) : GM;
    const int in = bid - grp * per;
    bm = first + in % gsz;
    bn = in / gsz;
  }
  const int z = blockIdx.y;

  int64_t a_off; int wz;
  if (!resolve_group(p, z, bm, a_off, wz)) return;
  const T* A = (const T*)p.A + a_off;
  const T* W = (const T*)p.W + (int64_t)wz * p.sW + p.wx;
  if (p.grp_ext && bn * TBN >= p.N) return;

  // staging assignment: 4 chunks of A and 4 of W per thread
  const T* ga[4];
  const T* gw[4];
  int so[4];
  int kc;  // this thread's k offset (elements) inside a slab
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
    int ar = bm * TBM + row; ar = ar < p.M ? ar : p.M - 1;
    int wr = bn * TBN + row; wr = wr < p.N ? wr : p.N - 1;
    ga[i] = A + (int64_t)ar * p.lda + ch * VEC;
    gw[i] = W + (int64_t)wr * p.ldw + ch * VEC;
    so[i] = lds_off(row, ch);
  }
  kc = (tid & 7) * VEC;
  // transposed operands (backward GEMMs): 16-byte chunks run along the OUTPUT index, one reduction row each
  constexpr int CPRT = 128 / VEC;  // chunks per reduction row of a 128-wide tile
  const T* gat[4];
  const T* gwt[4];
  int tk[4], tcol[4];
  bool aok[4], wok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    tk[i] = c / CPRT;
    tcol[i] = (c % CPRT) * VEC;
    const int am = bm * TBM + tcol[i], wn = bn * TBN + tcol[i];
    aok[i] = am < p.M; wok[i] = wn < p.N;
    gat[i] = A + (int64_t)tk[i] * p.lda + (aok[i] ? am : 0);
    gwt[i] = W + (int64_t)tk[i] * p.ldw + (wok[i] ? wn : 0);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (p.K + BK - 1) / BK;
  uint4 ra[4], rw[4];

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    const bool ok = (k0 + kc) < p.K;  // K % VEC == 0, so a chunk is entirely in or out
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (p.ta) ra[i] = (aok[i] && k0 + tk[i] < p.K) ? *(const uint4*)(gat[i] + (int64_t)k0 * p.lda) : make_uint4(0, 0, 0, 0);
      else ra[i] = ok ? *(const uint4*)(ga[i] + k0) : make_uint4(0, 0, 0, 0);
      if (p.tw) rw[i] = (wok[i] && k0 + tk[i] < p.K) ? *(const uint4*)(gwt[i] + (int64_t)k0 * p.ldw) : make_uint4(0, 0, 0, 0);
      else rw[i] = ok ? *(const uint4*)(gw[i] + k0) : make_uint4(0, 0, 0, 0);
    }
  };
  // transposed chunk -> LDS: element e belongs to tile row tcol+e, reduction index tk
  auto scatter = [&](unsigned char* tile, const uint4& u, int col0, int k) {
    T e[VEC];
    *(uint4*)e = u;
#pragma unroll
    for (int j = 0; j < VEC; ++j) *(T*)(tile + lds_off(col0 + j, k / VEC) + (k % VEC) * (int)sizeof(T)) = e[j];
  };
  // bf16: a transposed operand keeps its [reduction row][128 outputs] shape in LDS (256-byte rows, the guide's image (b):
  // chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) — one 16-byte store per chunk instead of eight 2-byte scatters — and
  // the MFMA fragments are gathered column-wise by ds_read_b64_tr_b16 (conflict-free: a 32-lane half reads two 4-row
  // blocks 8 rows apart).  The 2-byte scatter remains for fp32 (no 32-bit transposed read).
  constexpr bool TRT = sizeof(T) == 2;
  auto t_off = [](int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (p.ta) {
        if constexpr (TRT) *(uint4*)(&smem[buf][0][t_off(tk[i], tcol[i] / VEC)]) = ra[i];
        else scatter(&smem[buf][0][0], ra[i], tcol[i], tk[i]);
      } else *(uint4*)(&smem[buf][0][so[i]]) = ra[i];
      if (p.tw) {
        if constexpr (TRT) *(uint4*)(&smem[buf][1][t_off(tk[i], tcol[i] / VEC)]) = rw[i];
        else scatter(&smem[buf][1][0], rw[i], tcol[i], tk[i]);
      } else *(uint4*)(&smem[buf][1][so[i]]) = rw[i];
    }
  };
  // fragment of 16 outputs x 32 reduction steps out of a transposed-image tile: rows s*32 + 8q .. +7, columns col0 + r
  const int qq = r >> 2, pp = r & 3;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_g_t;
  // the four 16-output fragments of a wave's 64 outputs: all eight reads in flight, one wait
  auto tr_frags = [&](const unsigned char* tile, int s_, int col0, uint4 (&f)[4]) {
    const uint32_t base = (uint32_t)(uintptr_t)(lds_ptr_t)tile;
    const int r0 = s_ * 32 + 8 * q + qq;
    u32x2_g_t lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch0 = (col0 + 16 * j) / 8;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[j]) : "v"(base + (uint32_t)(t_off(r0, ch0 + (pp >> 1)) + 8 * (pp & 1))) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi[j]) : "v"(base + (uint32_t)(t_off(r0 + 4, ch0 + (pp >> 1)) + 8 * (pp & 1))) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = make_uint4(lo[j].x, lo[j].y, hi[j].x, hi[j].y);
  };

  gload(0);
  sstore(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
    const unsigned char* sa = &smem[buf][0][0];
    const unsigned char* sw = &smem[buf][1][0];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 fa[4], fb[4];
      if (TRT && p.ta) {
        if constexpr (TRT) {
          tr_frags(sa, s, wm * 64, fa);
        }
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[m] = *(const uint4*)(sa + lds_off(wm * 64 + m * 16 + r, s * 4 + q));
      }
      if (TRT && p.tw) {
        if constexpr (TRT) {
          tr_frags(sw, s, wn * 64, fb);
        }
      } else {
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[n] = *(const uint4*)(sw + lds_off(wn * 64 + n * 16 + r, s * 4 + q));
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) MMA<T>::step(acc[m][n], fa[m], fb[n]);
    }
    if (kt + 1 < nkt) sstore(buf ^ 1);
    __syncthreads();
  }

  if constexpr (ACT != SL_ACT_SILU_MUL) {
    if (!p.direct_epi && tile_epilogue_rows<T, ACT, 4>(p, acc, bm * TBM + wm * 64, bn * TBN + wn * 64, lane, z, wz, (float*)&smem[0][0][0] + wave * 4096)) return;
  }
  tile_epilogue<T, ACT>(p, acc, bm, bn, wm, wn, q, r, z, wz);
}

// ----------------------------------------------------------------------------------------------
// tiled kernel, direct-to-LDS staging (global_load_lds_dwordx4): same tile, same swizzled LDS image — the
// swizzle moves to the per-lane SOURCE address because the LDS side of an LDS-DMA is lane-linear — no
// staging VGPRs, no ds_write pass.  Used when K is a whole number of 128-byte slabs and no operand is
// transposed (every forward GEMM of the encoder / prefill at model shapes).
// ----------------------------------------------------------------------------------------------


// DMAB (round 6, with ASMLDS, bf16): the eight DMA requests of slab kt + 1 are issued BETWEEN the MFMAs of slab kt's first k-step (one per two,
// MFMAs as asm statements so that the order holds) instead of in a burst at the top of the iteration: a request costs the issuing wave
// ~60-185 cycles (guide: LDS-DMA piece issue cost) during which its SIMD's matrix pipe has only the other block's wave to draw from.  Same
// MFMA order, same bits.  SL_GLDS_DMAB=1 selects it; default off — with two blocks per CU the other block covers the burst (7 984 x 3 072 x 1 024: 795 against 845 TF/s, the rest within 1 %, profiles/r06_ah_gemm_vs_vendor_mid.txt); at ONE block per CU the same placement is worth +20 % (gemm128.hip).
template <typename T>
__device__ __forceinline__ void mfma_fence_step(f32x4& acc, const u32x4_t& a, const u32x4_t& b) {
  if constexpr (sizeof(T) == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "memory");
}

template <typename T, int ACT, bool ASMLDS = false, bool DMAB = false>
__global__ __launch_bounds__(256, 2) void gemm_tiled_glds_kernel(GemmP p) {
  static_assert(!DMAB || (ASMLDS && sizeof(T) == 2), "the interleaved form is the bf16 asm-read loop");
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int BK = TROWB / (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][TBM * TROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, q = lane >> 4;
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int qn = nt >> 3, rn = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  int bm, bn;   // 8 x 8 patches of tiles per XCD at a time (see gemm_tiled256_kernel)
  {
    constexpr int GM = 8;
    const int per = GM * p.tiles_n, grp = bid / per, first = grp * GM;
    const int gsz = (p.tiles_m - first) < GM ? (p.tiles_m - first) : GM;
    const int in = bid - grp * per;
    bm = first + in % gsz;
    bn = in / gsz;
  }
  const int z = blockIdx.y;
  int64_t a_off; int wz;
  if (!resolve_group(p, z, bm, a_off, wz)) return;
  const T* A = (const T*)p.A + a_off;
  const T* W = (const T*)p.W + (int64_t)wz * p.sW + p.wx;
  if (p.grp_ext && bn * TBN >= p.N) return;

  // LDS chunk c = tid + 256 i sits at (row c>>3, physical chunk c&7) and must hold logical chunk (c&7)^(row&7)
  const T* ga[4];
  const T* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 3, ch = (c & 7) ^ (row & 7);
    int ar = bm * TBM + row; ar = ar < p.M ? ar : p.M - 1;
    int wr = bn * TBN + row; wr = wr < p.N ? wr : p.N - 1;
    ga[i] = A + (int64_t)ar * p.lda + ch * VEC;
    gw[i] = W + (int64_t)wr * p.ldw + ch * VEC;
  }
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;  // this wave's 1 KiB piece inside a 4 KiB group

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = p.K / BK;
  auto issue = [&](int kt, int buf) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[i] + k0), (lds_ptr_t)(&smem[buf][0][i * 4096 + wave_lds]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(gw[i] + k0), (lds_ptr_t)(&smem[buf][1][i * 4096 + wave_lds]), 16, 0, 0);
    }
  };

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if constexpr (DMAB) {
    const uint32_t x0 = (uint32_t)((q ^ (r & 7)) << 4), x1 = (uint32_t)(((4 + q) ^ (r & 7)) << 4);
    const uint32_t sb0 = (uint32_t)(uintptr_t)(lds_ptr_t)(&smem[0][0][0]);
    const uint32_t ra0 = sb0 + (uint32_t)((wm * 64 + r) * TROWB), rb0 = sb0 + (uint32_t)(TBM * TROWB + (wn * 64 + r) * TROWB);
    u32x4_t a0[4], b0[4], a1[4], b1[4];
    auto reads = [&](int buf) {
      const uint32_t ra = ra0 + (uint32_t)(buf * 2 * TBM * TROWB), rb = rb0 + (uint32_t)(buf * 2 * TBM * TROWB);
      SL_LDS_RD(a0[0], ra + x0, 0); SL_LDS_RD(a0[1], ra + x0, 2048); SL_LDS_RD(a0[2], ra + x0, 4096); SL_LDS_RD(a0[3], ra + x0, 6144);
      SL_LDS_RD(b0[0], rb + x0, 0); SL_LDS_RD(b0[1], rb + x0, 2048); SL_LDS_RD(b0[2], rb + x0, 4096); SL_LDS_RD(b0[3], rb + x0, 6144);
      SL_LDS_RD(a1[0], ra + x1, 0); SL_LDS_RD(a1[1], ra + x1, 2048); SL_LDS_RD(a1[2], ra + x1, 4096); SL_LDS_RD(a1[3], ra + x1, 6144);
      SL_LDS_RD(b1[0], rb + x1, 0); SL_LDS_RD(b1[1], rb + x1, 2048); SL_LDS_RD(b1[2], rb + x1, 4096); SL_LDS_RD(b1[3], rb + x1, 6144);
      lds_wait8<8>(a0[0], a0[1], a0[2], a0[3], b0[0], b0[1], b0[2], b0[3]);
    };
    auto step1 = [&]() {
      lds_wait8<0>(a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) mfma_fence_step<T>(acc[m][n], a1[m], b1[n]);
    };
    for (int kt = 0; kt + 1 < nkt; ++kt) {
      const int buf = kt & 1;
      reads(buf);
      const int k0 = (kt + 1) * BK;
      unsigned char* dst = &smem[buf ^ 1][0][0] + wave_lds;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          mfma_fence_step<T>(acc[m][n], a0[m], b0[n]);
          if (n == 1) __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[m] + k0), (lds_ptr_t)(dst + m * 4096), 16, 0, 0);
          if (n == 3) __builtin_amdgcn_global_load_lds((glb_ptr_t)(gw[m] + k0), (lds_ptr_t)(dst + TBM * TROWB + m * 4096), 16, 0, 0);
        }
      step1();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    reads((nkt - 1) & 1);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) mfma_fence_step<T>(acc[m][n], a0[m], b0[n]);
    step1();
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // XDL write -> VALU read wait states the compiler cannot see behind asm MFMAs
    __builtin_amdgcn_s_barrier();
  } else
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) issue(kt + 1, buf ^ 1);
    if constexpr (ASMLDS) {
      // both 64-byte k-steps of the slab are requested up front; the MFMAs of step 0 run under the reads of step 1,
      // and the DMA of slab kt+1 (issued above) runs under all of it
      const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(&smem[buf][0][0]);
      const uint32_t x0 = (uint32_t)((q ^ (r & 7)) << 4), x1 = (uint32_t)(((4 + q) ^ (r & 7)) << 4);
      const uint32_t ra = sb + (uint32_t)((wm * 64 + r) * TROWB), rb = sb + (uint32_t)(TBM * TROWB + (wn * 64 + r) * TROWB);
      u32x4_t a0[4], b0[4], a1[4], b1[4];
      SL_LDS_RD(a0[0], ra + x0, 0); SL_LDS_RD(a0[1], ra + x0, 2048); SL_LDS_RD(a0[2], ra + x0, 4096); SL_LDS_RD(a0[3], ra + x0, 6144);
      SL_LDS_RD(b0[0], rb + x0, 0); SL_LDS_RD(b0[1], rb + x0, 2048); SL_LDS_RD(b0[2], rb + x0, 4096); SL_LDS_RD(b0[3], rb + x0, 6144);
      SL_LDS_RD(a1[0], ra + x1, 0); SL_LDS_RD(a1[1], ra + x1, 2048); SL_LDS_RD(a1[2], ra + x1, 4096); SL_LDS_RD(a1[3], ra + x1, 6144);
      SL_LDS_RD(b1[0], rb + x1, 0); SL_LDS_RD(b1[1], rb + x1, 2048); SL_LDS_RD(b1[2], rb + x1, 4096); SL_LDS_RD(b1[3], rb + x1, 6144);
      lds_wait8<8>(a0[0], a0[1], a0[2], a0[3], b0[0], b0[1], b0[2], b0[3]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) MMA<T>::step(acc[m][n], as_uint4(a0[m]), as_uint4(b0[n]));
      __builtin_amdgcn_sched_barrier(0);
      lds_wait8<0>(a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) MMA<T>::step(acc[m][n], as_uint4(a1[m]), as_uint4(b1[n]));
      __builtin_amdgcn_sched_barrier(0);   // keep the DMA wait and the barrier BELOW the MFMAs (they carry no data dependence)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      const unsigned char* sa = &smem[buf][0][0];
      const unsigned char* sw = &smem[buf][1][0];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint4 fa[4], fb[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[m] = *(const uint4*)(sa + lds_off(wm * 64 + m * 16 + r, s * 4 + q));
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[n] = *(const uint4*)(sw + lds_off(wn * 64 + n * 16 + r, s * 4 + q));
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) MMA<T>::step(acc[m][n], fa[m], fb[n]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  if constexpr (ACT != SL_ACT_SILU_MUL) {
    if (!p.direct_epi && tile_epilogue_rows<T, ACT, 4, ASMLDS>(p, acc, bm * TBM + wm * 64, bn * TBN + wn * 64, lane, z, wz, (float*)&smem[0][0][0] + wave * 4096)) return;
  }
  tile_epilogue<T, ACT>(p, acc, bm, bn, wm, wn, q, r, z, wz);
}

// ----------------------------------------------------------------------------------------------
// skinny kernel (M <= 16*MT): HBM-bound weight streaming for decode.
//   block = RF 16-row weight fragments x NW waves; wave w takes 64-byte k-steps w, w+NW, ... with U steps
//   of loads in flight; W fragments go global -> VGPR -> MFMA (A operand), the M activation rows are the
//   B operand (re-read from L2); partial sums meet once in LDS.
//   PACKED: W is stored fragment-major [ceil(N/16)][K/KSTEP][64 lanes][16 B] (sl_pack_weight layout), so
//   every wave-level weight load is one contiguous 1 KiB (measured +25..45 % over row-major 16 x 64 B).
//   Epilogues: NONE/GELU (+bias, +residual), SILU_MUL (fragment pairs = gate, up), ROPE_KV (fragment
//   pairs = the two rotate_half halves of a head; q is written rotated, k/v go straight into the cache).
//   fuse_rms: the RMSNorm gain is pre-folded into W and the per-row rsqrt(mean(x^2)+eps) is computed
//   from the x fragments the block loads anyway, then applied to the accumulators.
// ----------------------------------------------------------------------------------------------

template <typename T, int MT, int ACT, int RF, int NW, int U, bool PACKED, bool KCONT = false>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmP p, SkinnyX sx) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int KSTEP = MMA<T>::KSTEP;
  constexpr int RB = 16 * RF;
  constexpr bool PAIRS = (ACT == SL_ACT_SILU_MUL || ACT == SL_ACT_ROPE_KV);
  static_assert(!PAIRS || RF % 2 == 0, "pair epilogues need an even number of fragments");
  __shared__ float red[NW][RB][MT * 16 + 1];
  __shared__ float red_ss[NW][4][MT * 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int z = blockIdx.y;
  const int n0 = blockIdx.x * RB;
  const int nks_full = p.K / KSTEP;  // full 64-byte steps

  const T* A = (const T*)p.A + (int64_t)z * p.sA;
  const T* W = (const T*)p.W + (int64_t)z * p.sW;

  const T* wp[RF];
  int64_t wstep;  // elements between consecutive k-steps of one fragment row
  if constexpr (PACKED) {
    const int nfrag = (p.N + 15) >> 4;
    wstep = 64 * VEC;
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      int fi = (n0 >> 4) + f; fi = fi < nfrag ? fi : nfrag - 1;
      wp[f] = W + (int64_t)fi * nks_full * (64 * VEC) + lane * VEC;
    }
  } else {
    wstep = KSTEP;
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      int wr = n0 + f * 16 + r; wr = wr < p.N ? wr : p.N - 1;
      wp[f] = W + (int64_t)wr * p.ldw + q * VEC;
    }
  }
  const T* xp[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int xr = t * 16 + r; xr = xr < p.M ? xr : p.M - 1;
    xp[t] = A + (int64_t)xr * p.lda + q * VEC;
  }

  f32x4 acc[RF][MT];
  float ss[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    ss[t] = 0.f;
#pragma unroll
    for (int f = 0; f < RF; ++f) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool fuse = sx.fuse_rms != 0;

  auto sumsq = [&](const uint4& u, float& s) {
    float e[VEC];
    Vec16<T>::unpack(u, e);
#pragma unroll
    for (int j = 0; j < VEC; ++j) s = fmaf(e[j], e[j], s);
  };

  // K split over the block's waves: interleaved 64-byte steps (w, w+NW, ...) or one contiguous slice per wave
  int ks, kend;
  constexpr int KSTR = KCONT ? 1 : NW;
  if constexpr (KCONT) {
    const int per = (nks_full + NW - 1) / NW;
    ks = wave * per;
    kend = (ks + per) < nks_full ? (ks + per) : nks_full;
  } else {
    ks = wave;
    kend = nks_full;
  }
  for (; ks + (U - 1) * KSTR < kend; ks += U * KSTR) {
    uint4 fw[U][RF], fx[U][MT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t kk = ks + u * KSTR;
#pragma unroll
      for (int f = 0; f < RF; ++f) fw[u][f] = ld_nt16(wp[f] + kk * wstep);
#pragma unroll
      for (int t = 0; t < MT; ++t) fx[u][t] = *(const uint4*)(xp[t] + kk * KSTEP);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int t = 0; t < MT; ++t) MMA<T>::step(acc[f][t], fw[u][f], fx[u][t]);
      if (fuse) {
#pragma unroll
        for (int t = 0; t < MT; ++t) sumsq(fx[u][t], ss[t]);
      }
    }
  }
  for (; ks < kend; ks += KSTR) {
    uint4 fw[RF], fx[MT];
#pragma unroll
    for (int f = 0; f < RF; ++f) fw[f] = ld_nt16(wp[f] + (int64_t)ks * wstep);
#pragma unroll
    for (int t = 0; t < MT; ++t) fx[t] = *(const uint4*)(xp[t] + (int64_t)ks * KSTEP);
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
      for (int t = 0; t < MT; ++t) MMA<T>::step(acc[f][t], fw[f], fx[t]);
    if (fuse) {
#pragma unroll
      for (int t = 0; t < MT; ++t) sumsq(fx[t], ss[t]);
    }
  }
  if constexpr (!PACKED) {
    // K tail (K % KSTEP != 0): one predicated step, taken by the wave whose turn it is
    if (nks_full * KSTEP < p.K && wave == (KCONT ? NW - 1 : nks_full % NW)) {
      const int64_t k = (int64_t)nks_full * KSTEP;
      const bool ok = (k + q * VEC) < p.K;
      uint4 fw[RF], fx[MT];
#pragma unroll
      for (int f = 0; f < RF; ++f) fw[f] = ok ? *(const uint4*)(wp[f] + k) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < MT; ++t) fx[t] = ok ? *(const uint4*)(xp[t] + k) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int t = 0; t < MT; ++t) MMA<T>::step(acc[f][t], fw[f], fx[t]);
      if (fuse) {
#pragma unroll
        for (int t = 0; t < MT; ++t) sumsq(fx[t], ss[t]);
      }
    }
  }

  // D[weight row 4q+i][x row r]  ->  red[wave][f*16 + 4q+i][t*16 + r]
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave][f * 16 + q * 4 + i][t * 16 + r] = acc[f][t][i];
  if (fuse) {
#pragma unroll
    for (int t = 0; t < MT; ++t) red_ss[wave][q][t * 16 + r] = ss[t];
  }
  __syncthreads();

  void* Cb = p.out_f32 ? (void*)((float*)p.C + (int64_t)z * p.sC) : (void*)((T*)p.C + (int64_t)z * p.sC);
  const T* bias = p.bias ? (const T*)p.bias + (int64_t)z * p.sBias : nullptr;
  const void* Rb = p.res ? (const void*)((const T*)p.res + (int64_t)z * p.sR) : nullptr;
  auto row_scale = [&](int m) -> float {
    if (!fuse) return 1.0f;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red_ss[w][0][m] + red_ss[w][1][m] + red_ss[w][2][m] + red_ss[w][3][m];
    return rsqrtf(s / (float)p.K + sx.eps);
  };
  auto rsum = [&](int n, int m) -> float {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][n][m];
    return v;
  };

  if constexpr (PAIRS) {
    constexpr int NP = RF / 2;
    for (int o = tid; o < NP * 16 * p.M; o += NW * 64) {
      const int n = o & 15, pr = (o >> 4) % NP, m = o / (16 * NP);
      const float rs = row_scale(m);
      float a = rsum((2 * pr) * 16 + n, m) * rs, b = rsum((2 * pr + 1) * 16 + n, m) * rs;
      if constexpr (ACT == SL_ACT_SILU_MUL) {
        const int ocol = ((n0 >> 4) / 2 + pr) * 16 + n;
        if (ocol >= (p.N >> 1)) continue;
        if (bias) { a += to_f32(bias[n0 + (2 * pr) * 16 + n]); b += to_f32(bias[n0 + (2 * pr + 1) * 16 + n]); }
        store_out<T>(p, Cb, Rb, m, ocol, silu(a) * b);
      } else {  // ROPE_KV: global fragment gf; a head is 8 fragments (D = 128)
        const int gf = (n0 >> 4) + 2 * pr;
        if (gf * 16 >= p.N) continue;
        const int hh = gf >> 3, j = (gf & 7) >> 1;
        const int pos = sx.pos[m];
        if (hh < sx.nh + sx.nkv) {
          const int d = j * 16 + n;  // a = x[d], b = x[d + 64]
          const float c = sx.cos[(int64_t)pos * 64 + d], s = sx.sin[(int64_t)pos * 64 + d];
          const float o1 = a * c - b * s, o2 = b * c + a * s;
          if (hh < sx.nh) {
            T* dst = (T*)Cb + (int64_t)m * p.ldc + hh * 128 + d;
            dst[0] = from_f32<T>(o1); dst[64] = from_f32<T>(o2);
          } else {
            T* dst = (T*)sx.kc + (((int64_t)sx.seq[m] * sx.nkv + (hh - sx.nh)) * sx.max_ctx + pos) * 128 + d;
            dst[0] = from_f32<T>(o1); dst[64] = from_f32<T>(o2);
          }
        } else {  // v rows are in natural order: fragments 2j, 2j+1
          const int d = (gf & 7) * 16 + n;
          T* dst = (T*)sx.vc + (((int64_t)sx.seq[m] * sx.nkv + (hh - sx.nh - sx.nkv)) * sx.max_ctx + pos) * 128 + d;
          dst[0] = from_f32<T>(a); dst[16] = from_f32<T>(b);
        }
      }
    }
  } else {
    for (int o = tid; o < RB * p.M; o += NW * 64) {
      const int n = o % RB, m = o / RB;
      const int col = n0 + n;
      if (col >= p.N) continue;
      float v = rsum(n, m) * row_scale(m);
      if (bias) v += to_f32(bias[col]);
      if constexpr (ACT == SL_ACT_GELU) v = gelu_act<T>(v);
      store_out<T>(p, Cb, Rb, m, col, v);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// host dispatch
// ----------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------
// stream-K admission (sl_gemm_ex_args.sk_ws): returns the grid size, 0 = keep one block per tile
// ----------------------------------------------------------------------------------------------
static int sk_cu_count() {
  static int cus[SL_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SL_MAX_DEVICES) return 256;
  if (!cus[dev]) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus[dev];
}
extern "C" size_t sl_gemm_streamk_workspace_bytes(void) { return SK_FLAG_BYTES + (size_t)SK_WS_SLOTS * SK_SLOT_BYTES; }

static int sk_grid(const GemmP& p, int batch, int bk, size_t ws_bytes) {
  const int mode = sl_env().stream_k;          // SL_STREAM_K: 0 = never, 1 = rule (default), 2 = whenever the form allows
  if (!mode || p.ta || p.tw || p.grp || batch != 1 || p.K % bk || p.ln_mr || p.stats_out || p.amax_val || p.aux || p.N < 192 || p.post || p.colsum) return 0;
  const int cus = sk_cu_count() < 256 ? sk_cu_count() : 256;
  const int64_t tm = (p.M + XBM - 1) / XBM, tn = (p.N + XBN - 1) / XBN, nt = tm * tn, nkt = p.K / bk;
  if (nkt < 8 || tm * XBM * 4 > (int64_t)p.M * 5 + 4 * XBM) return 0;      // short reductions; rows padded by more than a quarter (+ one tile)
  int G = cus;
  if (nt * 4 < G) G = (int)(nt * 4);                                        // a tile's slabs go to at most ~4 blocks: the owner reads the others' sums serially
  if (nt * nkt < (int64_t)G * 4) return 0;
  if (ws_bytes < SK_FLAG_BYTES + (size_t)G * SK_SLOT_BYTES) return 0;
  if (mode == 2) return G;
  // Measured (profiles/r04_f_streamk.txt): cutting K across blocks gives up what the XCD tile patches buy — neighbouring blocks no longer
  // stream the SAME slabs, so every block pulls its 64 KiB per slab from beyond L2 (3 200 x 3 072 x 16 384: 156 tiles, all 256 CUs busy,
  // 2.6 GB through the Infinity Cache in 310 us = its bandwidth; 0.94 x the one-block-per-tile time) — and a tile's fp32 hand-off costs
  // 10-20 us.  It pays only where a few tiles sit under a long reduction (400 x 3 072 x 16 384: 24 tiles, 1.6 x).
  return (nt * 8 <= cus && nkt >= 128) ? G : 0;
}

// ----------------------------------------------------------------------------------------------
// Plain split-K for products of FEW tiles (round 5): the per-rank KD window (634 LLM rows) has N = 3 072 products of 36 big / 120
// small tiles under K = 3 072 ... 16 384 — a third of the chip busy (634 x 3 072 x 8 192: 425 TF/s against the vendor library's 729,
// x 16 384: 294 against 672; profiles/r05_b_gemm_vs_vendor_pad_rule.txt).  The reduction is cut into S equal runs of whole slabs and
// the S partial products run as ONE BATCHED launch of the ordinary tiled kernels (batch index = K run: A and W advance by K / S
// columns, fp32 partial tiles go to the caller's workspace), so blocks of a run still share their slabs in L2 — which the stream-K form
// above gives up — and a second launch adds the runs in run order and applies bias / residual / rounding: deterministic, no atomics.
// Costs S x M x N x 4 bytes written and read back, which is why it is for few tiles only.
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, int64_t slab, GemmP p) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (int64_t)p.M * p.N) return;
  const int64_t row = i / p.N;
  const int col = (int)(i - row * p.N);          // N % 4 == 0 (admission): the four values are one row's
  f32x4 acc = *(const f32x4*)(part + i);
  for (int z = 1; z < S; ++z) {
    const f32x4 v = *(const f32x4*)(part + (int64_t)z * slab + i);
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  float v4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v4[e] = acc[e] + (p.bias ? to_f32(((const T*)p.bias)[col + e]) : 0.f);
  if (p.post == SL_POST_SILU_MUL_BWD) {
    const int64_t o = row * p.post_ld + 32 * (col >> 4) + (col & 15);
    T* op = (T*)p.C + row * p.ldc + 32 * (col >> 4) + (col & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float dg, du;
      post_silu_bwd<T>(v4[e], to_f32(((const T*)p.post_in)[o + e]), to_f32(((const T*)p.post_in)[o + 16 + e]), dg, du);
      op[e] = from_f32<T>(dg); op[16 + e] = from_f32<T>(du);
    }
    return;
  }
  if (p.post) post_apply<T, 4>(p, row, col, v4);
#pragma unroll
  for (int e = 0; e < 4; ++e) store_out<T>(p, p.C, p.res, row, col + e, v4[e]);
}

// the ring form of the 128-tile kernel (gemm128.hip): whole-slab untransposed ungrouped products whose 128 x 128 tiles (x batch) give every CU
// at most one block, with enough slabs behind the ring's prologue
static bool ring_ok(const GemmP& p, int batch, int bk) {
  if (!sl_env().glds_ring || g_disable_glds || p.ta || p.tw || p.grp || p.K % bk || p.amax_val) return false;
  const int64_t t128 = (int64_t)((p.M + TBM - 1) / TBM) * ((p.N + TBN - 1) / TBN) * batch;
  // up to `ring_max_tiles` (default 256 = one block per CU; above it the blocks beyond the first round wait for a CU: SL_GLDS_RING_MAX_TILES)
  const int64_t cap = sl_env().ring_max_tiles > 0 ? sl_env().ring_max_tiles : sk_cu_count();
  return t128 <= cap && p.K / bk >= 8;
}

static int splitk_runs(const GemmP& p, int batch, int bk, size_t ws_bytes) {
  if (p.colsum) return 0;        // column sums are taken in the tile epilogues (one adder per wave and column), not in the reduce pass
  if (!sl_env().split_k || p.ta || p.tw || p.grp || batch != 1 || p.K % bk || p.ln_mr || p.stats_out || p.amax_val || p.aux || p.N < 128 || (p.N & 3)) return 0;
  const int64_t t128 = (int64_t)((p.M + TBM - 1) / TBM) * ((p.N + TBN - 1) / TBN);
  const int nkt = p.K / bk;
  if (t128 > 256 || nkt < 32) return 0;          // above half the 512 slots (two blocks per CU) the chip is busy enough; short reductions
  // slots: two blocks per CU of the two-stage kernel, ONE of the ring form (whose runs then also need no second round)
  // Measured with the ring form (profiles/r06_ag_gemm_vs_vendor_small.txt): one block per CU wins up to K = 8 192 (634 x 3 072 x 8 192: 725 against
  // 707 TF/s on two blocks per CU of the two-stage kernel), two blocks per CU from 192 slabs (x 16 384: 786 against 724)
  const bool ring = sl_env().glds_ring && !g_disable_glds && !p.amax_val;
  const int slots = sl_env().splitk_slots > 0 ? sl_env().splitk_slots : ((ring && nkt < 192) ? 256 : 512);
  int S = (int)(slots / t128);
  if (S > 8) S = 8;
  if (S > nkt / 12) S = nkt / 12;                // every run keeps >= 12 slabs (768 k) behind its prologue
  while (S > 1 && (nkt % S || (size_t)S * p.M * p.N * sizeof(float) > ws_bytes)) --S;
  // the reduce launch costs what ~25-30 slabs of the ring loop cost (634 x 3 072 x 3 072: 494 TF/s unsplit, 447 in two runs): cut only where the
  // runs save more slabs than that per block — or where the consumer sums the runs itself (deferred_splits: no reduce launch)
  if (ring && slots == 256 && !p.defer && S >= 2 && (int64_t)nkt * (S - 1) < 32 * (int64_t)S) return 0;
  return S >= 2 ? S : 0;
}

// K runs of 256-tile products that fill 50-66 % of ONE round (round 6).  The LLM data gradients of a 16-sample KD window are 3 200 x 3 072 under
// K = 3 072 / 5 120 / 16 384: 13 x 12 = 156 tiles on 256 CUs, 280 us at K = 16 384 whatever the kernel does inside a tile.  Cut into S = 3 runs of
// whole slabs (86 / 86 / 84 of the 256 — uneven: GemmP.krun) the 468 blocks make two rounds of a third of the length each: 2/3 of the time, and the
// three fp32 partial products stay in the workspace for the RMSNorm backward behind the product, which sums them while loading (deferred_splits);
// a caller without deferred_splits gets the reduce launch (the same bits: the unfused tape of the A/B tests), which gives most of the gain back.
// From 48 slabs per run up.  Measured: 16-sample KD window 88.7 -> 88.4 ms (profiles/r06_an_kd_windows.txt) — a third of what the round count
// promises: 118 MB of partial tiles are written and read back per product.
static int splitk256_runs(const GemmP& p, int batch, int bk, size_t ws_bytes, int* krun) {
  if (!sl_env().split_k256 || p.colsum || p.ta || p.tw || p.grp || batch != 1 || p.K % bk || p.ln_mr || p.stats_out || p.amax_val || p.aux || p.post ||
      p.bias || p.res || p.out_f32 || (p.N & 3) || g_disable_glds || sl_env().disable_t256 || !sl_env().t256_phased)
    return 0;
  const int64_t t256 = (int64_t)((p.M + XBM - 1) / XBM) * ((p.N + XBN - 1) / XBN);
  // 129 ... 200 tiles only: fewer tiles belong to the 128-tile rule below (more blocks, the ring form), and the admission must not depend on
  // whether the product carries a post-op (the fused and the unfused tape cut the same products the same way)
  if (t256 <= 128 || t256 > 200) return 0;
  const int nkt = p.K / bk;
  int best = 0;
  double best_cost = 0.7;                       // rounds(S) / S must fall below it
  for (int S = 2; S <= 4; ++S) {
    if (nkt / S < 48 || (size_t)S * p.M * p.N * sizeof(float) > ws_bytes) continue;
    const double cost = (double)((t256 * S + 255) / 256) / S;
    if (cost < best_cost) { best_cost = cost; best = S; }
  }
  if (best) *krun = (nkt + best - 1) / best;
  return best;
}

// both operands K-major (weight gradients): gemm_tiled_tt_kernel, with the reduction cut into S runs of whole slabs when the tiles alone
// leave CUs idle (two blocks per CU: 512 slots) and the caller supplied a workspace
template <typename T>
static bool tt_ok(const GemmP& p, int batch) {
  if (sizeof(T) != 2 || !sl_env().wgrad_tr || !p.ta || !p.tw || p.grp || p.aux || p.ln_mr || p.stats_out || p.amax_val) return false;
  // batched (round 6: the positional conv's weight gradient, 16 groups x 64 output rows per utterance, was 16 launches of the register-staged
  // loader at 84 TF/s): one K run, 64 output rows allowed (the tile's upper half reads zeros), no rider, 8-element aligned batch strides
  if (batch != 1) {
    if (!sl_env().tt_batched || p.colsum || p.bias || (p.M != 64 && p.M % TBM) || p.N % TBN || p.lda % 8 || p.ldw % 8 || p.K < 128 || (p.sA & 7) || (p.sW & 7) ||
        p.wx || p.cx || p.rx)
      return false;
    return true;
  }
  return p.M % TBM == 0 && p.N % TBN == 0 && p.lda % 8 == 0 && p.ldw % 8 == 0 && p.K >= 128 && p.wx == 0 && p.cx == 0 && p.rx == 0;
}

template <typename T>
static int launch_tt(GemmP& p, hipStream_t st, void* sk_ws, size_t sk_ws_bytes, int batch = 1) {
  p.tiles_m = (p.M + TBM - 1) / TBM;
  p.tiles_n = p.N / TBN;
  const int nt = p.tiles_m * p.tiles_n, nkt = (p.K + 63) / 64;
  if (batch > 1) return sl_gemm_tt_kernel_launch(p, nt, 1, nkt, st, batch);
  int S = 1;
  const size_t ws = sk_ws && sk_ws_bytes > SK_FLAG_BYTES ? sk_ws_bytes - SK_FLAG_BYTES : 0;
  // K runs fill 512 block slots (two blocks per CU of the two-stage kernel); launches that end at <= 256 blocks take the ring form (gemm_tt.hip).
  // Filling only 256 slots so that every product rides the ring was measured mixed at 7 984 tokens (4 096 x 1 024: 746 against 675 TF/s,
  // 3 072 x 1 024: 571 against 622, 1 024 x 4 096: 749 against 824; profiles/r06_ak_wgrad_ring.txt) — SL_SPLITK_SLOTS=256 selects it.
  const int slots = sl_env().splitk_slots > 0 ? sl_env().splitk_slots : 512;
  if (ws && nt < slots && !(p.N & 3)) {
    S = slots / nt;
    const int smax = sl_env().tt_max_splits > 0 ? sl_env().tt_max_splits : 8;
    if (S > smax) S = smax;
    if (S > nkt / 12) S = nkt / 12;
    while (S > 1 && (size_t)S * p.M * p.N * sizeof(float) > ws) --S;
    if (S < 1) S = 1;
  }
  const int spr = (nkt + S - 1) / S;
  S = (nkt + spr - 1) / spr;
  if (S == 1) return sl_gemm_tt_kernel_launch(p, nt, 1, spr, st);
  float* part = (float*)((unsigned char*)sk_ws + SK_FLAG_BYTES);
  GemmP q = p;
  q.C = part; q.ldc = p.N; q.sC = (int64_t)p.M * p.N; q.out_f32 = 1;
  q.bias = nullptr; q.sBias = 0; q.res = nullptr; q.ldr = 0; q.sR = 0; q.res_f32 = 0;
  SL_TRY(sl_gemm_tt_kernel_launch(q, nt, S, spr, st));
  const int64_t vecs = ((int64_t)p.M * p.N + 3) / 4;
  hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, st, part, S, (int64_t)p.M * p.N, p);
  SL_CHECK_LAUNCH("splitk_reduce");
  return 0;
}

template <typename T, int ACT>
static int launch_tiled(GemmP& p, int batch, hipStream_t st, void* sk_ws = nullptr, size_t sk_ws_bytes = 0) {
  constexpr int BK_ = TROWB / (int)sizeof(T);
  if constexpr (ACT == SL_ACT_NONE && sizeof(T) == 2) {
    if (tt_ok<T>(p, batch)) return launch_tt<T>(p, st, sk_ws, sk_ws_bytes, batch);
  }
  if constexpr (ACT == SL_ACT_NONE && sizeof(T) == 2) {
    if (sk_ws) {
      int krun = 0;
      const int S = splitk256_runs(p, batch, BK_, sk_ws_bytes > SK_FLAG_BYTES ? sk_ws_bytes - SK_FLAG_BYTES : 0, &krun);
      if (S) {
        GemmP q = p;
        q.krun = krun; q.sA = 0; q.sW = 0;
        q.C = (unsigned char*)sk_ws + SK_FLAG_BYTES; q.ldc = p.N; q.sC = (int64_t)p.M * p.N; q.out_f32 = 1;
        q.defer = nullptr;
        q.tiles_m = (p.M + XBM - 1) / XBM;
        q.tiles_n = (p.N + XBN - 1) / XBN;
        SL_TRY((sl_gemm256_launch<T, SL_ACT_NONE>(q, SL_T256_PHASED, dim3(q.tiles_m * q.tiles_n, S), nullptr, st)));
        if (p.defer) { *p.defer = S; return 0; }      // the consumer sums the runs (speechllm.h deferred_splits)
        const int64_t vecs = ((int64_t)p.M * p.N + 3) / 4;
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, st, (const float*)q.C, S, (int64_t)p.M * p.N, p);
        SL_CHECK_LAUNCH("splitk_reduce");
        return 0;
      }
    }
  }
  if constexpr (ACT == SL_ACT_NONE) {
    if (sk_ws) {
      const int S = splitk_runs(p, batch, BK_, sk_ws_bytes > SK_FLAG_BYTES ? sk_ws_bytes - SK_FLAG_BYTES : 0);
      if (S) {
        float* part = (float*)((unsigned char*)sk_ws + SK_FLAG_BYTES);      // behind the stream-K flags (which stay zero)
        GemmP q = p;
        q.K = p.K / S; q.sA = q.K; q.sW = q.K;                                // run z reads columns [z K/S, (z+1) K/S) of A and W
        q.C = part; q.ldc = p.N; q.sC = (int64_t)p.M * p.N; q.out_f32 = 1;
        q.bias = nullptr; q.sBias = 0; q.res = nullptr; q.ldr = 0; q.sR = 0; q.res_f32 = 0;
        q.post = 0; q.drop_thr24 = 0; q.post_in = nullptr; q.colsum = nullptr;      // the reduce pass applies them
        q.defer = nullptr;
        SL_TRY((launch_tiled<T, SL_ACT_NONE>(q, S, st)));
        if (p.defer && !p.bias && !p.res && !p.post && !p.colsum && !p.out_f32) { *p.defer = S; return 0; }      // the consumer sums the runs (speechllm.h deferred_splits)
        const int64_t vecs = ((int64_t)p.M * p.N + 3) / 4;
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, st, part, S, (int64_t)p.M * p.N, p);
        SL_CHECK_LAUNCH("splitk_reduce");
        return 0;
      }
    }
  }
  if (sk_ws) {
    const int G = sk_grid(p, batch, BK_, sk_ws_bytes);
    if (G) {
      p.tiles_m = (p.M + XBM - 1) / XBM;
      p.tiles_n = (p.N + XBN - 1) / XBN;
      if constexpr (sizeof(T) == 2 && ACT != SL_ACT_SILU_MUL) {
        const bool al = !(p.N & 7) && !(p.ldc & 7) && !((uintptr_t)p.C & 15) && (!p.res || (!(p.ldr & 7) && !((uintptr_t)p.res & 15)));
        if (al && !p.out_f32 && !p.res_f32 && !p.direct_epi && !sl_env().no_swap_epilogue) {
          return sl_gemm256_launch<T, ACT>(p, SL_T256_SK_SW, dim3(G), sk_ws, st);
        }
      }
      return sl_gemm256_launch<T, ACT>(p, SL_T256_SK, dim3(G), sk_ws, st);
    }
  }
  // large products: 256^2 tiles once they alone give every CU >= 2 tiles (ragged batches: sized by the largest group)
  if (!p.ta && !p.tw && p.K % BK_ == 0 && !p.grp_ext && g_disable_glds == 0 && !sl_env().disable_t256 && !((p.post || p.colsum) && !sl_env().t256_phased)) {
    const int famM = (p.grp || batch != 1) ? p.M : sl_family_rows(p.M);      // the tile family follows the pinned rows (common.h sl_family_rows)
    const int64_t t256 = (int64_t)((famM + XBM - 1) / XBM) * ((p.N + XBN - 1) / XBN) * batch;
    const int64_t min_tiles = sl_env().t256_min_tiles;   // tuning switches
    const int min_k = sl_env().t256_min_k;
    // rows padded to 256 vs to 128: short (grouped) products such as the 123-row projector would half-fill the big tile
    const int64_t m128 = (int64_t)((famM + TBM - 1) / TBM) * TBM, m256 = (int64_t)((famM + XBM - 1) / XBM) * XBM;
    // Mid-size products (KD windows: M = 2-8 k rows): neither tile count fills the chip evenly, so the choice is made on whole
    // rounds of tiles — 256 slots of one 256^2 tile per CU against 512 slots of 128^2 tiles (two blocks per CU, each at ~0.85 of
    // the big tile's rate per flop): 5072 x 3072 is 240 big tiles = one round (1.09 PF/s; 960 small ones = two rounds, 0.99),
    // 3200 x 5120 is 260 big tiles = two rounds, the second almost empty (0.63 PF/s; small tiles 0.98).  tools/sweep_t256.py.
    bool by_rounds = false;
    if (t256 < min_tiles && !p.grp && batch == 1) {
      const int64_t t128 = (int64_t)((famM + TBM - 1) / TBM) * ((p.N + TBN - 1) / TBN);
      const int64_t r256 = (t256 + 255) / 256, r128 = (t128 + 511) / 512;
      by_rounds = (double)r256 * (XBM * XBN) * 0.85 < (double)r128 * 2.0 * (TBM * TBN);
    }
    // rows padded to 256: the whole-rounds comparison above already prices the padding (it counts TILES), so a product it picks may pad
    // freely — 634 x 16 384 x 3 072 (the per-rank KD window) is 192 big tiles = one round against 640 small ones = two; only products
    // admitted by their tile count alone (grouped / batched: the 123-row projector) keep the 1/8 bound
    const bool pad_ok = m256 <= m128 + m128 / 8 || (by_rounds && sl_env().t256_by_rounds_pad);
    if ((t256 >= min_tiles || by_rounds) && p.N >= 192 && p.K >= min_k && pad_ok) {   // 1024: with the row epilogue the big tile also wins at K = 1024..1536 (+10..20 %)
      p.tiles_m = (p.M + XBM - 1) / XBM;
      p.tiles_n = (p.N + XBN - 1) / XBN;
      const int phased = sl_env().t256_phased;   // 0: round-3 one-barrier-per-slab loop (A/B)
      if constexpr (sizeof(T) == 2 && ACT != SL_ACT_SILU_MUL) {
        // swapped-operand form (register epilogue, 16-byte stores): plain bf16 stores on 8-element aligned rows, one of the forms
        // {bias}, {bias, residual}, {LayerNorm fold}, {bias, residual, row statistics}, {bias, pre-activation copy}
        const bool al = !(p.N & 7) && !(p.ldc & 7) && !(p.sC & 7) && !((uintptr_t)p.C & 15) &&
                        (!p.res || (!(p.ldr & 7) && !(p.sR & 7) && !((uintptr_t)p.res & 15)));
        const bool form = p.aux ? (!p.ln_mr && !p.stats_out && !p.res && !((uintptr_t)p.aux & 15))        // {bias, pre-activation copy}: the training forward's FFN1
                                : !p.ln_mr ? (!p.stats_out || p.res) : (!p.res && !p.stats_out);
        // post-ops on the swapped-operand kernels: dropout in {bias, pre-activation copy, GELU} or {bias, residual}; GELU' (+ colsum_out) and SwiGLU' on
        // the plain product; operand rows 16-byte aligned.  Everything else with a post-op takes the LDS-turned rows epilogue.
        bool post_ok = !p.post && !p.colsum;      // (the phased kernel only: the one-barrier-per-slab A/B form keeps the plain epilogues)
        if (p.post == SL_POST_DROPOUT) post_ok = !p.colsum && !p.ln_mr && !p.stats_out && (ACT == SL_ACT_GELU ? (p.aux && !p.res) : (p.res && !p.aux));
        if (p.post == SL_POST_GELU_BWD || p.post == SL_POST_SILU_MUL_BWD)
          post_ok = ACT == SL_ACT_NONE && !(p.post_ld & 7) && !((uintptr_t)p.post_in & 15) && (p.post == SL_POST_GELU_BWD || !p.colsum);
        if (al && form && !p.grp && !p.out_f32 && !p.res_f32 && !p.amax_val && !p.direct_epi && !sl_env().no_swap_epilogue && post_ok && (phased || (!p.post && !p.colsum))) {
#ifdef SL_GEMM_DEBUG
          if constexpr (ACT == SL_ACT_NONE) {       // instrumented / knocked-out builds (tools/gemm_stamps.py, tools/gemm_knockout.py), never in the product .so
            const int ko = sl_env().gemm_ko;
            if (phased && !p.ln_mr && (sl_env().gemm_stamp_ptr || ko)) {
              p.stamp = (uint32_t*)(uintptr_t)sl_env().gemm_stamp_ptr;
              const dim3 g(p.tiles_m * p.tiles_n, batch);
              return sl_gemm256_launch<T, ACT>(p, SL_T256_DBG + (p.stamp ? 8 : (ko >= 1 && ko <= 4 ? ko : 6)), g, nullptr, st);
            }
          }
#endif
          return sl_gemm256_launch<T, ACT>(p, phased ? SL_T256_PHASED_SW : SL_T256_PLAIN_SW, dim3(p.tiles_m * p.tiles_n, batch), nullptr, st);
        }
      }
      return sl_gemm256_launch<T, ACT>(p, phased ? SL_T256_PHASED : SL_T256_PLAIN, dim3(p.tiles_m * p.tiles_n, batch), nullptr, st);
    }
  }
  p.tiles_m = (p.M + TBM - 1) / TBM;
  p.tiles_n = (p.N + TBN - 1) / TBN;
  dim3 grid(p.tiles_m * p.tiles_n, batch);
  constexpr int BK = TROWB / (int)sizeof(T);
  if ((p.post || p.colsum) && !(!p.ta && !p.tw && p.K % BK == 0 && (!p.grp_ext || p.grp_kslab) && !g_disable_glds)) {
    sl_set_error("sl_gemm_ex: post_op / colsum_out need whole 128-byte K slabs (K %% %d == 0) and the LDS-DMA kernels (SL_DISABLE_GLDS unset)", BK);
    return SL_ERR_UNSUPPORTED;
  }
  // at most one block per CU: the ring form keeps NS - 1 slabs of DMA in flight per block (gemm128.hip; same bits as the two-stage kernel)
  if constexpr (sizeof(T) == 2) {
    if (ring_ok(p, batch, BK)) return sl_gemm128_ring_launch<T, ACT>(p, sl_env().glds_ring, grid, st);
  }
  if (!p.ta && !p.tw && p.K % BK == 0 && !p.grp_ext && g_disable_glds == 2)
    hipLaunchKernelGGL((gemm_tiled_glds_kernel<T, ACT, false>), grid, dim3(256), 0, st, p);
  else if (!p.ta && !p.tw && p.K % BK == 0 && (!p.grp_ext || p.grp_kslab) && !g_disable_glds) {   // per-group K: only the register path handles K tails (groups_ext = 2: the caller vouches for whole slabs)
    if constexpr (sizeof(T) == 2) {
      if (sl_env().glds_dmab && !p.grp_ext) {
        hipLaunchKernelGGL((gemm_tiled_glds_kernel<T, ACT, true, true>), grid, dim3(256), 0, st, p);
        SL_CHECK_LAUNCH("gemm_tiled");
        return 0;
      }
    }
    hipLaunchKernelGGL((gemm_tiled_glds_kernel<T, ACT, true>), grid, dim3(256), 0, st, p);
  }
  else
    hipLaunchKernelGGL((gemm_tiled_kernel<T, ACT>), grid, dim3(256), 0, st, p);
  SL_CHECK_LAUNCH("gemm_tiled");
  return 0;
}

template <typename T, int MT, int ACT, int RF, int NW, int U, bool PACKED, bool KCONT = false>
static int launch_skinny_cfg(GemmP& p, const SkinnyX& sx, int batch, hipStream_t st) {
  dim3 grid((p.N + 16 * RF - 1) / (16 * RF), batch);
  hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, ACT, RF, NW, U, PACKED, KCONT>), grid, dim3(NW * 64), 0, st, p, sx);
  SL_CHECK_LAUNCH("gemm_skinny");
  return 0;
}

// Structure per shape, from the sweeps in tools/tune_skinny.hip (MI355X, bf16, packed weights):
//   M <= 16:  gate/up, lm_head (>= 1024 fragments): 4 fragments x 4 waves x 4 steps in flight   (5.3 / 6.2 TB/s)
//             qkv (320 fragments):                   2 fragments x 8 waves x 4 steps            (2.9 TB/s)
//             N = hidden (o, down; 192 fragments):   1 fragment x 16 waves x 2 steps            (2.7 / 3.3 TB/s)
//             (2 x 16 x 4 and contiguous K slices looked 5-12 % better in the bare sweep but were slower inside the
//              real kernel with its fused-norm / pair epilogues: decode step 2.19 vs 1.98 ms)
//   M 17..64: activations are re-read per fragment from L2, so fewer fragments per wave and wide K splits win.
template <typename T, int MT, int ACT, bool PACKED>
static int launch_skinny_mt(GemmP& p, const SkinnyX& sx, int batch, hipStream_t st) {
  constexpr bool PAIRS = (ACT == SL_ACT_SILU_MUL || ACT == SL_ACT_ROPE_KV);
  const int nfrag = (p.N + 15) / 16 * batch;
  if constexpr (MT == 1) {
    if constexpr (PACKED && sizeof(T) == 2 && !PAIRS) {
      // N = hidden (o, down) at M <= 8: three or four 64-byte steps of loads in flight per wave, chosen so the wave's share
      // of K divides evenly (o: 6 steps = 2 x 3, down: 16 = 4 x 4): 1.575 -> 1.49 ms per decode step at M = 1, +3 % at M = 16
      // where the x fragments crowd the loads.  The same sweep over the qkv and gate/up structures (2x16x3, 2x8x6, 2x8x3;
      // 4x4x6, 4x4x3, 2x8x3, 2x8x6, 4x8x3) moved nothing (profiles/r04_r_skinny_small_m.txt).  SL_SKINNY_ALT=1: old structure.
      if (nfrag < 256 && sl_family_rows(p.M) <= 8 && !(sl_env().skinny_alt & 1)) {
        const int per_wave = p.K / 32 / 16;
        if (per_wave % 3 == 0) return launch_skinny_cfg<T, MT, ACT, 1, 16, 3, PACKED>(p, sx, batch, st);
        return launch_skinny_cfg<T, MT, ACT, 1, 16, 4, PACKED>(p, sx, batch, st);
      }
    }
    if (nfrag >= 1024) return launch_skinny_cfg<T, MT, ACT, 4, 4, 4, PACKED>(p, sx, batch, st);
    if (nfrag >= 256 || PAIRS) return launch_skinny_cfg<T, MT, ACT, 2, 8, 4, PACKED>(p, sx, batch, st);
    if constexpr (!PAIRS) return launch_skinny_cfg<T, MT, ACT, 1, 16, 2, PACKED>(p, sx, batch, st);
  } else if constexpr (MT == 2) {
    if (nfrag >= 1024) return launch_skinny_cfg<T, MT, ACT, 2, 8, 4, PACKED>(p, sx, batch, st);
    if (nfrag >= 256 || PAIRS) return launch_skinny_cfg<T, MT, ACT, 2, 16, 2, PACKED>(p, sx, batch, st);
    if constexpr (!PAIRS) return launch_skinny_cfg<T, MT, ACT, 1, 16, 2, PACKED>(p, sx, batch, st);
  } else {
    if (nfrag >= 1024) return launch_skinny_cfg<T, MT, ACT, 4, 8, 2, PACKED>(p, sx, batch, st);
    if (nfrag >= 256 || PAIRS) return launch_skinny_cfg<T, MT, ACT, 2, 16, 2, PACKED>(p, sx, batch, st);
    if constexpr (!PAIRS) return launch_skinny_cfg<T, MT, ACT, 1, 16, 2, PACKED>(p, sx, batch, st);
  }
  return 0;
}

template <typename T, int ACT>
static int launch_skinny(GemmP& p, const SkinnyX& sx, int batch, bool packed, hipStream_t st) {
  if (packed) {
    if constexpr (ACT == SL_ACT_GELU) {
      sl_set_error("sl_gemm: packed weights are not built with the GELU epilogue");
      return SL_ERR_UNSUPPORTED;
    } else {
      if (sl_family_rows(p.M) <= 16) return launch_skinny_mt<T, 1, ACT, true>(p, sx, batch, st);
      if (sl_family_rows(p.M) <= 32) return launch_skinny_mt<T, 2, ACT, true>(p, sx, batch, st);
      return launch_skinny_mt<T, 4, ACT, true>(p, sx, batch, st);
    }
  }
  if constexpr (ACT == SL_ACT_ROPE_KV) {
    sl_set_error("sl_gemm: the ROPE_KV epilogue needs packed weights");
    return SL_ERR_UNSUPPORTED;
  } else {
    if (sl_family_rows(p.M) <= 16) return launch_skinny_mt<T, 1, ACT, false>(p, sx, batch, st);
    if (sl_family_rows(p.M) <= 32) return launch_skinny_mt<T, 2, ACT, false>(p, sx, batch, st);
    return launch_skinny_mt<T, 4, ACT, false>(p, sx, batch, st);
  }
}

template <typename T>
static int gemm_typed(const sl_gemm_args* a, GemmP& p, const SkinnyX& sx, hipStream_t st, void* sk_ws, size_t sk_ws_bytes) {
  const bool skinny = sl_family_rows(a->M) <= 64 && !p.ta && !p.tw && !p.aux && !p.res_f32 && !p.grp && !p.ln_mr && !p.stats_out && !p.post && !p.colsum;  // backward features and the LayerNorm fold live in the tiled kernel
  const bool packed = a->w_layout == SL_W_PACKED;
  if (!skinny && (packed || a->act == SL_ACT_ROPE_KV || sx.fuse_rms)) {
    sl_set_error("sl_gemm: packed weights / ROPE_KV / fused RMSNorm need plain operands (no transposes / groups), M=%d", a->M);
    return SL_ERR_UNSUPPORTED;
  }
  switch (a->act) {
    case SL_ACT_NONE: return skinny ? launch_skinny<T, SL_ACT_NONE>(p, sx, a->batch, packed, st) : launch_tiled<T, SL_ACT_NONE>(p, a->batch, st, sk_ws, sk_ws_bytes);
    case SL_ACT_GELU: return skinny ? launch_skinny<T, SL_ACT_GELU>(p, sx, a->batch, packed, st) : launch_tiled<T, SL_ACT_GELU>(p, a->batch, st, sk_ws, sk_ws_bytes);
    case SL_ACT_SILU_MUL: return skinny ? launch_skinny<T, SL_ACT_SILU_MUL>(p, sx, a->batch, packed, st) : launch_tiled<T, SL_ACT_SILU_MUL>(p, a->batch, st, sk_ws, sk_ws_bytes);
    case SL_ACT_ROPE_KV: return launch_skinny<T, SL_ACT_ROPE_KV>(p, sx, a->batch, packed, st);
  }
  sl_set_error("sl_gemm: unknown act %d", a->act);
  return SL_ERR_ARG;
}

// true when a plain (M, N, K) product of this dtype is served by one of the LDS-DMA tiled kernels, whose rows epilogue carries the
// LayerNorm fold (ln_* / stats_out)
// shapes for which the training tapes may hand a product its post-ops (train_tape.hip fuse_ok): the tiled kernels' row range, whole K slabs
bool sl_gemm_post_ok(int64_t M, int N, int K, int dtype) {
  return sl_family_rows((int)(M > 0x7fffffff ? 0x7fffffff : M)) > 64 && M > 64 && N % 16 == 0 && K % (dtype == SL_F32 ? 32 : 64) == 0 && sl_env().disable_glds == 0;
}

bool sl_gemm_rows_epilogue_ok(int M, int N, int K, int dtype) {
  // any row count: a product that carries ln_* / stats_out is kept on the tiled kernels even below 65 rows (gemm_typed), so that the fold is a
  // property of the MODEL — a short utterance encoded alone takes the same epilogues, hence the same bits, as inside a batch
  if (dtype != SL_BF16 || M <= 0 || (N & 63) || sl_env().disable_glds != 0 || sl_env().direct_epilogue != 0) return false;
  return K % (TROWB / 2) == 0;
}

int sl_gemm_impl(const sl_gemm_args* a, const sl_gemm_fused* fx, const sl_gemm_ex_args* ex, hipStream_t st) {
  SL_CHECK_ARG(a != nullptr, "sl_gemm: null args");
  g_disable_glds = sl_env().disable_glds;   // 1: register staging, 2: glds with compiler-visible LDS reads
  SL_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0 && a->batch > 0, "sl_gemm: bad shape M=%d N=%d K=%d batch=%d", a->M, a->N, a->K, a->batch);
  SL_CHECK_ARG(a->dtype == SL_F32 || a->dtype == SL_BF16, "sl_gemm: bad dtype %d", a->dtype);
  const int vec = a->dtype == SL_F32 ? 4 : 8;
  // a K-contiguous (non-transposed) operand is read in 16-byte chunks along K; transposed ones along the output index
  const bool both_t = ex && ex->trans_a && ex->trans_w;
  SL_CHECK_ARG(both_t || a->K % vec == 0, "sl_gemm: K=%d must be a multiple of %d", a->K, vec);
  SL_CHECK_ARG(a->lda % vec == 0 && a->strideA % vec == 0, "sl_gemm: lda/strideA must keep rows 16-byte aligned");
  SL_CHECK_ARG(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0, "sl_gemm: A and W must be 16-byte aligned");
  if (a->w_layout == SL_W_PACKED) {
    SL_CHECK_ARG(a->K % (4 * vec) == 0, "sl_gemm: packed weights need K %% %d == 0", 4 * vec);
  } else {
    SL_CHECK_ARG(a->w_layout == SL_W_ROWMAJOR, "sl_gemm: unknown w_layout %d", a->w_layout);
    SL_CHECK_ARG(a->ldw % vec == 0 && a->strideW % vec == 0, "sl_gemm: ldw/strideW must keep rows 16-byte aligned");
  }
  if (a->act == SL_ACT_SILU_MUL) SL_CHECK_ARG(a->N % 32 == 0, "sl_gemm: SILU_MUL needs N %% 32 == 0 (16-row gate/up blocks)");
  GemmP p;
  p.A = a->A; p.lda = a->lda; p.sA = a->strideA;
  p.W = a->W; p.ldw = a->ldw; p.sW = a->strideW;
  p.C = a->C; p.ldc = a->ldc; p.sC = a->strideC;
  p.bias = a->bias; p.sBias = a->strideBias;
  p.res = a->residual; p.ldr = a->ldr; p.sR = a->strideR;
  p.M = a->M; p.N = a->N; p.K = a->K; p.out_f32 = a->out_f32;
  p.tiles_m = p.tiles_n = 0;
  p.ta = p.tw = 0; p.aux = nullptr; p.res_f32 = 0; p.grp = nullptr; p.w_mod = 1; p.cx = p.rx = p.wx = 0; p.grp_ext = 0; p.grp_kslab = 0;
  p.post = 0; p.drop_thr24 = 0; p.drop_scale = 1.f; p.drop_seed = 0; p.drop_ld = 0; p.post_in = nullptr; p.post_ld = 0; p.colsum = nullptr; p.defer = nullptr;
  p.krun = 0; p.stamp = nullptr; p.amax_val = nullptr; p.amax_idx = nullptr; p.ln_mr = nullptr; p.ln_u = nullptr; p.ln_c = nullptr; p.stats_out = nullptr;
  const int direct_epi = sl_env().direct_epilogue;
  p.direct_epi = direct_epi;
  const int gm_env = sl_env().gemm_gm;
  p.gm = gm_env;
  if (ex) {
    p.ta = ex->trans_a; p.tw = ex->trans_w; p.aux = ex->aux_out; p.res_f32 = ex->residual_f32;
    p.grp = ex->groups; p.w_mod = ex->w_mod > 0 ? ex->w_mod : 1;
    if (ex->deferred_splits) { *ex->deferred_splits = 0; p.defer = ex->deferred_splits; }
    p.grp_ext = ex->groups && ex->groups_ext;
    p.grp_kslab = ex->groups && ex->groups_ext == 2;
    SL_CHECK_ARG(!(p.ta || p.tw || p.aux) || a->act != SL_ACT_SILU_MUL, "sl_gemm_ex: transposed operands / aux_out are not combined with SILU_MUL");
    SL_CHECK_ARG(!p.res_f32 || a->out_f32, "sl_gemm_ex: residual_f32 needs out_f32");
    SL_CHECK_ARG(!(p.ta || p.tw) || a->w_layout == SL_W_ROWMAJOR, "sl_gemm_ex: transposed operands need row-major storage");
    if (ex->ln_mr || ex->ln_u || ex->ln_c || ex->stats_out) {
      // both sides of the LayerNorm fold live in the rows epilogue of the LDS-DMA tiled kernels (bf16, 4-column vectors)
      SL_CHECK_ARG(sl_gemm_rows_epilogue_ok(a->M, a->N, a->K, a->dtype) && a->batch == 1 && !ex->groups && !ex->trans_a && !ex->trans_w && !ex->aux_out &&
                       a->act != SL_ACT_SILU_MUL && a->w_layout == SL_W_ROWMAJOR && a->ldc % 4 == 0 && ((uintptr_t)a->C & 7) == 0 &&
                       (!a->residual || (a->ldr % 4 == 0 && ((uintptr_t)a->residual & 7) == 0)),
                   "sl_gemm_ex: ln_* / stats_out need a plain bf16 row-major product the LDS-DMA tiled kernels take (M=%d N=%d K=%d), 4-element aligned rows", a->M, a->N, a->K);
      SL_CHECK_ARG((!ex->ln_mr && !ex->ln_u && !ex->ln_c) || (ex->ln_mr && ex->ln_u && ex->ln_c && !a->bias),
                   "sl_gemm_ex: the LayerNorm fold needs ln_mr, ln_u and ln_c together (the bias is inside ln_c)");
      p.ln_mr = ex->ln_mr; p.ln_u = ex->ln_u; p.ln_c = ex->ln_c; p.stats_out = ex->stats_out;
    }
    if (ex->colsum_out && ex->trans_a && ex->trans_w && !ex->post_op) {
      // the bias gradient riding on the weight-gradient product: colsum_out (M) += sum over K of A-stored[k][m] — the token-major kernel only
      p.colsum = ex->colsum_out;
      SL_CHECK_ARG(a->dtype == SL_BF16 && tt_ok<bf16_t>(p, a->batch) && a->act == SL_ACT_NONE,
                   "sl_gemm_ex: colsum_out with trans_a + trans_w needs the token-major weight-gradient kernel (bf16, M and N multiples of 128, K >= 128, 8-element aligned rows; sl_gemm_tt_ok)");
    } else if (ex->post_op || ex->colsum_out) {
      SL_CHECK_ARG(ex->post_op >= SL_POST_NONE && ex->post_op <= SL_POST_SILU_MUL_BWD, "sl_gemm_ex: unknown post_op %d", ex->post_op);
      SL_CHECK_ARG(!ex->trans_a && !ex->trans_w && !ex->groups && !ex->ln_mr && !ex->ln_u && !ex->ln_c && !ex->stats_out && !ex->amax_val && !ex->amax_idx &&
                       a->batch == 1 && a->w_layout == SL_W_ROWMAJOR && a->act != SL_ACT_SILU_MUL && a->act != SL_ACT_ROPE_KV && a->M > 64,
                   "sl_gemm_ex: post_op / colsum_out need one plain row-major product on the tiled kernels (M > 64, no transposes / groups / ln_* / stats_out / amax_*)");
      SL_CHECK_ARG(ex->drop_p >= 0.f && ex->drop_p < 1.f, "sl_gemm_ex: drop_p %f outside [0, 1)", (double)ex->drop_p);
      if (ex->post_op == SL_POST_DROPOUT) SL_CHECK_ARG(ex->drop_p > 0.f && ex->drop_ld >= a->N, "sl_gemm_ex: SL_POST_DROPOUT needs drop_p > 0 and drop_ld >= N");
      if (ex->post_op == SL_POST_GELU_BWD)
        SL_CHECK_ARG(ex->post_in && ex->post_ld >= a->N && !a->residual && !a->bias && a->act == SL_ACT_NONE && !ex->aux_out && !a->out_f32 && (ex->drop_p == 0.f || ex->drop_ld >= a->N),
                     "sl_gemm_ex: SL_POST_GELU_BWD needs post_in (M, N), the plain epilogue (no bias / residual / act / aux_out) and an output in the storage type");
      if (ex->post_op == SL_POST_SILU_MUL_BWD)
        SL_CHECK_ARG(ex->post_in && ex->post_ld >= 2 * (int64_t)a->N && a->ldc >= 2 * (int64_t)a->N && a->N % 16 == 0 && !a->residual && !a->bias && a->act == SL_ACT_NONE &&
                         !ex->aux_out && !a->out_f32 && !ex->colsum_out && ex->drop_p == 0.f,
                     "sl_gemm_ex: SL_POST_SILU_MUL_BWD needs post_in = gu (M, 2 N), ldc >= 2 N, N %% 16 == 0 and the plain epilogue");
      p.post = ex->post_op; p.post_in = ex->post_in; p.post_ld = ex->post_ld; p.colsum = ex->colsum_out;
      if (ex->drop_p > 0.f && ex->post_op != SL_POST_NONE) {
        p.drop_thr24 = (uint32_t)((double)ex->drop_p * 16777216.0);
        p.drop_scale = 1.0f / (1.0f - ex->drop_p);
        p.drop_seed = ex->drop_seed; p.drop_ld = ex->drop_ld;
      }
    }
    if (ex->amax_val || ex->amax_idx) {
      SL_CHECK_ARG(ex->amax_val && ex->amax_idx && a->act == SL_ACT_NONE && a->batch == 1 && !ex->groups && !ex->trans_a && !ex->trans_w && !ex->aux_out &&
                       !a->residual && sl_family_rows(a->M) > 64 && a->w_layout == SL_W_ROWMAJOR,
                   "sl_gemm_ex: amax_val / amax_idx (fused row-wise top-1) need both pointers, the plain epilogue without residual, one "
                   "un-grouped row-major product and M > 64 (M=%d act=%d batch=%d)", a->M, a->act, a->batch);
      p.amax_val = ex->amax_val; p.amax_idx = ex->amax_idx;
    }
  }
  SL_CHECK_ARG(p.amax_val || a->C, "sl_gemm: null C");
  if (sl_env().gemm_log)     // SL_GEMM_LOG=1: one line per product on stderr (tools/kd_gemm_shapes.py turns a KD window's lines into a per-shape table)
    fprintf(stderr, "SLGEMM M=%d N=%d K=%d batch=%d act=%d ta=%d tw=%d res=%d resf32=%d outf32=%d aux=%d grp=%d bias=%d packed=%d dt=%d\n", a->M, a->N, a->K, a->batch, a->act,
            p.ta, p.tw, a->residual != nullptr, p.res_f32, a->out_f32, p.aux != nullptr, p.grp ? (p.grp_ext ? 2 : 1) : 0, a->bias != nullptr, a->w_layout == SL_W_PACKED, a->dtype);
  SkinnyX sx;
  memset(&sx, 0, sizeof(sx));
  if (fx) {
    sx.fuse_rms = fx->fuse_rms; sx.eps = fx->rms_eps;
    sx.cos = fx->rope_cos; sx.sin = fx->rope_sin; sx.pos = fx->tok_pos; sx.seq = fx->tok_seq;
    sx.kc = fx->k_cache; sx.vc = fx->v_cache; sx.nh = fx->n_heads; sx.nkv = fx->n_kv_heads; sx.max_ctx = fx->max_ctx;
    sx.rstd_in = fx->rstd_in; sx.rstd_out = fx->rstd_out; sx.norm_out = fx->norm_out; sx.norm_gain = fx->norm_gain;
    SL_CHECK_ARG((!fx->norm_out && !fx->norm_gain) || (fx->norm_out && fx->norm_gain && fx->rstd_out && !a->out_f32), "sl_gemm: norm_out needs norm_gain, rstd_out and an output in the storage type");
  }
  if (a->act == SL_ACT_ROPE_KV) {
    SL_CHECK_ARG(fx && fx->rope_cos && fx->rope_sin && fx->tok_pos && fx->tok_seq && fx->k_cache && fx->v_cache,
                 "sl_gemm: ROPE_KV epilogue needs the sl_gemm_fused tables");
    SL_CHECK_ARG(a->N == (fx->n_heads + 2 * fx->n_kv_heads) * 128 && a->batch == 1, "sl_gemm: ROPE_KV expects N = (n_heads + 2 n_kv) * 128");
  }
  // packed weights with more than g_stream_min_m rows: LDS-staged streaming kernel (gemm_stream.hip)
  if (a->w_layout == SL_W_PACKED && sl_family_rows(a->M) > stream_min_m() && a->batch == 1 && !ex && a->act != SL_ACT_GELU &&
      a->K % (a->dtype == SL_F32 ? 32 : 64) == 0)
    return sl_gemm_stream_launch(p, sx, a->dtype, a->act, fx ? fx->split_ws : nullptr, fx ? fx->split_ws_bytes : 0, st);
  SL_CHECK_ARG(a->w_layout != SL_W_PACKED || sl_family_rows(a->M) <= 64, "sl_gemm: packed weights with M=%d > 64 need batch 1 and K %% 64 == 0", a->M);
  SL_CHECK_ARG(!sx.rstd_in && !sx.rstd_out && !sx.norm_out, "sl_gemm: rstd_in / rstd_out / norm_out are features of the streaming path (M > %d rows, packed weights)", stream_min_m());
  void* sk_ws = ex ? ex->sk_ws : nullptr;
  const size_t sk_ws_bytes = ex ? ex->sk_ws_bytes : 0;
  SL_CHECK_ARG(!sk_ws || ((uintptr_t)sk_ws & 15) == 0, "sl_gemm_ex: sk_ws must be 16-byte aligned");
  if (a->dtype == SL_F32) return gemm_typed<float>(a, p, sx, st, sk_ws, sk_ws_bytes);
  return gemm_typed<bf16_t>(a, p, sx, st, sk_ws, sk_ws_bytes);
}

extern "C" int32_t sl_gemm_split_count(int32_t M, int32_t N, int32_t K, int32_t dtype) {
  M = sl_family_rows(M);
  if (M <= stream_min_m() || M <= 0 || N <= 0 || K <= 0 || K % (dtype == SL_F32 ? 32 : 64) != 0) return 1;
  return sl_gemm_stream_splits(M, N, K, dtype);
}

extern "C" size_t sl_gemm_split_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t dtype) {
  if (sl_family_rows(M) <= stream_min_m() || M <= 0 || N <= 0 || K <= 0) return 0;
  return sl_gemm_stream_ws_bytes(M, N, K, dtype);
}

extern "C" int32_t sl_gemm_ln_fold_ok(int32_t M, int32_t N, int32_t K, int32_t dtype) { return sl_gemm_rows_epilogue_ok(M, N, K, dtype) ? 1 : 0; }

extern "C" int sl_gemm(const sl_gemm_args* a, sl_stream stream) { return sl_gemm_impl(a, nullptr, nullptr, (hipStream_t)stream); }

extern "C" int sl_gemm_ex(const sl_gemm_args* a, const sl_gemm_ex_args* ex, sl_stream stream) {
  SL_CHECK_ARG(ex != nullptr, "sl_gemm_ex: null ex args");
  return sl_gemm_impl(a, nullptr, ex, (hipStream_t)stream);
}

extern "C" int sl_gemm_fused_decode(const sl_gemm_args* a, const sl_gemm_fused* fx, sl_stream stream) {
  SL_CHECK_ARG(fx != nullptr, "sl_gemm_fused_decode: null fused args");
  return sl_gemm_impl(a, fx, nullptr, (hipStream_t)stream);
}
