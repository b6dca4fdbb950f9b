// gemm_tt.hip — the weight-gradient product on token-major operands (gemm_tiled_tt_kernel) and its launch entry, split from gemm.hip so that
// the three GEMM translation units compile side by side.
#include <atomic>
#include <type_traits>
#include "common.h"
#include "gemm_internal.h"
#include "gemm_epilogue.h"

// ----------------------------------------------------------------------------------------------
// Both operands K-MAJOR (the weight-gradient product dW (No, Ni) += dY^T X with dY (tokens, No) and X (tokens, Ni) row-major, reduction
// over the token rows): 128 x 128 tile, LDS-DMA staging of [64 token rows][128 columns] slabs (256-byte rows), fragments by
// ds_read_b64_tr_b16 — a 16-lane group gathers 4 token rows x 16 columns and lane i receives column i's four values, i.e. four
// consecutive k of output row i; two reads make the 16-byte MFMA operand.  No transposed copies of dY / X are made (sl_transpose_pad
// read + wrote each of them once per product: 4 % of a KD window).  bf16 only.
//   LDS image: row = token row of the slab (256 B = eight 32-byte slots of 16 columns); slot s of row r sits at physical slot
//   s ^ f(r), f = (r & 3) | (((r >> 3) & 1) << 2): the 16 row segments one wave-wide read touches (rows 8g + {0..3} (+4), g = 0..3) fall on
//   every 32-byte bank group exactly twice — the rate of a 512-byte read.  As in the kernels above the swizzle is applied to the
//   per-lane SOURCE address of the DMA.
//   K (token) tail: rows past K are fetched from a 16-byte zero constant.  blockIdx.y = K run (split-K: fp32 partial tiles to
//   C + run * sC, summed by splitk_reduce_kernel), every run a whole number of slabs.
// ----------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

#define SL_LDS_RD_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_tt_t;
template <int N>
__device__ __forceinline__ void lds_wait_tr16(u32x2_tt_t (&a)[8], u32x2_tt_t (&b)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%16)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]),
                 "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
               : "n"(N));
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void gemm_tiled_tt_kernel(GemmP p, int slabs_per_run) {
  using T = bf16_t;
  constexpr int BK = 64;                       // token rows per slab
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][BK * 256];   // [buf][A|W]: 64 rows x 256 B = 16 KiB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, q = lane >> 4;
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int qn = nt >> 3, rn = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  const int bm = bid % p.tiles_m, bn = bid / p.tiles_m;
  // blockIdx.y = K run (split-K); blockIdx.z = batch index (round 6: the positional conv's weight gradient is 16 groups of 64 output rows per
  // utterance — batched launches have one K run, and `z` below indexes C / residual by the batch strides)
  const int zb = blockIdx.z;
  const int z = gridDim.z > 1 ? zb : (int)blockIdx.y;
  const int nkt_all = (p.K + BK - 1) / BK;
  const int kt0 = (gridDim.z > 1 ? 0 : (int)blockIdx.y) * slabs_per_run;
  int kt1 = kt0 + slabs_per_run;
  kt1 = kt1 < nkt_all ? kt1 : nkt_all;
  const T* A = (const T*)p.A + (int64_t)zb * p.sA;
  const T* W = (const T*)p.W + (int64_t)zb * p.sW;

  // LDS chunk c = tid + 256 i sits at (row c >> 4, physical chunk c & 15) and must hold logical chunk ((pc >> 1) ^ f(row)) << 1 | (pc & 1)
  const T* ga[4];
  const T* gw[4];
  int grow[4];
  bool a_in[4];       // this chunk's 8 output rows (columns of dY) lie inside M (M = 64: the upper half of the tile reads the zero constant)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 4, pc = c & 15;
    const int f = (row & 3) | (((row >> 3) & 1) << 2);
    const int lc = (((pc >> 1) ^ f) << 1) | (pc & 1);
    grow[i] = row;
    a_in[i] = bm * TBM + lc * 8 < p.M;
    ga[i] = A + (int64_t)row * p.lda + bm * TBM + lc * 8;
    gw[i] = W + (int64_t)row * p.ldw + bn * TBN + lc * 8;
  }
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;
  const T* zero = (const T*)g_zero16;

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // The bias gradient rides along (sl_gemm_ex_args.colsum_out with both operands transposed): db[m] = sum over tokens of dY[token][m] is the
  // product of the A fragments with a B fragment of ones.  The waves of the first column tile that hold the tile's left half (wn = 0)
  // take it: 8 more MFMAs per slab beside their 32, for one in 2 tiles_n waves — instead of a sl_colsum launch that re-reads dY.
  const bool do_cs = p.colsum != nullptr && bn == 0 && wn == 0;
  const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  f32x4 accb[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int kt, int buf) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool in = k0 + grow[i] < p.K;
      const T* sa = (in && a_in[i]) ? ga[i] + (int64_t)k0 * p.lda : zero;
      const T* sw = in ? gw[i] + (int64_t)k0 * p.ldw : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)sa, (lds_ptr_t)(&smem[buf][0][i * 4096 + wave_lds]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)sw, (lds_ptr_t)(&smem[buf][1][i * 4096 + wave_lds]), 16, 0, 0);
    }
  };

  // fragment addresses: lane (li = r, g = q) asks for token row 8 g + (li >> 2) (+ 4 for the upper half, + 32 for the second k-step: immediates),
  // piece li & 3 of the 32-byte slot of its 16 columns; slot (4 wm + m) ^ f, f = (li >> 2) | ((g & 1) << 2)
  const int qq = r >> 2, pp = r & 3;
  const int f = qq | ((q & 1) << 2);
  uint32_t aa[4], ab[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    aa[m] = (uint32_t)((8 * q + qq) * 256 + (((4 * wm + m) ^ f) << 5) + pp * 8);
    ab[m] = (uint32_t)((8 * q + qq) * 256 + (((4 * wn + m) ^ f) << 5) + pp * 8);
  }
  const uint32_t sb0 = (uint32_t)(uintptr_t)(lds_ptr_t)(&smem[0][0][0]);

  if (kt0 < kt1) {
    issue(kt0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) issue(kt + 1, buf ^ 1);
    const uint32_t sa = sb0 + (uint32_t)buf * (2 * BK * 256), sw = sa + BK * 256;
    u32x2_tt_t a0[8], b0[8], a1[8], b1[8];
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(a0[2 * m], sa + aa[m], 0); SL_LDS_RD_TR(a0[2 * m + 1], sa + aa[m], 1024); }
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(b0[2 * m], sw + ab[m], 0); SL_LDS_RD_TR(b0[2 * m + 1], sw + ab[m], 1024); }
    lds_wait_tr16<0>(a0, b0);
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(a1[2 * m], sa + aa[m], 8192); SL_LDS_RD_TR(a1[2 * m + 1], sa + aa[m], 9216); }
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(b1[2 * m], sw + ab[m], 8192); SL_LDS_RD_TR(b1[2 * m + 1], sw + ab[m], 9216); }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        MMA<T>::step(acc[m][n], make_uint4(a0[2 * m].x, a0[2 * m].y, a0[2 * m + 1].x, a0[2 * m + 1].y), make_uint4(b0[2 * n].x, b0[2 * n].y, b0[2 * n + 1].x, b0[2 * n + 1].y));
    if (do_cs) {
#pragma unroll
      for (int m = 0; m < 4; ++m) MMA<T>::step(accb[m], make_uint4(a0[2 * m].x, a0[2 * m].y, a0[2 * m + 1].x, a0[2 * m + 1].y), ones);
    }
    __builtin_amdgcn_sched_barrier(0);
    lds_wait_tr16<0>(a1, b1);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        MMA<T>::step(acc[m][n], make_uint4(a1[2 * m].x, a1[2 * m].y, a1[2 * m + 1].x, a1[2 * m + 1].y), make_uint4(b1[2 * n].x, b1[2 * n].y, b1[2 * n + 1].x, b1[2 * n + 1].y));
    if (do_cs) {
#pragma unroll
      for (int m = 0; m < 4; ++m) MMA<T>::step(accb[m], make_uint4(a1[2 * m].x, a1[2 * m].y, a1[2 * m + 1].x, a1[2 * m + 1].y), ones);
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (do_cs && r == 0) {       // every column of accb holds the row sums; lane (r = 0, q) has rows 4 q + i of each fragment.  One adder per K run.
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = bm * TBM + wm * 64 + m * 16 + 4 * q + i;
        if (row < p.M) atomicAdd(p.colsum + row, accb[m][i]);
      }
  }
  GemmP pe = p;
  pe.colsum = nullptr;         // (the epilogues' colsum_out is the stored values' column sum: not this product's meaning of the field)
  if (!p.direct_epi && tile_epilogue_rows<T, ACT, 4>(pe, acc, bm * TBM + wm * 64, bn * TBN + wn * 64, lane, z, 0, (float*)&smem[0][0][0] + wave * 4096)) return;
  tile_epilogue<T, ACT>(pe, acc, bm, bn, wm, wn, q, r, z, 0);
}


// ----------------------------------------------------------------------------------------------
// Ring form of the kernel above for launches of at most ONE block per CU (round 6; the NT twin is gemm128.hip, where the reasoning and the
// measurements live): same tile, LDS image, fragment order and epilogue — the same bits for the same K runs — but four 32 KiB stages in dynamic
// LDS, the fragment reads software-pipelined across the barrier (k-step 1 under the MFMAs of k-step 0, the next slab's k-step 0 under those of
// k-step 1), MFMAs as asm statements and a slab's eight DMA requests spread between them (its W half inside k-step 0 of iteration nk - 2, its
// A half inside k-step 1 of iteration nk - 3).  32 ds_read_b64_tr_b16 are in flight at the waits and lgkmcnt counts to 15: "the older 16 have
// landed" is written lgkmcnt(15) (one read more than needed).  Runs shorter than four slabs take a plain loop.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ void tt_mfma(f32x4& acc, const u32x2_tt_t& alo, const u32x2_tt_t& ahi, const u32x2_tt_t& blo, const u32x2_tt_t& bhi) {
  const u32x4_t a = {alo.x, alo.y, ahi.x, ahi.y}, b = {blo.x, blo.y, bhi.x, bhi.y};
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "memory");
}
// B operand in a register tuple the caller built with VALU moves (the fragment of ones): the hazard recognizer does not know the asm statement
// is a matrix instruction and leaves out the VALU-write -> MFMA-read wait states (the first rider MFMA read the PREVIOUS tenant of the tuple:
// a quarter of the bias gradient wrong) — they are part of the statement
__device__ __forceinline__ void tt_mfma_b4(f32x4& acc, const u32x2_tt_t& alo, const u32x2_tt_t& ahi, const u32x4_t& b) {
  const u32x4_t a = {alo.x, alo.y, ahi.x, ahi.y};
  asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "memory");
}
template <int N> __device__ __forceinline__ void tt_vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int ACT>
__global__ __launch_bounds__(256, 1) void gemm_tiled_tt_ring_kernel(GemmP p, int slabs_per_run) {
  using T = bf16_t;
  constexpr int BK = 64, NS = 4, HALF = BK * 256, STAGE = 2 * HALF;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, q = lane >> 4;
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int qn = nt >> 3, rn = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  const int bm = bid % p.tiles_m, bn = bid / p.tiles_m;
  const int z = blockIdx.y;
  const int nkt_all = (p.K + BK - 1) / BK;
  const int kt0 = z * slabs_per_run;
  int kt1 = kt0 + slabs_per_run;
  kt1 = kt1 < nkt_all ? kt1 : nkt_all;
  const int n = kt1 > kt0 ? kt1 - kt0 : 0;
  const T* A = (const T*)p.A;
  const T* W = (const T*)p.W;

  const T* ga[4];
  const T* gw[4];
  int grow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 4, pc = c & 15;
    const int f = (row & 3) | (((row >> 3) & 1) << 2);
    const int lc = (((pc >> 1) ^ f) << 1) | (pc & 1);
    grow[i] = row;
    ga[i] = A + (int64_t)row * p.lda + bm * TBM + lc * 8;
    gw[i] = W + (int64_t)row * p.ldw + bn * TBN + lc * 8;
  }
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;
  const T* zero = (const T*)g_zero16;

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int nn = 0; nn < 4; ++nn) acc[m][nn] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_cs = p.colsum != nullptr && bn == 0 && wn == 0;      // the bias gradient rides along (see the kernel above)
  u32x4_t ones = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
  asm volatile("" : "+v"(ones));        // one tuple, written here once (not re-materialised in front of each use)
  f32x4 accb[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) accb[m] = f32x4{0.f, 0.f, 0.f, 0.f};

  // request i (0..3) of the A / W half of run-relative slab s (off = the slab's row offset in elements, uniform; rows past K read the zero constant)
  auto dmaA = [&](int s, int i, int64_t off, int k0) {
    const T* src = ga[i] + off;
    src = k0 + grow[i] < p.K ? src : zero;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + (s % NS) * STAGE + i * 4096 + wave_lds), 16, 0, 0);
  };
  auto dmaW = [&](int s, int i, int64_t off, int k0) {
    const T* src = gw[i] + off;
    src = k0 + grow[i] < p.K ? src : zero;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)(smem + (s % NS) * STAGE + HALF + i * 4096 + wave_lds), 16, 0, 0);
  };
  auto slabA = [&](int s) { const int k0 = (kt0 + s) * BK; const int64_t off = (int64_t)k0 * p.lda;
#pragma unroll
    for (int i = 0; i < 4; ++i) dmaA(s, i, off, k0); };
  auto slabW = [&](int s) { const int k0 = (kt0 + s) * BK; const int64_t off = (int64_t)k0 * p.ldw;
#pragma unroll
    for (int i = 0; i < 4; ++i) dmaW(s, i, off, k0); };

  const int qq = r >> 2, pp = r & 3;
  const int f = qq | ((q & 1) << 2);
  uint32_t aa[4], ab[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    aa[m] = (uint32_t)((8 * q + qq) * 256 + (((4 * wm + m) ^ f) << 5) + pp * 8);
    ab[m] = (uint32_t)((8 * q + qq) * 256 + (((4 * wn + m) ^ f) << 5) + pp * 8);
  }
  const uint32_t sb0 = (uint32_t)(uintptr_t)(lds_ptr_t)smem;

  u32x2_tt_t a0[8], b0[8], a1[8], b1[8];
  auto rd0 = [&](int s) {
    const uint32_t sa = sb0 + (uint32_t)((s % NS) * STAGE), sw = sa + HALF;
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(a0[2 * m], sa + aa[m], 0); SL_LDS_RD_TR(a0[2 * m + 1], sa + aa[m], 1024); }
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(b0[2 * m], sw + ab[m], 0); SL_LDS_RD_TR(b0[2 * m + 1], sw + ab[m], 1024); }
  };
  auto rd1 = [&](int s) {
    const uint32_t sa = sb0 + (uint32_t)((s % NS) * STAGE), sw = sa + HALF;
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(a1[2 * m], sa + aa[m], 8192); SL_LDS_RD_TR(a1[2 * m + 1], sa + aa[m], 9216); }
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(b1[2 * m], sw + ab[m], 8192); SL_LDS_RD_TR(b1[2 * m + 1], sw + ab[m], 9216); }
  };
  // 16 MFMAs of one k-step; DMAW / DMAA: the W (A) half of run-relative slab `s` goes between them, one request per four MFMAs
  auto mma = [&](u32x2_tt_t (&a)[8], u32x2_tt_t (&b)[8], auto pending, auto dma_w, auto dma_a, int s) {
    const int k0 = (kt0 + s) * BK;
    const int64_t offa = (int64_t)k0 * p.lda, offw = (int64_t)k0 * p.ldw;
    if constexpr (decltype(pending)::value) lds_wait_tr16<15>(a, b);      // the 16 reads issued behind these may stay in flight
    else lds_wait_tr16<0>(a, b);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        tt_mfma(acc[m][nn], a[2 * m], a[2 * m + 1], b[2 * nn], b[2 * nn + 1]);
        if constexpr (decltype(dma_w)::value) { if (nn == 1) dmaW(s, m, offw, k0); }
        if constexpr (decltype(dma_a)::value) { if (nn == 1) dmaA(s, m, offa, k0); }
      }
    }
    if (do_cs) {
#pragma unroll
      for (int m = 0; m < 4; ++m) tt_mfma_b4(accb[m], a[2 * m], a[2 * m + 1], ones);
    }
  };
  using yes = std::integral_constant<bool, true>;
  using no = std::integral_constant<bool, false>;

  if (n >= 4) {
    slabA(0); slabW(0); slabA(1); slabW(1); slabA(2);
    tt_vm_wait<12>();                  // slab 0 has landed: slab 1 and A(2) are younger
    __builtin_amdgcn_s_barrier();
    rd0(0);
    int s = 0;
    for (; s + 3 < n; ++s) {
      rd1(s);
      mma(a0, b0, yes{}, yes{}, no{}, s + 2);
      tt_vm_wait<8>();                 // slab s + 1 has landed: A(s + 2), W(s + 2) are younger
      __builtin_amdgcn_s_barrier();    // behind it every wave has retired its reads of slab s - 1
      rd0(s + 1);
      mma(a1, b1, yes{}, no{}, yes{}, s + 3);
    }
    rd1(s);                            // s = n - 3
    mma(a0, b0, yes{}, yes{}, no{}, s + 2);
    tt_vm_wait<8>();
    __builtin_amdgcn_s_barrier();
    rd0(s + 1);
    mma(a1, b1, yes{}, no{}, no{}, 0);
    ++s;
    rd1(s);                            // s = n - 2
    mma(a0, b0, yes{}, no{}, no{}, 0);
    tt_vm_wait<0>();
    __builtin_amdgcn_s_barrier();
    rd0(s + 1);
    mma(a1, b1, yes{}, no{}, no{}, 0);
    ++s;
    rd1(s);
    mma(a0, b0, yes{}, no{}, no{}, 0);
    mma(a1, b1, no{}, no{}, no{}, 0);
  } else if (n > 0) {                  // short runs: every slab requested up front (n <= 3 stages), then the slabs one after the other
    for (int s = 0; s < n; ++s) { slabA(s); slabW(s); }
    tt_vm_wait<0>();
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < n; ++s) {
      rd0(s);
      rd1(s);
      mma(a0, b0, yes{}, no{}, no{}, 0);
      mma(a1, b1, no{}, no{}, no{}, 0);
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // XDL write -> VALU read wait states the compiler cannot see behind asm MFMAs
  __syncthreads();
  if (do_cs && r == 0) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = bm * TBM + wm * 64 + m * 16 + 4 * q + i;
        if (row < p.M) atomicAdd(p.colsum + row, accb[m][i]);
      }
  }
  GemmP pe = p;
  pe.colsum = nullptr;
  if (!p.direct_epi && tile_epilogue_rows<T, ACT, 4>(pe, acc, bm * TBM + wm * 64, bn * TBN + wn * 64, lane, z, 0, (float*)smem + wave * 4096)) return;
  tile_epilogue<T, ACT>(pe, acc, bm, bn, wm, wn, q, r, z, 0);
}

// launch of the kernel above (gemm.hip launch_tt decides the K runs and issues the reduce pass): grid (tiles, runs)
int sl_gemm_tt_kernel_launch(const GemmP& p, int nt, int S, int slabs_per_run, hipStream_t st, int batch) {
  if (batch > 1) {        // batched (one K run): the two-stage kernel, batch index on blockIdx.z
    hipLaunchKernelGGL((gemm_tiled_tt_kernel<SL_ACT_NONE>), dim3(nt, 1, batch), dim3(256), 0, st, p, slabs_per_run);
    SL_CHECK_LAUNCH("gemm_tiled_tt(batch)");
    return 0;
  }
  if (sl_env().glds_ring && sl_env().tt_ring && (int64_t)nt * S <= 256 && slabs_per_run >= 8) {       // at most one block per CU: the ring form
    constexpr int LDS_BYTES = 4 * 2 * 64 * 256;
    static std::atomic<uint64_t> attr_set{0};   // one bit per device: the opt-in to > 64 KiB of dynamic LDS is per device
    int devid = 0;
    SL_HIP(hipGetDevice(&devid));
    if (devid < 0 || devid >= 64 || !((attr_set.load(std::memory_order_relaxed) >> devid) & 1)) {
      SL_HIP(hipFuncSetAttribute((const void*)gemm_tiled_tt_ring_kernel<SL_ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
      if (devid >= 0 && devid < 64) attr_set.fetch_or(1ull << devid, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((gemm_tiled_tt_ring_kernel<SL_ACT_NONE>), dim3(nt, S), dim3(256), LDS_BYTES, st, p, slabs_per_run);
    SL_CHECK_LAUNCH("gemm_tiled_tt_ring");
    return 0;
  }
  hipLaunchKernelGGL((gemm_tiled_tt_kernel<SL_ACT_NONE>), dim3(nt, S), dim3(256), 0, st, p, slabs_per_run);
  SL_CHECK_LAUNCH("gemm_tiled_tt");
  return 0;
}
