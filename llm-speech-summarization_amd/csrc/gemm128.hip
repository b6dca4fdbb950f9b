// gemm128.hip — the 128 x 128 tile with a RING of K slabs for products of at most one block per CU (round 6).
//
// Why a second 128-tile kernel.  gemm_tiled_glds_kernel (gemm.hip) keeps two 32 KiB stages and closes every slab with
// s_waitcnt vmcnt(0) + barrier: one slab of DMA in flight per block, which two resident blocks per CU turn into two.  The
// per-rank KD window (2 samples: 634 LLM rows, 998 encoder rows) is made of products with FEWER tiles than CUs — 634 x 5 120 x 3 072
// is 200 tiles, 634 x 3 072 x 3 072 is 120 — so every CU holds ONE block, and that block waits out the whole L2 / fabric latency of
// each slab before it multiplies: 35 us where the bytes the 200 blocks pull (314 MB at ~74 GB/s per CU) take 21
// (profiles/r05_d_gemm_vs_vendor_split_k.txt: 0.61-0.84 x the vendor library on these rows).
// This kernel: same tile, wave grid (2 x 2 waves of 64 x 64), swizzled LDS image, fragment order and epilogues — so the SAME BITS as
// gemm_tiled_glds_kernel — but NS stages (dynamic LDS, NS x 32 KiB, one block per CU) with NS - 1 slabs requested ahead and a COUNTED
// vmcnt: iteration kt waits only for slab kt (8 DMA instructions per slab and thread: vmcnt(8 (NS - 2)) in the steady loop), passes
// ONE barrier — behind it every wave has retired its fragment reads of slab kt - 1, whose slot the DMA of slab kt + NS - 1 may now
// overwrite — requests that slab and multiplies slab kt.  The DMA queue never drains inside the loop.
// That loop is the PIPE = false form (SL_GLDS_RING=104), +6-7 % on the two-stage kernel.  The default (PIPE, SPLIT) adds, measured step by step on
// 634 x 5 120 x 3 072 (598 TF/s on the two-stage kernel, vendor 756-778; profiles/r06_af / ag / aj):
//   * fragment reads software-pipelined across the barrier, MFMAs as asm statements (below)                          -> 692
//   * the slab's eight DMA requests BETWEEN the MFMAs instead of in a burst behind the barrier (SL_GLDS_RING=204)      -> 757-778
//   * those requests spread over two 16-MFMA phases, one per four MFMAs (A half / W half in different iterations)      -> 802-857
// which is the L2 -> CU fetch bound of a 128^2 tile (32 KiB per slab at ~31 B/clk).  Three stages (SL_GLDS_RING=3): 623.
// launch_tiled (gemm.hip) takes it for whole-slab, untransposed, ungrouped bf16 products of <= 256 tiles (SL_GLDS_RING=0: off; more tiles run in two
// rounds and lose to the two-stage kernel, profiles/r06_ap).
#include <atomic>
#include <type_traits>
#include "common.h"
#include "gemm_internal.h"
#include "gemm_epilogue.h"

// The pipelined loop issues its MFMAs as asm statements: with the builtin, the fragments that live across the loop's back edge made the
// register allocator rotate the accumulators through copies (92 v_accvgpr_* moves per slab beside 32 MFMAs); the asm form accumulates in
// place, in the order written.  The compiler cannot see these as matrix instructions: mfma_asm_drain() covers the XDL-write -> VALU-read
// wait states in front of the epilogue.
template <typename T>
__device__ __forceinline__ void mfma_asm(f32x4& acc, const u32x4_t& a, const u32x4_t& b) {
  static_assert(std::is_same<T, bf16_t>::value, "the ring form is built for bf16");
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <typename T>      // same, as a compiler-level memory fence: the DMA requests placed between two of these stay between them
__device__ __forceinline__ void mfma_asm_fence(f32x4& acc, const u32x4_t& a, const u32x4_t& b) {
  static_assert(std::is_same<T, bf16_t>::value, "the ring form is built for bf16");
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "memory");
}
__device__ __forceinline__ void mfma_asm_drain() { asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); }

template <int N> __device__ __forceinline__ void vm_wait_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <typename T, int ACT, int NS, bool PIPE, bool SPLIT = false, bool DMA_BETWEEN = true>
__global__ __launch_bounds__(256, 1) void gemm_tiled_ring_kernel(GemmP p) {
  static_assert(NS >= 3 && NS <= 4, "8 (NS - 2) DMA instructions stay in flight; NS x 32 KiB of LDS");
  static_assert(!SPLIT || (PIPE && NS == 4), "the split request schedule is written for the pipelined four-stage ring");
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int BK = TROWB / (int)sizeof(T);
  constexpr int HALF = TBM * TROWB;            // 16 KiB: one operand's slab
  constexpr int STAGE = 2 * HALF;
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
  int bm, bn;   // 8-row patches of tiles per XCD (as gemm_tiled_glds_kernel)
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

  // LDS chunk c = tid + 256 i sits at (row c>>3, physical chunk c&7) and holds logical chunk (c&7)^(row&7)
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
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = p.K / BK;
  auto issue = [&](int kt) {
    const int k0 = kt * BK;
    unsigned char* dst = smem + (kt % NS) * STAGE + wave_lds;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[i] + k0), (lds_ptr_t)(dst + i * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(gw[i] + k0), (lds_ptr_t)(dst + HALF + i * 4096), 16, 0, 0);
    }
  };

  __builtin_amdgcn_sched_barrier(0);
  if constexpr (SPLIT) {      // slabs 0, 1 and the A half of slab 2 (nkt >= 4: launch_tiled admits the ring from 8 slabs)
    issue(0); issue(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[i] + 2 * BK), (lds_ptr_t)(smem + 2 * STAGE + wave_lds + i * 4096), 16, 0, 0);
  } else {
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
      if (s < nkt) issue(s);
  }

  const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const uint32_t x0 = (uint32_t)((q ^ (r & 7)) << 4), x1 = (uint32_t)(((4 + q) ^ (r & 7)) << 4);
  const uint32_t ra0 = sb + (uint32_t)((wm * 64 + r) * TROWB), rb0 = sb + (uint32_t)(HALF + (wn * 64 + r) * TROWB);

  if constexpr (PIPE) {
    // One wave per SIMD: nothing else covers a wave's fragment reads, so the loop is software-pipelined ACROSS the barrier — the reads of
    // k-step 1 run under the MFMAs of k-step 0, and the reads of the NEXT slab's k-step 0 (behind the wait + barrier that admit that slab)
    // under the MFMAs of k-step 1.  What stays exposed per slab is the barrier skew and the issue of 8 DMA + 8 read instructions.
    u32x4_t a0[4], b0[4], a1[4], b1[4];
    auto rd0 = [&](int kt) {       // k-step 0 of slab kt -> a0, b0
      const uint32_t so = (uint32_t)((kt % NS) * STAGE), ra = ra0 + so, rb = rb0 + so;
      SL_LDS_RD(a0[0], ra + x0, 0); SL_LDS_RD(a0[1], ra + x0, 2048); SL_LDS_RD(a0[2], ra + x0, 4096); SL_LDS_RD(a0[3], ra + x0, 6144);
      SL_LDS_RD(b0[0], rb + x0, 0); SL_LDS_RD(b0[1], rb + x0, 2048); SL_LDS_RD(b0[2], rb + x0, 4096); SL_LDS_RD(b0[3], rb + x0, 6144);
    };
    auto rd1 = [&](int kt) {       // k-step 1 of slab kt -> a1, b1
      const uint32_t so = (uint32_t)((kt % NS) * STAGE), ra = ra0 + so, rb = rb0 + so;
      SL_LDS_RD(a1[0], ra + x1, 0); SL_LDS_RD(a1[1], ra + x1, 2048); SL_LDS_RD(a1[2], ra + x1, 4096); SL_LDS_RD(a1[3], ra + x1, 6144);
      SL_LDS_RD(b1[0], rb + x1, 0); SL_LDS_RD(b1[1], rb + x1, 2048); SL_LDS_RD(b1[2], rb + x1, 4096); SL_LDS_RD(b1[3], rb + x1, 6144);
    };
    auto mma0 = [&]() {
      lds_wait8<8>(a0[0], a0[1], a0[2], a0[3], b0[0], b0[1], b0[2], b0[3]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) mfma_asm<T>(acc[m][n], a0[m], b0[n]);
    };
    auto mma1 = [&](auto last) {
      if constexpr (decltype(last)::value) lds_wait8<0>(a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3]);
      else lds_wait8<8>(a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) mfma_asm<T>(acc[m][n], a1[m], b1[n]);
    };
    // k-step 1 with the eight DMA requests of slab `nk` BETWEEN its MFMAs (one per two): a request costs the issuing wave ~60-185 cycles
    // (guide: LDS-DMA piece issue cost), which the matrix pipe otherwise sits out idle behind the barrier
    auto mma1_dma = [&](int nk) {
      lds_wait8<8>(a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3]);
      const int k0 = nk * BK;
      unsigned char* dst = smem + (nk % NS) * STAGE + wave_lds;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          mfma_asm_fence<T>(acc[m][n], a1[m], b1[n]);
          if (n == 1) __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[m] + k0), (lds_ptr_t)(dst + m * 4096), 16, 0, 0);
          if (n == 3) __builtin_amdgcn_global_load_lds((glb_ptr_t)(gw[m] + k0), (lds_ptr_t)(dst + HALF + m * 4096), 16, 0, 0);
        }
    };
    // SPLIT: a slab's eight requests spread over TWO 16-MFMA phases, one per four MFMAs — its A half between the MFMAs of k-step 1 of iteration
    // nk - 3 (behind that iteration's barrier every wave has retired its reads of slab nk - 4, the slot's last tenant), its W half between the
    // MFMAs of k-step 0 of iteration nk - 2.  Behind W(kt + 1) the requests A(kt + 2), W(kt + 2) are younger: vmcnt(8) admits slab kt + 1.
    auto mma0_dmaW = [&](int nk) {
      lds_wait8<8>(a0[0], a0[1], a0[2], a0[3], b0[0], b0[1], b0[2], b0[3]);
      const int k0 = nk * BK;
      unsigned char* dst = smem + (nk % NS) * STAGE + HALF + wave_lds;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          mfma_asm_fence<T>(acc[m][n], a0[m], b0[n]);
          if (n == 1) __builtin_amdgcn_global_load_lds((glb_ptr_t)(gw[m] + k0), (lds_ptr_t)(dst + m * 4096), 16, 0, 0);
        }
    };
    auto mma1_dmaA = [&](int nk) {
      lds_wait8<8>(a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3]);
      const int k0 = nk * BK;
      unsigned char* dst = smem + (nk % NS) * STAGE + wave_lds;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          mfma_asm_fence<T>(acc[m][n], a1[m], b1[n]);
          if (n == 1) __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[m] + k0), (lds_ptr_t)(dst + m * 4096), 16, 0, 0);
        }
    };
    using yes = std::integral_constant<bool, true>;
    using no = std::integral_constant<bool, false>;
    if constexpr (SPLIT) {
      vm_wait_n<12>();                 // slab 0 has landed: slab 1 and A(2) are younger
      __builtin_amdgcn_s_barrier();
      rd0(0);
      int kt = 0;
      for (; kt + 3 < nkt; ++kt) {
        rd1(kt);
        mma0_dmaW(kt + 2);
        vm_wait_n<8>();
        __builtin_amdgcn_s_barrier();      // behind it: slab kt + 1 is visible, and every wave has retired its reads of slab kt - 1
        rd0(kt + 1);
        mma1_dmaA(kt + 3);
      }
      rd1(kt);                             // kt = nkt - 3: the last W half; A(nkt - 1) and it stay younger than slab nkt - 2
      mma0_dmaW(kt + 2);
      vm_wait_n<8>();
      __builtin_amdgcn_s_barrier();
      rd0(kt + 1);
      mma1(no{});
      ++kt;
      rd1(kt);                             // kt = nkt - 2
      mma0();
      vm_wait_n<0>();
      __builtin_amdgcn_s_barrier();
      rd0(kt + 1);
      mma1(no{});
      ++kt;
      rd1(kt);
      mma0();
      mma1(yes{});
      mfma_asm_drain();
    } else {
    {   // slab 0 has landed once at most min(NS - 2, nkt - 1) younger slabs are outstanding
      if (nkt - 1 >= NS - 2) vm_wait_n<8 * (NS - 2)>();
      else if (NS > 3 && nkt == 2) vm_wait_n<8>();
      else vm_wait_n<0>();
    }
    __builtin_amdgcn_s_barrier();
    rd0(0);
    // steady loop: slab kt + NS - 1 exists, so behind slab kt + 1 exactly NS - 3 younger slabs are outstanding
    int kt = 0;
    for (; kt + NS - 1 < nkt; ++kt) {
      rd1(kt);
      mma0();
      vm_wait_n<8 * (NS - 3)>();
      __builtin_amdgcn_s_barrier();      // behind it: slab kt + 1 is visible, and every wave has retired its reads of slab kt - 1
      rd0(kt + 1);
      if constexpr (DMA_BETWEEN) mma1_dma(kt + NS - 1);
      else { issue(kt + NS - 1); mma1(no{}); }
    }
    // drain: nothing left to request; min(NS - 3, nkt - 2 - kt) younger slabs behind slab kt + 1
    for (; kt + 1 < nkt; ++kt) {
      rd1(kt);
      mma0();
      if (NS > 3 && kt + 2 < nkt) vm_wait_n<8 * (NS - 3)>();
      else vm_wait_n<0>();
      __builtin_amdgcn_s_barrier();
      rd0(kt + 1);
      mma1(no{});
    }
    rd1(kt);
    mma0();
    mma1(yes{});
    mfma_asm_drain();
    }
  } else {
    for (int kt = 0; kt < nkt; ++kt) {
      {   // slab kt has landed once at most min(NS - 2, nkt - 1 - kt) younger slabs are outstanding (uniform)
        const int younger = nkt - 1 - kt;
        if (younger >= NS - 2) vm_wait_n<8 * (NS - 2)>();
        else if (NS > 3 && younger == 1) vm_wait_n<8>();
        else vm_wait_n<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (kt + NS - 1 < nkt) issue(kt + NS - 1);
      const uint32_t so = (uint32_t)((kt % NS) * STAGE);
      const uint32_t ra = ra0 + so, rb = rb0 + so;
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
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  __syncthreads();      // every wave's fragment reads are retired (and no DMA is outstanding): the rows epilogue turns its tile through LDS
  if constexpr (ACT != SL_ACT_SILU_MUL) {
    if (!p.direct_epi && tile_epilogue_rows<T, ACT, 4, true>(p, acc, bm * TBM + wm * 64, bn * TBN + wn * 64, lane, z, wz, (float*)smem + wave * 4096)) return;
  }
  tile_epilogue<T, ACT>(p, acc, bm, bn, wm, wn, q, r, z, wz);
}

template <typename T, int ACT, int NS, bool PIPE, bool SPLIT = false>
static int launch_ring(const GemmP& p, dim3 grid, hipStream_t st) {
  constexpr int LDS_BYTES = NS * 2 * TBM * TROWB;
  static std::atomic<uint64_t> attr_set{0};   // one bit per device: the opt-in to > 64 KiB of dynamic LDS is per device
  int devid = 0;
  SL_HIP(hipGetDevice(&devid));
  if (devid < 0 || devid >= 64 || !((attr_set.load(std::memory_order_relaxed) >> devid) & 1)) {
    SL_HIP(hipFuncSetAttribute((const void*)gemm_tiled_ring_kernel<T, ACT, NS, PIPE, SPLIT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    if (devid >= 0 && devid < 64) attr_set.fetch_or(1ull << devid, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL((gemm_tiled_ring_kernel<T, ACT, NS, PIPE, SPLIT>), grid, dim3(256), LDS_BYTES, st, p);
  SL_CHECK_LAUNCH("gemm_tiled_ring");
  return 0;
}

template <typename T, int ACT>
int sl_gemm128_ring_launch(const GemmP& p, int stages, dim3 grid, hipStream_t st) {
  if constexpr (sizeof(T) == 2) {
    if (stages == 3) return launch_ring<T, ACT, 3, true>(p, grid, st);
    if (stages == 104) return launch_ring<T, ACT, 4, false>(p, grid, st);      // the un-pipelined loop (A/B)
    if (stages == 204) return launch_ring<T, ACT, 4, true>(p, grid, st);       // all eight requests of a slab inside one 16-MFMA phase (A/B)
    return launch_ring<T, ACT, 4, true, true>(p, grid, st);
  } else {
    sl_set_error("sl_gemm: the ring form of the 128-tile kernel is built for 2-byte types only");
    return SL_ERR_UNSUPPORTED;
  }
}
template int sl_gemm128_ring_launch<bf16_t, SL_ACT_NONE>(const GemmP&, int, dim3, hipStream_t);
template int sl_gemm128_ring_launch<bf16_t, SL_ACT_GELU>(const GemmP&, int, dim3, hipStream_t);
template int sl_gemm128_ring_launch<bf16_t, SL_ACT_SILU_MUL>(const GemmP&, int, dim3, hipStream_t);
template int sl_gemm128_ring_launch<float, SL_ACT_NONE>(const GemmP&, int, dim3, hipStream_t);
template int sl_gemm128_ring_launch<float, SL_ACT_GELU>(const GemmP&, int, dim3, hipStream_t);
template int sl_gemm128_ring_launch<float, SL_ACT_SILU_MUL>(const GemmP&, int, dim3, hipStream_t);
