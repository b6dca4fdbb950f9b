// gemm256.hip — the 256 x 256 tile kernels (one-barrier-per-slab form, phased form, stream-K form) and their launch entry, split from
// gemm.hip so that the GEMM translation units compile side by side.
#include "common.h"
#include "gemm_internal.h"
#include "gemm_epilogue.h"

// ----------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves (2 x 4, 128 x 64 each), same K slabs / swizzled LDS image / LDS-DMA staging as above.
// Why: measured, a CU sustains only ~40 GB/s of operand fetches (L2 hits + HBM through one miss queue) — the 128^2 tile
// needs 32 KiB per 2*128*128*64 FLOP and tops out at 600-980 TF/s on that, not on the MFMA pipe.  The 256^2 tile halves
// the bytes per FLOP; one block per CU (128 KiB of LDS), two waves per SIMD.  (A four-stage ring of 64-byte slabs with
// three slabs of DMA in flight — swizzle c ^ ((row >> 2) & 2) for conflict-free reads of 64-byte rows — measured 5-8 %
// SLOWER: the limit is the fetch rate per CU, not its latency; what helped is sharing slabs in L2, below.)
// Also measured slower (-4..-7 %): issuing the DMA of slab k+2 in the middle of slab k behind an extra bare barrier (1.25-1.5
// product phases of cover instead of one).  PMC on 17408x16384x3072: MFMA busy 46 %, waves 30 % parked (vmcnt/barrier), 50 %
// issue-stalled behind the MFMA pipe, 20 % issuing; no LDS bank conflicts.
// A 256 x 128 tile with dedicated loader waves (8 compute + 4 loader waves, three 48 KiB slots, fragment reads interleaved with
// the MFMAs — the structure of gemm_stream_wide_kernel on row-major operands) measured 0.93-1.09 x this kernel on the
// encoder / prefill shapes (tools/bench_gemm_lw.py, round 2): it is bound by what a CU pulls from L2 (~31 B/clk) at 48 KiB per
// 1024 MFMA-cycles, this tile needs 64 KiB per 2048; removed again.
// A persistent form (one block per CU walking its tiles, the next tile's first slab requested before the current tile's
// epilogue, which then turns 32-row groups through the other staging buffer) measured within +-2 % of this kernel at
// K = 1024..8192: the vmcnt(0) that admits the prefetched slab also drains the epilogue's stores (one counter on gfx9).
// Round 3, again with the register epilogue of the swapped-operand form (no LDS in the epilogue, the K slabs of consecutive tiles as
// one double-buffered stream, scalar tile bases): 0.97-1.02 x on the encoder shapes (tools/time_fold_epilogue.py); removed again.
// A second build with one epilogue form per instantiation (0-44 bytes of spills instead of 20-96): QKV 748-761 vs 770 us, out_proj
// 300 vs 306-309, FFN1 + GELU 1 134 vs 1 131, FFN2 equal, the LayerNorm-fold forms 3 % slower — the 3-6 us per tile that in-kernel
// stamps show between a block's last store and its successor's first product do not turn into throughput; removed again.
// Where the time goes (127744 x 4096 x 1024, bias + GELU, 1180 us): product loop alone 820-845 us (1.27-1.3 PF/s), epilogue
// arithmetic without its stores +45 us, the stores +170..290 us — 128 KiB per tile leave a CU at ~24 GB/s, and neither spreading
// the first-round blocks of an XCD over a tile time nor a block that outlives its tile changes that.
// ----------------------------------------------------------------------------------------------

template <typename T, int ACT, bool SW = false>
__global__ __launch_bounds__(512, 1) void gemm_tiled256_kernel(GemmP p) {
  static_assert(!SW || (sizeof(T) == 2 && ACT != SL_ACT_SILU_MUL), "the swapped-operand form is the bf16 store epilogue");
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int BK = TROWB / (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][XBM * TROWB];   // [buf][A|W], 32 KiB each
  __shared__ float2 mr_s[XBM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int qn = nt >> 3, rn = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  // blocks that run together on an XCD (consecutive ids) cover an 8 x 4 patch of tiles, so they share A and W slabs in
  // that XCD's L2 (walking M only shares W: 33 slab streams per 32 blocks from beyond L2 instead of 12)
  int bm, bn;
  {
    const int GM = p.gm;
    const int per = GM * p.tiles_n, grp = bid / per, first = grp * GM;
    const int gsz = (p.tiles_m - first) < GM ? (p.tiles_m - first) : GM;
    const int in = bid - grp * per;
    bm = first + in % gsz;
    bn = in / gsz;
  }
  const int z = blockIdx.y;
  int64_t a_off; int wz;
  if (!resolve_group(p, z, bm, a_off, wz, XBM)) return;
  const T* A = (const T*)p.A + a_off;
  const T* W = (const T*)p.W + (int64_t)wz * p.sW + p.wx;
  if (p.grp_ext && bn * XBN >= p.N) return;

  // LDS chunk c = tid + 512 i sits at (row c>>3, physical chunk c&7) and must hold logical chunk (c&7)^(row&7)
  const T* ga[4];
  const T* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 512 * i, row = c >> 3, ch = (c & 7) ^ (row & 7);
    // SW: the W fragments are read at rows 8 (r >> 2) + (r & 3) + {0, 4, 32, 36}; the image is swizzled by those rows' (r & 3) and
    // bit 0 of (r >> 2), which keeps each 16-lane group of a ds_read_b128 on 16 different 16-byte slots of the 256-byte bank row
    const int chw = SW ? (c & 7) ^ ((row & 3) | (((row >> 3) & 1) << 2)) : ch;
    int ar = bm * XBM + row; ar = ar < p.M ? ar : p.M - 1;
    int wr = bn * XBN + row; wr = wr < p.N ? wr : p.N - 1;
    ga[i] = A + (int64_t)ar * p.lda + ch * VEC;
    gw[i] = W + (int64_t)wr * p.ldw + chw * VEC;
  }
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;  // this wave's 1 KiB piece inside an 8 KiB group

  f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = p.K / BK;
  auto issue = [&](int kt, int buf) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[i] + k0), (lds_ptr_t)(&smem[buf][0][i * 8192 + wave_lds]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(gw[i] + k0), (lds_ptr_t)(&smem[buf][1][i * 8192 + wave_lds]), 16, 0, 0);
    }
  };

  issue(0, 0);
  if (sizeof(T) == 2 && p.ln_mr && tid < XBM) {   // LayerNorm fold: this tile's {mean, rstd} pairs wait in LDS for the epilogue
    int row = bm * XBM + tid;
    row = row < p.M ? row : p.M - 1;
    mr_s[tid] = ((const float2*)p.ln_mr)[row];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) issue(kt + 1, buf ^ 1);
    const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(&smem[buf][0][0]);
    const uint32_t ra = sb + (uint32_t)((wm * 128 + r) * TROWB);
    const uint32_t rb = sb + (uint32_t)(XBM * TROWB + (wn * 64 + (SW ? 8 * (r >> 2) + (r & 3) : r)) * TROWB);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint32_t xs = (uint32_t)(((s * 4 + q) ^ (r & 7)) << 4);
      const uint32_t xw = SW ? (uint32_t)(((s * 4 + q) ^ ((r & 3) | (((r >> 2) & 1) << 2))) << 4) : xs;
      u32x4_t a[8], b[4];
      SL_LDS_RD(a[0], ra + xs, 0); SL_LDS_RD(a[1], ra + xs, 2048); SL_LDS_RD(a[2], ra + xs, 4096); SL_LDS_RD(a[3], ra + xs, 6144);
      if constexpr (SW) { SL_LDS_RD(b[0], rb + xw, 0); SL_LDS_RD(b[1], rb + xw, 512); SL_LDS_RD(b[2], rb + xw, 4096); SL_LDS_RD(b[3], rb + xw, 4608); }
      else { SL_LDS_RD(b[0], rb + xw, 0); SL_LDS_RD(b[1], rb + xw, 2048); SL_LDS_RD(b[2], rb + xw, 4096); SL_LDS_RD(b[3], rb + xw, 6144); }
      SL_LDS_RD(a[4], ra + xs, 8192); SL_LDS_RD(a[5], ra + xs, 10240); SL_LDS_RD(a[6], ra + xs, 12288); SL_LDS_RD(a[7], ra + xs, 14336);
      lds_wait8<4>(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if constexpr (SW) MMA<T>::step(acc[m][n], as_uint4(b[n]), as_uint4(a[m]));
          else MMA<T>::step(acc[m][n], as_uint4(a[m]), as_uint4(b[n]));
        }
      __builtin_amdgcn_sched_barrier(0);
      lds_wait8<0>(a[4], a[5], a[6], a[7], b[0], b[1], b[2], b[3]);
#pragma unroll
      for (int m = 4; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if constexpr (SW) MMA<T>::step(acc[m][n], as_uint4(b[n]), as_uint4(a[m]));
          else MMA<T>::step(acc[m][n], as_uint4(a[m]), as_uint4(b[n]));
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (SW) {
    const int rb0 = bm * XBM + wm * 128, cb0 = bn * XBN + wn * 64;
    const float2* mrl = mr_s + wm * 128;
    const bool res = p.res != nullptr, ln = p.ln_mr != nullptr, st = p.stats_out != nullptr;    // launch_tiled admits these five forms only
    if (p.aux) tile_epilogue_sw<ACT, EPI_AUX>(p, acc, rb0, cb0, lane, z, wz, mrl);
    else if (ln) tile_epilogue_sw<ACT, EPI_LN>(p, acc, rb0, cb0, lane, z, wz, mrl);
    else if (st) tile_epilogue_sw<ACT, EPI_RES | EPI_STATS>(p, acc, rb0, cb0, lane, z, wz, mrl);
    else if (res) tile_epilogue_sw<ACT, EPI_RES>(p, acc, rb0, cb0, lane, z, wz, mrl);
    else tile_epilogue_sw<ACT, 0>(p, acc, rb0, cb0, lane, z, wz, mrl);
    return;
  } else {
    if constexpr (ACT != SL_ACT_SILU_MUL) {
      if (!p.direct_epi && tile_epilogue_rows<T, ACT, 8>(p, acc, bm * XBM + wm * 128, bn * XBN + wn * 64, lane, z, wz, (float*)&smem[0][0][0] + wave * 4096,
                                                        sizeof(T) == 2 && p.ln_mr ? mr_s + wm * 128 : nullptr)) return;
    }
    tile_epilogue_g<T, ACT, 8, 4>(p, acc, bm * XBM + wm * 128, bn * XBN + wn * 64, q, r, z, wz);
  }
}

// ----------------------------------------------------------------------------------------------
// 256 x 256 tile, staggered two-phase main loop (round 4).  Same tile, wave grid (2 x 4 waves of 128 x 64), fragment layouts and
// epilogues as gemm_tiled256_kernel; what changes is how a K slab of 64 bytes per row moves through the block:
//   * the slab is cut into four 16 KiB PIECES — PA0 (A rows of the waves' upper 64 x 64 halves), PB0 (W rows of the waves' left 32
//     columns), PB1 (right 32 columns), PA1 (lower halves) — and a wave's 128 x 64 output into an upper and a lower 64 x 64 half,
//     one per PHASE: the upper phase reads PA0 + PB0 + PB1 into registers (16 ds_read_b128), the lower one PA1 (8; W's fragments
//     are kept).  A phase = {fragment reads, DMA of the two pieces 6 and 7 pieces ahead, counted vmcnt, lgkmcnt(0), barrier,
//     32 MFMAs, barrier}: the DMA stays in flight across barriers (never vmcnt(0) in the steady loop), eight LDS slots.
//   * waves 4-7 (the lower 128 rows) run ONE BARRIER behind waves 0-3: while one wave of a SIMD issues its 32 MFMAs its partner
//     issues its reads and its DMA, so the matrix pipe of a SIMD always has a wave to draw from (in the one-barrier-per-slab loop
//     all eight waves read together and multiply together: MFMA busy 0.40-0.49).
// Ordering (guide §5 "Read a staged buffer one phase AFTER the wait that retires it"): phase P issues pieces 2P + 6 and 2P + 7 and
// waits until piece 2P + 4 has landed (vmcnt(6): three pieces stay in flight); both halves of the block have made that wait once
// the lagging half's first barrier of phase P is passed, and the pieces of phase P + 1 (4t .. 4t + 2 for an upper phase 2t) are
// read behind it.  A slot is re-filled (piece n + 8, phase (n >> 1) + 1) one phase after its last read; that is enough because every
// wave retires its fragment reads (lgkmcnt(0)) in FRONT of the phase's first barrier.  The lagging half's MFMAs of its last phase
// are still running when the leading half leaves the loop: that half passes one more barrier before the epilogue touches LDS.
// What was measured on the way (profiles/r04_b_*.txt, tools/gemm_knockout.py, tools/gemm_qvar.py — MI355X, random bf16):
//   four phases of 16 MFMAs (the first form of this loop)   1 311-1 365 TF/s on 140 288 x 5 120 x 3 072 (round-3 loop 1 212, vendor 1 396-1 440)
//   knock-outs of that form: MFMAs + barriers alone 2 154 us, reads + DMA + barriers alone 2 010 us (= 67 GB/s per CU, the rate a CU
//     pulls from L2 when every CU streams: a 256^2 tile needs 64 KiB per 2 048 MFMA cycles = 32 B/clk), both 3 461 us: the loop is
//     bound by how well two equal costs overlap, and the eight barriers of a slab cost the matrix pipe ~56 cycles each
//   reads rebalanced 8/4/8/4 (next slab's W fragments early), DMA lead 5 / 7 / 8 pieces, no s_setprio: -13 .. +3 %
//   two phases of 32 MFMAs (half the barriers)              +3-6 %;  and without s_setprio around the clusters  +1-3 % more  <- this loop
// ----------------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ void lds_rd(u32x4_t& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void vm_wait_pieces(int n_out) {   // n_out pieces (two DMAs each per wave) may stay in flight; uniform
  switch (n_out) {
    case 0: vm_wait<0>(); break;
    case 1: vm_wait<2>(); break;
    case 2: vm_wait<4>(); break;
    default: vm_wait<6>(); break;
  }
}

// DBG (debug builds only, -DSL_GEMM_DEBUG): 8 = cycle stamps; knock-outs 1 = no fragment reads, 2 = no DMA, 4 = no MFMAs (results are then
// meaningless: timing experiments, tools/gemm_knockout.py)
//
// The main loop over `nkt` K slabs starting at element k_first, shared by the one-tile-per-block kernel and the stream-K kernel.
// gp[kind][i]: this thread's two source rows of piece kind {PA0, PB0, PB1, PA1} at k = 0; all eight waves call it together and
// leave it together (the leading half waits for the lagging one), with every DMA landed and every fragment read retired.
template <typename T, bool SW, int DBG>
__device__ __forceinline__ void t256_mainloop(unsigned char* smem, const T* const (&gp)[4][2], int64_t k_first, int nkt, int wave, int lane,
                                              f32x4 (&acc)[8][4], uint32_t* stamps) {
  constexpr bool STAMP = (DBG & 8) != 0, KO_RD = (DBG & 1) != 0, KO_DMA = (DBG & 2) != 0, KO_MMA = (DBG & 4) != 0;
  constexpr int BK = TROWB / (int)sizeof(T);
  constexpr int PIECE = 128 * TROWB;            // 16 KiB
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  auto stamp = [&](int i) {
    if constexpr (STAMP) {
      if ((wave & 3) == 0 && lane == 0) stamps[(wave >> 2) * 32 + i] = (uint32_t)__builtin_amdgcn_s_memtime();
    }
  };
  const int NP = 4 * nkt;
  // piece n = 4 * slab + kind goes to slot n & 7
  auto issue = [&](int n, int kind) {
    if constexpr (KO_DMA) return;
    const int64_t k0 = k_first + (int64_t)(n >> 2) * BK;
    unsigned char* dst = smem + (n & 7) * PIECE + wave * 1024;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(gp[kind][0] + k0), (lds_ptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(gp[kind][1] + k0), (lds_ptr_t)(dst + 8192), 16, 0, 0);
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int n = 0; n < 6; ++n)
    if (n < NP) issue(n, n & 3);
  {
    const int last = 5 < (NP - 1) ? 5 : (NP - 1);
    vm_wait_pieces(last - 2 > 0 ? last - 2 : 0);        // pieces 0, 1, 2 have landed
  }
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();           // the lower half of the block runs one barrier behind

  const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const uint32_t ra0 = sb + (uint32_t)((wm * 64 + r) * TROWB);
  const uint32_t rb0 = sb + (uint32_t)((wn * 32 + (SW ? 8 * (r >> 2) + (r & 3) : r)) * TROWB);
  const uint32_t ka = (uint32_t)(r & 7), kw = SW ? (uint32_t)((r & 3) | (((r >> 2) & 1) << 2)) : ka;
  const uint32_t xa0 = ((uint32_t)q ^ ka) << 4, xa1 = ((uint32_t)(4 + q) ^ ka) << 4;
  const uint32_t xw0 = ((uint32_t)q ^ kw) << 4, xw1 = ((uint32_t)(4 + q) ^ kw) << 4;
  constexpr int BN1 = SW ? 512 : 2048;               // second W fragment of a 32-column half: +4 rows (swapped form) / +16 rows

  u32x4_t a[8], b0[4], b1[4];                        // a[4 s + m'], b[2 s + n']
  if constexpr (KO_RD) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = u32x4_t{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; asm volatile("" : "+v"(a[i])); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { b0[i] = a[i]; b1[i] = a[4 + i]; asm volatile("" : "+v"(b0[i]), "+v"(b1[i])); }
  }

  auto rd_a = [&](uint32_t base) {                   // base = address of the piece's row (wm * 64 + r)
    if constexpr (KO_RD) return;
    lds_rd<0>(a[0], base + xa0); lds_rd<2048>(a[1], base + xa0); lds_rd<4096>(a[2], base + xa0); lds_rd<6144>(a[3], base + xa0);
    lds_rd<0>(a[4], base + xa1); lds_rd<2048>(a[5], base + xa1); lds_rd<4096>(a[6], base + xa1); lds_rd<6144>(a[7], base + xa1);
  };
  auto rd_b = [&](u32x4_t (&b)[4], uint32_t base) {
    if constexpr (KO_RD) return;
    lds_rd<0>(b[0], base + xw0); lds_rd<BN1>(b[1], base + xw0);
    lds_rd<0>(b[2], base + xw1); lds_rd<BN1>(b[3], base + xw1);
  };
  auto mma_q = [&](int mi, int nj, u32x4_t (&b)[4]) {     // quadrant (mi, nj): 16 MFMAs, both 64-byte k-steps
    if constexpr (KO_MMA) { asm volatile("" : "+v"(a[0]), "+v"(b[0])); return; }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          if constexpr (SW) MMA<T>::step(acc[mi * 4 + m][nj * 2 + n], as_uint4(b[2 * s + n]), as_uint4(a[4 * s + m]));
          else MMA<T>::step(acc[mi * 4 + m][nj * 2 + n], as_uint4(a[4 * s + m]), as_uint4(b[2 * s + n]));
        }
  };
  // the rest of a phase's load segment: DMA of pieces 2P + 6 and 2P + 7, the wait that retires piece 2P + 4, the fragment reads
  // retired (every fragment register tied to the wait so no MFMA moves above it), the barrier
  auto stage = [&](int P, bool steady) {
    const int n0 = 2 * P + 6;
    if (steady) {
      issue(n0, n0 & 3); issue(n0 + 1, (n0 + 1) & 3);
      vm_wait<6>();
    } else {
      if (n0 < NP) issue(n0, n0 & 3);
      if (n0 + 1 < NP) issue(n0 + 1, (n0 + 1) & 3);
      const int last = (n0 + 1) < (NP - 1) ? (n0 + 1) : (NP - 1);
      vm_wait_pieces(last - (2 * P + 4) > 0 ? last - (2 * P + 4) : 0);
    }
    lds_wait8<0>(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
    lds_wait8<0>(b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  auto close = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); };
  // stamps (instrumented build): slabs 4 and 5, per phase {load segment start, first barrier passed, MFMAs issued} -> entries 2 .. 13
  auto tile = [&](int t, bool steady) {
    const uint32_t bo = (uint32_t)(t & 1) * (4 * PIECE);
    const bool st = STAMP && (t == 4 || t == 5);
    const int sb_ = 2 + (t - 4) * 6;
    if (st) stamp(sb_ + 0);
    rd_b(b0, rb0 + bo + 1 * PIECE);
    rd_a(ra0 + bo + 0 * PIECE);
    rd_b(b1, rb0 + bo + 2 * PIECE);
    stage(2 * t, steady);
    if (st) stamp(sb_ + 1);
    mma_q(0, 0, b0);
    mma_q(0, 1, b1);
    if (st) stamp(sb_ + 2);
    close();
    if (st) stamp(sb_ + 3);
    rd_a(ra0 + bo + 3 * PIECE);
    stage(2 * t + 1, steady);
    if (st) stamp(sb_ + 4);
    mma_q(1, 1, b1);
    mma_q(1, 0, b0);
    if (st) stamp(sb_ + 5);
    close();
  };
  const int nsteady = nkt - 2;                         // slabs whose two phases both issue: 2 (2 t + 1) + 7 <= NP - 1
  int t = 0;
  stamp(1);
  for (; t < nsteady; ++t) tile(t, true);
  for (; t < nkt; ++t) tile(t, false);
  stamp(14);
  if (wm == 0) __builtin_amdgcn_s_barrier();           // the lagging half's last MFMA segment ends behind this one
}

// staging pointers of tile (bm, bn): LDS chunk c = tid + 512 i of a piece sits at (piece row c >> 3, physical chunk c & 7) and holds the
// logical chunk (c & 7) ^ key(piece row).  Piece rows: A pieces = [half of the block 0/1][64 rows], W pieces = [wave column 0..3][32 rows].
template <typename T, bool SW>
__device__ __forceinline__ void t256_stage_ptrs(const GemmP& p, const T* A, const T* W, int bm, int bn, int tid, const T* (&gp)[4][2]) {
  constexpr int VEC = Vec16<T>::VEC;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 512 * i, rho = c >> 3, pc = c & 7;
    const int cha = pc ^ (rho & 7);
    const int chw = SW ? pc ^ ((rho & 3) | (((rho >> 3) & 1) << 2)) : cha;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ar = bm * XBM + (rho >> 6) * 128 + h * 64 + (rho & 63); ar = ar < p.M ? ar : p.M - 1;
      int wr = bn * XBN + (rho >> 5) * 64 + h * 32 + (rho & 31); wr = wr < p.N ? wr : p.N - 1;
      gp[h ? 3 : 0][i] = A + (int64_t)ar * p.lda + cha * VEC;
      gp[h ? 2 : 1][i] = W + (int64_t)wr * p.ldw + chw * VEC;
    }
  }
}

// tile index (after the XCD remap) -> tile coordinates: XCD patches of GM tile rows (see gemm_tiled256_kernel)
__device__ __forceinline__ void t256_tile_coords(const GemmP& p, int v, int& bm, int& bn) {
  const int GM = p.gm;
  const int per = GM * p.tiles_n, grp = v / per, first = grp * GM;
  const int gsz = (p.tiles_m - first) < GM ? (p.tiles_m - first) : GM;
  const int in = v - grp * per;
  bm = first + in % gsz;
  bn = in / gsz;
}
__device__ __forceinline__ int xcd_remap(int bid, int n) {      // bijective: the blocks of one XCD (bid % 8 equal) get consecutive indices
  const int qn = n >> 3, rn = n & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
}

template <typename T, int ACT, bool SW, bool POSTS = false>
__device__ __forceinline__ void t256_epilogue(const GemmP& p, f32x4 (&acc)[8][4], int bm, int bn, int wave, int lane, int z, int wz, unsigned char* smem,
                                              float2* mr_s) {
  const int wm = wave >> 2, wn = wave & 3;
  if constexpr (SW) {
    const int rb0_ = bm * XBM + wm * 128, cb0 = bn * XBN + wn * 64;
    const float2* mrl = mr_s + wm * 128;
    const bool res = p.res != nullptr, ln = p.ln_mr != nullptr, st = p.stats_out != nullptr;    // launch_tiled admits these five forms only
    bool done = false;
    if constexpr (POSTS) {
     if (p.post) {         // launch_tiled admits exactly these post-op forms on the swapped-operand kernels (the phased kernel only)
      done = true;
      if constexpr (ACT == SL_ACT_GELU) {
        tile_epilogue_sw<ACT, EPI_AUX | EPI_DROP>(p, acc, rb0_, cb0, lane, z, wz, mrl);        // FFN1 forward: mid = dropout(gelu(pre)), pre kept
      } else {
        if (p.post == SL_POST_DROPOUT) tile_epilogue_sw<ACT, EPI_RES | EPI_DROP>(p, acc, rb0_, cb0, lane, z, wz, mrl);   // h = residual + dropout(sublayer)
        else if (p.post == SL_POST_GELU_BWD) tile_epilogue_sw<ACT, EPI_GBWD>(p, acc, rb0_, cb0, lane, z, wz, mrl);
        else tile_epilogue_sw<ACT, EPI_SBWD>(p, acc, rb0_, cb0, lane, z, wz, mrl);
      }
     }
    }
    if (done) return;
    if (p.aux) tile_epilogue_sw<ACT, EPI_AUX>(p, acc, rb0_, cb0, lane, z, wz, mrl);
    else if (ln) tile_epilogue_sw<ACT, EPI_LN>(p, acc, rb0_, cb0, lane, z, wz, mrl);
    else if (st) tile_epilogue_sw<ACT, EPI_RES | EPI_STATS>(p, acc, rb0_, cb0, lane, z, wz, mrl);
    else if (res) tile_epilogue_sw<ACT, EPI_RES>(p, acc, rb0_, cb0, lane, z, wz, mrl);
    else tile_epilogue_sw<ACT, 0>(p, acc, rb0_, cb0, lane, z, wz, mrl);
  } else {
    if constexpr (ACT != SL_ACT_SILU_MUL) {
      // the LDS-turned rows epilogue uses 16 KiB per wave of the piece slots (every DMA has landed: the last phases wait vmcnt(0))
      if (!p.direct_epi && tile_epilogue_rows<T, ACT, 8, POSTS>(p, acc, bm * XBM + wm * 128, bn * XBN + wn * 64, lane, z, wz, (float*)smem + wave * 4096,
                                                        sizeof(T) == 2 && p.ln_mr ? mr_s + wm * 128 : nullptr)) return;
    }
    tile_epilogue_g<T, ACT, 8, 4>(p, acc, bm * XBM + wm * 128, bn * XBN + wn * 64, lane >> 4, lane & 15, z, wz);
  }
}

template <typename T, int ACT, bool SW = false, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm_tiled256p_kernel(GemmP p) {
  constexpr bool STAMP = (DBG & 8) != 0;
  static_assert(!SW || (sizeof(T) == 2 && ACT != SL_ACT_SILU_MUL), "the swapped-operand form is the bf16 store epilogue");
  constexpr int BK = TROWB / (int)sizeof(T);
  constexpr int PIECE = 128 * TROWB;            // 16 KiB
  // one LDS object (a second one beside an LDS-DMA target can cost a vmcnt(0) per k-step, guide §5 item 4a): 8 piece slots + {mean, rstd}
  __shared__ __attribute__((aligned(16))) unsigned char smem[8 * PIECE + XBM * 8];
  float2* mr_s = (float2*)(smem + 8 * PIECE);   // instrumented build: the stamps of lane 0 of waves 0 and 4 live here (no fold in that build)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (STAMP) {
    if ((wave & 3) == 0 && lane == 0) ((uint32_t*)mr_s)[(wave >> 2) * 32] = (uint32_t)__builtin_amdgcn_s_memtime();
  }
  int bm, bn;
  t256_tile_coords(p, xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n), bm, bn);
  const int z = blockIdx.y;
  int64_t a_off; int wz;
  if (!resolve_group(p, z, bm, a_off, wz, XBM)) return;
  const T* A = (const T*)p.A + a_off;
  const T* W = (const T*)p.W + (int64_t)wz * p.sW + p.wx;
  if (p.grp_ext && bn * XBN >= p.N) return;
  const T* gp[4][2];     // [PA0, PB0, PB1, PA1][i]
  t256_stage_ptrs<T, SW>(p, A, W, bm, bn, tid, gp);

  f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (sizeof(T) == 2 && p.ln_mr && tid < XBM) {   // LayerNorm fold: this tile's {mean, rstd} pairs wait in LDS for the epilogue
    int row = bm * XBM + tid;                     // (before the first DMA: the compiler drains vmcnt for this load's use)
    row = row < p.M ? row : p.M - 1;
    mr_s[tid] = ((const float2*)p.ln_mr)[row];
  }
  int64_t k_first = 0;
  int nkt = p.K / BK;
  if (p.krun > 0) {      // K runs of uneven length (gemm.hip splitk256_runs): this block's run of whole slabs
    k_first = (int64_t)z * p.krun * BK;
    const int left = nkt - z * p.krun;
    nkt = left < p.krun ? left : p.krun;
  }
  t256_mainloop<T, SW, DBG>(smem, gp, k_first, nkt, wave, lane, acc, (uint32_t*)mr_s);
  if constexpr (STAMP) {
    if ((wave & 3) == 0 && lane < 32 && p.stamp) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      p.stamp[((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + (wave >> 2)) * 32 + lane] = ((const uint32_t*)mr_s)[(wave >> 2) * 32 + lane];
    }
  }
  t256_epilogue<T, ACT, SW, DBG == 0>(p, acc, bm, bn, wave, lane, z, wz, smem, mr_s);
}

// ----------------------------------------------------------------------------------------------
// Stream-K form of the kernel above (un-grouped bf16 / fp32 products whose 256^2 tiles do not fill the chip evenly: KD windows of
// 2-8 k rows, the per-rank KD regime of a few hundred rows, weight gradients of 16-64 tiles under K = 8 000, decode projections).
// The tiles' K slabs form one sequence of tiles x slabs units, cut into `gridDim.x` equal contiguous ranges, one per block (one
// block per CU); a block walks its range from the top down, tile segment by tile segment, each segment through the main loop above.
//   * a segment that is a whole tile: the usual epilogue;
//   * a segment that does not reach its tile's last slab (only a block's FIRST segment can be one): the accumulators go to the
//     block's slot of the workspace as fp32 (16-byte write-through stores), every wave drains, one lane raises the block's flag;
//   * a segment that ends its tile but does not start it (only a block's LAST segment): the tile's other segments belong to the
//     blocks just below, which produced them first thing — the owner polls their flags (one lane, relaxed, s_sleep), takes ONE
//     agent-scope acquire, adds the partial sums in descending block order (a fixed order: results are reproducible, though not
//     bit-identical to the unsplit kernel), clears the flags and runs the epilogue.
// Waiting is only ever for work that was started before the waiter's own: no cycle, and with at most one block per CU resident
// (130 KiB of LDS) every block of a grid of <= #CUs blocks is resident or becomes resident as soon as any kernel's block retires.
// Workspace (caller-owned, zero-initialised once): [flags: 1 KiB][gridDim.x slots of 256 KiB].  Guide §6 Guideline 16 (R1).
// ----------------------------------------------------------------------------------------------

template <typename T, int ACT, bool SW>
__global__ __launch_bounds__(512, 2) void gemm_tiled256sk_kernel(GemmP p, unsigned char* ws) {
  static_assert(!SW || (sizeof(T) == 2 && ACT != SL_ACT_SILU_MUL), "the swapped-operand form is the bf16 store epilogue");
  constexpr int BK = TROWB / (int)sizeof(T);
  constexpr int PIECE = 128 * TROWB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[8 * PIECE + XBM * 8];
  float2* mr_s = (float2*)(smem + 8 * PIECE);
  typedef __attribute__((address_space(1))) unsigned int gu32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, nkt = p.K / BK, nt = p.tiles_m * p.tiles_n;
  const int vb = xcd_remap(blockIdx.x, G);                 // blocks of one XCD take consecutive ranges: their tiles share A / W slabs in its L2
  const int64_t U = (int64_t)nt * nkt;
  const int64_t u_lo = U * vb / G;
  int64_t u_hi = U * (vb + 1) / G;
  gu32* flags = (gu32*)ws;
  float* slots = (float*)(ws + SK_FLAG_BYTES);
  const T* A = (const T*)p.A;
  const T* W = (const T*)p.W;
  bool first = true;
  while (u_hi > u_lo) {
    const int tile = (int)((u_hi - 1) / nkt);
    const int64_t t0 = (int64_t)tile * nkt;
    const int s1 = (int)(u_hi - t0), s0 = (int)((u_lo > t0 ? u_lo : t0) - t0);
    int bm, bn;
    t256_tile_coords(p, tile, bm, bn);               // unit order = the XCD-patch tile order: the blocks of one XCD (consecutive vb) work on neighbouring tiles
    const T* gp[4][2];
    t256_stage_ptrs<T, SW>(p, A, W, bm, bn, tid, gp);
    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!first) __builtin_amdgcn_s_barrier();              // the previous segment's epilogue may still be turning rows through LDS in another wave
    first = false;
    t256_mainloop<T, SW, 0>(smem, gp, (int64_t)s0 * BK, s1 - s0, wave, lane, acc, nullptr);
    if (s1 < nkt) {
      // partial sums -> this block's slot, in register order: [wave][fragment][lane] x 16 bytes, write-through (sc1)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slots + (int64_t)vb * (XBM * XBN), 0, (int)SK_SLOT_BYTES, 0x00020000);
      const int off = (wave * 32 * 64 + lane) * 16;
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[m][n]), rs, off + (m * 4 + n) * 1024, 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave drains, then the workgroup's barrier, then ONE flag store
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + vb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (s0 > 0) {
        // owner: the rest of this tile sits in the slots of the blocks below, down to the one that holds the tile's first slab
        int v_first = vb - 1;
        while (U * v_first / G > t0) --v_first;
        if (wave == 0) {
          for (int v = vb - 1; v >= v_first; --v) {
            if (lane == 0) {
              while (__hip_atomic_load(flags + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        for (int v = vb - 1; v >= v_first; --v) {
          const f32x4* src = (const f32x4*)(slots + (int64_t)v * (XBM * XBN)) + wave * 32 * 64 + lane;
#pragma unroll
          for (int m = 0; m < 8; m += 2) {       // eight fragments (32 registers) in flight at a time: all 32 at once would need 128, fewer leaves the
            f32x4 t8[2][4];                        // read latency-bound (guide: >= 8 loads per lane outstanding on a handed-off tile)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int n = 0; n < 4; ++n) t8[h][n] = __builtin_nontemporal_load(src + ((m + h) * 4 + n) * 64);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int n = 0; n < 4; ++n) acc[m + h][n] += t8[h][n];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (tid < vb - v_first) __hip_atomic_store(flags + (vb - 1 - tid), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed: ready for the next launch
      }
      t256_epilogue<T, ACT, SW>(p, acc, bm, bn, wave, lane, 0, 0, smem, mr_s);
    }
    u_hi = t0 + s0;
  }
}


// kind: SL_T256_* (gemm_internal.h).  grid = (tiles, batch) for the tile kernels, (G) for the stream-K form.
template <typename T, int ACT>
int sl_gemm256_launch(const GemmP& p, int kind, dim3 grid, void* sk_ws, hipStream_t st) {
  constexpr bool SWOK = sizeof(T) == 2 && ACT != SL_ACT_SILU_MUL;
  switch (kind) {
    case SL_T256_PHASED: hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, false>), grid, dim3(512), 0, st, p); break;
    case SL_T256_PLAIN: hipLaunchKernelGGL((gemm_tiled256_kernel<T, ACT>), grid, dim3(512), 0, st, p); break;
    case SL_T256_SK: hipLaunchKernelGGL((gemm_tiled256sk_kernel<T, ACT, false>), grid, dim3(512), 0, st, p, (unsigned char*)sk_ws); break;
    default:
      if constexpr (SWOK) {
        if (kind == SL_T256_PHASED_SW) { hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true>), grid, dim3(512), 0, st, p); break; }
        if (kind == SL_T256_PLAIN_SW) { hipLaunchKernelGGL((gemm_tiled256_kernel<T, ACT, true>), grid, dim3(512), 0, st, p); break; }
        if (kind == SL_T256_SK_SW) { hipLaunchKernelGGL((gemm_tiled256sk_kernel<T, ACT, true>), grid, dim3(512), 0, st, p, (unsigned char*)sk_ws); break; }
#ifdef SL_GEMM_DEBUG
        if constexpr (ACT == SL_ACT_NONE) {       // instrumented / knocked-out builds (tools/gemm_stamps.py, tools/gemm_knockout.py), never in the product .so
          if (kind == SL_T256_DBG + 8) { hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 8>), grid, dim3(512), 0, st, p); break; }
          if (kind == SL_T256_DBG + 1) { hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 1>), grid, dim3(512), 0, st, p); break; }
          if (kind == SL_T256_DBG + 2) { hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 2>), grid, dim3(512), 0, st, p); break; }
          if (kind == SL_T256_DBG + 3) { hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 3>), grid, dim3(512), 0, st, p); break; }
          if (kind == SL_T256_DBG + 4) { hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 4>), grid, dim3(512), 0, st, p); break; }
          if (kind == SL_T256_DBG + 6) { hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 6>), grid, dim3(512), 0, st, p); break; }
        }
#endif
      }
      sl_set_error("sl_gemm: 256-tile kernel kind %d is not built for this type / epilogue", kind);
      return SL_ERR_UNSUPPORTED;
  }
  SL_CHECK_LAUNCH("gemm_tiled256");
  return 0;
}
template int sl_gemm256_launch<bf16_t, SL_ACT_NONE>(const GemmP&, int, dim3, void*, hipStream_t);
template int sl_gemm256_launch<bf16_t, SL_ACT_GELU>(const GemmP&, int, dim3, void*, hipStream_t);
template int sl_gemm256_launch<bf16_t, SL_ACT_SILU_MUL>(const GemmP&, int, dim3, void*, hipStream_t);
template int sl_gemm256_launch<float, SL_ACT_NONE>(const GemmP&, int, dim3, void*, hipStream_t);
template int sl_gemm256_launch<float, SL_ACT_GELU>(const GemmP&, int, dim3, void*, hipStream_t);
template int sl_gemm256_launch<float, SL_ACT_SILU_MUL>(const GemmP&, int, dim3, void*, hipStream_t);
