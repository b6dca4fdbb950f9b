// gemm_stream.hip — C = epilogue(x . W^T) for mid-size row counts (large-batch decode: 16 < M <= a few hundred
// activation rows against the fragment-packed decode weights).  Still HBM-bound (ridge is ~300 rows on MI355X),
// but unlike the skinny kernel the activations no longer fit the "re-read them per fragment from L2" model:
//   * block = NWV compute waves + 1 loader wave.  The compute waves walk the SAME k-range over different weight
//     fragments (RF fragments = 16*RF weight rows each), so one x slab serves 16*RF*NWV weight rows.
//   * the loader wave DMAs x slabs ([16*MT rows][128 B of K], XOR-swizzled as in the tiled kernel) straight into a
//     ring of three LDS slabs, two stages ahead of the MFMAs.  LDS-DMA lives in the loader wave only: in a wave that
//     also reads LDS or streams weights the compiler guards every LDS read and barrier after a global_load_lds with
//     vmcnt(0), which drains the weight prefetch each stage.
//   * weight fragments go global -> VGPR -> MFMA (one contiguous 1 KiB per wave-load, packed layout), D stages
//     ahead in a register ring; nothing of W ever touches LDS.
//   * when the grid would not fill the CUs, K is split over `splits` blocks: those write fp32 partials and
//     gemm_stream_reduce_kernel applies the epilogue.
// (Tried at 384+ rows against 8192+ weight rows: 256-weight-row blocks — 4 waves x 4 fragments, all waves DMA-ing the slab
// and reading it with compiler-invisible ds_read_b128, no loader wave — to halve the activation bytes per CU: 74 vs 67 us
// at M = 512, 137 vs 102 us at M = 768; dropped.)
// Epilogues are the decode ones of gemm.hip: +bias/+residual, SILU_MUL (fragment pairs gate/up), ROPE_KV (q rotated,
// k/v appended to the cache), and the fused RMSNorm row scale (sum of squares taken from the staged x slabs).
#include <stdlib.h>

#include <atomic>

#include "gemm_internal.h"

struct StreamX {
  float* part;     // [splits][M][np] fp32 partial sums
  float* part_ss;  // [splits][M] partial sums of squares (fused RMSNorm)
  int splits, sps; // K splits over blocks, 128-byte stages per split
  int np;          // row stride of `part` in floats (fragments * 16)
  int* cnt;        // in-kernel K-split fix-up (wide form): [1024] tile arrival counters + [64] row-block counters, zero between launches
  float* ss_part;  // [n-blocks][M] sums of squares of the stored rows per 128-column block (row statistics of the fix-up)
};

// Hand-off of partial records between workgroups inside one launch (K-split fix-up).  A release / acquire fence pair at agent
// scope writes back and invalidates the XCD's whole L2 (measured: +80-95 us on a 30 us GEMM); instead every handed-off byte is
// stored and loaded with sc1 (agent scope: through the non-coherent L2 to the memory side) and drained with s_waitcnt before /
// after the arrival counter (MI355X_MICROARCH.md, Correctness boundaries, second valid form).
__device__ __forceinline__ void st_sc1_16(float* ptr, const f32x4& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory");
}
__device__ __forceinline__ void ld_sc1_16(f32x4& v, const float* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void vm_drain16(f32x4 (&v)[16]) {   // the 16 sc1 loads above have landed; ties the registers to the wait
  asm volatile("s_waitcnt vmcnt(0)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]), "+v"(v[9]), "+v"(v[10]),
                 "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
               :
               : "memory");
}

// final values for 4 consecutive output columns n4..n4+3 of fragment gf (pairs: gf = first fragment), row m
template <typename T, int ACT>
__device__ __forceinline__ void stream_epilogue4(const GemmP& p, const SkinnyX& sx, int m, int gf, int n4, const float (&a)[4], const float (&b)[4]) {
  const T* bias = (const T*)p.bias;
  if constexpr (ACT == SL_ACT_SILU_MUL) {
    const int ocol = (gf >> 1) * 16 + n4;
    if (ocol >= (p.N >> 1)) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float g = a[i], u = b[i];
      if (bias) { g += to_f32(bias[gf * 16 + n4 + i]); u += to_f32(bias[(gf + 1) * 16 + n4 + i]); }
      store_out<T>(p, p.C, p.res, m, ocol + i, silu(g) * u);
    }
  } else if constexpr (ACT == SL_ACT_ROPE_KV) {
    if (gf * 16 >= p.N) return;
    const int hh = gf >> 3, j = (gf & 7) >> 1;   // a head is 8 fragments (D = 128); pair (2j, 2j+1) = dims d, d+64
    const int pos = sx.pos[m];
    if (hh < sx.nh + sx.nkv) {
      T* dst = hh < sx.nh ? (T*)p.C + (int64_t)m * p.ldc + hh * 128
                          : (T*)sx.kc + (((int64_t)sx.seq[m] * sx.nkv + (hh - sx.nh)) * sx.max_ctx + pos) * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = j * 16 + n4 + i;
        const float c = sx.cos[(int64_t)pos * 64 + d], s = sx.sin[(int64_t)pos * 64 + d];
        dst[d] = from_f32<T>(a[i] * c - b[i] * s);
        dst[d + 64] = from_f32<T>(b[i] * c + a[i] * s);
      }
    } else {  // v rows are in natural order: fragments 2j, 2j+1 = dims d, d+16
      T* dst = (T*)sx.vc + (((int64_t)sx.seq[m] * sx.nkv + (hh - sx.nh - sx.nkv)) * sx.max_ctx + pos) * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int d = (gf & 7) * 16 + n4 + i;
        dst[d] = from_f32<T>(a[i]);
        dst[d + 16] = from_f32<T>(b[i]);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int col = gf * 16 + n4 + i;
      if (col >= p.N) continue;
      float v = a[i];
      if (bias) v += to_f32(bias[col]);
      if constexpr (ACT == SL_ACT_GELU) v = gelu_act<T>(v);
      store_out<T>(p, p.C, p.res, m, col, v);
    }
  }
}

// what a compute wave does with its accumulators: K-split partial records, or scale + epilogue.
// acc[f][t][i] = D[weight row (fi0 + f) * 16 + 4q + i][x row m0 + t * 16 + r]
template <typename T, int MT, int ACT, int RF>
__device__ __forceinline__ void stream_finish(const GemmP& p, const SkinnyX& sx, const StreamX& s, f32x4 (&acc)[RF][MT], const float (&ssum)[MT],
                                              bool fuse, int m0, int sp, int fi0, int nfrag, bool writes_ss, int q, int r) {
  constexpr bool PAIRS = (ACT == SL_ACT_SILU_MUL || ACT == SL_ACT_ROPE_KV);
  if (s.splits > 1) {
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = m0 + t * 16 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int f = 0; f < RF; ++f)
        if (fi0 + f < nfrag) {
          float* dst = s.part + ((int64_t)sp * p.M + m) * s.np + (fi0 + f) * 16 + 4 * q;
          if (s.cnt) st_sc1_16(dst, acc[f][t]); else *(f32x4*)dst = acc[f][t];   // s.cnt: another workgroup of this launch reads the record
        }
      if (fuse && writes_ss && q == 0) s.part_ss[(int64_t)sp * p.M + m] = ssum[t];
    }
    return;
  }
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = m0 + t * 16 + r;
    if (m >= p.M) continue;
    const float rs = sx.rstd_in ? sx.rstd_in[m] : (fuse ? rsqrtf(ssum[t] / (float)p.K + sx.eps) : 1.0f);
    if constexpr (PAIRS) {
#pragma unroll
      for (int pr = 0; pr < RF / 2; ++pr) {
        float a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = acc[2 * pr][t][i] * rs; b[i] = acc[2 * pr + 1][t][i] * rs; }
        stream_epilogue4<T, ACT>(p, sx, m, fi0 + 2 * pr, 4 * q, a, b);
      }
    } else {
#pragma unroll
      for (int f = 0; f < RF; ++f) {
        float a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = acc[f][t][i] * rs;
        stream_epilogue4<T, ACT>(p, sx, m, fi0 + f, 4 * q, a, a);
      }
    }
  }
}

template <typename T, int MT, int ACT, int RF, int NWV, int D, int NL, bool FUSE>
__global__ __launch_bounds__(64 * (NWV + NL)) void gemm_stream_kernel(GemmP p, SkinnyX sx, StreamX s) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int KSTEP = MMA<T>::KSTEP;
  constexpr bool PAIRS = (ACT == SL_ACT_SILU_MUL || ACT == SL_ACT_ROPE_KV);
  static_assert(!PAIRS || RF % 2 == 0, "pair epilogues need an even number of fragments");
  static_assert(MT % NL == 0 && MT / NL <= 8, "a loader wave stages at most 128 rows");
  constexpr int MTL = MT / NL;                // row tiles per loader wave (NL of them: one wave's LDS-DMA stream tops out at ~25 GB/s)
  constexpr int SLAB = MT * 16 * TROWB;       // bytes of one x slab
  constexpr int NXL = 2 * MTL;                // 16-byte x chunks per loader lane per stage
  constexpr int NSLOT = 3;                    // LDS slabs: one being read, two being filled
  constexpr int NSS = (MT + NWV - 1) / NWV;   // row tiles whose RMSNorm statistics a compute wave owns
  __shared__ __attribute__((aligned(16))) unsigned char smem[NSLOT * SLAB];
  __shared__ float ssl[MT * 16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave;                        // compute waves 0..NWV-1; wave NWV stages x
  const int r = lane & 15, q = lane >> 4;
  // block id -> (n-block, row block): the row blocks of one n-block sit 8 ids apart, i.e. on the same XCD (ids round-robin
  // over the 8 XCDs), so the second row block's weight stream hits that XCD's L2 instead of going back to HBM / MALL
  const int mblocks = (p.M + MT * 16 - 1) / (MT * 16);
  const int xj = blockIdx.x >> 3, nb = (xj / mblocks) * 8 + (blockIdx.x & 7);
  if (nb * NWV * RF >= ((p.N + 15) >> 4)) return;       // padding of the n-block count to a multiple of 8
  const int m0 = (xj % mblocks) * (MT * 16);
  const int sp = blockIdx.z;
  const int nfrag = (p.N + 15) >> 4;
  const int nks = p.K / KSTEP;                // 64-byte k-steps; a stage is two of them
  const int fi0 = (nb * NWV + wn) * RF;

  const int g_lo = sp * s.sps;
  const int g_hi = min(nks >> 1, g_lo + s.sps);
  const int per = g_hi - g_lo;
  const int n_it = per;                       // stages (= barriers) of this block; the ring's last partial turn is peeled
  const int g_last = g_hi - 1;

  f32x4 acc[RF][MT];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int f = 0; f < RF; ++f) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr bool fuse = FUSE;   // RMSNorm statistics taken here (sx.fuse_rms without sx.rstd_in: the producer of A did not supply them);
                                // compile-time so that the main loop stays one basic block for the issue-order directives

  if (wave >= NWV) {
    // ---- x loader.  Its vmcnt queue holds only x loads: a wave that also streamed weights would, waiting for an
    // x slab issued two stages ago, drain every older weight load with it (vmcnt completes in issue order), which caps
    // the weight prefetch at ~3 stages whatever the ring depth (measured: D = 4, 6, 8 identical at 23 GB/s per CU).
    // The slabs go global -> LDS by LDS-DMA (no staging VGPRs, no ds_write pass: a single wave moving 16 KiB per stage
    // through registers was the critical path at ~0.5 us per stage); three LDS slots keep two stages of DMA in flight.
    const T* A = (const T*)p.A;
    const int lw = wave - NWV;                        // loader index: rows [16 MTL lw, 16 MTL (lw + 1)) of the block
    const int ch = (lane & 7) ^ ((lane >> 3) & 7);   // LDS chunk c = lane + 64 i: row (lane >> 3) + 8 i, logical chunk ch
    const T* gx[NXL];
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      int xr_ = m0 + lw * (MTL * 16) + (lane >> 3) + 8 * i; xr_ = xr_ < p.M ? xr_ : p.M - 1;
      gx[i] = A + (int64_t)xr_ * p.lda + ch * VEC;
    }
    unsigned char* lbase = smem + lw * (MTL * 16 * TROWB);
    auto dma = [&](int it) {   // x slab of stage g_lo + it -> slot it % NSLOT
      int stg = g_lo + it; stg = stg < g_last ? stg : g_last;
      unsigned char* dst = lbase + (it % NSLOT) * SLAB;
#pragma unroll
      for (int i = 0; i < NXL; ++i)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(gx[i] + (int64_t)stg * (2 * KSTEP)), (lds_ptr_t)(dst + i * 1024), 16, 0, 0);
    };
    dma(0);
    dma(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NXL) : "memory");
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it < n_it; ++it) {
      dma(it + 2);             // into the slot every wave finished reading one barrier ago
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NXL) : "memory");   // stage it+1 has landed
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    // ---- compute waves: weight fragments global -> VGPR ring (D stages = D * RF * 2 KiB per wave in flight) -> MFMA
    const T* W = (const T*)p.W;
    const T* wp[RF];
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      int fi = fi0 + f; fi = fi < nfrag ? fi : nfrag - 1;
      wp[f] = W + (int64_t)fi * nks * (64 * VEC) + lane * VEC;
    }
    f32x4 ssacc[NSS];
#pragma unroll
    for (int j = 0; j < NSS; ++j) ssacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint4 wr[D][2][RF];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      int stg = g_lo + d; stg = stg < g_last ? stg : g_last;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int f = 0; f < RF; ++f) wr[d][s2][f] = ld_nt16(wp[f] + (int64_t)(stg * 2 + s2) * (64 * VEC));
    }
    __syncthreads();
    // One wave per SIMD: nothing else hides the LDS latency, so the activation fragments of the NEXT k-step are requested
    // before the MFMAs of the current one (two register sets).  The stage's barrier sits between its two k-steps: by then this
    // wave's reads of the slab are complete (so the loader may refill it one barrier later) and the next slab has landed, whose
    // first fragments are requested right behind the barrier, under the second k-step's MFMAs.  (Requesting each k-step's
    // fragments just ahead of its own MFMAs left the matrix core idle for the LDS round trip twice per stage: 0.31 of the
    // kernel's cycles in MFMAs at 512 rows, PMC in profiles/.)
    uint4 fx[2][MT], fsx[2][NSS];
    auto rd_x = [&](int it, int s2, int set) {
      const unsigned char* sl_ = smem + (it % NSLOT) * SLAB;
#pragma unroll
      for (int t = 0; t < MT; ++t) fx[set][t] = *(const uint4*)(sl_ + lds_off(t * 16 + r, s2 * 4 + q));
      // fused-RMSNorm statistics on the matrix core: diag(X_t X_t^T) = the row sums of squares of row tile t.  Wave w owns
      // tiles w, w + NWV, ...; it re-reads them from LDS at a wave-dependent address (selecting among the fx registers
      // instead needs per-wave branches, which made the compiler shuttle the accumulators between AGPRs and VGPRs: +40 %)
      if constexpr (FUSE) {
#pragma unroll
        for (int j = 0; j < NSS; ++j) {
          const int t = (j * NWV + wn) < MT ? (j * NWV + wn) : MT - 1;
          fsx[set][j] = *(const uint4*)(sl_ + lds_off(t * 16 + r, s2 * 4 + q));
        }
      }
    };
    auto mma_x = [&](int d, int s2, int set) {
#pragma unroll
      for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int f = 0; f < RF; ++f) MMA<T>::step(acc[f][t], wr[d][s2][f], fx[set][t]);
      if constexpr (FUSE) {
#pragma unroll
        for (int j = 0; j < NSS; ++j) MMA<T>::step(ssacc[j], fsx[set][j], fsx[set][j]);
      }
    };
    // issue order inside a k-step: RF MFMAs, one fragment read, RF MFMAs, ... — the four waves' ds_read_b128 bursts no longer
    // collide on the LDS port with the matrix core idle behind them (in-order issue: 8 reads x 4 waves queued ~130 cycles)
    auto order = [&]() {
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int t = 0; t < MT; ++t) {
          __builtin_amdgcn_sched_group_barrier(0x008, RF, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if constexpr (FUSE) {
          __builtin_amdgcn_sched_group_barrier(0x008, NSS, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, NSS, 0);
        }
      }
    };
    auto stage = [&](const int d, const int it) {
      __builtin_amdgcn_sched_barrier(0);
      rd_x(it, 1, 1);
      mma_x(d, 0, 0);
      order();
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave is done with slab `it`
      __builtin_amdgcn_s_barrier();
      rd_x(it + 1, 0, 0);
      mma_x(d, 1, 1);
      order();
      __builtin_amdgcn_sched_barrier(0);
      int stg = g_lo + it + D; stg = stg < g_last ? stg : g_last;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int f = 0; f < RF; ++f) wr[d][s2][f] = ld_nt16(wp[f] + (int64_t)(stg * 2 + s2) * (64 * VEC));
    };
    rd_x(0, 0, 0);
    const int n_full = n_it / D * D;
    for (int it0 = 0; it0 < n_full; it0 += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) stage(d, it0 + d);
    }
#pragma unroll
    for (int d = 0; d < D - 1; ++d)
      if (n_full + d < n_it) stage(d, n_full + d);
    if constexpr (FUSE) {   // lane (c = r, q) holds D[4q + i][c]: the diagonal element of row r sits in lane q == r >> 2, i = r & 3
#pragma unroll
      for (int j = 0; j < NSS; ++j) {
        const int t = j * NWV + wn;
        if (t < MT && (r >> 2) == q) ssl[t * 16 + r] = ssacc[j][r & 3];
      }
    }
  }
  __syncthreads();
  if (wave >= NWV) return;

  float ssum[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    ssum[t] = 0.f;
    if (fuse) ssum[t] = ssl[t * 16 + r];
  }
  stream_finish<T, MT, ACT, RF>(p, sx, s, acc, ssum, fuse, m0, sp, fi0, nfrag, nb == 0 && wn == 0, q, r);
}

// ----------------------------------------------------------------------------------------------
// Wide form for >= 385 rows: 256 x rows x 128 weight rows per block, 8 compute waves (4 x 2: 64 rows x 4 fragments each) and
// 4 loader waves; x AND the packed weight fragments go global -> LDS by LDS-DMA (a fragment k-step is 1 KiB in MFMA operand
// order, so its LDS image is read back lane-contiguous, conflict-free), three 48 KiB slots.
// Why (tools/ko_gateup.py knock-outs with in-kernel cycle stamps, profiles/r02_ko_gateup.txt): at 512 rows the 128 x 128
// block's main loop is bound by what a CU can pull from L2, ~31 B/clk (~17 TB/s chip-wide, the rate of reads of lines that
// every workgroup shares): 32 KiB per stage take ~1030 cycles with every CU streaming, whatever the loader-wave count, ring
// depth or memory level behind L2, against 528 cycles of MFMAs.  The 256 x 128 block moves 24 KiB per such stage of MFMAs;
// one block per CU (gate/up at 512 rows = 256 blocks) also pays the prologue / epilogue once instead of twice.
// ----------------------------------------------------------------------------------------------
template <typename T, int ACT>
__global__ __launch_bounds__(768) void gemm_stream_wide_kernel(GemmP p, SkinnyX sx, StreamX s) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int KSTEP = MMA<T>::KSTEP;
  constexpr int CW = 8;                       // compute waves; waves 8..11 are the four loaders
  constexpr int MTW = 4, RFW = 4;             // row tiles and weight fragments per compute wave
  constexpr int BM = 256, BF = 8;             // block: x rows, weight fragments
  constexpr int SLAB_X = BM * TROWB, SLAB_W = BF * 2 * 1024, SLOT = SLAB_X + SLAB_W;
  constexpr int NSLOT = 3;
  constexpr int NDMA = 12;                    // LDS-DMA instructions per loader wave per stage: 8 of x (64 rows), 4 of W (2 fragments)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int fx_last[2];
  __shared__ float fx_ss[4][2][64];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int mblocks = (p.M + BM - 1) / BM;
  const int nfrag = (p.N + 15) >> 4;
  const int xj = blockIdx.x >> 3, nb = (xj / mblocks) * 8 + (blockIdx.x & 7);   // row blocks of one n-block share an XCD (its L2)
  if (nb * BF >= nfrag) return;
  const int mblock = xj % mblocks;
  const int m0 = mblock * BM;
  const int sp = blockIdx.z;
  const int nks = p.K / KSTEP;
  const int g_lo = sp * s.sps;
  const int g_hi = min(nks >> 1, g_lo + s.sps);
  const int n_it = g_hi - g_lo;
  const int g_last = g_hi - 1;
  const int wm = (wave >> 1) & 3, wn = wave & 1;
  const int mw = m0 + wm * 64, fi0 = nb * BF + wn * RFW;

  f32x4 acc[RFW][MTW];
  float ssum[MTW];
#pragma unroll
  for (int t = 0; t < MTW; ++t) {
    ssum[t] = 0.f;
#pragma unroll
    for (int f = 0; f < RFW; ++f) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  if (wave >= CW) {
    const int lw = wave - CW;
    const T* A = (const T*)p.A;
    const T* W = (const T*)p.W;
    const int ch = (lane & 7) ^ ((lane >> 3) & 7);
    const T* gx[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      int xr_ = m0 + lw * 64 + (lane >> 3) + 8 * i; xr_ = xr_ < p.M ? xr_ : p.M - 1;
      gx[i] = A + (int64_t)xr_ * p.lda + ch * VEC;
    }
    const T* gw[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      int fi = nb * BF + lw * 2 + f; fi = fi < nfrag ? fi : nfrag - 1;
      gw[f] = W + (int64_t)fi * nks * (64 * VEC) + lane * VEC;
    }
    unsigned char* xbase = smem + lw * (64 * TROWB);
    unsigned char* wbase = smem + SLAB_X + lw * (2 * 2 * 1024);
    auto dma = [&](int it) {
      int stg = g_lo + it; stg = stg < g_last ? stg : g_last;
      const int so = (it % NSLOT) * SLOT;
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((glb_ptr_t)(gx[i] + (int64_t)stg * (2 * KSTEP)), (lds_ptr_t)(xbase + so + i * 1024), 16, 0, 0);
#pragma unroll
      for (int f = 0; f < 2; ++f)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
          __builtin_amdgcn_global_load_lds((glb_ptr_t)(gw[f] + (int64_t)(stg * 2 + s2) * (64 * VEC)), (lds_ptr_t)(wbase + so + (f * 2 + s2) * 1024), 16, 0, 0);
    };
    dma(0);
    dma(1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it < n_it; ++it) {
      dma(it + 2);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");   // stage it+1 has landed
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    uint4 fx[2][MTW], fw[2][RFW];
    auto rd = [&](int it, int s2, int set) {
      const unsigned char* sl_ = smem + (it % NSLOT) * SLOT;
#pragma unroll
      for (int t = 0; t < MTW; ++t) fx[set][t] = *(const uint4*)(sl_ + lds_off(wm * 64 + t * 16 + r, s2 * 4 + q));
#pragma unroll
      for (int f = 0; f < RFW; ++f) fw[set][f] = *(const uint4*)(sl_ + SLAB_X + ((wn * RFW + f) * 2 + s2) * 1024 + lane * 16);
    };
    auto mma = [&](int set) {
#pragma unroll
      for (int t = 0; t < MTW; ++t)
#pragma unroll
        for (int f = 0; f < RFW; ++f) MMA<T>::step(acc[f][t], fw[set][f], fx[set][t]);
      if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int i = 0; i < MTW + RFW; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
      }
    };
    __builtin_amdgcn_s_barrier();   // slab 0 has landed
    rd(0, 0, 0);
    for (int it = 0; it < n_it; ++it) {
      __builtin_amdgcn_sched_barrier(0);
      rd(it, 1, 1);
      mma(0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave is done with slab `it`
      __builtin_amdgcn_s_barrier();
      rd(it + 1, 0, 0);
      mma(1);
      __builtin_amdgcn_sched_barrier(0);
    }
    stream_finish<T, MTW, ACT, RFW>(p, sx, s, acc, ssum, false, mw, sp, fi0, nfrag, false, q, r);
  }
  if (!(s.splits > 1 && s.cnt)) return;

  // ---- in-kernel fix-up of a K split, OFF by default (SL_STREAM_FIXUP=1 turns it on; all twelve waves walk the same barriers,
  // the loaders do none of the work).  The block that arrives last at its tile's counter sums the S partial records — in split
  // order, its own included: bit for bit the sum gemm_stream_reduce_kernel forms — and applies the epilogue.
  // Measured at 512 rows (tools/tune_stream.py, bf16): with the separate reduce launch qkv / o / down take 31 / 26 / 40 us;
  // closed in-kernel 49 / 47-52 / 65-69 us with sc1 hand-off, 111 / 121 / 136 us with agent-scope fences (those write back and
  // invalidate the XCD's whole L2).  The partial records have to cross to the memory side either way, and the last block of
  // each of 48-80 tiles then pulls S x 128 KiB through ONE CU's miss path behind an atomic round trip, where the reduce launch
  // spreads the same bytes over every CU: on this chip the launch boundary is the cheaper synchronisation.  Kept as a tested
  // option (tests/test_kernels_gpu.py: identical stores to the reduce pass) for shapes where the balance may differ.
  const bool cw = wave < CW;
  const int tile = nb * mblocks + mblock;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial-record stores (sc1) have reached the memory side
  __builtin_amdgcn_s_barrier();
  if (tid == 0) {
    const int old = atomicAdd(&s.cnt[tile], 1);
    fx_last[0] = old == s.splits - 1;
    if (old == s.splits - 1) s.cnt[tile] = 0;   // every split has arrived: leave the counter as it was found
  }
  __builtin_amdgcn_s_barrier();
  if (!fx_last[0]) return;
  bool rowstat = false;
  if constexpr (ACT == SL_ACT_NONE) rowstat = sx.rstd_out != nullptr;
  if (cw) {
#pragma unroll
    for (int t = 0; t < MTW; ++t)
#pragma unroll
      for (int f = 0; f < RFW; ++f) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int sp2 = 0; sp2 < s.splits; ++sp2) {
      f32x4 v[MTW * RFW];
#pragma unroll
      for (int t = 0; t < MTW; ++t) {
        int m = mw + t * 16 + r; m = m < p.M ? m : p.M - 1;           // rows / fragments past the edge re-read the last one
#pragma unroll
        for (int f = 0; f < RFW; ++f) {
          const int fi = fi0 + f < nfrag ? fi0 + f : nfrag - 1;
          ld_sc1_16(v[t * RFW + f], s.part + ((int64_t)sp2 * p.M + m) * s.np + fi * 16 + 4 * q);
        }
      }
      vm_drain16(v);
#pragma unroll
      for (int t = 0; t < MTW; ++t)
#pragma unroll
        for (int f = 0; f < RFW; ++f) acc[f][t] += v[t * RFW + f];
    }
    if (!rowstat) {
      StreamX s1 = s;
      s1.splits = 1;
      stream_finish<T, MTW, ACT, RFW>(p, sx, s1, acc, ssum, false, mw, sp, fi0, nfrag, false, q, r);
    }
  }
  if (!rowstat) return;
  if constexpr (ACT == SL_ACT_NONE) {
    if (cw) {
      // plain (+bias, +residual) epilogue that keeps the stored values for the row statistics (as gemm_stream_reduce_kernel<ROWSTAT>)
      const T* bias = (const T*)p.bias;
#pragma unroll
      for (int t = 0; t < MTW; ++t) {
        const int m = mw + t * 16 + r;
        float sq = 0.f;
        if (m < p.M) {
#pragma unroll
          for (int f = 0; f < RFW; ++f) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int col = (fi0 + f) * 16 + 4 * q + i;
              if (col >= p.N) continue;
              float v = acc[f][t][i];
              if (bias) v += to_f32(bias[col]);
              if (p.res) v += p.res_f32 ? ((const float*)p.res)[(int64_t)m * p.ldr + col] : to_f32(((const T*)p.res)[(int64_t)m * p.ldr + col]);
              if (p.out_f32) {
                ((float*)p.C)[(int64_t)m * p.ldc + col] = v;
              } else {
                const T o = from_f32<T>(v);
                ((T*)p.C)[(int64_t)m * p.ldc + col] = o;
                v = to_f32(o);
              }
              sq = fmaf(v, v, sq);
            }
          }
        }
        sq += __shfl_xor(sq, 16);
        sq += __shfl_xor(sq, 32);
        if (q == 0) fx_ss[wm][wn][t * 16 + r] = sq;
      }
    }
    // second level: this tile's per-row sums over its 128 columns -> ss_part; the block that completes a row block's last tile
    // adds the n-block partials in n-block order (deterministic) and writes rstd_out for the 256 rows
    const int nbv = (nfrag + BF - 1) / BF;
    __builtin_amdgcn_s_barrier();
    if (tid < 256) {
      const int m = m0 + tid;
      if (m < p.M)
        __hip_atomic_store(&s.ss_part[(int64_t)nb * p.M + m], fx_ss[tid >> 6][0][tid & 63] + fx_ss[tid >> 6][1][tid & 63], __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid == 0) {
      const int old = atomicAdd(&s.cnt[1024 + mblock], 1);
      fx_last[1] = old == nbv - 1;
      if (old == nbv - 1) s.cnt[1024 + mblock] = 0;
    }
    __builtin_amdgcn_s_barrier();
    if (!fx_last[1]) return;
    if (tid < 256) {
      const int m = m0 + tid;
      if (m < p.M) {
        float t = 0.f;
        for (int j = 0; j < nbv; ++j) t += __hip_atomic_load(&s.ss_part[(int64_t)j * p.M + m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sx.rstd_out[m] = rsqrtf(t / (float)p.N + sx.eps);
      }
    }
  }
}

// sums the K-split partials and applies the epilogue: one thread per (row, fragment or fragment pair, 4-column group).
// ROWSTAT: one block per output row (blockDim = 4 * fragments <= 1024); the block also reduces the squares of the values
// it stores (as rounded to T) and emits the row's RMSNorm scale for the GEMM that consumes these rows next.
template <typename T, int ACT, bool ROWSTAT>
__global__ __launch_bounds__(ROWSTAT ? 1024 : 256) void gemm_stream_reduce_kernel(GemmP p, SkinnyX sx, StreamX s) {
  constexpr bool PAIRS = (ACT == SL_ACT_SILU_MUL || ACT == SL_ACT_ROPE_KV);
  static_assert(!ROWSTAT || ACT == SL_ACT_NONE, "row statistics ride on the plain (+residual) epilogue");
  const int nfrag = (p.N + 15) >> 4;
  const int nunits = PAIRS ? (nfrag + 1) / 2 : nfrag;
  int q, unit, m;
  if constexpr (ROWSTAT) {
    m = blockIdx.x; q = threadIdx.x & 3; unit = threadIdx.x >> 2;
  } else {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    q = (int)(idx & 3);
    unit = (int)((idx >> 2) % nunits);
    m = (int)(idx / (4 * (int64_t)nunits));
    if (m >= p.M) return;
  }
  const bool live = unit < nunits;
  const int gf = PAIRS ? 2 * unit : unit;
  f32x4 a4 = {0.f, 0.f, 0.f, 0.f}, b4 = {0.f, 0.f, 0.f, 0.f};
  float ssum = 0.f;
  const bool stats_here = sx.fuse_rms && !sx.rstd_in;
  if (live) {
    // the partial records of up to 8 splits are requested together (a load-add chain per split left one 16-byte load in
    // flight per thread); summed in split order
    constexpr int UN = 8;
    for (int sp0 = 0; sp0 < s.splits; sp0 += UN) {
      f32x4 va[UN], vb[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int sp = sp0 + u < s.splits ? sp0 + u : s.splits - 1;
        const float* row = s.part + ((int64_t)sp * p.M + m) * s.np + gf * 16 + 4 * q;
        va[u] = *(const f32x4*)row;
        if constexpr (PAIRS) vb[u] = *(const f32x4*)(row + 16);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (sp0 + u < s.splits) {
          a4 += va[u];
          if constexpr (PAIRS) b4 += vb[u];
        }
      }
    }
    if (stats_here)
      for (int sp = 0; sp < s.splits; ++sp) ssum += s.part_ss[(int64_t)sp * p.M + m];
  }
  const float rs = sx.rstd_in ? sx.rstd_in[m] : (stats_here ? rsqrtf(ssum / (float)p.K + sx.eps) : 1.0f);
  float a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = a4[i] * rs; b[i] = b4[i] * rs; }
  if constexpr (ROWSTAT) {
    // same arithmetic as stream_epilogue4's plain branch, keeping the stored values for the statistics
    __shared__ float wsum[17];
    float sq = 0.f;
    float kept[4] = {0.f, 0.f, 0.f, 0.f};      // the values as stored, for the normalised copy (sx.norm_out)
    if (live) {
      const T* bias = (const T*)p.bias;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = gf * 16 + 4 * q + i;
        if (col >= p.N) continue;
        float v = a[i];
        if (bias) v += to_f32(bias[col]);
        if (p.res) v += p.res_f32 ? ((const float*)p.res)[(int64_t)m * p.ldr + col] : to_f32(((const T*)p.res)[(int64_t)m * p.ldr + col]);
        if (p.out_f32) {
          ((float*)p.C)[(int64_t)m * p.ldc + col] = v;
        } else {
          const T o = from_f32<T>(v);
          ((T*)p.C)[(int64_t)m * p.ldc + col] = o;
          v = to_f32(o);
        }
        kept[i] = v;
        sq = fmaf(v, v, sq);
      }
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += wsum[w];
      const float rstd = rsqrtf(t / (float)p.N + sx.eps);
      sx.rstd_out[m] = rstd;
      wsum[16] = rstd;
    }
    if (sx.norm_out) {
      // the RMS-normalised row beside the row itself, in HF's rounding order: weight * (x * rstd).to(dtype) (hf:...llama.py:60-71) —
      // what a separate sl_rmsnorm launch over these rows would write (11 us per 1 024 x 3 072 in the decode graph)
      __syncthreads();
      const float rstd = wsum[16];
      if (live) {
        const int col0 = gf * 16 + 4 * q;
        if (col0 + 3 < p.N) {
          float o[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) o[i] = to_f32(((const T*)sx.norm_gain)[col0 + i]) * to_f32(from_f32<T>(kept[i] * rstd));
          T* dst = (T*)sx.norm_out + (int64_t)m * p.N + col0;        // 8 (bf16) / 16 (f32) bytes per lane, 64 / 128 contiguous bytes per fragment
          if constexpr (sizeof(T) == 2) *(uint2*)dst = make_uint2(pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]));
          else *(f32x4*)dst = f32x4{o[0], o[1], o[2], o[3]};
        } else {
          for (int i = 0; i < 4 && col0 + i < p.N; ++i)
            ((T*)sx.norm_out)[(int64_t)m * p.N + col0 + i] = from_f32<T>(to_f32(((const T*)sx.norm_gain)[col0 + i]) * to_f32(from_f32<T>(kept[i] * rstd)));
        }
      }
    }
  } else {
    if (live) stream_epilogue4<T, ACT>(p, sx, m, gf, 4 * q, a, b);
  }
}

// ----------------------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------------------
struct StreamCfg { int mt, splits, nwv, d, nl; int wide, wsplits; };   // wide: the 256 x 128 form applies (its own K-split count)

// Measured on MI355X (tools/tune_stream.py, bf16, M = 64..256): one CU pulls at most ~23 GB/s from HBM (plus the x
// slabs it re-reads from L2 through the same miss queue) whatever the prefetch depth, so the only lever is how many
// CUs stream.  Wide matrices (>= 512 blocks of 128 weight rows: lm_head) use 4 compute waves x 32 rows; the layer
// projections use 2 compute waves x 32 rows (64-row blocks) and split K until ~200 blocks exist — beyond that the
// fp32 partials (written, then re-read by the reduce kernel) cost more than the extra CUs bring.
static StreamCfg stream_cfg(int M, int N, int K, int kstep, bool have_ws, int dtype) {
  StreamCfg c;
  // above 128 rows: 128-row blocks x 4 compute waves (128 weight rows): the row blocks of an n-block re-read its weights
  // from L2, and each CU stages half the activation bytes of a 256-row block (measured at M = 256: gate/up 38 vs 50 us)
  c.mt = M <= 32 ? 2 : (M <= 64 ? 4 : 8);
  const int nfrag = (N + 15) / 16;
  const int nst = K / (2 * kstep);
  c.d = 4;
  c.nl = sl_env().stream_nl ? sl_env().stream_nl : 1;
  // tuning override "splits,nwv[,mt]" (tools/tune_stream.py), read once into sl_env()
  const int sp_env = sl_env().stream_splits, nwv_env = sl_env().stream_nwv;
  if (sl_env().stream_mt && M > 64) c.mt = sl_env().stream_mt;
  const int mblocks = (M + c.mt * 16 - 1) / (c.mt * 16);
  c.nwv = ((nfrag + 7) / 8 * mblocks >= 512 || M > 128) && c.mt <= 8 ? 4 : 2;   // 4 compute + 2 loader waves of 256-row blocks spill
  if (nwv_env && c.mt <= 8) c.nwv = nwv_env;
  const int base = mblocks * ((nfrag + c.nwv * 2 - 1) / (c.nwv * 2));
  int splits = 1;
  if (have_ws) {
    splits = 208 / base;
    const int max_splits = nst / 4 > 0 ? nst / 4 : 1;   // >= 4 stages (512 B of K per row) per split
    if (splits > max_splits) splits = max_splits;
    if (splits > 16) splits = 16;
    if (splits < 1) splits = 1;
  }
  if (sp_env && (have_ws || sp_env == 1)) splits = sp_env < nst ? sp_env : nst;
  c.splits = splits;
  // the 256 x 128 form (bf16, > 384 rows): unsplit when its blocks cover at least half the CUs, else K split towards ~240 blocks of >= 6 stages
  c.wide = 0; c.wsplits = 1;
  if (dtype == SL_BF16 && M > 384 && sl_env().stream_wide != 0) {
    const int wbase = ((M + 255) / 256) * ((nfrag + 7) / 8);
    if (wbase >= 128 || sl_env().stream_wide == 2) {   // unsplit from half the CUs up (qkv at 1024 rows: 160 blocks, 53 us; 61 us split in two)
      c.wide = 1;
    }
    if (wbase < 128 && have_ws) {
      int ws_ = 240 / wbase;
      const int max_ws = nst / 6 > 0 ? nst / 6 : 1;
      if (ws_ > max_ws) ws_ = max_ws;
      if (ws_ > 8) ws_ = 8;
      if (sl_env().stream_wsplits) ws_ = sl_env().stream_wsplits < nst ? sl_env().stream_wsplits : nst;
      if (ws_ > 1) { c.wide = 1; c.wsplits = ws_; }
    }
  }
  return c;
}

// the K-split workspace opens with the fix-up's counters (8 KiB, zero whenever no launch is in flight: zeroed once by the owner
// of the buffer, restored by every launch) and its row-statistics partials ([<= 32 n-blocks][M] floats); the partial records follow
static size_t stream_ws_header(int M) { return 8192 + (((size_t)32 * M * sizeof(float) + 255) & ~(size_t)255); }

int sl_gemm_stream_splits(int M, int N, int K, int dtype) {
  const StreamCfg c = stream_cfg(sl_family_rows(M), N, K, dtype == SL_F32 ? 16 : 32, true, dtype);
  return c.wide ? c.wsplits : c.splits;
}

size_t sl_gemm_stream_ws_bytes(int M, int N, int K, int dtype) {
  const size_t np = (size_t)((N + 15) / 16) * 16;
  const StreamCfg c = stream_cfg(sl_family_rows(M), N, K, dtype == SL_F32 ? 16 : 32, true, dtype);
  const int most = c.wide && c.wsplits > c.splits ? c.wsplits : c.splits;   // either form may run (the wide one not when the kernel takes the RMSNorm statistics itself)
  size_t splits = most > 1 ? (size_t)most : 0;
  return stream_ws_header(M) + splits * ((size_t)M * np + (size_t)M) * sizeof(float) + 256;
}

// second pass of a K-split launch: sums the partial records and applies the epilogue
template <typename T, int ACT>
static int launch_stream_reduce(GemmP& p, const SkinnyX& sx, const StreamX& s, hipStream_t st) {
  if (s.splits <= 1) return 0;
  constexpr bool PAIRS = (ACT == SL_ACT_SILU_MUL || ACT == SL_ACT_ROPE_KV);
  const int nfrag = (p.N + 15) / 16;
  const int64_t nunits = PAIRS ? (nfrag + 1) / 2 : nfrag;
  const int64_t threads = (int64_t)p.M * nunits * 4;
  if constexpr (ACT == SL_ACT_NONE) {
    if (sx.rstd_out) {   // checked by the caller: 4 * fragments <= 1024
      hipLaunchKernelGGL((gemm_stream_reduce_kernel<T, ACT, true>), dim3(p.M), dim3((unsigned)((nunits * 4 + 63) / 64 * 64)), 0, st, p, sx, s);
      SL_CHECK_LAUNCH("gemm_stream_reduce(rowstat)");
      return 0;
    }
  }
  hipLaunchKernelGGL((gemm_stream_reduce_kernel<T, ACT, false>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, p, sx, s);
  SL_CHECK_LAUNCH("gemm_stream_reduce");
  return 0;
}

template <typename T, int MT, int ACT, int RF, int NWV, int D, int NL>
static int launch_stream_cfg(GemmP& p, const SkinnyX& sx, const StreamX& s, hipStream_t st) {
  const int nfrag = (p.N + 15) / 16;
  const int nblocks = (nfrag + NWV * RF - 1) / (NWV * RF), mblocks = (p.M + MT * 16 - 1) / (MT * 16);
  dim3 grid((nblocks + 7) / 8 * 8 * mblocks, 1, s.splits);
  if (sx.fuse_rms != 0 && sx.rstd_in == nullptr)
    hipLaunchKernelGGL((gemm_stream_kernel<T, MT, ACT, RF, NWV, D, NL, true>), grid, dim3(64 * (NWV + NL)), 0, st, p, sx, s);
  else
    hipLaunchKernelGGL((gemm_stream_kernel<T, MT, ACT, RF, NWV, D, NL, false>), grid, dim3(64 * (NWV + NL)), 0, st, p, sx, s);
  SL_CHECK_LAUNCH("gemm_stream");
  return launch_stream_reduce<T, ACT>(p, sx, s, st);
}

template <typename T, int MT, int ACT>
static int launch_stream_mt(GemmP& p, const SkinnyX& sx, const StreamX& s, const StreamCfg& c, hipStream_t st) {
  if constexpr (sizeof(T) == 4) {
    return launch_stream_cfg<T, MT, ACT, 2, (MT > 8 ? 2 : 4), 2, (MT > 8 ? 2 : 1)>(p, sx, s, st);   // fp32 parity mode: one structure
  } else {
    if constexpr (MT > 8) {
      return launch_stream_cfg<T, MT, ACT, 2, 2, 4, 2>(p, sx, s, st);   // 256-row blocks: 4 compute + 2 loader waves would spill
    } else if constexpr (MT == 8) {
      if (c.nwv == 2) return launch_stream_cfg<T, MT, ACT, 2, 2, 4, 1>(p, sx, s, st);
      if (c.nl == 4) return launch_stream_cfg<T, MT, ACT, 2, 4, 4, 4>(p, sx, s, st);
      if (c.nl == 2) return launch_stream_cfg<T, MT, ACT, 2, 4, 4, 2>(p, sx, s, st);
      return launch_stream_cfg<T, MT, ACT, 2, 4, 4, 1>(p, sx, s, st);
    } else {
      if (c.nwv == 2) return launch_stream_cfg<T, MT, ACT, 2, 2, 4, 1>(p, sx, s, st);
      return launch_stream_cfg<T, MT, ACT, 2, 4, 4, 1>(p, sx, s, st);
    }
  }
}

template <typename T, int ACT>
static int launch_stream(GemmP& p, const SkinnyX& sx, const StreamX& s, const StreamCfg& c, hipStream_t st) {
  switch (c.mt) {
    case 2: return launch_stream_mt<T, 2, ACT>(p, sx, s, c, st);
    case 4: return launch_stream_mt<T, 4, ACT>(p, sx, s, c, st);
    case 8: return launch_stream_mt<T, 8, ACT>(p, sx, s, c, st);
    default: return launch_stream_mt<T, 16, ACT>(p, sx, s, c, st);
  }
}

template <typename T>
static int stream_typed(GemmP& p, const SkinnyX& sx, int act, const StreamX& s, const StreamCfg& c, hipStream_t st) {
  switch (act) {
    case SL_ACT_NONE: return launch_stream<T, SL_ACT_NONE>(p, sx, s, c, st);
    case SL_ACT_SILU_MUL: return launch_stream<T, SL_ACT_SILU_MUL>(p, sx, s, c, st);
    case SL_ACT_ROPE_KV: return launch_stream<T, SL_ACT_ROPE_KV>(p, sx, s, c, st);
  }
  sl_set_error("sl_gemm: packed weights are not built with act %d for M > 16 streaming", act);
  return SL_ERR_UNSUPPORTED;
}

// the 256 x 128 form applies to this call: the shape qualifies (stream_cfg) and the kernel does not have to take the RMSNorm
// statistics itself (row scales, if any, come from the producer)
static bool stream_wide_ok(const SkinnyX& sx, int act, const StreamCfg& c, bool have_ws) {
  if (!c.wide || (sx.fuse_rms != 0 && sx.rstd_in == nullptr)) return false;
  if (act != SL_ACT_NONE && act != SL_ACT_SILU_MUL && act != SL_ACT_ROPE_KV) return false;
  return c.wsplits == 1 || have_ws;
}

template <int ACT>
static int launch_stream_wide_act(GemmP& p, const SkinnyX& sx, const StreamX& s, hipStream_t st) {
  const int nfrag = (p.N + 15) / 16, mblocks = (p.M + 255) / 256, nblocks = (nfrag + 7) / 8;
  constexpr int LDS_BYTES = 3 * (256 * TROWB + 8 * 2 * 1024);
  dim3 grid((nblocks + 7) / 8 * 8 * mblocks, 1, s.splits);
  static std::atomic<uint64_t> attr_set{0};   // one bit per device: the opt-in to 144 KiB of dynamic LDS is per device
  int devid = 0;
  SL_HIP(hipGetDevice(&devid));
  if (devid < 0 || devid >= 64 || !((attr_set.load(std::memory_order_relaxed) >> devid) & 1)) {
    SL_HIP(hipFuncSetAttribute((const void*)gemm_stream_wide_kernel<bf16_t, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    if (devid >= 0 && devid < 64) attr_set.fetch_or(1ull << devid, std::memory_order_relaxed);
  }
  hipLaunchKernelGGL((gemm_stream_wide_kernel<bf16_t, ACT>), grid, dim3(768), LDS_BYTES, st, p, sx, s);
  SL_CHECK_LAUNCH("gemm_stream_wide");
  if (s.cnt) return 0;   // the kernel's last-arriving blocks did the reduce pass
  return launch_stream_reduce<bf16_t, ACT>(p, sx, s, st);
}

static int launch_stream_wide(GemmP& p, const SkinnyX& sx, int act, const StreamX& s, hipStream_t st) {
  switch (act) {
    case SL_ACT_SILU_MUL: return launch_stream_wide_act<SL_ACT_SILU_MUL>(p, sx, s, st);
    case SL_ACT_ROPE_KV: return launch_stream_wide_act<SL_ACT_ROPE_KV>(p, sx, s, st);
    default: return launch_stream_wide_act<SL_ACT_NONE>(p, sx, s, st);
  }
}

int sl_gemm_stream_launch(GemmP& p, const SkinnyX& sx, int dtype, int act, void* split_ws, size_t split_ws_bytes, hipStream_t st) {
  const int kstep = dtype == SL_F32 ? 16 : 32;
  SL_CHECK_ARG(p.K % (2 * kstep) == 0, "sl_gemm: streaming path needs K %% %d == 0 (K=%d)", 2 * kstep, p.K);
  StreamCfg c = stream_cfg(sl_family_rows(p.M), p.N, p.K, kstep, split_ws != nullptr, dtype);      // the block shape and K-split of the pinned family (common.h)
  StreamX s;
  const bool wide = stream_wide_ok(sx, act, c, split_ws != nullptr);
  if (wide) c.splits = c.wsplits;

  s.np = (p.N + 15) / 16 * 16;
  const size_t per_split = ((size_t)p.M * s.np + (size_t)p.M) * sizeof(float);
  const size_t hdr = stream_ws_header(p.M);
  split_ws_bytes = (split_ws && split_ws_bytes > hdr) ? split_ws_bytes - hdr : 0;   // what is left for partial records
  if (c.splits > 1 && (size_t)c.splits * per_split > split_ws_bytes) c.splits = (int)(split_ws_bytes / per_split);
  if (c.splits < 1) c.splits = 1;
  const int nst = p.K / (2 * kstep);
  s.sps = (nst + c.splits - 1) / c.splits;
  s.splits = (nst + s.sps - 1) / s.sps;   // no empty splits
  c.splits = s.splits;
  SL_CHECK_ARG(sx.norm_out == nullptr || sx.rstd_out != nullptr, "sl_gemm: norm_out comes from the pass that forms rstd_out (pass both)");
  if (sx.rstd_out) {
    SL_CHECK_ARG(act == SL_ACT_NONE && (p.N + 15) / 16 * 4 <= 1024 && s.splits > 1,
                 "sl_gemm: rstd_out needs the plain epilogue, N <= 4096 and a K-split (splits=%d; see sl_gemm_split_count)", s.splits);
  }
  s.part = split_ws ? (float*)((unsigned char*)split_ws + hdr) : nullptr;
  s.part_ss = s.part ? s.part + (size_t)s.splits * p.M * s.np : nullptr;
  // in-kernel fix-up (wide form): the tile / row-block counters and the row-statistics partials must fit the header
  s.cnt = nullptr; s.ss_part = nullptr;
  // (norm_out is written by the reduce launch only: a call that asks for it keeps that launch)
  if (wide && s.splits > 1 && sl_env().stream_fixup != 0 && sx.norm_out == nullptr) {
    const int nbv = ((p.N + 15) / 16 + 7) / 8, mbl = (p.M + 255) / 256;
    if (nbv * mbl <= 1024 && mbl <= 64 && (sx.rstd_out == nullptr || nbv <= 32)) {
      s.cnt = (int*)split_ws;
      s.ss_part = (float*)((unsigned char*)split_ws + 8192);
    }
  }
  if (dtype == SL_F32) return stream_typed<float>(p, sx, act, s, c, st);
  if (wide) return launch_stream_wide(p, sx, act, s, st);
  return stream_typed<bf16_t>(p, sx, act, s, c, st);
}
