// gemm.hip — C = act(A . W^T + bias) + residual for gfx950, bf16 (v_mfma_f32_16x16x32_bf16) and exact
// fp32 (v_mfma_f32_16x16x4_f32).  Two kernels:
//   * gemm_tiled_kernel : 128x128 block tile, 4 waves (2x2, 64x64 each), 128-byte K slabs staged
//     through XOR-swizzled LDS (conflict-free ds_read_b128), register double-buffering of the global
//     loads.  MFMA-bound shapes: HuBERT conv-as-GEMM, encoder layers, Llama prefill.
//   * gemm_skinny_kernel: M <= 64 rows (KV-cached decode, M = batch).  HBM-bound weight streaming:
//     one block owns 16 (or 32 gate/up) weight rows, its waves interleave 64-byte K steps, weight
//     fragments go global -> VGPR -> MFMA with no LDS round trip, partial sums meet in LDS once.
// A rows may overlap (lda < K): that is how the strided convolutions run without im2col.
#include "common.h"

struct GemmP {
  const void* A; int64_t lda, sA;
  const void* W; int64_t ldw, sW;
  void* C; int64_t ldc, sC;
  const void* bias; int64_t sBias;
  const void* res; int64_t ldr, sR;
  int M, N, K, out_f32;
  int tiles_m, tiles_n;
};

// ----------------------------------------------------------------------------------------------
// epilogue helper: +bias, act, +residual, store (T or float)
// ----------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void store_out(const GemmP& p, void* Cb, const void* Rb, int64_t row, int64_t col, float v) {
  if (Rb) v += to_f32(((const T*)Rb)[row * p.ldr + col]);
  if (p.out_f32)
    ((float*)Cb)[row * p.ldc + col] = v;
  else
    ((T*)Cb)[row * p.ldc + col] = from_f32<T>(v);
}

// ----------------------------------------------------------------------------------------------
// tiled kernel
// ----------------------------------------------------------------------------------------------
constexpr int TBM = 128, TBN = 128, TROWB = 128;  // tile rows, tile cols, bytes of K per LDS row

// byte offset of 16-byte chunk `ch` (0..7) of tile row `row` in a [128][128 B] swizzled LDS tile
__device__ __forceinline__ int lds_off(int row, int ch) { return row * TROWB + ((ch ^ (row & 7)) << 4); }

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
  const int bn = bid / p.tiles_m, bm = bid - bn * p.tiles_m;
  const int z = blockIdx.y;

  const T* A = (const T*)p.A + (int64_t)z * p.sA;
  const T* W = (const T*)p.W + (int64_t)z * p.sW;

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
      ra[i] = ok ? *(const uint4*)(ga[i] + k0) : make_uint4(0, 0, 0, 0);
      rw[i] = ok ? *(const uint4*)(gw[i] + k0) : make_uint4(0, 0, 0, 0);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(uint4*)(&smem[buf][0][so[i]]) = ra[i];
      *(uint4*)(&smem[buf][1][so[i]]) = rw[i];
    }
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
#pragma unroll
      for (int m = 0; m < 4; ++m) fa[m] = *(const uint4*)(sa + lds_off(wm * 64 + m * 16 + r, s * 4 + q));
#pragma unroll
      for (int n = 0; n < 4; ++n) fb[n] = *(const uint4*)(sw + lds_off(wn * 64 + n * 16 + r, s * 4 + q));
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) MMA<T>::step(acc[m][n], fa[m], fb[n]);
    }
    if (kt + 1 < nkt) sstore(buf ^ 1);
    __syncthreads();
  }

  // epilogue
  void* Cb = p.out_f32 ? (void*)((float*)p.C + (int64_t)z * p.sC) : (void*)((T*)p.C + (int64_t)z * p.sC);
  const T* bias = p.bias ? (const T*)p.bias + (int64_t)z * p.sBias : nullptr;
  const void* Rb = p.res ? (const void*)((const T*)p.res + (int64_t)z * p.sR) : nullptr;
  const int row0 = bm * TBM + wm * 64 + q * 4;
  const int col0 = bn * TBN + wn * 64 + r;
  if constexpr (ACT == SL_ACT_SILU_MUL) {
    const int nout = p.N >> 1;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      const int gcol = col0 + (2 * pr) * 16, ucol = gcol + 16;
      const int ocol = ((bn * TBN + wn * 64) >> 1) + pr * 16 + r;
      if (ocol >= nout) continue;
      const float bg = bias ? to_f32(bias[gcol]) : 0.f, bu = bias ? to_f32(bias[ucol]) : 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + m * 16 + i;
          if (row < p.M) store_out<T>(p, Cb, Rb, row, ocol, silu(acc[m][2 * pr][i] + bg) * (acc[m][2 * pr + 1][i] + bu));
        }
    }
  } else {
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int col = col0 + n * 16;
      if (col >= p.N) continue;
      const float b = bias ? to_f32(bias[col]) : 0.f;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + m * 16 + i;
          if (row < p.M) {
            float v = acc[m][n][i] + b;
            if constexpr (ACT == SL_ACT_GELU) v = gelu_erf(v);
            store_out<T>(p, Cb, Rb, row, col, v);
          }
        }
    }
  }
}

// ----------------------------------------------------------------------------------------------
// skinny kernel (M <= 16*MT)
// ----------------------------------------------------------------------------------------------
template <typename T, int MT, int ACT, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmP p) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int KSTEP = MMA<T>::KSTEP;
  constexpr int RF = (ACT == SL_ACT_SILU_MUL) ? 2 : 1;  // 16-row weight fragments per block
  constexpr int RB = 16 * RF;
  constexpr int U = 4;  // k-steps in flight per wave
  __shared__ float red[NW][RB][MT * 16 + 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int z = blockIdx.y;
  const int n0 = blockIdx.x * RB;

  const T* A = (const T*)p.A + (int64_t)z * p.sA;
  const T* W = (const T*)p.W + (int64_t)z * p.sW;

  const T* wp[RF];
#pragma unroll
  for (int f = 0; f < RF; ++f) {
    int wr = n0 + f * 16 + r; wr = wr < p.N ? wr : p.N - 1;
    wp[f] = W + (int64_t)wr * p.ldw + q * VEC;
  }
  const T* xp[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int xr = t * 16 + r; xr = xr < p.M ? xr : p.M - 1;
    xp[t] = A + (int64_t)xr * p.lda + q * VEC;
  }

  f32x4 acc[RF][MT];
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nks_full = p.K / KSTEP;  // full 64-byte steps
  int ks = wave;
  // main: U interleaved steps per iteration, all loads issued before the first MFMA
  for (; ks + (U - 1) * NW < nks_full; ks += U * NW) {
    uint4 fw[U][RF], fx[U][MT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = (int64_t)(ks + u * NW) * KSTEP;
#pragma unroll
      for (int f = 0; f < RF; ++f) fw[u][f] = ld_nt16(wp[f] + k);
#pragma unroll
      for (int t = 0; t < MT; ++t) fx[u][t] = *(const uint4*)(xp[t] + k);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int t = 0; t < MT; ++t) MMA<T>::step(acc[f][t], fw[u][f], fx[u][t]);
  }
  for (; ks < nks_full; ks += NW) {
    const int64_t k = (int64_t)ks * KSTEP;
    uint4 fw[RF], fx[MT];
#pragma unroll
    for (int f = 0; f < RF; ++f) fw[f] = ld_nt16(wp[f] + k);
#pragma unroll
    for (int t = 0; t < MT; ++t) fx[t] = *(const uint4*)(xp[t] + k);
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
      for (int t = 0; t < MT; ++t) MMA<T>::step(acc[f][t], fw[f], fx[t]);
  }
  // K tail (K % KSTEP != 0): one predicated step, taken by the wave whose turn it is
  if (nks_full * KSTEP < p.K && (nks_full % NW) == wave) {
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
  }

  // D[weight row 4q+i][x row r]  ->  red[wave][f*16 + 4q+i][t*16 + r]
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave][f * 16 + q * 4 + i][t * 16 + r] = acc[f][t][i];
  __syncthreads();

  void* Cb = p.out_f32 ? (void*)((float*)p.C + (int64_t)z * p.sC) : (void*)((T*)p.C + (int64_t)z * p.sC);
  const T* bias = p.bias ? (const T*)p.bias + (int64_t)z * p.sBias : nullptr;
  const void* Rb = p.res ? (const void*)((const T*)p.res + (int64_t)z * p.sR) : nullptr;
  for (int o = tid; o < 16 * p.M; o += NW * 64) {
    const int m = o >> 4, n = o & 15;
    if constexpr (ACT == SL_ACT_SILU_MUL) {
      const int ocol = blockIdx.x * 16 + n;
      if (ocol >= (p.N >> 1)) continue;
      float g = 0.f, u = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) { g += red[w][n][m]; u += red[w][16 + n][m]; }
      if (bias) { g += to_f32(bias[n0 + n]); u += to_f32(bias[n0 + 16 + n]); }
      store_out<T>(p, Cb, Rb, m, ocol, silu(g) * u);
    } else {
      const int col = n0 + n;
      if (col >= p.N) continue;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[w][n][m];
      if (bias) v += to_f32(bias[col]);
      if constexpr (ACT == SL_ACT_GELU) v = gelu_erf(v);
      store_out<T>(p, Cb, Rb, m, col, v);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// host dispatch
// ----------------------------------------------------------------------------------------------
template <typename T, int ACT>
static int launch_tiled(GemmP& p, int batch, hipStream_t st) {
  p.tiles_m = (p.M + TBM - 1) / TBM;
  p.tiles_n = (p.N + TBN - 1) / TBN;
  dim3 grid(p.tiles_m * p.tiles_n, batch);
  hipLaunchKernelGGL((gemm_tiled_kernel<T, ACT>), grid, dim3(256), 0, st, p);
  SL_CHECK_LAUNCH("gemm_tiled");
  return 0;
}

template <typename T, int MT, int ACT>
static int launch_skinny_mt(GemmP& p, int batch, hipStream_t st) {
  const int RB = (ACT == SL_ACT_SILU_MUL) ? 32 : 16;
  const int nblk = (p.N + RB - 1) / RB;
  dim3 grid(nblk, batch);
  // few row blocks (N = hidden): spread K over 8 waves so the chip still has enough loads in flight
  if (nblk * batch < 1024 && p.K >= 2048)
    hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, ACT, 8>), grid, dim3(512), 0, st, p);
  else
    hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, ACT, 4>), grid, dim3(256), 0, st, p);
  SL_CHECK_LAUNCH("gemm_skinny");
  return 0;
}

template <typename T, int ACT>
static int launch_skinny(GemmP& p, int batch, hipStream_t st) {
  if (p.M <= 16) return launch_skinny_mt<T, 1, ACT>(p, batch, st);
  if (p.M <= 32) return launch_skinny_mt<T, 2, ACT>(p, batch, st);
  return launch_skinny_mt<T, 4, ACT>(p, batch, st);
}

template <typename T>
static int gemm_typed(const sl_gemm_args* a, GemmP& p, hipStream_t st) {
  const bool skinny = a->M <= 64;
  switch (a->act) {
    case SL_ACT_NONE: return skinny ? launch_skinny<T, SL_ACT_NONE>(p, a->batch, st) : launch_tiled<T, SL_ACT_NONE>(p, a->batch, st);
    case SL_ACT_GELU: return skinny ? launch_skinny<T, SL_ACT_GELU>(p, a->batch, st) : launch_tiled<T, SL_ACT_GELU>(p, a->batch, st);
    case SL_ACT_SILU_MUL: return skinny ? launch_skinny<T, SL_ACT_SILU_MUL>(p, a->batch, st) : launch_tiled<T, SL_ACT_SILU_MUL>(p, a->batch, st);
  }
  sl_set_error("sl_gemm: unknown act %d", a->act);
  return SL_ERR_ARG;
}

extern "C" int sl_gemm(const sl_gemm_args* a, sl_stream stream) {
  SL_CHECK_ARG(a != nullptr, "sl_gemm: null args");
  SL_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0 && a->batch > 0, "sl_gemm: bad shape M=%d N=%d K=%d batch=%d", a->M, a->N, a->K, a->batch);
  SL_CHECK_ARG(a->dtype == SL_F32 || a->dtype == SL_BF16, "sl_gemm: bad dtype %d", a->dtype);
  const int vec = a->dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(a->K % vec == 0, "sl_gemm: K=%d must be a multiple of %d", a->K, vec);
  SL_CHECK_ARG(a->lda % vec == 0 && a->ldw % vec == 0 && a->strideA % vec == 0 && a->strideW % vec == 0,
               "sl_gemm: lda/ldw/strides must keep rows 16-byte aligned");
  SL_CHECK_ARG(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0, "sl_gemm: A and W must be 16-byte aligned");
  if (a->act == SL_ACT_SILU_MUL) SL_CHECK_ARG(a->N % 32 == 0, "sl_gemm: SILU_MUL needs N %% 32 == 0 (16-row gate/up blocks)");
  GemmP p;
  p.A = a->A; p.lda = a->lda; p.sA = a->strideA;
  p.W = a->W; p.ldw = a->ldw; p.sW = a->strideW;
  p.C = a->C; p.ldc = a->ldc; p.sC = a->strideC;
  p.bias = a->bias; p.sBias = a->strideBias;
  p.res = a->residual; p.ldr = a->ldr; p.sR = a->strideR;
  p.M = a->M; p.N = a->N; p.K = a->K; p.out_f32 = a->out_f32;
  p.tiles_m = p.tiles_n = 0;
  hipStream_t st = (hipStream_t)stream;
  if (a->dtype == SL_F32) return gemm_typed<float>(a, p, st);
  return gemm_typed<bf16_t>(a, p, st);
}
