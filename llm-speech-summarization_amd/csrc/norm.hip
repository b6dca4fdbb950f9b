// norm.hip — LayerNorm (+optional GELU) and Llama RMSNorm.  HBM-bound row kernels: one wave per row,
// 16-byte loads, the whole row lives in registers between the statistics and the write (one read, one
// write of HBM per element).  Statistics in fp32 as the reference's fp32 / autocast-to-fp32 path does.
#include <stdlib.h>
#include "common.h"

constexpr int NORM_MAXF = 64;  // floats per lane -> rows of up to 4096 elements

template <typename T, bool RMS>
__global__ __launch_bounds__(256) void norm_rows_kernel(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ g,
                                                        const T* __restrict__ b, int64_t rows, int cols, float eps, int gelu,
                                                        float* __restrict__ rstd_out = nullptr) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int MAXCH = NORM_MAXF / VEC;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = cols / VEC;  // 16-byte chunks per row
  const T* xr = x + row * cols;
  T* yr = y + row * cols;

  float v[MAXCH][VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      const uint4 u = *(const uint4*)(xr + ch * VEC);
      Vec16<T>::unpack(u, v[i]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) s += RMS ? v[i][e] * v[i][e] : v[i][e];
    }
  }
  s = wave_sum(s);
  float mean = 0.f, rstd;
  if constexpr (RMS) {
    rstd = rsqrtf(s / (float)cols + eps);
  } else {
    mean = s / (float)cols;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { const float d = v[i][e] - mean; s2 += d * d; }
      }
    }
    s2 = wave_sum(s2);
    rstd = rsqrtf(s2 / (float)cols + eps);
  }
  if (rstd_out && lane == 0) rstd_out[row] = rstd;     // the scale alone, for a consumer that folds the gain into its weights
  if (y == nullptr) return;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      float gg[VEC], bb[VEC], o[VEC];
      Vec16<T>::unpack(*(const uint4*)(g + ch * VEC), gg);
      if constexpr (!RMS) Vec16<T>::unpack(*(const uint4*)(b + ch * VEC), bb);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        if constexpr (RMS) {
          // hf:models/llama/modeling_llama.py:66-67: normalised value is cast to the input dtype first
          const float t = to_f32(from_f32<T>(v[i][e] * rstd));
          o[e] = gg[e] * t;
        } else {
          float t = (v[i][e] - mean) * rstd * gg[e] + bb[e];
          o[e] = gelu ? gelu_act<T>(t) : t;
        }
      }
      *(uint4*)(yr + ch * VEC) = Vec16<T>::pack(o);
    }
  }
}

// Short rows (cols = CH * 64 lanes * 16 bytes: 512 / 1024 bf16 — the conv stack's and the encoder's LayerNorms): one
// row per wave leaves 1-2 KiB in flight per wave and pays two dependent 6-step wave reductions, the gain/bias loads
// and the wave launch per row (2.5 TB/s at 512 columns).  Here a wave keeps gain / bias in registers, walks the rows
// in groups of R with all R rows requested up front, and runs the R reductions interleaved.
template <typename T, bool RMS, int CH, int R>
__global__ __launch_bounds__(256) void norm_rows_multi_kernel(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ g,
                                                              const T* __restrict__ b, int64_t rows, float eps, int gelu) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int cols = CH * 64 * VEC;
  const int lane = threadIdx.x & 63;
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (int64_t)gridDim.x * 4;
  float gg[CH][VEC], bb[CH][VEC];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    Vec16<T>::unpack(*(const uint4*)(g + (lane + 64 * c) * VEC), gg[c]);
    if constexpr (!RMS) Vec16<T>::unpack(*(const uint4*)(b + (lane + 64 * c) * VEC), bb[c]);
  }
  for (int64_t r0 = wave_id * R; r0 < rows; r0 += n_waves * R) {
    float v[R][CH][VEC];
    float s[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int64_t row = r0 + j < rows ? r0 + j : rows - 1;
#pragma unroll
      for (int c = 0; c < CH; ++c) Vec16<T>::unpack(ld_nt16(x + row * cols + (lane + 64 * c) * VEC), v[j][c]);   // streamed once
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      s[j] = 0.f;
#pragma unroll
      for (int c = 0; c < CH; ++c)
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[j] += RMS ? v[j][c][e] * v[j][c][e] : v[j][c][e];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
      for (int j = 0; j < R; ++j) s[j] += __shfl_xor(s[j], o, 64);
    float mean[R], rstd[R];
    if constexpr (RMS) {
#pragma unroll
      for (int j = 0; j < R; ++j) { mean[j] = 0.f; rstd[j] = rsqrtf(s[j] / (float)cols + eps); }
    } else {
      float s2[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        mean[j] = s[j] / (float)cols;
        s2[j] = 0.f;
#pragma unroll
        for (int c = 0; c < CH; ++c)
#pragma unroll
          for (int e = 0; e < VEC; ++e) { const float d = v[j][c][e] - mean[j]; s2[j] += d * d; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int j = 0; j < R; ++j) s2[j] += __shfl_xor(s2[j], o, 64);
#pragma unroll
      for (int j = 0; j < R; ++j) rstd[j] = rsqrtf(s2[j] / (float)cols + eps);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) {
      if (r0 + j >= rows) break;
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if constexpr (RMS) {
            const float t = to_f32(from_f32<T>(v[j][c][e] * rstd[j]));
            o[e] = gg[c][e] * t;
          } else {
            const float t = (v[j][c][e] - mean[j]) * rstd[j] * gg[c][e] + bb[c][e];
            o[e] = gelu ? gelu_act<T>(t) : t;
          }
        }
        const uint4 pk = Vec16<T>::pack(o);
        __builtin_nontemporal_store(u32x4_t{pk.x, pk.y, pk.z, pk.w}, (u32x4_t*)(y + (r0 + j) * cols + (lane + 64 * c) * VEC));
      }
    }
  }
}

template <typename T, bool RMS, int CH, int R>
static int launch_norm_multi(const void* x, void* y, const void* g, const void* b, int64_t rows, float eps, int gelu, hipStream_t st) {
  const int64_t groups = ceil_div64(rows, R);
  const unsigned grid = (unsigned)(ceil_div64(groups, 4) < 8192 ? ceil_div64(groups, 4) : 8192);   // <= 32 blocks per CU, rows strided over waves
  hipLaunchKernelGGL((norm_rows_multi_kernel<T, RMS, CH, R>), dim3(grid), dim3(256), 0, st, (const T*)x, (T*)y, (const T*)g, (const T*)b, rows,
                     eps, gelu);
  SL_CHECK_LAUNCH("norm_rows_multi");
  return 0;
}

template <typename T, bool RMS>
static int launch_norm(const void* x, void* y, const void* g, const void* b, int64_t rows, int cols, float eps, int gelu,
                       hipStream_t st) {
  constexpr int VEC = Vec16<T>::VEC;
  SL_CHECK_ARG(cols % VEC == 0 && cols <= 64 * NORM_MAXF, "norm: cols=%d must be a multiple of %d and <= %d", cols, VEC, 64 * NORM_MAXF);
  if (rows == 0) return 0;
  const int single = sl_env().norm_single_row;   // A/B switch
  if (!single && rows >= 4096) {
    if (cols == 64 * VEC) return launch_norm_multi<T, RMS, 1, 8>(x, y, g, b, rows, eps, gelu, st);
    if (cols == 128 * VEC) return launch_norm_multi<T, RMS, 2, 4>(x, y, g, b, rows, eps, gelu, st);
  }
  hipLaunchKernelGGL((norm_rows_kernel<T, RMS>), dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, st, (const T*)x, (T*)y,
                     (const T*)g, (const T*)b, rows, cols, eps, gelu);
  SL_CHECK_LAUNCH("norm_rows");
  return 0;
}

extern "C" int sl_layernorm(const void* x, void* y, const void* gamma, const void* beta, int64_t rows, int32_t cols, float eps,
                            int32_t gelu, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(x && y && gamma && beta && rows >= 0 && cols > 0, "sl_layernorm: bad arguments");
  SL_DISPATCH_DTYPE(dtype, T, return (launch_norm<T, false>(x, y, gamma, beta, rows, cols, eps, gelu, (hipStream_t)stream)));
}

extern "C" int sl_rmsnorm(const void* x, void* y, const void* w, int64_t rows, int32_t cols, float eps, int32_t dtype,
                          sl_stream stream) {
  SL_CHECK_ARG(x && y && w && rows >= 0 && cols > 0, "sl_rmsnorm: bad arguments");
  SL_DISPATCH_DTYPE(dtype, T, return (launch_norm<T, true>(x, y, w, nullptr, rows, cols, eps, 0, (hipStream_t)stream)));
}

// RMSNorm scale of every row (rstd_out, fp32) and / or the normalised rows (y with gain w; y = w = NULL: the scale only).  The
// decode step above ~1 500 rows runs its o / down projections unsplit (no reduce pass to take the statistics in), so the scale
// the gain-folded qkv / gate-up weights need comes from this one-read pass instead (runtime.hip llama_layer).
int sl_rmsnorm_rstd_impl(const void* x, void* y, const void* w, float* rstd_out, int64_t rows, int32_t cols, float eps, int32_t dtype, hipStream_t st) {
  SL_CHECK_ARG(x && rows >= 0 && cols > 0 && (y == nullptr) == (w == nullptr) && (y || rstd_out), "sl_rmsnorm_rstd: bad arguments");
  if (rows == 0) return 0;
  SL_DISPATCH_DTYPE(dtype, T, {
    constexpr int VEC = Vec16<T>::VEC;
    SL_CHECK_ARG(cols % VEC == 0 && cols <= 64 * NORM_MAXF, "norm: cols=%d must be a multiple of %d and <= %d", cols, VEC, 64 * NORM_MAXF);
    hipLaunchKernelGGL((norm_rows_kernel<T, true>), dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, st, (const T*)x, (T*)y, (const T*)w, (const T*)nullptr,
                       rows, cols, eps, 0, rstd_out);
    SL_CHECK_LAUNCH("rmsnorm_rstd");
    return 0;
  });
}

// ----------------------------------------------------------------------------------------------
// LayerNorm folded into the Linears around it (sl_gemm_ex_args.ln_* / stats_out): the row statistics
// ----------------------------------------------------------------------------------------------
// {mean, rstd} of a row from the {sum, sum of squares} pairs its producer GEMM left per 64-column segment; one thread per row
__global__ __launch_bounds__(256) void ln_stats_finalize_kernel(const float* __restrict__ stats, int segs, int64_t rows, int cols, float eps,
                                                                float* __restrict__ mr) {
  const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (row >= rows) return;
  const float2* s = (const float2*)stats + row * segs;
  float s1 = 0.f;
  for (int i = 0; i < segs; ++i) s1 += s[i].x;
  const float mean = s1 / (float)cols;
  // segments merged the way Chan et al. merge partial variances: M2 = sum_s [ (sumsq_s - sum_s^2 / 64) + 64 (mean_s - mean)^2 ] — the
  // one-pass E[x^2] - mean^2 over the whole row cancels catastrophically once |mean| >> std; here only a segment's own 64 values do
  float m2 = 0.f;
  for (int i = 0; i < segs; ++i) {
    const float2 v = s[i];
    const float ms = v.x * (1.0f / 64.0f), d = ms - mean;
    m2 += (v.y - v.x * ms) + 64.0f * d * d;
  }
  float var = m2 / (float)cols;
  var = var > 0.f ? var : 0.f;
  ((float2*)mr)[row] = make_float2(mean, rsqrtf(var + eps));
}

// the same pair from the rows themselves: one wave per row, the row in registers, two-pass variance
template <typename T>
__global__ __launch_bounds__(256) void ln_rowstats_kernel(const T* __restrict__ x, int64_t rows, int cols, float eps, float* __restrict__ mr) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int MAXCH = NORM_MAXF / VEC;
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int nch = cols / VEC;
  float xv[MAXCH][VEC];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nch) {
      Vec16<T>::unpack(*(const uint4*)(x + row * cols + ch * VEC), xv[i]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) s += xv[i][e];
    }
  }
  const float mean = wave_sum(s) / (float)cols;
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i)
    if (lane + 64 * i < nch) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) { const float d = xv[i][e] - mean; s2 += d * d; }
    }
  const float var = wave_sum(s2) / (float)cols;
  if (lane == 0) ((float2*)mr)[row] = make_float2(mean, rsqrtf(var + eps));
}

extern "C" int sl_layernorm_stats_finalize(const float* stats, int32_t segs, int64_t rows, int32_t cols, float eps, float* mr, sl_stream stream) {
  SL_CHECK_ARG(stats && mr && segs > 0 && rows >= 0 && cols == 64 * segs, "sl_layernorm_stats_finalize: bad arguments (segs=%d cols=%d)", segs, cols);
  if (rows == 0) return 0;
  hipLaunchKernelGGL(ln_stats_finalize_kernel, dim3((unsigned)ceil_div64(rows, 256)), dim3(256), 0, (hipStream_t)stream, stats, segs, rows, cols, eps, mr);
  SL_CHECK_LAUNCH("ln_stats_finalize");
  return 0;
}

extern "C" int sl_layernorm_stats(const void* x, int64_t rows, int32_t cols, float eps, float* mr, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(x && mr && rows >= 0 && cols > 0, "sl_layernorm_stats: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(cols % vec == 0 && cols <= 64 * NORM_MAXF, "sl_layernorm_stats: cols=%d must be a multiple of %d and <= %d", cols, vec, 64 * NORM_MAXF);
  if (rows == 0) return 0;
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((ln_rowstats_kernel<T>), dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const T*)x, rows, cols, eps, mr);
  });
  SL_CHECK_LAUNCH("ln_rowstats");
  return 0;
}

// ----------------------------------------------------------------------------------------------
// LayerNorm fold, weight side (sl_hubert_fold's tensors): for a Linear W (N, K) behind LayerNorm(gain, beta)
//   Wf[n][k] = round_T(W[n][k] * gain[k]),   u[n] = sum_k float(Wf[n][k]),   c[n] = sum_k W[n][k] * beta[k] + bias[n]
// One wave per output row, fp32 sums in a fixed order (lane-strided, then the wave tree).  Replaces three torch launches per tensor
// (an element-wise product, a row reduction and a vendor GEMV) that ran after every optimizer step followed by an inference encode.
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ln_fold_build_kernel(const T* __restrict__ W, const T* __restrict__ gain, const T* __restrict__ beta,
                                                            const T* __restrict__ bias, T* __restrict__ Wf, float* __restrict__ u, float* __restrict__ c,
                                                            int N, int K) {
  constexpr int VEC = Vec16<T>::VEC;
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  float su = 0.f, sc = 0.f;
  for (int k = lane * VEC; k < K; k += 64 * VEC) {
    float w[VEC], g[VEC], b[VEC], f[VEC];
    Vec16<T>::unpack(*(const uint4*)(W + (int64_t)n * K + k), w);
    Vec16<T>::unpack(*(const uint4*)(gain + k), g);
    Vec16<T>::unpack(*(const uint4*)(beta + k), b);
#pragma unroll
    for (int e = 0; e < VEC; ++e) { f[e] = to_f32(from_f32<T>(w[e] * g[e])); su += f[e]; sc = fmaf(w[e], b[e], sc); }
    *(uint4*)(Wf + (int64_t)n * K + k) = Vec16<T>::pack(f);
  }
  su = wave_sum(su); sc = wave_sum(sc);
  if (lane == 0) { u[n] = su; c[n] = sc + (bias ? to_f32(bias[n]) : 0.f); }
}

extern "C" int sl_layernorm_fold_build(const void* W, const void* gain, const void* beta, const void* bias, void* Wf, float* u, float* c, int32_t N,
                                       int32_t K, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(W && gain && beta && Wf && u && c && N > 0 && K > 0, "sl_layernorm_fold_build: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(K % vec == 0, "sl_layernorm_fold_build: K=%d must be a multiple of %d", K, vec);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((ln_fold_build_kernel<T>), dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const T*)W, (const T*)gain, (const T*)beta,
                       (const T*)bias, (T*)Wf, u, c, N, K);
  });
  SL_CHECK_LAUNCH("ln_fold_build");
  return 0;
}
