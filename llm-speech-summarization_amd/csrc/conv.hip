// conv.hip — HuBERT front-end pieces that are not plain GEMMs (all HBM-bound byte movers):
//   * conv0: Conv1d(1 -> C, k=10, s=5) + LayerNorm(C) + GELU fused; channel-last output so layers
//     1..6 can run as implicit GEMMs on overlapping rows.  One wave owns a strip of time steps; its 64
//     lanes hold C/64 channels each (all taps in registers), the waveform strip sits in two VGPRs and
//     is broadcast with v_readlane, LayerNorm statistics are wave reductions, each time step leaves as
//     one contiguous, fully coalesced row write (1 KiB for C=512 bf16).
//   * posconv_stage: regroup (T,H) -> (groups, T+k, H/groups) with zero halo.
//   * avgpool_rows: AvgPool1d over time / ctc range mean.
#include "common.h"

template <typename T, int CPL, int K, int STRIDE>
__device__ __forceinline__ void conv0_strip(const float* __restrict__ wave, int64_t n_samples, const float* __restrict__ w,
                                            const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                            T* __restrict__ out, int64_t L, float eps, int64_t strip) {
  constexpr int C = 64 * CPL;
  constexpr int TS = (128 - (K - STRIDE)) / STRIDE;  // time steps per wave strip (24 for k=10, s=5)
  const int lane = threadIdx.x & 63;
  const int64_t t0 = strip * TS;
  if (t0 >= L) return;

  float wr[CPL][K], br[CPL], gr[CPL], ber[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int ch = lane * CPL + c;
#pragma unroll
    for (int j = 0; j < K; ++j) wr[c][j] = w[ch * K + j];
    br[c] = bias[ch]; gr[c] = gamma[ch]; ber[c] = beta[ch];
  }
  // waveform strip: samples [t0*S, t0*S + 128)
  const int64_t s0 = t0 * STRIDE;
  const int64_t i0 = s0 + lane, i1 = s0 + 64 + lane;
  const float x0 = i0 < n_samples ? wave[i0] : 0.f;
  const float x1 = i1 < n_samples ? wave[i1] : 0.f;
  const int b0 = __builtin_bit_cast(int, x0), b1 = __builtin_bit_cast(int, x1);

  const int nt = (int)((L - t0) < TS ? (L - t0) : TS);
  for (int tt = 0; tt < nt; ++tt) {
    float xs[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int idx = tt * STRIDE + j;  // wave-uniform
      const int lo = __builtin_amdgcn_readlane(b0, idx & 63), hi = __builtin_amdgcn_readlane(b1, idx & 63);
      xs[j] = __builtin_bit_cast(float, idx < 64 ? lo : hi);
    }
    float y[CPL];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      float a = br[c];
#pragma unroll
      for (int j = 0; j < K; ++j) a = fmaf(wr[c][j], xs[j], a);
      y[c] = a;
      s += a;
    }
    const float mean = wave_sum(s) / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) { const float d = y[c] - mean; s2 += d * d; }
    const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
    T* orow = out + (t0 + tt) * C + lane * CPL;
    float o[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) o[c] = gelu_act<T>((y[c] - mean) * rstd * gr[c] + ber[c]);
    constexpr int VEC = Vec16<T>::VEC;
    if constexpr (CPL % VEC == 0) {
#pragma unroll
      for (int c = 0; c < CPL; c += VEC) *(uint4*)(orow + c) = Vec16<T>::pack(&o[c]);
    } else {
#pragma unroll
      for (int c = 0; c < CPL; ++c) orow[c] = from_f32<T>(o[c]);
    }
  }
}

template <typename T, int CPL, int K, int STRIDE>
__global__ __launch_bounds__(256) void conv0_ln_gelu_kernel(const float* __restrict__ wave, int64_t n_samples,
                                                            const float* __restrict__ w, const float* __restrict__ bias,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            T* __restrict__ out, int64_t L, float eps) {
  conv0_strip<T, CPL, K, STRIDE>(wave, n_samples, w, bias, gamma, beta, out, L, eps, (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
}

// the whole ragged batch in one launch (utterance = blockIdx.y): one utterance alone is ~1300 waves, i.e. one or two
// per SIMD — nothing hides the LayerNorm reductions' latency and the second round of waves doubles the makespan
template <typename T, int CPL, int K, int STRIDE>
__global__ __launch_bounds__(256) void conv0_ln_gelu_batch_kernel(const float* __restrict__ waves, const int64_t* __restrict__ soff,
                                                                  const int64_t* __restrict__ row0, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, T* __restrict__ out, float eps) {
  const int u = blockIdx.y;
  const int64_t s0 = soff[u], r0 = row0[u];
  conv0_strip<T, CPL, K, STRIDE>(waves + s0, soff[u + 1] - s0, w, bias, gamma, beta, out + r0 * (64 * CPL), row0[u + 1] - r0, eps,
                                 (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6));
}

template <typename T, int CPL>
static int launch_conv0_batch(const float* waves, const int64_t* soff, const int64_t* row0, int n_utt, int64_t max_L, const float* w,
                              const float* b, const float* g, const float* be, void* out, float eps, hipStream_t st) {
  constexpr int TS = (128 - 5) / 5;
  const int64_t strips = ceil_div64(max_L, TS);
  hipLaunchKernelGGL((conv0_ln_gelu_batch_kernel<T, CPL, 10, 5>), dim3((unsigned)ceil_div64(strips, 4), n_utt), dim3(256), 0, st, waves, soff,
                     row0, w, b, g, be, (T*)out, eps);
  SL_CHECK_LAUNCH("conv0_ln_gelu_batch");
  return 0;
}

extern "C" int sl_hubert_conv0_batch(const float* waves, const int64_t* sample_offsets_dev, const int64_t* row_offsets_dev, int32_t n_utt, int64_t max_L,
                          const float* w, const float* bias, const float* gamma, const float* beta, void* out, int32_t C, int32_t k,
                          int32_t stride, float eps, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(waves && sample_offsets_dev && row_offsets_dev && w && bias && gamma && beta && out && n_utt > 0, "sl_hubert_conv0_batch: bad arguments");
  SL_CHECK_ARG(k == 10 && stride == 5, "sl_hubert_conv0_batch: only the HuBERT layer-0 geometry k=10, stride=5 is built (got k=%d s=%d)", k, stride);
  hipStream_t st = (hipStream_t)stream;
  SL_DISPATCH_DTYPE(dtype, T, {
    switch (C) {
      case 64: return launch_conv0_batch<T, 1>(waves, sample_offsets_dev, row_offsets_dev, n_utt, max_L, w, bias, gamma, beta, out, eps, st);
      case 128: return launch_conv0_batch<T, 2>(waves, sample_offsets_dev, row_offsets_dev, n_utt, max_L, w, bias, gamma, beta, out, eps, st);
      case 256: return launch_conv0_batch<T, 4>(waves, sample_offsets_dev, row_offsets_dev, n_utt, max_L, w, bias, gamma, beta, out, eps, st);
      case 512: return launch_conv0_batch<T, 8>(waves, sample_offsets_dev, row_offsets_dev, n_utt, max_L, w, bias, gamma, beta, out, eps, st);
      default: sl_set_error("sl_hubert_conv0_batch: C=%d must be 64, 128, 256 or 512", C); return SL_ERR_ARG;
    }
  });
}

template <typename T, int CPL>
static int launch_conv0(const float* wave, int64_t n, const float* w, const float* b, const float* g, const float* be, void* out,
                        int64_t L, float eps, hipStream_t st) {
  constexpr int TS = (128 - 5) / 5;
  const int64_t strips = ceil_div64(L, TS);
  hipLaunchKernelGGL((conv0_ln_gelu_kernel<T, CPL, 10, 5>), dim3((unsigned)ceil_div64(strips, 4)), dim3(256), 0, st, wave, n, w, b,
                     g, be, (T*)out, L, eps);
  SL_CHECK_LAUNCH("conv0_ln_gelu");
  return 0;
}

extern "C" int sl_hubert_conv0(const float* wave, int64_t n_samples, const float* w, const float* bias, const float* gamma,
                               const float* beta, void* out, int32_t C, int32_t k, int32_t stride, float eps, int32_t dtype,
                               sl_stream stream) {
  SL_CHECK_ARG(wave && w && bias && gamma && beta && out, "sl_hubert_conv0: null pointer");
  SL_CHECK_ARG(k == 10 && stride == 5, "sl_hubert_conv0: only the HuBERT layer-0 geometry k=10, stride=5 is built (got k=%d s=%d)", k, stride);
  SL_CHECK_ARG(n_samples >= k, "sl_hubert_conv0: n_samples=%lld shorter than the kernel", (long long)n_samples);
  const int64_t L = (n_samples - k) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  SL_DISPATCH_DTYPE(dtype, T, {
    switch (C) {
      case 64: return launch_conv0<T, 1>(wave, n_samples, w, bias, gamma, beta, out, L, eps, st);
      case 128: return launch_conv0<T, 2>(wave, n_samples, w, bias, gamma, beta, out, L, eps, st);
      case 256: return launch_conv0<T, 4>(wave, n_samples, w, bias, gamma, beta, out, L, eps, st);
      case 512: return launch_conv0<T, 8>(wave, n_samples, w, bias, gamma, beta, out, L, eps, st);
      default: sl_set_error("sl_hubert_conv0: C=%d must be 64, 128, 256 or 512", C); return SL_ERR_ARG;
    }
  });
}

// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void posconv_stage_kernel(const T* __restrict__ x, T* __restrict__ xg, int64_t T_, int H,
                                                            int groups, int k) {
  constexpr int VEC = Vec16<T>::VEC;
  const int Hg = H / groups, cpr = Hg / VEC;  // chunks per staged row
  const int64_t rows = T_ + k;
  const int64_t total = (int64_t)groups * rows * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % cpr);
    const int64_t rr = (i / cpr) % rows;
    const int g = (int)(i / (cpr * rows));
    const int64_t t = rr - k / 2;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t >= 0 && t < T_) v = *(const uint4*)(x + t * H + g * Hg + ch * VEC);
    *(uint4*)(xg + ((int64_t)g * rows + rr) * Hg + ch * VEC) = v;
  }
}

// whole ragged batch in one launch: utterance u = blockIdx.y, frames cu[u] .. cu[u] + klen[u]; its staged rows follow
// those of the utterances before it, each with k rows of padding ((cu[u] + u k) rows in)
template <typename T>
__global__ __launch_bounds__(256) void posconv_stage_batch_kernel(const T* __restrict__ x, T* __restrict__ xg, const int32_t* __restrict__ cu,
                                                                  const int32_t* __restrict__ klen, int H, int groups, int k) {
  constexpr int VEC = Vec16<T>::VEC;
  const int u = blockIdx.y;
  const int64_t T_ = klen[u], tok0 = cu[u];
  x += tok0 * H;
  xg += (tok0 + (int64_t)u * k) * H;
  const int Hg = H / groups, cpr = Hg / VEC;
  const int64_t rows = T_ + k;
  const int64_t total = (int64_t)groups * rows * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % cpr);
    const int64_t rr = (i / cpr) % rows;
    const int g = (int)(i / (cpr * rows));
    const int64_t t = rr - k / 2;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (t >= 0 && t < T_) v = *(const uint4*)(x + t * H + g * Hg + ch * VEC);
    *(uint4*)(xg + ((int64_t)g * rows + rr) * Hg + ch * VEC) = v;
  }
}

extern "C" int sl_posconv_stage_batch(const void* x, void* xg, const int32_t* cu, const int32_t* klen, int32_t n_utt, int64_t max_T, int32_t H,
                           int32_t groups, int32_t k, int32_t dtype, sl_stream stream) {
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(x && xg && cu && klen && n_utt > 0 && groups > 0 && H % groups == 0 && (H / groups) % vec == 0, "sl_posconv_stage_batch: bad arguments");
  const int64_t total = (int64_t)groups * (max_T + k) * (H / groups / vec);
  const unsigned gx = (unsigned)(ceil_div64(total, 256) < 64 ? ceil_div64(total, 256) : 64);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((posconv_stage_batch_kernel<T>), dim3(gx, n_utt), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)xg, cu, klen, H, groups, k);
  });
  SL_CHECK_LAUNCH("posconv_stage_batch");
  return 0;
}

extern "C" int sl_posconv_stage(const void* x, void* xg, int64_t T_, int32_t H, int32_t groups, int32_t k, int32_t dtype,
                                sl_stream stream) {
  SL_CHECK_ARG(x && xg && T_ > 0 && groups > 0 && H % groups == 0, "sl_posconv_stage: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG((H / groups) % vec == 0, "sl_posconv_stage: H/groups=%d must be a multiple of %d", H / groups, vec);
  const int64_t total = (int64_t)groups * (T_ + k) * (H / groups / vec);
  const unsigned grid = (unsigned)(ceil_div64(total, 256) < 4096 ? ceil_div64(total, 256) : 4096);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((posconv_stage_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)xg, T_, H, groups, k);
  });
  SL_CHECK_LAUNCH("posconv_stage");
  return 0;
}

// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void avgpool_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t T_, int H, int kernel,
                                                           int stride, const int32_t* __restrict__ ranges, int64_t P) {
  constexpr int VEC = Vec16<T>::VEC;
  const int cpr = H / VEC;
  const int64_t total = P * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % cpr);
    const int64_t p = i / cpr;
    int64_t s = ranges ? ranges[2 * p] : p * stride;
    int64_t e = ranges ? ranges[2 * p + 1] : s + kernel;
    if (e > T_) e = T_;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int64_t t = s; t < e; ++t) {
      float f[VEC];
      Vec16<T>::unpack(*(const uint4*)(x + t * H + ch * VEC), f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += f[j];
    }
    const float inv = 1.0f / (float)(ranges ? (e - s) : kernel);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] *= inv;
    *(uint4*)(y + p * H + ch * VEC) = Vec16<T>::pack(acc);
  }
}

// AvgPool1d over time for the whole ragged batch: utterance u = blockIdx.y pools its klen[u] frames (rows cu[u]..) into
// rec[4u] rows that start rec[4u+1] ELEMENTS into y (the records of the grouped projector GEMM that follows)
template <typename T>
__global__ __launch_bounds__(256) void avgpool_batch_kernel(const T* __restrict__ x, T* __restrict__ y, const int32_t* __restrict__ cu,
                                                            const int32_t* __restrict__ klen, const int64_t* __restrict__ rec, int H, int kernel,
                                                            int stride) {
  constexpr int VEC = Vec16<T>::VEC;
  const int u = blockIdx.y;
  const int64_t T_ = klen[u], P = rec[4 * u];
  x += (int64_t)cu[u] * H;
  y += rec[4 * u + 1];
  const int cpr = H / VEC;
  const int64_t total = P * cpr;
  const float inv = 1.0f / (float)kernel;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % cpr);
    const int64_t pi = i / cpr;
    const int64_t s0 = pi * stride;
    int64_t e = s0 + kernel;
    if (e > T_) e = T_;
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    for (int64_t t = s0; t < e; ++t) {
      float f[VEC];
      Vec16<T>::unpack(*(const uint4*)(x + t * H + ch * VEC), f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += f[j];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] *= inv;
    *(uint4*)(y + pi * H + ch * VEC) = Vec16<T>::pack(acc);
  }
}

extern "C" int sl_avgpool_batch(const void* x, void* y, const int32_t* cu, const int32_t* klen, const int64_t* rec, int32_t n_utt, int64_t max_P, int32_t H,
                     int32_t kernel, int32_t stride, int32_t dtype, sl_stream stream) {
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(x && y && cu && klen && rec && n_utt > 0 && H % vec == 0 && kernel > 0 && stride > 0, "sl_avgpool_batch: bad arguments");
  if (max_P <= 0) return 0;
  const int64_t total = max_P * (H / vec);
  const unsigned gx = (unsigned)(ceil_div64(total, 256) < 64 ? ceil_div64(total, 256) : 64);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((avgpool_batch_kernel<T>), dim3(gx, n_utt), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, cu, klen, rec, H, kernel, stride);
  });
  SL_CHECK_LAUNCH("avgpool_batch");
  return 0;
}

extern "C" int sl_avgpool_rows(const void* x, void* y, int64_t T_, int32_t H, int32_t kernel, int32_t stride, const int32_t* ranges,
                               int64_t P, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(x && y && T_ > 0 && H > 0 && P >= 0, "sl_avgpool_rows: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(H % vec == 0, "sl_avgpool_rows: H=%d must be a multiple of %d", H, vec);
  if (!ranges) SL_CHECK_ARG(kernel > 0 && stride > 0 && (P == 0 || (P - 1) * stride + kernel <= T_), "sl_avgpool_rows: window past the end");
  if (P == 0) return 0;
  const int64_t total = P * (H / vec);
  const unsigned grid = (unsigned)(ceil_div64(total, 256) < 4096 ? ceil_div64(total, 256) : 4096);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((avgpool_rows_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, T_, H, kernel, stride,
                       ranges, P);
  });
  SL_CHECK_LAUNCH("avgpool_rows");
  return 0;
}
