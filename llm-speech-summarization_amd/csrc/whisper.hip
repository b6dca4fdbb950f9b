// whisper.hip — Whisper log-mel front end (hf:models/whisper/feature_extraction_whisper.py:135-168) on the GPU.
// The STFT is a GEMM: frames are overlapping rows of the reflect-padded waveform (lda = hop < n_fft, the same
// implicit-im2col trick as the HuBERT convolutions) against a windowed DFT basis [cos | sin] (2*(n_fft/2+1), n_fft),
// in exact fp32 MFMA; then power, the slaney mel projection (second fp32 GEMM), log10 / dynamic-range clamp.
#include "common.h"

int sl_gemm_impl(const sl_gemm_args* a, const sl_gemm_fused* fx, const sl_gemm_ex_args* ex, hipStream_t st);

// zero-pad / trim to n_total samples, then reflect-pad by `pad` on both sides (torch.stft center=True)
__global__ __launch_bounds__(256) void whisper_pad_kernel(const float* __restrict__ audio, int64_t n_samples, float* __restrict__ out, int64_t n_total,
                                                          int pad) {
  const int64_t total = n_total + 2 * pad;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int64_t j = i - pad;
    if (j < 0) j = -j;
    if (j >= n_total) j = 2 * (n_total - 1) - j;
    out[i] = j < n_samples ? audio[j] : 0.f;
  }
}

// spec (frames, ld_spec) = [re(0..nb) | im(0..nb)]  ->  power (frames, ld_pw) with zero padding columns
__global__ __launch_bounds__(256) void whisper_power_kernel(const float* __restrict__ spec, int64_t ld_spec, float* __restrict__ pw, int64_t ld_pw,
                                                            int64_t frames, int nb) {
  const int64_t total = frames * ld_pw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / ld_pw;
    const int k = (int)(i % ld_pw);
    float v = 0.f;
    if (k < nb) { const float re = spec[t * ld_spec + k], im = spec[t * ld_spec + nb + k]; v = re * re + im * im; }
    pw[i] = v;
  }
}

// in place: x = log10(max(x, 1e-10)); *gmax = max over everything (single block)
__global__ __launch_bounds__(1024) void whisper_log_max_kernel(float* __restrict__ x, int64_t n, float* __restrict__ gmax) {
  __shared__ float sh[16];
  float m = -INFINITY;
  for (int64_t i = threadIdx.x; i < n; i += 1024) {
    const float v = log10f(fmaxf(x[i], 1e-10f));
    x[i] = v;
    m = fmaxf(m, v);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 16; ++w) m = fmaxf(m, sh[w]);
    *gmax = m;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void whisper_finish_kernel(const float* __restrict__ x, const float* __restrict__ gmax, T* __restrict__ out, int64_t n) {
  const float lo = *gmax - 8.0f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = from_f32<T>((fmaxf(x[i], lo) + 4.0f) / 4.0f);
}

static inline size_t rup256(size_t v) { return (v + 255) & ~(size_t)255; }

extern "C" size_t sl_whisper_logmel_workspace_bytes(int32_t n_fft, int32_t hop, int32_t n_frames, int32_t n_mel) {
  const int nb = n_fft / 2 + 1, ld_spec = (2 * nb + 3) & ~3, ld_pw = (nb + 3) & ~3;
  const size_t n_total = (size_t)n_frames * hop;
  return rup256((n_total + n_fft) * 4) + rup256((size_t)n_frames * ld_spec * 4) + rup256((size_t)n_frames * ld_pw * 4) +
         rup256((size_t)n_frames * n_mel * 4) + 256;
}

extern "C" int sl_whisper_logmel(const float* audio, int64_t n_samples, const float* dft_basis, const float* mel_w, void* mel_out, int32_t n_fft,
                                 int32_t hop, int32_t n_frames, int32_t n_mel, void* workspace, size_t workspace_bytes, int32_t dtype,
                                 sl_stream stream) {
  SL_CHECK_ARG(audio && dft_basis && mel_w && mel_out && workspace && n_samples > 0, "sl_whisper_logmel: bad arguments");
  SL_CHECK_ARG(n_fft % 4 == 0 && hop % 4 == 0 && n_mel > 0 && n_frames > 0, "sl_whisper_logmel: n_fft and hop must be multiples of 4");
  SL_CHECK_ARG(workspace_bytes >= sl_whisper_logmel_workspace_bytes(n_fft, hop, n_frames, n_mel), "sl_whisper_logmel: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int nb = n_fft / 2 + 1, ld_spec = (2 * nb + 3) & ~3, ld_pw = (nb + 3) & ~3;
  const int64_t n_total = (int64_t)n_frames * hop;
  unsigned char* p = (unsigned char*)workspace;
  float* padded = (float*)p; p += rup256((n_total + n_fft) * 4);
  float* spec = (float*)p;   p += rup256((size_t)n_frames * ld_spec * 4);
  float* pw = (float*)p;     p += rup256((size_t)n_frames * ld_pw * 4);
  float* mel = (float*)p;    p += rup256((size_t)n_frames * n_mel * 4);
  float* gmax = (float*)p;
  hipLaunchKernelGGL(whisper_pad_kernel, dim3(1024), dim3(256), 0, st, audio, n_samples, padded, n_total, n_fft / 2);
  SL_CHECK_LAUNCH("whisper_pad");
  sl_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = padded; a.lda = hop; a.W = dft_basis; a.ldw = n_fft; a.C = spec; a.ldc = ld_spec;
  a.M = n_frames; a.N = 2 * nb; a.K = n_fft; a.batch = 1; a.dtype = SL_F32;   // frame t = padded[t*hop : t*hop + n_fft]; the last STFT frame is dropped
  SL_TRY(sl_gemm_impl(&a, nullptr, nullptr, st));
  hipLaunchKernelGGL(whisper_power_kernel, dim3(2048), dim3(256), 0, st, spec, (int64_t)ld_spec, pw, (int64_t)ld_pw, (int64_t)n_frames, nb);
  SL_CHECK_LAUNCH("whisper_power");
  memset(&a, 0, sizeof(a));
  a.A = pw; a.lda = ld_pw; a.W = mel_w; a.ldw = ld_pw; a.C = mel; a.ldc = n_mel;
  a.M = n_frames; a.N = n_mel; a.K = ld_pw; a.batch = 1; a.dtype = SL_F32;
  SL_TRY(sl_gemm_impl(&a, nullptr, nullptr, st));
  hipLaunchKernelGGL(whisper_log_max_kernel, dim3(1), dim3(1024), 0, st, mel, (int64_t)n_frames * n_mel, gmax);
  SL_CHECK_LAUNCH("whisper_log_max");
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((whisper_finish_kernel<T>), dim3(512), dim3(256), 0, st, mel, gmax, (T*)mel_out, (int64_t)n_frames * n_mel);
  });
  SL_CHECK_LAUNCH("whisper_finish");
  return 0;
}
