// tools/tune_skinny.hip — standalone structure sweep for the decode weight-streaming GEMM (bf16, M<=16).
// Build: hipcc -O3 --offload-arch=gfx950 tools/tune_skinny.hip -o /tmp/tune_skinny ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
__device__ __forceinline__ uint4 ld_nt16(const void* p) { u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint4 ld16(const void* p) { return *(const uint4*)p; }
__device__ __forceinline__ void mma(f32x4& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}
// KMODE 0: wave w takes k-steps w, w+NW, ... ; 1: contiguous slice.  NT: nontemporal weight loads.
// PACKED: W stored fragment-major [N/16][K/32][64 lanes][8] so one wave load = 1 KiB contiguous.
template <int RF, int NW, int U, int KMODE, int NT, int PACKED, int MT = 1>
__global__ __launch_bounds__(NW * 64) void k(const uint16_t* __restrict__ W, const uint16_t* __restrict__ X, uint16_t* __restrict__ C, int M, int N, int K) {
  __shared__ float red[NW][RF * 16][MT * 16 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * RF * 16;
  const int nks = K / 32;
  const uint16_t* wp[RF];
#pragma unroll
  for (int f = 0; f < RF; ++f) {
    if (PACKED) wp[f] = W + ((size_t)(n0 / 16 + f) * nks) * 512 + lane * 8;
    else wp[f] = W + (size_t)(n0 + f * 16 + r) * K + q * 8;
  }
  const uint16_t* xp[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) { int xr = t * 16 + r; xr = xr < M ? xr : M - 1; xp[t] = X + (size_t)xr * K + q * 8; }
  f32x4 acc[RF][MT];
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[f][t] = f32x4{0, 0, 0, 0};
  int ks, kend, kstride;
  if (KMODE == 0) { ks = wave; kend = nks; kstride = NW; }
  else { const int per = (nks + NW - 1) / NW; ks = wave * per; kend = ks + per < nks ? ks + per : nks; kstride = 1; }
  for (; ks + (U - 1) * kstride < kend; ks += U * kstride) {
    uint4 fw[U][RF], fx[U][MT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t kk = (size_t)(ks + u * kstride);
#pragma unroll
      for (int f = 0; f < RF; ++f) { const uint16_t* a = PACKED ? wp[f] + kk * 512 : wp[f] + kk * 32; fw[u][f] = NT ? ld_nt16(a) : ld16(a); }
#pragma unroll
      for (int t = 0; t < MT; ++t) fx[u][t] = ld16(xp[t] + kk * 32);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int t = 0; t < MT; ++t) mma(acc[f][t], fw[u][f], fx[u][t]);
  }
  for (; ks < kend; ks += kstride) {
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      const uint16_t* a = PACKED ? wp[f] + (size_t)ks * 512 : wp[f] + (size_t)ks * 32;
      const uint4 wv = NT ? ld_nt16(a) : ld16(a);
#pragma unroll
      for (int t = 0; t < MT; ++t) mma(acc[f][t], wv, ld16(xp[t] + (size_t)ks * 32));
    }
  }
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave][f * 16 + q * 4 + i][t * 16 + r] = acc[f][t][i];
  __syncthreads();
  for (int o = tid; o < RF * 16 * M; o += NW * 64) {
    const int m = o / (RF * 16), n = o % (RF * 16);
    float v = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][n][m];
    __bf16 h = (__bf16)v;
    C[(size_t)m * N + n0 + n] = __builtin_bit_cast(uint16_t, h);
  }
}
// plain streaming read (upper bound for this access shape): every lane loads 16 B, grid-stride, sum.
__global__ __launch_bounds__(256) void stream_read(const uint4* __restrict__ p, size_t n16, uint4* out) {
  uint4 acc = make_uint4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { uint4 v = ld_nt16(p + i); acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
  if (acc.x == 0x12345678u) out[0] = acc;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <int RF, int NW, int U, int KMODE, int NT, int PACKED, int MT = 1>
static void run(const char* name, const std::vector<uint16_t*>& Ws, uint16_t* X, uint16_t* C, int M, int N, int K) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 5, L = (int)Ws.size();
  dim3 grid(N / (RF * 16));
  for (int i = 0; i < L; ++i) hipLaunchKernelGGL((k<RF, NW, U, KMODE, NT, PACKED, MT>), grid, dim3(NW * 64), 0, 0, Ws[i], X, C, M, N, K);
  CK(hipEventRecord(e0, 0));
  for (int rp = 0; rp < reps; ++rp)
    for (int i = 0; i < L; ++i) hipLaunchKernelGGL((k<RF, NW, U, KMODE, NT, PACKED, MT>), grid, dim3(NW * 64), 0, 0, Ws[i], X, C, M, N, K);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / (reps * L), gbs = (double)N * K * 2 / (us * 1e-6) / 1e9;
  printf("  %-34s M=%2d N=%6d K=%5d blocks=%5d  %7.2f us  %7.1f GB/s\n", name, M, N, K, grid.x, us, gbs);
}
int main() {
  struct Shape { int N, K; } shapes[] = {{3072, 3072}, {5120, 3072}, {3072, 8192}, {16384, 3072}};
  const int L = 24;  // rotate through L weight matrices (> 256 MB Infinity Cache in total for the big ones)
  for (auto s : shapes) {
    std::vector<uint16_t*> Ws(L);
    const size_t bytes = (size_t)s.N * s.K * 2;
    for (auto& w : Ws) { CK(hipMalloc(&w, bytes)); CK(hipMemset(w, 0x3c, bytes)); }
    uint16_t *X, *C; CK(hipMalloc(&X, 64 * s.K * 2)); CK(hipMemset(X, 0x3c, 64 * s.K * 2)); CK(hipMalloc(&C, 64 * s.N * 2));
    printf("shape N=%d K=%d (%.1f MB)\n", s.N, s.K, bytes / 1e6);
    {  // streaming upper bound
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); uint4* o; CK(hipMalloc(&o, 16));
      for (int i = 0; i < L; ++i) hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, (const uint4*)Ws[i], bytes / 16, o);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < 5 * L; ++i) hipLaunchKernelGGL(stream_read, dim3(2048), dim3(256), 0, 0, (const uint4*)Ws[i % L], bytes / 16, o);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  %-34s %51.2f us  %7.1f GB/s\n", "plain nt stream read (2048 blocks)", ms * 1e3 / (5 * L), bytes / (ms * 1e-3 / (5 * L)) / 1e9);
    }
    for (int M : {1, 16}) {
      run<1, 16, 2, 0, 1, 1, 1>("rf1 nw16 u2 (current small-N)", Ws, X, C, M, s.N, s.K);
      run<1, 16, 3, 0, 1, 1, 1>("rf1 nw16 u3", Ws, X, C, M, s.N, s.K);
      run<1, 16, 6, 0, 1, 1, 1>("rf1 nw16 u6", Ws, X, C, M, s.N, s.K);
      run<1, 16, 8, 0, 1, 1, 1>("rf1 nw16 u8", Ws, X, C, M, s.N, s.K);
      run<1, 16, 6, 1, 1, 1, 1>("rf1 nw16 u6 contiguous", Ws, X, C, M, s.N, s.K);
      run<1, 8, 4, 0, 1, 1, 1>("rf1 nw8 u4", Ws, X, C, M, s.N, s.K);
      run<1, 8, 6, 0, 1, 1, 1>("rf1 nw8 u6", Ws, X, C, M, s.N, s.K);
      run<1, 8, 12, 0, 1, 1, 1>("rf1 nw8 u12", Ws, X, C, M, s.N, s.K);
      run<2, 16, 2, 0, 1, 1, 1>("rf2 nw16 u2", Ws, X, C, M, s.N, s.K);
      run<2, 16, 4, 0, 1, 1, 1>("rf2 nw16 u4", Ws, X, C, M, s.N, s.K);
      run<2, 8, 4, 0, 1, 1, 1>("rf2 nw8 u4", Ws, X, C, M, s.N, s.K);
      run<2, 8, 8, 0, 1, 1, 1>("rf2 nw8 u8", Ws, X, C, M, s.N, s.K);
      run<4, 4, 4, 0, 1, 1, 1>("rf4 nw4 u4 (current big-N)", Ws, X, C, M, s.N, s.K);
      run<4, 4, 8, 0, 1, 1, 1>("rf4 nw4 u8", Ws, X, C, M, s.N, s.K);
      run<4, 8, 4, 0, 1, 1, 1>("rf4 nw8 u4", Ws, X, C, M, s.N, s.K);
    }
    for (auto w : Ws) CK(hipFree(w));
    CK(hipFree(X)); CK(hipFree(C));
  }
  return 0;
}
