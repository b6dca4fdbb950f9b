// tools/probe_persistent_chain.hip — timing probe for a persistent batch-1 decode chain (no numerics, no product code).
// Question: does ONE launch that walks the 4 weight-streaming products of every layer (qkv, o, gate/up, down: 201 MB per layer,
// 28 layers, bf16, fragment-packed) behind grid-wide hand-offs beat the same products as 112 launches inside a HIP graph?
//   * "launches": the decode step's own kernel structures (tools/tune_skinny.hip shapes: qkv 2x8x4, o 1x16x3, gate/up 4x4x4,
//     down 1x16x4), one launch per product, captured in a graph and replayed.
//   * "persistent": 256 workgroups x 16 waves; each product's 1 KiB wave-loads are cut into 4 096 equal contiguous runs (one per
//     wave); a wave issues the first U loads of the NEXT product before it waits at the grid barrier (sc1 arrival counters
//     sharded by XCD, bounded spin), so the weight stream keeps running through the hand-off; partial sums leave as sc1 records.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_persistent_chain.hip -o tools/probe_persistent_chain
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ __forceinline__ uint4 ld_nt16(const void* p) { u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p); return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void mma(f32x4& acc, const uint4& a, const uint4& b) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc, 0, 0, 0);
}

// ---- the launch-per-product form (same structure as gemm_skinny_kernel, M = 1) ----
template <int RF, int NW, int U>
__global__ __launch_bounds__(NW * 64) void gemv_kernel(const uint16_t* __restrict__ W, const uint16_t* __restrict__ X, uint16_t* __restrict__ C, int N, int K) {
  __shared__ float red[NW][RF * 16][17];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * RF * 16, nks = K / 32;
  const uint16_t* wp[RF];
#pragma unroll
  for (int f = 0; f < RF; ++f) wp[f] = W + ((size_t)(n0 / 16 + f) * nks) * 512 + lane * 8;
  const uint16_t* xp = X + q * 8;
  f32x4 acc[RF];
#pragma unroll
  for (int f = 0; f < RF; ++f) acc[f] = f32x4{0, 0, 0, 0};
  int ks = wave;
  for (; ks + (U - 1) * NW < nks; ks += U * NW) {
    uint4 fw[U][RF], fx[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t kk = (size_t)(ks + u * NW);
#pragma unroll
      for (int f = 0; f < RF; ++f) fw[u][f] = ld_nt16(wp[f] + kk * 512);
      fx[u] = *(const uint4*)(xp + kk * 32);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int f = 0; f < RF; ++f) mma(acc[f], fw[u][f], fx[u]);
  }
  for (; ks < nks; ks += NW) {
#pragma unroll
    for (int f = 0; f < RF; ++f) mma(acc[f], ld_nt16(wp[f] + (size_t)ks * 512), *(const uint4*)(xp + (size_t)ks * 32));
  }
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) red[wave][f * 16 + q * 4 + i][r] = acc[f][i];
  __syncthreads();
  for (int o = tid; o < RF * 16; o += NW * 64) {
    float v = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][o][0];
    __bf16 h = (__bf16)v;
    C[n0 + o] = __builtin_bit_cast(uint16_t, h);
  }
}

// ---- the persistent form ----
struct Op { const uint16_t* W; int nfrag, ksteps; };
struct Chain { Op op[4]; };   // one layer; layer l's weights sit at W + l * layer_stride (elements)

__device__ __forceinline__ void st_sc1_16(float* ptr, const f32x4& v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int U, bool PREFETCH, bool BARRIER>
__global__ __launch_bounds__(1024) void chain_kernel(Chain ch, size_t layer_stride, int n_layers, const uint16_t* __restrict__ X, float* __restrict__ part,
                                                     unsigned* __restrict__ bar, unsigned* __restrict__ err) {
  extern __shared__ unsigned char pad_lds[];   // sized by the host to keep one workgroup per CU
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, q = lane >> 4;
  const int gw = blockIdx.x * 16 + wave, NWAVES = gridDim.x * 16;
  const int shard = blockIdx.x & 7, per_shard = gridDim.x / 8;
  if (tid == 0) pad_lds[0] = 0;
  uint4 w[U];
  unsigned gen = 0;
  bool dead = false;
  // run of product `o` (0..4*n_layers-1) for this wave: [lo, hi) in 1 KiB wave-loads, fragment-major
  auto run = [&](int o, const uint16_t*& base, int& lo, int& hi, int& ksteps) {
    const Op& p = ch.op[o & 3];
    const long T = (long)p.nfrag * p.ksteps;
    lo = (int)((long)gw * T / NWAVES); hi = (int)((long)(gw + 1) * T / NWAVES);
    base = p.W + (size_t)(o >> 2) * layer_stride; ksteps = p.ksteps;
  };
  const int n_ops = 4 * n_layers;
  const uint16_t* base; int lo, hi, ksteps;
  run(0, base, lo, hi, ksteps);
  bool have = false;
  for (int o = 0; o < n_ops; ++o) {
    const int n = hi - lo;
    if (!have) {
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = u < n ? u : n - 1; w[u] = ld_nt16(base + (size_t)(lo + i) * 512 + lane * 8); }
    }
    f32x4 acc = {0, 0, 0, 0};
    int frag = lo / ksteps, seg = 0;
    float* rec = part + ((size_t)(o & 1) * NWAVES + gw) * 2 * 16;
    for (int c = 0; c < n; c += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = c + u;
        if (i < n) {
          const int idx = lo + i, f = idx / ksteps, ks = idx - f * ksteps;
          if (f != frag) {
            if (r == 0 && seg < 2) st_sc1_16(rec + seg * 16 + q * 4, acc);
            ++seg; frag = f; acc = f32x4{0, 0, 0, 0};
          }
          const uint4 fx = *(const uint4*)(X + (size_t)ks * 32 + q * 8);
          mma(acc, w[u], fx);
          const int nx = i + U;
          if (nx < n) w[u] = ld_nt16(base + (size_t)(lo + nx) * 512 + lane * 8);
        }
      }
    }
    if (r == 0 && seg < 2) st_sc1_16(rec + seg * 16 + q * 4, acc);
    if (o + 1 == n_ops) break;
    // hand-off: records drained, next product's first loads in flight, then the grid barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    run(o + 1, base, lo, hi, ksteps);
    have = false;
    if (PREFETCH && wave != 0) {
      const int n2 = hi - lo;
#pragma unroll
      for (int u = 0; u < U; ++u) { const int i = u < n2 ? u : n2 - 1; w[u] = ld_nt16(base + (size_t)(lo + i) * 512 + lane * 8); }
      have = true;
    }
    if (BARRIER) {
      ++gen;
      __builtin_amdgcn_s_barrier();
      if (wave == 0) {
        if (lane == 0) __hip_atomic_fetch_add(&bar[shard * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < 8 && !dead) {
          const unsigned target = gen * per_shard;
          int spins = 0;
          while ((int)(ld_sc1(&bar[lane * 32]) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 200000) { dead = true; atomicAdd(err, 1u); break; }
          }
        }
      }
      __builtin_amdgcn_s_barrier();
    }
  }
}

int main(int argc, char** argv) {
  const int L = argc > 1 ? atoi(argv[1]) : 28;
  const int reps = 10;
  struct S { int N, K; } sh[4] = {{5120, 3072}, {3072, 3072}, {16384, 3072}, {3072, 8192}};
  size_t off[4], layer_elems = 0;
  for (int i = 0; i < 4; ++i) { off[i] = layer_elems; layer_elems += (size_t)sh[i].N * sh[i].K; }
  uint16_t* W; CK(hipMalloc(&W, layer_elems * 2 * L)); CK(hipMemset(W, 0x3c, layer_elems * 2 * L));
  uint16_t *X, *C; CK(hipMalloc(&X, 16384 * 2)); CK(hipMemset(X, 0x3c, 16384 * 2)); CK(hipMalloc(&C, 16384 * 2));
  float* part; CK(hipMalloc(&part, (size_t)2 * 4096 * 2 * 16 * 4)); unsigned *bar, *err; CK(hipMalloc(&bar, 8 * 32 * 4)); CK(hipMalloc(&err, 4));
  const double bytes = (double)layer_elems * 2 * L;
  printf("%d layers, %.1f MB per layer, %.2f GB per pass\n", L, layer_elems * 2 / 1e6, bytes / 1e9);
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  {  // launches in a graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < L; ++l) {
      const uint16_t* wl = W + (size_t)l * layer_elems;
      hipLaunchKernelGGL((gemv_kernel<2, 8, 4>), dim3(5120 / 32), dim3(512), 0, st, wl + off[0], X, C, 5120, 3072);
      hipLaunchKernelGGL((gemv_kernel<1, 16, 3>), dim3(3072 / 16), dim3(1024), 0, st, wl + off[1], X, C, 3072, 3072);
      hipLaunchKernelGGL((gemv_kernel<4, 4, 4>), dim3(16384 / 64), dim3(256), 0, st, wl + off[2], X, C, 16384, 3072);
      hipLaunchKernelGGL((gemv_kernel<1, 16, 4>), dim3(3072 / 16), dim3(1024), 0, st, wl + off[3], X, C, 3072, 8192);
    }
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < reps; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-52s %8.1f us per pass  %6.2f TB/s  %6.2f us per layer\n", "graph of 4 launches per layer", ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12, ms * 1e3 / reps / L);
  }
  Chain ch;
  for (int i = 0; i < 4; ++i) ch.op[i] = Op{W + off[i], sh[i].N / 16, sh[i].K / 32};
  auto time_chain = [&](const char* name, auto kern) {
    const int lds = 96 * 1024;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    float best = 1e9f;
    for (int i = 0; i < reps + 2; ++i) {
      CK(hipMemsetAsync(bar, 0, 8 * 32 * 4, st)); CK(hipMemsetAsync(err, 0, 4, st));
      CK(hipEventRecord(e0, st));
      hipLaunchKernelGGL(kern, dim3(256), dim3(1024), lds, st, ch, layer_elems, L, X, part, bar, err);
      CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      if (i >= 2 && ms < best) best = ms;
    }
    unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("  %-52s %8.1f us per pass  %6.2f TB/s  %6.2f us per layer%s\n", name, best * 1e3, bytes / (best * 1e-3) / 1e12, best * 1e3 / L, herr ? "  (SPIN BOUND HIT)" : "");
  };
  time_chain("one launch, no barriers (stream ceiling), U=8", chain_kernel<8, true, false>);
  time_chain("one launch, grid barriers, no prefetch, U=8", chain_kernel<8, false, true>);
  time_chain("one launch, grid barriers + prefetch, U=8", chain_kernel<8, true, true>);
  time_chain("one launch, grid barriers + prefetch, U=4", chain_kernel<4, true, true>);
  time_chain("one launch, grid barriers + prefetch, U=12", chain_kernel<12, true, true>);
  return 0;
}
