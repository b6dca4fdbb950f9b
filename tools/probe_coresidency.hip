// tools/probe_coresidency.hip — do workgroups of TWO kernels launched on two streams share a CU on MI355X when their resources fit together?
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 tools/probe_coresidency.hip -o gpurun_out/probe_coresidency && gpurun_out/probe_coresidency
//
// "hog"   = the shape of a 256 x 256 GEMM block: 512 threads, HOG_LDS bytes of LDS (130 KiB: one block per CU), <= 128 registers; every block
//           spins for `hog_us` of wall clock.  Grid = one block per CU, or several rounds of them.
// "small" = the shape of a decode-attention block: 256 threads, SMALL_LDS bytes of LDS (19.5 KiB with 64-key chunks, 38.7 KiB with 128-key
//           chunks); every block spins for `small_us`.  Grid = `small_blocks`.
// Each kernel records, per block, the CU it ran on (XCC_ID / SE / CU from HW_ID) and its start / end wall clock.  Three timings:
//   hog alone, small alone, both at once on two streams.  If small blocks start while hog blocks of the SAME CU are still running, the two
//   kernels share CUs; the report counts such blocks and prints the three wall times.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Rec { unsigned long long t0, t1; unsigned int hw, pad; };

__device__ __forceinline__ unsigned int hw_id() {
  unsigned int v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  unsigned int x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return (v & 0xffffu) | ((x & 0xfu) << 16);      // HW_ID[15:0]: wave, simd, pipe, cu, sh, se ; XCC id above
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void spin_kernel(Rec* rec, long long ticks) {
  extern __shared__ unsigned char lds[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) lds[0] = (unsigned char)t0;        // the allocation is real
  while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    Rec r; r.t0 = t0; r.t1 = wall_clock64(); r.hw = hw_id(); r.pad = lds[0];
    rec[blockIdx.x] = r;
  }
}

static double ms_between(hipEvent_t a, hipEvent_t b) { float f; CK(hipEventElapsedTime(&f, a, b)); return f; }

int main(int argc, char** argv) {
  const int hog_lds = argc > 1 ? atoi(argv[1]) : 130 * 1024, small_lds = argc > 2 ? atoi(argv[2]) : 19968;
  const int hog_blocks = argc > 3 ? atoi(argv[3]) : 512, small_blocks = argc > 4 ? atoi(argv[4]) : 2048;
  const double hog_us = argc > 5 ? atof(argv[5]) : 400.0, small_us = argc > 6 ? atof(argv[6]) : 50.0;
  const double tick_per_us = 100.0;     // wall_clock64 runs at 100 MHz
  CK(hipFuncSetAttribute((const void*)spin_kernel<512>, hipFuncAttributeMaxDynamicSharedMemorySize, hog_lds));
  CK(hipFuncSetAttribute((const void*)spin_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, small_lds));
  Rec *rh, *rs;
  CK(hipMalloc(&rh, sizeof(Rec) * hog_blocks)); CK(hipMalloc(&rs, sizeof(Rec) * small_blocks));
  hipStream_t sa, sb; CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
  hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  auto hog = [&](hipStream_t s) { hipLaunchKernelGGL(spin_kernel<512>, dim3(hog_blocks), dim3(512), hog_lds, s, rh, (long long)(hog_us * tick_per_us)); };
  auto sm = [&](hipStream_t s) { hipLaunchKernelGGL(spin_kernel<256>, dim3(small_blocks), dim3(256), small_lds, s, rs, (long long)(small_us * tick_per_us)); };
  hog(sa); sm(sb); CK(hipDeviceSynchronize());      // warm-up
  CK(hipEventRecord(e0, sa)); hog(sa); CK(hipEventRecord(e1, sa)); CK(hipDeviceSynchronize());
  const double t_hog = ms_between(e0, e1);
  CK(hipEventRecord(f0, sb)); sm(sb); CK(hipEventRecord(f1, sb)); CK(hipDeviceSynchronize());
  const double t_small = ms_between(f0, f1);
  // both: the hog first, the small kernel right behind it on the other stream
  CK(hipEventRecord(e0, sa)); hog(sa); CK(hipEventRecord(e1, sa));
  CK(hipEventRecord(f0, sb)); sm(sb); CK(hipEventRecord(f1, sb));
  CK(hipDeviceSynchronize());
  const double t_both_h = ms_between(e0, e1), t_both_s = ms_between(f0, f1), t_span = ms_between(e0, f1) > ms_between(e0, e1) ? ms_between(e0, f1) : ms_between(e0, e1);
  std::vector<Rec> H(hog_blocks), S(small_blocks);
  CK(hipMemcpy(H.data(), rh, sizeof(Rec) * hog_blocks, hipMemcpyDeviceToHost));
  CK(hipMemcpy(S.data(), rs, sizeof(Rec) * small_blocks, hipMemcpyDeviceToHost));
  // a small block shares a CU with a hog block if they have the same (xcc, se, sh, cu) and their intervals overlap
  auto cu_of = [](unsigned hw) { return ((hw >> 16) & 0xf) << 12 | ((hw >> 8) & 0xf) | ((hw >> 12) & 0x1) << 4 | ((hw >> 13) & 0x7) << 5; };   // cu[11:8], sh[12], se[15:13]
  int shared = 0, any_overlap = 0;
  for (const Rec& s : S) {
    bool sh = false, ov = false;
    for (const Rec& h : H) {
      const bool overlap = s.t0 < h.t1 && h.t0 < s.t1;
      ov = ov || overlap;
      if (overlap && cu_of(s.hw) == cu_of(h.hw)) { sh = true; break; }
    }
    shared += sh; any_overlap += ov;
  }
  std::vector<int> cus;
  for (const Rec& h : H) cus.push_back(cu_of(h.hw));
  std::sort(cus.begin(), cus.end()); cus.erase(std::unique(cus.begin(), cus.end()), cus.end());
  printf("hog: %d blocks x 512 threads, %d B LDS, %.0f us each; small: %d blocks x 256 threads, %d B LDS, %.0f us each; %zu distinct CUs seen\n", hog_blocks, hog_lds, hog_us,
         small_blocks, small_lds, small_us, cus.size());
  printf("alone: hog %.3f ms, small %.3f ms (sum %.3f)\n", t_hog, t_small, t_hog + t_small);
  printf("together: hog %.3f ms, small %.3f ms, first launch to last completion %.3f ms\n", t_both_h, t_both_s, t_span);
  printf("small blocks that ran while some hog block was running: %d of %d; on the SAME CU as a running hog block: %d\n", any_overlap, small_blocks, shared);
  return 0;
}
