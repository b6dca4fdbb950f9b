// tools/probe_store_rate.hip — what a CU's store path sustains, in the patterns a GEMM tile's epilogue produces.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_store_rate.hip -o gpurun_out/probe_store_rate ; run on the GPU box.
// One block of 512 threads per CU (130 KiB of LDS keeps it alone there); each wave writes its 128 x 64 bf16 sub-tile of a 256 x 256
// output tile REP times, either as rows (4 rows x 128 B per instruction: the LDS-turned epilogue), as the swapped-operand form
// (16 rows x 64 B per instruction, two instructions per 16 rows), or as 2-byte elements (the direct epilogue).  Patches are 128 KiB per CU and
// rewritten.  argv: ldc [CUs storing at the same time].
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(uint16_t* C, int ldc, int rep, int tiles_per_cu) {
  __shared__ unsigned char pad[130 * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3, q = lane >> 4, r = lane & 15;
  if (tid == 100000) pad[0] = 1;
  u32x4_t v = {(unsigned)tid, 1u, 2u, 3u};
  for (int it = 0; it < rep; ++it) {
    uint16_t* base = C + ((size_t)blockIdx.x * tiles_per_cu + (it % tiles_per_cu)) * 256 * (size_t)ldc;     // a 256-row band per (CU, tile)
    if (MODE == 0) {            // rows: lane (q, r): row t*4+q, 8 bytes at column 4r  (16 lanes = 128 B of a row)
#pragma unroll 4
      for (int t = 0; t < 32; ++t) {
        uint16_t* p = base + (size_t)(wm * 128 + t * 4 + q) * ldc + wn * 64 + r * 4;
        *(uint2*)p = make_uint2(v.x + t, v.y);
      }
    } else if (MODE == 1) {     // swapped operands: lane (q, r): row m*16+r, 16 bytes at column 8q (+32)
#pragma unroll 4
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          uint16_t* p = base + (size_t)(wm * 128 + m * 16 + r) * ldc + wn * 64 + 32 * h + 8 * q;
          *(u32x4_t*)p = u32x4_t{v.x + m, v.y, v.z, v.w};
        }
    } else if (MODE == 2) {     // rows, 16 bytes per lane: lane l: row t*8 + l/8, column 8 (l%8): 8 rows x 128 B per instruction
#pragma unroll 4
      for (int t = 0; t < 16; ++t) {
        uint16_t* p = base + (size_t)(wm * 128 + t * 8 + (lane >> 3)) * ldc + wn * 64 + (lane & 7) * 8;
        *(u32x4_t*)p = u32x4_t{v.x + t, v.y, v.z, v.w};
      }
    } else {                    // direct: lane (q, r): column r of rows 4q+i, 2 bytes
#pragma unroll 2
      for (int m = 0; m < 8; ++m)
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int i = 0; i < 4; ++i) base[(size_t)(wm * 128 + m * 16 + 4 * q + i) * ldc + wn * 64 + n * 16 + r] = (uint16_t)(v.x + i);
    }
    __syncthreads();            // a tile's stores are issued together, as an epilogue does
  }
}

int main(int argc, char** argv) {
  const int ldc = argc > 1 ? atoi(argv[1]) : 4096, rep = 64;
  const int nblk = argc > 2 ? atoi(argv[2]) : 256;       // blocks (= CUs) storing at the same time
  for (int tiles_per_cu = 1; tiles_per_cu <= 64; tiles_per_cu *= 8) {
    uint16_t* C;
    const size_t bytes = (size_t)256 * tiles_per_cu * 256 * ldc * 2;
    if (hipMalloc(&C, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(C, 0, bytes);
    for (int mode = 0; mode < 4; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      for (int w = 0; w < 2; ++w) {
        if (w) hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nblk), dim3(512), 0, 0, C, ldc, rep, tiles_per_cu);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nblk), dim3(512), 0, 0, C, ldc, rep, tiles_per_cu);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nblk), dim3(512), 0, 0, C, ldc, rep, tiles_per_cu);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(nblk), dim3(512), 0, 0, C, ldc, rep, tiles_per_cu);
        if (w) hipEventRecord(e1);
      }
      hipDeviceSynchronize();
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      const double us_tile = ms * 1e3 / rep, gbs_cu = 131072.0 / (us_tile * 1e-6) / 1e9;
      const char* names[4] = {"rows 8 B/lane (4 rows x 128 B)", "swapped 16 B/lane (16 rows x 64 B)", "rows 16 B/lane (8 rows x 128 B)", "direct 2 B/lane"};
      printf("%3d CUs  ldc %5d  distinct tiles per CU %2d  %-36s %7.2f us per 128 KiB tile  %6.1f GB/s per CU  %5.2f TB/s together\n", nblk, ldc, tiles_per_cu,
             names[mode], us_tile, gbs_cu, gbs_cu * nblk / 1e3);
    }
    hipFree(C);
  }
  return 0;
}
