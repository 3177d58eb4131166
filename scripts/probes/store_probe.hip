// Store-pattern probe (build: hipcc --offload-arch=gfx950 -O3 store_probe.hip -o store_probe): every wave writes
// 16 KiB (16 dwordx4 store instructions) of a bf16 [M][N] matrix the way a 256x256 GEMM epilogue tile does, with two
// lane->address maps: 0 = 16 rows x 64 B per instruction, 1 = 8 rows x 128 B.  Measured: 6.0 vs 6.6 TB/s (MI355X).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int MODE>
__global__ __launch_bounds__(512) void k(uint4* out, long ld16 /* row stride in 16-B units */) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wr = wid >> 2, wc = wid & 3;
  const long tile_m = (long)blockIdx.y * 256 + wr * 128, tile_n16 = (long)blockIdx.x * 32 + wc * 8;  // 64 cols = 8 x 16 B
  uint4 v = make_uint4(lane, wid, blockIdx.x, blockIdx.y);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    long row, c16;
    if (MODE == 0) { row = (i >> 1) * 16 + (lane & 15); c16 = (i & 1) * 4 + (lane >> 4); }
    else if (MODE == 1) { row = i * 8 + (lane >> 3); c16 = lane & 7; }
    else { row = i * 8 + (lane >> 3); c16 = lane & 7; }
    out[(tile_m + row) * ld16 + tile_n16 + c16] = v;
  }
}
int main() {
  const long M = 34176 / 256 * 256, N = 4096;
  uint4* d; hipMalloc(&d, M * N * 2);
  dim3 grid(N / 256, M / 256);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int mode = 0; mode < 2; ++mode) for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    for (int it = 0; it < 10; ++it) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, grid, dim3(512), 0, 0, d, N / 8);
      else hipLaunchKernelGGL(k<1>, grid, dim3(512), 0, 0, d, N / 8);
    }
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("mode %d: %.1f us per launch, %.2f TB/s\n", mode, ms * 100, M * N * 2 / (ms / 10 * 1e-3) / 1e12);
  }
  return 0;
}
