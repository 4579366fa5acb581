// What does a CU's store path sustain for a GEMM epilogue?  One 512-thread workgroup per CU writes 256x256 bf16 "tiles" (128 KiB each:
// 16 dwordx4 store instructions per wave) of a [M, N] bf16 matrix, tile after tile in the persistent kernel's order, with NO compute in
// between.  Patterns of one store instruction (64 lanes x 16 B = 1 KiB):
//   0: 8 rows x 128 B  (the staged epilogue of gemm_bf16_t256: a wave owns a 128 x 64 sub-tile -> one line per row)
//   1: 2 rows x 512 B  (whole tile rows: 4 contiguous lines per row)
//   2: 16 rows x 64 B  (half lines: the v_permlane16_swap form without an LDS transpose)
//   3: 1 KiB contiguous (a tile stored as one dense 128 KiB block — what a blocked output layout would give)
// Prints bytes / clk / CU at the measured duration and an assumed 2.0 GHz, and GB/s for the chip.
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/probe_store.hip -o scripts/probe/probe_store
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int AUX>
__global__ __launch_bounds__(512) void store_tiles(char* C, int tiles_m, int tiles_n, long long ldc_bytes, int reps) {
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  const int wr = w >> 2, wc = w & 3;
  const int nt = tiles_m * tiles_n;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)C, 0, 0xfffffff0u, 0x00020000);
  u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
  for (int r = 0; r < reps; ++r)
    for (int t = blockIdx.x; t < nt; t += gridDim.x) {
      const int tm = t % tiles_m, tn = t / tiles_m;
      const long long base = (long long)tm * 256 * ldc_bytes + (long long)tn * 512;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        long long off;
        if (PAT == 0) off = base + (long long)(wr * 128 + i * 8 + (l >> 3)) * ldc_bytes + wc * 128 + (l & 7) * 16;
        else if (PAT == 1) off = base + (long long)(w * 32 + i * 2 + (l >> 5)) * ldc_bytes + (l & 31) * 16;
        else if (PAT == 2) off = base + (long long)(wr * 128 + (i >> 1) * 16 + (l >> 2)) * ldc_bytes + wc * 128 + (i & 1) * 64 + (l & 3) * 16;
        else off = (long long)t * 131072 + (w * 16 + i) * 1024 + l * 16;
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)off, 0, AUX);
        v[0] += 1;
      }
    }
}

int main() {
  const int M = 42752, N = 3072;                 // 167 x 12 tiles
  const int tiles_m = M / 256, tiles_n = N / 256;
  char* C;
  hipMalloc(&C, (size_t)M * N * 2);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 8;
  int grid = 256;
  auto run = [&](auto kern, const char* name) {
    for (int it = 0; it < 2; ++it) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kern, dim3(grid), dim3(512), 0, 0, C, tiles_m, tiles_n, (long long)N * 2, reps);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)reps * tiles_m * tiles_n * 131072.0;
    printf("grid %3d %-28s %8.1f us  %7.1f GB/s chip  %6.1f GB/s per CU  = %5.1f B/clk/CU at 2.0 GHz   (%.2f us per 128 KiB tile and CU)\n", grid, name, ms * 1e3,
           bytes / ms / 1e6, bytes / ms / 1e6 / grid, bytes / ms / 1e6 / grid / 2.0, ms * 1e3 / (reps * tiles_m * tiles_n / (double)grid));
  };
  for (int g : {8, 16, 32, 64, 128}) {
    grid = g;
    run(store_tiles<0, 0>, "8 rows x 128 B");
    run(store_tiles<3, 0>, "1 KiB contiguous");
  }
  grid = 256;
  run(store_tiles<0, 0>, "8 rows x 128 B");
  run(store_tiles<1, 0>, "2 rows x 512 B");
  run(store_tiles<2, 0>, "16 rows x 64 B");
  run(store_tiles<3, 0>, "1 KiB contiguous");
  run(store_tiles<0, 2>, "8 rows x 128 B, nt");
  run(store_tiles<0, 16>, "8 rows x 128 B, sc1");
  run(store_tiles<0, 1>, "8 rows x 128 B, sc0");
  run(store_tiles<0, 17>, "8 rows x 128 B, sc0 sc1");
  return 0;
}
