// What bounds the skinny GEMMs of BASELINE configs[1]?  Per-CU ingest of a GEMM-shaped operand stream (every K-step a workgroup pulls
// ROWS rows x 128 B, row stride = K * 2 bytes, 16 workgroups share each row block, as the 96x192-tile launch over (1536, 3072, 15360)
// does) with NO math: LDS-DMA (buffer_load ... lds, 16 B per lane) against global_load_dwordx4 -> VGPR -> ds_write_b128, 4 or 8 waves
// per workgroup, one or two workgroups per CU, ring depth 2-4 K-steps in flight.  Prints GB/s per CU and for the chip.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define LDSP __attribute__((address_space(3)))
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int ROWS, int ST, bool DMA>
__global__ __launch_bounds__(WAVES * 64) void ingest(const char* base, int K, int nblocks_rows, unsigned* sink) {
  constexpr int CH = (ROWS / 8 + WAVES - 1) / WAVES;   // 1-KiB chunks (8 rows x 128 B) per wave per K-step (8 waves x 5 chunks cover 288 rows with 4 spare)
  __shared__ __attribute__((aligned(16))) char smem[ST * CH * WAVES * 1024];
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  // sharing pattern (nblocks_rows): 0 = as the 96x192-tile GEMM over (1536, 3072, 15360): XCD x = blockIdx & 7 runs 4 m-tiles x 8 n-tiles, its
  // A rows (96 per tile) come from m-group x >> 1 and its W rows (192 per tile) from n-half x & 1 — 4 + 8 distinct row blocks per XCD, each
  // read by 8 / 4 workgroups; 1 = every workgroup of an XCD reads the SAME rows (pure L2 hits after the first reader)
  const int x = blockIdx.x & 7, i = (blockIdx.x >> 3) & 31;
  constexpr int RA = ROWS / 3, RW = ROWS - RA;          // 96 + 192 of 288
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffff0u, 0x00020000);
  unsigned voff[12];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
    const int row = (w * CH + c) * 8 + (l >> 3);        // row of this workgroup's stream
    long long grow;                                      // row of the buffer: A rows 0..1535, W rows 1536..4607
    if (nblocks_rows == 1) grow = (long long)x * ROWS + row;
    else if (row < RA) grow = ((x >> 1) * 4 + (i & 3)) * RA + row;
    else grow = 1536 + ((x & 1) * 8 + (i >> 2)) * RW + (row - RA);
    voff[c] = (unsigned)(grow * K * 2 + (l & 7) * 16);
  }
  const int nk = K / 64;
  unsigned acc = 0;
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      LDSP char* d = (LDSP char*)smem + buf * (CH * WAVES * 1024) + (w * CH + i) * 1024;
      if (DMA) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDSP void*)d, 16, voff[i], kt * 128, 0, 0);
      else {
        u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], kt * 128, 0);
        *(LDSP u32x4*)(d + l * 16) = v;
      }
    }
  };
  if (DMA) {
#pragma unroll
    for (int s = 0; s < ST - 1; ++s) stage(s, s);
    int buf = 0, nbuf = ST - 1;
    for (int kt = 0; kt < nk; ++kt) {
      if (ST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (ST == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CH) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * CH) : "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + ST - 1 < nk) stage(nbuf, kt + ST - 1);
      acc += *(const unsigned*)(smem + buf * (CH * WAVES * 1024) + threadIdx.x * 4);      // one LDS read per K-step (keeps the data live)
      buf = buf + 1 == ST ? 0 : buf + 1;
      nbuf = nbuf + 1 == ST ? 0 : nbuf + 1;
    }
  } else {
    // register path: loads of K-step kt+1 are in flight while K-step kt is written to LDS (depth 2 by registers)
    u32x4 r[2][CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) r[0][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], 0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (kt + h + 1 < nk) {
#pragma unroll
          for (int i = 0; i < CH; ++i) r[h ^ 1][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff[i], (kt + h + 1) * 128, 0);
        }
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int i = 0; i < CH; ++i) *(LDSP u32x4*)((LDSP char*)smem + (h & 1) * (CH * WAVES * 1024) + (w * CH + i) * 1024 + l * 16) = r[h][i];
        acc += *(const unsigned*)(smem + (h & 1) * (CH * WAVES * 1024) + threadIdx.x * 4);
      }
    }
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int WAVES, int ROWS, int ST, bool DMA>
static void run(const char* name, const char* buf, int K, int nrb, unsigned* sink, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((ingest<WAVES, ROWS, ST, DMA>), dim3(grid), dim3(WAVES * 64), 0, 0, buf, K, nrb, sink);
  hipEventRecord(a);
  const int it = 10;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL((ingest<WAVES, ROWS, ST, DMA>), dim3(grid), dim3(WAVES * 64), 0, 0, buf, K, nrb, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= it;
  const double bytes = (double)grid * (((ROWS / 8 + WAVES - 1) / WAVES) * WAVES * 8) * K * 2;
  printf("%-34s grid %4d: %7.1f us  %6.2f TB/s chip  %5.1f GB/s per CU (256)\n", name, grid, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}

int main() {
  const int K = 15360, RB = 16;
  const size_t bytes = (size_t)4800 * K * 2;
  char* buf; unsigned* sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 64); hipMemset(buf, 1, bytes);
  for (int pat = 0; pat < 2; ++pat) {
    printf("---- %s\n", pat == 0 ? "GEMM-like sharing (4 A + 8 W row blocks per XCD, 8 / 4 readers each)" : "all 32 workgroups of an XCD read the same rows");
    run<4, 288, 3, true>("DMA  4 waves ST3", buf, K, pat, sink, 256);
    run<4, 288, 4, true>("DMA  4 waves ST4", buf, K, pat, sink, 256);
    run<4, 288, 2, true>("DMA  4 waves ST2", buf, K, pat, sink, 256);
    run<8, 288, 3, true>("DMA  8 waves ST3", buf, K, pat, sink, 256);
    run<4, 288, 2, false>("VGPR 4 waves", buf, K, pat, sink, 256);
    run<8, 288, 2, false>("VGPR 8 waves", buf, K, pat, sink, 256);
  }
  return 0;
}
