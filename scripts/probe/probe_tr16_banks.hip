// throughput of ds_read_b64_tr_b16 under different per-lane address patterns (bank behaviour of the transpose read is not the
// plain (addr / 4) % 64 rule): 8 waves per workgroup, one workgroup per CU, each wave issues ROUNDS x 16 reads.
// Prints ns per wave-instruction per CU-wave for every pattern; pattern 0 (8 * lane: one contiguous 512-byte run) is the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define LDSP __attribute__((address_space(3)))
constexpr int ROUNDS = 2000;
__global__ __launch_bounds__(512) void k(const int* addr, int* sink, int b128) {
  __shared__ __attribute__((aligned(16))) char lds[65536];
  for (int i = threadIdx.x; i < 65536 / 4; i += 512) ((int*)lds)[i] = i;
  __syncthreads();
  const int a = addr[threadIdx.x & 63] + (threadIdx.x >> 6) * 0;      // all waves same pattern
  int acc = 0;
  if (!b128) {
    for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDSP s16x4*)((LDSP char*)lds + a + (i & 3) * 4096 + (i >> 2) * 2048 * 0 + ((i >> 2) & 1) * 2048));
        acc += v[0] + v[3];
      }
      asm volatile("" ::: "memory");
    }
  } else {
    for (int r = 0; r < ROUNDS; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        typedef int i32x4 __attribute__((ext_vector_type(4)));
        i32x4 v = *(LDSP i32x4*)((LDSP char*)lds + (threadIdx.x & 63) * 16 + i * 4096);
        acc += v[0] + v[3];
      }
      asm volatile("" ::: "memory");
    }
  }
  sink[blockIdx.x * 512 + threadIdx.x] = acc;
}
int main() {
  int* d_addr; int* d_sink;
  hipMalloc(&d_addr, 64 * 4); hipMalloc(&d_sink, 256 * 512 * 4);
  std::vector<std::vector<int>> pats; std::vector<const char*> names;
  auto add = [&](const char* n, auto f) { std::vector<int> v(64); for (int l = 0; l < 64; ++l) v[l] = f(l); pats.push_back(v); names.push_back(n); };
  add("0 contiguous 8*lane", [](int l) { return 8 * l; });
  add("1 rows 256 B apart, units XOR 4*(key&3)  (first VROW kernel)", [](int l) {
    int si = l & 15, g1 = (l >> 4) & 1, hh = l >> 5; return (4 * hh + (si >> 2)) * 256 + ((4 * (0 ^ (si >> 2)) + 2 * g1 + ((si >> 1) & 1)) << 4) + (si & 1) * 8; });
  add("2 rows 256 B apart, no swizzle", [](int l) { int si = l & 15, g1 = (l >> 4) & 1, hh = l >> 5; return (4 * hh + (si >> 2)) * 256 + (2 * g1 + ((si >> 1) & 1)) * 16 + (si & 1) * 8; });
  add("3 [d16 block][key][32 B]: groups 0/1 2048 B apart, quads contiguous", [](int l) { int si = l & 15, g1 = (l >> 4) & 1, hh = l >> 5; return g1 * 2048 + (4 * hh + (si >> 2)) * 32 + (si & 3) * 8; });
  add("4 as 3 with +128 B for odd d16 blocks", [](int l) { int si = l & 15, g1 = (l >> 4) & 1, hh = l >> 5; return g1 * (2048 + 128) + (4 * hh + (si >> 2)) * 32 + (si & 3) * 8; });
  add("5 [key quad][d32 block][4 keys][64 B]: a group reads 4 x 32 B at stride 64", [](int l) { int si = l & 15, g1 = (l >> 4) & 1, hh = l >> 5; return hh * 256 + (si >> 2) * 64 + g1 * 32 + (si & 3) * 8; });
  add("6 [key quad][d16 block g1][128 B], hh quads 256 B apart", [](int l) { int si = l & 15, g1 = (l >> 4) & 1, hh = l >> 5; return hh * 256 + g1 * 128 + (si >> 2) * 32 + (si & 3) * 8; });
  add("7 as 6 but hh 128 B apart, g1 256 B apart", [](int l) { int si = l & 15, g1 = (l >> 4) & 1, hh = l >> 5; return g1 * 256 + hh * 128 + (si >> 2) * 32 + (si & 3) * 8; });
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int b128 = 0; b128 < 2; ++b128)
    for (size_t p = 0; p < (b128 ? 1 : pats.size()); ++p) {
      hipMemcpy(d_addr, pats[p].data(), 256, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d_addr, d_sink, b128);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, d_addr, d_sink, b128);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double n = (double)ROUNDS * (b128 ? 8 : 16) * 8;   // wave-instructions per CU
      printf("%-75s %.2f ns per wave-instruction per CU (%.1f B/ns/CU)\n", b128 ? "ds_read_b128 16*lane (reference)" : names[p], ms * 1e6 / n,
             (b128 ? 1024 : 512) / (ms * 1e6 / n));
    }
  return 0;
}
