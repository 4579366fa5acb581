// probe of ds_read_b64_tr_b16 on gfx950: which source lane's element lands in which (lane, element) of the result.
// LDS holds lds[i] = i (16-bit); every lane supplies its own byte address.  Prints the observed mapping for two address patterns.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(uint16_t* out, const int* addr) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)((__attribute__((address_space(3))) char*)lds + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}
int main() {
  int h_addr[64];
  uint16_t h_out[256];
  int* d_addr; uint16_t* d_out;
  hipMalloc(&d_addr, sizeof(h_addr)); hipMalloc(&d_out, sizeof(h_out));
  for (int pat = 0; pat < 2; ++pat) {
    // pattern 0: lane s reads the 4 elements at element index 4 * s (a dense 64 x 4 block)
    // pattern 1: lane s reads 4 elements at 1000 * (s >> 4) + 100 * ((s & 15) >> 2) + 4 * (s & 3): "row (s&15)>>2, column piece s&3" of its group
    for (int s = 0; s < 64; ++s) h_addr[s] = 2 * (pat == 0 ? 4 * s : 1000 * (s >> 4) + 100 * ((s & 15) >> 2) + 4 * (s & 3));
    hipMemcpy(d_addr, h_addr, sizeof(h_addr), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_addr);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("pattern %d\n", pat);
    int ok = 1;
    for (int i = 0; i < 64; ++i) {
      printf("lane %2d:", i);
      for (int j = 0; j < 4; ++j) {
        printf(" %5d", h_out[i * 4 + j]);
        // hypothesis: out[lane i][j] = element (i & 3) of source lane 16 * (i >> 4) + 4 * j + ((i & 15) >> 2)
        const int src = 16 * (i >> 4) + 4 * j + ((i & 15) >> 2);
        const int want = h_addr[src] / 2 + (i & 3);
        ok &= (h_out[i * 4 + j] == want);
      }
      printf("\n");
    }
    printf("hypothesis out[i][j] = in[lane 16(i>>4) + 4j + ((i&15)>>2)][i&3]: %s\n", ok ? "HOLDS" : "FAILS");
  }
  return 0;
}
