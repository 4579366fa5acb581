// hipcc (ROCm 7.2.0, clang 20 / AMDGPU) folds the two results of __builtin_amdgcn_permlane32_swap into one when they are combined:
//     hipcc -O3 --offload-arch=gfx950 --cuda-device-only -S scripts/probe/permlane32_swap_fold_repro.hip -o -
// emits
//     v_permlane32_swap_b32_e32 v1, v2
//     v_mov_b32_e32 v2, v1            <- fmaxf(sw[0], sw[1]) became sw[0]: no v_max_f32 anywhere
//     s_nop 1
//     v_permlane32_swap_b32_e32 v1, v2
//     global_store_dword v0, v1, ...  offset:512
//     global_store_dword v0, v1, ...  offset:768   <- sw2[1] stored from the SAME register as sw2[0]
// (the optimised IR holds one `extractvalue { i32, i32 } %swap, 0` per call; -O0 IR extracts both elements correctly).  The hardware instruction
// swaps the upper 32 lanes of its first operand with the lower 32 lanes of its second: swap(a, b) = ((a.lo | b.lo), (a.hi | b.hi)) — its two
// results differ even for swap(x, x).  Using the results separately (csrc/attention.hip epilogue: {r0[0], r1[0], r0[1], r1[1]} stored as one
// 16-byte piece) compiles correctly.  The 64-query attention kernel's row maximum used the pattern below and saw one lane half's keys only
// (DESIGN.md "Round 4", (iii)); it issues the instruction from inline asm now.
#include <hip/hip_runtime.h>
__global__ void k(float* p) {
  float ra = p[threadIdx.x], rb = p[threadIdx.x + 64];
  const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, ra), __builtin_bit_cast(unsigned, rb), false, false);
  const float mx = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1]));
  const auto sw2 = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mx), __builtin_bit_cast(unsigned, mx), false, false);
  p[threadIdx.x + 128] = __builtin_bit_cast(float, sw2[0]);
  p[threadIdx.x + 192] = __builtin_bit_cast(float, sw2[1]);
}
