// drag_common.h — shared device helpers for libdomainrag_hip.so (gfx950 / CDNA4 only).
//
// Everything in csrc/ is written directly for MI355X: 64-wide wavefronts, MFMA
// matrix cores, LDS-DMA (buffer_load ... lds).  There is no CUDA path and no
// portability layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/domainrag_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;     // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(16))) float f32x16_t;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define DRAG_LDS __attribute__((address_space(3)))

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;

// round-to-nearest-even (the rule of torch's float->bfloat16 cast): gfx950 has it in hardware —
// the native casts below lower to v_cvt_pk_bf16_f32 (one instruction per PAIR in pack2bf)
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// round a float through bf16 (mirrors an intermediate bf16 tensor in the reference's bf16 pipeline)
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// --- activations (fp32 math; match torch's definitions) ---
// sigmoid(z) = 1 / (1 + 2^(-z*log2 e)) on the raw v_exp_f32 / v_rcp_f32 (1 ulp each; saturates correctly:
// z -> -inf gives rcp(inf) = 0, z -> +inf gives rcp(1) = 1)
__device__ __forceinline__ float fast_sigmoid(float z) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z));
}
__device__ __forceinline__ float act_gelu_tanh(float x) {
  // torch gelu(approximate="tanh"): 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715x^3); 0.5*(1+tanh u) = sigmoid(2u)
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  const float u2 = 2.0f * k0 * x * (1.0f + k1 * x * x);
  return x * fast_sigmoid(u2);
}
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f)); }
__device__ __forceinline__ float act_silu(float x) { return x * fast_sigmoid(x); }
__device__ __forceinline__ float act_quick_gelu(float x) { return x * fast_sigmoid(1.702f * x); }


__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case DRAG_ACT_GELU_TANH: return act_gelu_tanh(x);
    case DRAG_ACT_SILU: return act_silu(x);
    case DRAG_ACT_QUICK_GELU: return act_quick_gelu(x);
    case DRAG_ACT_GELU_ERF: return act_gelu_erf(x);
    default: return x;
  }
}

// XCD-aware block remap: dispatcher places block b on XCD b%8 (speed only, never
// correctness).  Gives every XCD a contiguous run of logical tile ids so that tiles
// sharing an operand panel hit the same private L2.  Bijective for any nwg.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, loc = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

// wave-wide reductions (64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, 64));
  return v;
}

// tuning switches (capi.hip): read once from the environment, changeable through drag_set_option
enum { DRAG_OPT_ATTN_SCHED = 0, DRAG_OPT_ATTN_W4 = 1, DRAG_OPT_ATTN_TUNE = 2, DRAG_OPT_ATTN_Q64 = 3, DRAG_OPT_GEMM_KERNEL = 4, DRAG_OPT_LN_GENERIC = 5, DRAG_OPT_GEMM_GROUP_M = 6, DRAG_OPT_TOPK_GRID = 7, DRAG_OPT_TOPK_DEPTH = 8, DRAG_OPT_ATTN_PERSIST = 9, DRAG_OPT_TOPK_SELECT = 10, DRAG_OPT_TOPK_DENSE_SAMPLE = 11, DRAG_OPT_TOPK_QT = 12, DRAG_OPT_GEMM_PAIR = 13, DRAG_OPT_GEMM_EPILOGUE = 14, DRAG_OPT_TOPK_PATH = 15, DRAG_OPT_TOPK_QREG = 16, DRAG_OPT_GEMM_W4 = 17, DRAG_OPT_ATTN_WALK = 18, DRAG_OPT_GEMM_SPLITK = 19, DRAG_OPT_ATTN_GEN = 20, DRAG_OPT_COUNT = 21 };
// measured defaults (scripts/bench_attn.py, B=8 S=5337: schedule 0 1112, 1 1139, 2 1165 TFLOP/s; 16-byte epilogue stores +0.1 % there,
// +4...7 % at S = 1753 / 729; static wave priority: no gain, -5 % on short sequences).  All settings give the same bits.
#define DRAG_ATTN_SCHED_DEFAULT 2
#define DRAG_ATTN_TUNE_DEFAULT 2
int drag_opt(int idx);
// Experiments: code paths whose A/B is recorded and negative (the persistent attention kernel -1 %, the V-one-step-ahead attention
// schedule 0, several query tiles per top-k workgroup -20 %) are compiled only into a
// -DDRAG_EXPERIMENTS library (DRAG_EXPERIMENTS=1 python -m domain_rag_amd.build); the product library neither carries their kernels nor accepts their
// switches (drag_set_option says so), so nothing has to keep them bit-identical through every test run.
#ifdef DRAG_EXPERIMENTS
#define DRAG_EXP 1
#else
#define DRAG_EXP 0
#endif

// error plumbing shared by the C ABI
void drag_set_error(const char* msg);
#define DRAG_CHECK(cond, msg)          \
  do {                                 \
    if (!(cond)) {                     \
      drag_set_error(msg);             \
      return -1;                       \
    }                                  \
  } while (0)
#define DRAG_LAUNCH_CHECK()                                  \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) {                                 \
      drag_set_error(hipGetErrorString(e__));                \
      return -2;                                             \
    }                                                        \
  } while (0)
