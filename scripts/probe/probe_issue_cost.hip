// Issue cost of the instructions the hand-placed attention loop is made of, for ONE wave per SIMD (and two), on gfx950:
//     hipcc -O3 --offload-arch=gfx950 scripts/probe/probe_issue_cost.hip -o /tmp/probe_issue && /tmp/probe_issue
// Every kernel runs ITER iterations of a block of 32 independent instructions of one kind (or a pattern) and reads s_memtime around the
// loop; cycles per instruction = (t1 - t0) / (ITER * instructions per block), median over the waves of the grid (256 workgroups).
// Patterns "mfma + k fillers" show how many VALU instructions hide behind one v_mfma_f32_32x32x16_bf16 of a lone wave.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
constexpr int ITER = 2000;

#define R8(x) x x x x x x x x
#define R32(x) R8(x) R8(x) R8(x) R8(x)

template <int KIND>
__global__ __launch_bounds__(512) void probe(long long* out, float seed, const void* src) {
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
  float a0 = seed + threadIdx.x, a1 = a0 * 2, a2 = a0 * 3, a3 = a0 * 5, b = seed * 0.5f, c = 1.0001f;
  f32x2_t p0 = {a0, a1}, p1 = {a2, a3}, pc = {c, b};
  f32x16_t acc0 = {}, acc1 = {}, acc2 = {}, acc3 = {};
  bf16x8_t fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (__bf16)(seed + i); fb[i] = (__bf16)(seed - i); }
  unsigned w0 = 0, w1 = 0;
  __shared__ __attribute__((aligned(16))) char lds[65536];
  const unsigned lds_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds + (threadIdx.x & 63) * 16 + ((threadIdx.x >> 6) & 3) * 1024;
  bf16x8_t r0, r1, r2, r3, fq;
  for (int i = 0; i < 8; ++i) fq[i] = (__bf16)(seed + 2 * i);
  asm volatile("; q fragment -> AGPR" : "=a"(fq) : "0"(fq));
  if (seed == 12345.f) lds[threadIdx.x] = 1;
  asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(p0), "+v"(p1), "+v"(pc), "+v"(fa), "+v"(fb), "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3));
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; ++it) {
    if constexpr (KIND == 0) asm volatile(R8("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(b));
    if constexpr (KIND == 1) asm volatile(R8("v_pk_fma_f32 %0, %0, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %0, %0, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\tv_pk_fma_f32 %1, %1, %2, %2 op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t") : "+v"(p0), "+v"(p1) : "v"(pc));
    if constexpr (KIND == 2) asm volatile(R8("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    if constexpr (KIND == 3) asm volatile(R8("v_exp_f32 %0, %0\n\tv_fma_f32 %1, %1, %4, %5\n\tv_exp_f32 %2, %2\n\tv_fma_f32 %3, %3, %4, %5\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(b));
    if constexpr (KIND == 4) asm volatile(R8("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
    if constexpr (KIND == 5) asm volatile(R8("v_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %2\n\tv_pk_add_f32 %0, %0, %2\n\tv_pk_add_f32 %1, %1, %2\n\t") : "+v"(p0), "+v"(p1) : "v"(pc));
    if constexpr (KIND == 6) asm volatile(R8("v_cvt_pk_bf16_f32 %0, %2, %3\n\tv_cvt_pk_bf16_f32 %1, %4, %5\n\tv_cvt_pk_bf16_f32 %0, %3, %2\n\tv_cvt_pk_bf16_f32 %1, %5, %4\n\t") : "+v"(w0), "+v"(w1) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    if constexpr (KIND == 7) asm volatile(R8("v_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %1, %1, %4, %5\n\tv_max3_f32 %2, %2, %4, %5\n\tv_max3_f32 %3, %3, %4, %5\n\t") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(b));
    if constexpr (KIND == 8) asm volatile(R32("s_nop 0\n\t"));
    // 32 MFMAs on four independent accumulators
    if constexpr (KIND == 9) asm volatile(R8("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\tv_mfma_f32_32x32x16_bf16 %1, %4, %5, %1\n\tv_mfma_f32_32x32x16_bf16 %2, %4, %5, %2\n\tv_mfma_f32_32x32x16_bf16 %3, %4, %5, %3\n\t") : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3) : "v"(fa), "v"(fb));
#define MF(n) "v_mfma_f32_32x32x16_bf16 %" #n ", %8, %9, %" #n "\n\t"
#define F1 "v_fma_f32 %4, %4, %10, %11\n\t"
#define F2 F1 "v_fma_f32 %5, %5, %10, %11\n\t"
#define F4 F2 "v_fma_f32 %6, %6, %10, %11\n\tv_fma_f32 %7, %7, %10, %11\n\t"
#define MOPS : "+v"(acc0), "+v"(acc1), "+v"(acc2), "+v"(acc3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(fa), "v"(fb), "v"(c), "v"(b)
    // 8 x (4 MFMAs, each followed by k v_fma fillers): cycles are reported per MFMA
    if constexpr (KIND == 10) asm volatile(R8(MF(0) F2 MF(1) F2 MF(2) F2 MF(3) F2) MOPS);
    if constexpr (KIND == 11) asm volatile(R8(MF(0) F4 MF(1) F4 MF(2) F4 MF(3) F4) MOPS);
    if constexpr (KIND == 12) asm volatile(R8(MF(0) F4 F2 MF(1) F4 F2 MF(2) F4 F2 MF(3) F4 F2) MOPS);
    if constexpr (KIND == 13) asm volatile(R8(MF(0) F4 F4 MF(1) F4 F4 MF(2) F4 F4 MF(3) F4 F4) MOPS);
    // one exponential + three v_fma behind each MFMA (the attention loop's gap), and two exponentials + two v_fma
#define E1 "v_exp_f32 %4, %4\n\t"
#define E2 "v_exp_f32 %5, %5\n\t"
    if constexpr (KIND == 14) asm volatile(R8(MF(0) E1 "v_fma_f32 %5, %5, %10, %11\n\tv_fma_f32 %6, %6, %10, %11\n\tv_fma_f32 %7, %7, %10, %11\n\t" MF(1) E1 "v_fma_f32 %5, %5, %10, %11\n\tv_fma_f32 %6, %6, %10, %11\n\tv_fma_f32 %7, %7, %10, %11\n\t" MF(2) E1 "v_fma_f32 %5, %5, %10, %11\n\tv_fma_f32 %6, %6, %10, %11\n\tv_fma_f32 %7, %7, %10, %11\n\t" MF(3) E1 "v_fma_f32 %5, %5, %10, %11\n\tv_fma_f32 %6, %6, %10, %11\n\tv_fma_f32 %7, %7, %10, %11\n\t") MOPS);
    if constexpr (KIND == 15) asm volatile(R8(MF(0) E1 "v_fma_f32 %6, %6, %10, %11\n\t" E2 "v_fma_f32 %7, %7, %10, %11\n\t" MF(1) E1 "v_fma_f32 %6, %6, %10, %11\n\t" E2 "v_fma_f32 %7, %7, %10, %11\n\t" MF(2) E1 "v_fma_f32 %6, %6, %10, %11\n\t" E2 "v_fma_f32 %7, %7, %10, %11\n\t" MF(3) E1 "v_fma_f32 %6, %6, %10, %11\n\t" E2 "v_fma_f32 %7, %7, %10, %11\n\t") MOPS);
    // 32 independent LDS fragment reads (the attention loop's ds_read_b128), retired once per block
    if constexpr (KIND == 16) {
      // (16 distinct destinations: reads into ONE register serialise on it — 16 cycles each instead of 4-5)
      bf16x8_t q[16];
      asm volatile("ds_read_b128 %0, %16\n\tds_read_b128 %1, %16 offset:1024\n\tds_read_b128 %2, %16 offset:2048\n\tds_read_b128 %3, %16 offset:3072\n\t"
                   "ds_read_b128 %4, %16 offset:4096\n\tds_read_b128 %5, %16 offset:5120\n\tds_read_b128 %6, %16 offset:6144\n\tds_read_b128 %7, %16 offset:7168\n\t"
                   "ds_read_b128 %8, %16 offset:8192\n\tds_read_b128 %9, %16 offset:9216\n\tds_read_b128 %10, %16 offset:10240\n\tds_read_b128 %11, %16 offset:11264\n\t"
                   "ds_read_b128 %12, %16 offset:12288\n\tds_read_b128 %13, %16 offset:13312\n\tds_read_b128 %14, %16 offset:14336\n\tds_read_b128 %15, %16 offset:15360\n\t"
                   "ds_read_b128 %0, %16 offset:16384\n\tds_read_b128 %1, %16 offset:17408\n\tds_read_b128 %2, %16 offset:18432\n\tds_read_b128 %3, %16 offset:19456\n\t"
                   "ds_read_b128 %4, %16 offset:20480\n\tds_read_b128 %5, %16 offset:21504\n\tds_read_b128 %6, %16 offset:22528\n\tds_read_b128 %7, %16 offset:23552\n\t"
                   "ds_read_b128 %8, %16 offset:24576\n\tds_read_b128 %9, %16 offset:25600\n\tds_read_b128 %10, %16 offset:26624\n\tds_read_b128 %11, %16 offset:27648\n\t"
                   "ds_read_b128 %12, %16 offset:28672\n\tds_read_b128 %13, %16 offset:29696\n\tds_read_b128 %14, %16 offset:30720\n\tds_read_b128 %15, %16 offset:31744\n\t"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(q[6]), "=&v"(q[7]), "=&v"(q[8]), "=&v"(q[9]),
                     "=&v"(q[10]), "=&v"(q[11]), "=&v"(q[12]), "=&v"(q[13]), "=&v"(q[14]), "=&v"(q[15])
                   : "v"(lds_addr));
      r0 = q[0]; r1 = q[5]; r2 = q[10]; r3 = q[15];
    }
    // the 64-query attention kernel's step (steps 4..15) four times: O accumulators in AGPRs, S in VGPRs, fillers in the kernel's order;
    // KIND 17 with the two fragment reads and the counted wait, KIND 18 without them
#define STEP(o0, o1, s0, s1, RD1, RD2, WAIT)                                                                                           \
  WAIT "v_fma_f32 %[y0], %[x0], %[c], %[b]\n\tv_fma_f32 %[y1], %[x1], %[c], %[b]\n\t"                                                   \
  "v_mfma_f32_32x32x16_bf16 %[" #o0 "], %[fa], %[fb], %[" #o0 "]\n\t" RD1 "v_exp_f32 %[y0], %[y0]\n\t"                                 \
  "v_mfma_f32_32x32x16_bf16 %[" #o1 "], %[fa], %[fb], %[" #o1 "]\n\t" RD2 "v_fma_f32 %[y2], %[x2], %[c], %[b]\n\tv_exp_f32 %[y1], %[y1]\n\tv_fma_f32 %[y3], %[x3], %[c], %[b]\n\t" \
  "v_mfma_f32_32x32x16_bf16 %[" #s0 "], %[fa], %[fq], %[" #s0 "]\n\t"                                                                  \
  "v_add_f32 %[t0], %[y0], %[y1]\n\tv_exp_f32 %[y2], %[y2]\n\tv_add_f32 %[ps0], %[ps0], %[t0]\n\tv_exp_f32 %[y3], %[y3]\n\t"          \
  "v_mfma_f32_32x32x16_bf16 %[" #s1 "], %[fa], %[fq], %[" #s1 "]\n\t"                                                                  \
  "v_cvt_pk_bf16_f32 %[w0], %[y0], %[y1]\n\tv_add_f32 %[t1], %[y2], %[y3]\n\tv_add_f32 %[ps1], %[ps1], %[t1]\n\tv_cvt_pk_bf16_f32 %[w1], %[y2], %[y3]\n\t"
#define STEP_OPS                                                                                                                        \
  : [oa] "+a"(acc0), [ob] "+a"(acc1), [sa] "+v"(acc2), [sb] "+v"(acc3), [y0] "=&v"(y0), [y1] "=&v"(y1), [y2] "=&v"(y2), [y3] "=&v"(y3), \
    [t0] "=&v"(t0), [t1] "=&v"(t1), [ps0] "+v"(a0), [ps1] "+v"(a1), [w0] "=&v"(w0), [w1] "=&v"(w1), [r0] "=&v"(r0), [r1] "=&v"(r1)       \
  : [x0] "v"(a2), [x1] "v"(a3), [x2] "v"(b), [x3] "v"(c), [c] "s"(seed), [b] "v"(b), [fa] "v"(fa), [fb] "v"(fb), [fq] "a"(fq), [ad] "v"(lds_addr)
#define RDA "ds_read_b128 %[r0], %[ad]\n\t"
#define RDB "ds_read_b128 %[r1], %[ad] offset:8192\n\t"
    if constexpr (KIND == 17) {
      float y0, y1, y2, y3, t0, t1;
      asm volatile(STEP(oa, ob, sa, sb, RDA, RDB, "s_waitcnt lgkmcnt(2)\n\t") STEP(oa, ob, sa, sb, RDA, RDB, "s_waitcnt lgkmcnt(2)\n\t")
                   STEP(oa, ob, sa, sb, RDA, RDB, "s_waitcnt lgkmcnt(2)\n\t") STEP(oa, ob, sa, sb, RDA, RDB, "s_waitcnt lgkmcnt(2)\n\t") "s_waitcnt lgkmcnt(0)" STEP_OPS);
    }
    // KIND 19: the step with one LDS-DMA piece (buffer_load_dwordx4 ... lds, 1 KiB per wave, source L2-resident) behind its third MFMA, as in
    // steps 0..7 of the kernel; KIND 20: two pieces per step
    if constexpr (KIND == 19 || KIND == 20) {
      float y0, y1, y2, y3, t0, t1;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        asm volatile(STEP(oa, ob, sa, sb, RDA, RDB, "s_waitcnt lgkmcnt(2)\n\t") STEP_OPS);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 32768 + k * 4096 + (threadIdx.x >> 6) * 1024), 16, (threadIdx.x & 63) * 16 + k * 1024, 0, 0, 0);
        if constexpr (KIND == 20)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds + 49152 + k * 4096 + (threadIdx.x >> 6) * 1024), 16, (threadIdx.x & 63) * 16 + k * 1024 + 8192, 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    // KIND 21: the kernel's steps 0..3 (two S MFMAs, the same 14 VALU, two reads); KIND 22: its top block (8 P V MFMAs, 4 v_max3 behind each)
#define STEP_LO(s0, s1)                                                                                                                 \
  "s_waitcnt lgkmcnt(2)\n\tv_fma_f32 %[y0], %[x0], %[c], %[b]\n\tv_fma_f32 %[y1], %[x1], %[c], %[b]\n\t"                                \
  "v_mfma_f32_32x32x16_bf16 %[" #s0 "], %[fa], %[fq], %[" #s0 "]\n\t" RDA RDB                                                            \
  "v_exp_f32 %[y0], %[y0]\n\tv_fma_f32 %[y2], %[x2], %[c], %[b]\n\tv_exp_f32 %[y1], %[y1]\n\tv_fma_f32 %[y3], %[x3], %[c], %[b]\n\tv_add_f32 %[t0], %[y0], %[y1]\n\t" \
  "v_mfma_f32_32x32x16_bf16 %[" #s1 "], %[fa], %[fq], %[" #s1 "]\n\t"                                                                  \
  "v_exp_f32 %[y2], %[y2]\n\tv_add_f32 %[ps0], %[ps0], %[t0]\n\tv_exp_f32 %[y3], %[y3]\n\tv_cvt_pk_bf16_f32 %[w0], %[y0], %[y1]\n\t"  \
  "v_add_f32 %[t1], %[y2], %[y3]\n\tv_add_f32 %[ps1], %[ps1], %[t1]\n\tv_cvt_pk_bf16_f32 %[w1], %[y2], %[y3]\n\t"
    if constexpr (KIND == 21) {
      float y0, y1, y2, y3, t0, t1;
      asm volatile(STEP_LO(sa, sb) STEP_LO(sa, sb) STEP_LO(sa, sb) STEP_LO(sa, sb) "s_waitcnt lgkmcnt(0)" STEP_OPS);
    }
#define TOPM(o) "v_mfma_f32_32x32x16_bf16 %[" #o "], %[fa], %[fb], %[" #o "]\n\tv_max3_f32 %[y0], %[x0], %[x1], %[x2]\n\tv_max3_f32 %[y1], %[x1], %[x2], %[x3]\n\tv_max3_f32 %[y2], %[x0], %[x2], %[x3]\n\tv_max3_f32 %[y3], %[x0], %[x1], %[x3]\n\t"
    if constexpr (KIND == 22) {
      float y0, y1, y2, y3, t0, t1;
      asm volatile(RDA RDB TOPM(oa) TOPM(ob) TOPM(oa) TOPM(ob) TOPM(oa) TOPM(ob) TOPM(oa) TOPM(ob) "s_waitcnt lgkmcnt(0)\n\t"
                   "v_max_f32 %[t0], %[y0], %[y1]\n\tv_max_f32 %[t1], %[y2], %[y3]\n\ts_nop 1\n\tv_permlane32_swap_b32 %[t0], %[t1]\n\tv_max_f32 %[t0], %[t0], %[t1]\n\tv_mov_b32 %[t1], %[t0]\n\t"
                   "s_nop 1\n\tv_permlane32_swap_b32 %[t0], %[t1]\n\tv_sub_f32 %[w0], %[t0], %[ps0]\n\tv_mul_f32 %[w0], %[w0], %[c]\n\tv_sub_f32 %[w1], %[t1], %[ps1]\n\tv_mul_f32 %[w1], %[w1], %[c]\n\t" STEP_OPS);
    }
    if constexpr (KIND == 18) {
      float y0, y1, y2, y3, t0, t1;
      asm volatile(STEP(oa, ob, sa, sb, "", "", "") STEP(oa, ob, sa, sb, "", "", "") STEP(oa, ob, sa, sb, "", "", "") STEP(oa, ob, sa, sb, "", "", "") STEP_OPS);
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  asm volatile("" ::"v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(p0), "v"(p1), "v"(w0), "v"(w1), "v"(acc0), "v"(acc1), "v"(acc2), "v"(acc3));
  if constexpr (KIND >= 16) asm volatile("" ::"v"(r0), "v"(r1));
  if constexpr (KIND == 16) asm volatile("" ::"v"(r2), "v"(r3));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
void run(const char* name, int per_block, long long* d, double cyc_ghz_ratio) {
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(threads), 0, 0, d, 1.0f, (const void*)(d + 65536));
    hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(threads), 0, 0, d, 1.0f, (const void*)(d + 65536));
    hipDeviceSynchronize();
    std::vector<long long> h(256 * threads / 64);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double ticks = (double)h[h.size() / 2] / ((double)ITER * per_block);
    printf("%-52s %d wave(s)/SIMD: %7.2f cycles per %s\n", name, threads / 256, ticks, per_block == 32 && name[0] != 'a' ? "instruction" : "MFMA");
    (void)cyc_ghz_ratio;
  }
}

int main() {
  long long* d;
  hipMalloc(&d, 4 << 20);
  // __builtin_readcyclecounter counts shader cycles here: 32 independent v_mfma_f32_32x32x16_bf16 read 32.00 per instruction (8 passes x 4)
  int khz = 0;
  hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
  int wall = 0;
  hipDeviceGetAttribute(&wall, hipDeviceAttributeWallClockRate, 0);
  printf("device clock rate %d kHz, wall clock rate %d kHz\n", khz, wall);
  const double ratio = wall > 0 ? (double)khz / wall : 1.0;      // shader cycles per counter tick at the nominal clock
  run<0>("v_fma_f32 (4 chains)", 32, d, ratio);
  run<1>("v_pk_fma_f32 op_sel broadcast (2 chains)", 32, d, ratio);
  run<2>("v_exp_f32 (4 chains)", 32, d, ratio);
  run<3>("v_exp_f32 / v_fma_f32 alternating", 32, d, ratio);
  run<4>("v_add_f32", 32, d, ratio);
  run<5>("v_pk_add_f32 (2 chains)", 32, d, ratio);
  run<6>("v_cvt_pk_bf16_f32", 32, d, ratio);
  run<7>("v_max3_f32", 32, d, ratio);
  run<8>("s_nop 0", 32, d, ratio);
  run<9>("v_mfma_f32_32x32x16_bf16 (4 accumulators)", 32, d, ratio);
  run<10>("MFMA + 2 v_fma", 32, d, ratio);
  run<11>("MFMA + 4 v_fma", 32, d, ratio);
  run<12>("MFMA + 6 v_fma", 32, d, ratio);
  run<13>("MFMA + 8 v_fma", 32, d, ratio);
  run<14>("MFMA + v_exp + 3 v_fma", 32, d, ratio);
  run<15>("MFMA + 2 (v_exp, v_fma)", 32, d, ratio);
  run<16>("ds_read_b128 (32 in flight, one wait)", 32, d, ratio);
  run<17>("attention step (4 MFMA, 14 VALU, 2 reads, wait)", 16, d, ratio);
  run<18>("attention step without the reads", 16, d, ratio);
  run<21>("attention steps 0..3 (2 MFMA, 14 VALU, 2 reads), /MFMA", 8, d, ratio);
  run<22>("attention top block (8 MFMA x 4 v_max3 + tail), /MFMA", 8, d, ratio);
  run<19>("attention step + 1 LDS-DMA piece", 16, d, ratio);
  run<20>("attention step + 2 LDS-DMA pieces", 16, d, ratio);
  return 0;
}
