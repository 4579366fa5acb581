// capi.hip — version / error plumbing of the C ABI (include/domainrag_hip.h).
#include "drag_common.h"
#include <stdlib.h>
#include <string.h>

static thread_local char g_err[512] = "";

void drag_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "unknown error", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* drag_last_error(void) { return g_err; }
extern "C" int drag_version(void) { return 100; }  // 0.1.0
extern "C" int drag_experiments_built(void) { return DRAG_EXP; }

// ---- tuning switches (measurement only: every setting computes the same values unless its comment says otherwise).  Initial
// values come from the environment ONCE; drag_set_option changes them at run time so one process can A/B kernels.
static int g_opt[DRAG_OPT_COUNT];
static bool g_opt_init = false;
static const char* const g_opt_names[DRAG_OPT_COUNT] = {"attn_sched", "attn_w4", "attn_tune", "attn_q64", "gemm_kernel", "ln_generic", "gemm_group_m", "topk_grid", "topk_depth", "attn_persist", "topk_select", "topk_dense_sample", "topk_qt", "gemm_pair", "gemm_epilogue", "topk_path", "topk_qreg", "gemm_w4", "attn_walk", "gemm_splitk", "attn_gen"};
static const char* const g_opt_env[DRAG_OPT_COUNT] = {"DRAG_ATTN_SCHED", "DRAG_ATTN_W4", "DRAG_ATTN_TUNE", "DRAG_ATTN_Q64", "DRAG_GEMM_KERNEL", "DRAG_LN_GENERIC", "DRAG_GEMM_GROUP_M", "DRAG_TOPK_GRID", "DRAG_TOPK_DEPTH", "DRAG_ATTN_PERSIST", "DRAG_TOPK_SELECT", "DRAG_TOPK_DENSE_SAMPLE", "DRAG_TOPK_QT", "DRAG_GEMM_PAIR", "DRAG_GEMM_EPILOGUE", "DRAG_TOPK_PATH", "DRAG_TOPK_QREG", "DRAG_GEMM_W4", "DRAG_ATTN_WALK", "DRAG_GEMM_SPLITK", "DRAG_ATTN_GEN"};

static void opt_init() {
  if (g_opt_init) return;
  for (int i = 0; i < DRAG_OPT_COUNT; ++i) {
    const char* e = getenv(g_opt_env[i]);
    g_opt[i] = e ? (*e ? atoi(e) : 1) : 0;
  }
  g_opt[DRAG_OPT_ATTN_SCHED] = getenv(g_opt_env[DRAG_OPT_ATTN_SCHED]) ? g_opt[DRAG_OPT_ATTN_SCHED] : DRAG_ATTN_SCHED_DEFAULT;
  g_opt[DRAG_OPT_ATTN_TUNE] = getenv(g_opt_env[DRAG_OPT_ATTN_TUNE]) ? g_opt[DRAG_OPT_ATTN_TUNE] : DRAG_ATTN_TUNE_DEFAULT;
  if (!DRAG_EXP) {        // the product library has no experiment kernels: their environment switches are ignored
    g_opt[DRAG_OPT_ATTN_PERSIST] = g_opt[DRAG_OPT_TOPK_QT] = 0;
    if (g_opt[DRAG_OPT_ATTN_SCHED] == 3) g_opt[DRAG_OPT_ATTN_SCHED] = DRAG_ATTN_SCHED_DEFAULT;
  }
  g_opt_init = true;
}

int drag_opt(int idx) {
  opt_init();
  return g_opt[idx];
}

extern "C" int drag_set_option(const char* name, int32_t value) {
  DRAG_CHECK(name != nullptr, "drag_set_option: null name");
  opt_init();
  for (int i = 0; i < DRAG_OPT_COUNT; ++i)
    if (strcmp(name, g_opt_names[i]) == 0) {
      const bool experiment = i == DRAG_OPT_ATTN_PERSIST || i == DRAG_OPT_TOPK_QT || (i == DRAG_OPT_ATTN_SCHED && value == 3);
      DRAG_CHECK(DRAG_EXP || !experiment || (value == 0 && i != DRAG_OPT_ATTN_SCHED),
                 "drag_set_option: attn_persist, topk_qt and attn_sched = 3 are experiments (measured non-improvements): build the library with DRAG_EXPERIMENTS=1");
      g_opt[i] = value;
      return 0;
    }
  drag_set_error("drag_set_option: unknown option (attn_sched, attn_w4, attn_tune, attn_q64, gemm_kernel, ln_generic, gemm_group_m, topk_grid, topk_depth, attn_persist, topk_select, topk_dense_sample, topk_qt, gemm_pair, gemm_epilogue, topk_path, topk_qreg, gemm_w4, attn_walk, gemm_splitk, attn_gen)");
  return -1;
}
