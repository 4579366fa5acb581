// capi.hip — version / error plumbing of the C ABI (include/domainrag_hip.h).
#include "drag_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void drag_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "unknown error", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

extern "C" const char* drag_last_error(void) { return g_err; }
extern "C" int drag_version(void) { return 100; }  // 0.1.0
