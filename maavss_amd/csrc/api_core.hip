// Error state + version entry points of the C-ABI (include/maavss.h).
#include "common.h"
#include "../../include/maavss.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void maavss_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// Deterministic mode (process-wide, default off): the M = batch Linear forms and the generic GEMM's automatic split-K accumulate their
// K slices with f32 atomicAdd -- results reproducible to summation order only.  With the flag set, maavss_gemm_f32 takes no path that
// uses atomics (one K slice per output element; explicit split_k > 1 requests are still honoured): bit-identical run to run, slower
// weight streaming on the few Linear shapes concerned.
static int g_deterministic = 0;
static float* g_det_ws = nullptr;       // caller-provided device scratch for the deterministic split-K partial sums
static int64_t g_det_ws_floats = 0;
static void* g_det_ws_stream = nullptr; // the one stream whose Linear kernels may use the scratch
int maavss_deterministic_flag(void) { return g_deterministic; }
float* maavss_deterministic_ws(int64_t* floats, void* stream) {
  // a launch on another stream than the registered one gets no scratch (-> single-slice path): two streams must never share it
  const bool ok = g_det_ws && stream == g_det_ws_stream;
  *floats = ok ? g_det_ws_floats : 0;
  return ok ? g_det_ws : nullptr;
}

extern "C" {
int maavss_set_deterministic(int on) { const int prev = g_deterministic; g_deterministic = on ? 1 : 0; return prev; }   // returns the previous setting
int maavss_get_deterministic(void) { return g_deterministic; }
// Optional device scratch for deterministic mode (the library never allocates): with it the M = batch Linear forms keep their
// split over K -- every slice writes its partial sums, a second kernel adds them in slice order -- instead of falling back to one
// slice per output element.  One scratch per process, bound to ONE stream: Linear kernels launched on any other stream do not see it
// and take the single-slice path (slower, still deterministic), so two streams can never race on it.
int maavss_set_deterministic_workspace(float* ws, int64_t bytes, void* stream) {
  if (ws != nullptr && (bytes < 0 || ((uintptr_t)ws & 15) != 0)) { maavss_set_error("set_deterministic_workspace: ws must be 16-byte aligned"); return MAAVSS_ERR_ARG; }
  g_det_ws = ws;
  g_det_ws_floats = ws ? bytes / 4 : 0;
  g_det_ws_stream = ws ? stream : nullptr;
  return MAAVSS_OK;
}
const char* maavss_last_error(void) { return g_err; }
int maavss_version(void) { return MAAVSS_ABI_VERSION; }
const char* maavss_arch(void) { return "gfx950"; }
}
