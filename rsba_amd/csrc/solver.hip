// LM / Schur solver state (placeholder until the normal-equation kernels land).
#include "handle.hpp"

int32_t rsba_gradient(rsba_handle*, double*) { return rsba_set_error(RSBA_ERR_UNSUPPORTED, "gradient: not built yet"); }
void rsba_destroy_solver(rsba_handle*) {}

extern "C" int32_t rsba_solve(rsba_handle*, const rsba_solver_options*, rsba_solver_summary*, rsba_iteration*, int32_t) {
  return rsba_set_error(RSBA_ERR_UNSUPPORTED, "solve: not built yet");
}
