#include "sacv_common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void sacv_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

SACV_API const char* sacv_last_error(void) { return g_err; }
SACV_API int sacv_abi_version(void) { return 5; }

// One plan per PROCESS (one process drives one GPU): PyTorch runs a backward pass on its autograd thread, not on the
// thread that ran the forward pass, and a per-thread plan sent the two passes of one step to different kernels.
SacvShare sacv_share = {3, 2, 3, 2, 0, 3, 1, 0};

// Launch plan of the calling host thread's subsequent launches (see SacvShare in sacv_common.h); values <= 0 keep the
// current setting (cu_reserve, split_policy, wino_big, nt_split: < 0).  Defaults (a launch alone on the chip): 3, 2, 3, 2, 0, 3, 1, 0.
SACV_API int sacv_launch_share(int nt64_wg_per_cu, int nt128_wg_per_cu, int wino_wg_per_cu, int wgrad_wg_per_cu,
                               int cu_reserve, int split_policy, int wino_big, int nt_split) {
  if (nt64_wg_per_cu > 3 || nt128_wg_per_cu > 2 || wino_wg_per_cu > 3 || wgrad_wg_per_cu > 2 || cu_reserve > 192) {
    sacv_set_error("launch_share: out of range (%d %d %d %d %d)", nt64_wg_per_cu, nt128_wg_per_cu, wino_wg_per_cu,
                   wgrad_wg_per_cu, cu_reserve);
    return SACV_ERR_ARG;
  }
  if (nt64_wg_per_cu > 0) sacv_share.nt64_wg_per_cu = nt64_wg_per_cu;
  if (nt128_wg_per_cu > 0) sacv_share.nt128_wg_per_cu = nt128_wg_per_cu;
  if (wino_wg_per_cu > 0) sacv_share.wino_wg_per_cu = wino_wg_per_cu;
  if (wgrad_wg_per_cu > 0) sacv_share.wgrad_wg_per_cu = wgrad_wg_per_cu;
  if (cu_reserve >= 0) sacv_share.cu_reserve = cu_reserve;
  if (split_policy >= 0) sacv_share.split_policy = split_policy;
  if (wino_big >= 0) sacv_share.wino_big = wino_big > 2 ? 2 : wino_big;
  if (nt_split >= 0) sacv_share.nt_split = nt_split ? 1 : 0;
  return SACV_OK;
}
