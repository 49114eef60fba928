// Error reporting + version for the C-ABI (include/ssunet_hip.h).
#include "common.h"

static thread_local char g_err[512] = "";

void ssg_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* ssg_last_error(void) { return g_err; }
extern "C" int ssg_abi_version(void) { return 9; }   // 9: ssg_conv_desc.bwd_x ... bwd_slope (batch-norm backward statistics in the input-gradient epilogue), ssg_conv2d_bwd_stats_ok, ssg_conv2d_thin_bf16*; 8: ssg_conv_desc / ssg_wgrad_desc .in_scale / in_shift / in_act / in_slope (fused batch-norm apply on the input), ssg_conv2d_in_affine_ok, ssg_conv2d_wgrad_in_affine_ok, split-pack codes 1016 / 1032; 7: split-pack format codes 1128 / 1064 (conv_igemm_halo_k32.hip), ssg_conv_set_k32_mode; 6: ssg_conv_desc.parity_merge (appended after w_split); 5: ssg_wgrad_desc.flags; 4: ssg_conv_desc.w_split (operands split into bf16 terms); 3: ws / ws_bytes (split-K)
