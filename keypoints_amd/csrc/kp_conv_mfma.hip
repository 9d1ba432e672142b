// bf16 MFMA implicit-GEMM convolution kernels (gfx950).  Placeholder until the tiled kernels land:
// every shape is reported unsupported so kp_conv_* falls through to the generic kernels.
#include "kp_common.h"

int kp_conv_fwd_mfma(const void*, const void*, const float*, void*, double*, int, int, int, int, int, int,
                     hipStream_t) {
    return KP_E_UNSUPPORTED;
}
int kp_conv_wgrad_mfma(const void*, const void*, float*, int, int, int, int, int, int, int, hipStream_t) {
    return KP_E_UNSUPPORTED;
}
