// Entry points that dispatch between kernel families + error plumbing.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include "kp_common.h"

int kp_conv_fwd_generic(const void*, const void*, const float*, void*, double*, int, int, int, int, int, int, int,
                        hipStream_t);
int kp_conv_wgrad_generic(const void*, const void*, float*, int, int, int, int, int, int, int, int, hipStream_t);
int kp_conv_fwd_mfma(const void*, const void*, const float*, void*, double*, int, int, int, int, int, int,
                     hipStream_t);   // bf16 only; returns KP_E_UNSUPPORTED when the shape is not eligible
int kp_conv_wgrad_mfma(const void*, const void*, float*, int, int, int, int, int, int, int, hipStream_t);

static thread_local char g_err[512] = "";

void kp_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int kp_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        kp_set_error("%s: %s", what, hipGetErrorString(e));
        return KP_E_LAUNCH;
    }
    return KP_OK;
}

static int use_mfma() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("KP_DISABLE_MFMA");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v;
}

extern "C" {

int kp_version(void) { return 100; }
const char* kp_last_error(void) { return g_err; }

int kp_device_cus(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}

int kp_conv_fwd(const void* x, const void* w, const float* bias, void* y, double* stats, int N, int Hi, int Wi, int C,
                int K, int R, int dtype, void* stream) {
    KP_REQUIRE(x && w && y, "kp_conv_fwd: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && K > 0 && (R == 1 || R == 3) && Hi >= R && Wi >= R, "kp_conv_fwd: bad dims N=%d Hi=%d Wi=%d C=%d K=%d R=%d", N, Hi, Wi, C, K, R);
    KP_REQUIRE(dtype == KP_F32 || dtype == KP_BF16, "kp_conv_fwd: bad dtype %d", dtype);
    if (dtype == KP_BF16 && use_mfma()) {
        int rc = kp_conv_fwd_mfma(x, w, bias, y, stats, N, Hi, Wi, C, K, R, (hipStream_t)stream);
        if (rc != KP_E_UNSUPPORTED) return rc;
    }
    return kp_conv_fwd_generic(x, w, bias, y, stats, N, Hi, Wi, C, K, R, dtype, (hipStream_t)stream);
}

int kp_conv_wgrad(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int R, int dy_pad,
                  int dtype, void* stream) {
    KP_REQUIRE(x && dy && scratch, "kp_conv_wgrad: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && K > 0 && (R == 1 || R == 3) && Ho > 0 && Wo > 0 && dy_pad >= 0, "kp_conv_wgrad: bad dims");
    KP_REQUIRE(dtype == KP_F32 || dtype == KP_BF16, "kp_conv_wgrad: bad dtype %d", dtype);
    if (dtype == KP_BF16 && use_mfma()) {
        int rc = kp_conv_wgrad_mfma(x, dy, scratch, N, Ho, Wo, C, K, R, dy_pad, (hipStream_t)stream);
        if (rc != KP_E_UNSUPPORTED) return rc;
    }
    return kp_conv_wgrad_generic(x, dy, scratch, N, Ho, Wo, C, K, R, dy_pad, dtype, (hipStream_t)stream);
}

}  // extern "C"
