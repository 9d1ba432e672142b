// Entry points that dispatch between kernel families + error plumbing.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include "kp_common.h"

int kp_conv_fwd_generic(const void*, const void*, const float*, void*, float*, int, int, int, int, int, int, int,
                        hipStream_t);
int kp_conv_wgrad_generic(const void*, const void*, float*, int, int, int, int, int, int, int, int, hipStream_t);
int kp_conv_fwd_mfma(const void*, const void*, const float*, void*, float*, int, int, int, int, int, int,
                     hipStream_t);   // bf16 only; returns KP_E_UNSUPPORTED when the shape is not eligible
int kp_conv_wgrad_mfma(const void*, const void*, float*, int, int, int, int, int, int, int, hipStream_t);
int kp_conv_dgrad_mfma(const void*, const void*, void*, int, int, int, int, int, hipStream_t);
int kp_conv_wgrad_patch(const void*, const void*, float*, int, int, int, int, int, int, hipStream_t);
long long kp_conv_wgrad_patch_partials(int, int, int, int, int, int);
int kp_conv_small_fwd(const void*, const void*, const float*, void*, float*, int, int, int, int, int, hipStream_t);
int kp_conv_small_wgrad(const void*, const void*, float*, int, int, int, int, int, int, hipStream_t);
int kp_conv_first_fwd(const void*, const void*, const float*, void*, float*, int, int, int, int, int, hipStream_t);
int kp_conv_first_wgrad(const void*, const void*, float*, int, int, int, int, int, int, hipStream_t);
long long kp_conv_first_wgrad_partials(int, int, int, int, int);

static thread_local char g_err[512] = "";
static thread_local long long g_last_rows = 0;
void kp_set_last_rows(long long r) { g_last_rows = r; }
long long* g_kp_trace = nullptr;   // kp_config_ptr(KP_CFG_TRACE, dev_ptr): s_memtime trace buffer (512 x i64) or NULL
int g_kp_wgrad_patch = 1;   // kp_config(KP_CFG_WGRAD_PATCH, v)
int g_kp_bn_fast = 1;       // kp_config(KP_CFG_BN_FAST, v)
int g_kp_debug = 0;   // kp_config(KP_CFG_DEBUG, mask): kernel ablation switches for measurements

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

// ---- in-library kernel timing (bench.py's roofline leg) -------------------------------------------
// HIP events are recorded on the launch stream right around the convolution kernels; kp_prof_collect()
// synchronises them after the timed region and sums time and algorithmic FLOPs per kernel family.
#include <vector>
struct ProfRec { int family; double flops, bytes; hipEvent_t e0, e1; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;

static hipEvent_t prof_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
}
struct ProfScope {
    bool on; ProfRec r; hipStream_t st;
    ProfScope(int family, double flops, hipStream_t s, double bytes = 0.0) : on(g_prof_on), st(s) {
        if (!on) return;
        r.family = family; r.flops = flops; r.bytes = bytes; r.e0 = prof_event(); r.e1 = prof_event();
        hipEventRecord(r.e0, st);
    }
    void cancel() {   // nothing was launched under this scope: give the events back
        if (!on) return;
        g_event_pool.push_back(r.e0); g_event_pool.push_back(r.e1);
        on = false;
    }
    void done(int family) {
        if (!on) return;
        r.family = family;
        hipEventRecord(r.e1, st);
        g_prof.push_back(r);
    }
};

static int use_mfma() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("KP_DISABLE_MFMA");
        v = (e && e[0] == '1') ? 0 : 1;
    }
    return v;
}

extern int g_kp_conv_patch;

extern "C" {

int kp_version(void) { return 101; }
long long kp_last_rows(void) { return g_last_rows; }
const char* kp_last_error(void) { return g_err; }

int kp_device_cus(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}

int kp_config(int key, int value) {
    if (key == KP_CFG_CONV_PATCH) { g_kp_conv_patch = value ? 1 : 0; return KP_OK; }
    if (key == KP_CFG_DEBUG) { g_kp_debug = value; return KP_OK; }
    if (key == KP_CFG_BN_FAST) { g_kp_bn_fast = value ? 1 : 0; return KP_OK; }
    if (key == KP_CFG_WGRAD_PATCH) { g_kp_wgrad_patch = value ? 1 : 0; return KP_OK; }
    kp_set_error("kp_config: unknown key %d", key);
    return KP_E_ARG;
}

int kp_config_ptr(int key, void* value) {
    if (key == KP_CFG_TRACE) { g_kp_trace = (long long*)value; return KP_OK; }
    kp_set_error("kp_config_ptr: unknown key %d", key);
    return KP_E_ARG;
}

int kp_prof_enable(int on) {
    for (auto& r : g_prof) { g_event_pool.push_back(r.e0); g_event_pool.push_back(r.e1); }
    g_prof.clear();
    g_prof_on = on != 0;
    return KP_OK;
}

int kp_prof_collect(int family, double* ms, double* flops, long long* launches) {
    double t = 0.0, f = 0.0; long long n = 0;
    for (auto& r : g_prof) {
        if (r.family != family) continue;
        if (hipEventSynchronize(r.e1) != hipSuccess) { kp_set_error("kp_prof_collect: event sync failed"); return KP_E_LAUNCH; }
        float dt = 0.f;
        hipEventElapsedTime(&dt, r.e0, r.e1);
        t += dt; f += r.flops; ++n;
    }
    if (ms) *ms = t;
    if (flops) *flops = f;
    if (launches) *launches = n;
    return KP_OK;
}

int kp_prof_bytes(int family, double* bytes) {
    double b = 0.0;
    for (auto& r : g_prof)
        if (r.family == family) b += r.bytes;
    if (bytes) *bytes = b;
    return KP_OK;
}

int kp_conv_fwd(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi, int C,
                int K, int R, int dtype, void* stream) {
    KP_REQUIRE(x && w && y, "kp_conv_fwd: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && K > 0 && (R == 1 || R == 3) && Hi >= R && Wi >= R, "kp_conv_fwd: bad dims N=%d Hi=%d Wi=%d C=%d K=%d R=%d", N, Hi, Wi, C, K, R);
    KP_REQUIRE(dtype == KP_F32 || dtype == KP_BF16, "kp_conv_fwd: bad dtype %d", dtype);
    const double flops = 2.0 * N * (double)(Hi - R + 1) * (Wi - R + 1) * C * K * R * R;
    const double esz = dtype == KP_BF16 ? 2.0 : 4.0;   // algorithmic HBM bytes: input + output + weights, each once
    const double bytes = esz * ((double)N * Hi * Wi * C + (double)N * (Hi - R + 1) * (Wi - R + 1) * K + (double)R * R * C * K);
    ProfScope ps(KP_PROF_CONV_GENERIC, flops, (hipStream_t)stream, bytes);
    if (stats) kp_set_last_rows(((long long)N * (Hi - R + 1) * (Wi - R + 1) + 63) / 64);
    if (dtype == KP_BF16 && use_mfma()) {
        int rc = kp_conv_fwd_mfma(x, w, bias, y, stats, N, Hi, Wi, C, K, R, (hipStream_t)stream);
        if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_CONV_MFMA); return rc; }
        if (R == 3 && C <= 4) {
            rc = kp_conv_first_fwd(x, w, bias, y, stats, N, Hi, Wi, C, K, (hipStream_t)stream);
            if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_CONV_FIRST); return rc; }
        }
        if (R == 3) {
            rc = kp_conv_small_fwd(x, w, bias, y, stats, N, Hi, Wi, C, K, (hipStream_t)stream);
            if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_CONV_SMALL); return rc; }
        }
    }
    if (getenv("KP_LOG_GENERIC")) fprintf(stderr, "generic conv: N=%d Hi=%d Wi=%d C=%d K=%d R=%d dtype=%d stats=%d\n", N, Hi, Wi, C, K, R, dtype, stats != nullptr);
    int rc = kp_conv_fwd_generic(x, w, bias, y, stats, N, Hi, Wi, C, K, R, dtype, (hipStream_t)stream);
    ps.done(KP_PROF_CONV_GENERIC);
    return rc;
}

int kp_conv_dgrad(const void* dy, const void* w_dgrad, void* gx, int N, int H, int W, int K, int C, int dtype,
                  void* stream) {
    KP_REQUIRE(dy && w_dgrad && gx, "kp_conv_dgrad: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && K > 0 && H > 0 && W > 0, "kp_conv_dgrad: bad dims");
    KP_REQUIRE(dtype == KP_F32 || dtype == KP_BF16, "kp_conv_dgrad: bad dtype %d", dtype);
    if (dtype == KP_BF16 && use_mfma()) {
        const double flops = 2.0 * N * (double)H * W * C * K * 9;
        const double bytes = 2.0 * ((double)N * (H + 4) * (W + 4) * K + (double)N * (H + 2) * (W + 2) * C + 9.0 * C * K);
        ProfScope ps(KP_PROF_CONV_MFMA, flops, (hipStream_t)stream, bytes);
        int rc = kp_conv_dgrad_mfma(dy, w_dgrad, gx, N, H, W, K, C, (hipStream_t)stream);
        if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_CONV_MFMA); return rc; }
        ps.cancel();
    }
    return kp_conv_fwd(dy, w_dgrad, nullptr, gx, nullptr, N, H + 4, W + 4, K, C, 3, dtype, stream);
}

int kp_conv_wgrad_partials(int N, int Ho, int Wo, int C, int K, int R, int dy_pad, int dtype) {
    if (dtype == KP_BF16 && use_mfma() && R == 3 && g_kp_wgrad_patch) {
        const long long p = kp_conv_wgrad_patch_partials(N, Ho, Wo, C, K, dy_pad);
        if (p > 0 && p < 0x7fffffffLL) return (int)p;
    }
    if (dtype == KP_BF16 && use_mfma() && R == 3 && C <= 4) {
        const long long p = kp_conv_first_wgrad_partials(N, Ho, Wo, C, K);
        if (p > 0) return (int)p;
    }
    return 1;
}

int kp_conv_wgrad(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int R, int dy_pad,
                  int dtype, void* stream) {
    KP_REQUIRE(x && dy && scratch, "kp_conv_wgrad: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && K > 0 && (R == 1 || R == 3) && Ho > 0 && Wo > 0 && dy_pad >= 0, "kp_conv_wgrad: bad dims");
    KP_REQUIRE(dtype == KP_F32 || dtype == KP_BF16, "kp_conv_wgrad: bad dtype %d", dtype);
    const double flops = 2.0 * N * (double)Ho * Wo * C * K * R * R;
    const double wesz = dtype == KP_BF16 ? 2.0 : 4.0;
    const double wbytes = wesz * ((double)N * (Ho + R - 1) * (Wo + R - 1) * C + (double)N * Ho * Wo * K) + 4.0 * R * R * C * K;
    ProfScope ps(KP_PROF_WGRAD_GENERIC, flops, (hipStream_t)stream, wbytes);
    if (dtype == KP_BF16 && use_mfma()) {
        int rc = (R == 3 && g_kp_wgrad_patch) ? kp_conv_wgrad_patch(x, dy, scratch, N, Ho, Wo, C, K, dy_pad, (hipStream_t)stream)
                                              : KP_E_UNSUPPORTED;
        if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_WGRAD_MFMA); return rc; }
        rc = kp_conv_wgrad_mfma(x, dy, scratch, N, Ho, Wo, C, K, R, dy_pad, (hipStream_t)stream);
        if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_WGRAD_MFMA); return rc; }
        if (R == 3 && C <= 4) {
            rc = kp_conv_first_wgrad(x, dy, scratch, N, Ho, Wo, C, K, dy_pad, (hipStream_t)stream);
            if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_CONV_FIRST); return rc; }
        }
        if (R == 3) {
            rc = kp_conv_small_wgrad(x, dy, scratch, N, Ho, Wo, C, K, dy_pad, (hipStream_t)stream);
            if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_CONV_SMALL); return rc; }
        }
    }
    if (getenv("KP_LOG_GENERIC")) fprintf(stderr, "generic wgrad: N=%d Ho=%d Wo=%d C=%d K=%d R=%d dp=%d dtype=%d\n", N, Ho, Wo, C, K, R, dy_pad, dtype);
    int rc = kp_conv_wgrad_generic(x, dy, scratch, N, Ho, Wo, C, K, R, dy_pad, dtype, (hipStream_t)stream);
    ps.done(KP_PROF_WGRAD_GENERIC);
    return rc;
}

}  // extern "C"
