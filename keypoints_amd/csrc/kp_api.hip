// Entry points that dispatch between kernel families + error plumbing.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include "kp_common.h"

int kp_bottleneck_fused_fwd(const float*, const float*, const float*, const float*, float*, float*, float*, float*, float*,
                            float*, float*, unsigned char*, int, int, int, int, int, float, float, hipStream_t);
int kp_bottleneck_fused_bwd(const float*, const float*, const float*, const float*, const float*, const float*, const float*,
                            const unsigned char*, const float*, const float*, const float*, float*, float*, int, int, int,
                            int, int, float, float, hipStream_t);

static thread_local char g_err[512] = "";
static thread_local long long g_last_rows = 0;
void kp_set_last_rows(long long r) { g_last_rows = r; }
static thread_local int g_query = 0;
int kp_query_mode() { return g_query; }
void kp_set_query_mode(int on) { g_query = on; }
static thread_local KpFinRequest* g_fin = nullptr;
KpFinRequest* kp_fin_request() { return g_fin; }
void kp_set_fin_request(KpFinRequest* r) { g_fin = r; }
kp_knob_ptr_t g_kp_trace{nullptr};   // kp_config_ptr(KP_CFG_TRACE, dev_ptr): s_memtime trace buffer (512 x i64) or NULL
kp_knob_t g_kp_wgrad_patch{1};   // kp_config(KP_CFG_WGRAD_PATCH, v)
kp_knob_t g_kp_small_band{1};    // kp_config(KP_CFG_SMALL_BAND, v)
kp_knob_t g_kp_bn_fast{1};       // kp_config(KP_CFG_BN_FAST, v)
kp_knob_t g_kp_bn_flat{1};       // kp_config(KP_CFG_BN_FLAT, v)
kp_knob_t g_kp_bred_patch{1};    // kp_config(KP_CFG_BRED_PATCH, v)
kp_knob_t g_kp_fin_nt{256};     // kp_config(KP_CFG_FIN_NT, 256 | 1024): threads per workgroup of the one-launch statistics / backward-sums finalize of the > 32-channel layers
kp_knob_t g_kp_fin_wide{0};      // kp_config(KP_CFG_FIN_WIDE, 1): the 1024-thread statistics finalize for every layer (measurement)
kp_knob_t g_kp_disable_mfma{0};  // kp_config(KP_CFG_DISABLE_MFMA, 1): every convolution on the generic VALU kernels (measurement / debugging)
kp_knob_t g_kp_wg16{7};          // kp_config(KP_CFG_WG16, bits): which small-channel weight gradients take the 16 x 16 x 32 kernel
kp_knob_t g_kp_first_rows{1};    // kp_config(KP_CFG_FIRST_ROWS, v): row kernel for single-channel first layers (1 default, 0 off, > 1: workgroups per launch)
kp_knob_t g_kp_first_ktn{2};     // kp_config(KP_CFG_FIRST_KTN, v): 32-channel blocks per wavefront in kp_conv_first_fwd_bnact (2 default, 1)
// Memory walk order of the passes (kp_common.h: KP_WALK_*; kp_config(KP_CFG_WALK, mask), 0 = every pass front to back):
// a pass that follows a producer sweeps the producer's output back to front, so that it starts on what is still in the
// 256 MiB Infinity Cache, and reads what nobody re-reads soon with streaming loads
kp_knob_t g_kp_walk{KP_WALK_DEFAULT};   // kp_config(KP_CFG_WALK, mask)
kp_knob_t g_kp_cus{0};     // kp_config(KP_CFG_CUS, n): CU budget that persistent kernels size their grids to (0 = the whole device)
kp_knob_t g_kp_debug{0};   // kp_config(KP_CFG_DEBUG, mask): kernel ablation switches for measurements

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

// ---- in-library kernel timing: storage (kp_prof.h holds the scope object the launch sites use) -----------------
#include "kp_prof.h"
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::mutex g_prof_mu;

static std::atomic<int> g_bytes_on{0};
static double g_bytes_total = 0.0;
static long long g_bytes_launches = 0;

void kp_count_bytes(double bytes, int launch) {
    if (!g_bytes_on.load(std::memory_order_relaxed) || kp_query_mode()) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_bytes_total += bytes;
    g_bytes_launches += launch;
}

static std::vector<hipEvent_t> g_event_pool;
hipEvent_t kp_prof_event() {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
void kp_prof_event_release(hipEvent_t e) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_event_pool.push_back(e);
}

kp_knob_t g_kp_stats_store{0};   // kp_config(KP_CFG_STATS_STORE, 1): measurement -- forward statistics of the 128-channel-tile patch kernel at the store stage (kp_conv_patch.hip, SST)
kp_knob_t g_kp_fin_fold{1};      // kp_config(KP_CFG_FIN_FOLD, v): 1 (default) kp_conv_fwd_fin folds the finalize into the convolution's tail where it can; 0: always the separate launch
kp_knob_t g_kp_conv_patch{1};    // kp_config(KP_CFG_CONV_PATCH, v): 1 (default) the LDS-patch kernel, 0 linear, 2 pair tile first

extern "C" {

int kp_version(void) { return 101; }
long long kp_last_rows(void) { return g_last_rows; }
const char* kp_last_error(void) { return g_err; }

int kp_device_cus(void) {
    if (g_kp_cus > 0) return g_kp_cus;
    static int cached = 0;
    if (cached > 0) return cached;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    cached = n;
    return n;
}

int kp_config(int key, int value) {
    if (key == KP_CFG_CONV_PATCH) { g_kp_conv_patch = value < 0 ? 0 : (value > 2 ? 2 : value); return KP_OK; }
    if (key == KP_CFG_DEBUG) { g_kp_debug = value; return KP_OK; }
    if (key == KP_CFG_STATS_STORE) { g_kp_stats_store = value ? 1 : 0; return KP_OK; }
    if (key == KP_CFG_FIN_FOLD) { g_kp_fin_fold = value ? 1 : 0; return KP_OK; }
    if (key == KP_CFG_FIN_NT) { g_kp_fin_nt = value == 256 ? 256 : 1024; return KP_OK; }
    if (key == KP_CFG_WALK) { g_kp_walk = value; return KP_OK; }
    if (key == KP_CFG_CUS) { g_kp_cus = value > 0 ? value : 0; return KP_OK; }
    if (key == KP_CFG_BN_FAST) { g_kp_bn_fast = value ? 1 : 0; return KP_OK; }
    if (key == KP_CFG_BN_FLAT) { g_kp_bn_flat = value ? 1 : 0; return KP_OK; }
    if (key == KP_CFG_FIN_WIDE) { g_kp_fin_wide = value ? 1 : 0; return KP_OK; }
    if (key == KP_CFG_DISABLE_MFMA) { g_kp_disable_mfma = value ? 1 : 0; return KP_OK; }
    if (key == KP_CFG_WG16) { g_kp_wg16 = value & 7; return KP_OK; }
    if (key == KP_CFG_FIRST_ROWS) { g_kp_first_rows = value < 0 ? 0 : value; return KP_OK; }
    if (key == KP_CFG_FIRST_KTN) { g_kp_first_ktn = value == 1 ? 1 : 2; return KP_OK; }
    if (key == KP_CFG_BRED_PATCH) { g_kp_bred_patch = value < 0 ? 0 : (value > 2 ? 2 : value); return KP_OK; }
    if (key == KP_CFG_WGRAD_PATCH) { g_kp_wgrad_patch = value ? 1 : 0; return KP_OK; }
    if (key == KP_CFG_SMALL_BAND) { g_kp_small_band = value ? 1 : 0; return KP_OK; }
    kp_set_error("kp_config: unknown key %d", key);
    return KP_E_ARG;
}

// The raw value of an integer knob (what the last kp_config(key, .) stored; the library's default if never set), so that a caller
// that changes a process-global knob for its own lifetime can put back what it found.  Negative: KP_E_ARG (unknown key).
int kp_config_get(int key) {
    if (key == KP_CFG_CUS) return g_kp_cus;
    if (key == KP_CFG_CONV_PATCH) return g_kp_conv_patch;
    if (key == KP_CFG_DEBUG) return g_kp_debug;
    if (key == KP_CFG_WALK) return g_kp_walk;
    if (key == KP_CFG_BN_FAST) return g_kp_bn_fast;
    if (key == KP_CFG_BRED_PATCH) return g_kp_bred_patch;
    if (key == KP_CFG_FIRST_ROWS) return g_kp_first_rows;
    kp_set_error("kp_config_get: unknown key %d", key);
    return KP_E_ARG;
}


int kp_config_ptr(int key, void* value) {
    if (key == KP_CFG_TRACE) { g_kp_trace = (long long*)value; return KP_OK; }
    kp_set_error("kp_config_ptr: unknown key %d", key);
    return KP_E_ARG;
}

int kp_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof) { g_event_pool.push_back(r.e0); g_event_pool.push_back(r.e1); }
    g_prof.clear();
    g_prof_on = on == 1;
    g_bytes_on.store(on == 2 ? 1 : 0);
    if (on == 2) { g_bytes_total = 0.0; g_bytes_launches = 0; }
    return KP_OK;
}

int kp_prof_step_bytes(double* bytes, long long* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (bytes) *bytes = g_bytes_total;
    if (launches) *launches = g_bytes_launches;
    return KP_OK;
}

int kp_prof_collect(int family, double* ms, double* flops, long long* launches) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double t = 0.0, f = 0.0; long long n = 0;
    for (auto& r : g_prof) {
        if (r.family != family) continue;
        if (hipEventSynchronize(r.e1) != hipSuccess) { kp_set_error("kp_prof_collect: event sync failed"); return KP_E_LAUNCH; }
        float dt = 0.f;
        (void)hipEventElapsedTime(&dt, r.e0, r.e1);
        t += dt; f += r.flops; ++n;
    }
    if (ms) *ms = t;
    if (flops) *flops = f;
    if (launches) *launches = n;
    return KP_OK;
}

int kp_prof_bytes(int family, double* bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double b = 0.0;
    for (auto& r : g_prof)
        if (r.family == family) b += r.bytes;
    if (bytes) *bytes = b;
    return KP_OK;
}

int kp_keypoint_bottleneck_fwd(const float* heat_s, const float* heat_t, const float* phi_s, const float* phi_t,
                               float* k_t, float* p_t, float* q_t, float* m_t, float* phi, float* mask_s, float* mask_t,
                               unsigned char* arg_t, int B, int C, int K, int H, int W, float sigma, float eps,
                               void* stream) {
    // algorithmic HBM bytes: 2 heat stacks + 2 feature stacks in; blended features, target maps, 2 masks, argmax,
    // keypoints and marginals out -- each once
    const double hw = (double)H * W;
    const double bytes = (double)B * (4.0 * (2.0 * K * hw + 2.0 * C * hw + C * hw + K * hw + 2.0 * hw) + hw + 4.0 * K * (2 + H + W));
    ProfScope ps(KP_PROF_BOTTLENECK, 0.0, (hipStream_t)stream, bytes);
    const int rc = kp_bottleneck_fused_fwd(heat_s, heat_t, phi_s, phi_t, k_t, p_t, q_t, m_t, phi, mask_s, mask_t, arg_t, B, C,
                                           K, H, W, sigma, eps, (hipStream_t)stream);
    ps.done(KP_PROF_BOTTLENECK);
    return rc;
}

int kp_keypoint_bottleneck_bwd(const float* gphi, const float* g_k, const float* g_m, const float* phi_s,
                               const float* phi_t, const float* mask_s, const float* mask_t, const unsigned char* arg_t,
                               const float* k_t, const float* p_t, const float* q_t, float* d_phi_t, float* d_heat_t,
                               int B, int C, int K, int H, int W, float sigma, float eps, void* stream) {
    // in: gphi, phi_s, phi_t, masks, argmax, keypoints, marginals; out: d_phi_t, d_heat_t
    const double hw = (double)H * W;
    const double bytes = (double)B * (4.0 * (3.0 * C * hw + 2.0 * hw + C * hw + K * hw) + hw + 4.0 * K * (2 + H + W));
    ProfScope ps(KP_PROF_BOTTLENECK, 0.0, (hipStream_t)stream, bytes);
    const int rc = kp_bottleneck_fused_bwd(gphi, g_k, g_m, phi_s, phi_t, mask_s, mask_t, arg_t, k_t, p_t, q_t, d_phi_t,
                                           d_heat_t, B, C, K, H, W, sigma, eps, (hipStream_t)stream);
    ps.done(KP_PROF_BOTTLENECK);
    return rc;
}

// ---- dtype dispatch of the entry points whose kernels exist once per 16-bit storage type ---------------------------
// name__bf16 (also serves KP_F32) and name__f16 come from the two builds of the same source (kp_h16_names.h).
#define KP_DISPATCH(name, DECL, CALL)                                                        \
    int name##__bf16 DECL;                                                                   \
    int name##__f16 DECL;                                                                    \
    int name DECL {                                                                          \
        if (dtype != KP_F32 && dtype != KP_BF16 && dtype != KP_F16) {                        \
            kp_set_error(#name ": bad dtype %d", dtype);                                     \
            return KP_E_ARG;                                                                 \
        }                                                                                    \
        return dtype == KP_F16 ? name##__f16 CALL : name##__bf16 CALL;                       \
    }

KP_DISPATCH(kp_nchw_to_nhwc, (const float* src, void* dst, int N, int C, int Cp, int H, int W, int pad, int dtype, void* stream),
            (src, dst, N, C, Cp, H, W, pad, dtype, stream))
KP_DISPATCH(kp_nhwc_grad_to_nchw, (const void* gsrc, float* dst, int N, int C, int Cp, int H, int W, int pad, int dtype, void* stream),
            (gsrc, dst, N, C, Cp, H, W, pad, dtype, stream))
KP_DISPATCH(kp_nchw2_to_nhwc, (const float* src0, int C0, const float* src1, int C1, void* dst, int N, int Cp, int H, int W, int pad, int dtype, void* stream),
            (src0, C0, src1, C1, dst, N, Cp, H, W, pad, dtype, stream))
KP_DISPATCH(kp_nhwc_grad_to_nchw2, (const void* gsrc, float* dst0, int C0, float* dst1, int C1, int N, int Cp, int H, int W, int pad, int dtype, void* stream),
            (gsrc, dst0, C0, dst1, C1, N, Cp, H, W, pad, dtype, stream))
KP_DISPATCH(kp_conv_first_fused_supported, (int N, int H, int W, int C, int K, int dtype), (N, H, W, C, K, dtype))
KP_DISPATCH(kp_conv_first_stats, (const void* x, const void* w, const float* bias, float* stats, int N, int H, int W, int C, int K, int dtype, void* stream),
            (x, w, bias, stats, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_conv_first_fwd_bnact, (const void* x, const void* w, const float* bias, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, void* a_out, int N, int H, int W, int C, int K, int dtype, void* stream),
            (x, w, bias, mean, invstd, gamma, beta, slope, a_out, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_conv_first_bwd_reduce, (const void* g, const void* x, const void* w, const float* bias, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, float* partials, int N, int H, int W, int C, int K, int dtype, void* stream),
            (g, x, w, bias, mean, invstd, gamma, beta, slope, partials, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_conv_first_bwd_wgrad, (const void* g, const void* x, const void* w, const float* bias, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, const float* sums, float* scratch, int N, int H, int W, int C, int K, int dtype, void* stream),
            (g, x, w, bias, mean, invstd, gamma, beta, slope, sums, scratch, N, H, W, C, K, dtype, stream))
int kp_conv_wgrad_pad_hw__bf16(int, int, int, int, int, int, int, int*, int*);
int kp_conv_wgrad_pad_hw__f16(int, int, int, int, int, int, int, int*, int*);
int kp_conv_wgrad_pad_hw(int N, int Ho, int Wo, int C, int K, int dy_pad, int dtype, int* Hp, int* Wp) {
    return dtype == KP_F16 ? kp_conv_wgrad_pad_hw__f16(N, Ho, Wo, C, K, dy_pad, dtype, Hp, Wp) : kp_conv_wgrad_pad_hw__bf16(N, Ho, Wo, C, K, dy_pad, dtype, Hp, Wp);
}
KP_DISPATCH(kp_pack_weights, (const float* w, void* w_fwd, void* w_dgrad, int K, int C, int Cp, int R, int dtype, void* stream),
            (w, w_fwd, w_dgrad, K, C, Cp, R, dtype, stream))
KP_DISPATCH(kp_pack_weights_batch, (int n, const float* const* w, void* const* w_fwd, void* const* w_dgrad, const int* K, const int* C, const int* Cp, int dtype, void* stream),
            (n, w, w_fwd, w_dgrad, K, C, Cp, dtype, stream))
KP_DISPATCH(kp_conv_fwd, (const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi, int C, int K, int R, int dtype, void* stream),
            (x, w, bias, y, stats, N, Hi, Wi, C, K, R, dtype, stream))
// conv + BatchNorm finalize as ONE call; the finalize runs in the tail of the convolution kernel where the launcher takes the request
// (KpFinRequest, kp_common.h), else as the kp_bn_finalize launch behind it -- the same numbers either way up to the order of the fp64 sums
int kp_conv_fwd_fin(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi, int C, int K, int R,
                    int dtype, double* ws, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                    long long* num_batches_tracked, float momentum, float eps, long long count, unsigned* tickets, void* stream) {
    if (!stats || !ws || !save_mean || !save_invstd || count <= 0) { kp_set_error("kp_conv_fwd_fin: bad args"); return KP_E_ARG; }
    KpFinRequest rq{save_mean, save_invstd, running_mean, running_var, num_batches_tracked, momentum, eps, count, tickets, 0};
    kp_set_fin_request((tickets && g_kp_fin_fold) ? &rq : nullptr);
    const int rc = kp_conv_fwd(x, w, bias, y, stats, N, Hi, Wi, C, K, R, dtype, stream);
    kp_set_fin_request(nullptr);
    if (rc != KP_OK || rq.taken) return rc;
    return kp_bn_finalize(stats, kp_last_rows(), ws, save_mean, save_invstd, running_mean, running_var, num_batches_tracked, momentum, eps,
                          count, K, stream);
}
// 1 if kp_conv_fwd_fin would fold the finalize into the convolution for this shape (tests / records; a pure query)
int kp_conv_fwd_fin_folds(int N, int Hi, int Wi, int C, int K, int R, int dtype) {
    if (!g_kp_fin_fold) return 0;
    unsigned dummy_t = 0; float dummy;
    KpFinRequest rq{&dummy, &dummy, nullptr, nullptr, nullptr, 0.f, 0.f, 1, &dummy_t, 0};
    kp_set_query_mode(1);
    kp_set_fin_request(&rq);
    const int rc = kp_conv_fwd(nullptr, nullptr, nullptr, nullptr, &dummy, N, Hi, Wi, C, K, R, dtype, nullptr);
    kp_set_fin_request(nullptr);
    kp_set_query_mode(0);
    return (rc == KP_OK && rq.taken) ? 1 : 0;
}
KP_DISPATCH(kp_conv_dgrad, (const void* dy, const void* w_dgrad, void* gx, int N, int H, int W, int K, int C, int dtype, void* stream),
            (dy, w_dgrad, gx, N, H, W, K, C, dtype, stream))
KP_DISPATCH(kp_conv_wgrad_partials, (int N, int Ho, int Wo, int C, int K, int R, int dy_pad, int dtype),
            (N, Ho, Wo, C, K, R, dy_pad, dtype))
KP_DISPATCH(kp_conv_wgrad, (const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int R, int dy_pad, int dtype, void* stream),
            (x, dy, scratch, N, Ho, Wo, C, K, R, dy_pad, dtype, stream))
KP_DISPATCH(kp_conv_fwd_act_supported, (int N, int H, int W, int C, int K, int dtype), (N, H, W, C, K, dtype))
KP_DISPATCH(kp_conv_fwd_act, (const void* x, const void* w, const float* scale, const float* shift, float slope, void* out, int out_pad, int N, int H, int W, int C, int K, int dtype, void* stream),
            (x, w, scale, shift, slope, out, out_pad, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_conv_bnin_supported, (int N, int H, int W, int C, int K, int dtype), (N, H, W, C, K, dtype))
KP_DISPATCH(kp_conv_fwd_bnin, (const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, const void* w, const float* bias, void* y, float* stats, int N, int H, int W, int C, int K, int dtype, void* stream),
            (yprev, mean, invstd, gamma, beta, slope, w, bias, y, stats, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_conv_wgrad_bnin, (const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int dy_pad, int dtype, void* stream),
            (yprev, mean, invstd, gamma, beta, slope, dy, scratch, N, Ho, Wo, C, K, dy_pad, dtype, stream))
KP_DISPATCH(kp_conv_dgrad_bnred_supported, (int N, int H, int W, int K, int C, int dtype), (N, H, W, K, C, dtype))
KP_DISPATCH(kp_conv_dgrad_bnred, (const void* dy, const void* w_dgrad, void* gx, const void* yred, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, float* partials, int N, int H, int W, int K, int C, int dtype, void* stream),
            (dy, w_dgrad, gx, yred, mean, invstd, gamma, beta, slope, partials, N, H, W, K, C, dtype, stream))
KP_DISPATCH(kp_bn_act_fwd, (const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, int tail, void* out, int out_pad, unsigned char* argmax, int N, int H, int W, int C, int dtype, void* stream),
            (y, mean, invstd, gamma, beta, slope, tail, out, out_pad, argmax, N, H, W, C, dtype, stream))
KP_DISPATCH(kp_bn_act_bwd_reduce, (const void* gsrc, int gp, int tail, const unsigned char* argmax, const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, float* partials, int N, int H, int W, int C, int dtype, void* stream),
            (gsrc, gp, tail, argmax, y, mean, invstd, gamma, beta, slope, partials, N, H, W, C, dtype, stream))
KP_DISPATCH(kp_bn_act_bwd_apply, (const void* gsrc, int gp, int tail, const unsigned char* argmax, const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, const float* sums, void* dy, int dp, int N, int H, int W, int C, int dtype, void* stream),
            (gsrc, gp, tail, argmax, y, mean, invstd, gamma, beta, slope, sums, dy, dp, N, H, W, C, dtype, stream))
KP_DISPATCH(kp_upsample2_bwd, (const void* gsrc, int gp, void* g, int N, int H, int W, int C, int dtype, void* stream),
            (gsrc, gp, g, N, H, W, C, dtype, stream))
KP_DISPATCH(kp_head_fwd, (const void* x, const float* w, const float* bias, float slope, float* out, int N, int H, int W, int C, int K, int dtype, void* stream),
            (x, w, bias, slope, out, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_head_bwd, (const float* gout, const float* out, const void* x, const float* w, float slope, void* dx, float* dw, float* db, float* ws, int N, int H, int W, int C, int K, int dtype, void* stream),
            (gout, out, x, w, slope, dx, dw, db, ws, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_head_bnin_supported, (int N, int H, int W, int C, int K, int dtype), (N, H, W, C, K, dtype))
KP_DISPATCH(kp_head_fwd_bnin, (const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope_in, const float* w, const float* bias, float slope, float* out, int N, int H, int W, int C, int K, int dtype, void* stream),
            (yprev, mean, invstd, gamma, beta, slope_in, w, bias, slope, out, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_head_bwd_bnin, (const float* gout, const float* out, const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope_in, const float* w, float slope, void* dx, float* dw, float* db, float* ws, int N, int H, int W, int C, int K, int dtype, void* stream),
            (gout, out, yprev, mean, invstd, gamma, beta, slope_in, w, slope, dx, dw, db, ws, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_nhwc_to_nchw, (const void* src, float* dst, int N, int C, int H, int W, int pad, int dtype, void* stream), (src, dst, N, C, H, W, pad, dtype, stream))
KP_DISPATCH(kp_head_t_supported, (int N, int H, int W, int C, int K, int dtype), (N, H, W, C, K, dtype))
KP_DISPATCH(kp_head_fwd_t, (const void* x, int bnin, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope_in, const float* w, const float* bias, float slope, void* out_t, int N, int H, int W, int C, int K, int dtype, void* stream),
            (x, bnin, mean, invstd, gamma, beta, slope_in, w, bias, slope, out_t, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_head_bwd_t, (const void* gout_t, const void* out_t, const void* x, int bnin, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope_in, const float* w, float slope, void* dx, float* dw, float* db, float* ws, int N, int H, int W, int C, int K, int dtype, void* stream),
            (gout_t, out_t, x, bnin, mean, invstd, gamma, beta, slope_in, w, slope, dx, dw, db, ws, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_conv_pair_supported, (int N, int H, int W, int C, int K, int first, int dtype), (N, H, W, C, K, first, dtype))
KP_DISPATCH(kp_conv_fwd_pair, (const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi, int C, int K, int dtype, void* stream),
            (x, w, bias, y, stats, N, Hi, Wi, C, K, dtype, stream))
KP_DISPATCH(kp_conv_fwd_bnin_pair, (const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope, const void* w, const float* bias, void* y, float* stats, int N, int H, int W, int C, int K, int dtype, void* stream),
            (yprev, mean, invstd, gamma, beta, slope, w, bias, y, stats, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_head_pair_supported, (int N, int H, int W, int C, int K, int t_out, int dtype), (N, H, W, C, K, t_out, dtype))
KP_DISPATCH(kp_head_fwd_bnin_pair, (const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope_in, const float* w, const float* bias, float slope, void* out0, void* out1, int t_out, int N, int H, int W, int C, int K, int dtype, void* stream),
            (yprev, mean, invstd, gamma, beta, slope_in, w, bias, slope, out0, out1, t_out, N, H, W, C, K, dtype, stream))
KP_DISPATCH(kp_keypoint_bottleneck_t_supported, (int B, int C, int K, int H, int W, int dtype), (B, C, K, H, W, dtype))

// the T form of the fused bottleneck (kp_bottleneck_t.hip): dtype dispatch + the roofline leg's byte count
int kp_keypoint_bottleneck_fwd_t__bf16(const float*, const float*, const void*, const void*, float*, float*, float*, float*, void*, float*, float*, unsigned char*, int, int, int, int, int, float, float, int, void*);
int kp_keypoint_bottleneck_fwd_t__f16(const float*, const float*, const void*, const void*, float*, float*, float*, float*, void*, float*, float*, unsigned char*, int, int, int, int, int, float, float, int, void*);
int kp_keypoint_bottleneck_bwd_t__bf16(const void*, const float*, const float*, const void*, const void*, const float*, const float*, const unsigned char*, const float*, const float*, const float*, void*, float*, int, int, int, int, int, float, float, int, void*);
int kp_keypoint_bottleneck_bwd_t__f16(const void*, const float*, const float*, const void*, const void*, const float*, const float*, const unsigned char*, const float*, const float*, const float*, void*, float*, int, int, int, int, int, float, float, int, void*);

int kp_keypoint_bottleneck_fwd_t(const float* heat_s, const float* heat_t, const void* phi_s, const void* phi_t, float* k_t,
                                 float* p_t, float* q_t, float* m_t, void* a, float* mask_s, float* mask_t, unsigned char* arg_t,
                                 int B, int C, int K, int H, int W, float sigma, float eps, int dtype, void* stream) {
    if (dtype != KP_BF16 && dtype != KP_F16) return KP_E_UNSUPPORTED;
    // 2 heat stacks (fp32) + 2 feature stacks (T) in; padded blend (T), target maps, 2 masks (fp32), argmax, keypoints, marginals out
    const double hw = (double)H * W;
    const double bytes = (double)B * (4.0 * (2.0 * K * hw + K * hw + 2.0 * hw) + 2.0 * (2.0 * C * hw + C * (H + 2.0) * (W + 2.0)) + hw + 4.0 * K * (2 + H + W));
    ProfScope ps(KP_PROF_BOTTLENECK, 0.0, (hipStream_t)stream, bytes);
    const int rc = dtype == KP_F16
        ? kp_keypoint_bottleneck_fwd_t__f16(heat_s, heat_t, phi_s, phi_t, k_t, p_t, q_t, m_t, a, mask_s, mask_t, arg_t, B, C, K, H, W, sigma, eps, dtype, stream)
        : kp_keypoint_bottleneck_fwd_t__bf16(heat_s, heat_t, phi_s, phi_t, k_t, p_t, q_t, m_t, a, mask_s, mask_t, arg_t, B, C, K, H, W, sigma, eps, dtype, stream);
    ps.done(KP_PROF_BOTTLENECK);
    return rc;
}

int kp_keypoint_bottleneck_bwd_t(const void* ga, const float* g_k, const float* g_m, const void* phi_s, const void* phi_t,
                                 const float* mask_s, const float* mask_t, const unsigned char* arg_t, const float* k_t,
                                 const float* p_t, const float* q_t, void* d_phi_t, float* d_heat_t, int B, int C, int K, int H,
                                 int W, float sigma, float eps, int dtype, void* stream) {
    if (dtype != KP_BF16 && dtype != KP_F16) return KP_E_UNSUPPORTED;
    // in: padded gradient, phi_s, phi_t (T), masks, argmax, keypoints, marginals; out: d_phi_t (T), d_heat_t (fp32)
    const double hw = (double)H * W;
    const double bytes = (double)B * (2.0 * (C * (H + 2.0) * (W + 2.0) + 3.0 * C * hw) + 4.0 * (2.0 * hw + K * hw) + hw + 4.0 * K * (2 + H + W));
    ProfScope ps(KP_PROF_BOTTLENECK, 0.0, (hipStream_t)stream, bytes);
    const int rc = dtype == KP_F16
        ? kp_keypoint_bottleneck_bwd_t__f16(ga, g_k, g_m, phi_s, phi_t, mask_s, mask_t, arg_t, k_t, p_t, q_t, d_phi_t, d_heat_t, B, C, K, H, W, sigma, eps, dtype, stream)
        : kp_keypoint_bottleneck_bwd_t__bf16(ga, g_k, g_m, phi_s, phi_t, mask_s, mask_t, arg_t, k_t, p_t, q_t, d_phi_t, d_heat_t, B, C, K, H, W, sigma, eps, dtype, stream);
    ps.done(KP_PROF_BOTTLENECK);
    return rc;
}

long long kp_conv_first_fused_rows__bf16(int, int, int, int, int, int);
long long kp_conv_first_fused_rows__f16(int, int, int, int, int, int);
long long kp_conv_first_fused_rows(int N, int H, int W, int C, int K, int dtype) {
    return dtype == KP_F16 ? kp_conv_first_fused_rows__f16(N, H, W, C, K, dtype) : kp_conv_first_fused_rows__bf16(N, H, W, C, K, dtype);
}


long long kp_head_bwd_ws_floats__bf16(int, int, int, int, int, int);
long long kp_head_bwd_ws_floats__f16(int, int, int, int, int, int);
long long kp_head_bwd_ws_floats(int N, int H, int W, int C, int K, int dtype) {
    return dtype == KP_F16 ? kp_head_bwd_ws_floats__f16(N, H, W, C, K, dtype) : kp_head_bwd_ws_floats__bf16(N, H, W, C, K, dtype);
}

/* Row counts as pure functions of the shape (and the device): the public dispatch runs in query mode -- every launcher
 * records its row count and returns before launching.  Replaces the kp_last_rows() side channel. */
long long kp_conv_fwd_stats_rows(int N, int Hi, int Wi, int C, int K, int R, int dtype) {
    kp_set_query_mode(1);
    float dummy;   // non-NULL stats: the launchers only look at the pointer's truth value in query mode
    const int rc = kp_conv_fwd(nullptr, nullptr, nullptr, nullptr, &dummy, N, Hi, Wi, C, K, R, dtype, nullptr);
    kp_set_query_mode(0);
    return rc == KP_OK ? kp_last_rows() : (long long)rc;
}

long long kp_conv_dgrad_bnred_rows(int N, int H, int W, int K, int C, int dtype) {
    kp_set_query_mode(1);
    const int rc = kp_conv_dgrad_bnred(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, N, H, W, K, C, dtype, nullptr);
    kp_set_query_mode(0);
    return rc == KP_OK ? kp_last_rows() : (long long)rc;
}

long long kp_bn_bwd_rows(int gp, int tail, int N, int H, int W, int C, int dtype) {
    kp_set_query_mode(1);
    const int rc = kp_bn_act_bwd_reduce(nullptr, gp, tail, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, nullptr,
                                        N, H, W, C, dtype, nullptr);
    kp_set_query_mode(0);
    return rc == KP_OK ? kp_last_rows() : (long long)rc;
}

}  // extern "C"
