// Convolution entry points: dispatch between the kernel families of this build's 16-bit storage type (compiled once
// for bf16 and once for fp16; kp_api.hip holds the public names and the dtype dispatch).
#include <stdlib.h>
#include "kp_prof.h"

int kp_conv_fwd_generic(const void*, const void*, const float*, void*, float*, int, int, int, int, int, int, int,
                        hipStream_t);
int kp_conv_wgrad_generic(const void*, const void*, float*, int, int, int, int, int, int, int, int, hipStream_t);
int kp_conv_fwd_mfma(const void*, const void*, const float*, void*, float*, int, int, int, int, int, int,
                     hipStream_t);   // bf16 only; returns KP_E_UNSUPPORTED when the shape is not eligible
int kp_conv_dgrad_mfma(const void*, const void*, void*, int, int, int, int, int, hipStream_t);
int kp_conv_patch_launch_act(const void*, const void*, const float*, const float*, float, void*, int, int, int, int, int, int, int, int, int, int, int, hipStream_t);
int kp_conv_wgrad_mfma(const void*, const void*, float*, int, int, int, int, int, int, int, hipStream_t);
int kp_conv_wgrad_mfma_accepts(int, int, int, int, int, int);
int kp_conv_wgrad_patch(const void*, const void*, float*, int, int, int, int, int, int, hipStream_t);
long long kp_conv_wgrad_patch_partials(int, int, int, int, int, int);
int kp_conv_small_fwd(const void*, const void*, const float*, void*, float*, int, int, int, int, int, hipStream_t);
int kp_conv_small_wgrad(const void*, const void*, float*, int, int, int, int, int, int, hipStream_t);
int kp_conv_small_fwd_bnin_ok(int, int, int, int, int);
int kp_conv_small_fwd_bnin(const void*, const float*, const float*, const float*, const float*, float, const void*, const float*,
                           void*, float*, int, int, int, int, int, hipStream_t);
int kp_conv_small_fwd_bnin_pair(const void*, const float*, const float*, const float*, const float*, float, const void*, const float*,
                                void*, float*, int, int, int, int, int, hipStream_t);
int kp_conv_small_wgrad_bnin_ok(int, int, int, int, int);
int kp_conv_small_wgrad_bnin(const void*, const float*, const float*, const float*, const float*, float, const void*, float*, int,
                             int, int, int, int, int, hipStream_t);
int kp_conv_small_dgrad_bnred_ok(int, int, int, int, int);
int kp_conv_small_dgrad_bnred(const void*, const void*, void*, const void*, const float*, const float*, const float*, const float*, float,
                              float*, int, int, int, int, int, hipStream_t);
int kp_conv_first_fwd(const void*, const void*, const float*, void*, float*, int, int, int, int, int, hipStream_t, int pair = 0);
int kp_conv_first_wgrad(const void*, const void*, float*, int, int, int, int, int, int, hipStream_t);
long long kp_conv_first_wgrad_partials(int, int, int, int, int);
int kp_conv_small_first_wgrad(const void*, const void*, float*, int, int, int, int, int, int, hipStream_t);
long long kp_conv_small_first_wgrad_partials(int, int, int, int, int);
long long kp_conv_small_wgrad_partials(int, int, int, int, int);
long long kp_conv_wgrad_generic_partials(int, int, int, int, int, int);
// strict fp32 on the matrix cores (kp_conv_f32.hip, v_mfma_f32_32x32x2_f32): 3x3, C % 16 == 0, K % 64 == 0
int kp_conv_fwd_f32mfma(const float*, const float*, const float*, float*, float*, int, int, int, int, int, hipStream_t);
long long kp_conv_wgrad_f32mfma_partials(int, int, int, int, int);
int kp_conv_wgrad_f32mfma(const float*, const float*, float*, int, int, int, int, int, int, hipStream_t);


long long kp_conv_patch_bred_rows(int, int, int, int, int);
long long kp_conv_ring_bred_rows(int, int, int);
int kp_conv_patch_launch_bred(const void*, const void*, void*, const void*, const float*, const float*, const float*, const float*, float, float*,
                              int, int, int, int, int, hipStream_t);
int kp_conv_ring_dgrad_bred(const void*, const void*, void*, const void*, const float*, const float*, const float*, const float*, float, float*,
                            int, int, int, int, int, hipStream_t);
extern kp_knob_t g_kp_bred_patch;   // kp_config(KP_CFG_BRED_PATCH, v): 1 (default) the matrix-core layers' data gradient offers the fused form
extern kp_knob_t g_kp_wgrad_patch;

extern kp_knob_t g_kp_disable_mfma;   // kp_config(KP_CFG_DISABLE_MFMA, 1)
static int use_mfma() { return g_kp_disable_mfma ? 0 : 1; }

extern "C" {

int kp_conv_fwd(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi, int C,
                int K, int R, int dtype, void* stream) {
    KP_REQUIRE(kp_query_mode() || (x && w && y), "kp_conv_fwd: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && K > 0 && (R == 1 || R == 3) && Hi >= R && Wi >= R, "kp_conv_fwd: bad dims N=%d Hi=%d Wi=%d C=%d K=%d R=%d", N, Hi, Wi, C, K, R);
    KP_REQUIRE(dtype == KP_F32 || dtype == KP_H16, "kp_conv_fwd: bad dtype %d", dtype);
    const double flops = 2.0 * N * (double)(Hi - R + 1) * (Wi - R + 1) * C * K * R * R;
    const double esz = dtype == KP_H16 ? 2.0 : 4.0;   // algorithmic HBM bytes: input + output + weights, each once
    const double bytes = esz * ((double)N * Hi * Wi * C + (double)N * (Hi - R + 1) * (Wi - R + 1) * K + (double)R * R * C * K);
    ProfScope ps(KP_PROF_CONV_GENERIC, flops, (hipStream_t)stream, bytes);
    if (stats) kp_set_last_rows(((long long)N * (Hi - R + 1) * (Wi - R + 1) + 63) / 64);
    if (dtype == KP_H16 && use_mfma()) {
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
    if (dtype == KP_F32 && R == 3 && use_mfma()) {
        int rc = kp_conv_fwd_f32mfma((const float*)x, (const float*)w, bias, (float*)y, stats, N, Hi, Wi, C, K, (hipStream_t)stream);
        if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_CONV_F32MFMA); return rc; }
    }
    int rc = kp_conv_fwd_generic(x, w, bias, y, stats, N, Hi, Wi, C, K, R, dtype, (hipStream_t)stream);
    ps.done(KP_PROF_CONV_GENERIC);
    return rc;
}

int kp_conv_dgrad(const void* dy, const void* w_dgrad, void* gx, int N, int H, int W, int K, int C, int dtype,
                  void* stream) {
    KP_REQUIRE(dy && w_dgrad && gx, "kp_conv_dgrad: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && K > 0 && H > 0 && W > 0, "kp_conv_dgrad: bad dims");
    KP_REQUIRE(dtype == KP_F32 || dtype == KP_H16, "kp_conv_dgrad: bad dtype %d", dtype);
    if (dtype == KP_H16 && use_mfma()) {
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
    if (dtype == KP_H16 && use_mfma() && R == 3 && g_kp_wgrad_patch) {
        const long long p = kp_conv_wgrad_patch_partials(N, Ho, Wo, C, K, dy_pad);
        if (p > 0 && p < 0x7fffffffLL) return (int)p;
    }
    if (dtype == KP_H16 && use_mfma() && R == 3 && C <= 4 && (K == 16 || K == 32)) {   // Pong-class first layers: the 16 x 16 x 32 kernel
        const long long p = kp_conv_small_first_wgrad_partials(N, Ho, Wo, C, K);
        if (p > 0) return (int)p;
    }
    if (dtype == KP_H16 && use_mfma() && R == 3 && C <= 4) {
        const long long p = kp_conv_first_wgrad_partials(N, Ho, Wo, C, K);
        if (p > 0) return (int)p;
    }
    if (dtype == KP_H16 && use_mfma() && R == 3) {
        const long long p = kp_conv_small_wgrad_partials(N, Ho, Wo, C, K);
        if (p > 0) return (int)p;
    }
    // the per-tap MFMA kernel (kp_wgrad_mfma.hip): 64-multiple channels the tap-reuse kernel declines -- (64 x 64) gradient blocks (the
    // 64 -> 64 layers of vgg_cfg 'B' / 'D' / 'E', vgg.py:55-57) and maps smaller than its pixel block; one partial per pixel split
    if (dtype == KP_H16 && use_mfma()) {
        const int p = kp_conv_wgrad_mfma_accepts(N, Ho, Wo, C, K, R);
        if (p > 0) return p;
    }
    if (dtype == KP_F32 && R == 3 && use_mfma()) {
        const long long p = kp_conv_wgrad_f32mfma_partials(N, Ho, Wo, C, K);
        if (p > 0 && p < 0x7fffffffLL) return (int)p;
    }
    const long long p = kp_conv_wgrad_generic_partials(N, Ho, Wo, C, K, R);
    return p < 0x7fffffffLL ? (int)p : 1;
}

int kp_conv_wgrad(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int R, int dy_pad,
                  int dtype, void* stream) {
    KP_REQUIRE(x && dy && scratch, "kp_conv_wgrad: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && K > 0 && (R == 1 || R == 3) && Ho > 0 && Wo > 0 && dy_pad >= 0, "kp_conv_wgrad: bad dims");
    KP_REQUIRE(dtype == KP_F32 || dtype == KP_H16, "kp_conv_wgrad: bad dtype %d", dtype);
    const double flops = 2.0 * N * (double)Ho * Wo * C * K * R * R;
    const double wesz = dtype == KP_H16 ? 2.0 : 4.0;
    const double wbytes = wesz * ((double)N * (Ho + R - 1) * (Wo + R - 1) * C + (double)N * Ho * Wo * K) + 4.0 * R * R * C * K;
    ProfScope ps(KP_PROF_WGRAD_GENERIC, flops, (hipStream_t)stream, wbytes);
    if (dtype == KP_H16 && use_mfma()) {
        int rc = (R == 3 && g_kp_wgrad_patch) ? kp_conv_wgrad_patch(x, dy, scratch, N, Ho, Wo, C, K, dy_pad, (hipStream_t)stream)
                                              : KP_E_UNSUPPORTED;
        if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_WGRAD_MFMA); return rc; }
        rc = kp_conv_wgrad_mfma(x, dy, scratch, N, Ho, Wo, C, K, R, dy_pad, (hipStream_t)stream);
        if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_WGRAD_MFMA); return rc; }
        if (R == 3 && C <= 4 && (K == 16 || K == 32)) {
            rc = kp_conv_small_first_wgrad(x, dy, scratch, N, Ho, Wo, C, K, dy_pad, (hipStream_t)stream);
            if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_CONV_FIRST); return rc; }
        }
        if (R == 3 && C <= 4) {
            rc = kp_conv_first_wgrad(x, dy, scratch, N, Ho, Wo, C, K, dy_pad, (hipStream_t)stream);
            if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_CONV_FIRST); return rc; }
        }
        if (R == 3) {
            rc = kp_conv_small_wgrad(x, dy, scratch, N, Ho, Wo, C, K, dy_pad, (hipStream_t)stream);
            if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_WGRAD_SMALL); return rc; }
        }
    }
    if (dtype == KP_F32 && R == 3 && use_mfma()) {
        int rc = kp_conv_wgrad_f32mfma((const float*)x, (const float*)dy, scratch, N, Ho, Wo, C, K, dy_pad, (hipStream_t)stream);
        if (rc != KP_E_UNSUPPORTED) { ps.done(KP_PROF_WGRAD_F32MFMA); return rc; }
    }
    int rc = kp_conv_wgrad_generic(x, dy, scratch, N, Ho, Wo, C, K, R, dy_pad, dtype, (hipStream_t)stream);
    ps.done(KP_PROF_WGRAD_GENERIC);
    return rc;
}

// A 3x3 layer with matrix-core channel counts on a map that is NOT a multiple of the tap-reuse kernel's 8 x 16 pixel block: the
// per-tap fallback runs at ~50 TFLOP/s where the tap-reuse kernel reaches > 1000.  The zero border of dy makes padding exact: on a map
// padded with ZERO gradient rows / columns (and any finite input there) every added pixel contributes 0 to every tap.  -> 1 and the
// padded map size if kp_conv_wgrad on zero-padded copies (kp_pad_hw of x [N][Ho+2][Wo+2][C] and dy [N][Ho+2dp][Wo+2dp][K] into
// [N][Hp+2][Wp+2][C] / [N][Hp+2dp][Wp+2dp][K]) takes the tap-reuse kernel; 0: call kp_conv_wgrad as it is.
int kp_conv_wgrad_pad_hw(int N, int Ho, int Wo, int C, int K, int dy_pad, int dtype, int* Hp, int* Wp) {
    if (dtype != KP_H16 || !use_mfma() || !g_kp_wgrad_patch || N <= 0 || Ho <= 0 || Wo <= 0) return 0;
    if (kp_conv_wgrad_patch_partials(N, Ho, Wo, C, K, dy_pad) > 0) return 0;            // eligible as it is
    const int hp = (Ho + 7) / 8 * 8, wp = (Wo + 15) / 16 * 16;
    if (kp_conv_wgrad_patch_partials(N, hp, wp, C, K, dy_pad) <= 0) return 0;           // channels the kernel does not take
    if (Hp) *Hp = hp;
    if (Wp) *Wp = wp;
    return 1;
}

// ---- 3x3 layers fed by another 3x3 layer: the previous layer's BatchNorm-apply + activation + replicate pad fused into the
// consumer's staging (kp_conv_small.hip, BnIn).  Shapes without a fused kernel answer 0 / KP_E_UNSUPPORTED and the caller
// runs kp_bn_act_fwd + kp_conv_fwd / kp_conv_wgrad instead: same results, bit for bit.
int kp_conv_bnin_supported(int N, int H, int W, int C, int K, int dtype) {
    if (dtype != KP_H16 || !use_mfma() || N <= 0 || H <= 0 || W <= 0) return 0;
    return kp_conv_small_fwd_bnin_ok(N, H, W, C, K) && kp_conv_small_wgrad_bnin_ok(N, H, W, C, K);
}

int kp_conv_fwd_bnin(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     float slope, const void* w, const float* bias, void* y, float* stats, int N, int H, int W, int C, int K,
                     int dtype, void* stream) {
    KP_REQUIRE(kp_query_mode() || (yprev && w && y), "kp_conv_fwd_bnin: null pointer");
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "kp_conv_fwd_bnin: bad dims");
    KP_REQUIRE(!mean == !invstd && !mean == !gamma && !mean == !beta, "kp_conv_fwd_bnin: mean / invstd / gamma / beta must be all set or all NULL");
    if (!kp_conv_bnin_supported(N, H, W, C, K, dtype)) return KP_E_UNSUPPORTED;
    const double flops = 2.0 * N * (double)H * W * C * K * 9;
    const double bytes = 2.0 * ((double)N * H * W * C + (double)N * H * W * K + 9.0 * C * K);
    ProfScope ps(KP_PROF_CONV_SMALL, flops, (hipStream_t)stream, bytes);
    const int rc = kp_conv_small_fwd_bnin(yprev, mean, invstd, gamma, beta, slope, w, bias, y, stats, N, H, W, C, K, (hipStream_t)stream);
    ps.done(KP_PROF_CONV_SMALL);
    return rc;
}

// ---- two half-batches in one launch ("pair" forms) ------------------------------------------------------------------------
// TransporterNet.forward runs every encoder twice on the same weights -- source frame under no_grad, then target frame
// (transporter.py:36-39) -- with BatchNorm2d in training mode computing the batch statistics of EACH call.  The pair forms take
// the two half-batches as one tensor [N = 2 x N/2] and launch each half exactly as a separate call of N / 2 images would be
// launched (same grid, same work per workgroup, same statistics rows -- results bit-identical to the two calls), as ONE grid:
// half as many launches, each twice as long (a third of a 30-60 us launch of these layers is its ends).
int kp_conv_pair_supported(int N, int H, int W, int C, int K, int first, int dtype) {
    if (dtype != KP_H16 || !use_mfma() || N < 2 || (N & 1) || H < 1 || W < 1) return 0;
    if (first) {   // the first-layer VALU kernel (kp_conv_first.hip)
        const long long M = (long long)(N / 2) * H * W;
        const int CP = K / 8;
        return C >= 1 && C <= 4 && K % 8 == 0 && K <= 128 && (CP & (CP - 1)) == 0 && M % 64 == 0 && !(K == 64 && C <= 3 && W % 32 == 0 && M >= 4096);
    }
    return kp_conv_small_fwd_bnin_ok(N / 2, H, W, C, K) ? 1 : 0;
}

int kp_conv_fwd_pair(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi, int C, int K,
                     int dtype, void* stream) {
    KP_REQUIRE(kp_query_mode() || (x && w && y), "kp_conv_fwd_pair: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && K > 0 && Hi >= 3 && Wi >= 3, "kp_conv_fwd_pair: bad dims");
    if (!kp_conv_pair_supported(N, Hi - 2, Wi - 2, C, K, 1, dtype)) return KP_E_UNSUPPORTED;
    const double flops = 2.0 * N * (double)(Hi - 2) * (Wi - 2) * C * K * 9;
    const double bytes = 2.0 * ((double)N * Hi * Wi * C + (double)N * (Hi - 2) * (Wi - 2) * K + 9.0 * C * K);
    ProfScope ps(KP_PROF_CONV_FIRST, flops, (hipStream_t)stream, bytes);
    const int rc = kp_conv_first_fwd(x, w, bias, y, stats, N, Hi, Wi, C, K, (hipStream_t)stream, 1);
    ps.done(KP_PROF_CONV_FIRST);
    return rc;
}

int kp_conv_fwd_bnin_pair(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                          float slope, const void* w, const float* bias, void* y, float* stats, int N, int H, int W, int C, int K,
                          int dtype, void* stream) {
    KP_REQUIRE(kp_query_mode() || (yprev && w && y), "kp_conv_fwd_bnin_pair: null pointer");
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "kp_conv_fwd_bnin_pair: bad dims");
    KP_REQUIRE(!mean == !invstd && !mean == !gamma && !mean == !beta, "kp_conv_fwd_bnin_pair: mean / invstd / gamma / beta must be all set or all NULL");
    if (!kp_conv_pair_supported(N, H, W, C, K, 0, dtype)) return KP_E_UNSUPPORTED;
    const double flops = 2.0 * N * (double)H * W * C * K * 9;
    const double bytes = 2.0 * ((double)N * H * W * C + (double)N * H * W * K + 9.0 * C * K);
    ProfScope ps(KP_PROF_CONV_SMALL, flops, (hipStream_t)stream, bytes);
    const int rc = kp_conv_small_fwd_bnin_pair(yprev, mean, invstd, gamma, beta, slope, w, bias, y, stats, N, H, W, C, K, (hipStream_t)stream);
    ps.done(KP_PROF_CONV_SMALL);
    return rc;
}

int kp_conv_wgrad_bnin(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       float slope, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int dy_pad, int dtype,
                       void* stream) {
    KP_REQUIRE(yprev && dy && scratch, "kp_conv_wgrad_bnin: null pointer");
    KP_REQUIRE(N > 0 && Ho > 0 && Wo > 0 && C > 0 && K > 0 && dy_pad >= 0, "kp_conv_wgrad_bnin: bad dims");
    KP_REQUIRE(!mean == !invstd && !mean == !gamma && !mean == !beta, "kp_conv_wgrad_bnin: mean / invstd / gamma / beta must be all set or all NULL");
    if (!kp_conv_bnin_supported(N, Ho, Wo, C, K, dtype)) return KP_E_UNSUPPORTED;
    const double flops = 2.0 * N * (double)Ho * Wo * C * K * 9;
    const double bytes = 2.0 * ((double)N * Ho * Wo * C + (double)N * Ho * Wo * K) + 4.0 * 9 * C * K;
    ProfScope ps(KP_PROF_WGRAD_SMALL, flops, (hipStream_t)stream, bytes);
    const int rc = kp_conv_small_wgrad_bnin(yprev, mean, invstd, gamma, beta, slope, dy, scratch, N, Ho, Wo, C, K, dy_pad, (hipStream_t)stream);
    ps.done(KP_PROF_WGRAD_SMALL);
    return rc;
}

// ---- the data gradient of a 3x3 layer that also leaves the first pass of the PRODUCING layer's BatchNorm backward ------------
// (kp_conv_small.hip, BRED): gx as from kp_conv_dgrad, bit for bit; partials = kp_bn_act_bwd_reduce's sums of the layer below
// (gp = 1, tail none) up to the order of the fp32 additions -- that pass is then not run.  Rows written: the return value
// of kp_conv_dgrad_bnred_rows (= kp_last_rows() after the call), at most ceil(N * (H + 2) * (W + 2) / 64) rows of [2][C].
int kp_conv_dgrad_bnred_supported(int N, int H, int W, int K, int C, int dtype) {
    if (dtype != KP_H16 || !use_mfma() || N <= 0 || H <= 0 || W <= 0) return 0;
    if (kp_conv_small_dgrad_bnred_ok(N, H, W, K, C)) return 1;
    return g_kp_bred_patch && H >= 16 && W >= 16 && kp_conv_patch_bred_rows(N, H, W, K, C) > 0 ? 1 : 0;
}

int kp_conv_dgrad_bnred(const void* dy, const void* w_dgrad, void* gx, const void* yred, const float* mean, const float* invstd,
                        const float* gamma, const float* beta, float slope, float* partials, int N, int H, int W, int K, int C,
                        int dtype, void* stream) {
    KP_REQUIRE(kp_query_mode() || (dy && w_dgrad && gx && yred && partials), "kp_conv_dgrad_bnred: null pointer");
    KP_REQUIRE(N > 0 && C > 0 && K > 0 && H > 0 && W > 0, "kp_conv_dgrad_bnred: bad dims");
    KP_REQUIRE(!mean == !invstd && !mean == !gamma && !mean == !beta, "kp_conv_dgrad_bnred: mean / invstd / gamma / beta must be all set or all NULL");
    if (!kp_conv_dgrad_bnred_supported(N, H, W, K, C, dtype)) return KP_E_UNSUPPORTED;
    const double flops = 2.0 * N * (double)H * W * C * K * 9;
    const double bytes = 2.0 * ((double)N * (H + 4) * (W + 4) * K + (double)N * (H + 2) * (W + 2) * C + (double)N * H * W * C + 9.0 * C * K);
    if (kp_conv_small_dgrad_bnred_ok(N, H, W, K, C)) {
        ProfScope ps(KP_PROF_CONV_SMALL, flops, (hipStream_t)stream, bytes);
        const int rc = kp_conv_small_dgrad_bnred(dy, w_dgrad, gx, yred, mean, invstd, gamma, beta, slope, partials, N, H, W, K, C, (hipStream_t)stream);
        ps.done(KP_PROF_CONV_SMALL);
        return rc;
    }
    // matrix-core layers: interior (LDS-patch kernel, rows first) + border ring (its rows behind them)
    const long long prows = kp_conv_patch_bred_rows(N, H, W, K, C);
    kp_set_last_rows(prows + kp_conv_ring_bred_rows(N, H, W));
    KP_QUERY_RETURN();
    ProfScope ps(KP_PROF_CONV_MFMA, flops, (hipStream_t)stream, bytes);
    int rc = kp_conv_patch_launch_bred(dy, w_dgrad, gx, yred, mean, invstd, gamma, beta, slope, partials, N, H, W, K, C, (hipStream_t)stream);
    if (rc == KP_OK)
        rc = kp_conv_ring_dgrad_bred(dy, w_dgrad, gx, yred, mean, invstd, gamma, beta, slope, partials + prows * 2 * C, N, H, W, K, C,
                                     (hipStream_t)stream);
    ps.done(KP_PROF_CONV_MFMA);
    return rc;
}

// ---- inference: Conv2d(3x3) + eval-mode BatchNorm2d + LeakyReLU as ONE kernel (row f3 widened to the matrix-core layers) ---------------
// out[n][p + out_pad][q + out_pad][k] = act((x * w)[n][p][q][k] * scale[k] + shift[k]) into out [N][H + 2 out_pad][W + 2 out_pad][K]
// (the border ring is NOT written: kp_pad_border replicates into it).  x: [N][H+2][W+2][C] with its replicate border; w: the packed
// forward weights [K][3][3][C].  16-bit storage, C and K multiples of 64, H and W multiples of 16; KP_E_UNSUPPORTED otherwise.
int kp_conv_fwd_act_supported(int N, int H, int W, int C, int K, int dtype) {
    if (dtype != KP_H16 || !use_mfma() || N <= 0 || H <= 0 || W <= 0) return 0;
    kp_set_query_mode(1);
    const int rc = kp_conv_patch_launch_act(nullptr, nullptr, nullptr, nullptr, 0.f, nullptr, N, H + 2, W + 2, 0, H, W, 0, H, W, C, K, nullptr);
    kp_set_query_mode(0);
    return rc == KP_OK ? 1 : 0;
}

int kp_conv_fwd_act(const void* x, const void* w, const float* scale, const float* shift, float slope, void* out, int out_pad, int N,
                    int H, int W, int C, int K, int dtype, void* stream) {
    KP_REQUIRE(x && w && scale && shift && out && N > 0 && H > 0 && W > 0 && C > 0 && K > 0 && out_pad >= 0, "kp_conv_fwd_act: bad args");
    KP_REQUIRE(slope >= 0.f && slope <= 1.f, "kp_conv_fwd_act: activation slope must be in [0, 1]");
    if (dtype != KP_H16 || !use_mfma()) return KP_E_UNSUPPORTED;
    const double flops = 2.0 * N * (double)H * W * C * K * 9;
    const double bytes = 2.0 * ((double)N * (H + 2) * (W + 2) * C + (double)N * H * W * K + 9.0 * C * K);
    ProfScope ps(KP_PROF_CONV_MFMA, flops, (hipStream_t)stream, bytes);
    const int rc = kp_conv_patch_launch_act(x, w, scale, shift, slope, out, N, H + 2, W + 2, 0, H + 2 * out_pad, W + 2 * out_pad, out_pad, H, W,
                                            C, K, (hipStream_t)stream);
    if (rc == KP_E_UNSUPPORTED) { ps.cancel(); return rc; }
    ps.done(KP_PROF_CONV_MFMA);
    return rc;
}

}  // extern "C"
