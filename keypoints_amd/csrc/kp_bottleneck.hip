// Keypoint bottleneck kernels: marginal (log-)softmax -> (y,x) keypoints, Gaussian-like
// map render, 'max' transport blend, and their backward passes.
// Reference: keypoints/models/functional.py:5-72, keypoints/models/transporter.py:57-60.
//
// These are small fp32 NCHW tensors (K x h x w per image, h = w = 16 at 128x128 input):
// HBM/latency-class work.  Heat-maps are staged through LDS in row chunks (coalesced
// reads over H x W), marginals are reduced per wavefront.
#include "kp_common.h"
#include "kp_rowsum.h"

namespace {

constexpr int TPB = 256;
constexpr int CHUNK_FLOATS = 8192;   // 32 KiB LDS staging tile

// One workgroup per heat-map.  LDS: tile[rc][pitch] | rowv[H] | colv[W]
__global__ __launch_bounds__(TPB) void spatial_softmax_fwd_k(const float* __restrict__ heat, float* __restrict__ kout,
                                                             float* __restrict__ pout, float* __restrict__ qout,
                                                             int H, int W, int rc, int pitch, int log_domain) {
    extern __shared__ float sm[];
    float* tile = sm;
    float* rowv = sm + (size_t)rc * pitch;
    float* colv = rowv + H;
    const int bk = blockIdx.x, tid = threadIdx.x;
    const float* src = heat + (long long)bk * H * W;

    for (int j = tid; j < W; j += TPB) colv[j] = 0.f;
    for (int r0 = 0; r0 < H; r0 += rc) {
        const int rows = (H - r0 < rc) ? H - r0 : rc;
        __syncthreads();
        for (int e = tid; e < rows * W; e += TPB) tile[(e / W) * pitch + (e % W)] = src[(long long)r0 * W + e];
        __syncthreads();
        // row sums: eight rows per wavefront pass, a lane's partial over its columns lane, lane + 64, ... then the halving butterfly of
        // kp_rowsum.h -- the order of the fused kernels' wave_keypoint (kp_bottleneck_dev.h), to which this kernel stays bit-identical
        for (int b8 = tid >> 6; b8 * 8 < rows; b8 += TPB / 64) {
            const int lane = tid & 63;
            float pp[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = b8 * 8 + r;
                float s = 0.f;
                if (i < rows) {
                    s = lane < W ? tile[i * pitch + lane] : 0.f;
                    for (int j = lane + 64; j - lane < 256 || j < W; j += 64) s += j < W ? tile[i * pitch + j] : 0.f;
                }
                pp[r] = s;
            }
            int idx;
            const float tot = kp_rows8_over64(pp, lane, idx);
            if ((lane & 7) == 0 && b8 * 8 + idx < rows) rowv[r0 + b8 * 8 + idx] = tot / (float)W;
        }
        for (int j = tid; j < W; j += TPB) {      // column partial sums
            float s = colv[j];
            for (int i = 0; i < rows; ++i) s += tile[i * pitch + j];
            colv[j] = s;
        }
    }
    __syncthreads();
    for (int j = tid; j < W; j += TPB) colv[j] = colv[j] / (float)H;
    __syncthreads();

    // wave 0: rows (ky), wave 1: columns (kx)
    const int wave = tid >> 6, lane = tid & 63;
    if (wave < 2) {
        const int L = wave == 0 ? H : W;
        float* a = wave == 0 ? rowv : colv;
        float* dst = (wave == 0 ? pout + (long long)bk * H : qout + (long long)bk * W);
        float mx = -INFINITY;
        for (int i = lane; i < L; i += 64) mx = fmaxf(mx, a[i]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int i = lane; i < L; i += 64) se += expf(a[i] - mx);
        se = wave_sum(se);
        float key = 0.f;
        if (log_domain) {
            const float lse = logf(se);
            for (int i = lane; i < L; i += 64) {
                const float lp = (a[i] - mx) - lse;
                const float r = ruler_at(i, L);
                if (r > 0.f) key += expf(lp + logf(r));
                if (dst) dst[i] = expf(lp);
            }
        } else {
            for (int i = lane; i < L; i += 64) {
                const float pr = expf(a[i] - mx) / se;
                key += pr * ruler_at(i, L);
                if (dst) dst[i] = pr;
            }
        }
        key = wave_sum(key);
        if (lane == 0) kout[(long long)bk * 2 + wave] = key;
    }
}

__global__ void spatial_softmax_bwd_k(const float* __restrict__ gk, const float* __restrict__ k,
                                      const float* __restrict__ p, const float* __restrict__ q,
                                      float* __restrict__ dheat, int BK, int H, int W) {
    const long long total = (long long)BK * H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % W);
    const int i = (int)((t / W) % H);
    const long long bk = t / ((long long)W * H);
    const float ky = k[bk * 2], kx = k[bk * 2 + 1];
    const float gy = gk[bk * 2], gx = gk[bk * 2 + 1];
    dheat[t] = p[bk * H + i] * (ruler_at(i, H) - ky) * gy / (float)W + q[bk * W + j] * (ruler_at(j, W) - kx) * gx / (float)H;
}

template <int MODE>   // 0 gaussian-like, 1 point map
__global__ void render_fwd_k(const float* __restrict__ k, float* __restrict__ m, int BK, int H, int W, float inv2s2,
                             float eps) {
    const long long total = (long long)BK * H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % W);
    const int i = (int)((t / W) % H);
    const long long bk = t / ((long long)W * H);
    const float dy = ruler_at(i, H) - k[bk * 2], dx = ruler_at(j, W) - k[bk * 2 + 1];
    const float d2 = dy * dy + dx * dx;
    if (MODE == 0) m[t] = kp_gauss(d2 + eps, inv2s2);
    else m[t] = -d2;
}

__global__ __launch_bounds__(TPB) void gaussian_bwd_k(const float* __restrict__ gm, const float* __restrict__ k,
                                                      float* __restrict__ dk, int H, int W, float inv2s2, float eps) {
    const int bk = blockIdx.x;
    const float ky = k[bk * 2], kx = k[bk * 2 + 1];
    float sy = 0.f, sx = 0.f;
    for (int e = threadIdx.x; e < H * W; e += TPB) {
        const int i = e / W, j = e % W;
        const float dy = ruler_at(i, H) - ky, dx = ruler_at(j, W) - kx;
        const float d = kp_fast_sqrt(dy * dy + dx * dx + eps);
        const float mv = kp_fast_exp(-d * inv2s2);
        const float c = gm[(long long)bk * H * W + e] * mv * inv2s2 / d;
        sy += c * dy;
        sx += c * dx;
    }
    sy = wave_sum(sy);
    sx = wave_sum(sx);
    __shared__ float part[2][TPB / 64];
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = sy; part[1][threadIdx.x >> 6] = sx; }
    __syncthreads();
    if (threadIdx.x < 2) {
        float s = 0.f;
        for (int i = 0; i < TPB / 64; ++i) s += part[threadIdx.x][i];
        dk[bk * 2 + threadIdx.x] = s;
    }
}

__global__ void transport_max_fwd_k(const float* __restrict__ phi_s, const float* __restrict__ phi_t,
                                    const float* __restrict__ m_s, const float* __restrict__ m_t,
                                    float* __restrict__ phi, float* __restrict__ mask_s, float* __restrict__ mask_t,
                                    unsigned char* __restrict__ arg_s, unsigned char* __restrict__ arg_t, int B, int C,
                                    int K, int HW) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * HW) return;
    const long long b = t / HW;
    const int pix = (int)(t % HW);
    float bs = 0.f, bt = 0.f;
    int as = 0, at = 0;
    for (int k = 0; k < K; ++k) {
        const float vs = m_s[(b * K + k) * HW + pix], vt = m_t[(b * K + k) * HW + pix];
        if (k == 0 || vs > bs) { bs = vs; as = k; }
        if (k == 0 || vt > bt) { bt = vt; at = k; }
    }
    mask_s[t] = bs;
    mask_t[t] = bt;
    if (arg_s) arg_s[t] = (unsigned char)as;
    if (arg_t) arg_t[t] = (unsigned char)at;
    const float keep = (1.f - bs) * (1.f - bt);
    for (int c = 0; c < C; ++c) {
        const long long o = (b * C + c) * HW + pix;
        phi[o] = phi_s[o] * (1.f - bs) * (1.f - bt) + phi_t[o] * bt;
    }
    (void)keep;
}

__global__ void transport_max_bwd_k(const float* __restrict__ g, const float* __restrict__ phi_s,
                                    const float* __restrict__ phi_t, const float* __restrict__ mask_s,
                                    const float* __restrict__ mask_t, const unsigned char* __restrict__ arg_s,
                                    const unsigned char* __restrict__ arg_t, float* __restrict__ d_phi_s,
                                    float* __restrict__ d_phi_t, float* __restrict__ d_m_s, float* __restrict__ d_m_t,
                                    int B, int C, int K, int HW) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * HW) return;
    const long long b = t / HW;
    const int pix = (int)(t % HW);
    const float ms = mask_s[t], mt = mask_t[t];
    float gms = 0.f, gmt = 0.f;
    for (int c = 0; c < C; ++c) {
        const long long o = (b * C + c) * HW + pix;
        const float gv = g[o], ps = phi_s[o], pt = phi_t[o];
        gmt += gv * (pt - ps * (1.f - ms));
        gms -= gv * ps * (1.f - mt);
        if (d_phi_t) d_phi_t[o] = gv * mt;
        if (d_phi_s) d_phi_s[o] = gv * (1.f - ms) * (1.f - mt);
    }
    for (int k = 0; k < K; ++k) {
        const long long o = (b * K + k) * HW + pix;
        if (d_m_t) d_m_t[o] = (k == arg_t[t]) ? gmt : 0.f;
        if (d_m_s) d_m_s[o] = (k == arg_s[t]) ? gms : 0.f;
    }
}

// ---- the other three combine modes of TransporterNet.forward (transporter.py:41-55) ---------------------------------
// One thread per (image, pixel); channels / keypoints are walked in registers.  MODE 1 'sum_and_clamp':
// mask = clamp(sum_k m, 0, 1); MODE 2 'pretrained_network': mask = m (one channel, from MaskMaker); MODE 3 'loop':
// phi_0 = phi_s, phi_{i+1} = phi_i (1 - m_s,i)(1 - m_t,i) + phi_t m_t,i  for i < K (the masks returned are the last slices).
constexpr int LOOP_MAXK = 64;

// KM ('loop' only): a compile-time bound on K (4 / 16 / LOOP_MAXK).  The per-keypoint arrays of a thread are indexed in fully unrolled
// loops, so for K <= 16 they live in registers; with the run-time bound alone (rounds 1-5) the 5 x 64 floats of the backward sat in
// scratch memory and 'loop' -- the DEFAULT combine_method of TransporterNet.__init__, transporter.py:14 -- took 3.6 x the fused 'max'
// pair at config 2's bottleneck (tools/bench_f4.py, profiles/r06_f4_modes.txt).  Same expressions in the same order: bit-identical.
template <int MODE, int KM = LOOP_MAXK>
__global__ void transport_mode_fwd_k(const float* __restrict__ phi_s, const float* __restrict__ phi_t,
                                     const float* __restrict__ m_s, const float* __restrict__ m_t,
                                     float* __restrict__ phi, float* __restrict__ mask_s, float* __restrict__ mask_t,
                                     int B, int C, int K, int HW) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * HW) return;
    const long long b = t / HW;
    const int pix = (int)(t % HW);
    if (MODE == 3) {
        float vs[KM], vt[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k)
            if (k < K) { vs[k] = m_s[(b * K + k) * HW + pix]; vt[k] = m_t[(b * K + k) * HW + pix]; }
        for (int c = 0; c < C; ++c) {
            const long long o = (b * C + c) * HW + pix;
            float p = phi_s[o];
            const float pt = phi_t[o];
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) p = p * (1.f - vs[k]) * (1.f - vt[k]) + pt * vt[k];
            phi[o] = p;
        }
        mask_s[t] = m_s[(b * K + K - 1) * HW + pix];
        mask_t[t] = m_t[(b * K + K - 1) * HW + pix];
        return;
    }
    float ms, mt;
    if (MODE == 1) {
        float ss = 0.f, st = 0.f;
        for (int k = 0; k < K; ++k) { ss += m_s[(b * K + k) * HW + pix]; st += m_t[(b * K + k) * HW + pix]; }
        ms = fminf(fmaxf(ss, 0.f), 1.f);
        mt = fminf(fmaxf(st, 0.f), 1.f);
    } else {
        ms = m_s[b * HW + pix];
        mt = m_t[b * HW + pix];
    }
    mask_s[t] = ms;
    mask_t[t] = mt;
    for (int c = 0; c < C; ++c) {
        const long long o = (b * C + c) * HW + pix;
        phi[o] = phi_s[o] * (1.f - ms) * (1.f - mt) + phi_t[o] * mt;
    }
}

template <int MODE, int KM = LOOP_MAXK>
__global__ void transport_mode_bwd_k(const float* __restrict__ g, const float* __restrict__ phi_s,
                                     const float* __restrict__ phi_t, const float* __restrict__ m_s,
                                     const float* __restrict__ m_t, float* __restrict__ d_phi_s,
                                     float* __restrict__ d_phi_t, float* __restrict__ d_m_s, float* __restrict__ d_m_t,
                                     int B, int C, int K, int HW) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)B * HW) return;
    const long long b = t / HW;
    const int pix = (int)(t % HW);
    if (MODE == 3) {
        float vs[KM], vt[KM], gs[KM], gt[KM], pi[KM];
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            if (k < K) {
                vs[k] = m_s[(b * K + k) * HW + pix];
                vt[k] = m_t[(b * K + k) * HW + pix];
            }
            gs[k] = gt[k] = 0.f;
        }
        for (int c = 0; c < C; ++c) {
            const long long o = (b * C + c) * HW + pix;
            const float pt = phi_t[o];
            float p = phi_s[o];
#pragma unroll
            for (int k = 0; k < KM; ++k) {   // forward again, keeping every phi_i
                if (k < K) {
                    pi[k] = p;
                    p = p * (1.f - vs[k]) * (1.f - vt[k]) + pt * vt[k];
                }
            }
            float gp = g[o], gpt = 0.f;
#pragma unroll
            for (int k = KM - 1; k >= 0; --k) {
                if (k < K) {
                    gpt += gp * vt[k];
                    gt[k] += gp * (pt - pi[k] * (1.f - vs[k]));
                    gs[k] -= gp * pi[k] * (1.f - vt[k]);
                    gp = gp * (1.f - vs[k]) * (1.f - vt[k]);
                }
            }
            if (d_phi_s) d_phi_s[o] = gp;
            if (d_phi_t) d_phi_t[o] = gpt;
        }
#pragma unroll
        for (int k = 0; k < KM; ++k) {
            if (k < K) {
                if (d_m_s) d_m_s[(b * K + k) * HW + pix] = gs[k];
                if (d_m_t) d_m_t[(b * K + k) * HW + pix] = gt[k];
            }
        }
        return;
    }
    float ms, mt, ss = 0.f, st = 0.f;
    if (MODE == 1) {
        for (int k = 0; k < K; ++k) { ss += m_s[(b * K + k) * HW + pix]; st += m_t[(b * K + k) * HW + pix]; }
        ms = fminf(fmaxf(ss, 0.f), 1.f);
        mt = fminf(fmaxf(st, 0.f), 1.f);
    } else {
        ms = m_s[b * HW + pix];
        mt = m_t[b * HW + pix];
    }
    float gms = 0.f, gmt = 0.f;
    for (int c = 0; c < C; ++c) {
        const long long o = (b * C + c) * HW + pix;
        const float gv = g[o], ps = phi_s[o], pt = phi_t[o];
        gmt += gv * (pt - ps * (1.f - ms));
        gms -= gv * ps * (1.f - mt);
        if (d_phi_t) d_phi_t[o] = gv * mt;
        if (d_phi_s) d_phi_s[o] = gv * (1.f - ms) * (1.f - mt);
    }
    if (MODE == 1) {   // torch.clamp passes the gradient where min <= x <= max
        const float ps_ = (ss >= 0.f && ss <= 1.f) ? gms : 0.f, pt_ = (st >= 0.f && st <= 1.f) ? gmt : 0.f;
        for (int k = 0; k < K; ++k) {
            if (d_m_s) d_m_s[(b * K + k) * HW + pix] = ps_;
            if (d_m_t) d_m_t[(b * K + k) * HW + pix] = pt_;
        }
    } else {
        if (d_m_s) d_m_s[b * HW + pix] = gms;
        if (d_m_t) d_m_t[b * HW + pix] = gmt;
    }
}

}  // namespace

extern "C" {

int kp_spatial_softmax_fwd(const float* heat, float* k, float* p, float* q, int BK, int H, int W, int log_domain,
                           void* stream) {
    KP_REQUIRE(BK > 0 && H > 0 && W > 0, "kp_spatial_softmax_fwd: bad dims");
    const int pitch = W | 1;
    int rc = CHUNK_FLOATS / pitch;
    if (rc < 1) rc = 1;
    if (rc > H) rc = H;
    const size_t smem = ((size_t)rc * pitch + H + W) * sizeof(float);
    KP_REQUIRE(smem <= 150 * 1024, "kp_spatial_softmax_fwd: heat-map too large for the LDS staging tile");
    spatial_softmax_fwd_k<<<BK, TPB, smem, (hipStream_t)stream>>>(heat, k, p, q, H, W, rc, pitch, log_domain);
    return kp_check_launch("spatial_softmax_fwd");
}

int kp_spatial_softmax_bwd(const float* gk, const float* k, const float* p, const float* q, float* dheat, int BK,
                           int H, int W, void* stream) {
    KP_REQUIRE(BK > 0 && H > 0 && W > 0, "kp_spatial_softmax_bwd: bad dims");
    spatial_softmax_bwd_k<<<cdiv((long long)BK * H * W, TPB), TPB, 0, (hipStream_t)stream>>>(gk, k, p, q, dheat, BK, H, W);
    return kp_check_launch("spatial_softmax_bwd");
}

int kp_gaussian_fwd(const float* k, float* m, int BK, int H, int W, float sigma, float eps, void* stream) {
    KP_REQUIRE(BK > 0 && H > 0 && W > 0 && sigma > 0.f, "kp_gaussian_fwd: bad args");
    const float inv2s2 = (float)(1.0 / (2.0 * (double)sigma * (double)sigma));
    render_fwd_k<0><<<cdiv((long long)BK * H * W, TPB), TPB, 0, (hipStream_t)stream>>>(k, m, BK, H, W, inv2s2, eps);
    return kp_check_launch("gaussian_fwd");
}

int kp_gaussian_bwd(const float* gm, const float* k, float* dk, int BK, int H, int W, float sigma, float eps,
                    void* stream) {
    KP_REQUIRE(BK > 0 && H > 0 && W > 0 && sigma > 0.f, "kp_gaussian_bwd: bad args");
    const float inv2s2 = (float)(1.0 / (2.0 * (double)sigma * (double)sigma));
    gaussian_bwd_k<<<BK, TPB, 0, (hipStream_t)stream>>>(gm, k, dk, H, W, inv2s2, eps);
    return kp_check_launch("gaussian_bwd");
}

int kp_point_map_fwd(const float* k, float* m, int BK, int H, int W, void* stream) {
    KP_REQUIRE(BK > 0 && H > 0 && W > 0, "kp_point_map_fwd: bad dims");
    render_fwd_k<1><<<cdiv((long long)BK * H * W, TPB), TPB, 0, (hipStream_t)stream>>>(k, m, BK, H, W, 0.f, 0.f);
    return kp_check_launch("point_map_fwd");
}

int kp_transport_max_fwd(const float* phi_s, const float* phi_t, const float* m_s, const float* m_t, float* phi,
                         float* mask_s, float* mask_t, unsigned char* arg_s, unsigned char* arg_t, int B, int C, int K,
                         int H, int W, void* stream) {
    KP_REQUIRE(B > 0 && C > 0 && K > 0 && K <= 255 && H > 0 && W > 0, "kp_transport_max_fwd: bad dims");
    transport_max_fwd_k<<<cdiv((long long)B * H * W, TPB), TPB, 0, (hipStream_t)stream>>>(phi_s, phi_t, m_s, m_t, phi, mask_s, mask_t, arg_s, arg_t, B, C, K, H * W);
    return kp_check_launch("transport_max_fwd");
}

int kp_transport_max_bwd(const float* gphi, const float* phi_s, const float* phi_t, const float* mask_s,
                         const float* mask_t, const unsigned char* arg_s, const unsigned char* arg_t, float* d_phi_s,
                         float* d_phi_t, float* d_m_s, float* d_m_t, int B, int C, int K, int H, int W, void* stream) {
    KP_REQUIRE(B > 0 && C > 0 && K > 0 && K <= 255 && H > 0 && W > 0, "kp_transport_max_bwd: bad dims");
    transport_max_bwd_k<<<cdiv((long long)B * H * W, TPB), TPB, 0, (hipStream_t)stream>>>(gphi, phi_s, phi_t, mask_s, mask_t, arg_s, arg_t, d_phi_s, d_phi_t, d_m_s, d_m_t, B, C, K, H * W);
    return kp_check_launch("transport_max_bwd");
}

int kp_transport_fwd(int mode, const float* phi_s, const float* phi_t, const float* m_s, const float* m_t, float* phi,
                     float* mask_s, float* mask_t, int B, int C, int K, int H, int W, void* stream) {
    KP_REQUIRE(phi_s && phi_t && m_s && m_t && phi && mask_s && mask_t, "kp_transport_fwd: null pointer");
    KP_REQUIRE(B > 0 && C > 0 && K > 0 && H > 0 && W > 0, "kp_transport_fwd: bad dims");
    KP_REQUIRE(mode != KP_COMBINE_PRETRAINED || K == 1, "kp_transport_fwd: 'pretrained_network' takes one-channel masks");
    const unsigned grid = cdiv((long long)B * H * W, TPB);
    // (the backward keeps per-keypoint arrays in registers: refuse here what could not be differentiated, instead of failing
    //  inside backward() after the forward succeeded)
    KP_REQUIRE(mode != KP_COMBINE_LOOP || K <= LOOP_MAXK, "kp_transport_fwd: 'loop' supports up to %d keypoints", LOOP_MAXK);
    hipStream_t st = (hipStream_t)stream;
    if (mode == KP_COMBINE_SUM_AND_CLAMP) transport_mode_fwd_k<1><<<grid, TPB, 0, st>>>(phi_s, phi_t, m_s, m_t, phi, mask_s, mask_t, B, C, K, H * W);
    else if (mode == KP_COMBINE_PRETRAINED) transport_mode_fwd_k<2><<<grid, TPB, 0, st>>>(phi_s, phi_t, m_s, m_t, phi, mask_s, mask_t, B, C, K, H * W);
    else if (mode == KP_COMBINE_LOOP && K <= 4) transport_mode_fwd_k<3, 4><<<grid, TPB, 0, st>>>(phi_s, phi_t, m_s, m_t, phi, mask_s, mask_t, B, C, K, H * W);
    else if (mode == KP_COMBINE_LOOP && K <= 16) transport_mode_fwd_k<3, 16><<<grid, TPB, 0, st>>>(phi_s, phi_t, m_s, m_t, phi, mask_s, mask_t, B, C, K, H * W);
    else if (mode == KP_COMBINE_LOOP) transport_mode_fwd_k<3><<<grid, TPB, 0, st>>>(phi_s, phi_t, m_s, m_t, phi, mask_s, mask_t, B, C, K, H * W);
    else { kp_set_error("kp_transport_fwd: unknown mode %d (use kp_transport_max_fwd for 'max')", mode); return KP_E_ARG; }
    return kp_check_launch("transport_fwd");
}

int kp_transport_bwd(int mode, const float* gphi, const float* phi_s, const float* phi_t, const float* m_s,
                     const float* m_t, float* d_phi_s, float* d_phi_t, float* d_m_s, float* d_m_t, int B, int C, int K,
                     int H, int W, void* stream) {
    KP_REQUIRE(gphi && phi_s && phi_t && m_s && m_t, "kp_transport_bwd: null pointer");
    KP_REQUIRE(B > 0 && C > 0 && K > 0 && H > 0 && W > 0, "kp_transport_bwd: bad dims");
    KP_REQUIRE(mode != KP_COMBINE_PRETRAINED || K == 1, "kp_transport_bwd: 'pretrained_network' takes one-channel masks");
    KP_REQUIRE(mode != KP_COMBINE_LOOP || K <= LOOP_MAXK, "kp_transport_bwd: 'loop' supports up to %d keypoints", LOOP_MAXK);
    const unsigned grid = cdiv((long long)B * H * W, TPB);
    hipStream_t st = (hipStream_t)stream;
    if (mode == KP_COMBINE_SUM_AND_CLAMP) transport_mode_bwd_k<1><<<grid, TPB, 0, st>>>(gphi, phi_s, phi_t, m_s, m_t, d_phi_s, d_phi_t, d_m_s, d_m_t, B, C, K, H * W);
    else if (mode == KP_COMBINE_PRETRAINED) transport_mode_bwd_k<2><<<grid, TPB, 0, st>>>(gphi, phi_s, phi_t, m_s, m_t, d_phi_s, d_phi_t, d_m_s, d_m_t, B, C, K, H * W);
    else if (mode == KP_COMBINE_LOOP && K <= 4) transport_mode_bwd_k<3, 4><<<grid, TPB, 0, st>>>(gphi, phi_s, phi_t, m_s, m_t, d_phi_s, d_phi_t, d_m_s, d_m_t, B, C, K, H * W);
    else if (mode == KP_COMBINE_LOOP && K <= 16) transport_mode_bwd_k<3, 16><<<grid, TPB, 0, st>>>(gphi, phi_s, phi_t, m_s, m_t, d_phi_s, d_phi_t, d_m_s, d_m_t, B, C, K, H * W);
    else if (mode == KP_COMBINE_LOOP) transport_mode_bwd_k<3><<<grid, TPB, 0, st>>>(gphi, phi_s, phi_t, m_s, m_t, d_phi_s, d_phi_t, d_m_s, d_m_t, B, C, K, H * W);
    else { kp_set_error("kp_transport_bwd: unknown mode %d (use kp_transport_max_bwd for 'max')", mode); return KP_E_ARG; }
    return kp_check_launch("transport_bwd");
}

}  // extern "C"
