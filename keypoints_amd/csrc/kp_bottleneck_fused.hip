// The Transporter keypoint bottleneck as ONE launch per direction (keypoint_bottleneck_{fwd,bwd}).
//
// Reference statements replaced (combine_method='max', the default of transporter.make):
//   TransporterNet.extract       keypoints/models/transporter.py:27-32   (ssm + key2map on both branches)
//   spacial_logsoftmax           keypoints/models/functional.py:11-14,22-24,37-44
//   gaussian_like_function       keypoints/models/functional.py:49-63
//   the 'max' blend              keypoints/models/transporter.py:57-60
//
// One workgroup of 16 wavefronts per image pair.  Forward, phase A: the 2K heat-maps of the pair are spread over the
// wavefronts; a wavefront stages its map through a private LDS tile in row chunks (coalesced 16-byte reads over H x W),
// reduces both marginals, runs the two 1-D log-softmaxes with wave reductions and leaves the keypoint in LDS.  Phase B
// (registers only): a thread owns 4 consecutive pixels and a channel group; it renders the 2K Gaussians of its pixels
// from the LDS keypoints, takes both maxima, and streams its channels of phi_s / phi_t through the blend with
// 16-byte accesses.  The source branch's maps are never written to HBM (nothing downstream reads them), the target
// branch's are (TransporterNet.forward returns m_xt).  Every sum has a fixed order: results are reproducible, and
// the forward is bit-identical to the piecewise kernels of kp_bottleneck.hip (same expressions, same summation order).
//
// Backward (one workgroup per pair): the blend's adjoint reduces over channels per pixel (per channel-group partials
// in LDS, summed in a fixed order), routes that to argmax_k, folds it with the Gaussian's keypoint derivative (one
// wavefront per keypoint, wave reduction over pixels) and writes the heat-map gradient of the marginal log-softmax.
// The map gradient itself ([B,K,H,W], zero off the argmax) never exists in memory.
#include <mutex>
#include "kp_bottleneck_dev.h"

namespace {

template <int V>
__global__ __launch_bounds__(NT) void bottleneck_fwd_k(const float* __restrict__ heat_s, const float* __restrict__ heat_t,
                                                       const float* __restrict__ phi_s, const float* __restrict__ phi_t,
                                                       float* __restrict__ k_t, float* __restrict__ p_t,
                                                       float* __restrict__ q_t, float* __restrict__ m_t,
                                                       float* __restrict__ phi, float* __restrict__ mask_s,
                                                       float* __restrict__ mask_t, unsigned char* __restrict__ arg_t,
                                                       int C, int K, int H, int W, float inv2s2, float eps) {
    extern __shared__ float sm[];
    float* keys = sm;                     // [2][K][2]: branch, keypoint, (y, x)
    float* wbase = sm + 4 * K;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const int HW = H * W;

    // ---- phase A: 2K heat-maps -> keypoints ------------------------------------------------------------------
    {
        float* tile = wbase + wave * WAVE_FLOATS;
        float* rowv = tile + TILE_FLOATS;
        float* colv = rowv + MAXDIM;
        for (int mp = wave; mp < 2 * K; mp += NW) {
            const int br = mp / K, k = mp - br * K;
            const long long bk = (long long)b * K + k;
            const float* src = (br ? heat_t : heat_s) + bk * HW;
            float ky, kx;
            wave_keypoint(src, H, W, tile, rowv, colv, br ? p_t + bk * H : nullptr, br ? q_t + bk * W : nullptr, ky, kx);
            if ((tid & 63) == 0) {
                keys[mp * 2] = ky;
                keys[mp * 2 + 1] = kx;
                if (br) { k_t[bk * 2] = ky; k_t[bk * 2 + 1] = kx; }
            }
        }
    }
    __syncthreads();

    // ---- phase B: render, max over K, blend -----------------------------------------------------------------
    const int nq = HW / V;
    int G = NT / nq;
    if (G < 1) G = 1;
    if (G > C) G = C;
    const float* ks = keys;
    const float* kt = keys + 2 * K;
    for (int item = tid; item < nq * G; item += NT) {
        const int cg = item / nq, quad = item - cg * nq;
        const int pix0 = quad * V;
        float ry[V], rx[V], bs[V], bt[V];
        int at[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int pix = pix0 + v;
            const int i = pix / W;
            ry[v] = ruler_at(i, H);
            rx[v] = ruler_at(pix - i * W, W);
            bs[v] = bt[v] = 0.f;
            at[v] = 0;
        }
        for (int k = 0; k < K; ++k) {
            const float sy = ks[2 * k], sx = ks[2 * k + 1], ty = kt[2 * k], tx = kt[2 * k + 1];
            float mt[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float dy = ry[v] - sy, dx = rx[v] - sx;
                const float vs = kp_gauss(dy * dy + dx * dx + eps, inv2s2);
                dy = ry[v] - ty; dx = rx[v] - tx;
                mt[v] = kp_gauss(dy * dy + dx * dx + eps, inv2s2);
                if (k == 0 || vs > bs[v]) bs[v] = vs;
                if (k == 0 || mt[v] > bt[v]) { bt[v] = mt[v]; at[v] = k; }   // first maximum wins, like torch.max
            }
            if (k % G == cg) Vec<V>::st(m_t + ((long long)b * K + k) * HW + pix0, mt);
        }
        if (cg == 0) {
            Vec<V>::st(mask_s + (long long)b * HW + pix0, bs);
            Vec<V>::st(mask_t + (long long)b * HW + pix0, bt);
            Vec<V>::st8(arg_t + (long long)b * HW + pix0, at);
        }
        const float* ps = phi_s + (long long)b * C * HW + pix0;
        const float* pt = phi_t + (long long)b * C * HW + pix0;
        float* po = phi + (long long)b * C * HW + pix0;
        int c = cg;
        for (; c + 3 * G < C; c += 4 * G) {   // four channels in flight
            float a[4][V], t[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                Vec<V>::ld(ps + (long long)(c + u * G) * HW, a[u]);
                Vec<V>::ld(pt + (long long)(c + u * G) * HW, t[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int v = 0; v < V; ++v) a[u][v] = a[u][v] * (1.f - bs[v]) * (1.f - bt[v]) + t[u][v] * bt[v];
                Vec<V>::st(po + (long long)(c + u * G) * HW, a[u]);
            }
        }
        for (; c < C; c += G) {
            float a[V], t[V];
            Vec<V>::ld(ps + (long long)c * HW, a);
            Vec<V>::ld(pt + (long long)c * HW, t);
#pragma unroll
            for (int v = 0; v < V; ++v) a[v] = a[v] * (1.f - bs[v]) * (1.f - bt[v]) + t[v] * bt[v];
            Vec<V>::st(po + (long long)c * HW, a);
        }
    }
}

template <int V>
__global__ __launch_bounds__(NT) void bottleneck_bwd_k(const float* __restrict__ gphi, const float* __restrict__ g_k,
                                                       const float* __restrict__ g_m, const float* __restrict__ phi_s,
                                                       const float* __restrict__ phi_t, const float* __restrict__ mask_s,
                                                       const float* __restrict__ mask_t,
                                                       const unsigned char* __restrict__ arg_t,
                                                       const float* __restrict__ k_t, const float* __restrict__ p_t,
                                                       const float* __restrict__ q_t, float* __restrict__ d_phi_t,
                                                       float* __restrict__ d_heat, int C, int K, int H, int W, int G,
                                                       float inv2s2, float eps) {
    extern __shared__ float sm[];
    float* gpart = sm;                      // [G][HW] -> reduced into row 0
    float* dkv = sm + BWD_LDS_FLOATS;       // [K][2]
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int HW = H * W;
    const int nq = HW / V;

    // ---- 1: adjoint of the blend: d_phi_t = g * Mt ; gM[pix] = sum_c g_c (phi_t,c - phi_s,c (1 - Ms)) --------------
    for (int item = tid; item < nq * G; item += NT) {
        const int cg = item / nq, quad = item - cg * nq;
        const int pix0 = quad * V;
        float ms[V], mt[V], acc[V];
        Vec<V>::ld(mask_s + (long long)b * HW + pix0, ms);
        Vec<V>::ld(mask_t + (long long)b * HW + pix0, mt);
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = 0.f;
        const long long base = (long long)b * C * HW + pix0;
        int c = cg;
        for (; c + G < C; c += 2 * G) {   // two channels in flight (3 loads each)
            float g0[V], s0[V], t0[V], g1[V], s1[V], t1[V];
            Vec<V>::ld(gphi + base + (long long)c * HW, g0);
            Vec<V>::ld(phi_s + base + (long long)c * HW, s0);
            Vec<V>::ld(phi_t + base + (long long)c * HW, t0);
            Vec<V>::ld(gphi + base + (long long)(c + G) * HW, g1);
            Vec<V>::ld(phi_s + base + (long long)(c + G) * HW, s1);
            Vec<V>::ld(phi_t + base + (long long)(c + G) * HW, t1);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                acc[v] += g0[v] * (t0[v] - s0[v] * (1.f - ms[v]));
                acc[v] += g1[v] * (t1[v] - s1[v] * (1.f - ms[v]));
                g0[v] = g0[v] * mt[v];
                g1[v] = g1[v] * mt[v];
            }
            Vec<V>::st(d_phi_t + base + (long long)c * HW, g0);
            Vec<V>::st(d_phi_t + base + (long long)(c + G) * HW, g1);
        }
        for (; c < C; c += G) {
            float g0[V], s0[V], t0[V];
            Vec<V>::ld(gphi + base + (long long)c * HW, g0);
            Vec<V>::ld(phi_s + base + (long long)c * HW, s0);
            Vec<V>::ld(phi_t + base + (long long)c * HW, t0);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                acc[v] += g0[v] * (t0[v] - s0[v] * (1.f - ms[v]));
                g0[v] = g0[v] * mt[v];
            }
            Vec<V>::st(d_phi_t + base + (long long)c * HW, g0);
        }
#pragma unroll
        for (int v = 0; v < V; ++v) gpart[cg * HW + pix0 + v] = acc[v];
    }
    __syncthreads();
    if (G > 1) {
        for (int pix = tid; pix < HW; pix += NT) {
            float a = gpart[pix];
            for (int cg = 1; cg < G; ++cg) a += gpart[cg * HW + pix];
            gpart[pix] = a;
        }
        __syncthreads();
    }

    // ---- 2: through argmax_k and the Gaussian render onto the keypoints (one wavefront per keypoint) --------------
    for (int k = wave; k < K; k += NW) {
        const long long bk = (long long)b * K + k;
        const float ky = k_t[bk * 2], kx = k_t[bk * 2 + 1];
        float sy = 0.f, sx = 0.f;
        for (int pix = lane; pix < HW; pix += 64) {
            float gm = (arg_t[(long long)b * HW + pix] == k) ? gpart[pix] : 0.f;
            if (g_m) gm += g_m[bk * HW + pix];
            if (gm != 0.f) {
                const int i = pix / W;
                const float dy = ruler_at(i, H) - ky, dx = ruler_at(pix - i * W, W) - kx;
                const float d = kp_fast_sqrt(dy * dy + dx * dx + eps);
                const float cc = gm * kp_fast_exp(-d * inv2s2) * inv2s2 / d;
                sy += cc * dy;
                sx += cc * dx;
            }
        }
        sy = wave_sum(sy);
        sx = wave_sum(sx);
        if (lane == 0) {
            dkv[2 * k] = sy + (g_k ? g_k[bk * 2] : 0.f);
            dkv[2 * k + 1] = sx + (g_k ? g_k[bk * 2 + 1] : 0.f);
        }
    }
    __syncthreads();

    // ---- 3: marginal log-softmax backward: d_heat = p_i (r_i - ky) gy / W + q_j (c_j - kx) gx / H ------------------
    for (int item = tid; item < K * nq; item += NT) {
        const int k = item / nq, quad = item - k * nq;
        const int pix0 = quad * V;
        const long long bk = (long long)b * K + k;
        const float ky = k_t[bk * 2], kx = k_t[bk * 2 + 1];
        const float gy = dkv[2 * k], gx = dkv[2 * k + 1];
        float o[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const int pix = pix0 + v;
            const int i = pix / W, j = pix - i * W;
            o[v] = p_t[bk * H + i] * (ruler_at(i, H) - ky) * gy / (float)W + q_t[bk * W + j] * (ruler_at(j, W) - kx) * gx / (float)H;
        }
        Vec<V>::st(d_heat + bk * HW + pix0, o);
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// The four kernels take more than the default 64 KiB of dynamic LDS.  The attribute is per DEVICE (a code object is loaded per
// device) and must be in place whichever entry point a process calls first -- a backward without a forward in this process,
// or a first forward on another GPU of a multi-GPU process -- so BOTH entry points call this, once per device (call_once per
// device id; 64 devices is more than a node holds).
static int set_lds_attributes() {
    static std::once_flag once[64];
    static int status[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    std::call_once(once[dev], [dev] {
        const void* fns[] = {reinterpret_cast<const void*>(bottleneck_fwd_k<4>), reinterpret_cast<const void*>(bottleneck_fwd_k<1>),
                             reinterpret_cast<const void*>(bottleneck_bwd_k<4>), reinterpret_cast<const void*>(bottleneck_bwd_k<1>)};
        status[dev] = 0;
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) status[dev] = 1;
    });
    if (status[dev]) {
        kp_set_error("keypoint bottleneck: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed on device %d", dev);
        return KP_E_LAUNCH;
    }
    return KP_OK;
}

int kp_bottleneck_fused_fwd(const float* heat_s, const float* heat_t, const float* phi_s, const float* phi_t, float* k_t,
                            float* p_t, float* q_t, float* m_t, float* phi, float* mask_s, float* mask_t,
                            unsigned char* arg_t, int B, int C, int K, int H, int W, float sigma, float eps,
                            hipStream_t stream) {
    KP_REQUIRE(heat_s && heat_t && phi_s && phi_t && k_t && p_t && q_t && m_t && phi && mask_s && mask_t && arg_t,
               "kp_keypoint_bottleneck_fwd: null pointer");
    KP_REQUIRE(B > 0 && C > 0 && K > 0 && K <= 255 && H > 0 && W > 0 && sigma > 0.f, "kp_keypoint_bottleneck_fwd: bad dims");
    KP_REQUIRE(H <= MAXDIM && W <= MAXDIM, "kp_keypoint_bottleneck_fwd: heat-maps larger than %d x %d are not supported", MAXDIM, MAXDIM);
    const float inv2s2 = (float)(1.0 / (2.0 * (double)sigma * (double)sigma));
    const size_t smem = (4 * (size_t)K + (size_t)NW * WAVE_FLOATS) * sizeof(float);
    const bool vec = ((H * W) & 3) == 0 && aligned16(phi_s) && aligned16(phi_t) && aligned16(phi) && aligned16(m_t) &&
                     aligned16(mask_s) && aligned16(mask_t) && ((reinterpret_cast<uintptr_t>(arg_t) & 3) == 0);
    if (int rc = set_lds_attributes()) return rc;
    if (vec)
        bottleneck_fwd_k<4><<<B, NT, smem, stream>>>(heat_s, heat_t, phi_s, phi_t, k_t, p_t, q_t, m_t, phi, mask_s, mask_t, arg_t, C, K, H, W, inv2s2, eps);
    else
        bottleneck_fwd_k<1><<<B, NT, smem, stream>>>(heat_s, heat_t, phi_s, phi_t, k_t, p_t, q_t, m_t, phi, mask_s, mask_t, arg_t, C, K, H, W, inv2s2, eps);
    return kp_check_launch("keypoint_bottleneck_fwd");
}

int kp_bottleneck_fused_bwd(const float* gphi, const float* g_k, const float* g_m, const float* phi_s, const float* phi_t,
                            const float* mask_s, const float* mask_t, const unsigned char* arg_t, const float* k_t,
                            const float* p_t, const float* q_t, float* d_phi_t, float* d_heat, int B, int C, int K, int H,
                            int W, float sigma, float eps, hipStream_t stream) {
    KP_REQUIRE(gphi && phi_s && phi_t && mask_s && mask_t && arg_t && k_t && p_t && q_t && d_phi_t && d_heat,
               "kp_keypoint_bottleneck_bwd: null pointer");
    KP_REQUIRE(B > 0 && C > 0 && K > 0 && K <= 255 && H > 0 && W > 0 && sigma > 0.f, "kp_keypoint_bottleneck_bwd: bad dims");
    const float inv2s2 = (float)(1.0 / (2.0 * (double)sigma * (double)sigma));
    const int HW = H * W;
    const bool vec = (HW & 3) == 0 && aligned16(gphi) && aligned16(phi_s) && aligned16(phi_t) && aligned16(d_phi_t) &&
                     aligned16(d_heat) && aligned16(mask_s) && aligned16(mask_t);
    const int nq = HW / (vec ? 4 : 1);
    int G = NT / nq;
    if (G < 1) G = 1;
    if (G > C) G = C;
    KP_REQUIRE((long long)G * HW <= BWD_LDS_FLOATS, "kp_keypoint_bottleneck_bwd: %d x %d maps do not fit the LDS reduction buffer", H, W);
    const size_t smem = ((size_t)BWD_LDS_FLOATS + 2 * (size_t)K) * sizeof(float);
    if (int rc = set_lds_attributes()) return rc;
    if (vec)
        bottleneck_bwd_k<4><<<B, NT, smem, stream>>>(gphi, g_k, g_m, phi_s, phi_t, mask_s, mask_t, arg_t, k_t, p_t, q_t, d_phi_t, d_heat, C, K, H, W, G, inv2s2, eps);
    else
        bottleneck_bwd_k<1><<<B, NT, smem, stream>>>(gphi, g_k, g_m, phi_s, phi_t, mask_s, mask_t, arg_t, k_t, p_t, q_t, d_phi_t, d_heat, C, K, H, W, G, inv2s2, eps);
    return kp_check_launch("keypoint_bottleneck_bwd");
}
