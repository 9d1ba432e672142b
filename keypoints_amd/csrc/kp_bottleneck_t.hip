// The fused keypoint bottleneck (kp_bottleneck_fused.hip) on the 16-bit NHWC tensors of its neighbours -- gfx950.
//
// Reference statements replaced: the same as kp_bottleneck_fused.hip (keypoints/models/transporter.py:27-32,57-60,
// functional.py:11-14,22-24,37-63, combine_method='max').  What differs is the LAYOUT of the feature maps: the reference (and
// kp_keypoint_bottleneck_fwd) takes phi_s / phi_t as fp32 NCHW, which for the Pong-class nets (vgg.py:53,66: 16 feature channels
// over 84 x 84) means the encoder head writes 4 bytes per element in planes, this kernel reads them, writes the blend as fp32
// planes, and a layout kernel turns that into the decoder's padded 16-bit NHWC input -- and the same chain backwards.  Here
//   phi_s / phi_t  are the encoder head's output as it leaves the matrix cores: T [B][H][W][C]   (kp_head_fwd_t),
//   the blend      is written as the decoder's first-layer input:             T [B][H+2][W+2][C] with its replicate border,
//   the gradient   arrives as the decoder's data gradient over that padded domain (folded on the load path: the adjoint of
//                  the replicate pad) and leaves as T [B][H][W][C] for kp_head_bwd_t.
// Half the bytes per element at both ends and no layout kernel in between.  The heat-maps stay fp32 NCHW (marginals are
// row / column sums), phase A (heat-maps -> keypoints) is kp_bottleneck_dev.h's code, shared with the fp32 form.
//
// Phase B: a thread owns one 16-byte chunk (8 consecutive channels) of one pixel; the LP = C / 8 lanes of a pixel are adjacent
// lanes of a wavefront.  With LP >= 2 the two branches' Gaussian renders are split over lane parity (even lanes: source maps,
// odd lanes: target maps + argmax + the m_t planes) and exchanged with one DPP-class shuffle, so a pixel's 2K renders are issued
// once per lane pair instead of once per lane.  Same expressions, same order of operations as the fp32 form: masks, argmax,
// keypoints, marginals and m_t are bit-identical to it, the blend is its value rounded once to T.
#include <mutex>
#include "kp_bottleneck_dev.h"

namespace {

__device__ __forceinline__ void unpack8(const uint4 v, float (&o)[8]) {
    unpack_h16x2(v.x, o[0], o[1]);
    unpack_h16x2(v.y, o[2], o[3]);
    unpack_h16x2(v.z, o[4], o[5]);
    unpack_h16x2(v.w, o[6], o[7]);
}
__device__ __forceinline__ uint4 pack8(const float (&o)[8]) {
    return make_uint4(pack_h16x2(o[0], o[1]), pack_h16x2(o[2], o[3]), pack_h16x2(o[4], o[5]), pack_h16x2(o[6], o[7]));
}

template <int LP>
__global__ __launch_bounds__(NT) void bottleneck_fwd_t_k(const float* __restrict__ heat_s, const float* __restrict__ heat_t,
                                                         const h16_t* __restrict__ phi_s, const h16_t* __restrict__ phi_t,
                                                         float* __restrict__ k_t, float* __restrict__ p_t,
                                                         float* __restrict__ q_t, float* __restrict__ m_t,
                                                         h16_t* __restrict__ a, float* __restrict__ mask_s,
                                                         float* __restrict__ mask_t, unsigned char* __restrict__ arg_t,
                                                         int K, int H, int W, float inv2s2, float eps, int dbg) {
    constexpr int C = 8 * LP;
    extern __shared__ float sm[];
    float* keys = sm;                     // [2][K][2]: branch, keypoint, (y, x)
    float* wbase = sm + 4 * K;
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
    const int HW = H * W, Wp = W + 2;

    // ---- phase A: 2K heat-maps -> keypoints (shared with the fp32 form) -------------------------------------------
    // (dbg: kp_config(KP_CFG_DEBUG) ablation mask of tools/probes/bneck_t_abl.py -- 1: no phase A, 2: no phase B; 0 in normal operation)
    if (dbg & 1) {
        if (tid < 4 * K) keys[tid] = 0.5f;
    } else {
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
    if (dbg & 2) return;

    // ---- phase B: render, max over K, blend: one (pixel, 8-channel chunk) per thread ------------------------------
    const uint4* ps = reinterpret_cast<const uint4*>(phi_s + (long long)b * HW * C);
    const uint4* pt = reinterpret_cast<const uint4*>(phi_t + (long long)b * HW * C);
    uint4* pa = reinterpret_cast<uint4*>(a + (long long)b * (H + 2) * Wp * C);
    const float* ks = keys;
    const float* kt = keys + 2 * K;
    // (one workgroup per CU, 16 waves: a lane keeps the chunks of its NEXT TWO items in flight under the renders of this one --
    //  with one pair of 16-byte loads per lane the CU had 32 KB in flight, half of what its share of the HBM rate needs)
    const int total = HW * LP;
    uint4 ns[2], nt[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int it = tid + u * NT < total ? tid + u * NT : (tid < total ? tid : 0);   // (small maps: HW * LP < NT -- never past the image)
        ns[u] = ps[it]; nt[u] = pt[it];
    }
    for (int item = tid; item < total; item += 2 * NT) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int itm = item + u * NT;
        if (itm >= total) break;
        const uint4 rs = ns[u], rt = nt[u];
        {
            const int nx = itm + 2 * NT < total ? itm + 2 * NT : itm;
            ns[u] = ps[nx]; nt[u] = pt[nx];
        }
        const int pix = itm / LP, ch = itm - pix * LP;
        const int i = pix / W, j = pix - i * W;
        const float ry = ruler_at(i, H), rx = ruler_at(j, W);
        const bool do_s = LP == 1 || (ch & 1) == 0, do_t = LP == 1 || (ch & 1) == 1;
        float bs = 0.f, bt = 0.f;
        int at = 0;
        for (int k = 0; k < K; ++k) {
            if (do_s) {
                const float dy = ry - ks[2 * k], dx = rx - ks[2 * k + 1];
                const float vs = kp_gauss(dy * dy + dx * dx + eps, inv2s2);
                if (k == 0 || vs > bs) bs = vs;
            }
            if (do_t) {
                const float dy = ry - kt[2 * k], dx = rx - kt[2 * k + 1];
                const float mt = kp_gauss(dy * dy + dx * dx + eps, inv2s2);
                if (k == 0 || mt > bt) { bt = mt; at = k; }   // first maximum wins, like torch.max
                if (ch == (LP == 1 ? 0 : 1)) m_t[((long long)b * K + k) * HW + pix] = mt;
            }
        }
        if (LP > 1) {   // the lane pair (ch ^ 1) rendered the other branch of this pixel
            const float mine = do_s ? bs : bt;
            const float other = __shfl_xor(mine, 1, 64);
            const int oat = __shfl_xor(at, 1, 64);
            if (do_s) { bt = other; at = oat; } else { bs = other; }
        }
        if (ch == 0) {
            mask_s[(long long)b * HW + pix] = bs;
            mask_t[(long long)b * HW + pix] = bt;
            arg_t[(long long)b * HW + pix] = (unsigned char)at;
        }
        float s[8], t[8];
        unpack8(rs, s);
        unpack8(rt, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = s[e] * (1.f - bs) * (1.f - bt) + t[e] * bt;
        const uint4 o = pack8(s);
        // the pixel and the copies of it the replicate border holds
        const bool top = i == 0, bot = i == H - 1, lef = j == 0, rig = j == W - 1;
        auto put = [&](int pr, int pc) { pa[((long long)pr * Wp + pc) * LP + ch] = o; };
        put(i + 1, j + 1);
        if (top) put(0, j + 1);
        if (bot) put(H + 1, j + 1);
        if (lef) put(i + 1, 0);
        if (rig) put(i + 1, W + 1);
        if (top && lef) put(0, 0);
        if (top && rig) put(0, W + 1);
        if (bot && lef) put(H + 1, 0);
        if (bot && rig) put(H + 1, W + 1);
      }
    }
}

template <int LP, int V>
__global__ __launch_bounds__(NT) void bottleneck_bwd_t_k(const h16_t* __restrict__ ga, const float* __restrict__ g_k,
                                                         const float* __restrict__ g_m, const h16_t* __restrict__ phi_s,
                                                         const h16_t* __restrict__ phi_t, const float* __restrict__ mask_s,
                                                         const float* __restrict__ mask_t,
                                                         const unsigned char* __restrict__ arg_t,
                                                         const float* __restrict__ k_t, const float* __restrict__ p_t,
                                                         const float* __restrict__ q_t, h16_t* __restrict__ d_phi_t,
                                                         float* __restrict__ d_heat, int K, int H, int W, float inv2s2,
                                                         float eps, int dbg) {
    constexpr int C = 8 * LP;
    extern __shared__ float sm[];
    float* gpart = sm;                      // [HW]: sum over channels of g (phi_t - phi_s (1 - Ms))
    float* dkv = sm + BWD_LDS_FLOATS;       // [K][2]
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int HW = H * W, Wp = W + 2;

    // ---- 1: adjoint of the replicate pad (fold), then of the blend: d_phi_t = g Mt ; gM[pix] = sum_c g_c (phi_t,c - phi_s,c (1 - Ms))
    const uint4* pg = reinterpret_cast<const uint4*>(ga + (long long)b * (H + 2) * Wp * C);
    const uint4* ps = reinterpret_cast<const uint4*>(phi_s + (long long)b * HW * C);
    const uint4* pt = reinterpret_cast<const uint4*>(phi_t + (long long)b * HW * C);
    uint4* pd = reinterpret_cast<uint4*>(d_phi_t + (long long)b * HW * C);
    const int total = HW * LP;
    // the next item's chunks (phi_s, phi_t, the gradient's centre position) are requested before this item is worked on
    auto centre = [&](int it) { const int px = it / LP, c2 = it - px * LP, ii = px / W, jj = px - ii * W; return ((long long)(ii + 1) * Wp + jj + 1) * LP + c2; };
    uint4 ns = ps[tid < total ? tid : 0], nt = pt[tid < total ? tid : 0], ng = pg[centre(tid < total ? tid : 0)];
    // (dbg: kp_config(KP_CFG_DEBUG) ablation mask of tools/probes/bneck_t_abl.py -- 1: no step 1, 2: no steps 2-3; 0 in normal operation.
    //  Measured at config-2 size, cold: 122 us whole, 88 without steps 2-3, 41 without step 1; the marginals staged through LDS for
    //  step 3 instead of read from global memory in its item loop: no change, not kept)
    for (int item = (dbg & 1) ? total : tid; item < total; item += NT) {
        const int pix = item / LP, ch = item - pix * LP;
        const int i = pix / W, j = pix - i * W;
        const uint4 rs = ns, rt = nt, rg = ng;
        {
            const int nx = item + NT < total ? item + NT : item;
            ns = ps[nx]; nt = pt[nx]; ng = pg[centre(nx)];
        }
        const float ms = mask_s[(long long)b * HW + pix], mt = mask_t[(long long)b * HW + pix];
        // rows / columns of the padded domain that replicate onto (i, j): summed rows outer, columns inner, from 0 -- the order of
        // kp_nhwc_grad_to_nchw, so this kernel == that kernel + kp_keypoint_bottleneck_bwd on its fp32 output
        const int ilo = i == 0 ? 0 : i + 1, ihi = i == H - 1 ? H + 1 : i + 1;
        const int jlo = j == 0 ? 0 : j + 1, jhi = j == W - 1 ? W + 1 : j + 1;
        float g[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) g[e] = 0.f;
        for (int pr = ilo; pr <= ihi; ++pr)
            for (int pc = jlo; pc <= jhi; ++pc) {
                float v[8];
                unpack8((pr == i + 1 && pc == j + 1) ? rg : pg[((long long)pr * Wp + pc) * LP + ch], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) g[e] += v[e];
            }
        float s[8], t[8];
        unpack8(rs, s);
        unpack8(rt, t);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc += g[e] * (t[e] - s[e] * (1.f - ms));
            g[e] = g[e] * mt;
        }
        pd[item] = pack8(g);
#pragma unroll
        for (int o = 1; o < LP; o <<= 1) acc += __shfl_xor(acc, o, 64);   // the pixel's chunks: adjacent lanes, fixed order
        if (ch == 0) gpart[pix] = acc;
    }
    __syncthreads();
    if (dbg & 2) return;

    // ---- 2: through argmax_k and the Gaussian render onto the keypoints: NW / K wavefronts per keypoint (K < NW: with one
    // wavefront per keypoint 4 of the 16 waves of a Pong-class pair walked 7 056 pixels each while 12 idled), every part a
    // strided share of the pixels, the parts' sums added in part order
    {
        const int parts = K >= NW ? 1 : NW / K;
        float* dkp = dkv + 2 * K;                 // [K][parts][2]
        for (int k = wave / parts; k < K; k += NW / parts) {
            const int part = wave - (wave / parts) * parts;
            const long long bk = (long long)b * K + k;
            const float ky = k_t[bk * 2], kx = k_t[bk * 2 + 1];
            float sy = 0.f, sx = 0.f;
            for (int pix = part * 64 + lane; pix < HW; pix += 64 * parts) {
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
                dkp[(k * parts + part) * 2] = sy;
                dkp[(k * parts + part) * 2 + 1] = sx;
            }
        }
        __syncthreads();
        if (tid < K) {
            float sy = 0.f, sx = 0.f;
            for (int part = 0; part < parts; ++part) { sy += dkp[(tid * parts + part) * 2]; sx += dkp[(tid * parts + part) * 2 + 1]; }
            const long long bk = (long long)b * K + tid;
            dkv[2 * tid] = sy + (g_k ? g_k[bk * 2] : 0.f);
            dkv[2 * tid + 1] = sx + (g_k ? g_k[bk * 2 + 1] : 0.f);
        }
    }
    __syncthreads();

    // ---- 3: marginal log-softmax backward: d_heat = p_i (r_i - ky) gy / W + q_j (c_j - kx) gx / H ------------------
    const int nq = HW / V;
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

int set_lds_attributes_t() {
    static std::once_flag once[64];
    static int status[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    std::call_once(once[dev], [dev] {
        const void* fns[] = {reinterpret_cast<const void*>(bottleneck_fwd_t_k<1>), reinterpret_cast<const void*>(bottleneck_fwd_t_k<2>),
                             reinterpret_cast<const void*>(bottleneck_fwd_t_k<4>),
                             reinterpret_cast<const void*>(bottleneck_bwd_t_k<1, 1>), reinterpret_cast<const void*>(bottleneck_bwd_t_k<2, 1>),
                             reinterpret_cast<const void*>(bottleneck_bwd_t_k<4, 1>), reinterpret_cast<const void*>(bottleneck_bwd_t_k<1, 4>),
                             reinterpret_cast<const void*>(bottleneck_bwd_t_k<2, 4>), reinterpret_cast<const void*>(bottleneck_bwd_t_k<4, 4>)};
        status[dev] = 0;
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) status[dev] = 1;
    });
    if (status[dev]) {
        kp_set_error("keypoint bottleneck (T form): hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed on device %d", dev);
        return KP_E_LAUNCH;
    }
    return KP_OK;
}

}  // namespace

extern kp_knob_t g_kp_debug;   // kp_api.hip: kp_config(KP_CFG_DEBUG, mask)

extern "C" {

// 16-bit storage, 8 / 16 / 32 feature channels (the Pong-class nets, where the layout traffic is a measurable share of the
// step), maps the LDS reduction buffer of the backward holds
int kp_keypoint_bottleneck_t_supported(int B, int C, int K, int H, int W, int dtype) {
    if (dtype != KP_H16) return 0;
    return B > 0 && (C == 8 || C == 16 || C == 32) && K > 0 && K <= 255 && H > 0 && W > 0 && H <= MAXDIM && W <= MAXDIM &&
           (long long)H * W <= BWD_LDS_FLOATS;
}

int kp_keypoint_bottleneck_fwd_t(const float* heat_s, const float* heat_t, const void* phi_s, const void* phi_t, float* k_t,
                                 float* p_t, float* q_t, float* m_t, void* a, float* mask_s, float* mask_t,
                                 unsigned char* arg_t, int B, int C, int K, int H, int W, float sigma, float eps, int dtype,
                                 void* stream) {
    KP_REQUIRE(heat_s && heat_t && phi_s && phi_t && k_t && p_t && q_t && m_t && a && mask_s && mask_t && arg_t,
               "kp_keypoint_bottleneck_fwd_t: null pointer");
    KP_REQUIRE(sigma > 0.f, "kp_keypoint_bottleneck_fwd_t: sigma must be positive");
    if (!kp_keypoint_bottleneck_t_supported(B, C, K, H, W, dtype) || !aligned16(phi_s) || !aligned16(phi_t) || !aligned16(a))
        return KP_E_UNSUPPORTED;
    const float inv2s2 = (float)(1.0 / (2.0 * (double)sigma * (double)sigma));
    const size_t smem = (4 * (size_t)K + (size_t)NW * WAVE_FLOATS) * sizeof(float);
    if (int rc = set_lds_attributes_t()) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int dbg = g_kp_debug;
#define KP_BF(LPv)                                                                                                             \
    bottleneck_fwd_t_k<LPv><<<B, NT, smem, st>>>(heat_s, heat_t, (const h16_t*)phi_s, (const h16_t*)phi_t, k_t, p_t, q_t, m_t, \
                                                 (h16_t*)a, mask_s, mask_t, arg_t, K, H, W, inv2s2, eps, dbg)
    if (C == 8) KP_BF(1);
    else if (C == 16) KP_BF(2);
    else KP_BF(4);
#undef KP_BF
    return kp_check_launch("keypoint_bottleneck_fwd_t");
}

int kp_keypoint_bottleneck_bwd_t(const void* ga, const float* g_k, const float* g_m, const void* phi_s, const void* phi_t,
                                 const float* mask_s, const float* mask_t, const unsigned char* arg_t, const float* k_t,
                                 const float* p_t, const float* q_t, void* d_phi_t, float* d_heat_t, int B, int C, int K, int H,
                                 int W, float sigma, float eps, int dtype, void* stream) {
    KP_REQUIRE(ga && phi_s && phi_t && mask_s && mask_t && arg_t && k_t && p_t && q_t && d_phi_t && d_heat_t,
               "kp_keypoint_bottleneck_bwd_t: null pointer");
    KP_REQUIRE(sigma > 0.f, "kp_keypoint_bottleneck_bwd_t: sigma must be positive");
    if (!kp_keypoint_bottleneck_t_supported(B, C, K, H, W, dtype) || !aligned16(ga) || !aligned16(phi_s) || !aligned16(phi_t) ||
        !aligned16(d_phi_t))
        return KP_E_UNSUPPORTED;
    const float inv2s2 = (float)(1.0 / (2.0 * (double)sigma * (double)sigma));
    const size_t smem = ((size_t)BWD_LDS_FLOATS + 2 * (size_t)K + 2 * (size_t)(K >= NW ? K : NW)) * sizeof(float);   // + the parts' partial sums
    if (int rc = set_lds_attributes_t()) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int dbg = g_kp_debug;
    const bool vec = ((H * W) & 3) == 0 && aligned16(d_heat_t);
#define KP_BB(LPv, Vv)                                                                                                           \
    bottleneck_bwd_t_k<LPv, Vv><<<B, NT, smem, st>>>((const h16_t*)ga, g_k, g_m, (const h16_t*)phi_s, (const h16_t*)phi_t, mask_s, \
                                                     mask_t, arg_t, k_t, p_t, q_t, (h16_t*)d_phi_t, d_heat_t, K, H, W, inv2s2, eps, dbg)
    if (vec) {
        if (C == 8) KP_BB(1, 4);
        else if (C == 16) KP_BB(2, 4);
        else KP_BB(4, 4);
    } else {
        if (C == 8) KP_BB(1, 1);
        else if (C == 16) KP_BB(2, 1);
        else KP_BB(4, 1);
    }
#undef KP_BB
    return kp_check_launch("keypoint_bottleneck_bwd_t");
}

}  // extern "C"
