// Inference-only forward of a small-channel Unit as ONE kernel (the consumers are
// atari_demo.py:35-36 and tests/test_gl_hack.py:454-456 of the reference: `net.keypoint(x)` + spacial_logsoftmax on one
// frame at a time, BatchNorm in eval mode).
//
// The Unit (keypoints/models/knn.py:110-130): L x [ReplicationPad2d(1), Conv2d 3x3, BatchNorm2d (eval), (Leaky)ReLU] with
// no pooling / upsampling (the Pong-class tables of keypoints/models/vgg.py:64-70), then Conv2d 1x1 + LeakyReLU.  With the
// running statistics folded into the weights (kp_unit_infer_pack: w' = w * gamma / sqrt(var + eps), b' = (b - mean) * gamma /
// sqrt(var + eps) + beta) the whole chain of one output tile fits a workgroup's LDS:
//   * a workgroup owns a T x T tile of the heat-map and computes layer l on the tile grown by (L - l) pixels per side
//     (halo recomputation instead of a grid-wide exchange between layers: Pong-class layers cost ~1 MFLOP per tile);
//   * replicate padding is the clamp of the CENTRE coordinate: the value a layer holds at a position outside the image is
//     the value at the clamped position, which is what ReplicationPad2d feeds the next layer;
//   * activations live in LDS as [channel][pixel] fp32 (lanes = pixels: conflict-free), two buffers swapped per layer;
//   * a wavefront computes 8 output channels of 64 pixels: its weights are wave-uniform, so they are read through the
//     scalar cache (s_load) and enter the FMAs as SGPR operands -- no LDS or vector-memory traffic for weights;
//   * the last 3x3 layer keeps its channels in registers and applies the 1x1 head from there;
//   * arithmetic is fp32 throughout (the training kernels' 16-bit storage has no role at 0.07 GFLOP per frame).
// One launch per frame batch; spacial_logsoftmax (kp_spatial_softmax_fwd) follows as the second and last one.
#include <hip/hip_runtime.h>
#include "kp_common.h"

namespace {

constexpr int TPB = 256, OCT = 8, MAXL = 6;

struct InferDesc {
    int L;                 // 3x3 layers; index L is the 1x1 head
    int cin[MAXL + 1], cout[MAXL + 1];
    int woff[MAXL + 1], boff[MAXL + 1];   // float offsets into params
    float slope[MAXL + 1];
};

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Destination index of weight (k, c, tap) inside a layer's packed block (see the header comment of kp_unit_infer_pack):
// inner 3x3 layers  [c][k / 8][tap][k % 8]  (the 72 weights of one (input channel, channel group) are contiguous),
// the last 3x3 layer [c][tap][k]             (a wave holds all of its output channels),  head [c][k].
__device__ __forceinline__ int packed_index(int l, int L, int ci, int co, int k, int c, int t) {
    if (l == L) return c * co + k;
    if (l == L - 1) return (c * 9 + t) * co + k;
    return ((c * (co / OCT) + k / OCT) * 9 + t) * OCT + (k % OCT);
}

__global__ void unit_infer_pack_k(float* __restrict__ dst_w, float* __restrict__ dst_b, const float* __restrict__ w,
                                  const float* __restrict__ b, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const float* __restrict__ mean,
                                  const float* __restrict__ var, float eps, int l, int L, int ci, int co, int taps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int nw = co * ci * taps;
    if (idx >= nw + co) return;
    const int k = idx < nw ? idx / (ci * taps) : idx - nw;
    const float scale = var ? gamma[k] / sqrtf(var[k] + eps) : 1.f;
    if (idx < nw) {
        const int r = idx - k * ci * taps, c = r / taps, t = r - c * taps;      // source [k][c][tap]
        dst_w[packed_index(l, L, ci, co, k, c, t)] = w[idx] * scale;
    } else {
        const float bk = b ? b[k] : 0.f;
        dst_b[k] = var ? (bk - mean[k]) * scale + beta[k] : bk;
    }
}

// The last 3x3 layer and the 1x1 head: a wavefront computes ALL CO channels of its 64 pixels (one LDS read of an input
// value feeds CO FMAs), keeps them in registers and applies the head from there -- the widest activation of the chain is
// never stored.
// The same for an 8 x 8 tile (ONE block of 64 pixels): the four waves split the CO channels instead of the pixels, each
// forms the head's partial sums over its channels, and the partials meet in LDS (part[wave][k][pixel]).
template <int CO>
__device__ __forceinline__ void last_layer_and_head_split(const float* __restrict__ cur, float* __restrict__ part,
                                                          const float* __restrict__ params, float* __restrict__ heat,
                                                          const InferDesc& d, int b, int H, int W, int ty0, int tx0,
                                                          int wave, int lane, int tid) {
    constexpr int T = 8, Rin = T + 2, npi = Rin * Rin, CW = CO / 4;
    const int L = d.L, l = L - 1, ci = d.cin[l];
    const float* __restrict__ w = params + d.woff[l] + wave * CW;
    const float* __restrict__ bias = params + d.boff[l] + wave * CW;
    const float* __restrict__ wh = params + d.woff[L];
    const float* __restrict__ bh = params + d.boff[L];
    const float slope = d.slope[l], hslope = d.slope[L];
    const int K = d.cout[L];
    {
        const int i = lane / T, j = lane - i * T;
        const int cy = clampi(ty0 + i, 0, H - 1), cx = clampi(tx0 + j, 0, W - 1);
        const float* a0 = cur + (cy - ty0) * Rin + (cx - tx0);
        float acc[CW];
#pragma unroll
        for (int e = 0; e < CW; ++e) acc[e] = bias[e];
        for (int c = 0; c < ci; ++c) {
            const float* ac = a0 + c * npi;
            const float* wc = w + (long long)c * 9 * CO;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float a = ac[(t / 3) * Rin + (t % 3)];
#pragma unroll
                for (int e = 0; e < CW; ++e) acc[e] = fmaf(a, wc[t * CO + e], acc[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < CW; ++e) acc[e] = acc[e] > 0.f ? acc[e] : acc[e] * slope;
        for (int k = 0; k < K; ++k) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < CW; ++e) s = fmaf(acc[e], wh[(wave * CW + e) * K + k], s);
            part[(wave * K + k) * 64 + lane] = s;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < K * 64; idx += TPB) {
        const int k = idx >> 6, pl = idx & 63;
        const int gy = ty0 + pl / T, gx = tx0 + pl % T;
        if (gy < H && gx < W) {
            // (((bias + wave 0) + wave 1) + wave 2) + wave 3: a fixed order
            float s = bh[k];
#pragma unroll
            for (int v = 0; v < 4; ++v) s += part[(v * K + k) * 64 + pl];
            heat[(((long long)b * K + k) * H + gy) * W + gx] = s > 0.f ? s : s * hslope;
        }
    }
}

template <int CO>
__device__ __forceinline__ void last_layer_and_head(const float* __restrict__ cur, const float* __restrict__ params,
                                                    float* __restrict__ heat, const InferDesc& d, int b, int H, int W,
                                                    int T, int ty0, int tx0, int Rin, int wave, int lane) {
    const int L = d.L, l = L - 1, ci = d.cin[l];
    const float* __restrict__ w = params + d.woff[l];
    const float* __restrict__ bias = params + d.boff[l];
    const float* __restrict__ wh = params + d.woff[L];
    const float* __restrict__ bh = params + d.boff[L];
    const float slope = d.slope[l], hslope = d.slope[L];
    const int K = d.cout[L], npo = T * T, npi = Rin * Rin;
    for (int blk = wave; blk * 64 < npo; blk += TPB / 64) {
        const int p = blk * 64 + lane;
        const int pp = p < npo ? p : npo - 1;
        const int i = pp / T, j = pp - i * T;
        const int gy = ty0 + i, gx = tx0 + j;
        const int cy = clampi(gy, 0, H - 1), cx = clampi(gx, 0, W - 1);
        const float* a0 = cur + (cy - ty0) * Rin + (cx - tx0);
        float acc[CO];
#pragma unroll
        for (int e = 0; e < CO; ++e) acc[e] = bias[e];
        for (int c = 0; c < ci; ++c) {
            const float* ac = a0 + c * npi;
            const float* wc = w + (long long)c * 9 * CO;
            // TU taps' weights (TU * CO SGPRs) are in flight at a time: 32 + 32 of the ~100 a wave has
#pragma unroll 1
            for (int r = 0; r < 3; ++r) {
                const float* ar = ac + r * Rin;
                const float* wr = wc + r * 3 * CO;
#pragma unroll(CO >= 32 ? 1 : 3)
                for (int s_ = 0; s_ < 3; ++s_) {
                    const float a = ar[s_];
#pragma unroll
                    for (int e = 0; e < CO; ++e) acc[e] = fmaf(a, wr[s_ * CO + e], acc[e]);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < CO; ++e) acc[e] = acc[e] > 0.f ? acc[e] : acc[e] * slope;
        if (p < npo && gy < H && gx < W) {
            float* hp = heat + ((long long)b * K * H + gy) * W + gx;
            for (int k = 0; k < K; ++k) {
                float s = bh[k];
#pragma unroll
                for (int e = 0; e < CO; ++e) s = fmaf(acc[e], wh[e * K + k], s);
                hp[(long long)k * H * W] = s > 0.f ? s : s * hslope;
            }
        }
    }
}

__global__ __launch_bounds__(TPB) void unit_infer_k(const float* __restrict__ x, const float* __restrict__ params,
                                                    float* __restrict__ heat, const InferDesc d, int Cin, int H, int W,
                                                    int T, int tilesx, int tilesy, int capa, int capb) {
    extern __shared__ float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = blockIdx.x % (tilesx * tilesy), b = blockIdx.x / (tilesx * tilesy);
    const int ty0 = (tile / tilesx) * T, tx0 = (tile % tilesx) * T;
    const int L = d.L;
    float* cur = smem;          // input, outputs of layers 2, 4, ..
    float* nxt = smem + capa;   // outputs of layers 1, 3, ..

    // input region: the tile grown by L, coordinates clamped into the image (the first layer's replicate padding)
    int Rin = T + 2 * L;
    {
        const int np = Rin * Rin;
        const float* xb = x + (long long)b * Cin * H * W;
        for (int idx = tid; idx < Cin * np; idx += TPB) {
            const int c = idx / np, p = idx - c * np, i = p / Rin, j = p - i * Rin;
            const int gy = clampi(ty0 - L + i, 0, H - 1), gx = clampi(tx0 - L + j, 0, W - 1);
            cur[idx] = xb[((long long)c * H + gy) * W + gx];
        }
    }
    __syncthreads();

    for (int l = 0; l < L - 1; ++l) {
        const int Rout = Rin - 2, ci = d.cin[l], co = d.cout[l];
        const float* __restrict__ w = params + d.woff[l];
        const float* __restrict__ bias = params + d.boff[l];
        const float slope = d.slope[l];
        const int oy = ty0 - (L - 1 - l), ox = tx0 - (L - 1 - l);   // image coordinates of output position (0, 0)
        const int npo = Rout * Rout, npi = Rin * Rin;
        const int nblk = (npo + 63) >> 6, ngrp = co / OCT;
        for (int u = wave; u < nblk * ngrp; u += TPB / 64) {          // wave-uniform: (pixel block, channel group)
            const int g = u % ngrp, blk = u / ngrp;
            const int p = blk * 64 + lane;
            const int pp = p < npo ? p : npo - 1;
            const int i = pp / Rout, j = pp - i * Rout;
            // centre clamped into the image; its 3x3 window starts at (cy - 1, cx - 1), the input region at (oy - 1, ox - 1)
            const int cy = clampi(oy + i, 0, H - 1), cx = clampi(ox + j, 0, W - 1);
            const float* a0 = cur + (cy - oy) * Rin + (cx - ox);
            float acc[OCT];
#pragma unroll
            for (int e = 0; e < OCT; ++e) acc[e] = bias[g * OCT + e];
            for (int c = 0; c < ci; ++c) {
                const float* ac = a0 + c * npi;
                const float* wc = w + ((long long)c * ngrp + g) * (9 * OCT);   // 72 contiguous weights
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const float a = ac[(t / 3) * Rin + (t % 3)];
#pragma unroll
                    for (int e = 0; e < OCT; ++e) acc[e] = fmaf(a, wc[t * OCT + e], acc[e]);
                }
            }
            if (p < npo) {
#pragma unroll
                for (int e = 0; e < OCT; ++e) nxt[(g * OCT + e) * npo + p] = acc[e] > 0.f ? acc[e] : acc[e] * slope;
            }
        }
        __syncthreads();
        float* t_ = cur; cur = nxt; nxt = t_;
        Rin = Rout;
    }
    // Rin == T + 2 here
    const int co = d.cout[L - 1];
    if (T == 8) {
        float* part = smem + capa + capb;
        if (co == 32) last_layer_and_head_split<32>(cur, part, params, heat, d, b, H, W, ty0, tx0, wave, lane, tid);
        else if (co == 16) last_layer_and_head_split<16>(cur, part, params, heat, d, b, H, W, ty0, tx0, wave, lane, tid);
        else last_layer_and_head_split<8>(cur, part, params, heat, d, b, H, W, ty0, tx0, wave, lane, tid);
        return;
    }
    if (co == 32) last_layer_and_head<32>(cur, params, heat, d, b, H, W, T, ty0, tx0, Rin, wave, lane);
    else if (co == 16) last_layer_and_head<16>(cur, params, heat, d, b, H, W, T, ty0, tx0, Rin, wave, lane);
    else last_layer_and_head<8>(cur, params, heat, d, b, H, W, T, ty0, tx0, Rin, wave, lane);
}

static int fill_desc(InferDesc& d, int Cin, int L, const int* widths, const float* slopes) {
    if (L < 1 || L > MAXL || Cin < 1) return 0;
    d.L = L;
    int c = Cin, off = 0;
    for (int l = 0; l <= L; ++l) {
        const int k = widths[l];
        if (k < 1 || (l < L && k % OCT != 0) || (l == L - 1 && k != 8 && k != 16 && k != 32)) return 0;
        d.cin[l] = c; d.cout[l] = k;
        d.woff[l] = off; off += (l < L ? 9 : 1) * c * k;
        d.boff[l] = off; off += k;
        d.slope[l] = slopes ? slopes[l] : 0.f;
        c = k;
    }
    return off;
}

}  // namespace

extern "C" {

int kp_unit_infer_fwd(const float* x, const float* params, float* heat, int B, int Cin, int H, int W, int L,
                      const int* widths, const float* slopes, void* stream);

long long kp_unit_infer_param_floats(int Cin, int L, const int* widths) {
    InferDesc d;
    if (!widths) return KP_E_ARG;
    const int n = fill_desc(d, Cin, L, widths, nullptr);
    return n > 0 ? n : KP_E_UNSUPPORTED;
}

int kp_unit_infer_pack(float* params, int layer, int Cin, int L, const int* widths, const float* w, const float* b,
                       const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                       void* stream) {
    KP_REQUIRE(params && widths && w, "kp_unit_infer_pack: null pointer");
    KP_REQUIRE(!var || (gamma && beta && mean), "kp_unit_infer_pack: BatchNorm needs gamma, beta, mean and var");
    InferDesc d;
    if (fill_desc(d, Cin, L, widths, nullptr) <= 0 || layer < 0 || layer > L) {
        kp_set_error("kp_unit_infer_pack: unsupported chain or layer index %d of %d", layer, L);
        return KP_E_UNSUPPORTED;
    }
    const int ci = d.cin[layer], co = d.cout[layer], taps = layer < L ? 9 : 1;
    const int n = co * ci * taps + co;
    unit_infer_pack_k<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(params + d.woff[layer], params + d.boff[layer], w, b,
                                                                        gamma, beta, mean, var, eps, layer, L, ci, co, taps);
    return kp_check_launch("unit_infer_pack");
}

int kp_spatial_softmax_fwd(const float* heat, float* k, float* p, float* q, int BK, int H, int W, int log_domain,
                           void* stream);

int kp_unit_infer_keypoints_fwd(const float* x, const float* params, float* heat, float* k, float* p, float* q, int B,
                                int Cin, int H, int W, int L, const int* widths, const float* slopes, void* stream) {
    KP_REQUIRE(k && p && q && widths, "kp_unit_infer_keypoints_fwd: null pointer");
    const int rc = kp_unit_infer_fwd(x, params, heat, B, Cin, H, W, L, widths, slopes, stream);
    if (rc != KP_OK) return rc;
    return kp_spatial_softmax_fwd(heat, k, p, q, B * widths[L], H, W, 1, stream);
}

int kp_unit_infer_fwd(const float* x, const float* params, float* heat, int B, int Cin, int H, int W, int L,
                      const int* widths, const float* slopes, void* stream) {
    KP_REQUIRE(x && params && heat && widths && slopes, "kp_unit_infer_fwd: null pointer");
    KP_REQUIRE(B > 0 && H > 0 && W > 0, "kp_unit_infer_fwd: bad dims B=%d H=%d W=%d", B, H, W);
    InferDesc d;
    if (fill_desc(d, Cin, L, widths, slopes) <= 0) {
        kp_set_error("kp_unit_infer_fwd: 1..%d 3x3 layers with widths that are multiples of %d (got L=%d)", MAXL, OCT, L);
        return KP_E_UNSUPPORTED;
    }
    // LDS: buffer A holds the input and the outputs of layers 2, 4, .., buffer B those of layers 1, 3, ..; the last 3x3
    // layer's output stays in registers.  Tile edge 16 when that still gives every CU two workgroups and fits 48 KB
    // (three workgroups per CU), else 8 (one frame of 84 x 84: 121 tiles instead of 36).
    auto plan = [&](int T, int& tx, int& ty, int& ca, int& cb) {
        tx = (W + T - 1) / T; ty = (H + T - 1) / T;
        int R = T + 2 * L;
        ca = Cin * R * R; cb = 0;
        for (int l = 0; l < L - 1; ++l) {
            R -= 2;
            const int n = d.cout[l] * R * R;
            int& cap = (l & 1) ? ca : cb;
            cap = n > cap ? n : cap;
        }
    };
    int T = 16, tx, ty, ca, cb;
    plan(T, tx, ty, ca, cb);
    if ((long long)B * tx * ty < 512 || (size_t)(ca + cb) * 4 > 48 * 1024) { T = 8; plan(T, tx, ty, ca, cb); }
    const size_t lds = (size_t)(ca + cb + (T == 8 ? 4 * d.cout[L] * 64 : 0)) * sizeof(float);
    if (lds > 160 * 1024 || (long long)B * tx * ty > 0x7fffffffLL) {
        kp_set_error("kp_unit_infer_fwd: the layer chain does not fit one workgroup's LDS (%zu bytes)", lds);
        return KP_E_UNSUPPORTED;
    }
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)unit_infer_k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_set = true;
    }
    unit_infer_k<<<(unsigned)(B * tx * ty), TPB, lds, (hipStream_t)stream>>>(x, params, heat, d, Cin, H, W, T, tx, ty, ca, cb);
    return kp_check_launch("unit_infer");
}

}  // extern "C"
