// BatchNorm(train) + activation (+ max-pool / bilinear x2) forward and backward: the HBM-bound passes
// between the convolutions.  Reference semantics: keypoints/models/vgg.py:24-38, knn.py:115-119
// (nn.BatchNorm2d train mode, LeakyReLU / ReLU, MaxPool2d(2,2), UpsamplingBilinear2d(x2)).
//
// Mapping: one workgroup per image row; a thread owns ONE 16-byte channel vector (8 bf16 / 4 fp32) and
// walks the pixels of the row, so the per-channel constants (mean, invstd, gamma, beta) are loaded once
// per thread and every global access is a full 16-byte lane access on consecutive addresses.
// All reductions are deterministic (per-row partial sums, then an fp64 tree) -- no atomics.
#include <stdlib.h>
#include "kp_common.h"

namespace {

constexpr int TPB = 256;
constexpr int RED_R = 64;   // row groups of the two-stage column reduction
[[maybe_unused]] constexpr long long WIDE_ROWS = 8192;   // up to here the one-launch wide reduction is used

__device__ __forceinline__ void fold_range(int h, int H, int p, int& lo, int& hi) {
    lo = (h == 0) ? 0 : h + p;
    hi = (h == H - 1) ? H + 2 * p - 1 : h + p;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- V-wide channel vectors (V * sizeof(T) == 16, or V == 1 for odd channel counts) -----------------
template <typename T, int V> struct Vec;
template <> struct Vec<float, 4> {
    static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) { ld4<float>(p, o); }
    static __device__ __forceinline__ void st(float* p, const float (&o)[4]) { st4<float>(p, o); }
};
template <> struct Vec<h16_t, 8> {
    static __device__ __forceinline__ void ld(const h16_t* p, float (&o)[8]) {
        const uint4 v = *reinterpret_cast<const uint4*>(p);
        unpack_h16x2(v.x, o[0], o[1]);
        unpack_h16x2(v.y, o[2], o[3]);
        unpack_h16x2(v.z, o[4], o[5]);
        unpack_h16x2(v.w, o[6], o[7]);
    }
    static __device__ __forceinline__ void st(h16_t* p, const float (&o)[8]) {
        uint4 v;
        v.x = pack_h16x2(o[0], o[1]);
        v.y = pack_h16x2(o[2], o[3]);
        v.z = pack_h16x2(o[4], o[5]);
        v.w = pack_h16x2(o[6], o[7]);
        *reinterpret_cast<uint4*>(p) = v;
    }
};
template <typename T> struct Vec<T, 1> {
    static __device__ __forceinline__ void ld(const T* p, float (&o)[1]) { o[0] = ::ld<T>(p); }
    static __device__ __forceinline__ void st(T* p, const float (&o)[1]) { ::st<T>(p, o[0]); }
};

struct BnP {
    const float *mean, *invstd, *gamma, *beta;
};

template <int V>
struct BnC {   // per-thread channel constants
    float mu[V], is[V], ga[V], be[V];
    bool affine;
    __device__ __forceinline__ void load(const BnP& b, int c) {
        affine = b.mean != nullptr;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            mu[v] = affine ? b.mean[c + v] : 0.f;
            is[v] = affine ? b.invstd[c + v] : 1.f;
            ga[v] = affine ? b.gamma[c + v] : 1.f;
            be[v] = affine ? b.beta[c + v] : 0.f;
        }
    }
    __device__ __forceinline__ float z(float y, int v) const { return affine ? (y - mu[v]) * is[v] * ga[v] + be[v] : y; }
    __device__ __forceinline__ float xhat(float y, int v) const { return (y - mu[v]) * is[v]; }
};

// thread -> (channel vector, pixel lane) for a row-walking kernel
struct RowMap {
    int cv, pl, npl;
    bool active;
    __device__ __forceinline__ RowMap(int CV) {
        if (CV >= TPB) { cv = threadIdx.x; pl = 0; npl = 1; active = true; }
        else { cv = threadIdx.x % CV; pl = threadIdx.x / CV; npl = TPB / CV; active = pl < npl; }
    }
};

// torch's upsample_bilinear2d(align_corners=True) source index / lambda for output index o
__device__ __forceinline__ void bilin(int o, int in_size, int out_size, int& i0, int& ip, float& l0, float& l1) {
    const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    const float r = scale * (float)o;
    i0 = (int)r;
    ip = (i0 < in_size - 1) ? 1 : 0;
    l1 = r - (float)i0;
    l0 = 1.f - l1;
}

// ---- statistics: [rows][2][C] fp32 partials -> [2C] fp64 ------------------------------------------
__global__ __launch_bounds__(TPB) void reduce_rows_k(const float* __restrict__ partials, double* __restrict__ ws,
                                                     long long rows, int C2) {
    // grid (ceil(C2/64), RED_R); block = 64 columns x 4 row lanes
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const long long per = (rows + RED_R - 1) / RED_R;
    const long long r0 = (long long)blockIdx.y * per, r1 = (r0 + per < rows) ? r0 + per : rows;
    double s = 0.0;
    if (col < C2)
        for (long long r = r0 + rl; r < r1; r += 4) s += (double)partials[r * C2 + col];
    __shared__ double red[4][64];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < C2) ws[(long long)blockIdx.y * C2 + col] = red[0][col & 63] + red[1][col & 63] + red[2][col & 63] + red[3][col & 63];
}

__global__ void bn_finalize_k(const double* __restrict__ ws, float* __restrict__ mean, float* __restrict__ invstd,
                              float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt,
                              float momentum, float eps, long long count, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < C) {
        double s = 0.0, q = 0.0;
        for (int r = 0; r < RED_R; ++r) { s += ws[(long long)r * 2 * C + c]; q += ws[(long long)r * 2 * C + C + c]; }
        const double mu = s / (double)count;
        double var = q / (double)count - mu * mu;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
        if (rvar) {
            const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    }
    if (nbt && c == 0) *nbt += 1;
}

__global__ void bn_bwd_finalize_k(const double* __restrict__ ws, float* __restrict__ sums, float* __restrict__ dgamma,
                                  float* __restrict__ dbeta, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int r = 0; r < RED_R; ++r) { s += ws[(long long)r * 2 * C + c]; q += ws[(long long)r * 2 * C + C + c]; }
    sums[c] = (float)s;
    sums[C + c] = (float)q;
    if (dbeta) dbeta[c] += (float)s;
    if (dgamma) dgamma[c] += (float)q;
}

// Single-launch variant for a moderate number of rows: one workgroup per 32 channels reduces its 64 columns
// (sum | sumsq of each channel) over all rows in fp64 (4 row lanes, fixed order -> deterministic) and finalizes.
// MODE 0: BatchNorm forward statistics; MODE 1: backward sums (dbeta | dgamma).
template <int MODE>
__global__ __launch_bounds__(TPB) void bn_reduce_finalize_k(const float* __restrict__ partials, long long rows, int C,
                                                            float* __restrict__ o0, float* __restrict__ o1,
                                                            float* __restrict__ r0p, float* __restrict__ r1p,
                                                            long long* __restrict__ nbt, float momentum, float eps,
                                                            long long count) {
    const int cl = threadIdx.x & 31, which = (threadIdx.x >> 5) & 1, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 32 + cl;
    double s = 0.0;
    if (c < C) {
        const float* col = partials + (long long)which * C + c;
        for (long long r = rl; r < rows; r += 4) s += (double)col[r * 2 * C];
    }
    __shared__ double red[4][64];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (threadIdx.x < 32 && c < C) {
        const double sum = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
        const double sq = red[0][32 + cl] + red[1][32 + cl] + red[2][32 + cl] + red[3][32 + cl];
        if (MODE == 0) {
            const double mu = sum / (double)count;
            double var = sq / (double)count - mu * mu;
            if (var < 0.0) var = 0.0;
            o0[c] = (float)mu;
            o1[c] = (float)(1.0 / sqrt(var + (double)eps));
            if (r0p) r0p[c] = (1.f - momentum) * r0p[c] + momentum * (float)mu;
            if (r1p) {
                const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
                r1p[c] = (1.f - momentum) * r1p[c] + momentum * (float)unbiased;
            }
        } else {
            o0[c] = (float)sum;        // sums[c]
            o0[C + c] = (float)sq;     // sums[C + c]
            if (r0p) r0p[c] += (float)sum;   // dbeta
            if (r1p) r1p[c] += (float)sq;    // dgamma
        }
    }
    if (MODE == 0 && nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

// Single launch for up to a few thousand rows (the persistent conv kernels write 4 rows per workgroup = 1024): one
// 1024-thread workgroup per 16 channels, 32 row lanes x (16 sum | 16 sumsq) columns, 8 independent loads in flight
// per lane; the 32 lanes are folded through LDS in index order (deterministic).  (A "last workgroup finalizes"
// two-stage kernel was measured 5x slower: device-scope fences between XCDs cost more than the second launch.)
// CPW = channels per workgroup (16, 8 or 4): narrow layers get as many workgroups as wide ones (a 64-channel layer with
// 8192 partial rows on 4 workgroups of 16 channels took 26 us, four times the 512-channel case).
// NT = threads per workgroup: 1024, or 256 for the 16 / 32-channel layers of the Pong-class nets.  A 16-wave workgroup needs
// four free wave slots on EVERY SIMD of one CU; beside the small-channel weight gradient (one wave per SIMD holding ~470 of its
// 512 registers, 152 KB of LDS) it never finds them and the launch -- 6 us of work on the critical path of the backward -- sat
// out the whole weight gradient: 80-100 us per layer in the config-2 timeline (tools/probes/cfg2_timeline.sh).  Four waves of
// 16 registers do fit.  (For model F the 256-thread form measured no difference in round 2; its layers keep 1024.)
// HALVES == 2 (kp_bn_finalize_pair, MODE 0): partials = [2][rows][2][C], the statistics of two half-batches that went through a
// layer in one launch; the workgroup finalizes half 0, then half 1 -- o0 / o1 are [2][C], the running statistics take the two
// momentum updates in that order (two forward calls of nn.BatchNorm2d in training mode: source frame, then target frame)
// (HALVES is a template parameter: as a run-time argument the loop made every finalize of model F 0.5 us longer -- 56 launches on
// the critical path of a step, each amplified by the side streams' schedule: +0.15 ms per step, tools/r04_run24.sh)
template <int MODE, int CPW, int NT = 1024, int HALVES = 1>
__global__ __launch_bounds__(NT * HALVES) void bn_reduce_finalize_wide_k(const float* __restrict__ partials, long long rows, int C,
                                                                           float* __restrict__ o0, float* __restrict__ o1,
                                                                           float* __restrict__ r0p, float* __restrict__ r1p,
                                                                           long long* __restrict__ nbt, float momentum, float eps,
                                                                           long long count) {
    // HALVES == 2 (kp_bn_finalize_pair): NT threads per half reduce their half's rows AT THE SAME TIME, each exactly as the one-half
    // launch would; only the updates of the running statistics take turns (half 0 first)
    const int hf = HALVES > 1 ? (int)threadIdx.x / NT : 0;
    const int tx = HALVES > 1 ? (int)threadIdx.x - hf * NT : (int)threadIdx.x;
    if (HALVES > 1 && hf) { partials += rows * 2 * C; o0 += C; o1 += C; }
    constexpr int RL = NT / (2 * CPW);   // row lanes
    const int cl = tx & (CPW - 1), which = (tx / CPW) & 1, rl = tx / (2 * CPW);
    const int c = blockIdx.x * CPW + cl;
    double s = 0.0;
    if (c < C) {
        const float* col = partials + (long long)which * C + c;
        const long long stride = 2LL * C;
        long long r = rl;
        for (; r + 7 * RL < rows; r += 8 * RL) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = col[(r + RL * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (double)v[u];
        }
        for (; r < rows; r += RL) s += (double)col[r * stride];
    }
    __shared__ double red[HALVES][RL][2 * CPW];
    red[hf][rl][tx & (2 * CPW - 1)] = s;
    __syncthreads();
    // fold the row lanes in index order (deterministic): 8 partial chains per column, then their sum
    __shared__ double red2[HALVES][8][2 * CPW];
    if (tx < 8 * 2 * CPW) {
        const int col2 = tx % (2 * CPW), part = tx / (2 * CPW);
        double a = 0.0;
        for (int r = part * (RL / 8); r < (part + 1) * (RL / 8); ++r) a += red[hf][r][col2];
        red2[hf][part][col2] = a;
    }
    __syncthreads();
#pragma unroll
    for (int turn = 0; turn < HALVES; ++turn) {
        if (turn) __syncthreads();
        if (hf == turn && tx < CPW && c < C) {
            double sum = 0.0, sq = 0.0;
            for (int r = 0; r < 8; ++r) { sum += red2[hf][r][cl]; sq += red2[hf][r][CPW + cl]; }
            if (MODE == 0) {
                const double mu = sum / (double)count;
                double var = sq / (double)count - mu * mu;
                if (var < 0.0) var = 0.0;
                o0[c] = (float)mu;
                o1[c] = (float)(1.0 / sqrt(var + (double)eps));
                if (r0p) r0p[c] = (1.f - momentum) * r0p[c] + momentum * (float)mu;
                if (r1p) {
                    const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
                    r1p[c] = (1.f - momentum) * r1p[c] + momentum * (float)unbiased;
                }
            } else {
                o0[c] = (float)sum;
                o0[C + c] = (float)sq;
                if (r0p) r0p[c] += (float)sum;   // dbeta
                if (r1p) r1p[c] += (float)sq;    // dgamma
            }
        }
        if (MODE == 0 && nbt && blockIdx.x == 0 && hf == turn && tx == 0) *nbt += 1;
    }
}

// ---- forward ----------------------------------------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(TPB) void bn_act_fwd_k(const T* __restrict__ y, BnP bn, float slope, int tail,
                                                    T* __restrict__ out, int p, unsigned char* __restrict__ argmax,
                                                    int H, int W, int C) {
    const int Ho = tail == KP_TAIL_MAXPOOL2 ? H / 2 : (tail == KP_TAIL_UPSAMPLE2 ? 2 * H : H);
    const int Wo = tail == KP_TAIL_MAXPOOL2 ? W / 2 : (tail == KP_TAIL_UPSAMPLE2 ? 2 * W : W);
    const int Hp = Ho + 2 * p, Wp = Wo + 2 * p, CV = C / V;
    const int n = blockIdx.x / Hp, i = blockIdx.x % Hp;
    const int oh = clampi(i - p, 0, Ho - 1);
    const RowMap rm(CV);
    if (!rm.active) return;
    const T* yn = y + (long long)n * H * W * C;
    T* orow = out + (long long)blockIdx.x * Wp * C;
    for (int cv = rm.cv; cv < CV; cv += TPB) {
        const int c = cv * V;
        BnC<V> k; k.load(bn, c);
        if (tail == KP_TAIL_NONE) {
            const T* yr = yn + (long long)oh * W * C + c;
            for (int j = rm.pl; j < Wp; j += rm.npl) {
                const int ow = clampi(j - p, 0, Wo - 1);
                float v[V], res[V];
                Vec<T, V>::ld(yr + (long long)ow * C, v);
#pragma unroll
                for (int e = 0; e < V; ++e) res[e] = act_fwd(k.z(v[e], e), slope);
                Vec<T, V>::st(orow + (long long)j * C + c, res);
            }
        } else if (tail == KP_TAIL_MAXPOOL2) {
            const T* y0 = yn + (long long)(2 * oh) * W * C + c;
            const T* y1 = y0 + (long long)W * C;
            for (int j = rm.pl; j < Wp; j += rm.npl) {
                const int ow = clampi(j - p, 0, Wo - 1);
                float res[V];
                int best[V];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[V];
                    Vec<T, V>::ld(((q >> 1) ? y1 : y0) + (long long)(2 * ow + (q & 1)) * C, v);
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        float a = act_fwd(k.z(v[e], e), slope);
                        // compared in fp32, before the storage rounding: two window values that round to the same
                        // 16-bit number still resolve to the fp32 maximum, which is where the fp32 reference
                        // (torch max_pool2d) routes the gradient; the stored maximum is the same either way
                        if (q == 0 || a > res[e] || a != a) { res[e] = a; best[e] = q; }
                    }
                }
                Vec<T, V>::st(orow + (long long)j * C + c, res);
                if (argmax && i - p == oh && j - p == ow) {
                    unsigned char* am = argmax + (((long long)n * Ho + oh) * Wo + ow) * C + c;
                    if constexpr (V == 8) {
                        uint2 pk;
                        pk.x = best[0] | (best[1] << 8) | (best[2] << 16) | (best[3] << 24);
                        pk.y = best[4] | (best[5] << 8) | (best[6] << 16) | (best[7] << 24);
                        *reinterpret_cast<uint2*>(am) = pk;
                    } else {
#pragma unroll
                        for (int e = 0; e < V; ++e) am[e] = (unsigned char)best[e];
                    }
                }
            }
        } else {
            int h0, hp; float lh0, lh1;
            bilin(oh, H, Ho, h0, hp, lh0, lh1);
            const T* y0 = yn + (long long)h0 * W * C + c;
            const T* y1 = y0 + (long long)hp * W * C;
            for (int j = rm.pl; j < Wp; j += rm.npl) {
                const int ow = clampi(j - p, 0, Wo - 1);
                int w0, wp; float lw0, lw1;
                bilin(ow, W, Wo, w0, wp, lw0, lw1);
                float v00[V], v01[V], v10[V], v11[V], res[V];
                Vec<T, V>::ld(y0 + (long long)w0 * C, v00);
                Vec<T, V>::ld(y0 + (long long)(w0 + wp) * C, v01);
                Vec<T, V>::ld(y1 + (long long)w0 * C, v10);
                Vec<T, V>::ld(y1 + (long long)(w0 + wp) * C, v11);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float a00 = act_fwd(k.z(v00[e], e), slope), a01 = act_fwd(k.z(v01[e], e), slope);
                    const float a10 = act_fwd(k.z(v10[e], e), slope), a11 = act_fwd(k.z(v11[e], e), slope);
                    res[e] = lh0 * (lw0 * a00 + lw1 * a01) + lh1 * (lw0 * a10 + lw1 * a11);
                }
                Vec<T, V>::st(orow + (long long)j * C + c, res);
            }
        }
    }
}

// incoming gradient w.r.t. this layer's activation output at pre-tail pixel (n,h,w), channels c..c+V
template <typename T, int V>
__device__ __forceinline__ void load_g(const T* __restrict__ gsrc, int gp, int tail,
                                       const unsigned char* __restrict__ argmax, int n, int h, int w, int c, int H,
                                       int W, int C, float (&g)[V]) {
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = 0.f;
    int Hs = H, Ws = W, sh = h, sw = w;
    unsigned take = 0xffu;
    if (tail == KP_TAIL_MAXPOOL2) {
        Hs = H / 2; Ws = W / 2; sh = h >> 1; sw = w >> 1;
        if (sh >= Hs || sw >= Ws) return;
        const unsigned pos = ((h & 1) << 1) | (w & 1);
        const unsigned char* am = argmax + (((long long)n * Hs + sh) * Ws + sw) * C + c;
        take = 0;
        if constexpr (V == 8) {
            const uint2 pk = *reinterpret_cast<const uint2*>(am);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                take |= (((pk.x >> (8 * e)) & 0xffu) == pos) ? (1u << e) : 0u;
                take |= (((pk.y >> (8 * e)) & 0xffu) == pos) ? (1u << (e + 4)) : 0u;
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) take |= (am[e] == pos) ? (1u << e) : 0u;
        }
        if (!take) return;
    }
    const int Hp = Hs + 2 * gp, Wp = Ws + 2 * gp;
    int ilo, ihi, jlo, jhi;
    fold_range(sh, Hs, gp, ilo, ihi);
    fold_range(sw, Ws, gp, jlo, jhi);
    for (int i = ilo; i <= ihi; ++i)
        for (int j = jlo; j <= jhi; ++j) {
            float v[V];
            Vec<T, V>::ld(gsrc + (((long long)n * Hp + i) * Wp + j) * C + c, v);
#pragma unroll
            for (int e = 0; e < V; ++e) g[e] += v[e];
        }
#pragma unroll
    for (int e = 0; e < V; ++e) if (!((take >> e) & 1u)) g[e] = 0.f;
}

// ---- backward pass 1: per-row partial sums of (ghat, ghat * xhat) -----------------------------------
template <typename T, int V>
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_k(const T* __restrict__ gsrc, int gp, int tail,
                                                       const unsigned char* __restrict__ argmax,
                                                       const T* __restrict__ y, BnP bn, float slope,
                                                       float* __restrict__ partials, int H, int W, int C) {
    const int CV = C / V;
    const int n = blockIdx.x / H, h = blockIdx.x % H;
    const RowMap rm(CV);
    extern __shared__ float red[];   // [npl][lanes][2V]
    const int lanes = CV < TPB ? CV : TPB;
    const T* yr = y + (long long)blockIdx.x * W * C;
    for (int cv0 = 0; cv0 < CV; cv0 += TPB) {
        const int cv = cv0 + rm.cv;
        const int c = cv * V;
        float sb[V], sg[V];
#pragma unroll
        for (int e = 0; e < V; ++e) sb[e] = sg[e] = 0.f;
        if (rm.active && cv < CV) {
            BnC<V> k; k.load(bn, c);
            for (int w = rm.pl; w < W; w += rm.npl) {
                float g[V], yv[V];
                load_g<T, V>(gsrc, gp, tail, argmax, n, h, w, c, H, W, C, g);
                Vec<T, V>::ld(yr + (long long)w * C + c, yv);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float gh = g[e] * act_grad(k.z(yv[e], e), slope);
                    sb[e] += gh;
                    sg[e] += gh * k.xhat(yv[e], e);
                }
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                red[((rm.pl * lanes + rm.cv) * 2 + 0) * V + e] = sb[e];
                red[((rm.pl * lanes + rm.cv) * 2 + 1) * V + e] = sg[e];
            }
        }
        __syncthreads();
        if (rm.active && rm.pl == 0 && cv < CV) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float a = 0.f, b = 0.f;
                for (int r2 = 0; r2 < rm.npl; ++r2) {
                    a += red[((r2 * lanes + rm.cv) * 2 + 0) * V + e];
                    b += red[((r2 * lanes + rm.cv) * 2 + 1) * V + e];
                }
                partials[((long long)blockIdx.x * 2 + 0) * C + c + e] = a;
                partials[((long long)blockIdx.x * 2 + 1) * C + c + e] = b;
            }
        }
        __syncthreads();
    }
}

// ---- backward pass 2: dy = gamma*invstd*(ghat - dbeta/M - xhat*dgamma/M), zero border of dp ------------
template <typename T, int V>
__global__ __launch_bounds__(TPB) void bn_bwd_apply_k(const T* __restrict__ gsrc, int gp, int tail,
                                                      const unsigned char* __restrict__ argmax,
                                                      const T* __restrict__ y, BnP bn, float slope,
                                                      const float* __restrict__ sums, T* __restrict__ dy, int dp,
                                                      int N, int H, int W, int C) {
    const int Hp = H + 2 * dp, Wp = W + 2 * dp, CV = C / V;
    const int n = blockIdx.x / Hp, i = blockIdx.x % Hp;
    const int h = i - dp;
    const RowMap rm(CV);
    if (!rm.active) return;
    T* drow = dy + (long long)blockIdx.x * Wp * C;
    const bool border_row = h < 0 || h >= H;
    const float invM = 1.0f / (float)((long long)N * H * W);
    for (int cv = rm.cv; cv < CV; cv += TPB) {
        const int c = cv * V;
        float zero[V];
#pragma unroll
        for (int e = 0; e < V; ++e) zero[e] = 0.f;
        if (border_row) {
            for (int j = rm.pl; j < Wp; j += rm.npl) Vec<T, V>::st(drow + (long long)j * C + c, zero);
            continue;
        }
        BnC<V> k; k.load(bn, c);
        float db[V], dg[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            db[e] = k.affine ? sums[c + e] * invM : 0.f;
            dg[e] = k.affine ? sums[C + c + e] * invM : 0.f;
        }
        const T* yr = y + ((long long)n * H + h) * W * C + c;
        for (int j = rm.pl; j < Wp; j += rm.npl) {
            const int w = j - dp;
            if (w < 0 || w >= W) { Vec<T, V>::st(drow + (long long)j * C + c, zero); continue; }
            float g[V], yv[V], res[V];
            load_g<T, V>(gsrc, gp, tail, argmax, n, h, w, c, H, W, C, g);
            Vec<T, V>::ld(yr + (long long)w * C, yv);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const float gh = g[e] * act_grad(k.z(yv[e], e), slope);
                res[e] = k.affine ? k.ga[e] * k.is[e] * (gh - db[e] - k.xhat(yv[e], e) * dg[e]) : gh;
            }
            Vec<T, V>::st(drow + (long long)j * C + c, res);
        }
    }
}

// ---- adjoint of bilinear x2 (align_corners=True) ------------------------------------------------------
template <typename T, int V>
__global__ __launch_bounds__(TPB) void upsample2_bwd_k(const T* __restrict__ gsrc, int gp, T* __restrict__ g, int H,
                                                       int W, int C) {
    const int CV = C / V, Ho = 2 * H, Wo = 2 * W, Hp = Ho + 2 * gp, Wp = Wo + 2 * gp;
    const int n = blockIdx.x / H, h = blockIdx.x % H;
    const RowMap rm(CV);
    if (!rm.active) return;
    const int i_lo = 2 * h - 2 < 0 ? 0 : 2 * h - 2, i_hi = 2 * h + 2 > Ho - 1 ? Ho - 1 : 2 * h + 2;
    for (int cv = rm.cv; cv < CV; cv += TPB) {
        const int c = cv * V;
        for (int w = rm.pl; w < W; w += rm.npl) {
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = 0.f;
            const int j_lo = 2 * w - 2 < 0 ? 0 : 2 * w - 2, j_hi = 2 * w + 2 > Wo - 1 ? Wo - 1 : 2 * w + 2;
            for (int i = i_lo; i <= i_hi; ++i) {
                int h0, hp; float lh0, lh1;
                bilin(i, H, Ho, h0, hp, lh0, lh1);
                const float wh = (h0 == h ? lh0 : 0.f) + (h0 + hp == h ? lh1 : 0.f);
                if (wh == 0.f) continue;
                int ilo, ihi; fold_range(i, Ho, gp, ilo, ihi);
                for (int j = j_lo; j <= j_hi; ++j) {
                    int w0, wp; float lw0, lw1;
                    bilin(j, W, Wo, w0, wp, lw0, lw1);
                    const float ww = (w0 == w ? lw0 : 0.f) + (w0 + wp == w ? lw1 : 0.f);
                    if (ww == 0.f) continue;
                    int jlo, jhi; fold_range(j, Wo, gp, jlo, jhi);
                    for (int a = ilo; a <= ihi; ++a)
                        for (int b = jlo; b <= jhi; ++b) {
                            float v[V];
                            Vec<T, V>::ld(gsrc + (((long long)n * Hp + a) * Wp + b) * C + c, v);
#pragma unroll
                            for (int e = 0; e < V; ++e) acc[e] += wh * ww * v[e];
                        }
                }
            }
            Vec<T, V>::st(g + ((long long)blockIdx.x * W + w) * C + c, acc);
        }
    }
}

template <typename T> constexpr int vec_for(int C) { return (C % (16 / (int)sizeof(T)) == 0) ? 16 / (int)sizeof(T) : 1; }

}  // namespace

// dispatch on (dtype, vector width): F32 -> V in {4,1}; BF16 -> V in {8,1}
#define KP_BN_DISPATCH(dtype, C, CALL)                                            \
    do {                                                                          \
        if ((dtype) == KP_F32) {                                                  \
            using T = float;                                                      \
            if ((C) % 4 == 0) { constexpr int V = 4; CALL; } else { constexpr int V = 1; CALL; } \
        } else {                                                                  \
            using T = h16_t;                                                     \
            if ((C) % 8 == 0) { constexpr int V = 8; CALL; } else { constexpr int V = 1; CALL; } \
        }                                                                         \
    } while (0)

static int reduce_rows(const float* partials, double* ws, long long rows, int C2, hipStream_t st) {
    reduce_rows_k<<<dim3(cdiv(C2, 64), RED_R), TPB, 0, st>>>(partials, ws, rows, C2);
    return kp_check_launch("reduce_rows");
}

int kp_bn_act_fwd_fast(const void*, const float*, const float*, const float*, const float*, float, int, void*, int,
                       unsigned char*, int, int, int, int, hipStream_t);
int kp_bn_bwd_fast(int, const void*, int, int, const unsigned char*, const void*, const float*, const float*,
                   const float*, const float*, float, float*, const float*, void*, int, int, int, int, int, hipStream_t);
int kp_upsample2_bwd_fast(const void*, int, void*, int, int, int, int, hipStream_t);
extern kp_knob_t g_kp_bn_fast;
extern kp_knob_t g_kp_fin_nt;     // kp_config(KP_CFG_FIN_NT, 256): measurement
extern kp_knob_t g_kp_fin_wide;   // kp_config(KP_CFG_FIN_WIDE, 1): measurement -- the 1024-thread finalize for every layer

extern "C" {

#ifndef KP_F16_BUILD   // no dtype: defined once, by the bf16 build of this source
int kp_bn_ws_doubles(int C) { return RED_R * 2 * C; }

int kp_bn_finalize(const float* stats, long long rows, double* ws, float* save_mean, float* save_invstd,
                   float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                   long long count, int C, void* stream) {
    KP_REQUIRE(C > 0 && count > 0 && rows > 0 && stats && ws, "kp_bn_finalize: bad args");
    kp_count_bytes(8.0 * rows * C);
    hipStream_t st = (hipStream_t)stream;
    const bool narrow = C <= 32 && !g_kp_fin_wide;   // Pong-class layers: the 4-wave kernel below at every row count (its pair form,
                                                     // kp_bn_finalize_pair, must be this call bit for bit)
    if (rows <= 128 && !narrow) {
        bn_reduce_finalize_k<0><<<cdiv(C, 32), TPB, 0, st>>>(stats, rows, C, save_mean, save_invstd, running_mean,
                                                             running_var, num_batches_tracked, momentum, eps, count);
        return kp_check_launch("bn_reduce_finalize");
    }
    if (rows <= WIDE_ROWS) {
#define KP_WIDE0(CPWv) bn_reduce_finalize_wide_k<0, CPWv><<<cdiv(C, CPWv), 1024, 0, st>>>(stats, rows, C, save_mean, save_invstd, running_mean, \
                                                                               running_var, num_batches_tracked, momentum, eps, count)
        if (narrow)   // a 4-wave workgroup (see the kernel)
            bn_reduce_finalize_wide_k<0, 4, 256><<<cdiv(C, 4), 256, 0, st>>>(stats, rows, C, save_mean, save_invstd, running_mean,
                                                                              running_var, num_batches_tracked, momentum, eps, count);
        else if (g_kp_fin_nt == 256)   // (default since round 6: a 4-wave workgroup becomes resident beside a persistent matrix-core kernel at once)
            bn_reduce_finalize_wide_k<0, 4, 256><<<cdiv(C, 4), 256, 0, st>>>(stats, rows, C, save_mean, save_invstd, running_mean,
                                                                              running_var, num_batches_tracked, momentum, eps, count);
        else if (C >= 256) KP_WIDE0(16); else if (C >= 128) KP_WIDE0(8); else KP_WIDE0(4);
#undef KP_WIDE0
        return kp_check_launch("bn_reduce_finalize_wide");
    }
    int rc = reduce_rows(stats, ws, rows, 2 * C, st);
    if (rc) return rc;
    bn_finalize_k<<<cdiv(C, 128), 128, 0, st>>>(ws, save_mean, save_invstd, running_mean, running_var,
                                                num_batches_tracked, momentum, eps, count, C);
    return kp_check_launch("bn_finalize");
}

// The statistics of TWO half-batches (kp_conv_fwd_pair / kp_conv_fwd_bnin_pair: stats = [2][rows][2][C]) in one launch: save_mean /
// save_invstd are [2][C]; the running statistics (and num_batches_tracked) are updated twice, half 0 first.  Each half's numbers
// are those of kp_bn_finalize on its rows, bit for bit.  Only the one-launch reduction of the narrow layers has this form
// (rows <= 8192 per half, C <= 32: the Pong-class layers); otherwise KP_E_UNSUPPORTED.
int kp_bn_finalize_pair_supported(long long rows, int C) {
    return (rows > 0 && rows <= WIDE_ROWS && C > 0 && C <= 32 && !g_kp_fin_wide) ? 1 : 0;
}

int kp_bn_finalize_pair(const float* stats, long long rows, float* save_mean, float* save_invstd, float* running_mean,
                        float* running_var, long long* num_batches_tracked, float momentum, float eps, long long count, int C,
                        void* stream) {
    KP_REQUIRE(C > 0 && count > 0 && rows > 0 && stats && save_mean && save_invstd, "kp_bn_finalize_pair: bad args");
    if (!kp_bn_finalize_pair_supported(rows, C)) return KP_E_UNSUPPORTED;
    kp_count_bytes(16.0 * rows * C);
    bn_reduce_finalize_wide_k<0, 4, 256, 2><<<cdiv(C, 4), 512, 0, (hipStream_t)stream>>>(stats, rows, C, save_mean, save_invstd, running_mean,
                                                                                        running_var, num_batches_tracked, momentum, eps, count);
    return kp_check_launch("bn_reduce_finalize_pair");
}

#endif

int kp_bn_act_fwd(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  float slope, int tail, void* out, int out_pad, unsigned char* argmax, int N, int H, int W, int C,
                  int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && out_pad >= 0 && tail >= 0 && tail <= 2, "kp_bn_act_fwd: bad dims");
    KP_REQUIRE(tail != KP_TAIL_MAXPOOL2 || (H >= 2 && W >= 2), "kp_bn_act_fwd: maxpool needs H,W >= 2");
    kp_count_bytes((dtype == KP_F32 ? 4.0 : 2.0) * ((double)N * H * W * C + (double)N * ((tail == 1 ? H / 2 : tail == 2 ? 2 * H : H) + 2 * out_pad) * ((tail == 1 ? W / 2 : tail == 2 ? 2 * W : W) + 2 * out_pad) * C) + (argmax ? (double)N * (H / 2) * (W / 2) * C : 0.0));
    const int Ho = tail == KP_TAIL_MAXPOOL2 ? H / 2 : (tail == KP_TAIL_UPSAMPLE2 ? 2 * H : H);
    BnP bn{mean, invstd, gamma, beta};
    const unsigned grid = (unsigned)((long long)N * (Ho + 2 * out_pad));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == KP_H16 && g_kp_bn_fast) {
        int rc = kp_bn_act_fwd_fast(y, mean, invstd, gamma, beta, slope, tail, out, out_pad, argmax, N, H, W, C, st);
        if (rc != KP_E_UNSUPPORTED) return rc;
    }
    KP_BN_DISPATCH(dtype, C, (bn_act_fwd_k<T, V><<<grid, TPB, 0, st>>>((const T*)y, bn, slope, tail, (T*)out, out_pad, argmax, H, W, C)));
    return kp_check_launch("bn_act_fwd");
}

int kp_bn_act_bwd_reduce(const void* gsrc, int gp, int tail, const unsigned char* argmax, const void* y,
                         const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                         float* partials, int N, int H, int W, int C, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && gp >= 0 && (tail == 0 || tail == 1) && (partials || kp_query_mode()), "kp_bn_act_bwd_reduce: bad args");
    KP_REQUIRE(tail == 0 || argmax != nullptr || kp_query_mode(), "kp_bn_act_bwd_reduce: maxpool needs argmax");
    kp_count_bytes((dtype == KP_F32 ? 4.0 : 2.0) * (((double)N * ((tail == 1 ? H / 2 : H) + 2 * gp) * ((tail == 1 ? W / 2 : W) + 2 * gp) * C) + (double)N * H * W * C) + (tail == 1 ? (double)N * (H / 2) * (W / 2) * C : 0.0));
    BnP bn{mean, invstd, gamma, beta};
    const unsigned grid = (unsigned)((long long)N * H);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == KP_H16 && g_kp_bn_fast) {
        int rc = kp_bn_bwd_fast(0, gsrc, gp, tail, argmax, y, mean, invstd, gamma, beta, slope, partials, nullptr, nullptr,
                                0, N, H, W, C, st);
        if (rc != KP_E_UNSUPPORTED) return rc;
    }
    kp_set_last_rows((long long)N * H);
    KP_QUERY_RETURN();
    KP_BN_DISPATCH(dtype, C, (bn_bwd_reduce_k<T, V><<<grid, TPB, (size_t)TPB * 2 * V * sizeof(float), st>>>((const T*)gsrc, gp, tail, argmax, (const T*)y, bn, slope, partials, H, W, C)));
    return kp_check_launch("bn_bwd_reduce");
}

#ifndef KP_F16_BUILD
int kp_bn_bwd_finalize(const float* partials, double* ws, long long rows, int C, float* sums, float* dgamma,
                       float* dbeta, void* stream) {
    KP_REQUIRE(partials && ws && sums && rows > 0 && C > 0, "kp_bn_bwd_finalize: bad args");
    kp_count_bytes(8.0 * rows * C);
    hipStream_t st = (hipStream_t)stream;
    if (rows <= 128) {
        bn_reduce_finalize_k<1><<<cdiv(C, 32), TPB, 0, st>>>(partials, rows, C, sums, nullptr, dbeta, dgamma, nullptr, 0.f,
                                                             0.f, 1);
        return kp_check_launch("bn_bwd_reduce_finalize");
    }
    if (rows <= WIDE_ROWS) {
#define KP_WIDE1(CPWv) bn_reduce_finalize_wide_k<1, CPWv><<<cdiv(C, CPWv), 1024, 0, st>>>(partials, rows, C, sums, nullptr, dbeta, dgamma, \
                                                                               nullptr, 0.f, 0.f, 1)
        if (C <= 32 && !g_kp_fin_wide)
            bn_reduce_finalize_wide_k<1, 4, 256><<<cdiv(C, 4), 256, 0, st>>>(partials, rows, C, sums, nullptr, dbeta, dgamma, nullptr,
                                                                              0.f, 0.f, 1);
        else if (g_kp_fin_nt == 256)
            bn_reduce_finalize_wide_k<1, 4, 256><<<cdiv(C, 4), 256, 0, st>>>(partials, rows, C, sums, nullptr, dbeta, dgamma, nullptr,
                                                                              0.f, 0.f, 1);
        else if (C >= 256) KP_WIDE1(16); else if (C >= 128) KP_WIDE1(8); else KP_WIDE1(4);
#undef KP_WIDE1
        return kp_check_launch("bn_bwd_reduce_finalize_wide");
    }
    int rc = reduce_rows(partials, ws, rows, 2 * C, st);
    if (rc) return rc;
    bn_bwd_finalize_k<<<cdiv(C, 128), 128, 0, st>>>(ws, sums, dgamma, dbeta, C);
    return kp_check_launch("bn_bwd_finalize");
}

#endif

int kp_bn_act_bwd_apply(const void* gsrc, int gp, int tail, const unsigned char* argmax, const void* y,
                        const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                        const float* sums, void* dy, int dp, int N, int H, int W, int C, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && gp >= 0 && dp >= 0 && (tail == 0 || tail == 1), "kp_bn_act_bwd_apply: bad args");
    KP_REQUIRE(mean == nullptr || sums != nullptr, "kp_bn_act_bwd_apply: sums required with BatchNorm");
    kp_count_bytes((dtype == KP_F32 ? 4.0 : 2.0) * (((double)N * ((tail == 1 ? H / 2 : H) + 2 * gp) * ((tail == 1 ? W / 2 : W) + 2 * gp) * C) + (double)N * H * W * C + (double)N * (H + 2 * dp) * (W + 2 * dp) * C) + (tail == 1 ? (double)N * (H / 2) * (W / 2) * C : 0.0));
    BnP bn{mean, invstd, gamma, beta};
    const unsigned grid = (unsigned)((long long)N * (H + 2 * dp));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == KP_H16 && g_kp_bn_fast) {
        int rc = kp_bn_bwd_fast(1, gsrc, gp, tail, argmax, y, mean, invstd, gamma, beta, slope, nullptr, sums, dy, dp, N,
                                H, W, C, st);
        if (rc != KP_E_UNSUPPORTED) return rc;
    }
    KP_BN_DISPATCH(dtype, C, (bn_bwd_apply_k<T, V><<<grid, TPB, 0, st>>>((const T*)gsrc, gp, tail, argmax, (const T*)y, bn, slope, sums, (T*)dy, dp, N, H, W, C)));
    return kp_check_launch("bn_bwd_apply");
}

int kp_upsample2_bwd(const void* gsrc, int gp, void* g, int N, int H, int W, int C, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && gp >= 0, "kp_upsample2_bwd: bad dims");
    kp_count_bytes((dtype == KP_F32 ? 4.0 : 2.0) * ((double)N * (2 * H + 2 * gp) * (2 * W + 2 * gp) * C + (double)N * H * W * C));
    const unsigned grid = (unsigned)((long long)N * H);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == KP_H16 && g_kp_bn_fast) {
        int rc = kp_upsample2_bwd_fast(gsrc, gp, g, N, H, W, C, st);
        if (rc != KP_E_UNSUPPORTED) return rc;
    }
    KP_BN_DISPATCH(dtype, C, (upsample2_bwd_k<T, V><<<grid, TPB, 0, st>>>((const T*)gsrc, gp, (T*)g, H, W, C)));
    return kp_check_launch("upsample2_bwd");
}

}  // extern "C"
