// HBM-bound kernels around the convolutions: layout changes, weight packs, BatchNorm(train) +
// activation (+ max-pool / bilinear x2) forward and backward, the 1x1 head, L2 loss and Adam.
// Reference semantics: keypoints/models/vgg.py:16-39, knn.py:110-130, transporter.py:47,56-60.
#include "kp_common.h"

namespace {

constexpr int TPB = 256;

// rows of the padded image that replicate-padding maps onto interior row h (contiguous range)
__device__ __forceinline__ void fold_range(int h, int H, int p, int& lo, int& hi) {
    lo = (h == 0) ? 0 : h + p;
    hi = (h == H - 1) ? H + 2 * p - 1 : h + p;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---------------------------------------------------------------- layout
template <typename T>
__global__ void nchw_to_nhwc_k(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int H, int W, int p) {
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    const long long total = (long long)N * Hp * Wp;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % Wp);
    const int i = (int)((t / Wp) % Hp);
    const int n = (int)(t / ((long long)Wp * Hp));
    const int h = clampi(i - p, 0, H - 1), w = clampi(j - p, 0, W - 1);
    const float* s = src + ((long long)n * C * H + h) * W + w;
    T* d = dst + t * C;
    for (int c = 0; c < C; ++c) st<T>(d + c, s[(long long)c * H * W]);
}

template <typename T>
__global__ void nhwc_grad_to_nchw_k(const T* __restrict__ g, float* __restrict__ dst, int N, int C, int H, int W,
                                    int p) {
    const int Wp = W + 2 * p, Hp = H + 2 * p;
    const long long total = (long long)N * H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int w = (int)(t % W);
    const int h = (int)((t / W) % H);
    const int n = (int)(t / ((long long)W * H));
    int ilo, ihi, jlo, jhi;
    fold_range(h, H, p, ilo, ihi);
    fold_range(w, W, p, jlo, jhi);
    for (int c = 0; c < C; ++c) {
        float a = 0.f;
        for (int i = ilo; i <= ihi; ++i)
            for (int j = jlo; j <= jhi; ++j) a += ld<T>(g + (((long long)n * Hp + i) * Wp + j) * C + c);
        dst[(((long long)n * C + c) * H + h) * W + w] = a;
    }
}

// ---------------------------------------------------------------- weights
template <typename T>
__global__ void pack_weights_k(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int K, int C,
                               int R) {
    const long long total = (long long)K * R * R * C;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % C);
    const int s = (int)((t / C) % R);
    const int r = (int)((t / ((long long)C * R)) % R);
    const int k = (int)(t / ((long long)C * R * R));
    const float v = w[(((long long)k * C + c) * R + r) * R + s];
    st<T>(wf + t, v);
    if (wd) st<T>(wd + (((long long)c * R + (R - 1 - r)) * R + (R - 1 - s)) * K + k, v);
}

__global__ void wgrad_unpack_k(const float* __restrict__ scratch, float* __restrict__ dw, int K, int C, int R) {
    const long long total = (long long)K * C * R * R;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int s = (int)(t % R);
    const int r = (int)((t / R) % R);
    const int c = (int)((t / ((long long)R * R)) % C);
    const int k = (int)(t / ((long long)R * R * C));
    dw[t] += scratch[(((long long)k * R + r) * R + s) * C + c];
}

// ---------------------------------------------------------------- BatchNorm
__global__ __launch_bounds__(TPB) void bn_finalize_k(const float* __restrict__ partials, float* __restrict__ mean,
                                                     float* __restrict__ invstd, float* __restrict__ rmean,
                                                     float* __restrict__ rvar, long long* __restrict__ nbt,
                                                     float momentum, float eps, long long count, long long rows,
                                                     int C) {
    // block = 8 channels x 32 row-lanes; partials[row][2][C]
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int c = blockIdx.x * 8 + cl;
    double s = 0.0, q = 0.0;
    if (c < C)
        for (long long r = rl; r < rows; r += 32) {
            s += (double)partials[(r * 2 + 0) * C + c];
            q += (double)partials[(r * 2 + 1) * C + c];
        }
    __shared__ double red[2][32][8];
    red[0][rl][cl] = s;
    red[1][rl][cl] = q;
    __syncthreads();
    if (threadIdx.x < 8 && c < C) {
        double ss = 0.0, qq = 0.0;
        for (int i = 0; i < 32; ++i) { ss += red[0][i][cl]; qq += red[1][i][cl]; }
        const double mu = ss / (double)count;
        double var = qq / (double)count - mu * mu;
        if (var < 0.0) var = 0.0;
        mean[c] = (float)mu;
        invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean) rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mu;
        if (rvar) {
            const double unbiased = count > 1 ? var * (double)count / (double)(count - 1) : var;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unbiased;
        }
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

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

template <typename T, int V>
__device__ __forceinline__ void ldv(const T* p, float (&o)[V]) {
    if constexpr (V == 4) ld4<T>(p, o);
    else o[0] = ld<T>(p);
}
template <typename T, int V>
__device__ __forceinline__ void stv(T* p, const float (&o)[V]) {
    if constexpr (V == 4) st4<T>(p, o);
    else st<T>(p, o[0]);
}

// torch's upsample_bilinear2d(align_corners=True) source index / lambda for output index o
__device__ __forceinline__ void bilin(int o, int in_size, int out_size, int& i0, int& ip, float& l0, float& l1) {
    const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
    const float r = scale * (float)o;
    i0 = (int)r;
    ip = (i0 < in_size - 1) ? 1 : 0;
    l1 = r - (float)i0;
    l0 = 1.f - l1;
}

template <typename T, int V>
__global__ void bn_act_fwd_k(const T* __restrict__ y, BnP bn, float slope, int tail, T* __restrict__ out, int p,
                             unsigned char* __restrict__ argmax, int N, int H, int W, int C) {
    const int Ho = tail == KP_TAIL_MAXPOOL2 ? H / 2 : (tail == KP_TAIL_UPSAMPLE2 ? 2 * H : H);
    const int Wo = tail == KP_TAIL_MAXPOOL2 ? W / 2 : (tail == KP_TAIL_UPSAMPLE2 ? 2 * W : W);
    const int Hp = Ho + 2 * p, Wp = Wo + 2 * p, CV = C / V;
    const long long total = (long long)N * Hp * Wp * CV;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % CV) * V;
    long long r = t / CV;
    const int j = (int)(r % Wp); r /= Wp;
    const int i = (int)(r % Hp);
    const int n = (int)(r / Hp);
    const int oh = clampi(i - p, 0, Ho - 1), ow = clampi(j - p, 0, Wo - 1);
    BnC<V> k; k.load(bn, c);
    float res[V];
    const T* yn = y + (long long)n * H * W * C + c;
    if (tail == KP_TAIL_NONE) {
        float v[V]; ldv<T, V>(yn + ((long long)oh * W + ow) * C, v);
#pragma unroll
        for (int e = 0; e < V; ++e) res[e] = act_fwd(k.z(v[e], e), slope);
    } else if (tail == KP_TAIL_MAXPOOL2) {
        int best[V];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[V]; ldv<T, V>(yn + ((long long)(2 * oh + (q >> 1)) * W + 2 * ow + (q & 1)) * C, v);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                // compare what the next layer will actually see (storage precision)
                float a = act_fwd(k.z(v[e], e), slope);
                if constexpr (sizeof(T) == 2) a = bf2f(f2bf(a));
                if (q == 0 || a > res[e] || a != a) { res[e] = a; best[e] = q; }
            }
        }
        if (argmax && i - p == oh && j - p == ow) {
#pragma unroll
            for (int e = 0; e < V; ++e) argmax[(((long long)n * Ho + oh) * Wo + ow) * C + c + e] = (unsigned char)best[e];
        }
    } else {
        int h0, hp, w0, wp; float lh0, lh1, lw0, lw1;
        bilin(oh, H, Ho, h0, hp, lh0, lh1);
        bilin(ow, W, Wo, w0, wp, lw0, lw1);
        float v00[V], v01[V], v10[V], v11[V];
        ldv<T, V>(yn + ((long long)h0 * W + w0) * C, v00);
        ldv<T, V>(yn + ((long long)h0 * W + w0 + wp) * C, v01);
        ldv<T, V>(yn + ((long long)(h0 + hp) * W + w0) * C, v10);
        ldv<T, V>(yn + ((long long)(h0 + hp) * W + w0 + wp) * C, v11);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float a00 = act_fwd(k.z(v00[e], e), slope), a01 = act_fwd(k.z(v01[e], e), slope);
            const float a10 = act_fwd(k.z(v10[e], e), slope), a11 = act_fwd(k.z(v11[e], e), slope);
            res[e] = lh0 * (lw0 * a00 + lw1 * a01) + lh1 * (lw0 * a10 + lw1 * a11);
        }
    }
    stv<T, V>(out + t * V, res);
}

// incoming gradient w.r.t. this layer's activation output at pre-tail pixel (n,h,w), channels c..c+V
template <typename T, int V>
__device__ __forceinline__ void load_g(const T* __restrict__ gsrc, int gp, int tail,
                                       const unsigned char* __restrict__ argmax, int n, int h, int w, int c, int H,
                                       int W, int C, float (&g)[V]) {
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = 0.f;
    int Hs = H, Ws = W, sh = h, sw = w;
    bool take[V];
#pragma unroll
    for (int e = 0; e < V; ++e) take[e] = true;
    if (tail == KP_TAIL_MAXPOOL2) {
        Hs = H / 2; Ws = W / 2; sh = h >> 1; sw = w >> 1;
        if (sh >= Hs || sw >= Ws) return;
        const int pos = ((h & 1) << 1) | (w & 1);
        bool any = false;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            take[e] = argmax[(((long long)n * Hs + sh) * Ws + sw) * C + c + e] == pos;
            any |= take[e];
        }
        if (!any) return;
    }
    const int Hp = Hs + 2 * gp, Wp = Ws + 2 * gp;
    int ilo, ihi, jlo, jhi;
    fold_range(sh, Hs, gp, ilo, ihi);
    fold_range(sw, Ws, gp, jlo, jhi);
    for (int i = ilo; i <= ihi; ++i)
        for (int j = jlo; j <= jhi; ++j) {
            float v[V]; ldv<T, V>(gsrc + (((long long)n * Hp + i) * Wp + j) * C + c, v);
#pragma unroll
            for (int e = 0; e < V; ++e) g[e] += v[e];
        }
#pragma unroll
    for (int e = 0; e < V; ++e) if (!take[e]) g[e] = 0.f;
}

template <typename T, int V>
__global__ __launch_bounds__(TPB) void bn_bwd_reduce_k(const T* __restrict__ gsrc, int gp, int tail,
                                                       const unsigned char* __restrict__ argmax,
                                                       const T* __restrict__ y, BnP bn, float slope,
                                                       double* __restrict__ sums, int N, int H, int W, int C,
                                                       int pix_per_block) {
    const int CV = C / V;
    const int cols = CV < TPB ? CV : TPB, rows = TPB / cols;
    const int col = threadIdx.x % cols, row = threadIdx.x / cols;
    const long long M = (long long)N * H * W;
    const long long mbeg = (long long)blockIdx.x * pix_per_block;
    const long long mend = mbeg + pix_per_block < M ? mbeg + pix_per_block : M;
    extern __shared__ double red[];   // [rows][2*V] per col chunk
    for (int cv = col; cv < CV; cv += cols) {
        const int c = cv * V;
        BnC<V> k; k.load(bn, c);
        float sb[V], sg[V];
#pragma unroll
        for (int e = 0; e < V; ++e) sb[e] = sg[e] = 0.f;
        if (row < rows) {
            for (long long m = mbeg + row; m < mend; m += rows) {
                const int w = (int)(m % W);
                const int h = (int)((m / W) % H);
                const int n = (int)(m / ((long long)W * H));
                float g[V], yv[V];
                load_g<T, V>(gsrc, gp, tail, argmax, n, h, w, c, H, W, C, g);
                ldv<T, V>(y + m * C + c, yv);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const float gh = g[e] * act_grad(k.z(yv[e], e), slope);
                    sb[e] += gh;
                    sg[e] += gh * k.xhat(yv[e], e);
                }
            }
        }
        // reduce over rows through LDS
        __syncthreads();
        if (row < rows) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                red[((row * cols + col) * V + e) * 2 + 0] = (double)sb[e];
                red[((row * cols + col) * V + e) * 2 + 1] = (double)sg[e];
            }
        }
        __syncthreads();
        if (row == 0) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                double a = 0.0, b = 0.0;
                for (int r2 = 0; r2 < rows; ++r2) {
                    a += red[((r2 * cols + col) * V + e) * 2 + 0];
                    b += red[((r2 * cols + col) * V + e) * 2 + 1];
                }
                atomicAdd(sums + c + e, a);
                atomicAdd(sums + C + c + e, b);
            }
        }
    }
}

template <typename T, int V>
__global__ void bn_bwd_apply_k(const T* __restrict__ gsrc, int gp, int tail, const unsigned char* __restrict__ argmax,
                               const T* __restrict__ y, BnP bn, float slope, const double* __restrict__ sums,
                               float* __restrict__ dgamma, float* __restrict__ dbeta, T* __restrict__ dy, int dp,
                               int N, int H, int W, int C) {
    const int Hp = H + 2 * dp, Wp = W + 2 * dp, CV = C / V;
    const long long total = (long long)N * Hp * Wp * CV;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && sums != nullptr) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            if (dbeta) dbeta[c] += (float)sums[c];
            if (dgamma) dgamma[c] += (float)sums[C + c];
        }
    }
    if (t >= total) return;
    const int c = (int)(t % CV) * V;
    long long r = t / CV;
    const int j = (int)(r % Wp); r /= Wp;
    const int i = (int)(r % Hp);
    const int n = (int)(r / Hp);
    const int h = i - dp, w = j - dp;
    float res[V];
#pragma unroll
    for (int e = 0; e < V; ++e) res[e] = 0.f;
    if (h >= 0 && h < H && w >= 0 && w < W) {
        BnC<V> k; k.load(bn, c);
        float g[V], yv[V];
        load_g<T, V>(gsrc, gp, tail, argmax, n, h, w, c, H, W, C, g);
        ldv<T, V>(y + (((long long)n * H + h) * W + w) * C + c, yv);
        const float invM = 1.0f / (float)((long long)N * H * W);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float gh = g[e] * act_grad(k.z(yv[e], e), slope);
            if (k.affine) {
                const float db = (float)sums[c + e], dg = (float)sums[C + c + e];
                res[e] = k.ga[e] * k.is[e] * (gh - db * invM - k.xhat(yv[e], e) * dg * invM);
            } else {
                res[e] = gh;
            }
        }
    }
    stv<T, V>(dy + t * V, res);
}

template <typename T, int V>
__global__ void upsample2_bwd_k(const T* __restrict__ gsrc, int gp, T* __restrict__ g, int N, int H, int W, int C) {
    const int CV = C / V, Ho = 2 * H, Wo = 2 * W, Hp = Ho + 2 * gp, Wp = Wo + 2 * gp;
    const long long total = (long long)N * H * W * CV;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % CV) * V;
    long long r = t / CV;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    float acc[V];
#pragma unroll
    for (int e = 0; e < V; ++e) acc[e] = 0.f;
    const int i_lo = 2 * h - 2 < 0 ? 0 : 2 * h - 2, i_hi = 2 * h + 2 > Ho - 1 ? Ho - 1 : 2 * h + 2;
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
                    float v[V]; ldv<T, V>(gsrc + (((long long)n * Hp + a) * Wp + b) * C + c, v);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += wh * ww * v[e];
                }
        }
    }
    stv<T, V>(g + t * V, acc);
}

// ---------------------------------------------------------------- 1x1 head
constexpr int HPB = 16;   // pixels per head block

template <typename T>
__global__ __launch_bounds__(TPB) void head_fwd_k(const T* __restrict__ x, const float* __restrict__ w,
                                                  const float* __restrict__ bias, float slope,
                                                  float* __restrict__ out, long long M, int HW, int C, int K) {
    extern __shared__ float xs[];   // [HPB][C+1]
    const long long m0 = (long long)blockIdx.x * HPB;
    for (int e = threadIdx.x; e < HPB * C; e += TPB) {
        const int p = e / C, c = e % C;
        xs[p * (C + 1) + c] = (m0 + p < M) ? ld<T>(x + (m0 + p) * C + c) : 0.f;
    }
    __syncthreads();
    const int p = threadIdx.x & (HPB - 1), kk = threadIdx.x / HPB;
    const long long m = m0 + p;
    if (m >= M) return;
    const long long n = m / HW, hw = m % HW;
    for (int k = kk; k < K; k += TPB / HPB) {
        float acc = 0.f;
        const float* wr = w + (long long)k * C;
        for (int c = 0; c < C; ++c) acc = fmaf(xs[p * (C + 1) + c], wr[c], acc);
        acc += bias ? bias[k] : 0.f;
        out[(n * K + k) * HW + hw] = act_fwd(acc, slope);
    }
}

template <typename T>
__global__ __launch_bounds__(TPB) void head_bwd_dx_k(const float* __restrict__ gout, const float* __restrict__ out,
                                                     const float* __restrict__ w, float slope, T* __restrict__ dx,
                                                     long long M, int HW, int C, int K) {
    extern __shared__ float dz[];   // [HPB][K+1]
    const long long m0 = (long long)blockIdx.x * HPB;
    for (int e = threadIdx.x; e < HPB * K; e += TPB) {
        const int p = e % HPB, k = e / HPB;
        const long long m = m0 + p;
        float v = 0.f;
        if (m < M) {
            const long long idx = ((m / HW) * K + k) * HW + (m % HW);
            v = gout[idx] * act_grad(out[idx], slope);
        }
        dz[p * (K + 1) + k] = v;
    }
    __syncthreads();
    const int p = threadIdx.x & (HPB - 1), cc = threadIdx.x / HPB;
    const long long m = m0 + p;
    if (m >= M) return;
    for (int c = cc; c < C; c += TPB / HPB) {
        float acc = 0.f;
        for (int k = 0; k < K; ++k) acc = fmaf(dz[p * (K + 1) + k], w[(long long)k * C + c], acc);
        st<T>(dx + m * C + c, acc);
    }
}

constexpr int HPAIRS = 32;   // (k,c) pairs per thread in the head wgrad

template <typename T>
__global__ __launch_bounds__(TPB) void head_bwd_dw_k(const float* __restrict__ gout, const float* __restrict__ out,
                                                     const T* __restrict__ x, float slope, float* __restrict__ dw,
                                                     float* __restrict__ db, long long M, int HW, int C, int K) {
    extern __shared__ float sm[];   // dz [HPB][K] then xs [HPB][C]
    float* dz = sm;
    float* xs = sm + HPB * K;
    const long long pair0 = (long long)blockIdx.y * TPB * HPAIRS;
    const long long npairs = (long long)K * C;
    float acc[HPAIRS];
#pragma unroll
    for (int i = 0; i < HPAIRS; ++i) acc[i] = 0.f;
    float dbacc = 0.f;   // thread tid < K owns db[tid]
    const long long tiles = (M + HPB - 1) / HPB;
    for (long long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        const long long m0 = tile * HPB;
        __syncthreads();
        for (int e = threadIdx.x; e < HPB * K; e += TPB) {
            const int p = e % HPB, k = e / HPB;
            const long long m = m0 + p;
            float v = 0.f;
            if (m < M) {
                const long long idx = ((m / HW) * K + k) * HW + (m % HW);
                v = gout[idx] * act_grad(out[idx], slope);
            }
            dz[p * K + k] = v;
        }
        for (int e = threadIdx.x; e < HPB * C; e += TPB) {
            const int p = e / C, c = e % C;
            xs[p * C + c] = (m0 + p < M) ? ld<T>(x + (m0 + p) * C + c) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < HPAIRS; ++i) {
            const long long pr = pair0 + threadIdx.x + (long long)i * TPB;
            if (pr < npairs) {
                const int k = (int)(pr / C), c = (int)(pr % C);
                float a = acc[i];
#pragma unroll
                for (int p = 0; p < HPB; ++p) a = fmaf(dz[p * K + k], xs[p * C + c], a);
                acc[i] = a;
            }
        }
        if (blockIdx.y == 0 && threadIdx.x < K) {   // K <= TPB (checked by the launcher)
#pragma unroll
            for (int p = 0; p < HPB; ++p) dbacc += dz[p * K + threadIdx.x];
        }
    }
#pragma unroll
    for (int i = 0; i < HPAIRS; ++i) {
        const long long pr = pair0 + threadIdx.x + (long long)i * TPB;
        if (pr < npairs) atomicAdd(dw + pr, acc[i]);
    }
    if (blockIdx.y == 0 && threadIdx.x < K && db) atomicAdd(db + threadIdx.x, dbacc);
}

// ---------------------------------------------------------------- loss / Adam
__global__ __launch_bounds__(TPB) void l2_loss_k(const float* __restrict__ x, const float* __restrict__ t,
                                                 const float* __restrict__ mask, double* __restrict__ loss_sum,
                                                 float* __restrict__ grad, float scale, size_t n) {
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = x[i] - t[i];
        const float mk = mask ? mask[i] : 1.f;
        acc += (double)(d * d * mk);
        if (grad) grad[i] = 2.f * d * mk * scale;
    }
    acc = wave_sum_d(acc);
    __shared__ double part[TPB / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < TPB / 64; ++i) s += part[i];
        atomicAdd(loss_sum, s);
    }
}

__global__ void adam_k(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                       float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float bc1,
                       float bc2_sqrt) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] -= (lr / bc1) * (mi / denom);
}

}  // namespace

#define DISPATCH_T(dtype, ...)                         \
    do {                                               \
        if ((dtype) == KP_F32) { using T = float; __VA_ARGS__; } \
        else { using T = bf16_t; __VA_ARGS__; }        \
    } while (0)

extern "C" {

int kp_nchw_to_nhwc(const float* src, void* dst, int N, int C, int H, int W, int pad, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && pad >= 0, "kp_nchw_to_nhwc: bad dims");
    const long long total = (long long)N * (H + 2 * pad) * (W + 2 * pad);
    DISPATCH_T(dtype, nchw_to_nhwc_k<T><<<cdiv(total, TPB), TPB, 0, (hipStream_t)stream>>>(src, (T*)dst, N, C, H, W, pad));
    return kp_check_launch("nchw_to_nhwc");
}

int kp_nhwc_grad_to_nchw(const void* gsrc, float* dst, int N, int C, int H, int W, int pad, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && pad >= 0, "kp_nhwc_grad_to_nchw: bad dims");
    const long long total = (long long)N * H * W;
    DISPATCH_T(dtype, nhwc_grad_to_nchw_k<T><<<cdiv(total, TPB), TPB, 0, (hipStream_t)stream>>>((const T*)gsrc, dst, N, C, H, W, pad));
    return kp_check_launch("nhwc_grad_to_nchw");
}

int kp_pack_weights(const float* w, void* w_fwd, void* w_dgrad, int K, int C, int R, int dtype, void* stream) {
    KP_REQUIRE(K > 0 && C > 0 && (R == 1 || R == 3), "kp_pack_weights: bad dims");
    const long long total = (long long)K * C * R * R;
    DISPATCH_T(dtype, pack_weights_k<T><<<cdiv(total, TPB), TPB, 0, (hipStream_t)stream>>>(w, (T*)w_fwd, (T*)w_dgrad, K, C, R));
    return kp_check_launch("pack_weights");
}

int kp_wgrad_unpack(const float* scratch, float* dw, int K, int C, int R, void* stream) {
    KP_REQUIRE(K > 0 && C > 0 && (R == 1 || R == 3), "kp_wgrad_unpack: bad dims");
    const long long total = (long long)K * C * R * R;
    wgrad_unpack_k<<<cdiv(total, TPB), TPB, 0, (hipStream_t)stream>>>(scratch, dw, K, C, R);
    return kp_check_launch("wgrad_unpack");
}

int kp_bn_finalize(const float* stats, float* save_mean, float* save_invstd, float* running_mean, float* running_var,
                   long long* num_batches_tracked, float momentum, float eps, long long count, int C, void* stream) {
    KP_REQUIRE(C > 0 && count > 0, "kp_bn_finalize: bad dims");
    bn_finalize_k<<<cdiv(C, 8), TPB, 0, (hipStream_t)stream>>>(stats, save_mean, save_invstd, running_mean, running_var,
                                                               num_batches_tracked, momentum, eps, count,
                                                               (count + 63) / 64, C);
    return kp_check_launch("bn_finalize");
}

int kp_bn_act_fwd(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  float slope, int tail, void* out, int out_pad, unsigned char* argmax, int N, int H, int W, int C,
                  int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && out_pad >= 0 && tail >= 0 && tail <= 2, "kp_bn_act_fwd: bad dims");
    KP_REQUIRE(tail != KP_TAIL_MAXPOOL2 || (H >= 2 && W >= 2), "kp_bn_act_fwd: maxpool needs H,W >= 2");
    const int Ho = tail == KP_TAIL_MAXPOOL2 ? H / 2 : (tail == KP_TAIL_UPSAMPLE2 ? 2 * H : H);
    const int Wo = tail == KP_TAIL_MAXPOOL2 ? W / 2 : (tail == KP_TAIL_UPSAMPLE2 ? 2 * W : W);
    BnP bn{mean, invstd, gamma, beta};
    const long long pix = (long long)N * (Ho + 2 * out_pad) * (Wo + 2 * out_pad);
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0)
        DISPATCH_T(dtype, bn_act_fwd_k<T, 4><<<cdiv(pix * (C / 4), TPB), TPB, 0, st>>>((const T*)y, bn, slope, tail, (T*)out, out_pad, argmax, N, H, W, C));
    else
        DISPATCH_T(dtype, bn_act_fwd_k<T, 1><<<cdiv(pix * C, TPB), TPB, 0, st>>>((const T*)y, bn, slope, tail, (T*)out, out_pad, argmax, N, H, W, C));
    return kp_check_launch("bn_act_fwd");
}

int kp_bn_act_bwd_reduce(const void* gsrc, int gp, int tail, const unsigned char* argmax, const void* y,
                         const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                         double* sums, int N, int H, int W, int C, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && gp >= 0 && (tail == 0 || tail == 1), "kp_bn_act_bwd_reduce: bad args");
    KP_REQUIRE(tail == 0 || argmax != nullptr, "kp_bn_act_bwd_reduce: maxpool needs argmax");
    KP_REQUIRE((C % 4 == 0 ? C / 4 : C) <= TPB, "kp_bn_act_bwd_reduce: C too large (<= 1024, or <= 256 if C %% 4 != 0)");
    BnP bn{mean, invstd, gamma, beta};
    const long long M = (long long)N * H * W;
    long long ppb = (M + 1023) / 1024;
    if (ppb < 64) ppb = 64;
    const unsigned grid = cdiv(M, ppb);
    hipStream_t st = (hipStream_t)stream;
    const int V = (C % 4 == 0) ? 4 : 1;
    const size_t smem = (size_t)TPB * V * 2 * sizeof(double);
    if (V == 4)
        DISPATCH_T(dtype, bn_bwd_reduce_k<T, 4><<<grid, TPB, smem, st>>>((const T*)gsrc, gp, tail, argmax, (const T*)y, bn, slope, sums, N, H, W, C, (int)ppb));
    else
        DISPATCH_T(dtype, bn_bwd_reduce_k<T, 1><<<grid, TPB, smem, st>>>((const T*)gsrc, gp, tail, argmax, (const T*)y, bn, slope, sums, N, H, W, C, (int)ppb));
    return kp_check_launch("bn_bwd_reduce");
}

int kp_bn_act_bwd_apply(const void* gsrc, int gp, int tail, const unsigned char* argmax, const void* y,
                        const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                        const double* sums, float* dgamma, float* dbeta, void* dy, int dp, int N, int H, int W, int C,
                        int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && gp >= 0 && dp >= 0 && (tail == 0 || tail == 1), "kp_bn_act_bwd_apply: bad args");
    BnP bn{mean, invstd, gamma, beta};
    const long long pix = (long long)N * (H + 2 * dp) * (W + 2 * dp);
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0)
        DISPATCH_T(dtype, bn_bwd_apply_k<T, 4><<<cdiv(pix * (C / 4), TPB), TPB, 0, st>>>((const T*)gsrc, gp, tail, argmax, (const T*)y, bn, slope, sums, dgamma, dbeta, (T*)dy, dp, N, H, W, C));
    else
        DISPATCH_T(dtype, bn_bwd_apply_k<T, 1><<<cdiv(pix * C, TPB), TPB, 0, st>>>((const T*)gsrc, gp, tail, argmax, (const T*)y, bn, slope, sums, dgamma, dbeta, (T*)dy, dp, N, H, W, C));
    return kp_check_launch("bn_bwd_apply");
}

int kp_upsample2_bwd(const void* gsrc, int gp, void* g, int N, int H, int W, int C, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && gp >= 0, "kp_upsample2_bwd: bad dims");
    const long long pix = (long long)N * H * W;
    hipStream_t st = (hipStream_t)stream;
    if (C % 4 == 0)
        DISPATCH_T(dtype, upsample2_bwd_k<T, 4><<<cdiv(pix * (C / 4), TPB), TPB, 0, st>>>((const T*)gsrc, gp, (T*)g, N, H, W, C));
    else
        DISPATCH_T(dtype, upsample2_bwd_k<T, 1><<<cdiv(pix * C, TPB), TPB, 0, st>>>((const T*)gsrc, gp, (T*)g, N, H, W, C));
    return kp_check_launch("upsample2_bwd");
}

int kp_head_fwd(const void* x, const float* w, const float* bias, float slope, float* out, int N, int H, int W, int C,
                int K, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "kp_head_fwd: bad dims");
    KP_REQUIRE(C <= 2048, "kp_head_fwd: C too large for the LDS tile");
    const long long M = (long long)N * H * W;
    const size_t smem = (size_t)HPB * (C + 1) * sizeof(float);
    DISPATCH_T(dtype, head_fwd_k<T><<<cdiv(M, HPB), TPB, smem, (hipStream_t)stream>>>((const T*)x, w, bias, slope, out, M, H * W, C, K));
    return kp_check_launch("head_fwd");
}

int kp_head_bwd(const float* gout, const float* out, const void* x, const float* w, float slope, void* dx, float* dw,
                float* db, int N, int H, int W, int C, int K, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "kp_head_bwd: bad dims");
    KP_REQUIRE(K <= 256 && C <= 2048, "kp_head_bwd: K <= 256, C <= 2048 supported");
    const long long M = (long long)N * H * W;
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        const size_t smem = (size_t)HPB * (K + 1) * sizeof(float);
        DISPATCH_T(dtype, head_bwd_dx_k<T><<<cdiv(M, HPB), TPB, smem, st>>>(gout, out, w, slope, (T*)dx, M, H * W, C, K));
        int rc = kp_check_launch("head_bwd_dx");
        if (rc) return rc;
    }
    if (dw) {
        const long long tiles = (M + HPB - 1) / HPB;
        const unsigned gx = (unsigned)(tiles < 512 ? tiles : 512);
        const unsigned gy = cdiv((long long)K * C, (long long)TPB * HPAIRS);
        const size_t smem = (size_t)HPB * (K + C) * sizeof(float);
        DISPATCH_T(dtype, head_bwd_dw_k<T><<<dim3(gx, gy), TPB, smem, st>>>(gout, out, (const T*)x, slope, dw, db, M, H * W, C, K));
        return kp_check_launch("head_bwd_dw");
    }
    return KP_OK;
}

int kp_l2_loss(const float* x, const float* t, const float* mask, double* loss_sum, float* grad, float scale, size_t n,
               void* stream) {
    KP_REQUIRE(n > 0, "kp_l2_loss: empty");
    unsigned grid = cdiv(n, TPB * 4);
    if (grid > 2048) grid = 2048;
    l2_loss_k<<<grid, TPB, 0, (hipStream_t)stream>>>(x, t, mask, loss_sum, grad, scale, n);
    return kp_check_launch("l2_loss");
}

int kp_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                 float bc1, float bc2, void* stream) {
    KP_REQUIRE(n > 0, "kp_adam_step: empty");
    adam_k<<<cdiv(n, TPB), TPB, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, b1, b2, eps, bc1, sqrtf(bc2));
    return kp_check_launch("adam_step");
}

}  // extern "C"
