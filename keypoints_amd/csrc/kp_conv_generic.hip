// Generic (any C, any K, fp32 or bf16 storage, fp32 VALU accumulate) implicit-GEMM
// convolution kernels.  They are the strict-fp32 parity path and the fallback for
// shapes the MFMA kernels (kp_conv_mfma.hip) do not take (C = 1/3/74, K = 1/3/10...).
//
// All convolutions here are "valid" correlations over an image whose border is
// already physically present (replicated for the forward pass, zero for the
// data-gradient), so the inner loops carry no boundary logic: SURVEY.md 8(a) a1/a2.
#include "kp_common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, TPB = 256;

// y[m][k] = bias[k] + sum_{tap,c} x[pix(m) + tap][c] * w[k][tap][c]
template <typename T, int R>
__global__ __launch_bounds__(TPB) void conv_valid_generic(const T* __restrict__ x, const T* __restrict__ w,
                                                          const float* __restrict__ bias, T* __restrict__ y,
                                                          float* __restrict__ partials, int N, int Hi, int Wi, int C,
                                                          int K) {
    const int Ho = Hi - R + 1, Wo = Wi - R + 1;
    const long long M = (long long)N * Ho * Wo;
    __shared__ float As[BK][BM + 4];
    __shared__ float Bs[BK][BN + 4];
    __shared__ float red[2][16][BN];

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const long long m0 = (long long)blockIdx.x * BM;
    const int k0 = blockIdx.y * BN;

    // loader role: one pixel / one cout, 4 consecutive channels
    const int lrow = tid >> 2, lc = (tid & 3) * 4;
    const long long lm = m0 + lrow;
    long long a_base = -1;
    if (lm < M) {
        int q = (int)(lm % Wo);
        long long t = lm / Wo;
        int p = (int)(t % Ho);
        int n = (int)(t / Ho);
        a_base = (((long long)n * Hi + p) * Wi + q) * C;
    }
    const int lk = k0 + lrow;
    const long long b_base = (lk < K) ? (long long)lk * R * R * C : -1;

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (int tap = 0; tap < R * R; ++tap) {
        const int r = tap / R, s = tap % R;
        const long long a_tap = ((long long)r * Wi + s) * C;
        const long long b_tap = (long long)tap * C;
        for (int c0 = 0; c0 < C; c0 += BK) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = c0 + lc + j;
                float av = 0.f, bv = 0.f;
                if (c < C) {
                    if (a_base >= 0) av = ld<T>(x + a_base + a_tap + c);
                    if (b_base >= 0) bv = ld<T>(w + b_base + b_tap + c);
                }
                As[lc + j][lrow] = av;
                Bs[lc + j][lrow] = bv;
            }
            __syncthreads();
#pragma unroll
            for (int kk = 0; kk < BK; ++kk) {
                float a[4], b[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = As[kk][ty * 4 + i];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = Bs[kk][tx * 4 + j];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
            }
            __syncthreads();
        }
    }

    float csum[4] = {0.f, 0.f, 0.f, 0.f}, csq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + tx * 4 + j;
            if (k >= K) continue;
            float v = acc[i][j] + (bias ? bias[k] : 0.f);
            st<T>(y + m * K + k, v);
            csum[j] += v;
            csq[j] += v * v;
        }
    }
    if (partials) {   // one row of per-channel (sum, sumsq) per 64-pixel tile: deterministic, no atomics
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            red[0][ty][tx * 4 + j] = csum[j];
            red[1][ty][tx * 4 + j] = csq[j];
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, col = tid % BN;
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += red[which][i][col];
            const int k = k0 + col;
            if (k < K) partials[((long long)blockIdx.x * 2 + which) * K + k] = s;
        }
    }
}

// scratch[k][tap][c] += sum_m dy[m][k] * x[pix(m) + tap][c]
template <typename T, int R>
__global__ __launch_bounds__(TPB) void conv_wgrad_generic(const T* __restrict__ x, const T* __restrict__ dy,
                                                          float* __restrict__ scratch, int N, int Ho, int Wo, int C,
                                                          int K, int dy_pad, int pix_per_block) {
    const int Hi = Ho + R - 1, Wi = Wo + R - 1;
    const int Hd = Ho + 2 * dy_pad, Wd = Wo + 2 * dy_pad;
    const long long M = (long long)N * Ho * Wo;
    constexpr int PB = 16;
    __shared__ float Ds[PB][BM + 4];   // dy  [pixel][k]
    __shared__ float Xs[PB][BN + 4];   // x   [pixel][c]

    const int tid = threadIdx.x;
    const int tx = tid & 15, ty = tid >> 4;
    const int ctiles = (C + BN - 1) / BN;
    const int tap = blockIdx.x / ctiles;
    const int c0 = (blockIdx.x % ctiles) * BN;
    const int k0 = blockIdx.y * BM;
    const int r = tap / R, s = tap % R;
    const long long mbeg = (long long)blockIdx.z * pix_per_block;
    const long long mend = (mbeg + pix_per_block < M) ? mbeg + pix_per_block : M;

    const int lp = tid >> 4, lv = (tid & 15) * 4;   // loader: pixel lp, 4 consecutive channels from lv

    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

    for (long long mb = mbeg; mb < mend; mb += PB) {
        const long long m = mb + lp;
        long long xo = -1, dyo = -1;
        if (m < mend) {
            int q = (int)(m % Wo);
            long long t = m / Wo;
            int p = (int)(t % Ho);
            int n = (int)(t / Ho);
            xo = (((long long)n * Hi + p + r) * Wi + q + s) * C;
            dyo = (((long long)n * Hd + p + dy_pad) * Wd + q + dy_pad) * K;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + lv + j, c = c0 + lv + j;
            Ds[lp][lv + j] = (dyo >= 0 && k < K) ? ld<T>(dy + dyo + k) : 0.f;
            Xs[lp][lv + j] = (xo >= 0 && c < C) ? ld<T>(x + xo + c) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < PB; ++pp) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = Ds[pp][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = Xs[pp][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = k0 + ty * 4 + i;
        if (k >= K) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + tx * 4 + j;
            // this pixel split's partial gradient (plain store: every (k, tap, c) of a partial has exactly one writer)
            if (c < C) scratch[(long long)blockIdx.z * K * R * R * C + ((long long)k * R * R + tap) * C + c] = acc[i][j];
        }
    }
}

template <typename T>
int launch_fwd(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi, int C,
               int K, int R, hipStream_t st) {
    KP_QUERY_RETURN();
    const long long M = (long long)N * (Hi - R + 1) * (Wi - R + 1);
    dim3 grid(cdiv(M, BM), cdiv(K, BN));
    if (R == 3)
        conv_valid_generic<T, 3><<<grid, TPB, 0, st>>>((const T*)x, (const T*)w, bias, (T*)y, stats, N, Hi, Wi, C, K);
    else
        conv_valid_generic<T, 1><<<grid, TPB, 0, st>>>((const T*)x, (const T*)w, bias, (T*)y, stats, N, Hi, Wi, C, K);
    return kp_check_launch("conv_valid_generic");
}

// pixel splits of the weight-gradient launch = partial gradients in its scratch
static long long wgrad_plan(int N, int Ho, int Wo, int C, int K, int R, long long& ppb) {
    const long long M = (long long)N * Ho * Wo;
    const int ctiles = (C + BN - 1) / BN, ktiles = (K + BM - 1) / BM;
    const long long tiles = (long long)ctiles * ktiles * R * R;
    long long splits = (2048 + tiles - 1) / tiles;     // aim for ~2048 blocks, at least 256 pixels per block
    ppb = (M + splits - 1) / splits;
    if (ppb < 256) ppb = 256;
    ppb = (ppb + 15) / 16 * 16;
    return (M + ppb - 1) / ppb;
}

template <typename T>
int launch_wgrad(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int R, int dy_pad,
                 hipStream_t st) {
    const int ctiles = (C + BN - 1) / BN, ktiles = (K + BM - 1) / BM;
    long long ppb;
    const long long splits = wgrad_plan(N, Ho, Wo, C, K, R, ppb);
    dim3 grid((unsigned)(ctiles * R * R), (unsigned)ktiles, (unsigned)splits);
    if (R == 3)
        conv_wgrad_generic<T, 3><<<grid, TPB, 0, st>>>((const T*)x, (const T*)dy, scratch, N, Ho, Wo, C, K, dy_pad,
                                                       (int)ppb);
    else
        conv_wgrad_generic<T, 1><<<grid, TPB, 0, st>>>((const T*)x, (const T*)dy, scratch, N, Ho, Wo, C, K, dy_pad,
                                                       (int)ppb);
    return kp_check_launch("conv_wgrad_generic");
}

}  // namespace

int kp_conv_fwd_generic(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi,
                        int C, int K, int R, int dtype, hipStream_t st) {
    return dtype == KP_F32 ? launch_fwd<float>(x, w, bias, y, stats, N, Hi, Wi, C, K, R, st)
                           : launch_fwd<h16_t>(x, w, bias, y, stats, N, Hi, Wi, C, K, R, st);
}

long long kp_conv_wgrad_generic_partials(int N, int Ho, int Wo, int C, int K, int R) {
    long long ppb;
    return wgrad_plan(N, Ho, Wo, C, K, R, ppb);
}

int kp_conv_wgrad_generic(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int R,
                          int dy_pad, int dtype, hipStream_t st) {
    return dtype == KP_F32 ? launch_wgrad<float>(x, dy, scratch, N, Ho, Wo, C, K, R, dy_pad, st)
                           : launch_wgrad<h16_t>(x, dy, scratch, N, Ho, Wo, C, K, R, dy_pad, st);
}
