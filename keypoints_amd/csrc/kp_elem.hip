// Small kernels around the convolutions: layout changes, weight packs, the 1x1 head, L2 loss and Adam
// (BatchNorm / activation / pool / upsample live in kp_bn.hip).
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
__global__ void nchw_to_nhwc_k(const float* __restrict__ src, T* __restrict__ dst, int N, int C, int Cp, int H, int W,
                               int p) {
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    const long long total = (long long)N * Hp * Wp;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int j = (int)(t % Wp);
    const int i = (int)((t / Wp) % Hp);
    const int n = (int)(t / ((long long)Wp * Hp));
    const int h = clampi(i - p, 0, H - 1), w = clampi(j - p, 0, W - 1);
    const float* s = src + ((long long)n * C * H + h) * W + w;
    T* d = dst + t * Cp;
    for (int c = 0; c < C; ++c) st<T>(d + c, s[(long long)c * H * W]);
    for (int c = C; c < Cp; ++c) st<T>(d + c, 0.f);   // channel padding up to an MFMA-friendly width
}

// 16-bit storage, Cp a multiple of 8: one thread per (padded pixel, 8-channel chunk), lanes = adjacent pixels.  Eight coalesced
// plane loads, ONE 16-byte store -- the per-pixel kernel above walks all C channels with 2-byte stores at a stride of the pixel
// pitch (a scatter: 68 us for the 256 x 16 x 84 x 84 decoder input of the Pong-class nets, 2.6 TB/s of a 174 MB job).
__global__ __launch_bounds__(256) void nchw_to_nhwc_v8_k(const float* __restrict__ src, h16_t* __restrict__ dst, int N, int C, int Cp,
                                                         int H, int W, int p) {
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    const long long total = (long long)N * Hp * Wp;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c0 = blockIdx.y * 8;
    const int j = (int)(t % Wp);
    const int i = (int)((t / Wp) % Hp);
    const int n = (int)(t / ((long long)Wp * Hp));
    const int h = clampi(i - p, 0, H - 1), w = clampi(j - p, 0, W - 1);
    const float* s = src + (((long long)n * C + c0) * H + h) * W + w;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = c0 + e < C ? s[(long long)e * H * W] : 0.f;   // channel padding: zeros
    uint4 o;
    o.x = pack_h16x2(v[0], v[1]); o.y = pack_h16x2(v[2], v[3]); o.z = pack_h16x2(v[4], v[5]); o.w = pack_h16x2(v[6], v[7]);
    *reinterpret_cast<uint4*>(dst + t * Cp + c0) = o;
}

template <typename T>
__global__ void nhwc_grad_to_nchw_k(const T* __restrict__ g, float* __restrict__ dst, int N, int C, int Cp, int H,
                                    int W, int p) {
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
            for (int j = jlo; j <= jhi; ++j) a += ld<T>(g + (((long long)n * Hp + i) * Wp + j) * Cp + c);
        dst[(((long long)n * C + c) * H + h) * W + w] = a;
    }
}

// The same with one thread per (pixel, group of 4 channels): lanes = adjacent pixels, a vector load per folded position,
// four coalesced plane stores -- the per-pixel kernel above walks all C channels serially with 2-byte loads (44 us for the
// decoder's [128][64][16][16] input gradient, 128 workgroups; this one: C / 4 times the threads, 1-4 loads each).
template <typename T>
__global__ void nhwc_grad_to_nchw_v4_k(const T* __restrict__ g, float* __restrict__ dst, int N, int C, int Cp, int H,
                                       int W, int p) {
    const int Wp = W + 2 * p, Hp = H + 2 * p;
    const long long total = (long long)N * H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = blockIdx.y * 4;
    if (t >= total) return;
    const int w = (int)(t % W);
    const int h = (int)((t / W) % H);
    const int n = (int)(t / ((long long)W * H));
    int ilo, ihi, jlo, jhi;
    fold_range(h, H, p, ilo, ihi);
    fold_range(w, W, p, jlo, jhi);
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = ilo; i <= ihi; ++i)
        for (int j = jlo; j <= jhi; ++j) {
            float v[4];
            ld4<T>(g + (((long long)n * Hp + i) * Wp + j) * Cp + c0, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] += v[e];
        }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (c0 + e < C) dst[(((long long)n * C + c0 + e) * H + h) * W + w] = a[e];
}

// interior of a padded T image -> fp32 planes (no fold: the image is a VALUE, e.g. the blended feature map the T form of the keypoint
// bottleneck wrote as the decoder's padded input, which TransporterNet.forward also returns as fp32 NCHW, transporter.py:64)
template <typename T>
__global__ void nhwc_to_nchw_v4_k(const T* __restrict__ src, float* __restrict__ dst, int N, int C, int H, int W, int p) {
    const int Wp = W + 2 * p, Hp = H + 2 * p;
    const long long total = (long long)N * H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = blockIdx.y * 4;
    if (t >= total) return;
    const int w = (int)(t % W);
    const int h = (int)((t / W) % H);
    const int n = (int)(t / ((long long)W * H));
    float v[4];
    ld4<T>(src + (((long long)n * Hp + h + p) * Wp + w + p) * C + c0, v);
#pragma unroll
    for (int e = 0; e < 4; ++e) dst[(((long long)n * C + c0 + e) * H + h) * W + w] = v[e];
}

// Two NCHW fp32 sources as ONE padded NHWC image: channels [0, C0) from s0, [C0, C0 + C1) from s1, zeros up to Cp -- torch.cat((z, m),
// dim=1) in front of the KeyNet decoder (keynet.py:29) folded into the layout change its first layer needs anyway.  One thread per
// (padded pixel, group of 4 channels); lanes = adjacent pixels (coalesced plane loads).  Same values, same rounding as
// kp_nchw_to_nhwc of the concatenated tensor.
template <typename T>
__global__ void nchw2_to_nhwc_k(const float* __restrict__ s0, int C0, const float* __restrict__ s1, int C1, T* __restrict__ dst, int N,
                                int Cp, int H, int W, int p) {
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    const long long total = (long long)N * Hp * Wp;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c0 = blockIdx.y * 4;
    const int j = (int)(t % Wp);
    const int i = (int)((t / Wp) % Hp);
    const int n = (int)(t / ((long long)Wp * Hp));
    const int h = clampi(i - p, 0, H - 1), w = clampi(j - p, 0, W - 1);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        if (c >= Cp) break;
        float v = 0.f;   // channel padding: zeros
        if (c < C0) v = s0[(((long long)n * C0 + c) * H + h) * W + w];
        else if (c < C0 + C1) v = s1[(((long long)n * C1 + (c - C0)) * H + h) * W + w];
        st<T>(dst + t * Cp + c, v);
    }
}

// Its adjoint: the fold of kp_nhwc_grad_to_nchw, channels [0, C0) to d0 and [C0, C0 + C1) to d1 (either may be NULL: that input
// carries no gradient).  Sums in the order of nhwc_grad_to_nchw_k (rows, then columns of the folded range).
template <typename T>
__global__ void nhwc_grad_to_nchw2_k(const T* __restrict__ g, float* __restrict__ d0, int C0, float* __restrict__ d1, int C1, int N,
                                     int Cp, int H, int W, int p) {
    const int Wp = W + 2 * p, Hp = H + 2 * p;
    const long long total = (long long)N * H * W;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = blockIdx.y * 4;
    if (t >= total) return;
    const int w = (int)(t % W);
    const int h = (int)((t / W) % H);
    const int n = (int)(t / ((long long)W * H));
    int ilo, ihi, jlo, jhi;
    fold_range(h, H, p, ilo, ihi);
    fold_range(w, W, p, jlo, jhi);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = c0 + e;
        if (c >= C0 + C1) break;
        float* d = c < C0 ? d0 : d1;
        if (!d) continue;
        float a = 0.f;
        for (int i = ilo; i <= ihi; ++i)
            for (int j = jlo; j <= jhi; ++j) a += ld<T>(g + (((long long)n * Hp + i) * Wp + j) * Cp + c);
        if (c < C0) d[(((long long)n * C0 + c) * H + h) * W + w] = a;
        else d[(((long long)n * C1 + (c - C0)) * H + h) * W + w] = a;
    }
}

// [N][Hs][Ws][C] -> the top-left corner of [N][Hd][Wd][C], the rest zero: pads a map to the pixel-block multiple the tap-reuse weight
// gradient wants (kp_conv_wgrad_pad_hw).  One thread per 16-byte chunk (C * element size a multiple of 16).
__global__ __launch_bounds__(256) void pad_hw_k(const uint4* __restrict__ src, uint4* __restrict__ dst, int Hs, int Ws, int Hd, int Wd,
                                                int cpp, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % cpp);
    const long long pix = t / cpp;
    const int j = (int)(pix % Wd);
    const int i = (int)((pix / Wd) % Hd);
    const long long n = pix / ((long long)Wd * Hd);
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (i < Hs && j < Ws) v = src[((n * Hs + i) * Ws + j) * cpp + c];
    dst[t] = v;
}

// Replicate border of a padded NHWC tensor [N][H+2p][W+2p][C] from its interior (nn.ReplicationPad2d: every ring pixel = the clamped
// interior pixel).  One thread per 16-byte chunk of a RING pixel; the interior is not touched.
__global__ __launch_bounds__(256) void pad_border_k(uint4* __restrict__ buf, int H, int W, int p, int cpp, long long total) {
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    const int Hp = H + 2 * p, Wp = W + 2 * p;
    const int ring = Hp * Wp - H * W;
    const int c = (int)(t % cpp);
    const long long rp = t / cpp;
    const int r = (int)(rp % ring);
    const long long n = rp / ring;
    int i, j;   // ring pixel r: the p top rows, the p bottom rows, then the left / right columns of the middle rows
    if (r < p * Wp) { i = r / Wp; j = r - i * Wp; }
    else if (r < 2 * p * Wp) { const int q = r - p * Wp; i = H + p + q / Wp; j = q % Wp; }
    else { const int q = r - 2 * p * Wp; i = p + q / (2 * p); const int jj = q % (2 * p); j = jj < p ? jj : W + jj; }
    const int si = min(max(i, p), H + p - 1), sj = min(max(j, p), W + p - 1);
    buf[((n * Hp + i) * Wp + j) * cpp + c] = buf[((n * Hp + si) * Wp + sj) * cpp + c];
}

// ---------------------------------------------------------------- weights
template <typename T>
__global__ void pack_weights_k(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int K, int C,
                               int Cp, int R) {
    const long long total = (long long)K * R * R * Cp;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int c = (int)(t % Cp);
    const int s = (int)((t / Cp) % R);
    const int r = (int)((t / ((long long)Cp * R)) % R);
    const int k = (int)(t / ((long long)Cp * R * R));
    const float v = c < C ? w[(((long long)k * C + c) * R + r) * R + s] : 0.f;   // zero weights for padded channels
    st<T>(wf + t, v);
    if (wd) st<T>(wd + (((long long)c * R + (R - 1 - r)) * R + (R - 1 - s)) * K + k, v);
}

// All 3x3 layers of a Unit in ONE launch (a Unit forward packed its 8-9 layers with as many ~12 us launches).
constexpr int PACK_MAX = 24;
struct PackBatch {
    const float* w[PACK_MAX]; void* wf[PACK_MAX]; void* wd[PACK_MAX];
    int K[PACK_MAX], C[PACK_MAX], Cp[PACK_MAX];
    unsigned first_block[PACK_MAX + 1];
    int n;
};
// Through an LDS tile of 32 output channels x 32 input channels x 9 taps: the per-element kernel (pack_weights_k) reads the
// source at a 36-byte stride and scatters the data-gradient pack in 2-byte stores at a stride of 9 K elements (72 us for the
// 8-9 M weights of a model-F Unit).  Here the source rows come in 1152-byte runs, both packs leave in 64-byte runs; the tile
// pitch is odd (289 words), so the c-major and the k-major sweeps over it are both conflict-free.
constexpr int PT = 32, PPITCH = PT * 9 + 1;
// NT = 1024 for the nets of a few tiles (a Pong-class Transporter is 8 tiles: the launch takes as long as ONE workgroup walks its
// 3 x 9 216 tile elements -- 36 trips per sweep with 256 threads, 20 us at the start of every forward)
template <typename T, int NT = 256>
__global__ __launch_bounds__(NT) void pack_weights_tile_k(const PackBatch B) {
    __shared__ float tile[PT * PPITCH];
    int l = 0;
    while (l + 1 < B.n && blockIdx.x >= B.first_block[l + 1]) ++l;   // (uniform; n <= 24)
    const int K = B.K[l], C = B.C[l], Cp = B.Cp[l];
    const int ctiles = (Cp + PT - 1) / PT;
    const int tb = blockIdx.x - B.first_block[l];
    const int k0 = (tb / ctiles) * PT, c0 = (tb % ctiles) * PT;
    const int tid = threadIdx.x;
    const int cw = (C - c0 < PT ? C - c0 : PT) * 9;          // source floats per k row in this tile (may be <= 0: padding tile)
    const float* __restrict__ w = B.w[l];
    // full tiles of 4-aligned layers (every matrix-core layer): 16-byte source loads, 8-byte stores of four packed elements (the
    // 4-byte loads / 2-byte stores below: 107 us for model F's 17 M weights, 1.3 TB/s)
    const bool wide = sizeof(T) == 2 && cw == PT * 9 && k0 + PT <= K && c0 + PT <= Cp && (C & 3) == 0 && (Cp & 3) == 0 && (K & 3) == 0;
    if (wide) {
        for (int e = tid; e < PT * (PT * 9 / 4); e += NT) {
            const int kk = e / (PT * 9 / 4), j4 = e - kk * (PT * 9 / 4);
            const float4 v = *reinterpret_cast<const float4*>(w + ((long long)(k0 + kk) * C + c0) * 9 + j4 * 4);
            float* t = tile + kk * PPITCH + j4 * 4;
            t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
        }
    } else {
        for (int e = tid; e < PT * PT * 9; e += NT) {
            const int kk = e / (PT * 9), j = e - kk * (PT * 9);   // j = cc * 9 + tap
            float v = 0.f;
            if (k0 + kk < K && j < cw) v = w[((long long)(k0 + kk) * C + c0) * 9 + j];
            tile[kk * PPITCH + j] = v;
        }
    }
    __syncthreads();
    T* __restrict__ wf = (T*)B.wf[l];
    T* __restrict__ wd = (T*)B.wd[l];
    if (wide) {
        for (int e = tid; e < PT * 9 * (PT / 4); e += NT) {        // wf[k][tap][c .. c + 3]: lanes walk c
            const int c4 = e % (PT / 4), tap = (e / (PT / 4)) % 9, kk = e / ((PT / 4) * 9);
            const float* t = tile + kk * PPITCH + c4 * 4 * 9 + tap;
            uint2 pk;
            pk.x = pack_h16x2(t[0], t[9]); pk.y = pack_h16x2(t[18], t[27]);
            *reinterpret_cast<uint2*>(wf + ((long long)(k0 + kk) * 9 + tap) * Cp + c0 + c4 * 4) = pk;
        }
        if (wd) {
            for (int e = tid; e < PT * 9 * (PT / 4); e += NT) {    // wd[c][8 - tap][k .. k + 3]: lanes walk k
                const int k4 = e % (PT / 4), tap = (e / (PT / 4)) % 9, cc = e / ((PT / 4) * 9);
                const float* t = tile + k4 * 4 * PPITCH + cc * 9 + tap;
                uint2 pk;
                pk.x = pack_h16x2(t[0], t[PPITCH]); pk.y = pack_h16x2(t[2 * PPITCH], t[3 * PPITCH]);
                *reinterpret_cast<uint2*>(wd + ((long long)(c0 + cc) * 9 + (8 - tap)) * K + k0 + k4 * 4) = pk;
            }
        }
        return;
    }
    for (int e = tid; e < PT * 9 * PT; e += NT) {             // wf[k][tap][c]: lanes walk c
        const int cc = e % PT, tap = (e / PT) % 9, kk = e / (PT * 9);
        if (k0 + kk < K && c0 + cc < Cp) st<T>(wf + ((long long)(k0 + kk) * 9 + tap) * Cp + c0 + cc, tile[kk * PPITCH + cc * 9 + tap]);
    }
    if (wd) {
        for (int e = tid; e < PT * 9 * PT; e += NT) {         // wd[c][8 - tap][k]: lanes walk k
            const int kk = e % PT, tap = (e / PT) % 9, cc = e / (PT * 9);
            if (k0 + kk < K && c0 + cc < Cp) st<T>(wd + ((long long)(c0 + cc) * 9 + (8 - tap)) * K + k0 + kk, tile[kk * PPITCH + cc * 9 + tap]);
        }
    }
}

// dw[k][c][r][s] += sum_p scratch[p][k][r][s][c].  A workgroup = 256 / Q element groups (VEC consecutive c of one
// (k, r, s) of the scratch layout: coalesced 16-byte lanes) x Q partial lanes; lane q sums the partials p = q, q + Q, ...
// (4 loads in flight), the Q lanes are combined through LDS in a fixed order -> deterministic.  Q = 4, or 16 / 64 (a 1024-thread workgroup) for the few-KB
// gradients of the Pong-class layers with hundreds of partials (18 workgroups walking 128 dependent rounds each took 14 us).
template <int VEC, int Q, int NT = 256>
__global__ __launch_bounds__(NT) void wgrad_unpack_k(const float* __restrict__ scratch, int partials,
                                                     float* __restrict__ dw, int K, int Cd, int C, int R) {
    // C = channels of the scratch layout (possibly padded), Cd <= C = channels of dw
    constexpr int EG = NT / Q;
    const long long total = (long long)K * C * R * R;
    const int el = threadIdx.x % EG, q = threadIdx.x / EG;
    const long long t = ((long long)blockIdx.x * EG + el) * VEC;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    if (t < total) {
        int p = q;
        if (VEC == 4) {
            for (; p + 3 * Q < partials; p += 4 * Q) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(scratch + (long long)(p + Q * u) * total + t);
#pragma unroll
                for (int u = 0; u < 4; ++u) { acc[0] += v[u].x; acc[1] += v[u].y; acc[2] += v[u].z; acc[3] += v[u].w; }
            }
        }
        if (VEC == 1) {   // many small partials (first layer: one per workgroup): 8 loads in flight per lane
            for (; p + 7 * Q < partials; p += 8 * Q) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = scratch[(long long)(p + Q * u) * total + t];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc[0] += v[u];
            }
        }
        for (; p < partials; p += Q)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[e] += scratch[(long long)p * total + t + e];
    }
    __shared__ float red[Q][EG][VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[q][el][e] = acc[e];
    __syncthreads();
    if (q == 0 && t < total) {
        const int c = (int)(t % C);
        const int s = (int)((t / C) % R);
        const int r = (int)((t / ((long long)C * R)) % R);
        const int k = (int)(t / ((long long)C * R * R));
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            if (c + e < Cd) {
                float sum = ((red[0][el][e] + red[1][el][e]) + red[2][el][e]) + red[3][el][e];
#pragma unroll
                for (int g = 4; g < Q; g += 4) sum += ((red[g][el][e] + red[g + 1][el][e]) + red[g + 2][el][e]) + red[g + 3][el][e];
                dw[(((long long)k * Cd + c + e) * R + r) * R + s] += sum;
            }
    }
}

// The same sum for the MFMA kernels' layouts (C % 64 == 0), with both sides coalesced: a workgroup owns output channel k and
// a 128-channel chunk; for fixed k both layouts are one contiguous run of R*R*C floats, so the (tap, c) -> (c, tap)
// transposition happens in LDS.  Partials are summed per element in the same order as above: 4 interleaved running
// sums p = q, q+4, ... combined as ((s0 + s1) + s2) + s3 -- bitwise the same gradients.
__global__ __launch_bounds__(256) void wgrad_unpack_tile_k(const float* __restrict__ scratch, int partials,
                                                           float* __restrict__ dw, int K, int Cd, int C, int R) {
    constexpr int CH = 128;
    __shared__ float tile[9 * CH];
    const int RR = R * R;
    const int k = blockIdx.x, c0 = blockIdx.y * CH;
    const int cw = C - c0 < CH ? C - c0 : CH;             // channels of this chunk
    const long long pstride = (long long)K * RR * C;
    const float* base = scratch + ((long long)k * RR) * C + c0;
    // (four channels per lane with 16-byte loads: 20.9 -> 29.9 us -- 288 lanes' worth of work per workgroup leaves too few loads in flight)
    for (int e = threadIdx.x; e < RR * cw; e += 256) {
        const int rs = e / cw, cl = e - rs * cw;
        const float* src = base + (long long)rs * C + cl;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        int p = 0;
        for (; p + 8 <= partials; p += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(long long)(p + u) * pstride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u & 3] += v[u];
        }
        for (; p < partials; ++p) s[p & 3] += src[(long long)p * pstride];
        tile[rs * CH + cl] = ((s[0] + s[1]) + s[2]) + s[3];
    }
    __syncthreads();
    const int cd = Cd - c0 < cw ? Cd - c0 : cw;           // channels that exist in dw (the scratch may be padded)
    float* dst = dw + ((long long)k * Cd + c0) * RR;
    for (int o = threadIdx.x; o < cd * RR; o += 256) {
        const int cl = o / RR, rs = o - cl * RR;
        dst[o] += tile[rs * CH + cl];
    }
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
                                                     float* __restrict__ db, float* __restrict__ ws, long long M,
                                                     int HW, int C, int K) {
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
        if (pr < npairs) {
            if (ws) ws[(long long)blockIdx.x * (npairs + K) + pr] = acc[i];   // this workgroup column's partial
            else atomicAdd(dw + pr, acc[i]);
        }
    }
    if (blockIdx.y == 0 && threadIdx.x < K) {
        if (ws) ws[(long long)blockIdx.x * (npairs + K) + npairs + threadIdx.x] = dbacc;
        else if (db) atomicAdd(db + threadIdx.x, dbacc);
    }
}

// dw[i] += sum_p ws[p][i], db[k] += sum_p ws[p][K*C + k], p ascending: the fixed-order second stage of the head's
// weight gradient (first stage: one partial per workgroup / pixel split, plain stores)
template <int PL>
__global__ __launch_bounds__(16 * PL) void head_dw_reduce_k(const float* __restrict__ ws, int P, long long n, long long KC,
                                                            float* __restrict__ dw, float* __restrict__ db) {
    // a workgroup owns 16 consecutive elements (one 64-byte sector per partial); PL lanes of partials per element, each summing
    // p = pl, pl + PL, ... in order (4 / 8 loads in flight), then the PL lane sums in order: a fixed summation tree.  PL = 64 for
    // the Pong-class heads (2 048 partials of 528 floats: with 16 lanes a thread walked 32 dependent rounds, 12 us per launch,
    // three launches on the critical path of a config-2 step)
    constexpr int U = PL == 16 ? 4 : 8;
    __shared__ float red[PL][17];
    const int e = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const long long i = (long long)blockIdx.x * 16 + e;
    float s[U];
#pragma unroll
    for (int u = 0; u < U; ++u) s[u] = 0.f;
    if (i < n) {
        int p = pl;
        for (; p + (U - 1) * PL < P; p += U * PL) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ws[(long long)(p + PL * u) * n + i];
#pragma unroll
            for (int u = 0; u < U; ++u) s[u] += v[u];
        }
        for (int u = 0; p < P; p += PL, ++u) s[u & (U - 1)] += ws[(long long)p * n + i];
    }
    float t = ((s[0] + s[1]) + s[2]) + s[3];
    if (U == 8) t += ((s[4] + s[5]) + s[6]) + s[7];
    red[pl][e] = t;
    __syncthreads();
    if (pl == 0 && i < n) {
        float tot = 0.f;
#pragma unroll
        for (int r = 0; r < PL; ++r) tot += red[r][e];
        if (i < KC) dw[i] += tot;
        else if (db) db[i - KC] += tot;
    }
}
static void head_dw_reduce(const float* ws, long long parts, long long n, long long KC, float* dw, float* db, hipStream_t st) {
    if (parts > 256 && n <= 4096) head_dw_reduce_k<64><<<cdiv(n, 16), 1024, 0, st>>>(ws, (int)parts, n, KC, dw, db);
    else head_dw_reduce_k<16><<<cdiv(n, 16), 256, 0, st>>>(ws, (int)parts, n, KC, dw, db);
}

// ---------------------------------------------------------------- loss / Adam
// 16-byte lanes (n4 = number of whole float4 groups when all pointers are 16-byte aligned, else 0), the remainder element
// by element; per-thread fp64 partial sums, one fp64 atomic per workgroup (<= 512 of them: at 2048 the atomics on one
// address were most of the kernel's 40 us at 128 x 3 x 128 x 128).
// loss_out != NULL (kp_l2_loss_mean): loss_sum is a persistent workspace of two doubles, [0] the sum and [1] (as an integer) the
// number of workgroups that have added theirs; the LAST one to arrive writes loss_out = (float)(sum / n) -- the mean in fp64, rounded
// once -- and leaves both zero for the next call: no zero-fill and no division launch around the kernel.
__global__ __launch_bounds__(TPB) void l2_loss_k(const float* __restrict__ x, const float* __restrict__ t,
                                                 const float* __restrict__ mask, double* __restrict__ loss_sum,
                                                 float* __restrict__ grad, float scale, size_t n, size_t n4,
                                                 float* __restrict__ loss_out = nullptr) {
    double acc = 0.0;
    const size_t gtid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = gtid; i < n4; i += stride) {
        const float4 a = reinterpret_cast<const float4*>(x)[i], b = reinterpret_cast<const float4*>(t)[i];
        const float4 m = mask ? reinterpret_cast<const float4*>(mask)[i] : make_float4(1.f, 1.f, 1.f, 1.f);
        const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
        acc += ((double)(d0 * d0 * m.x) + (double)(d1 * d1 * m.y)) + ((double)(d2 * d2 * m.z) + (double)(d3 * d3 * m.w));
        if (grad)
            reinterpret_cast<float4*>(grad)[i] = make_float4(2.f * d0 * m.x * scale, 2.f * d1 * m.y * scale,
                                                             2.f * d2 * m.z * scale, 2.f * d3 * m.w * scale);
    }
    for (size_t i = n4 * 4 + gtid; i < n; i += stride) {
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
        if (loss_out) {
            __threadfence();
            unsigned long long* cnt = reinterpret_cast<unsigned long long*>(loss_sum + 1);
            if (atomicAdd(cnt, 1ull) == (unsigned long long)gridDim.x - 1) {
                __threadfence();
                const double tot = atomicAdd(loss_sum, 0.0);   // (device-scope read of the finished sum)
                loss_out[0] = (float)(tot / (double)n);
                atomicExch(reinterpret_cast<unsigned long long*>(loss_sum), 0ull);
                atomicExch(cnt, 0ull);
            }
        }
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

// y = x * s[0] (s on the device: the upstream gradient of the scalar loss, no host round trip)
__global__ void scale_k(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, size_t n) {
    const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float f = s[0];
    if (i + 4 <= n) {
        float4 v = *reinterpret_cast<const float4*>(x + i);
        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
        *reinterpret_cast<float4*>(y + i) = v;
    } else {
        for (size_t j = i; j < n; ++j) y[j] = x[j] * f;
    }
}

}  // namespace

#define DISPATCH_T(dtype, ...)                         \
    do {                                               \
        if ((dtype) == KP_F32) { using T = float; __VA_ARGS__; } \
        else { using T = h16_t; __VA_ARGS__; }        \
    } while (0)

int kp_head_fwd_fast(const void*, const float*, const float*, float, float*, long long, int, int, int, hipStream_t, const KpBnIn* = nullptr);
int kp_head_fwd_mfma(const void*, const float*, const float*, float, float*, long long, int, int, int, hipStream_t, const KpBnIn* = nullptr, const void* x1 = nullptr, float* out1 = nullptr);
int kp_head_dx_mfma(const float*, const float*, const float*, float, void*, long long, int, int, int, hipStream_t);
int kp_head_dw_mfma(const float*, const float*, const void*, float, float*, float*, float*, long long, int, int, int, hipStream_t, const KpBnIn* = nullptr);
bool kp_head_small_shape(long long, int, int);
bool kp_head_mfma_bnin_shape(long long, int, int, int);
bool kp_head_mfma_shape(long long, int, int, int);
bool kp_head_mfma_fwd_shape(long long, int, int, int);
long long kp_head_dw_parts_mfma(long long, int, int, int);
long long kp_head_dw_parts_fast(long long, int, int);
int kp_head_dx_fast(const float*, const float*, const float*, float, void*, long long, int, int, int, hipStream_t);
int kp_head_dw_fast(const float*, const float*, const void*, float, float*, float*, float*, long long, int, int, int, hipStream_t, const KpBnIn* = nullptr);
bool kp_head_t_shape(long long, int, int, int);
int kp_head_fwd_mfma_t(const void*, const float*, const float*, float, void*, long long, int, int, int, hipStream_t, const KpBnIn*, const void* x1 = nullptr, void* out1 = nullptr);
int kp_head_dx_mfma_t(const void*, const void*, const float*, float, void*, long long, int, int, int, hipStream_t);
int kp_head_dw_mfma_t(const void*, const void*, const void*, float, float*, float*, float*, long long, int, int, int, hipStream_t, const KpBnIn*);

extern "C" {

int kp_nchw_to_nhwc(const float* src, void* dst, int N, int C, int Cp, int H, int W, int pad, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && C > 0 && Cp >= C && H > 0 && W > 0 && pad >= 0, "kp_nchw_to_nhwc: bad dims");
    kp_count_bytes(4.0 * N * C * H * W + (dtype == KP_F32 ? 4.0 : 2.0) * N * (H + 2 * pad) * (W + 2 * pad) * Cp);
    const long long total = (long long)N * (H + 2 * pad) * (W + 2 * pad);
    if (dtype == KP_H16 && Cp % 8 == 0 && Cp / 8 <= 65535 && ((size_t)dst & 15) == 0) {   // (same values, same rounding: pack_h16x2 = st<h16_t>)
        nchw_to_nhwc_v8_k<<<dim3(cdiv(total, 256), Cp / 8), 256, 0, (hipStream_t)stream>>>(src, (h16_t*)dst, N, C, Cp, H, W, pad);
        return kp_check_launch("nchw_to_nhwc_v8");
    }
    DISPATCH_T(dtype, nchw_to_nhwc_k<T><<<cdiv(total, TPB), TPB, 0, (hipStream_t)stream>>>(src, (T*)dst, N, C, Cp, H, W, pad));
    return kp_check_launch("nchw_to_nhwc");
}

int kp_nchw2_to_nhwc(const float* src0, int C0, const float* src1, int C1, void* dst, int N, int Cp, int H, int W, int pad, int dtype,
                     void* stream) {
    KP_REQUIRE(src0 && src1 && dst && N > 0 && C0 > 0 && C1 > 0 && Cp >= C0 + C1 && H > 0 && W > 0 && pad >= 0 && cdiv(Cp, 4) <= 65535,
               "kp_nchw2_to_nhwc: bad args");
    kp_count_bytes(4.0 * N * (C0 + C1) * H * W + (dtype == KP_F32 ? 4.0 : 2.0) * N * (H + 2 * pad) * (W + 2 * pad) * Cp);
    const long long total = (long long)N * (H + 2 * pad) * (W + 2 * pad);
    DISPATCH_T(dtype, nchw2_to_nhwc_k<T><<<dim3(cdiv(total, TPB), cdiv(Cp, 4)), TPB, 0, (hipStream_t)stream>>>(src0, C0, src1, C1, (T*)dst, N, Cp, H, W, pad));
    return kp_check_launch("nchw2_to_nhwc");
}

int kp_nhwc_grad_to_nchw2(const void* gsrc, float* dst0, int C0, float* dst1, int C1, int N, int Cp, int H, int W, int pad, int dtype,
                          void* stream) {
    KP_REQUIRE(gsrc && (dst0 || dst1) && N > 0 && C0 > 0 && C1 > 0 && Cp >= C0 + C1 && H > 0 && W > 0 && pad >= 0 && cdiv(C0 + C1, 4) <= 65535,
               "kp_nhwc_grad_to_nchw2: bad args");
    kp_count_bytes(4.0 * N * ((dst0 ? C0 : 0) + (dst1 ? C1 : 0)) * H * W + (dtype == KP_F32 ? 4.0 : 2.0) * N * (H + 2 * pad) * (W + 2 * pad) * Cp);
    const long long total = (long long)N * H * W;
    DISPATCH_T(dtype, nhwc_grad_to_nchw2_k<T><<<dim3(cdiv(total, TPB), cdiv(C0 + C1, 4)), TPB, 0, (hipStream_t)stream>>>((const T*)gsrc, dst0, C0, dst1, C1, N, Cp, H, W, pad));
    return kp_check_launch("nhwc_grad_to_nchw2");
}

#ifndef KP_F16_BUILD   // bytes only: defined once, by the bf16 build of this source
int kp_pad_hw(const void* src, void* dst, int N, int Hs, int Ws, int Hd, int Wd, int C, int dtype, void* stream) {
    const int esz = dtype == KP_F32 ? 4 : 2;
    KP_REQUIRE(src && dst && N > 0 && Hs > 0 && Ws > 0 && Hd >= Hs && Wd >= Ws && C > 0 && (C * esz) % 16 == 0 && ((((size_t)src | (size_t)dst) & 15) == 0),
               "kp_pad_hw: bad args (C * element size a multiple of 16, 16-byte aligned tensors)");
    const int cpp = C * esz / 16;
    const long long total = (long long)N * Hd * Wd * cpp;
    kp_count_bytes((double)esz * C * ((double)N * Hs * Ws + (double)N * Hd * Wd));
    pad_hw_k<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>((const uint4*)src, (uint4*)dst, Hs, Ws, Hd, Wd, cpp, total);
    return kp_check_launch("pad_hw");
}
#endif

#ifndef KP_F16_BUILD   // bytes only
int kp_pad_border(void* buf, int N, int H, int W, int C, int pad, int dtype, void* stream) {
    const int esz = dtype == KP_F32 ? 4 : 2;
    KP_REQUIRE(buf && N > 0 && H > 0 && W > 0 && C > 0 && pad >= 0 && (C * esz) % 16 == 0 && (((size_t)buf & 15) == 0),
               "kp_pad_border: bad args (C * element size a multiple of 16, a 16-byte aligned tensor)");
    if (pad == 0) return KP_OK;
    const int cpp = C * esz / 16;
    const long long ring = (long long)(H + 2 * pad) * (W + 2 * pad) - (long long)H * W;
    const long long total = (long long)N * ring * cpp;
    kp_count_bytes(2.0 * esz * C * (double)N * ring);
    pad_border_k<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>((uint4*)buf, H, W, pad, cpp, total);
    return kp_check_launch("pad_border");
}
#endif

int kp_nhwc_grad_to_nchw(const void* gsrc, float* dst, int N, int C, int Cp, int H, int W, int pad, int dtype,
                         void* stream) {
    KP_REQUIRE(N > 0 && C > 0 && Cp >= C && H > 0 && W > 0 && pad >= 0, "kp_nhwc_grad_to_nchw: bad dims");
    kp_count_bytes(4.0 * N * C * H * W + (dtype == KP_F32 ? 4.0 : 2.0) * N * (H + 2 * pad) * (W + 2 * pad) * Cp);
    const long long total = (long long)N * H * W;
    if (Cp % 4 == 0 && C > 4 && Cp / 4 <= 65535) {   // same sums in the same order (rows, then columns of the folded range)
        DISPATCH_T(dtype, nhwc_grad_to_nchw_v4_k<T><<<dim3(cdiv(total, TPB), Cp / 4), TPB, 0, (hipStream_t)stream>>>((const T*)gsrc, dst, N, C, Cp, H, W, pad));
        return kp_check_launch("nhwc_grad_to_nchw_v4");
    }
    DISPATCH_T(dtype, nhwc_grad_to_nchw_k<T><<<cdiv(total, TPB), TPB, 0, (hipStream_t)stream>>>((const T*)gsrc, dst, N, C, Cp, H, W, pad));
    return kp_check_launch("nhwc_grad_to_nchw");
}

int kp_pack_weights(const float* w, void* w_fwd, void* w_dgrad, int K, int C, int Cp, int R, int dtype, void* stream) {
    KP_REQUIRE(K > 0 && C > 0 && Cp >= C && (R == 1 || R == 3), "kp_pack_weights: bad dims");
    kp_count_bytes(4.0 * K * C * R * R + (dtype == KP_F32 ? 4.0 : 2.0) * (w_dgrad ? 2.0 : 1.0) * K * R * R * Cp);
    const long long total = (long long)K * Cp * R * R;
    DISPATCH_T(dtype, pack_weights_k<T><<<cdiv(total, TPB), TPB, 0, (hipStream_t)stream>>>(w, (T*)w_fwd, (T*)w_dgrad, K, C, Cp, R));
    return kp_check_launch("pack_weights");
}

int kp_pack_weights_batch(int n, const float* const* w, void* const* w_fwd, void* const* w_dgrad, const int* K, const int* C,
                          const int* Cp, int dtype, void* stream) {
    KP_REQUIRE(n >= 1 && n <= PACK_MAX && w && w_fwd && w_dgrad && K && C && Cp, "kp_pack_weights_batch: bad args (1 <= n <= %d)", PACK_MAX);
    PackBatch B;
    B.n = n;
    unsigned blocks = 0;
    for (int i = 0; i < n; ++i) {
        KP_REQUIRE(K[i] > 0 && C[i] > 0 && Cp[i] >= C[i] && w[i] && w_fwd[i], "kp_pack_weights_batch: bad layer %d", i);
        B.w[i] = w[i]; B.wf[i] = w_fwd[i]; B.wd[i] = w_dgrad[i]; B.K[i] = K[i]; B.C[i] = C[i]; B.Cp[i] = Cp[i];
        B.first_block[i] = blocks;
        blocks += cdiv((size_t)K[i], PT) * cdiv((size_t)Cp[i], PT);
    }
    B.first_block[n] = blocks;
    for (int i = 0; i < n; ++i) kp_count_bytes(4.0 * K[i] * C[i] * 9 + (dtype == KP_F32 ? 4.0 : 2.0) * (w_dgrad[i] ? 2.0 : 1.0) * K[i] * 9 * Cp[i], i == 0);
    if (blocks <= 64) { DISPATCH_T(dtype, (pack_weights_tile_k<T, 1024><<<blocks, 1024, 0, (hipStream_t)stream>>>(B))); }
    else { DISPATCH_T(dtype, pack_weights_tile_k<T><<<blocks, 256, 0, (hipStream_t)stream>>>(B)); }
    return kp_check_launch("pack_weights_batch");
}

#ifndef KP_F16_BUILD   // no dtype: defined once, by the bf16 build of this source
int kp_wgrad_unpack(const float* scratch, int partials, float* dw, int K, int C, int Cp, int R, void* stream) {
    KP_REQUIRE(K > 0 && C > 0 && Cp >= C && (R == 1 || R == 3) && partials >= 1, "kp_wgrad_unpack: bad dims");
    kp_count_bytes(4.0 * partials * K * R * R * Cp + 8.0 * K * C * R * R);
    const long long total = (long long)K * Cp * R * R;
    if (Cp % 64 == 0 && K <= 65535) {
        wgrad_unpack_tile_k<<<dim3(K, cdiv(Cp, 128)), 256, 0, (hipStream_t)stream>>>(scratch, partials, dw, K, C, Cp, R);
        return kp_check_launch("wgrad_unpack_tile");
    }
    if (Cp % 4 == 0)
        if (partials >= 128 && total <= 16384)   // (64 lanes in a 1024-thread workgroup: measured no faster, 7.4 vs 7.2 us at 512 partials)
            wgrad_unpack_k<4, 16><<<cdiv(total / 4, 16), 256, 0, (hipStream_t)stream>>>(scratch, partials, dw, K, C, Cp, R);
        else
            wgrad_unpack_k<4, 4><<<cdiv(total / 4, 64), 256, 0, (hipStream_t)stream>>>(scratch, partials, dw, K, C, Cp, R);
    else if (partials >= 256 && total <= 16384)   // (first layers: thousands of partials of a few hundred floats: 9.0 -> 5.1 us)
        wgrad_unpack_k<1, 64, 1024><<<cdiv(total, 16), 1024, 0, (hipStream_t)stream>>>(scratch, partials, dw, K, C, Cp, R);
    else
        wgrad_unpack_k<1, 4><<<cdiv(total, 64), 256, 0, (hipStream_t)stream>>>(scratch, partials, dw, K, C, Cp, R);
    return kp_check_launch("wgrad_unpack");
}

#endif

int kp_head_fwd(const void* x, const float* w, const float* bias, float slope, float* out, int N, int H, int W, int C,
                int K, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "kp_head_fwd: bad dims");
    KP_REQUIRE(C <= 2048, "kp_head_fwd: C too large for the LDS tile");
    kp_count_bytes((dtype == KP_F32 ? 4.0 : 2.0) * N * H * W * C + 4.0 * N * H * W * K);
    const long long M = (long long)N * H * W;
    if (dtype == KP_H16) {
        int rc = kp_head_fwd_mfma(x, w, bias, slope, out, M, H * W, C, K, (hipStream_t)stream);
        if (rc != KP_E_UNSUPPORTED) return rc;
        rc = kp_head_fwd_fast(x, w, bias, slope, out, M, H * W, C, K, (hipStream_t)stream);
        if (rc != KP_E_UNSUPPORTED) return rc;
    }
    const size_t smem = (size_t)HPB * (C + 1) * sizeof(float);
    DISPATCH_T(dtype, head_fwd_k<T><<<cdiv(M, HPB), TPB, smem, (hipStream_t)stream>>>((const T*)x, w, bias, slope, out, M, H * W, C, K));
    return kp_check_launch("head_fwd");
}

// ---- the head fed by a convolution OUTPUT: BatchNorm-apply + activation of the producing layer on the head's load path ----
// (the last 3x3 layer of a Unit: its activation pass -- one launch, write + read of the layer's output -- is not run, the
// activation tensor never exists).  Only the 16-bit fast paths have this form: kp_head_bnin_supported tells.
// (the fused form must be the kernel the plain call would pick for the shape -- the matrix-core head where it applies, else the
//  K <= 16 kernels: two kernels round differently)
static bool head_bnin_ok(long long M, int HW, int C, int K, int dtype) {
    if (dtype != KP_H16) return false;
    const bool fwd = kp_head_mfma_fwd_shape(M, HW, C, K) ? C == 32 : kp_head_small_shape(M, C, K);
    const bool bwd = kp_head_mfma_shape(M, HW, C, K) ? kp_head_mfma_bnin_shape(M, HW, C, K) : kp_head_small_shape(M, C, K);
    return fwd && bwd;
}
int kp_head_bnin_supported(int N, int H, int W, int C, int K, int dtype) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0) return 0;
    return head_bnin_ok((long long)N * H * W, H * W, C, K, dtype) ? 1 : 0;
}

int kp_head_fwd_bnin(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     float slope_in, const float* w, const float* bias, float slope, float* out, int N, int H, int W, int C,
                     int K, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "kp_head_fwd_bnin: bad dims");
    KP_REQUIRE(slope_in >= 0.f && slope_in <= 1.f, "kp_head_fwd_bnin: activation slope of the producing layer must be in [0, 1]");
    KP_REQUIRE((mean != nullptr) == (invstd != nullptr) && (mean != nullptr) == (gamma != nullptr) && (mean != nullptr) == (beta != nullptr),
               "kp_head_fwd_bnin: mean / invstd / gamma / beta: all or none");
    kp_count_bytes((dtype == KP_F32 ? 4.0 : 2.0) * N * H * W * C + 4.0 * N * H * W * K);
    const long long M = (long long)N * H * W;
    if (!head_bnin_ok(M, H * W, C, K, dtype)) return KP_E_UNSUPPORTED;
    const KpBnIn bn{mean, invstd, gamma, beta, slope_in};
    int rc = kp_head_fwd_mfma(yprev, w, bias, slope, out, M, H * W, C, K, (hipStream_t)stream, &bn);
    if (rc == KP_E_UNSUPPORTED) rc = kp_head_fwd_fast(yprev, w, bias, slope, out, M, H * W, C, K, (hipStream_t)stream, &bn);
    return rc;
}

// partial gradients (workspace rows of K*C + K floats) the weight-gradient kernel of this shape writes in deterministic mode
static long long head_dw_parts(long long M, int HW, int C, int K, int dtype) {
    if (dtype == KP_H16) {
        long long p = kp_head_dw_parts_mfma(M, HW, C, K);
        if (p <= 0) p = kp_head_dw_parts_fast(M, C, K);
        if (p > 0) return p;
    }
    const long long tiles = (M + HPB - 1) / HPB;
    return tiles < 512 ? tiles : 512;
}

long long kp_head_bwd_ws_floats(int N, int H, int W, int C, int K, int dtype) {
    if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0) return KP_E_ARG;
    return head_dw_parts((long long)N * H * W, H * W, C, K, dtype) * ((long long)K * C + K);
}

static int head_bwd_impl(const float* gout, const float* out, const void* x, const float* w, float slope, void* dx, float* dw,
                         float* db, float* ws, int N, int H, int W, int C, int K, int dtype, void* stream, const KpBnIn* bn) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "kp_head_bwd: bad dims");
    KP_REQUIRE(K <= 256 && C <= 2048, "kp_head_bwd: K <= 256, C <= 2048 supported");
    const long long M = (long long)N * H * W;
    hipStream_t st = (hipStream_t)stream;
    if (dx && dtype == KP_H16) {
        int rc = kp_head_dx_mfma(gout, out, w, slope, dx, M, H * W, C, K, st);
        if (rc == KP_E_UNSUPPORTED) rc = kp_head_dx_fast(gout, out, w, slope, dx, M, H * W, C, K, st);
        if (rc == KP_OK) dx = nullptr;
        else if (rc != KP_E_UNSUPPORTED) return rc;
    }
    // rows of the deterministic workspace: sized by kp_head_bwd_ws_floats from the plan of the kernel this SHAPE selects.  A
    // faster kernel may still decline for a reason that is not the shape (gout / out not 16-byte aligned: sub-allocated
    // pointers of a foreign caller) and hand over to the generic kernel below, whose own grid would be min(tiles, 512) rows:
    // that grid is clamped to the rows the workspace holds, and the reduction sums the rows that were actually written.
    long long parts = (dw && ws) ? head_dw_parts(M, H * W, C, K, dtype) : 0;
    float* const dw_out = dw;
    if (dw && dtype == KP_H16) {
        int rc = kp_head_dw_mfma(gout, out, x, slope, dw, db, ws, M, H * W, C, K, st, bn);
        if (rc == KP_E_UNSUPPORTED) rc = kp_head_dw_fast(gout, out, x, slope, dw, db, ws, M, H * W, C, K, st, bn);
        if (rc == KP_OK) dw = nullptr;
        else if (rc != KP_E_UNSUPPORTED) return rc;
    }
    if (bn && dw) {   // the generic kernels below read a stored activation
        kp_set_error("kp_head_bwd_bnin: no fast weight-gradient kernel took this call (gout / out must be 16-byte aligned)");
        return KP_E_UNSUPPORTED;
    }
    if (dx) {
        const size_t smem = (size_t)HPB * (K + 1) * sizeof(float);
        DISPATCH_T(dtype, head_bwd_dx_k<T><<<cdiv(M, HPB), TPB, smem, st>>>(gout, out, w, slope, (T*)dx, M, H * W, C, K));
        int rc = kp_check_launch("head_bwd_dx");
        if (rc) return rc;
    }
    if (dw) {
        const long long tiles = (M + HPB - 1) / HPB;
        unsigned gx = (unsigned)(tiles < 512 ? tiles : 512);
        if (parts > 0) {
            if ((long long)gx > parts) gx = (unsigned)parts;   // the kernel strides over the tiles with its grid: any width is complete
            parts = gx;
        }
        const unsigned gy = cdiv((long long)K * C, (long long)TPB * HPAIRS);
        const size_t smem = (size_t)HPB * (K + C) * sizeof(float);
        DISPATCH_T(dtype, head_bwd_dw_k<T><<<dim3(gx, gy), TPB, smem, st>>>(gout, out, (const T*)x, slope, dw, db, ws, M, H * W, C, K));
        int rc = kp_check_launch("head_bwd_dw");
        if (rc) return rc;
    }
    if (parts > 0) {   // deterministic mode: sum the partial gradients in a fixed order
        const long long n = (long long)K * C + K;
        head_dw_reduce(ws, parts, n, (long long)K * C, dw_out, db, st);
        return kp_check_launch("head_dw_reduce");
    }
    return KP_OK;
}

int kp_head_bwd(const float* gout, const float* out, const void* x, const float* w, float slope, void* dx, float* dw,
                float* db, float* ws, int N, int H, int W, int C, int K, int dtype, void* stream) {
    kp_count_bytes(8.0 * N * H * W * K + (dtype == KP_F32 ? 4.0 : 2.0) * N * H * W * C * (dx ? 2.0 : 1.0));
    return head_bwd_impl(gout, out, x, w, slope, dx, dw, db, ws, N, H, W, C, K, dtype, stream, nullptr);
}

int kp_head_bwd_bnin(const float* gout, const float* out, const void* yprev, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, float slope_in, const float* w, float slope, void* dx, float* dw,
                     float* db, float* ws, int N, int H, int W, int C, int K, int dtype, void* stream) {
    KP_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "kp_head_bwd_bnin: bad dims");
    KP_REQUIRE(slope_in >= 0.f && slope_in <= 1.f, "kp_head_bwd_bnin: activation slope of the producing layer must be in [0, 1]");
    KP_REQUIRE((mean != nullptr) == (invstd != nullptr) && (mean != nullptr) == (gamma != nullptr) && (mean != nullptr) == (beta != nullptr),
               "kp_head_bwd_bnin: mean / invstd / gamma / beta: all or none");
    kp_count_bytes(8.0 * N * H * W * K + (dtype == KP_F32 ? 4.0 : 2.0) * N * H * W * C * (dx ? 2.0 : 1.0));
    if (!head_bnin_ok((long long)N * H * W, H * W, C, K, dtype)) return KP_E_UNSUPPORTED;
    const KpBnIn bn{mean, invstd, gamma, beta, slope_in};
    return head_bwd_impl(gout, out, yprev, w, slope, dx, dw, db, ws, N, H, W, C, K, dtype, stream, &bn);
}

int kp_nhwc_to_nchw(const void* src, float* dst, int N, int C, int H, int W, int pad, int dtype, void* stream) {
    KP_REQUIRE(src && dst && N > 0 && C > 0 && C % 4 == 0 && C / 4 <= 65535 && H > 0 && W > 0 && pad >= 0, "kp_nhwc_to_nchw: bad dims (C a multiple of 4)");
    kp_count_bytes((dtype == KP_F32 ? 4.0 : 2.0) * N * H * W * C + 4.0 * N * H * W * C);
    const long long total = (long long)N * H * W;
    DISPATCH_T(dtype, nhwc_to_nchw_v4_k<T><<<dim3(cdiv(total, TPB), C / 4), TPB, 0, (hipStream_t)stream>>>((const T*)src, dst, N, C, H, W, pad));
    return kp_check_launch("nhwc_to_nchw");
}

// ---- the head between T tensors (16-bit NHWC in, 16-bit NHWC out): kp_head_mfma.hip ------------------------------------
int kp_head_t_supported(int N, int H, int W, int C, int K, int dtype) {
    if (dtype != KP_H16 || N <= 0 || H <= 0 || W <= 0 || C <= 0 || K <= 0) return 0;
    return kp_head_t_shape((long long)N * H * W, H * W, C, K) ? 1 : 0;
}

static int head_t_bn(const char* who, int bnin, const float* mean, const float* invstd, const float* gamma, const float* beta, float slope_in) {
    if (!bnin) return KP_OK;
    KP_REQUIRE(slope_in >= 0.f && slope_in <= 1.f, "%s: activation slope of the producing layer must be in [0, 1]", who);
    KP_REQUIRE((mean != nullptr) == (invstd != nullptr) && (mean != nullptr) == (gamma != nullptr) && (mean != nullptr) == (beta != nullptr),
               "%s: mean / invstd / gamma / beta: all or none", who);
    return KP_OK;
}

int kp_head_fwd_t(const void* x, int bnin, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  float slope_in, const float* w, const float* bias, float slope, void* out_t, int N, int H, int W, int C, int K,
                  int dtype, void* stream) {
    KP_REQUIRE(x && w && out_t && N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "kp_head_fwd_t: bad args");
    if (int rc = head_t_bn("kp_head_fwd_t", bnin, mean, invstd, gamma, beta, slope_in)) return rc;
    if (!kp_head_t_supported(N, H, W, C, K, dtype)) return KP_E_UNSUPPORTED;
    kp_count_bytes((dtype == KP_F32 ? 4.0 : 2.0) * N * H * W * (C + K));
    const KpBnIn bn{mean, invstd, gamma, beta, slope_in};
    return kp_head_fwd_mfma_t(x, w, bias, slope, out_t, (long long)N * H * W, H * W, C, K, (hipStream_t)stream, bnin ? &bn : nullptr);
}

// The two forward forms above for TWO half-batches in one launch (see kp_conv_fwd_pair): yprev [N = 2 x N/2][H][W][C] with mean /
// invstd [2][C]; the halves' outputs go to out0 / out1 (fp32 NCHW [N/2][K][H][W] each; T [N/2][H][W][K] for the _t form).  Each half
// is launched as a separate call of N / 2 images would be: bit-identical.  Only the 32-channel matrix-core head has this form.
int kp_head_pair_supported(int N, int H, int W, int C, int K, int t_out, int dtype) {
    if (dtype != KP_H16 || N < 2 || (N & 1) || H <= 0 || W <= 0 || C != 32) return 0;
    const long long M = (long long)(N / 2) * H * W;
    return (t_out ? kp_head_t_shape(M, H * W, C, K) : kp_head_mfma_fwd_shape(M, H * W, C, K)) ? 1 : 0;
}

int kp_head_fwd_bnin_pair(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                          float slope_in, const float* w, const float* bias, float slope, void* out0, void* out1, int t_out, int N,
                          int H, int W, int C, int K, int dtype, void* stream) {
    KP_REQUIRE(yprev && w && out0 && out1 && N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "kp_head_fwd_bnin_pair: bad args");
    if (int rc = head_t_bn("kp_head_fwd_bnin_pair", 1, mean, invstd, gamma, beta, slope_in)) return rc;
    if (!kp_head_pair_supported(N, H, W, C, K, t_out, dtype)) return KP_E_UNSUPPORTED;
    kp_count_bytes((dtype == KP_F32 ? 4.0 : 2.0) * N * H * W * C + (t_out ? (dtype == KP_F32 ? 4.0 : 2.0) : 4.0) * N * H * W * K);
    const long long M = (long long)(N / 2) * H * W;
    const KpBnIn bn{mean, invstd, gamma, beta, slope_in};
    const void* x1 = (const h16_t*)yprev + M * C;
    if (t_out) return kp_head_fwd_mfma_t(yprev, w, bias, slope, out0, M, H * W, C, K, (hipStream_t)stream, &bn, x1, out1);
    return kp_head_fwd_mfma(yprev, w, bias, slope, (float*)out0, M, H * W, C, K, (hipStream_t)stream, &bn, x1, (float*)out1);
}

int kp_head_bwd_t(const void* gout_t, const void* out_t, const void* x, int bnin, const float* mean, const float* invstd,
                  const float* gamma, const float* beta, float slope_in, const float* w, float slope, void* dx, float* dw, float* db,
                  float* ws, int N, int H, int W, int C, int K, int dtype, void* stream) {
    KP_REQUIRE(gout_t && out_t && x && w && N > 0 && H > 0 && W > 0 && C > 0 && K > 0, "kp_head_bwd_t: bad args");
    if (int rc = head_t_bn("kp_head_bwd_t", bnin, mean, invstd, gamma, beta, slope_in)) return rc;
    if (!kp_head_t_supported(N, H, W, C, K, dtype)) return KP_E_UNSUPPORTED;
    // every sub-kernel's alignment rule, checked BEFORE the first launch: KP_E_UNSUPPORTED means nothing ran (the caller's fallback
    // accumulates into the same `+=` gradient slots)
    if ((((size_t)gout_t | (size_t)out_t | (size_t)x | (size_t)dx) & 15) != 0) return KP_E_UNSUPPORTED;
    kp_count_bytes((dtype == KP_F32 ? 4.0 : 2.0) * N * H * W * (2.0 * K + C * (dx ? 2.0 : 1.0)));
    const long long M = (long long)N * H * W;
    hipStream_t st = (hipStream_t)stream;
    const KpBnIn bn{mean, invstd, gamma, beta, slope_in};
    if (dx)
        if (int rc = kp_head_dx_mfma_t(gout_t, out_t, w, slope, dx, M, H * W, C, K, st)) return rc;
    if (dw) {
        if (int rc = kp_head_dw_mfma_t(gout_t, out_t, x, slope, dw, db, ws, M, H * W, C, K, st, bnin ? &bn : nullptr)) return rc;
        if (ws) {   // deterministic mode: sum the partial gradients in a fixed order
            const long long n = (long long)K * C + K, parts = kp_head_dw_parts_mfma(M, H * W, C, K);
            head_dw_reduce(ws, parts, n, (long long)K * C, dw, db, st);
            return kp_check_launch("head_dw_reduce");
        }
    }
    return KP_OK;
}

#ifndef KP_F16_BUILD   // no dtype: defined once, by the bf16 build of this source
int kp_l2_loss(const float* x, const float* t, const float* mask, double* loss_sum, float* grad, float scale, size_t n,
               void* stream) {
    KP_REQUIRE(n > 0, "kp_l2_loss: empty");
    kp_count_bytes(4.0 * n * (mask ? 3.0 : 2.0) + (grad ? 4.0 * n : 0.0));
    const bool al = ((((size_t)x | (size_t)t | (size_t)mask | (size_t)grad) & 15) == 0);
    const size_t n4 = al ? n / 4 : 0;
    unsigned grid = cdiv(n, TPB * 16);
    if (grid > 512) grid = 512;
    l2_loss_k<<<grid, TPB, 0, (hipStream_t)stream>>>(x, t, mask, loss_sum, grad, scale, n, n4);
    return kp_check_launch("l2_loss");
}

int kp_l2_loss_mean(const float* x, const float* t, const float* mask, double* ws2, float* loss, float* grad, size_t n, void* stream) {
    KP_REQUIRE(x && t && ws2 && loss && n > 0, "kp_l2_loss_mean: bad args");
    kp_count_bytes(4.0 * n * (mask ? 3.0 : 2.0) + (grad ? 4.0 * n : 0.0));
    const bool al = ((((size_t)x | (size_t)t | (size_t)mask | (size_t)grad) & 15) == 0);
    const size_t n4 = al ? n / 4 : 0;
    unsigned grid = cdiv(n, TPB * 16);
    if (grid > 512) grid = 512;
    l2_loss_k<<<grid, TPB, 0, (hipStream_t)stream>>>(x, t, mask, ws2, grad, 1.0f / (float)n, n, n4, loss);
    return kp_check_launch("l2_loss_mean");
}

int kp_scale(const float* x, const float* s, float* y, size_t n, void* stream) {
    KP_REQUIRE(x && s && y && n > 0, "kp_scale: bad args");
    KP_REQUIRE((((size_t)x | (size_t)y) & 15) == 0, "kp_scale: pointers must be 16-byte aligned");
    kp_count_bytes(8.0 * n);
    scale_k<<<cdiv((n + 3) / 4, TPB), TPB, 0, (hipStream_t)stream>>>(x, s, y, n);
    return kp_check_launch("scale");
}

int kp_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float b1, float b2, float eps,
                 float bc1, float bc2, void* stream) {
    KP_REQUIRE(n > 0, "kp_adam_step: empty");
    kp_count_bytes(28.0 * n);
    adam_k<<<cdiv(n, TPB), TPB, 0, (hipStream_t)stream>>>(p, g, m, v, n, lr, b1, b2, eps, bc1, sqrtf(bc2));
    return kp_check_launch("adam_step");
}

#endif

}  // extern "C"
