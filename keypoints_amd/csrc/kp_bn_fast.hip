// Fast bf16 paths of the BatchNorm(train) + activation passes (kp_bn.hip holds the general kernels and the
// entry points; this file is their HBM-roofline version for the shapes the networks actually run).
//
// Same mapping idea as kp_bn.hip -- a thread owns ONE 16-byte channel vector (8 bf16) and walks pixels, so the
// per-channel constants live in registers -- but shaped for bandwidth:
//   * a workgroup takes several image rows (>= 16 vectors per thread), not one;
//   * the pixel loop is unrolled by 4 with all loads issued before the first use (4-8 x 16 B in flight per lane);
//   * tail / border-width are template parameters; the replicate-pad adjoint ("fold") of an incoming gradient
//     is one unconditional centre load plus a rare edge path;
//   * the backward reduction leaves ONE partial row per workgroup (kp_last_rows() = number of workgroups),
//     which also shortens the fp64 column reduction that follows.
// Arithmetic is expression-for-expression the same as the general kernels (only the summation order of the
// fold at image edges differs).
#include "kp_common.h"

namespace {

constexpr int TPB = 256, V = 8, UNR = 4;
constexpr int UNB = 4;   // pixels in flight per thread in the backward kernels (2 and 1 measured slower)

struct BnP {
    const float *mean, *invstd, *gamma, *beta;
};

struct BnC {
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

__device__ __forceinline__ void unpack8(const uint4 v, float (&o)[V]) {
    unpack_h16x2(v.x, o[0], o[1]);
    unpack_h16x2(v.y, o[2], o[3]);
    unpack_h16x2(v.z, o[4], o[5]);
    unpack_h16x2(v.w, o[6], o[7]);
}
__device__ __forceinline__ uint4 pack8(const float (&o)[V]) {
    uint4 v;
    v.x = pack_h16x2(o[0], o[1]); v.y = pack_h16x2(o[2], o[3]);
    v.z = pack_h16x2(o[4], o[5]); v.w = pack_h16x2(o[6], o[7]);
    return v;
}
__device__ __forceinline__ uint4 ldv(const h16_t* p) { return *reinterpret_cast<const uint4*>(p); }
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// streaming load: the line is not kept in the caches behind it (last use of the data for a long time)
__device__ __forceinline__ uint4 ldv_nt(const h16_t* p, int nt) {
    if (!nt) return ldv(p);
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void stv(h16_t* p, const uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- forward, tail none: out[n][i][j] = act(bn(y[n][clamp(i-p)][clamp(j-p)])) ------------------------------
__global__ __launch_bounds__(TPB) void bn_act_fwd_none_fast_k(const h16_t* __restrict__ y, BnP bn, float slope,
                                                              h16_t* __restrict__ out, int p, int H, int W, int C,
                                                              long long total_rows, int rpb, int walk) {
    const int CV = C / V, Hp = H + 2 * p, Wp = W + 2 * p;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV, npl = TPB / CV;
    const int c = cv * V;
    BnC k; k.load(bn, c);
    const unsigned bid = (walk & 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int nt = walk & 2;
    const long long r0 = (long long)bid * rpb;
    for (int rr = 0; rr < rpb; ++rr) {
        const long long row = r0 + rr;
        if (row >= total_rows) break;
        const long long n = row / Hp;
        const int i = (int)(row - n * Hp);
        const int oh = clampi(i - p, 0, H - 1);
        const h16_t* yr = y + ((n * H + oh) * W) * (long long)C + c;
        h16_t* orow = out + row * Wp * (long long)C + c;
        for (int j0 = pl; j0 < Wp; j0 += UNR * npl) {
            uint4 raw[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int j = j0 + u * npl;
                if (j < Wp) raw[u] = ldv_nt(yr + (long long)clampi(j - p, 0, W - 1) * C, nt);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int j = j0 + u * npl;
                if (j < Wp) {
                    float v[V], res[V];
                    unpack8(raw[u], v);
#pragma unroll
                    for (int e = 0; e < V; ++e) res[e] = act_fwd(k.z(v[e], e), slope);
                    stv(orow + (long long)j * C, pack8(res));
                }
            }
        }
    }
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

// ---- forward with a 2x2 max-pool (TAIL 1) or bilinear x2 (TAIL 2) tail: 4 source vectors per output vector ----
template <int TAIL>
__global__ __launch_bounds__(TPB) void bn_act_fwd_tail_fast_k(const h16_t* __restrict__ y, BnP bn, float slope,
                                                              h16_t* __restrict__ out, int p,
                                                              unsigned char* __restrict__ argmax, int H, int W, int C,
                                                              long long total_rows, int rpb, int walk) {
    constexpr int U2 = 2;   // output pixels in flight per thread (8 loads)
    const int Ho = TAIL == 1 ? H / 2 : 2 * H, Wo = TAIL == 1 ? W / 2 : 2 * W;
    const int CV = C / V, Hp = Ho + 2 * p, Wp = Wo + 2 * p;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV, npl = TPB / CV;
    const int c = cv * V;
    BnC k; k.load(bn, c);
    const unsigned bid = (walk & 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int nt = walk & 2;
    const long long r0 = (long long)bid * rpb;
    for (int rr = 0; rr < rpb; ++rr) {
        const long long row = r0 + rr;
        if (row >= total_rows) break;
        const long long n = row / Hp;
        const int i = (int)(row - n * Hp);
        const int oh = clampi(i - p, 0, Ho - 1);
        const h16_t* yn = y + n * H * W * (long long)C + c;
        h16_t* orow = out + row * Wp * (long long)C + c;
        const h16_t *y0, *y1;
        float lh0 = 0.f, lh1 = 0.f;
        if (TAIL == 1) {
            y0 = yn + (long long)(2 * oh) * W * C;
            y1 = y0 + (long long)W * C;
        } else {
            int h0, hp;
            bilin(oh, H, Ho, h0, hp, lh0, lh1);
            y0 = yn + (long long)h0 * W * C;
            y1 = y0 + (long long)hp * W * C;
        }
        for (int j0 = pl; j0 < Wp; j0 += U2 * npl) {
            uint4 raw[U2][4];
            float lw0[U2], lw1[U2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int j = j0 + u * npl;
                if (j < Wp) {
                    const int ow = clampi(j - p, 0, Wo - 1);
                    int wa, wb;
                    if (TAIL == 1) { wa = 2 * ow; wb = 2 * ow + 1; lw0[u] = lw1[u] = 0.f; }
                    else { int w0, wp; bilin(ow, W, Wo, w0, wp, lw0[u], lw1[u]); wa = w0; wb = w0 + wp; }
                    raw[u][0] = ldv_nt(y0 + (long long)wa * C, nt);
                    raw[u][1] = ldv_nt(y0 + (long long)wb * C, nt);
                    raw[u][2] = ldv_nt(y1 + (long long)wa * C, nt);
                    raw[u][3] = ldv_nt(y1 + (long long)wb * C, nt);
                }
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int j = j0 + u * npl;
                if (j < Wp) {
                    float a[4][V], res[V];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v[V];
                        unpack8(raw[u][q], v);
#pragma unroll
                        for (int e = 0; e < V; ++e) a[q][e] = act_fwd(k.z(v[e], e), slope);
                    }
                    if (TAIL == 1) {
                        int best[V];
#pragma unroll
                        for (int e = 0; e < V; ++e) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float aq = a[q][e];   // compared in fp32 (see kp_bn.hip): ties of the 16-bit
                                                            // values resolve to the fp32 maximum, like the reference
                                if (q == 0 || aq > res[e] || aq != aq) { res[e] = aq; best[e] = q; }
                            }
                        }
                        const int ow = clampi(j - p, 0, Wo - 1);
                        if (argmax && i - p == oh && j - p == ow) {
                            uint2 pk;
                            pk.x = best[0] | (best[1] << 8) | (best[2] << 16) | (best[3] << 24);
                            pk.y = best[4] | (best[5] << 8) | (best[6] << 16) | (best[7] << 24);
                            *reinterpret_cast<uint2*>(argmax + ((n * Ho + oh) * Wo + ow) * (long long)C + c) = pk;
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < V; ++e)
                            res[e] = lh0 * (lw0[u] * a[0][e] + lw1[u] * a[1][e]) + lh1 * (lw0[u] * a[2][e] + lw1[u] * a[3][e]);
                    }
                    stv(orow + (long long)j * C, pack8(res));
                }
            }
        }
    }
}

// ---- forward with the bilinear x2 tail, 2x2 blocks of the (padded) output ----------------------------------
// With align_corners=True and a border of 1, the two output rows 2b, 2b+1 of the padded image interpolate between
// the SAME two source rows (and likewise for columns), so a 2x2 output block needs 2x2 source vectors: 4 loads and
// 4 BN+activation evaluations for 4 outputs instead of 16.  The interpolation indices / weights still come from
// the exact per-output formula (bilin); an output whose taps fall outside the block's source pair (possible only
// at odd paddings / image ends) takes the direct path.
__global__ __launch_bounds__(TPB) void bn_act_fwd_up_fast_k(const h16_t* __restrict__ y, BnP bn, float slope,
                                                            h16_t* __restrict__ out, int p, int H, int W, int C,
                                                            long long total_rows, int rpb, int walk) {
    const int Ho = 2 * H, Wo = 2 * W;
    const int CV = C / V, Hp = Ho + 2 * p, Wp = Wo + 2 * p, HB = (Hp + 1) / 2, WB = (Wp + 1) / 2;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV, npl = TPB / CV;
    const int c = cv * V;
    BnC k; k.load(bn, c);
    const unsigned bid = (walk & 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int nt = walk & 2;
    const long long r0 = (long long)bid * rpb;
    for (int rr = 0; rr < rpb; ++rr) {
        const long long row = r0 + rr;              // block row: (n, bi)
        if (row >= total_rows) break;
        const long long n = row / HB;
        const int bi = (int)(row - n * HB);
        const h16_t* yn = y + n * H * W * (long long)C + c;
        int hh0[2], hhp[2]; float lh0[2], lh1[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) bilin(clampi(2 * bi + a - p, 0, Ho - 1), H, Ho, hh0[a], hhp[a], lh0[a], lh1[a]);
        const int rA = hh0[0], rB = rA + 1 < H ? rA + 1 : H - 1;
        const h16_t* yA = yn + (long long)rA * W * C;
        const h16_t* yB = yn + (long long)rB * W * C;
        for (int bj = pl; bj < WB; bj += npl) {
            int ww0[2], wwp[2]; float lw0[2], lw1[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) bilin(clampi(2 * bj + b - p, 0, Wo - 1), W, Wo, ww0[b], wwp[b], lw0[b], lw1[b]);
            const int cA = ww0[0], cB = cA + 1 < W ? cA + 1 : W - 1;
            float s[2][2][V];   // activated sources [row A/B][col A/B]
            {
                const uint4 rAA = ldv_nt(yA + (long long)cA * C, nt), rAB = ldv_nt(yA + (long long)cB * C, nt);
                const uint4 rBA = ldv_nt(yB + (long long)cA * C, nt), rBB = ldv_nt(yB + (long long)cB * C, nt);
                float v[V];
                unpack8(rAA, v);
#pragma unroll
                for (int e = 0; e < V; ++e) s[0][0][e] = act_fwd(k.z(v[e], e), slope);
                unpack8(rAB, v);
#pragma unroll
                for (int e = 0; e < V; ++e) s[0][1][e] = act_fwd(k.z(v[e], e), slope);
                unpack8(rBA, v);
#pragma unroll
                for (int e = 0; e < V; ++e) s[1][0][e] = act_fwd(k.z(v[e], e), slope);
                unpack8(rBB, v);
#pragma unroll
                for (int e = 0; e < V; ++e) s[1][1][e] = act_fwd(k.z(v[e], e), slope);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int i = 2 * bi + a;
                if (i >= Hp) continue;
                const int r0i = hh0[a], r1i = hh0[a] + hhp[a];
                const bool rok = (r0i == rA || r0i == rB) && (r1i == rA || r1i == rB);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int j = 2 * bj + b;
                    if (j >= Wp) continue;
                    const int c0i = ww0[b], c1i = ww0[b] + wwp[b];
                    const bool cok = (c0i == cA || c0i == cB) && (c1i == cA || c1i == cB);
                    float res[V];
                    if (rok && cok) {
                        const int ra = r0i == rA ? 0 : 1, rb = r1i == rA ? 0 : 1, ca = c0i == cA ? 0 : 1, cb = c1i == cA ? 0 : 1;
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const float a00 = ra ? (ca ? s[1][1][e] : s[1][0][e]) : (ca ? s[0][1][e] : s[0][0][e]);
                            const float a01 = ra ? (cb ? s[1][1][e] : s[1][0][e]) : (cb ? s[0][1][e] : s[0][0][e]);
                            const float a10 = rb ? (ca ? s[1][1][e] : s[1][0][e]) : (ca ? s[0][1][e] : s[0][0][e]);
                            const float a11 = rb ? (cb ? s[1][1][e] : s[1][0][e]) : (cb ? s[0][1][e] : s[0][0][e]);
                            res[e] = lh0[a] * (lw0[b] * a00 + lw1[b] * a01) + lh1[a] * (lw0[b] * a10 + lw1[b] * a11);
                        }
                    } else {   // direct path
                        const h16_t* q0 = yn + (long long)r0i * W * C;
                        const h16_t* q1 = yn + (long long)r1i * W * C;
                        float v00[V], v01[V], v10[V], v11[V];
                        unpack8(ldv(q0 + (long long)c0i * C), v00); unpack8(ldv(q0 + (long long)c1i * C), v01);
                        unpack8(ldv(q1 + (long long)c0i * C), v10); unpack8(ldv(q1 + (long long)c1i * C), v11);
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const float a00 = act_fwd(k.z(v00[e], e), slope), a01 = act_fwd(k.z(v01[e], e), slope);
                            const float a10 = act_fwd(k.z(v10[e], e), slope), a11 = act_fwd(k.z(v11[e], e), slope);
                            res[e] = lh0[a] * (lw0[b] * a00 + lw1[b] * a01) + lh1[a] * (lw0[b] * a10 + lw1[b] * a11);
                        }
                    }
                    stv(out + ((n * Hp + i) * Wp + j) * (long long)C + c, pack8(res));
                }
            }
        }
    }
}

// ---- backward ---------------------------------------------------------------------------------------------
// Incoming gradient for pre-tail pixel (h, w): TAIL none -> source pixel (h, w); TAIL pool -> source pixel
// (h/2, w/2) taken only where argmax == position.  GP = 1: the source is a padded-domain gradient
// [Hs+2][Ws+2] whose border ring folds onto the edge pixels.
template <int GP>
__device__ __forceinline__ const h16_t* g_centre(const h16_t* gsrc, long long n, int sh, int sw, int Hs, int Ws, int C) {
    return gsrc + ((n * (Hs + 2 * GP) + sh + GP) * (Ws + 2 * GP) + sw + GP) * (long long)C;
}
// adds the ring neighbours of an edge pixel (everything in the fold range except the centre)
__device__ __forceinline__ void g_edge_add(const h16_t* gsrc, long long n, int sh, int sw, int Hs, int Ws, int C, int c,
                                           float (&g)[V]) {
    const int ilo = (sh == 0) ? 0 : sh + 1, ihi = (sh == Hs - 1) ? Hs + 1 : sh + 1;
    const int jlo = (sw == 0) ? 0 : sw + 1, jhi = (sw == Ws - 1) ? Ws + 1 : sw + 1;
    for (int i = ilo; i <= ihi; ++i)
        for (int j = jlo; j <= jhi; ++j) {
            if (i == sh + 1 && j == sw + 1) continue;
            float v[V];
            unpack8(ldv(gsrc + ((n * (Hs + 2) + i) * (Ws + 2) + j) * (long long)C + c), v);
#pragma unroll
            for (int e = 0; e < V; ++e) g[e] += v[e];
        }
}
// MODE 0: per-workgroup partial sums of (ghat, ghat*xhat).  MODE 1: dy = gamma*invstd*(ghat - db - xhat*dg).
// Rows: MODE 0 iterates the N*H pre-tail rows; MODE 1 iterates the N*(H+2dp) output rows (border rows zeroed).
template <int POOL, int GP, int MODE>
__global__ __launch_bounds__(TPB) void bn_bwd_fast_k(const h16_t* __restrict__ gsrc, const unsigned char* __restrict__ argmax,
                                                     const h16_t* __restrict__ y, BnP bn, float slope,
                                                     float* __restrict__ partials, const float* __restrict__ sums,
                                                     h16_t* __restrict__ dy, int dp, int N, int H, int W, int C,
                                                     long long total_rows, int rpb, int walk) {
    const int CV = C / V;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV, npl = TPB / CV;
    const int c = cv * V;
    const int Hs = POOL ? H / 2 : H, Ws = POOL ? W / 2 : W;
    const int Hp = H + 2 * dp, Wp = W + 2 * dp;
    BnC k; k.load(bn, c);
    float sb[V], sg[V], db[V], dg[V];
#pragma unroll
    for (int e = 0; e < V; ++e) sb[e] = sg[e] = db[e] = dg[e] = 0.f;
    if (MODE == 1) {
        const float invM = 1.0f / (float)((long long)N * H * W);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            db[e] = k.affine ? sums[c + e] * invM : 0.f;
            dg[e] = k.affine ? sums[C + c + e] * invM : 0.f;
        }
    }
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const unsigned bid = (walk & 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int nt = walk & 2;
    const long long r0 = (long long)bid * rpb;
    for (int rr = 0; rr < rpb; ++rr) {
        const long long row = r0 + rr;
        if (row >= total_rows) break;
        long long n; int h;
        h16_t* drow = nullptr;
        if (MODE == 1) {
            n = row / Hp;
            h = (int)(row - n * Hp) - dp;
            drow = dy + row * Wp * (long long)C + c;
            if (h < 0 || h >= H) {   // zero border row
                for (int j = pl; j < Wp; j += npl) stv(drow + (long long)j * C, zero4);
                continue;
            }
            for (int j = pl; j < dp; j += npl) {   // zero border columns
                stv(drow + (long long)j * C, zero4);
                stv(drow + (long long)(W + dp + j) * C, zero4);
            }
        } else {
            n = row / H;
            h = (int)(row - n * H);
        }
        const int sh = POOL ? (h >> 1) : h;
        const bool hedge = GP && (sh == 0 || sh == Hs - 1);
        const h16_t* yr = y + ((n * H + h) * W) * (long long)C + c;
        for (int w0 = pl; w0 < W; w0 += UNB * npl) {
            uint4 graw[UNB], yraw[UNB];
            uint2 amraw[UNB];
#pragma unroll
            for (int u = 0; u < UNB; ++u) {
                const int w = w0 + u * npl;
                if (w < W) {
                    const int sw = POOL ? (w >> 1) : w;
                    graw[u] = ldv_nt(g_centre<GP>(gsrc, n, sh, sw, Hs, Ws, C) + c, nt);
                    yraw[u] = ldv_nt(yr + (long long)w * C, nt);
                    if (POOL) amraw[u] = *reinterpret_cast<const uint2*>(argmax + ((n * Hs + sh) * Ws + sw) * (long long)C + c);
                }
            }
#pragma unroll
            for (int u = 0; u < UNB; ++u) {
                const int w = w0 + u * npl;
                if (w < W) {
                    const int sw = POOL ? (w >> 1) : w;
                    float g[V], yv[V];
                    unpack8(graw[u], g);
                    unpack8(yraw[u], yv);
                    if (GP && (hedge || sw == 0 || sw == Ws - 1)) g_edge_add(gsrc, n, sh, sw, Hs, Ws, C, c, g);
                    if (POOL) {
                        const unsigned pos = ((h & 1) << 1) | (w & 1);
                        unsigned take = 0;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            take |= (((amraw[u].x >> (8 * e)) & 0xffu) == pos) ? (1u << e) : 0u;
                            take |= (((amraw[u].y >> (8 * e)) & 0xffu) == pos) ? (1u << (e + 4)) : 0u;
                        }
#pragma unroll
                        for (int e = 0; e < V; ++e) if (!((take >> e) & 1u)) g[e] = 0.f;
                    }
                    if (MODE == 0) {
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const float gh = g[e] * act_grad(k.z(yv[e], e), slope);
                            sb[e] += gh;
                            sg[e] += gh * k.xhat(yv[e], e);
                        }
                    } else {
                        float res[V];
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            const float gh = g[e] * act_grad(k.z(yv[e], e), slope);
                            res[e] = k.affine ? k.ga[e] * k.is[e] * (gh - db[e] - k.xhat(yv[e], e) * dg[e]) : gh;
                        }
                        stv(drow + (long long)(w + dp) * C, pack8(res));
                    }
                }
            }
        }
    }
    if (MODE == 0) {
        // workgroup reduction over the pixel lanes -> partial row blockIdx.x (kp_last_rows() = grid size)
        __shared__ float red[TPB * 2 * V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            red[(pl * CV + cv) * 2 * V + e] = sb[e];
            red[(pl * CV + cv) * 2 * V + V + e] = sg[e];
        }
        __syncthreads();
        if (pl == 0) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float a = 0.f, b = 0.f;
                for (int r2 = 0; r2 < npl; ++r2) {
                    a += red[(r2 * CV + cv) * 2 * V + e];
                    b += red[(r2 * CV + cv) * 2 * V + V + e];
                }
                partials[((long long)bid * 2 + 0) * C + c + e] = a;
                partials[((long long)bid * 2 + 1) * C + c + e] = b;
            }
        }
    }
}

// MaxPool2d(2,2) tail, walking POOLED pixels: one load of the pooled gradient and of the argmax codes serves the four
// pre-pool positions (the per-pre-pool-pixel form above re-reads them four times: 12 loads per pooled pixel, here 6).
// MODE 0 rows = N*(H/2) pooled rows; MODE 1 also rows = N*(H/2): a pooled row writes pre-pool rows 2sh, 2sh+1 of dy
// (+ the zero border columns, and the zero border rows when it is the first / last pooled row of an image).
template <int GP, int MODE>
__global__ __launch_bounds__(TPB) void bn_bwd_pool_fast_k(const h16_t* __restrict__ gsrc, const unsigned char* __restrict__ argmax,
                                                          const h16_t* __restrict__ y, BnP bn, float slope,
                                                          float* __restrict__ partials, const float* __restrict__ sums,
                                                          h16_t* __restrict__ dy, int dp, int N, int H, int W, int C,
                                                          long long total_rows, int rpb, int walk) {
    constexpr int U2 = MODE == 0 ? 1 : 2;   // reduce: occupancy beats loads in flight (measured 212 -> 192 us), apply: the reverse
    const int CV = C / V;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV, npl = TPB / CV;
    const int c = cv * V;
    const int Hs = H / 2, Ws = W / 2, Wp = W + 2 * dp, Hp = H + 2 * dp;
    BnC k; k.load(bn, c);
    float sb[V], sg[V], db[V], dg[V];
#pragma unroll
    for (int e = 0; e < V; ++e) sb[e] = sg[e] = db[e] = dg[e] = 0.f;
    if (MODE == 1) {
        const float invM = 1.0f / (float)((long long)N * H * W);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            db[e] = k.affine ? sums[c + e] * invM : 0.f;
            dg[e] = k.affine ? sums[C + c + e] * invM : 0.f;
        }
    }
    const uint4 zero4 = make_uint4(0, 0, 0, 0);
    const unsigned bid = (walk & 1) ? gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const int nt = walk & 2;
    const long long r0 = (long long)bid * rpb;
    for (int rr = 0; rr < rpb; ++rr) {
        const long long row = r0 + rr;
        if (row >= total_rows) break;
        const long long n = row / Hs;
        const int sh = (int)(row - n * Hs);
        const bool hedge = GP && (sh == 0 || sh == Hs - 1);
        const h16_t* y0 = y + ((n * H + 2 * sh) * W) * (long long)C + c;
        const h16_t* y1 = y0 + (long long)W * C;
        h16_t* d0 = nullptr;
        if (MODE == 1) {
            d0 = dy + ((n * Hp + 2 * sh + dp) * Wp) * (long long)C + c;   // pre-pool row 2sh (incl. left border)
            for (int j = pl; j < dp; j += npl) {                             // zero border columns of both rows
                stv(d0 + (long long)j * C, zero4); stv(d0 + (long long)(W + dp + j) * C, zero4);
                stv(d0 + (long long)(Wp + j) * C, zero4); stv(d0 + (long long)(Wp + W + dp + j) * C, zero4);
            }
            if (sh == 0)
                for (int r = 0; r < dp; ++r)
                    for (int j = pl; j < Wp; j += npl) stv(dy + ((n * Hp + r) * Wp + j) * (long long)C + c, zero4);
            if (sh == Hs - 1)
                for (int r = 0; r < dp; ++r)
                    for (int j = pl; j < Wp; j += npl) stv(dy + ((n * Hp + H + dp + r) * Wp + j) * (long long)C + c, zero4);
        }
        for (int w0 = pl; w0 < Ws; w0 += U2 * npl) {
            uint4 graw[U2], yraw[U2][4];
            uint2 amraw[U2];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int sw = w0 + u * npl;
                if (sw < Ws) {
                    graw[u] = ldv_nt(g_centre<GP>(gsrc, n, sh, sw, Hs, Ws, C) + c, nt);
                    amraw[u] = *reinterpret_cast<const uint2*>(argmax + ((n * Hs + sh) * Ws + sw) * (long long)C + c);
                    yraw[u][0] = ldv_nt(y0 + (long long)(2 * sw) * C, nt);
                    yraw[u][1] = ldv_nt(y0 + (long long)(2 * sw + 1) * C, nt);
                    yraw[u][2] = ldv_nt(y1 + (long long)(2 * sw) * C, nt);
                    yraw[u][3] = ldv_nt(y1 + (long long)(2 * sw + 1) * C, nt);
                }
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int sw = w0 + u * npl;
                if (sw < Ws) {
                    float g[V];
                    unpack8(graw[u], g);
                    if (GP && (hedge || sw == 0 || sw == Ws - 1)) g_edge_add(gsrc, n, sh, sw, Hs, Ws, C, c, g);
                    unsigned code[V];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        code[e] = (amraw[u].x >> (8 * e)) & 0xffu;
                        code[e + 4] = (amraw[u].y >> (8 * e)) & 0xffu;
                    }
                    if (MODE == 0) {
                        // only the argmax position of a window carries gradient: pick its y per channel
                        float y4[4][V], ys[V];
#pragma unroll
                        for (int pos = 0; pos < 4; ++pos) unpack8(yraw[u][pos], y4[pos]);
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            ys[e] = code[e] == 0 ? y4[0][e] : (code[e] == 1 ? y4[1][e] : (code[e] == 2 ? y4[2][e] : y4[3][e]));
                            const float gh = g[e] * act_grad(k.z(ys[e], e), slope);
                            sb[e] += gh;
                            sg[e] += gh * k.xhat(ys[e], e);
                        }
                    } else {
#pragma unroll
                        for (int pos = 0; pos < 4; ++pos) {
                            float yv[V], res[V];
                            unpack8(yraw[u][pos], yv);
#pragma unroll
                            for (int e = 0; e < V; ++e) {
                                const float gh = (code[e] == (unsigned)pos ? g[e] : 0.f) * act_grad(k.z(yv[e], e), slope);
                                res[e] = k.affine ? k.ga[e] * k.is[e] * (gh - db[e] - k.xhat(yv[e], e) * dg[e]) : gh;
                            }
                            stv(d0 + ((long long)(pos >> 1) * Wp + dp + 2 * sw + (pos & 1)) * C, pack8(res));
                        }
                    }
                }
            }
        }
    }
    if (MODE == 0) {
        __shared__ float red[TPB * 2 * V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            red[(pl * CV + cv) * 2 * V + e] = sb[e];
            red[(pl * CV + cv) * 2 * V + V + e] = sg[e];
        }
        __syncthreads();
        if (pl == 0) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float a = 0.f, b = 0.f;
                for (int r2 = 0; r2 < npl; ++r2) {
                    a += red[(r2 * CV + cv) * 2 * V + e];
                    b += red[(r2 * CV + cv) * 2 * V + V + e];
                }
                partials[((long long)bid * 2 + 0) * C + c + e] = a;
                partials[((long long)bid * 2 + 1) * C + c + e] = b;
            }
        }
    }
}

bool fast_shape(int C) { return C % V == 0 && (C / V) <= TPB && TPB % (C / V) == 0; }

int rows_per_block(long long total_rows, int vec_per_row, int cus) {
    // >= 16 vectors per thread, but keep >= ~4 workgroups per CU
    int rpb = (16 * TPB + vec_per_row - 1) / vec_per_row;
    if (rpb < 1) rpb = 1;
    const long long max_rpb = total_rows / ((long long)cus * 4);
    if (rpb > max_rpb) rpb = (int)(max_rpb > 1 ? max_rpb : 1);
    return rpb;
}


// ---- adjoint of the bilinear x2 tail (align_corners=True) with the replicate-pad border folded in ---------------
// Separable: workgroup = (image, coarse row h, 64-channel chunk).  Phase 1 folds the <= 5 fine rows that touch h
// (weights Ah) and the padded border into T[fine column][64 ch] (fp32, LDS), every fine pixel read as a coalesced
// 16-byte vector; phase 2 applies the column weights Aw from LDS.  The general kernel gathers up to 25 fine pixels
// per output vector straight from memory with the index logic in the inner loop.
__global__ __launch_bounds__(256) void upsample2_bwd_fast_k(const h16_t* __restrict__ gsrc, int gp, h16_t* __restrict__ g,
                                                            int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) float T[];   // [Wo][64]
    const int Ho = 2 * H, Wo = 2 * W, Hp = Ho + 2 * gp, Wp = Wo + 2 * gp, chunks = C / 64;
    // XCD x (workgroups with blockIdx % 8 == x) walks a contiguous range of (image, row, chunk) items, so that the
    // workgroups of neighbouring coarse rows, which share 3 of their 5 fine rows, meet in one L2
    int b;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = blockIdx.x % 8, i = blockIdx.x / 8;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int ch = b % chunks; b /= chunks;
    const int h = b % H;
    const long long n = b / H;
    const int cv = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c = ch * 64 + cv * 8;
    const int i_lo = 2 * h - 2 < 0 ? 0 : 2 * h - 2, i_hi = 2 * h + 2 > Ho - 1 ? Ho - 1 : 2 * h + 2;
    // rows away from the top / bottom border: exactly the 5 fine rows 2h-2 .. 2h+2, no vertical fold -> their loads
    // are issued together (the general loop below waits for every load before it decides on the next one)
    const bool vint = i_lo > 0 && i_hi < Ho - 1 && i_hi - i_lo == 4;
    float rw[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        int h0, hp; float lh0, lh1;
        bilin(i_lo + r, H, Ho, h0, hp, lh0, lh1);
        rw[r] = (h0 == h ? lh0 : 0.f) + (h0 + hp == h ? lh1 : 0.f);
    }
    if (vint) {
        const h16_t* rowp = gsrc + ((n * Hp + i_lo + gp) * Wp) * C + c;
        for (int j = pl; j < Wo; j += 32) {
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = 0.f;
            const int jlo = (j == 0) ? 0 : j + gp, jhi = (j == Wo - 1) ? Wo + 2 * gp - 1 : j + gp;
            for (int bb = jlo; bb <= jhi; ++bb) {
                uint4 raw[5];
#pragma unroll
                for (int r = 0; r < 5; ++r) raw[r] = ldv(rowp + ((long long)r * Wp + bb) * C);
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    float v[V];
                    unpack8(raw[r], v);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] = fmaf(rw[r], v[e], acc[e]);
                }
            }
            float4* t = reinterpret_cast<float4*>(T + j * 64 + cv * 8);
            t[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            t[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    } else
    for (int j = pl; j < Wo; j += 32) {
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        const int jlo = (j == 0) ? 0 : j + gp, jhi = (j == Wo - 1) ? Wo + 2 * gp - 1 : j + gp;
        for (int i = i_lo; i <= i_hi; ++i) {
            int h0, hp; float lh0, lh1;
            bilin(i, H, Ho, h0, hp, lh0, lh1);
            const float wh = (h0 == h ? lh0 : 0.f) + (h0 + hp == h ? lh1 : 0.f);
            if (wh == 0.f) continue;
            const int ilo = (i == 0) ? 0 : i + gp, ihi = (i == Ho - 1) ? Ho + 2 * gp - 1 : i + gp;
            for (int a = ilo; a <= ihi; ++a)
                for (int bb = jlo; bb <= jhi; ++bb) {
                    float v[V];
                    unpack8(ldv(gsrc + ((n * Hp + a) * Wp + bb) * C + c), v);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] = fmaf(wh, v[e], acc[e]);
                }
        }
        float4* t = reinterpret_cast<float4*>(T + j * 64 + cv * 8);
        t[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        t[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    __syncthreads();
    for (int w = pl; w < W; w += 32) {
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        const int j_lo = 2 * w - 2 < 0 ? 0 : 2 * w - 2, j_hi = 2 * w + 2 > Wo - 1 ? Wo - 1 : 2 * w + 2;
        for (int j = j_lo; j <= j_hi; ++j) {
            int w0, wp; float lw0, lw1;
            bilin(j, W, Wo, w0, wp, lw0, lw1);
            const float ww = (w0 == w ? lw0 : 0.f) + (w0 + wp == w ? lw1 : 0.f);
            if (ww == 0.f) continue;
            const float4* t = reinterpret_cast<const float4*>(T + j * 64 + cv * 8);
            const float4 t0 = t[0], t1 = t[1];
            acc[0] = fmaf(ww, t0.x, acc[0]); acc[1] = fmaf(ww, t0.y, acc[1]); acc[2] = fmaf(ww, t0.z, acc[2]); acc[3] = fmaf(ww, t0.w, acc[3]);
            acc[4] = fmaf(ww, t1.x, acc[4]); acc[5] = fmaf(ww, t1.y, acc[5]); acc[6] = fmaf(ww, t1.z, acc[6]); acc[7] = fmaf(ww, t1.w, acc[7]);
        }
        stv(g + ((n * H + h) * W + w) * C + c, pack8(acc));
    }
}

}  // namespace

int kp_bn_act_fwd_fast(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       float slope, int tail, void* out, int out_pad, unsigned char* argmax, int N, int H, int W, int C,
                       hipStream_t st) {
    if (!fast_shape(C)) return KP_E_UNSUPPORTED;
    if (tail == KP_TAIL_MAXPOOL2 && ((H | W) & 1)) return KP_E_UNSUPPORTED;
    extern int g_kp_walk;
    const int walk = ((g_kp_walk & KP_WALK_FWD_REVERSE) ? 1 : 0) | ((g_kp_walk & KP_WALK_STREAM_LOADS) ? 2 : 0);
    BnP bn{mean, invstd, gamma, beta};
    const int Ho = tail == KP_TAIL_MAXPOOL2 ? H / 2 : (tail == KP_TAIL_UPSAMPLE2 ? 2 * H : H);
    const int Wo = tail == KP_TAIL_MAXPOOL2 ? W / 2 : (tail == KP_TAIL_UPSAMPLE2 ? 2 * W : W);
    const long long rows = (long long)N * (Ho + 2 * out_pad);
    const int work = tail == KP_TAIL_NONE ? 1 : 3;   // 4 source vectors per output vector
    const int rpb = rows_per_block(rows, (Wo + 2 * out_pad) * (C / V) * work, kp_device_cus());
    const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
    if (tail == KP_TAIL_NONE)
        bn_act_fwd_none_fast_k<<<grid, TPB, 0, st>>>((const h16_t*)y, bn, slope, (h16_t*)out, out_pad, H, W, C, rows, rpb, walk);
    else if (tail == KP_TAIL_MAXPOOL2)
        bn_act_fwd_tail_fast_k<1><<<grid, TPB, 0, st>>>((const h16_t*)y, bn, slope, (h16_t*)out, out_pad, argmax, H, W, C, rows, rpb, walk);
    else {
        const int HB = (Ho + 2 * out_pad + 1) / 2, WB = (Wo + 2 * out_pad + 1) / 2;
        const long long brows = (long long)N * HB;
        const int brpb = rows_per_block(brows, WB * (C / V) * 4, kp_device_cus());
        bn_act_fwd_up_fast_k<<<(unsigned)((brows + brpb - 1) / brpb), TPB, 0, st>>>((const h16_t*)y, bn, slope, (h16_t*)out,
                                                                                  out_pad, H, W, C, brows, brpb, walk);
    }
    return kp_check_launch("bn_act_fwd_fast");
}

// mode 0: reduce (partials [N*H][2][C], fully written); mode 1: apply
int kp_bn_bwd_fast(int mode, const void* gsrc, int gp, int tail, const unsigned char* argmax, const void* y,
                   const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                   float* partials, const float* sums, void* dy, int dp, int N, int H, int W, int C, hipStream_t st) {
    if (!(tail == KP_TAIL_NONE || tail == KP_TAIL_MAXPOOL2) || gp < 0 || gp > 1 || !fast_shape(C)) return KP_E_UNSUPPORTED;
    if (tail == KP_TAIL_MAXPOOL2 && ((H | W) & 1)) return KP_E_UNSUPPORTED;
    BnP bn{mean, invstd, gamma, beta};
    extern int g_kp_walk;
    const int walk = ((g_kp_walk & (mode ? KP_WALK_APPLY_REVERSE : KP_WALK_REDUCE_REVERSE)) ? 1 : 0) | ((g_kp_walk & KP_WALK_STREAM_LOADS) ? 2 : 0);
    const int pool = tail == KP_TAIL_MAXPOOL2;
    if (pool) {   // walk pooled rows
        const long long prows = (long long)N * (H / 2);
        const int prpb = rows_per_block(prows, (W / 2) * (C / V) * 4, kp_device_cus());
        const unsigned pgrid = (unsigned)((prows + prpb - 1) / prpb);
        if (mode == 0) kp_set_last_rows(pgrid);
        KP_QUERY_RETURN();
#define KP_BWDP(GPv, MODEv)                                                                                               \
    bn_bwd_pool_fast_k<GPv, MODEv><<<pgrid, TPB, 0, st>>>((const h16_t*)gsrc, argmax, (const h16_t*)y, bn, slope, partials, \
                                                         sums, (h16_t*)dy, dp, N, H, W, C, prows, prpb, walk)
        if (mode == 0) { if (gp) KP_BWDP(1, 0); else KP_BWDP(0, 0); }
        else { if (gp) KP_BWDP(1, 1); else KP_BWDP(0, 1); }
#undef KP_BWDP
        return kp_check_launch("bn_bwd_pool_fast");
    }
    const long long rows = mode ? (long long)N * (H + 2 * dp) : (long long)N * H;
    const int rpb = rows_per_block(rows, W * (C / V), kp_device_cus());
    const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
    if (mode == 0) kp_set_last_rows(grid);
    KP_QUERY_RETURN();
#define KP_BWD(POOLv, GPv, MODEv)                                                                                    \
    bn_bwd_fast_k<POOLv, GPv, MODEv><<<grid, TPB, 0, st>>>((const h16_t*)gsrc, argmax, (const h16_t*)y, bn, slope,  \
                                                           partials, sums, (h16_t*)dy, dp, N, H, W, C, rows, rpb, walk)
    if (mode == 0) {
        if (pool) { if (gp) KP_BWD(1, 1, 0); else KP_BWD(1, 0, 0); }
        else { if (gp) KP_BWD(0, 1, 0); else KP_BWD(0, 0, 0); }
    } else {
        if (pool) { if (gp) KP_BWD(1, 1, 1); else KP_BWD(1, 0, 1); }
        else { if (gp) KP_BWD(0, 1, 1); else KP_BWD(0, 0, 1); }
    }
#undef KP_BWD
    return kp_check_launch("bn_bwd_fast");
}

int kp_upsample2_bwd_fast(const void* gsrc, int gp, void* g, int N, int H, int W, int C, hipStream_t st) {
    const long long blocks = (long long)N * H * (C / 64);
    if (C % 64 != 0 || W > 128 || blocks > 0x7fffffffLL) return KP_E_UNSUPPORTED;
    upsample2_bwd_fast_k<<<(unsigned)blocks, 256, (size_t)2 * W * 64 * sizeof(float), st>>>((const h16_t*)gsrc, gp, (h16_t*)g, H, W, C);
    return kp_check_launch("upsample2_bwd_fast");
}
