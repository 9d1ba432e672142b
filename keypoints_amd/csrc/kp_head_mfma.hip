// MFMA paths of the 1x1 head (out_block = Conv2d(core_out, out, 1) + LeakyReLU, knn.py:123-125) for wide inputs
// (C a multiple of 128) and up to 64 output channels: the encoder heads (512 -> 64) and the keypoint heads (512 -> K); and
// for the 32-channel heads with more than 4 outputs (Pong-class encoder heads, 32 -> 16) -- gfx950.
//
// The head keeps fp32 weights and fp32 NCHW outputs / output gradients.  fp32 operands enter the bf16 matrix cores as
// a (hi, lo) pair of bf16 values (v = hi + lo up to 2^-17 relative): two or three MFMAs per product instead of one,
// which is free here -- all three kernels are bound by streaming the [pixels][C] bf16 activation.
//   fwd : D[k][px] = sum_c W[k][c] X[px][c]            A = W (hi, lo)   B = X (exact bf16)          2 MFMA / step
//   dx  : D[c][px] = sum_k W[k][c] dZ[px][k]           A = W^T (hi, lo) B = dZ (hi, lo)             3 MFMA / step
//   dw  : D[k][c]  = sum_px dZ[px][k] X[px][c]         A = dZ^T (hi, lo) B = X^T (exact bf16)       2 MFMA / step
// A wavefront owns 32 consecutive pixels (image and offset per lane: the group may straddle two images).  The accumulator layout (lane = pixel,
// register = output channel) is exactly the fp32 NCHW layout of `out`: 128-byte runs per register store; the NCHW
// gradient gives dZ^T fragments (8 consecutive pixels of one channel) as two 16-byte loads.  No LDS, no barriers.
#include "kp_mfma_util.h"

#ifndef KP_HEAD_DET_SPLITS
#define KP_HEAD_DET_SPLITS 128   // pixel splits (= partial gradients in the workspace) of the deterministic weight gradient
#endif

namespace {

// 8 fp32 values -> bf16 hi / lo fragments (round-to-nearest-even both times)
__device__ __forceinline__ void split8(const float (&v)[8], h16x8& hi, h16x8& lo) {
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = pack_h16x2(v[2 * i], v[2 * i + 1]);
        float h0, h1;
        unpack_h16x2(h[i], h0, h1);
        const float r0 = v[2 * i] - h0;
        const float r1 = v[2 * i + 1] - h1;
        l[i] = pack_h16x2(r0, r1);
    }
    const u32x4 hv = {h[0], h[1], h[2], h[3]}, lv = {l[0], l[1], l[2], l[3]};
    hi = __builtin_bit_cast(h16x8, hv);
    lo = __builtin_bit_cast(h16x8, lv);
}

__device__ __forceinline__ f32x16 mma(h16x8 a, h16x8 b, f32x16 c) {
    return kp_mfma16(a, b, c, 0, 0, 0);
}

// rows k of W[K][C] (fp32), 8 consecutive c starting at c0: zero rows beyond K
__device__ __forceinline__ void load_w8(const float* __restrict__ w, int k, int K, int C, int c0, float (&v)[8]) {
    if (k < K) {
        const float4 a = *reinterpret_cast<const float4*>(w + (long long)k * C + c0);
        const float4 b = *reinterpret_cast<const float4*>(w + (long long)k * C + c0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
}

// ---- forward -----------------------------------------------------------------------------------------------------
template <int KT>   // 32-channel tiles of the output (K <= 32 * KT)
__global__ __launch_bounds__(256) void head_fwd_mfma_k(const h16_t* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float slope,
                                                       float* __restrict__ out, long long groups, int HW, int C, int K) {
    const int lane = threadIdx.x & 63;
    const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= groups) return;
    const int px = lane & 31, h = lane >> 5;
    const long long m0 = g * 32;
    const h16_t* xp = x + (m0 + px) * C + h * 8;
    f32x16 acc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[kt][i] = 0.f;
    for (int c0 = 0; c0 < C; c0 += 16) {
        const h16x8 xb = *reinterpret_cast<const h16x8*>(xp + c0);
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            float wv[8];
            load_w8(w, kt * 32 + px, K, C, c0 + h * 8, wv);
            h16x8 whi, wlo;
            split8(wv, whi, wlo);
            acc[kt] = mma(whi, xb, acc[kt]);
            acc[kt] = mma(wlo, xb, acc[kt]);
        }
    }
    // (image, offset) per lane: a group of 32 pixels may straddle two images when HW % 32 != 0 (84 x 84 maps)
    const long long n = (m0 + px) / HW, hw = (m0 + px) - n * HW;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = kt * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
            if (k < K) out[(n * K + k) * HW + hw] = act_fwd(acc[kt][i] + (bias ? bias[k] : 0.f), slope);
        }
}

// The 32-channel head (Pong-class encoder heads: 32 -> 16 over 256 x 84 x 84) as a loop: the kernel above takes ONE group of 32
// pixels per wavefront and loads + splits its weight fragments for it (2 x ~50 VALU, 16 loads per 4 MFMAs); here a wavefront keeps
// the (hi, lo) fragments of both 16-channel steps in registers and walks groups wave, wave + waves, ..., the next group's
// pixels requested before the current one is multiplied.  Same operands into the same MFMAs in the same order: bit-identical.
// BNIN: x is the producing layer's conv output, `bn` its BatchNorm / activation (KpBnIn, kp_common.h), applied to the two
// 16-byte pixel fragments of a lane -- always the same 2 x 8 channels, constants in registers.
// TOUT (kp_head_fwd_t, K == 16): `out` is T [M][16] -- the pixel-major 16-bit tensor the keypoint bottleneck's T form and the
// decoder read (kp_bottleneck_t.hip) -- instead of fp32 planes: the two half-waves' 8-byte pieces of a pixel (channels 4h .. 4h + 3
// and 8 + 4h ..) are paired with v_permlane32_swap into one 16-byte store per lane, a wavefront writes 1 KB contiguous.
template <int KT, bool BNIN, bool TOUT = false>
__global__ __launch_bounds__(256) void head_fwd_mfma32_k(const h16_t* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float slope,
                                                         float* __restrict__ out, long long groups, int HW, int K, KpBnIn bn,
                                                         int hb = 0, const h16_t* __restrict__ x1 = nullptr, float* __restrict__ out1 = nullptr) {
    constexpr int C = 32;
    static_assert(!TOUT || KT == 1, "T output: K == 16");
    const int lane = threadIdx.x & 63;
    // hb != 0 (kp_head_fwd_bnin_pair / kp_head_fwd_t with pair): the grid is two launches of hb workgroups, one per half-batch of
    // `groups` pixel groups -- the second half reads x1, writes out1 and takes the second row of the [2][C] mean / invstd
    int bid = blockIdx.x, nblk = gridDim.x;
    if (hb) {
        nblk = hb;
        if (bid >= hb) {
            bid -= hb; x = x1; out = out1;
            if (BNIN && bn.mean) { bn.mean += C; bn.invstd += C; }
        }
    }
    const long long waves = (long long)nblk * 4;
    long long g = (long long)bid * 4 + (threadIdx.x >> 6);
    if (g >= groups) return;
    const int px = lane & 31, h = lane >> 5;
    h16x8 whi[KT][2], wlo[KT][2];
    float bk[KT][16];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            float wv[8];
            load_w8(w, kt * 32 + px, K, C, s * 16 + h * 8, wv);
            split8(wv, whi[kt][s], wlo[kt][s]);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = kt * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
            bk[kt][i] = (bias && k < K) ? bias[k] : 0.f;
        }
    }
    KpBnIn8 k0, k1;
    if (BNIN) { k0.load(bn, h * 8); k1.load(bn, 16 + h * 8); }
    const h16_t* xp = x + px * C + h * 8;
    uint4 a0 = *reinterpret_cast<const uint4*>(xp + g * 32 * C), a1 = *reinterpret_cast<const uint4*>(xp + g * 32 * C + 16);
    for (; g < groups; g += waves) {
        uint4 c0 = a0, c1 = a1;
        const long long gn = g + waves < groups ? g + waves : g;   // (the last trip re-reads its own pixels)
        a0 = *reinterpret_cast<const uint4*>(xp + gn * 32 * C);
        a1 = *reinterpret_cast<const uint4*>(xp + gn * 32 * C + 16);
        if (BNIN) { c0 = k0.apply(c0, bn.slope); c1 = k1.apply(c1, bn.slope); }
        const h16x8 x0 = __builtin_bit_cast(h16x8, u32x4{c0.x, c0.y, c0.z, c0.w}), x1 = __builtin_bit_cast(h16x8, u32x4{c1.x, c1.y, c1.z, c1.w});
        const long long m = g * 32 + px, n = m / HW, hw = m - n * HW;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
            acc = mma(whi[kt][0], x0, acc);
            acc = mma(wlo[kt][0], x0, acc);
            acc = mma(whi[kt][1], x1, acc);
            acc = mma(wlo[kt][1], x1, acc);
            if constexpr (TOUT) {
                uint32_t p[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    p[q] = pack_h16x2(act_fwd(acc[2 * q] + bk[0][2 * q], slope), act_fwd(acc[2 * q + 1] + bk[0][2 * q + 1], slope));
                // lanes h = 0: (p0, p1) = channels 0-3, (p2, p3) = 8-11; h = 1: 4-7, 12-15 -> h = 0: 0-7, h = 1: 8-15
                const auto s0 = __builtin_amdgcn_permlane32_swap(p[0], p[2], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(p[1], p[3], false, false);
                *reinterpret_cast<uint4*>(reinterpret_cast<h16_t*>(out) + m * 16 + h * 8) = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int k = kt * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                    if (k < K) out[(n * K + k) * HW + hw] = act_fwd(acc[i] + bk[kt][i], slope);
                }
            }
        }
    }
}

// dZ fragment of this lane for reduction slice kk: 8 consecutive output channels of pixel (n, hw)
__device__ __forceinline__ void load_dz_px(const float* __restrict__ gout, const float* __restrict__ out, long long n,
                                           int hw, int HW, int K, int k0, float slope, float (&v)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = k0 + i;
        if (k < K) {
            const long long idx = (n * K + k) * HW + hw;
            v[i] = gout[idx] * act_grad(out[idx], slope);
        } else {
            v[i] = 0.f;
        }
    }
}

// the same from T [M][K] gradient / output tensors (kp_head_bwd_t; K a multiple of 8): one 16-byte load each
__device__ __forceinline__ void load_dz_px_t(const h16_t* __restrict__ gout, const h16_t* __restrict__ out, long long m, int K,
                                             int k0, float slope, float (&v)[8]) {
    if (k0 < K) {
        const uint4 gq = *reinterpret_cast<const uint4*>(gout + m * K + k0), oq = *reinterpret_cast<const uint4*>(out + m * K + k0);
        const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, ow[4] = {oq.x, oq.y, oq.z, oq.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float g0, g1, o0, o1;
            unpack_h16x2(gw[i], g0, g1);
            unpack_h16x2(ow[i], o0, o1);
            v[2 * i] = g0 * act_grad(o0, slope);
            v[2 * i + 1] = g1 * act_grad(o1, slope);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 0.f;
    }
}

// ---- input gradient ------------------------------------------------------------------------------------------------
template <int KS, bool TIN = false>   // 16-channel reduction slices (K <= 16 * KS); TIN: gout / out are T [M][K]
__global__ __launch_bounds__(256) void head_dx_mfma_k(const float* __restrict__ gout, const float* __restrict__ out,
                                                      const float* __restrict__ w, float slope, h16_t* __restrict__ dx,
                                                      long long groups, int HW, int C, int K) {
    const int lane = threadIdx.x & 63;
    const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= groups) return;
    const int px = lane & 31, h = lane >> 5;
    const long long m0 = g * 32, n = (m0 + px) / HW;
    const int hw = (int)((m0 + px) - n * HW);
    h16x8 zhi[KS], zlo[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        float v[8];
        if constexpr (TIN) load_dz_px_t(reinterpret_cast<const h16_t*>(gout), reinterpret_cast<const h16_t*>(out), m0 + px, K, kk * 16 + h * 8, slope, v);
        else load_dz_px(gout, out, n, hw, HW, K, kk * 16 + h * 8, slope, v);
        split8(v, zhi[kk], zlo[kk]);
    }
    h16_t* dp = dx + (m0 + px) * C + 4 * h;
    for (int ct = 0; ct < C; ct += 32) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            // A = W^T: row = input channel ct + px, 8 consecutive output channels (coalesced over the lanes per load)
            float wv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = kk * 16 + h * 8 + i;
                wv[i] = k < K ? w[(long long)k * C + ct + px] : 0.f;
            }
            h16x8 whi, wlo;
            split8(wv, whi, wlo);
            acc = mma(whi, zhi[kk], acc);
            acc = mma(wlo, zhi[kk], acc);
            acc = mma(whi, zlo[kk], acc);
        }
        // lane = pixel, register i = input channel ct + 8*(i>>2) + 4*h + (i&3): 4 consecutive channels per 8-byte store
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint2 pk;
            pk.x = pack_h16x2(acc[4 * q], acc[4 * q + 1]);
            pk.y = pack_h16x2(acc[4 * q + 2], acc[4 * q + 3]);
            *reinterpret_cast<uint2*>(dp + ct + 8 * q) = pk;
        }
    }
}

// ---- weight / bias gradient ----------------------------------------------------------------------------------------
// wavefront = (pixel range, 32 * CT input channels): CT x KT accumulator tiles (CT = 4: 128-channel ranges of the wide
// heads; CT = 1: the 32-channel heads of the Pong-class nets); fp32 atomics into dw / db at the end, or per-split partials
template <int KT, int CT>
__global__ __launch_bounds__(256) void head_dw_mfma_k(const float* __restrict__ gout, const float* __restrict__ out,
                                                      const h16_t* __restrict__ x, float slope, float* __restrict__ dw,
                                                      float* __restrict__ db, float* __restrict__ ws, long long groups,
                                                      int gpw, int HW, int C, int K) {
    const int lane = threadIdx.x & 63;
    const int cranges = C / (32 * CT);
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int cr = (int)(wid % cranges);
    const long long sp = wid / cranges;
    const long long g0 = sp * gpw, g1 = g0 + gpw < groups ? g0 + gpw : groups;
    if (g0 >= g1) return;
    const int r = lane & 31, h = lane >> 5;     // r: output channel row (A) / input channel column (B)
    f32x16 acc[KT][CT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[kt][t][i] = 0.f;
    float bsum[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) bsum[kt] = 0.f;

    for (long long g = g0; g < g1; ++g) {
        const long long m0 = g * 32;
#pragma unroll
        for (int s = 0; s < 2; ++s) {           // two slices of 16 pixels
            const int p0 = s * 16 + h * 8;       // this lane's 8 consecutive pixels: inside ONE image (HW % 8 == 0)
            const long long n = (m0 + p0) / HW;
            const int hw0 = (int)((m0 + p0) - n * HW) - p0;
            h16x8 zhi[KT], zlo[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int k = kt * 32 + r;
                float v[8];
                if (k < K) {
                    const long long idx = (n * K + k) * HW + hw0 + p0;
                    const float4 ga = *reinterpret_cast<const float4*>(gout + idx), gb = *reinterpret_cast<const float4*>(gout + idx + 4);
                    const float4 oa = *reinterpret_cast<const float4*>(out + idx), ob = *reinterpret_cast<const float4*>(out + idx + 4);
                    v[0] = ga.x * act_grad(oa.x, slope); v[1] = ga.y * act_grad(oa.y, slope);
                    v[2] = ga.z * act_grad(oa.z, slope); v[3] = ga.w * act_grad(oa.w, slope);
                    v[4] = gb.x * act_grad(ob.x, slope); v[5] = gb.y * act_grad(ob.y, slope);
                    v[6] = gb.z * act_grad(ob.z, slope); v[7] = gb.w * act_grad(ob.w, slope);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = 0.f;
                }
                if (cr == 0) bsum[kt] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                split8(v, zhi[kt], zlo[kt]);
            }
            const h16_t* xp = x + (m0 + p0) * C + cr * (32 * CT) + r;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                h16x8 xb;      // column = input channel cr*128 + t*32 + r, 8 consecutive pixels (stride C)
#pragma unroll
                for (int i = 0; i < 8; ++i) xb[i] = (short)xp[(long long)i * C + t * 32];
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    acc[kt][t] = mma(zhi[kt], xb, acc[kt][t]);
                    acc[kt][t] = mma(zlo[kt], xb, acc[kt][t]);
                }
            }
        }
    }
    // D[k][c]: lane column = c (r), rows k = 8*(i>>2) + 4*h + (i&3)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int t = 0; t < CT; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = kt * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
                if (k < K) {
                    const long long e = (long long)k * C + cr * (32 * CT) + t * 32 + r;
                    if (ws) ws[sp * ((long long)K * C + K) + e] = acc[kt][t][i];   // this pixel split's partial (plain store)
                    else unsafeAtomicAdd(dw + e, acc[kt][t][i]);
                }
            }
    if ((db || ws) && cr == 0) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const float tot = bsum[kt] + __shfl_xor(bsum[kt], 32, 64);   // the two 8-pixel halves
            const int k = kt * 32 + r;
            if (h == 0 && k < K) {
                if (ws) ws[sp * ((long long)K * C + K) + (long long)K * C + k] = tot;
                else unsafeAtomicAdd(db + k, tot);
            }
        }
    }
}

// The 32-channel head's weight gradient (CT = 1 above) with the input tile staged through LDS: a wavefront copies the 32 pixels
// x 32 channels of a group (2 KB, two 16-byte loads per lane) into its own 2 KB of LDS, pixel-major, and reads the two B
// fragments (lane: channel lane % 32, 8 consecutive pixels) with hardware transpose reads (2 x ds_read_b64_tr_b16 each) -- the
// kernel above gathers them with sixteen 2-byte loads at a channel stride.  Same operands into the same MFMAs in the same
// order: bit-identical results.  BNIN: the producing layer's BatchNorm / activation (KpBnIn) applied to the 16-byte chunks
// between the load and the LDS store; a lane's chunk is always the same 8 channels (constants in registers) -- on the
// gathered elements the transform cost 70 us per launch (164 vs 93), here ~10.
template <int KT, bool BNIN>
__global__ __launch_bounds__(256) void head_dw_mfma32_k(const float* __restrict__ gout, const float* __restrict__ out,
                                                        const h16_t* __restrict__ x, float slope, float* __restrict__ dw,
                                                        float* __restrict__ db, float* __restrict__ ws, long long groups,
                                                        int gpw, int HW, int K, KpBnIn bn) {
    constexpr int C = 32;
    __shared__ __attribute__((aligned(16))) unsigned char tile[4][32 * C * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long sp = (long long)blockIdx.x * 4 + wave;
    const long long g0 = sp * gpw, g1 = g0 + gpw < groups ? g0 + gpw : groups;
    if (g0 >= g1) return;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc[KT];
    float bsum[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        bsum[kt] = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[kt][i] = 0.f;
    }
    KpBnIn8 k8;
    if (BNIN) k8.load(bn, (lane & 3) * 8);
    unsigned char* const tl = tile[wave];
    // transpose-read lane roles (as in wgrad3x3_small_band_k): t = lane & 15 -> (pixel t / 4, 4-channel chunk t % 4); bit 4 -> channels 16..31
    const int t16 = lane & 15, g16 = (lane >> 4) & 1;
    const int tr_off = (8 * h + (t16 >> 2)) * (C * 2) + (16 * g16 + (t16 & 3) * 4) * 2;
    for (long long g = g0; g < g1; ++g) {
        const long long m0 = g * 32;
        const uint4* xs = reinterpret_cast<const uint4*>(x + m0 * C);
        uint4 c0 = xs[lane], c1 = xs[lane + 64];             // chunk id = pixel * 4 + (8-channel chunk): lane % 4 is the chunk of both
        h16x8 zhi[2][KT], zlo[2][KT];
#pragma unroll
        for (int s = 0; s < 2; ++s) {           // two slices of 16 pixels
            const int p0 = s * 16 + h * 8;       // this lane's 8 consecutive pixels: inside ONE image (HW % 8 == 0)
            const long long n = (m0 + p0) / HW;
            const int hw0 = (int)((m0 + p0) - n * HW);
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const int k = kt * 32 + r;
                float v[8];
                if (k < K) {
                    const long long idx = (n * K + k) * HW + hw0;
                    const float4 ga = *reinterpret_cast<const float4*>(gout + idx), gb = *reinterpret_cast<const float4*>(gout + idx + 4);
                    const float4 oa = *reinterpret_cast<const float4*>(out + idx), ob = *reinterpret_cast<const float4*>(out + idx + 4);
                    v[0] = ga.x * act_grad(oa.x, slope); v[1] = ga.y * act_grad(oa.y, slope);
                    v[2] = ga.z * act_grad(oa.z, slope); v[3] = ga.w * act_grad(oa.w, slope);
                    v[4] = gb.x * act_grad(ob.x, slope); v[5] = gb.y * act_grad(ob.y, slope);
                    v[6] = gb.z * act_grad(ob.z, slope); v[7] = gb.w * act_grad(ob.w, slope);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = 0.f;
                }
                bsum[kt] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                split8(v, zhi[s][kt], zlo[s][kt]);
            }
        }
        if (BNIN) { c0 = k8.apply(c0, bn.slope); c1 = k8.apply(c1, bn.slope); }
        *reinterpret_cast<uint4*>(tl + lane * 16) = c0;
        *reinterpret_cast<uint4*>(tl + (lane + 64) * 16) = c1;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned char* b = tl + s * 16 * (C * 2) + tr_off;
            const h16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4*)(b));
            const h16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4*)(b + 4 * (C * 2)));
            const h16x8 xb = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                acc[kt] = mma(zhi[s][kt], xb, acc[kt]);
                acc[kt] = mma(zlo[s][kt], xb, acc[kt]);
            }
        }
    }
    // D[k][c]: lane column = c (r), rows k = 8*(i>>2) + 4*h + (i&3)
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int k = kt * 32 + 8 * (i >> 2) + 4 * h + (i & 3);
            if (k < K) {
                const long long e = (long long)k * C + r;
                if (ws) ws[sp * ((long long)K * C + K) + e] = acc[kt][i];   // this pixel split's partial (plain store)
                else unsafeAtomicAdd(dw + e, acc[kt][i]);
            }
        }
    if (db || ws) {
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            const float tot = bsum[kt] + __shfl_xor(bsum[kt], 32, 64);   // the two 8-pixel halves
            const int k = kt * 32 + r;
            if (h == 0 && k < K) {
                if (ws) ws[sp * ((long long)K * C + K) + (long long)K * C + k] = tot;
                else unsafeAtomicAdd(db + k, tot);
            }
        }
    }
}

// The 32 -> 16 head's weight / bias gradient from T [M][16] gradient / output tensors (kp_head_bwd_t).  The group's 32 x 16
// elements of dZ = gout * act'(out) are exactly one 16-byte chunk per lane (pixel lane / 2, channels 8 (lane % 2) ..): formed in
// fp32, split into the (hi, lo) pair and staged pixel-major through 2 x 1 KB of wave-private LDS beside the input tile; the A
// fragments (row = output channel, 8 consecutive pixels) come back through the same transpose reads as the B fragments
// (rows 16-31 of the 32 x 32 tile are zero).  The bias gradient is summed in the chunk layout (8 running sums per lane) and
// folded over the 32 lanes of a channel half at the end.
template <bool BNIN>
__global__ __launch_bounds__(256) void head_dw_mfma32_t_k(const h16_t* __restrict__ gout, const h16_t* __restrict__ out,
                                                          const h16_t* __restrict__ x, float slope, float* __restrict__ dw,
                                                          float* __restrict__ db, float* __restrict__ ws, long long groups,
                                                          int gpw, KpBnIn bn) {
    constexpr int C = 32, K = 16;
    __shared__ __attribute__((aligned(16))) unsigned char tile[4][32 * C * 2];
    __shared__ __attribute__((aligned(16))) unsigned char ztile[4][2][32 * K * 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long sp = (long long)blockIdx.x * 4 + wave;
    const long long g0 = sp * gpw, g1 = g0 + gpw < groups ? g0 + gpw : groups;
    if (g0 >= g1) return;
    const int r = lane & 31, h = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bz[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bz[e] = 0.f;
    KpBnIn8 k8;
    if (BNIN) k8.load(bn, (lane & 3) * 8);
    unsigned char* const tl = tile[wave];
    unsigned char* const zh = ztile[wave][0];
    unsigned char* const zl = ztile[wave][1];
    const int t16 = lane & 15, g16 = (lane >> 4) & 1;
    const int tr_off = (8 * h + (t16 >> 2)) * (C * 2) + (16 * g16 + (t16 & 3) * 4) * 2;
    const int tz_off = (8 * h + (t16 >> 2)) * (K * 2) + ((t16 & 3) * 4) * 2;
    for (long long g = g0; g < g1; ++g) {
        const long long m0 = g * 32;
        const uint4* xs = reinterpret_cast<const uint4*>(x + m0 * C);
        uint4 c0 = xs[lane], c1 = xs[lane + 64];
        const uint4 gq = reinterpret_cast<const uint4*>(gout + m0 * K)[lane], oq = reinterpret_cast<const uint4*>(out + m0 * K)[lane];
        {
            const uint32_t gw[4] = {gq.x, gq.y, gq.z, gq.w}, ow[4] = {oq.x, oq.y, oq.z, oq.w};
            float v[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float a0, a1, o0, o1;
                unpack_h16x2(gw[i], a0, a1);
                unpack_h16x2(ow[i], o0, o1);
                v[2 * i] = a0 * act_grad(o0, slope);
                v[2 * i + 1] = a1 * act_grad(o1, slope);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) bz[e] += v[e];
            h16x8 vhi, vlo;
            split8(v, vhi, vlo);
            *reinterpret_cast<h16x8*>(zh + lane * 16) = vhi;
            *reinterpret_cast<h16x8*>(zl + lane * 16) = vlo;
        }
        if (BNIN) { c0 = k8.apply(c0, bn.slope); c1 = k8.apply(c1, bn.slope); }
        *reinterpret_cast<uint4*>(tl + lane * 16) = c0;
        *reinterpret_cast<uint4*>(tl + (lane + 64) * 16) = c1;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned char* b = tl + s * 16 * (C * 2) + tr_off;
            const h16x4 q0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4*)(b));
            const h16x4 q1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4*)(b + 4 * (C * 2)));
            const h16x8 xb = {q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
            const unsigned char* za = zh + s * 16 * (K * 2) + tz_off;
            const unsigned char* zb = zl + s * 16 * (K * 2) + tz_off;
            const h16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4*)(za));
            const h16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4*)(za + 4 * (K * 2)));
            const h16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4*)(zb));
            const h16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) h16x4*)(zb + 4 * (K * 2)));
            h16x8 fhi = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            h16x8 flo = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            if (g16) {   // rows 16..31 of the A tile: no such output channel
                fhi = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
                flo = h16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
            acc = mma(fhi, xb, acc);
            acc = mma(flo, xb, acc);
        }
    }
    // D[k][c]: lane column = c (r), rows k = 8*(i>>2) + 4*h + (i&3): i < 8 are the 16 output channels
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = 8 * (i >> 2) + 4 * h + (i & 3);
        const long long e = (long long)k * C + r;
        if (ws) ws[sp * ((long long)K * C + K) + e] = acc[i];
        else unsafeAtomicAdd(dw + e, acc[i]);
    }
    if (db || ws) {
#pragma unroll
        for (int o = 2; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) bz[e] += __shfl_xor(bz[e], o, 64);
        if (lane < 2) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (ws) ws[sp * ((long long)K * C + K) + (long long)K * C + lane * 8 + e] = bz[e];
                else unsafeAtomicAdd(db + lane * 8 + e, bz[e]);
            }
        }
    }
}


// wide heads (C a multiple of 128), and the 32-channel heads with more than 4 outputs (Pong-class encoder heads: 32 -> 16 over
// 84 x 84 maps, where the K <= 16 VALU kernels of kp_head_fast.hip run at 1.8 TB/s).  Groups of 32 pixels may straddle images
// (per-lane image / offset); the weight gradient reads 8-pixel runs of the NCHW gradient: HW % 8 == 0.
bool mfma_shape(long long M, int HW, int C, int K) {
    const bool wide = C >= 128 && C % 128 == 0, narrow = C == 32 && K > 4;
    return (wide || narrow) && K >= 1 && K <= 64 && HW % 8 == 0 && M >= 32 && M % 32 == 0;
}
int chan_tiles(int C) { return C % 128 == 0 ? 4 : 1; }
bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr) { return (((size_t)a | (size_t)b | (size_t)c) & 15) == 0; }

}  // namespace

// bn != nullptr (kp_head_fwd_bnin / kp_head_bwd_bnin): the 32-channel shapes only
bool kp_head_mfma_bnin_shape(long long M, int HW, int C, int K) { return mfma_shape(M, HW, C, K) && C == 32; }
bool kp_head_mfma_shape(long long M, int HW, int C, int K) { return mfma_shape(M, HW, C, K); }
// the FORWARD also takes the 32-channel heads with K <= 4 (Pong-class keypoint heads, 32 -> K): the looped kernel streams their
// input at the rate of the 16-output head, the K <= 16 VALU kernel (kp_head_fast.hip) is 25 % slower once it applies the BatchNorm input
bool kp_head_mfma_fwd_shape(long long M, int HW, int C, int K) {
    return mfma_shape(M, HW, C, K) || (C == 32 && K >= 1 && K <= 4 && HW % 8 == 0 && M >= 32 && M % 32 == 0);
}

// x1 / out1 != nullptr (the pair forms): M is the pixel count of ONE half; the second half is read at x1, written at out1, with
// the second row of bn's [2][C] mean / invstd -- each half launched as a separate call of M pixels would be (bit-identical)
int kp_head_fwd_mfma(const void* x, const float* w, const float* bias, float slope, float* out, long long M, int HW,
                     int C, int K, hipStream_t st, const KpBnIn* bn, const void* x1, float* out1) {
    if (!kp_head_mfma_fwd_shape(M, HW, C, K) || !aligned16(x, w)) return KP_E_UNSUPPORTED;
    if (bn && C != 32) return KP_E_UNSUPPORTED;
    const bool pair = x1 != nullptr;
    if (pair && (C != 32 || !bn || !out1 || !aligned16(x1))) return KP_E_UNSUPPORTED;
    const long long groups = M / 32;
    const unsigned grid = (unsigned)((groups + 3) / 4);
    if (C == 32) {   // ~8 groups per wavefront
        long long wg = (groups / 8 + 3) / 4;
        wg = wg < 1 ? 1 : wg;
        const KpBnIn b = bn ? *bn : KpBnIn{};
#define KP_FW(KTv, Bv) head_fwd_mfma32_k<KTv, Bv><<<(unsigned)(pair ? 2 * wg : wg), 256, 0, st>>>((const h16_t*)x, w, bias, slope, out, groups, HW, K, b, pair ? (int)wg : 0, (const h16_t*)x1, out1)
        if (bn) { if (K <= 32) KP_FW(1, true); else KP_FW(2, true); }
        else { if (K <= 32) KP_FW(1, false); else KP_FW(2, false); }
#undef KP_FW
    }
    else if (K <= 32) head_fwd_mfma_k<1><<<grid, 256, 0, st>>>((const h16_t*)x, w, bias, slope, out, groups, HW, C, K);
    else head_fwd_mfma_k<2><<<grid, 256, 0, st>>>((const h16_t*)x, w, bias, slope, out, groups, HW, C, K);
    return kp_check_launch("head_fwd_mfma");
}

int kp_head_dx_mfma(const float* gout, const float* out, const float* w, float slope, void* dx, long long M, int HW,
                    int C, int K, hipStream_t st) {
    if (!mfma_shape(M, HW, C, K) || !aligned16(dx)) return KP_E_UNSUPPORTED;
    const long long groups = M / 32;
    const unsigned grid = (unsigned)((groups + 3) / 4);
#define KP_DX(KSv) head_dx_mfma_k<KSv><<<grid, 256, 0, st>>>(gout, out, w, slope, (h16_t*)dx, groups, HW, C, K)
    if (K <= 16) KP_DX(1);
    else if (K <= 32) KP_DX(2);
    else if (K <= 48) KP_DX(3);
    else KP_DX(4);
#undef KP_DX
    return kp_check_launch("head_dx_mfma");
}

// pixel splits of the weight-gradient kernel for this shape (0: not an MFMA shape).  Deterministic mode (a workspace of
// one partial gradient per split) caps them so that the workspace stays small.
static long long dw_mfma_plan(long long M, int HW, int C, int K, bool det, int& gpw) {
    if (!mfma_shape(M, HW, C, K)) return 0;
    const long long groups = M / 32;
    const int cranges = C / (32 * chan_tiles(C));
    long long splits = 2048 / cranges;          // ~2048 wavefronts: pixel splits x channel ranges
    // deterministic mode keeps one partial gradient per split in the workspace: capped at KP_HEAD_DET_SPLITS for the wide
    // heads (32 K floats each), not for the 32-channel ones (512 floats each: 2048 splits are 4 MB)
    if (det && splits > KP_HEAD_DET_SPLITS && (long long)K * C > 4096) splits = KP_HEAD_DET_SPLITS;
    if (splits > groups) splits = groups;
    gpw = (int)((groups + splits - 1) / splits);
    return (groups + gpw - 1) / gpw;
}
long long kp_head_dw_parts_mfma(long long M, int HW, int C, int K) { int gpw; return dw_mfma_plan(M, HW, C, K, true, gpw); }

int kp_head_dw_mfma(const float* gout, const float* out, const void* x, float slope, float* dw, float* db, float* ws,
                    long long M, int HW, int C, int K, hipStream_t st, const KpBnIn* bn) {
    if (!mfma_shape(M, HW, C, K) || !aligned16(gout, out)) return KP_E_UNSUPPORTED;
    if (bn && !kp_head_mfma_bnin_shape(M, HW, C, K)) return KP_E_UNSUPPORTED;
    const long long groups = M / 32;
    const int ct = chan_tiles(C), cranges = C / (32 * ct);
    int gpw;
    const long long splits = dw_mfma_plan(M, HW, C, K, ws != nullptr, gpw);
    const unsigned grid = (unsigned)((splits * cranges + 3) / 4);
#define KP_DW(KTv, CTv) head_dw_mfma_k<KTv, CTv><<<grid, 256, 0, st>>>(gout, out, (const h16_t*)x, slope, dw, db, ws, groups, gpw, HW, C, K)
    if (C == 32 && aligned16(x)) {   // (cranges == 1: a wavefront = a pixel split)
        const KpBnIn b = bn ? *bn : KpBnIn{};
        if (bn) {
            if (K <= 32) head_dw_mfma32_k<1, true><<<grid, 256, 0, st>>>(gout, out, (const h16_t*)x, slope, dw, db, ws, groups, gpw, HW, K, b);
            else head_dw_mfma32_k<2, true><<<grid, 256, 0, st>>>(gout, out, (const h16_t*)x, slope, dw, db, ws, groups, gpw, HW, K, b);
        } else {
            if (K <= 32) head_dw_mfma32_k<1, false><<<grid, 256, 0, st>>>(gout, out, (const h16_t*)x, slope, dw, db, ws, groups, gpw, HW, K, b);
            else head_dw_mfma32_k<2, false><<<grid, 256, 0, st>>>(gout, out, (const h16_t*)x, slope, dw, db, ws, groups, gpw, HW, K, b);
        }
    }
    else if (bn) return KP_E_UNSUPPORTED;   // (x of a foreign caller not 16-byte aligned: the gather kernel has no fused form)
    else if (K <= 32 && ct == 4) KP_DW(1, 4);
    else if (ct == 4) KP_DW(2, 4);
    else if (K <= 32) KP_DW(1, 1);
    else KP_DW(2, 1);
#undef KP_DW
    return kp_check_launch("head_dw_mfma");
}

// ---- T-layout forms (kp_head_fwd_t / kp_head_bwd_t): the 32 -> 16 encoder head of the Pong-class nets between 16-bit NHWC tensors ----
bool kp_head_t_shape(long long M, int HW, int C, int K) { return C == 32 && K == 16 && mfma_shape(M, HW, C, K); }

int kp_head_fwd_mfma_t(const void* x, const float* w, const float* bias, float slope, void* out_t, long long M, int HW, int C, int K,
                       hipStream_t st, const KpBnIn* bn, const void* x1, void* out1) {
    if (!kp_head_t_shape(M, HW, C, K) || !aligned16(x, w, out_t)) return KP_E_UNSUPPORTED;
    const bool pair = x1 != nullptr;
    if (pair && (!bn || !out1 || !aligned16(x1, out1))) return KP_E_UNSUPPORTED;
    const long long groups = M / 32;
    long long wg = (groups / 8 + 3) / 4;   // ~8 groups per wavefront, as the fp32-output form
    wg = wg < 1 ? 1 : wg;
    const KpBnIn b = bn ? *bn : KpBnIn{};
    const unsigned grid = (unsigned)(pair ? 2 * wg : wg);
    const int hb = pair ? (int)wg : 0;
    if (bn) head_fwd_mfma32_k<1, true, true><<<grid, 256, 0, st>>>((const h16_t*)x, w, bias, slope, (float*)out_t, groups, HW, K, b, hb, (const h16_t*)x1, (float*)out1);
    else head_fwd_mfma32_k<1, false, true><<<grid, 256, 0, st>>>((const h16_t*)x, w, bias, slope, (float*)out_t, groups, HW, K, b);
    return kp_check_launch("head_fwd_mfma_t");
}

int kp_head_dx_mfma_t(const void* gout_t, const void* out_t, const float* w, float slope, void* dx, long long M, int HW, int C, int K,
                      hipStream_t st) {
    if (!kp_head_t_shape(M, HW, C, K) || !aligned16(gout_t, out_t, dx)) return KP_E_UNSUPPORTED;
    const long long groups = M / 32;
    const unsigned grid = (unsigned)((groups + 3) / 4);
    head_dx_mfma_k<1, true><<<grid, 256, 0, st>>>((const float*)gout_t, (const float*)out_t, w, slope, (h16_t*)dx, groups, HW, C, K);
    return kp_check_launch("head_dx_mfma_t");
}

// (pixel splits / workspace rows: kp_head_dw_parts_mfma of the shape, the plan of the fp32-gradient form)
int kp_head_dw_mfma_t(const void* gout_t, const void* out_t, const void* x, float slope, float* dw, float* db, float* ws, long long M,
                      int HW, int C, int K, hipStream_t st, const KpBnIn* bn) {
    if (!kp_head_t_shape(M, HW, C, K) || !aligned16(gout_t, out_t, x)) return KP_E_UNSUPPORTED;
    const long long groups = M / 32;
    int gpw;
    const long long splits = dw_mfma_plan(M, HW, C, K, ws != nullptr, gpw);
    const unsigned grid = (unsigned)((splits + 3) / 4);
    const KpBnIn b = bn ? *bn : KpBnIn{};
    if (bn) head_dw_mfma32_t_k<true><<<grid, 256, 0, st>>>((const h16_t*)gout_t, (const h16_t*)out_t, (const h16_t*)x, slope, dw, db, ws, groups, gpw, b);
    else head_dw_mfma32_t_k<false><<<grid, 256, 0, st>>>((const h16_t*)gout_t, (const h16_t*)out_t, (const h16_t*)x, slope, dw, db, ws, groups, gpw, b);
    return kp_check_launch("head_dw_mfma_t");
}
