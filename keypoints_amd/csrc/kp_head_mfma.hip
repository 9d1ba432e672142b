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
// BNIN (C == 32 only: the Pong-class encoder heads): x is the producing layer's conv output, `bn` its BatchNorm / activation
// (KpBnIn, kp_common.h), applied to the pixel fragment of either 16-channel step; the constants of a lane's 2 x 8 channels
// live in registers.
template <int KT, bool BNIN>   // 32-channel tiles of the output (K <= 32 * KT)
__global__ __launch_bounds__(256) void head_fwd_mfma_k(const h16_t* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float slope,
                                                       float* __restrict__ out, long long groups, int HW, int C, int K, KpBnIn bn) {
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
    auto step = [&](int c0, const KpBnIn8* kk) {
        h16x8 xb = *reinterpret_cast<const h16x8*>(xp + c0);
        if (BNIN) {
            const u32x4 r = __builtin_bit_cast(u32x4, xb);
            const uint4 t = kk->apply(make_uint4(r[0], r[1], r[2], r[3]), bn.slope);
            xb = __builtin_bit_cast(h16x8, u32x4{t.x, t.y, t.z, t.w});
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            float wv[8];
            load_w8(w, kt * 32 + px, K, C, c0 + h * 8, wv);
            h16x8 whi, wlo;
            split8(wv, whi, wlo);
            acc[kt] = mma(whi, xb, acc[kt]);
            acc[kt] = mma(wlo, xb, acc[kt]);
        }
    };
    if constexpr (BNIN) {   // C == 32: two steps, either with its own constants in registers
        KpBnIn8 k0, k1;
        k0.load(bn, h * 8); k1.load(bn, 16 + h * 8);
        step(0, &k0); step(16, &k1);
    } else {
        for (int c0 = 0; c0 < C; c0 += 16) step(c0, nullptr);
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

// ---- input gradient ------------------------------------------------------------------------------------------------
template <int KS>   // 16-channel reduction slices (K <= 16 * KS)
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
        load_dz_px(gout, out, n, hw, HW, K, kk * 16 + h * 8, slope, v);
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
// BNIN (CT == 1, C == 32): a lane's input-channel column is fixed (r), its four constants are scalars of the lane
template <int KT, int CT, bool BNIN>
__global__ __launch_bounds__(256) void head_dw_mfma_k(const float* __restrict__ gout, const float* __restrict__ out,
                                                      const h16_t* __restrict__ x, float slope, float* __restrict__ dw,
                                                      float* __restrict__ db, float* __restrict__ ws, long long groups,
                                                      int gpw, int HW, int C, int K, KpBnIn bn) {
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
    float bmu = 0.f, bis = 1.f, bga = 1.f, bbe = -0.f;   // (the exact identity without BatchNorm)
    if (BNIN && bn.mean) { const int ch = cr * (32 * CT) + r; bmu = bn.mean[ch]; bis = bn.invstd[ch]; bga = bn.gamma[ch]; bbe = bn.beta[ch]; }

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
                for (int i = 0; i < 8; ++i) {
                    h16_t e = xp[(long long)i * C + t * 32];
                    if (BNIN) e = f2h(kp_bnin1(h2f(e), bmu, bis, bga, bbe, bn.slope));
                    xb[i] = (short)e;
                }
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
bool kp_head_mfma_bnin_shape(long long M, int HW, int C, int K) { return mfma_shape(M, HW, C, K) && C == 32 && K <= 32; }
bool kp_head_mfma_shape(long long M, int HW, int C, int K) { return mfma_shape(M, HW, C, K); }

int kp_head_fwd_mfma(const void* x, const float* w, const float* bias, float slope, float* out, long long M, int HW,
                     int C, int K, hipStream_t st, const KpBnIn* bn) {
    if (!mfma_shape(M, HW, C, K) || !aligned16(x, w)) return KP_E_UNSUPPORTED;
    if (bn && !kp_head_mfma_bnin_shape(M, HW, C, K)) return KP_E_UNSUPPORTED;
    const long long groups = M / 32;
    const unsigned grid = (unsigned)((groups + 3) / 4);
    if (bn) head_fwd_mfma_k<1, true><<<grid, 256, 0, st>>>((const h16_t*)x, w, bias, slope, out, groups, HW, C, K, *bn);
    else if (K <= 32) head_fwd_mfma_k<1, false><<<grid, 256, 0, st>>>((const h16_t*)x, w, bias, slope, out, groups, HW, C, K, KpBnIn{});
    else head_fwd_mfma_k<2, false><<<grid, 256, 0, st>>>((const h16_t*)x, w, bias, slope, out, groups, HW, C, K, KpBnIn{});
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
#define KP_DW(KTv, CTv) head_dw_mfma_k<KTv, CTv, false><<<grid, 256, 0, st>>>(gout, out, (const h16_t*)x, slope, dw, db, ws, groups, gpw, HW, C, K, KpBnIn{})
    if (bn) head_dw_mfma_k<1, 1, true><<<grid, 256, 0, st>>>(gout, out, (const h16_t*)x, slope, dw, db, ws, groups, gpw, HW, C, K, *bn);
    else if (K <= 32 && ct == 4) KP_DW(1, 4);
    else if (ct == 4) KP_DW(2, 4);
    else if (K <= 32) KP_DW(1, 1);
    else KP_DW(2, 1);
#undef KP_DW
    return kp_check_launch("head_dw_mfma");
}
