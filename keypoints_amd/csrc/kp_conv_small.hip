// 3x3 convolution for SMALL channel counts (C, K in {16, 32}: the Pong-class nets, vgg.py:53,66 VGG_PONG_*), bf16.
// These layers are HBM-bound (16.6 GFLOP against 174 MB per layer at 256 x 84 x 84), and their images (84 x 84) are
// not 16x16-tileable, so neither MFMA patch kernel applies and the general fp32 kernels sat ~30x off the roofline.
//
// forward / data-gradient (conv3x3_small_k): one wavefront per 32-pixel row segment, no LDS, no barriers.
//   With C = 16 one tap IS one MFMA reduction step (v_mfma_f32_32x32x16_bf16): the pixel operand of tap (r,s) is a
//   straight coalesced 16-byte-per-lane load of 32 consecutive padded-image pixels (1 KiB, L1/L2-hot after the first
//   tap), the 9 (18 for C = 32) weight fragments stay in registers for the whole kernel.  D = [32 k][32 px]; K = 16
//   uses half of it.  Epilogue: bias, running BN partial sums per lane (one partial row per wavefront at the end),
//   v_permlane32_swap pairs the two half-waves' 8-byte pieces into 16-byte stores.
// weight gradient (wgrad3x3_small_k): VALU, a thread owns (k, 8 input channels) and walks pixels: dy[px][k] scalar x
//   the 9 x 16-byte input vectors of the window (shared by all k-lanes: one L1 transaction), 72 accumulators;
//   block reduction over the pixel streams, fp32 atomics into the zeroed scratch.
#include "kp_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TPB = 256;

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// 16 per-lane values -> each lane ends with the sum over its 32-lane half of ONE value (index in `reg`)
__device__ __forceinline__ float reduce16_over32(float (&v)[16], int lane, int& reg) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float lo = v[i], hi = v[i + 8];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        a[i] = lo + hi;
    }
    const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
    float b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b3 ? a[i + 4] : a[i], send = b3 ? a[i] : a[i + 4];
        b[i] = keep + dpp_mov<0x128>(send);
    }
    float c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b2 ? b[i + 2] : b[i], send = b2 ? b[i] : b[i + 2];
        c[i] = keep + dpp_mov<0x141>(send);
    }
    float d = (b1 ? c[1] : c[0]) + dpp_mov<0x4E>(b1 ? c[0] : c[1]);
    d += dpp_mov<0xB1>(d);
    reg = (b4 ? 8 : 0) | (b3 ? 4 : 0) | (b2 ? 2 : 0) | (b1 ? 1 : 0);
    return d;
}

// CB = C / 16
template <int CB>
__global__ __launch_bounds__(TPB) void conv3x3_small_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                       const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                       float* __restrict__ partials, int N, int Hi, int Wi, int K,
                                                       long long nseg, int segs_per_row) {
    constexpr int C = CB * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int Ho = Hi - 2, Wo = Wi - 2;

    // weight fragments: lane (k = fr, half = fh): 8 input channels of W[k][tap][cb*16 + fh*8 ...]; rows k >= K are zero
    bf16x8 wf[9 * CB];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (fr < K) v = *reinterpret_cast<const bf16x8*>(w + ((long long)fr * 9 + t) * C + cb * 16 + fh * 8);
            wf[t * CB + cb] = v;
        }
    float bv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k = 8 * (i >> 2) + 4 * fh + (i & 3);
        bv[i] = (bias && k < K) ? bias[k] : 0.f;
    }
    float ssum[16], ssq[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) ssum[i] = ssq[i] = 0.f;

    const long long gw = (long long)blockIdx.x * 4 + wave, nw = (long long)gridDim.x * 4;
    for (long long seg = gw; seg < nseg; seg += nw) {
        const int sq = (int)(seg % segs_per_row);
        const long long rowid = seg / segs_per_row;
        const int p = (int)(rowid % Ho);
        const long long n = rowid / Ho;
        const int q = sq * 32 + fr;
        const bool ok = q < Wo;
        const int qc = ok ? q : Wo - 1;   // tail lanes load a valid pixel, masked at the store
        const bf16_t* xp = x + ((n * Hi + p) * Wi + qc) * (long long)C + fh * 8;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        bf16x8 xf[9 * CB];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                xf[t * CB + cb] = *reinterpret_cast<const bf16x8*>(xp + ((long long)(t / 3) * Wi + (t % 3)) * C + cb * 16);
#pragma unroll
        for (int t = 0; t < 9 * CB; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t], xf[t], acc, 0, 0, 0);
        // D[k][px]: lane: pixel = fr, k = 8*(reg>>2) + 4*fh + (reg&3)
        uint32_t pk[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = acc[g * 4 + e] + bv[g * 4 + e];
                if (ok) { ssum[g * 4 + e] += v[e]; ssq[g * 4 + e] += v[e] * v[e]; }
            }
            pk[2 * g] = pack_bf16x2(v[0], v[1]);
            pk[2 * g + 1] = pack_bf16x2(v[2], v[3]);
        }
        // pair the half-waves' 8-byte pieces: after the swaps lanes fh=0 hold channels 8g..8g+7 of chunk g (even g),
        // lanes fh=1 those of chunk g+1
        bf16_t* yp = y + ((n * Ho + p) * Wo + q) * (long long)K;
#pragma unroll
        for (int g = 0; g < 4; g += 2) {
            uint32_t a0 = pk[2 * g], a1 = pk[2 * g + 1], b0 = pk[2 * g + 2], b1 = pk[2 * g + 3];
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a0), "+v"(b0));
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a1), "+v"(b1));
            const int kc = 8 * (g + fh);
            if (ok && kc < K) *reinterpret_cast<uint4*>(yp + kc) = make_uint4(a0, a1, b0, b1);
        }
    }
    if (partials) {
        int r1, r2;
        const float ts = reduce16_over32(ssum, lane, r1);
        const float tq = reduce16_over32(ssq, lane, r2);
        const int k = 8 * (r1 >> 2) + 4 * fh + (r1 & 3);
        if ((lane & 1) == 0 && k < K) {
            partials[(gw * 2 + 0) * K + k] = ts;
            partials[(gw * 2 + 1) * K + k] = tq;
        }
    }
}

// scratch[k][tap][c] += sum_pixels dy[pixel][k] * x[pixel + tap][c];  thread = (k, group of 8 input channels)
template <int K, int C>
__global__ __launch_bounds__(TPB) void wgrad3x3_small_k(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                        float* __restrict__ scratch, int Ho, int Wo, int dy_pad,
                                                        long long total_rows, int rows_per_block) {
    constexpr int TPP = K * C / 8;          // threads per pixel stream
    constexpr int NS = TPB / TPP;           // pixel streams per block
    const int sid = threadIdx.x / TPP, r = threadIdx.x % TPP;
    const int k = r % K, cg = r / K;
    const int Hi = Ho + 2, Wi = Wo + 2, Hd = Ho + 2 * dy_pad, Wd = Wo + 2 * dy_pad;
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[t][e] = 0.f;
    const long long row0 = (long long)blockIdx.x * rows_per_block;
    for (int rr = sid; rr < rows_per_block; rr += NS) {
        const long long row = row0 + rr;
        if (row >= total_rows) break;
        const int p = (int)(row % Ho);
        const long long n = row / Ho;
        const bf16_t* dr = dy + (((n * Hd + p + dy_pad) * Wd) + dy_pad) * (long long)K + k;
        const bf16_t* xr = x + ((n * Hi + p) * Wi) * (long long)C + cg * 8;
        for (int q = 0; q < Wo; ++q) {
            const float d = bf2f(dr[(long long)q * K]);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const uint4 raw = *reinterpret_cast<const uint4*>(xr + ((long long)(t / 3) * Wi + q + (t % 3)) * C);
                const float xv[8] = {__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u),
                                     __uint_as_float(raw.y << 16), __uint_as_float(raw.y & 0xffff0000u),
                                     __uint_as_float(raw.z << 16), __uint_as_float(raw.z & 0xffff0000u),
                                     __uint_as_float(raw.w << 16), __uint_as_float(raw.w & 0xffff0000u)};
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[t][e] = fmaf(d, xv[e], acc[t][e]);
            }
        }
    }
    // reduce the NS pixel streams through LDS, then one atomic per (k, tap, c) per block
    __shared__ float red[TPB * 9];   // reused per channel e: [NS][TPP][9]
    for (int e = 0; e < 8; ++e) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t) red[(sid * TPP + r) * 9 + t] = acc[t][e];
        __syncthreads();
        if (sid == 0) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                float s = 0.f;
                for (int s2 = 0; s2 < NS; ++s2) s += red[(s2 * TPP + r) * 9 + t];
                unsafeAtomicAdd(scratch + ((long long)k * 9 + t) * C + cg * 8 + e, s);
            }
        }
    }
}

bool small_ch(int v) { return v == 16 || v == 32; }

}  // namespace

int kp_conv_small_fwd(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi,
                      int C, int K, hipStream_t st) {
    if (!small_ch(C) || !small_ch(K) || Hi < 3 || Wi < 3) return KP_E_UNSUPPORTED;
    const int Ho = Hi - 2, Wo = Wi - 2;
    const int spr = (Wo + 31) / 32;
    const long long nseg = (long long)N * Ho * spr;
    // stats rows = waves of the grid; the caller's buffer holds ceil(N*Ho*Wo/64) rows
    long long max_rows = ((long long)N * Ho * Wo + 63) / 64;
    long long blocks = (nseg + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (stats && blocks * 4 > max_rows) blocks = max_rows / 4;
    if (blocks < 1) { if (stats) return KP_E_UNSUPPORTED; blocks = 1; }
    if (stats) kp_set_last_rows(blocks * 4);
    if (C == 16)
        conv3x3_small_k<1><<<(unsigned)blocks, TPB, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, stats, N, Hi, Wi, K, nseg, spr);
    else
        conv3x3_small_k<2><<<(unsigned)blocks, TPB, 0, st>>>((const bf16_t*)x, (const bf16_t*)w, bias, (bf16_t*)y, stats, N, Hi, Wi, K, nseg, spr);
    return kp_check_launch("conv3x3_small");
}

int kp_conv_small_wgrad(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int dy_pad,
                        hipStream_t st) {
    if (!small_ch(C) || !small_ch(K)) return KP_E_UNSUPPORTED;
    const long long rows = (long long)N * Ho;
    const int ns = TPB / (K * C / 8);
    int rpb = (int)((rows + 2047) / 2048);
    if (rpb < ns) rpb = ns;
    rpb = (rpb + ns - 1) / ns * ns;
    const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
#define KP_SWG(Kv, Cv) wgrad3x3_small_k<Kv, Cv><<<grid, TPB, 0, st>>>((const bf16_t*)x, (const bf16_t*)dy, scratch, Ho, Wo, dy_pad, rows, rpb)
    if (K == 16 && C == 16) KP_SWG(16, 16);
    else if (K == 16) KP_SWG(16, 32);
    else if (C == 16) KP_SWG(32, 16);
    else KP_SWG(32, 32);
#undef KP_SWG
    return kp_check_launch("wgrad3x3_small");
}
