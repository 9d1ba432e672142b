// Strict-fp32 3x3 convolution on the matrix cores -- gfx950.
//
// The parity mode (set_precision('fp32'), the reference's apex O0 / plain fp32 run) keeps fp32 activations and weights
// and must accumulate in exact fp32.  v_mfma_f32_32x32x2_f32 does that (fp32 in, fp32 accumulate, an fmaf chain per
// output element) at the VALU's peak rate (64 FLOP/clk/SIMD, 157 TFLOP/s) but with the operand traffic of a matrix
// instruction: 2 VGPRs in per 4096 FLOP.  The generic VALU kernels (kp_conv_generic.hip) reach 0.23 of that peak; these
// take every 3x3 layer with C % 16 == 0 and K % 64 == 0 (all of model F but its 3-channel first layer).
//
//   forward / data-gradient ("valid" correlation over an image whose border is physically present, like the generic kernel):
//       workgroup = 4 wavefronts (2 pixel x 2 channel), tile = 128 consecutive output pixels (linear index over
//       N x Ho x Wo) x 128 (64) output channels; reduction steps = (tap, 16 input channels); both operands go
//       global -> registers -> LDS (double buffered, one barrier per step: the loads of step s+1 are issued before the
//       MFMAs of step s).  LDS rows are 64 B (16 channels of one pixel / one output channel) with the 16-byte slot XORed
//       by (row >> 2) & 3: every ds_read_b128 lane group hits 16 distinct bank quads.  A lane's ds_read_b128 carries 4
//       consecutive channels; MFMA i of a group takes element i from every lane, i.e. the reduction index is permuted
//       (lanes 0-31: channel i, lanes 32-63: channel 4 + i of the group) -- identically for both operands, so the sum is
//       the same set of products.  Operands are swapped (A = weights, B = pixels): a lane's accumulator registers hold 4
//       consecutive output channels of one pixel -> float4 stores; BatchNorm partial sums per 64-pixel row exactly as the
//       generic kernel defines them (kp_hip.h), reduced across lanes with VALU permutes (kp_mfma_util.h).
//   weight gradient: workgroup = one tap x 128 (64) output channels x 128 (64) input channels x one pixel split;
//       reduction steps = 16 pixels; A = dy^T and B = x are read from pixel-major LDS tiles with ds_read_b32 (a lane
//       reads ONE channel of pixel p or p + 1: 32 consecutive floats per half wave, conflict free); every split writes its
//       own partial gradient with plain stores (kp_wgrad_unpack sums them in a fixed order: deterministic).
#include <stdlib.h>
#include "kp_mfma_util.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int TPB = 256;
constexpr int BKC = 16;    // input channels per reduction step
constexpr int TM = 128;    // pixels per tile

__device__ __forceinline__ int swz(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

template <int BNT, bool STATS>
__global__ __launch_bounds__(TPB, 2) void conv3x3_f32_mfma_k(const float* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              float* __restrict__ partials, int Hi, int Wi, int C, int K,
                                                              int ktiles, long long M, long long rows) {
    constexpr int TN = BNT / 64;        // 32-channel MFMA tiles per wave
    constexpr int WN = BNT / 2;         // channels per wave
    constexpr int NB = BNT / 64;        // weight float4 loads per thread per step
    __shared__ __attribute__((aligned(16))) unsigned char As[2][TM * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[2][BNT * 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    const int Ho = Hi - 2, Wo = Wi - 2;
    const long long bid = blockIdx.x;
    const int kt = (int)(bid % ktiles);
    const long long m0 = (bid / ktiles) * TM;
    const int k0 = kt * BNT;

    // loader role: row = tid >> 2 (+ 64), 16-byte slot = tid & 3
    const int lrow = tid >> 2, lslot = tid & 3;
    const float* ap[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long long m = m0 + lrow + 64 * i;
        if (m > M - 1) m = M - 1;                       // tail tile: re-read the last pixel, its results are not stored
        const int q = (int)(m % Wo);
        const long long t = m / Wo;
        const int p = (int)(t % Ho);
        const long long n = t / Ho;
        ap[i] = x + ((n * Hi + p) * Wi + q) * (long long)C + lslot * 4;
    }
    const float* bp[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) bp[i] = w + (long long)(k0 + lrow + 64 * i) * 9 * C + lslot * 4;

    f32x16 acc[TN][2];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int cchunks = C / BKC, S = 9 * cchunks;
    f32x4 ra[2], rb[NB];
    auto gload = [&](int s) {
        const int tap = s / cchunks, c0 = (s - tap * cchunks) * BKC;
        const int r = tap / 3, sx = tap - 3 * r;
        const long long ao = ((long long)r * Wi + sx) * C + c0;
        const long long bo = (long long)tap * C + c0;
#pragma unroll
        for (int i = 0; i < 2; ++i) ra[i] = *reinterpret_cast<const f32x4*>(ap[i] + ao);
#pragma unroll
        for (int i = 0; i < NB; ++i) rb[i] = *reinterpret_cast<const f32x4*>(bp[i] + bo);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(As[buf] + swz(lrow + 64 * i, lslot)) = ra[i];
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<f32x4*>(Bs[buf] + swz(lrow + 64 * i, lslot)) = rb[i];
    };

    gload(0);
    lstore(0);
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        const int buf = s & 1;
        if (s + 1 < S) gload(s + 1);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            f32x4 xf[2], wf[TN];
#pragma unroll
            for (int b = 0; b < 2; ++b) xf[b] = *reinterpret_cast<const f32x4*>(As[buf] + swz(wm * 64 + b * 32 + fr, 2 * jj + fh));
#pragma unroll
            for (int a = 0; a < TN; ++a) wf[a] = *reinterpret_cast<const f32x4*>(Bs[buf] + swz(wn * WN + a * 32 + fr, 2 * jj + fh));
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[a][i], xf[b][i], acc[a][b], 0, 0, 0);
        }
        if (s + 1 < S) lstore(buf ^ 1);
        __syncthreads();
    }

    // epilogue: D[k_out][pixel]; lane: pixel = fr of fragment b, k_out = 8 * (reg >> 2) + 4 * fh + (reg & 3)
    float sm[STATS ? TN : 1][16], sq[STATS ? TN : 1][16];
    if constexpr (STATS) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) sm[a][i] = sq[a][i] = 0.f;
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const long long m = m0 + wm * 64 + b * 32 + fr;
        const bool live = m < M;
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            const int kb = k0 + wn * WN + a * 32 + 4 * fh;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v;
                const f32x4 bq = bias ? *reinterpret_cast<const f32x4*>(bias + kb + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[a][b][g * 4 + e] + bq[e];
                    if constexpr (STATS) {
                        const float u = live ? v[e] : 0.f;
                        sm[a][g * 4 + e] += u;
                        sq[a][g * 4 + e] += u * u;
                    }
                }
                if (live) *reinterpret_cast<f32x4*>(y + m * K + kb + 8 * g) = v;
            }
        }
    }
    if constexpr (STATS) {
        const long long prow = m0 / 64 + wm;   // one statistics row per 64 output pixels, like the generic kernel
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            int r1, r2;
            const float ts = reduce16_over32(sm[a], lane, r1);
            const float tq = reduce16_over32(sq[a], lane, r2);
            if ((lane & 1) == 0 && prow < rows) {
                const int k = k0 + wn * WN + a * 32 + 8 * (r1 >> 2) + 4 * fh + (r1 & 3);
                partials[(prow * 2 + 0) * K + k] = ts;
                partials[(prow * 2 + 1) * K + k] = tq;
            }
        }
    }
}

// scratch[split][k][tap][c] = sum over the split's pixels of dy[m][k] * x[pix(m) + tap][c]
constexpr int PB = 16;     // pixels per reduction step
// ROWT: Wo % 16 == 0 -- a 16-pixel step never leaves its image row, so a step's two row bases are scalars and a
// thread's offsets constants (the general path recomputes (n, p, q) per load: ~250 instructions per step)
template <int TK, int TC, bool ROWT>
__global__ __launch_bounds__(TPB, 2) void wgrad3x3_f32_mfma_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ scratch, int Ho, int Wo, int C, int K,
                                                               int dy_pad, int ctiles, int ktiles, int splits, long long M,
                                                               long long ppb) {
    constexpr int AK = TK / 64, AC = TC / 64;      // 32-wide MFMA tiles per wave along k / c
    __shared__ __attribute__((aligned(16))) float Ds[2][PB][TK];
    __shared__ __attribute__((aligned(16))) float Xs[2][PB][TC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wc = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    const int Hi = Ho + 2, Wi = Wo + 2, Hd = Ho + 2 * dy_pad, Wd = Wo + 2 * dy_pad;
    // XCD-aware order: the T = 9 * ctiles * ktiles workgroups of ONE pixel split run on one XCD (blockIdx % 8) at the same
    // time and walk the split's pixels in step, so each 16-pixel slice of dy / x is fetched into that XCD's L2 once and
    // re-read from there by the other taps / tiles (round-robin placement of a (tap, tile, split) grid had every XCD fetch
    // every slice: the kernel ran at the pace of the memory-side traffic)
    const int T = 9 * ctiles * ktiles;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int split = (jb / T) * 8 + xcd, tile = jb % T;
    if (split >= splits) return;
    const int kt = tile / (9 * ctiles), tx = tile - kt * 9 * ctiles;
    const int tap = tx / ctiles, c0 = (tx - tap * ctiles) * TC;
    const int k0 = kt * TK;
    const int r = tap / 3, sx = tap - 3 * r;
    const long long mbeg = (long long)split * ppb;
    const long long mend = mbeg + ppb < M ? mbeg + ppb : M;

    // loader role: float4 id = tid + 256 * i over a [PB][TK or TC] tile.  The pixel coordinates of a step's first pixel
    // (n, p, q) advance by carries (uniform: scalar registers); a thread adds its row of the tile with one more carry chain
    // -- no 64-bit divisions in the loop (they cost more than the step's 32 MFMAs)
    constexpr int LD = PB * TK / 4 / TPB, LX = PB * TC / 4 / TPB;
    f32x4 rd[LD], rx[LX];
    int bq, bp; long long bn;
    {
        bq = (int)(mbeg % Wo);
        const long long t = mbeg / Wo;
        bp = (int)(t % Ho);
        bn = t / Ho;
    }
    auto pixel = [&](int pr, int& q, int& p, long long& n) {
        q = bq + pr; p = bp; n = bn;
        while (q >= Wo) { q -= Wo; if (++p == Ho) { p = 0; ++n; } }
    };
    int doff[LD], xoff[LX];
#pragma unroll
    for (int i = 0; i < LD; ++i) { const int id = tid + TPB * i; doff[i] = (id / (TK / 4)) * K + (id % (TK / 4)) * 4; }
#pragma unroll
    for (int i = 0; i < LX; ++i) { const int id = tid + TPB * i; xoff[i] = (id / (TC / 4)) * C + (id % (TC / 4)) * 4; }
    auto gload = [&](long long mb) {
        if constexpr (ROWT) {
            const float* dyrow = dy + ((bn * Hd + bp + dy_pad) * Wd + bq + dy_pad) * (long long)K + k0;
            const float* xrow = x + ((bn * Hi + bp + r) * Wi + bq + sx) * (long long)C + c0;
#pragma unroll
            for (int i = 0; i < LD; ++i) rd[i] = *reinterpret_cast<const f32x4*>(dyrow + doff[i]);
#pragma unroll
            for (int i = 0; i < LX; ++i) rx[i] = *reinterpret_cast<const f32x4*>(xrow + xoff[i]);
            bq += PB;
            if (bq >= Wo) { bq = 0; if (++bp == Ho) { bp = 0; ++bn; } }
            return;
        }
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            const int id = tid + TPB * i, pr = id / (TK / 4), col = (id % (TK / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (mb + pr < mend) {
                int q, p; long long n;
                pixel(pr, q, p, n);
                v = *reinterpret_cast<const f32x4*>(dy + ((n * Hd + p + dy_pad) * Wd + q + dy_pad) * (long long)K + k0 + col);
            }
            rd[i] = v;
        }
#pragma unroll
        for (int i = 0; i < LX; ++i) {
            const int id = tid + TPB * i, pr = id / (TC / 4), col = (id % (TC / 4)) * 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (mb + pr < mend) {
                int q, p; long long n;
                pixel(pr, q, p, n);
                v = *reinterpret_cast<const f32x4*>(x + ((n * Hi + p + r) * Wi + q + sx) * (long long)C + c0 + col);
            }
            rx[i] = v;
        }
        // advance the base to the next step
        bq += PB;
        while (bq >= Wo) { bq -= Wo; if (++bp == Ho) { bp = 0; ++bn; } }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LD; ++i) {
            const int id = tid + TPB * i, pr = id / (TK / 4), col = (id % (TK / 4)) * 4;
            *reinterpret_cast<f32x4*>(&Ds[buf][pr][col]) = rd[i];
        }
#pragma unroll
        for (int i = 0; i < LX; ++i) {
            const int id = tid + TPB * i, pr = id / (TC / 4), col = (id % (TC / 4)) * 4;
            *reinterpret_cast<f32x4*>(&Xs[buf][pr][col]) = rx[i];
        }
    };

    f32x16 acc[AK][AC];
#pragma unroll
    for (int a = 0; a < AK; ++a)
#pragma unroll
        for (int b = 0; b < AC; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    gload(mbeg);
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (long long mb = mbeg; mb < mend; mb += PB) {
        const bool more = mb + PB < mend;
        if (more) gload(mb + PB);
        float df[2][AK], xf[2][AC];
#pragma unroll
        for (int a = 0; a < AK; ++a) df[0][a] = Ds[buf][fh][wk * (TK / 2) + a * 32 + fr];
#pragma unroll
        for (int b = 0; b < AC; ++b) xf[0][b] = Xs[buf][fh][wc * (TC / 2) + b * 32 + fr];
#pragma unroll
        for (int pp = 0; pp < PB; pp += 2) {
            const int cur = (pp >> 1) & 1, nxt = cur ^ 1;
            if (pp + 2 < PB) {
#pragma unroll
                for (int a = 0; a < AK; ++a) df[nxt][a] = Ds[buf][pp + 2 + fh][wk * (TK / 2) + a * 32 + fr];
#pragma unroll
                for (int b = 0; b < AC; ++b) xf[nxt][b] = Xs[buf][pp + 2 + fh][wc * (TC / 2) + b * 32 + fr];
            }
#pragma unroll
            for (int a = 0; a < AK; ++a)
#pragma unroll
                for (int b = 0; b < AC; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(df[cur][a], xf[cur][b], acc[a][b], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // D[k][c]: lane: c = fr, k = 8 * (reg >> 2) + 4 * fh + (reg & 3); plain stores, every element has one writer
    float* out = scratch + (long long)split * K * 9 * C;
#pragma unroll
    for (int a = 0; a < AK; ++a)
#pragma unroll
        for (int b = 0; b < AC; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = k0 + wk * (TK / 2) + a * 32 + 8 * (i >> 2) + 4 * fh + (i & 3);
                const int c = c0 + wc * (TC / 2) + b * 32 + fr;
                out[((long long)k * 9 + tap) * C + c] = acc[a][b][i];
            }
}

// Three taps of one filter row per workgroup (Wo % 16 == 0, K % 128 == 0, C % 64 == 0): the 16-pixel dy slice is staged
// once for the taps s = 0, 1, 2, whose x operands are the same 18-pixel slice of the padded input row read at a shift --
// 12.6 KB of L2 -> LDS traffic per 48 MFMAs per wave instead of 16 KB per 32.  Workgroup = filter row r x 128 output
// channels x 64 input channels x one pixel split; wave tile 64 k x 32 c x 3 taps (6 accumulators).
__global__ __launch_bounds__(TPB, 2) void wgrad3x3_f32_row3_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                               float* __restrict__ scratch, int Ho, int Wo, int C, int K,
                                                               int dy_pad, int ctiles, int ktiles, int splits, long long M,
                                                               long long ppb) {
    constexpr int TK = 128, TC = 64;
    __shared__ __attribute__((aligned(16))) float Ds[2][PB][TK];
    __shared__ __attribute__((aligned(16))) float Xs[2][PB + 2][TC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wc = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    const int Hi = Ho + 2, Wi = Wo + 2, Hd = Ho + 2 * dy_pad, Wd = Wo + 2 * dy_pad;
    const int T = 3 * ctiles * ktiles;
    const int xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int split = (jb / T) * 8 + xcd, tile = jb % T;
    if (split >= splits) return;
    const int kt = tile / (3 * ctiles), tx = tile - kt * 3 * ctiles;
    const int r = tx / ctiles, c0 = (tx - r * ctiles) * TC;
    const int k0 = kt * TK;
    const long long mbeg = (long long)split * ppb;
    const long long mend = mbeg + ppb < M ? mbeg + ppb : M;

    int bq = (int)(mbeg % Wo), bp; long long bn;
    { const long long t = mbeg / Wo; bp = (int)(t % Ho); bn = t / Ho; }
    // loaders: dy 16 x 128 floats = 512 float4 (2 per thread), x 18 x 64 floats = 288 float4 (1 per thread + 1 for 32 threads)
    const int doff0 = (tid >> 5) * K + (tid & 31) * 4, doff1 = doff0 + 8 * K;
    const int xr0 = tid >> 4, xc = (tid & 15) * 4;
    const int xoff0 = xr0 * C + xc, xoff1 = (xr0 + 16) * C + xc;
    const bool x2 = tid < 32;
    f32x4 rd[2], rx[2];
    auto gload = [&]() {
        const float* dyrow = dy + ((bn * Hd + bp + dy_pad) * Wd + bq + dy_pad) * (long long)K + k0;
        const float* xrow = x + ((bn * Hi + bp + r) * Wi + bq) * (long long)C + c0;
        rd[0] = *reinterpret_cast<const f32x4*>(dyrow + doff0);
        rd[1] = *reinterpret_cast<const f32x4*>(dyrow + doff1);
        rx[0] = *reinterpret_cast<const f32x4*>(xrow + xoff0);
        if (x2) rx[1] = *reinterpret_cast<const f32x4*>(xrow + xoff1);
        bq += PB;
        if (bq >= Wo) { bq = 0; if (++bp == Ho) { bp = 0; ++bn; } }
    };
    auto lstore = [&](int buf) {
        *reinterpret_cast<f32x4*>(&Ds[buf][tid >> 5][(tid & 31) * 4]) = rd[0];
        *reinterpret_cast<f32x4*>(&Ds[buf][(tid >> 5) + 8][(tid & 31) * 4]) = rd[1];
        *reinterpret_cast<f32x4*>(&Xs[buf][xr0][xc]) = rx[0];
        if (x2) *reinterpret_cast<f32x4*>(&Xs[buf][xr0 + 16][xc]) = rx[1];
    };
    f32x16 acc[2][3];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][t][i] = 0.f;
    gload();
    lstore(0);
    __syncthreads();
    int buf = 0;
    for (long long mb = mbeg; mb < mend; mb += PB) {
        const bool more = mb + PB < mend;
        if (more) gload();
        // fragments of pixel pair pp + 2 are read while the MFMAs of pair pp issue (two register sets)
        float df[2][2], xf[2][3];
#pragma unroll
        for (int a = 0; a < 2; ++a) df[0][a] = Ds[buf][fh][wk * 64 + a * 32 + fr];
#pragma unroll
        for (int t = 0; t < 3; ++t) xf[0][t] = Xs[buf][fh + t][wc * 32 + fr];
#pragma unroll
        for (int pp = 0; pp < PB; pp += 2) {
            const int cur = (pp >> 1) & 1, nxt = cur ^ 1;
            if (pp + 2 < PB) {
#pragma unroll
                for (int a = 0; a < 2; ++a) df[nxt][a] = Ds[buf][pp + 2 + fh][wk * 64 + a * 32 + fr];
#pragma unroll
                for (int t = 0; t < 3; ++t) xf[nxt][t] = Xs[buf][pp + 2 + fh + t][wc * 32 + fr];
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    acc[a][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(df[cur][a], xf[cur][t], acc[a][t], 0, 0, 0);
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* out = scratch + (long long)split * K * 9 * C;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = k0 + wk * 64 + a * 32 + 8 * (i >> 2) + 4 * fh + (i & 3);
                const int c = c0 + wc * 32 + fr;
                out[((long long)k * 9 + r * 3 + t) * C + c] = acc[a][t][i];
            }
}

bool fwd_ok(int N, int Hi, int Wi, int C, int K) {
    return C % BKC == 0 && K % 64 == 0 && Hi >= 3 && Wi >= 3 && (long long)N * Hi * Wi * C < (1LL << 40);
}

// pixel splits of the weight gradient: ONE resident round of workgroups (4 per CU: 117 VGPRs, 32 KB of LDS) -- the work per
// workgroup is uniform, so tiles x splits just above the resident count would leave a second, nearly empty round --
// and at least 512 pixels per split
bool wgrad_row3(int Wo, int C, int K) { return Wo % PB == 0 && K % 128 == 0 && C % 64 == 0; }

long long wgrad_plan(int N, int Ho, int Wo, int C, int K, long long& ppb, int& tk, int& tc) {
    const long long M = (long long)N * Ho * Wo;
    tk = K % 128 == 0 ? 128 : 64;
    tc = C % 128 == 0 ? 128 : 64;
    long long tiles = 9LL * (K / tk) * (C / tc);
    if (wgrad_row3(Wo, C, K)) { tc = 64; tiles = 3LL * (K / 128) * (C / 64); }
    // resident workgroups of the kernel this shape takes (occupancy query, once per kernel)
    static int slots_row3 = 0, slots_gen = 0;
    int& slots = wgrad_row3(Wo, C, K) ? slots_row3 : slots_gen;
    if (!slots) {
        int per_cu = 0;
        const hipError_t e = wgrad_row3(Wo, C, K)
            ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad3x3_f32_row3_k, TPB, 0)
            : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad3x3_f32_mfma_k<128, 128, true>, TPB, 0);
        int cus = kp_device_cus();
        if (cus <= 0) cus = 256;
        slots = (e == hipSuccess && per_cu > 0 ? per_cu : 3) * cus;
    }
    long long splits = slots / tiles;
    if (splits < 1) splits = 1;
    ppb = (M + splits - 1) / splits;
    if (ppb < 512) ppb = 512;
    ppb = (ppb + PB - 1) / PB * PB;
    return (M + ppb - 1) / ppb;
}

}  // namespace

// Returns KP_E_UNSUPPORTED when the shape is not eligible (the caller falls back to the generic kernel).
int kp_conv_fwd_f32mfma(const float* x, const float* w, const float* bias, float* y, float* stats, int N, int Hi, int Wi,
                        int C, int K, hipStream_t st) {
    if (!fwd_ok(N, Hi, Wi, C, K)) return KP_E_UNSUPPORTED;
    const long long M = (long long)N * (Hi - 2) * (Wi - 2);
    const long long rows = (M + 63) / 64;
    const int bnt = K % 128 == 0 ? 128 : 64;
    const int ktiles = K / bnt;
    const long long blocks = (M + TM - 1) / TM * ktiles;
    if (blocks >= (1LL << 31)) return KP_E_UNSUPPORTED;
    KP_QUERY_RETURN();
    const unsigned grid = (unsigned)blocks;
    if (bnt == 128 && stats) conv3x3_f32_mfma_k<128, true><<<grid, TPB, 0, st>>>(x, w, bias, y, stats, Hi, Wi, C, K, ktiles, M, rows);
    else if (bnt == 128) conv3x3_f32_mfma_k<128, false><<<grid, TPB, 0, st>>>(x, w, bias, y, stats, Hi, Wi, C, K, ktiles, M, rows);
    else if (stats) conv3x3_f32_mfma_k<64, true><<<grid, TPB, 0, st>>>(x, w, bias, y, stats, Hi, Wi, C, K, ktiles, M, rows);
    else conv3x3_f32_mfma_k<64, false><<<grid, TPB, 0, st>>>(x, w, bias, y, stats, Hi, Wi, C, K, ktiles, M, rows);
    return kp_check_launch("conv3x3_f32_mfma");
}

// number of partial gradient buffers of the fp32 matrix-core weight gradient (0: shape not eligible)
long long kp_conv_wgrad_f32mfma_partials(int N, int Ho, int Wo, int C, int K) {
    if (C % 64 != 0 || K % 64 != 0 || Ho < 1 || Wo < 1) return 0;
    long long ppb; int tk, tc;
    return wgrad_plan(N, Ho, Wo, C, K, ppb, tk, tc);
}

int kp_conv_wgrad_f32mfma(const float* x, const float* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int dy_pad,
                          hipStream_t st) {
    if (C % 64 != 0 || K % 64 != 0 || Ho < 1 || Wo < 1) return KP_E_UNSUPPORTED;
    long long ppb; int tk, tc;
    const long long splits = wgrad_plan(N, Ho, Wo, C, K, ppb, tk, tc);
    const long long M = (long long)N * Ho * Wo;
    const int ctiles = C / tc, ktiles = K / tk;
    if (wgrad_row3(Wo, C, K)) {
        const long long blocks3 = (splits + 7) / 8 * 8 * 3LL * ctiles * ktiles;
        if (blocks3 >= (1LL << 31)) return KP_E_UNSUPPORTED;
        wgrad3x3_f32_row3_k<<<(unsigned)blocks3, TPB, 0, st>>>(x, dy, scratch, Ho, Wo, C, K, dy_pad, ctiles, ktiles, (int)splits, M, ppb);
        return kp_check_launch("wgrad3x3_f32_row3");
    }
    const long long blocks = (splits + 7) / 8 * 8 * 9LL * ctiles * ktiles;
    if (blocks >= (1LL << 31)) return KP_E_UNSUPPORTED;
    const unsigned grid = (unsigned)blocks;
    const bool rowt = Wo % PB == 0;   // (then M and every split are multiples of 16 pixels as well)
#define KP_WG32(TKv, TCv)                                                                                                  \
    do {                                                                                                                   \
        if (rowt) wgrad3x3_f32_mfma_k<TKv, TCv, true><<<grid, TPB, 0, st>>>(x, dy, scratch, Ho, Wo, C, K, dy_pad, ctiles, ktiles, (int)splits, M, ppb);  \
        else wgrad3x3_f32_mfma_k<TKv, TCv, false><<<grid, TPB, 0, st>>>(x, dy, scratch, Ho, Wo, C, K, dy_pad, ctiles, ktiles, (int)splits, M, ppb);     \
    } while (0)
    if (tk == 128 && tc == 128) KP_WG32(128, 128);
    else if (tk == 128) KP_WG32(128, 64);
    else if (tc == 128) KP_WG32(64, 128);
    else KP_WG32(64, 64);
#undef KP_WG32
    return kp_check_launch("wgrad3x3_f32_mfma");
}
