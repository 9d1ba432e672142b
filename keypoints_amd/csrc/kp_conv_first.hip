// First layer of a Unit: Conv2d(Cin <= 4 -> K, 3x3) over the replicate-padded input image
// (knn.py:115: in_block = ReplicationPad2d(1), Conv2d(in_channels, core_in, 3), BN, LeakyReLU).
// With Cin = 3 the reduction is only 27 deep: 0.03 GMAC per image against 128 B of output per pixel, so both
// the forward and the weight-gradient are HBM-bound (write / read of the [N][H][W][K] bf16 tensor) and are
// written as VALU kernels shaped for coalesced 16-byte access rather than as MFMA GEMMs.
//
// forward : one thread per output pixel, 8 output channels at a time; weights are fp32 in LDS and read at
//           wave-uniform addresses (LDS broadcast); results go through a swizzled per-wave LDS tile so that
//           the global stores are whole 1-KiB runs; BatchNorm partial sums per 64 pixels by a halving butterfly.
// wgrad   : one lane per output channel k; a wavefront walks the pixels of an image row, reading dy[pixel][k]
//           coalesced and the 3x3xCin input window (fp32 in LDS, 16-byte aligned pixels) at uniform addresses;
//           27 accumulators per lane, block reduction in LDS, fp32 atomics into the zeroed scratch.
#include "kp_mfma_util.h"

namespace {
// The wave-private input strip is written as dwords and read as 16-bit elements: both through may_alias types, so type-based
// alias analysis cannot move the reads across the writes (kp_conv_first_fused.hip met that failure).
typedef uint32_t cf_u32_a __attribute__((may_alias));
typedef short cf_s16_a __attribute__((may_alias));
typedef unsigned short cf_u16_a __attribute__((may_alias));

constexpr int TPB = 256;

// 8 per-lane values -> each lane gets the sum over all 64 lanes of ONE of them (index returned in idx).
// Halving butterfly on VALU cross-lane ops only (v_permlane32_swap, v_permlane16_swap, DPP row_ror / half-mirror /
// quad_perm): the ds_bpermute version cost ~100 dependent LDS round trips per wavefront, more than the FMAs.
__device__ __forceinline__ float reduce8_over64(const float (&v)[8], int lane, int& idx) {
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    float a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // halves of 32 lanes: upper half of v[i] <-> lower half of v[i+4]
        float lo = v[i], hi = v[i + 4];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        a[i] = lo + hi;
    }
    float b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // rows of 16 lanes: odd rows of a[i] <-> even rows of a[i+2]
        float lo = a[i], hi = a[i + 2];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        b[i] = lo + hi;
    }
    float d = (b3 ? b[1] : b[0]) + dpp_mov<0x128>(b3 ? b[0] : b[1]);   // partner = lane ^ 8 (row_ror:8)
    d += dpp_mov<0x141>(d);    // + lane ^ 7 (row_half_mirror)
    d += dpp_mov<0x4E>(d);     // + lane ^ 2
    d += dpp_mov<0xB1>(d);     // + lane ^ 1
    idx = (b5 ? 4 : 0) | (b4 ? 2 : 0) | (b3 ? 1 : 0);
    return d;
}

template <int CIN>
__global__ __launch_bounds__(TPB) void conv3_first_fwd_k(const h16_t* __restrict__ x, const h16_t* __restrict__ w,
                                                         const float* __restrict__ bias, h16_t* __restrict__ y,
                                                         float* __restrict__ partials, int Hi, int Wi, int K,
                                                         long long M, int hb = 0, long long xh = 0, long long yh = 0) {
    // hb != 0 (kp_conv_first_fwd_pair): the grid is two launches of hb workgroups, one per half-batch of M pixels -- its own input
    // (+ xh elements), output (+ yh) and statistics rows (+ hb rows)
    int bid = blockIdx.x, nblk = gridDim.x;
    if (hb) {
        const int hf = bid >= hb ? 1 : 0;
        bid -= hf * hb; nblk = hb;
        x += hf * xh; y += hf * yh;
        if (partials) partials += (long long)hf * hb * 2 * K;
    }
    constexpr int J = 9 * CIN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* wl = reinterpret_cast<float*>(smem);                           // [J][K] fp32
    unsigned char* stage = smem + (size_t)J * K * sizeof(float);         // [4 waves][64 px][K] bf16, swizzled
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Ho = Hi - 2, Wo = Wi - 2;
    for (int i = tid; i < J * K; i += TPB) {   // w is [K][J] -> wl[j][k]
        const int k = i / J, j = i - k * J;
        wl[j * K + k] = h2f(w[i]);
    }
    __syncthreads();
    float* red = reinterpret_cast<float*>(stage + (size_t)4 * 64 * K * 2);   // [4 waves][2][K]: the workgroup's statistics
    if (partials)
        for (int i = lane; i < 2 * K; i += 64) red[wave * 2 * K + i] = 0.f;
    // a wavefront walks 64-pixel groups blockIdx * 4 + wave, + 4 * grid, ... (M % 64 == 0: groups are all-or-nothing); the
    // weights are staged once per workgroup, and the statistics row of a workgroup covers all its groups (<= 2 048 rows
    // for the finalize instead of M / 256: 7 056 at 256 x 84 x 84, 15 -> 8 us four times per step)
    for (long long m_wave = (long long)bid * TPB + wave * 64; m_wave < M; m_wave += (long long)nblk * TPB) {
    const long long m = m_wave + lane;
    const int q = (int)(m % Wo);
    const long long t = m / Wo;
    const int p = (int)(t % Ho);
    const long long n = t / Ho;
    float xin[J];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const h16_t* xr = x + ((n * Hi + p + r) * Wi + q) * CIN;
#pragma unroll
        for (int i = 0; i < 3 * CIN; ++i) xin[r * 3 * CIN + i] = h2f(xr[i]);
    }
    const int CP = K >> 3;                     // 16-byte chunks per pixel
    unsigned char* st = stage + (size_t)wave * 64 * K * 2;
    for (int k0 = 0; k0 < K; k0 += 8) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[k0 + e] : 0.f;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const float4 w0 = *reinterpret_cast<const float4*>(wl + j * K + k0);
            const float4 w1 = *reinterpret_cast<const float4*>(wl + j * K + k0 + 4);
            acc[0] = fmaf(xin[j], w0.x, acc[0]); acc[1] = fmaf(xin[j], w0.y, acc[1]);
            acc[2] = fmaf(xin[j], w0.z, acc[2]); acc[3] = fmaf(xin[j], w0.w, acc[3]);
            acc[4] = fmaf(xin[j], w1.x, acc[4]); acc[5] = fmaf(xin[j], w1.y, acc[5]);
            acc[6] = fmaf(xin[j], w1.z, acc[6]); acc[7] = fmaf(xin[j], w1.w, acc[7]);
        }
        if (partials) {
            float sq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) sq[e] = acc[e] * acc[e];
            int i1, i2;
            const float ts = reduce8_over64(acc, lane, i1);
            const float tq = reduce8_over64(sq, lane, i2);
            if ((lane & 7) == 0) {
                red[(wave * 2 + 0) * K + k0 + i1] += ts;
                red[(wave * 2 + 1) * K + k0 + i1] += tq;
            }
        }
        uint4 pk;
        pk.x = pack_h16x2(acc[0], acc[1]);
        pk.y = pack_h16x2(acc[2], acc[3]);
        pk.z = pack_h16x2(acc[4], acc[5]);
        pk.w = pack_h16x2(acc[6], acc[7]);
        const int ch = k0 >> 3;
        *reinterpret_cast<uint4*>(st + (size_t)lane * K * 2 + ((ch ^ (lane & (CP - 1))) << 4)) = pk;
    }
    // coalesced write-out: chunk id = i*64 + lane -> (pixel id / CP, chunk id % CP)
    h16_t* yw = y + m_wave * K;
    for (int id = lane; id < 64 * CP; id += 64) {
        const int px = id / CP, ch = id - px * CP;
        const uint4 v = *reinterpret_cast<const uint4*>(st + (size_t)px * K * 2 + ((ch ^ (px & (CP - 1))) << 4));
        *reinterpret_cast<uint4*>(yw + (size_t)id * 8) = v;
    }
    }   // groups
    // ONE statistics row per workgroup, its four wavefronts added in a fixed order
    if (partials) {
        __syncthreads();
        for (int i = tid; i < 2 * K; i += TPB)
            partials[(long long)bid * 2 * K + i] = ((red[i] + red[2 * K + i]) + red[4 * K + i]) + red[6 * K + i];
    }
}

// Single-channel inputs with 16 / 32 filters (the Pong-class first layers).  conv3_first_fwd_k reads every weight from LDS
// for every pixel (36 x ds_read_b128 per 64 pixels at K = 16: ~290 LDS-pipe cycles, the kernel's bound: 121 MB in 47 us).
// Here a lane owns ONE pixel x EIGHT channels: its 72 weights stay in registers, a wave-iteration covers 64 / KB consecutive
// pixels of an output row, and the 16 bytes a lane produces go out as one store (lanes of a wave: one contiguous KiB; no
// staging tile).  A wavefront walks whole output rows: the three input rows arrive as dwords (Wi even), are laid down in the
// wave's LDS strip as they lie in memory, and a lane picks its nine 16-bit values from there.  The fp32 accumulation order per
// output element is conv3_first_fwd_k's (bias, then the nine taps in order): the stored tensor is bit-identical.
constexpr int FR_RS = 512;        // bytes per staged input row (Wi <= 256)

template <int KB, bool WIDE>   // WIDE: input rows of more than 128 pixels (two dwords per lane and row)
__global__ __launch_bounds__(256, WIDE ? 3 : 4) void conv3_first_fwd_rows_k(const h16_t* __restrict__ x, const h16_t* __restrict__ w,
                                                              const float* __restrict__ bias, h16_t* __restrict__ y,
                                                              float* __restrict__ partials, int Hi, int Wi, long long rows,
                                                              int hb, long long xh, long long yh) {
    constexpr int K = KB * 8, PPI = 64 / KB;
    int bid = blockIdx.x, nblk = gridDim.x;
    if (hb) {   // kp_conv_fwd_pair: two launches of hb workgroups in one grid (see conv3_first_fwd_k)
        const int hf = bid >= hb ? 1 : 0;
        bid -= hf * hb; nblk = hb;
        x += hf * xh; y += hf * yh;
        if (partials) partials += (long long)hf * hb * 2 * K;
    }
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 3 * FR_RS + 4 * 2 * K * sizeof(float)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* xs = smem + wave * 3 * FR_RS;
    float* red = reinterpret_cast<float*>(smem + 4 * 3 * FR_RS);
    const int kb = lane % KB, p = lane / KB;
    const int Ho = Hi - 2, Wo = Wi - 2, ND = Wi >> 1;

    float wr[9][8], bs[8], sm[8], sq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bs[e] = bias ? bias[kb * 8 + e] : 0.f;
        sm[e] = sq[e] = 0.f;
#pragma unroll
        for (int j = 0; j < 9; ++j) wr[j][e] = h2f(w[(kb * 8 + e) * 9 + j]);
    }
    const long long nw = (long long)nblk * 4;
    long long row = (long long)bid * 4 + wave;
    uint32_t xr[3][WIDE ? 2 : 1] = {};
    auto fetch = [&](long long rw) {
        const long long n = rw / Ho;
        const int pr = (int)(rw - n * Ho);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(x + (n * Hi + pr) * Wi);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (lane < ND) xr[r][0] = src[r * ND + lane];
            if (WIDE && lane + 64 < ND) xr[r][WIDE ? 1 : 0] = src[r * ND + lane + 64];
        }
    };
    if (row < rows) fetch(row);
    for (; row < rows; row += nw) {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            if (lane < ND) *reinterpret_cast<cf_u32_a*>(xs + r * FR_RS + lane * 4) = xr[r][0];
            if (WIDE && lane + 64 < ND) *reinterpret_cast<cf_u32_a*>(xs + r * FR_RS + (lane + 64) * 4) = xr[r][WIDE ? 1 : 0];
        }
        if (row + nw < rows) fetch(row + nw);               // in flight during this row's arithmetic
        h16_t* yr = y + row * Wo * K + kb * 8;
        for (int q0 = 0; q0 < Wo; q0 += PPI) {
            const int q = q0 + p;
            const bool valid = q < Wo;
            const int qq = valid ? q : Wo - 1;
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = bs[e];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float xv = h2f((h16_t)*reinterpret_cast<const cf_u16_a*>(xs + r * FR_RS + (qq + t) * 2));
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = fmaf(xv, wr[r * 3 + t][e], acc[e]);
                }
            if (valid) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    sm[e] += acc[e];
                    sq[e] = fmaf(acc[e], acc[e], sq[e]);
                }
                uint4 pk;
                pk.x = pack_h16x2(acc[0], acc[1]);
                pk.y = pack_h16x2(acc[2], acc[3]);
                pk.z = pack_h16x2(acc[4], acc[5]);
                pk.w = pack_h16x2(acc[6], acc[7]);
                *reinterpret_cast<uint4*>(yr + (long long)q * K) = pk;
            }
        }
    }
    // ONE statistics row per workgroup: the lanes of a channel block, then the four wavefronts, in a fixed order
    if (partials) {
#pragma unroll
        for (int o = KB; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                sm[e] += __shfl_xor(sm[e], o, 64);
                sq[e] += __shfl_xor(sq[e], o, 64);
            }
        if (p == 0)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                red[(wave * 2 + 0) * K + kb * 8 + e] = sm[e];
                red[(wave * 2 + 1) * K + kb * 8 + e] = sq[e];
            }
        __syncthreads();
        for (int i = tid; i < 2 * K; i += 256)
            partials[(long long)bid * 2 * K + i] = ((red[i] + red[2 * K + i]) + red[4 * K + i]) + red[6 * K + i];
    }
}

// scratch[k][tap][c] += sum_pixels dy[pixel][k] * x[pixel + tap][c]
// NW wavefronts per workgroup.  The kernel is a latency chain per image row (stage three input rows, then walk the gradient
// row): with <= 512 workgroups (= partial gradients) of 4 waves only ~8 waves were resident per CU and the K = 16 / 32 launches
// of the Pong-class nets took 68 us for a 58 MB tensor (ablations: 15.7 us without the walk, 29 without the gradient loads, 41
// without the input loads).  Sixteen waves per workgroup keep the partial count and put 4x the rows in flight.
template <int CIN, int NW>
__global__ __launch_bounds__(NW * 64) void conv3_first_wgrad_k(const h16_t* __restrict__ x, const h16_t* __restrict__ dy,
                                                           float* __restrict__ scratch, int Ho, int Wo, int K,
                                                           int dy_pad, long long total_rows, int rows_per_block) {
    // every workgroup leaves ITS partial gradient in scratch[blockIdx.x][K][J] (plain stores; kp_wgrad_unpack adds them in order)
    constexpr int J = 9 * CIN;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Hi = Ho + 2, Wi = Wo + 2, Hd = Ho + 2 * dy_pad, Wd = Wo + 2 * dy_pad;
    const int KS = 64 / K;                                    // pixel streams per wavefront (K = 16, 32, 64)
    const int k = lane % K, stream = lane / K;
    const int PP = (CIN == 1 && KS > 1) ? 1 : 4;              // floats per staged pixel
    const int rowf = Wi * PP;                                 // floats per staged input row
    float* xw = reinterpret_cast<float*>(smem) + (size_t)wave * 3 * rowf;   // this wave's 3 input rows

    float acc[J];
#pragma unroll
    for (int j = 0; j < J; ++j) acc[j] = 0.f;

    const long long row0 = (long long)blockIdx.x * rows_per_block;
    for (int rr = wave; rr < rows_per_block; rr += NW) {
        const long long row = row0 + rr;
        if (row >= total_rows) break;
        const int p = (int)(row % Ho);
        const long long n = row / Ho;
        // stage input rows p..p+2 as fp32, 4 floats per pixel (channels >= CIN are never read)
        const h16_t* xr = x + ((n * Hi + p) * Wi) * CIN;
#pragma unroll
        for (int r = 0; r < 3; ++r)
            for (int i = lane; i < Wi * CIN; i += 64) {
                const int px = i / CIN, c = i - px * CIN;
                xw[r * rowf + px * PP + c] = h2f(xr[r * Wi * CIN + i]);
            }
        const h16_t* dr = dy + (((n * Hd + p + dy_pad) * Wd) + dy_pad) * (long long)K + k;
        if (KS == 1 && (Wo & 7) == 0) {
            // one stream per wave (K = 64): UQ pixels per iteration share their UQ+2 window columns (aligned 16-byte
            // broadcast reads) and the UQ dy loads are issued together (the loop is bound by their latency)
            constexpr int UQ = 8;
            for (int q = 0; q < Wo; q += UQ) {
                float d[UQ];
#pragma unroll
                for (int u = 0; u < UQ; ++u) d[u] = h2f(dr[(long long)(q + u) * K]);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    float4 col[UQ + 2];
#pragma unroll
                    for (int e = 0; e < UQ + 2; ++e) col[e] = *reinterpret_cast<const float4*>(xw + r * rowf + (q + e) * 4);
#pragma unroll
                    for (int u = 0; u < UQ; ++u)
#pragma unroll
                        for (int s = 0; s < 3; ++s) {
                            const float vv[4] = {col[u + s].x, col[u + s].y, col[u + s].z, col[u + s].w};
#pragma unroll
                            for (int c = 0; c < CIN; ++c)
                                acc[(r * 3 + s) * CIN + c] = fmaf(d[u], vv[c], acc[(r * 3 + s) * CIN + c]);
                        }
                }
            }
        } else if (KS > 1) {
            // K = 16 / 32 (Pong-class first layers): a pixel stream walks ONE contiguous quarter / half of the row, so its
            // 3-wide window slides in registers: one new column = 3 LDS reads per pixel instead of 9 (the interleaved walk
            // below re-read the whole window for every pixel and was LDS-bound: 102 us where the gradient tensor takes 15).
            const int seg = (Wo + KS - 1) / KS;
            const int q0 = stream * seg, q1 = q0 + seg < Wo ? q0 + seg : Wo;
            float w0[3][CIN], w1[3][CIN];
            auto column = [&](int r, int q, float (&o)[CIN]) {
                if (CIN == 1) { o[0] = xw[r * rowf + q]; }
                else {
                    const float4 v = *reinterpret_cast<const float4*>(xw + r * rowf + q * 4);
                    const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int c = 0; c < CIN; ++c) o[c] = vv[c];
                }
            };
            if (q0 < q1) {
#pragma unroll
                for (int r = 0; r < 3; ++r) { column(r, q0, w0[r]); column(r, q0 + 1, w1[r]); }
            }
#pragma unroll 8
            for (int q = q0; q < q1; ++q) {
                const float d = h2f(dr[(long long)q * K]);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    float nw[CIN];
                    column(r, q + 2, nw);
#pragma unroll
                    for (int c = 0; c < CIN; ++c) {
                        acc[(r * 3 + 0) * CIN + c] = fmaf(d, w0[r][c], acc[(r * 3 + 0) * CIN + c]);
                        acc[(r * 3 + 1) * CIN + c] = fmaf(d, w1[r][c], acc[(r * 3 + 1) * CIN + c]);
                        acc[(r * 3 + 2) * CIN + c] = fmaf(d, nw[c], acc[(r * 3 + 2) * CIN + c]);
                        w0[r][c] = w1[r][c];
                        w1[r][c] = nw[c];
                    }
                }
            }
        } else {
#pragma unroll 4
            for (int q = stream; q < Wo; q += KS) {
                const float d = h2f(dr[(long long)q * K]);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const float4 v = *reinterpret_cast<const float4*>(xw + r * rowf + (q + s) * 4);
                        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                        for (int c = 0; c < CIN; ++c) acc[(r * 3 + s) * CIN + c] = fmaf(d, vv[c], acc[(r * 3 + s) * CIN + c]);
                    }
            }
        }
    }
    // fold the pixel streams (lanes k, k+K, ...), then the 4 wavefronts
    for (int o = K; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < J; ++j) acc[j] += __shfl_xor(acc[j], o, 64);
    __syncthreads();                                          // everyone is done with the staged rows
    float* red = reinterpret_cast<float*>(smem);              // [NW][K][J]
    if (stream == 0)
#pragma unroll
        for (int j = 0; j < J; ++j) red[(wave * K + k) * J + j] = acc[j];
    __syncthreads();
    for (int i = tid; i < K * J; i += NW * 64) {
        float s = red[i];
#pragma unroll
        for (int w2 = 1; w2 < NW; ++w2) s += red[w2 * K * J + i];   // fixed order
        scratch[(long long)blockIdx.x * K * J + i] = s;       // a partial is [K][9][CIN] = [K][J]
    }
}


// ---------------------------------------------------------------------------------------------------------------
// MFMA versions for the image-sized first layer (K = 64, Cin <= 3, Wo % 32 == 0).  The VALU kernels above spend
// ~4000 issue cycles per 32 pixels on the 27 x 64 FMAs; as a (64 x 32) x (32 x 32px) MFMA product the arithmetic is
// 4 instructions and both kernels become what they should be: streaming the [N][H][W][64] bf16 tensor once.
// A wavefront works alone on groups of 32 consecutive pixels of an image row (no workgroup barriers in the loops):
// it copies the 3 x 34 pixel input window into its own LDS strip as it lies in memory (dword copies) and builds the
// reduction-major fragment (8 consecutive (tap, channel) indices of one pixel) with 2-byte LDS reads.
constexpr int FG = 32;            // pixels per group
constexpr int F_RS = 256;         // bytes per staged input row: (32 + 2) pixels * Cin * 2 B <= 204
constexpr int F_XSZ = 1024;       // 3 rows + a zero slot (offset 3 * F_RS) for the padded reduction indices

struct GroupPos { long long n; int p, q; };
__device__ __forceinline__ GroupPos group_pos(long long g, int Ho, int Wo) {
    const long long m0 = g * FG;
    GroupPos r;
    r.q = (int)(m0 % Wo);
    const long long t = m0 / Wo;
    r.p = (int)(t % Ho);
    r.n = t / Ho;
    return r;
}
__device__ __forceinline__ void group_next(GroupPos& c, int Ho, int Wo) {   // the following group, without divisions
    c.q += FG;
    if (c.q >= Wo) { c.q = 0; if (++c.p >= Ho) { c.p = 0; ++c.n; } }
}

template <int CIN>
__global__ __launch_bounds__(256) void conv3_first_fwd_mfma_k(const h16_t* __restrict__ x, const h16_t* __restrict__ w,
                                                              const float* __restrict__ bias, h16_t* __restrict__ y,
                                                              float* __restrict__ partials, int Hi, int Wi,
                                                              long long groups, int gpw) {
    constexpr int J = 9 * CIN, NS = (J + 15) / 16, K = 64, KT = 2;
    constexpr int ND = (FG + 2) * CIN * 2 / 4;            // dwords per staged row
    constexpr int WSZ = F_XSZ + FG * K * 2;               // per wave: input strip + 32 px x 64 ch output staging
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * WSZ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* xs = smem + wave * WSZ;
    unsigned char* stg = xs + F_XSZ;
    const int px = lane & 31, h = lane >> 5;
    const int Ho = Hi - 2, Wo = Wi - 2;
    if (lane < 4) *reinterpret_cast<cf_u32_a*>(xs + 3 * F_RS + lane * 4) = 0u;

    // weights [K][J] -> A fragments (row = output channel, 8 consecutive reduction indices per lane), once per wave
    h16x8 wa[KT][NS];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = s * 16 + h * 8 + i;
                wa[kt][s][i] = j < J ? (short)w[(kt * 32 + px) * J + j] : (short)0;
            }
    // accumulator layout: lane = pixel, register i = channel 8*(i>>2) + 4*h + (i&3) (+32 kt); start from the bias
    float bs[KT][16];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) bs[kt][i] = bias ? bias[kt * 32 + 8 * (i >> 2) + 4 * h + (i & 3)] : 0.f;
    // LDS byte offsets of this lane's 8 * NS reduction elements (lane constants)
    int xo[NS][8];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = s * 16 + h * 8 + i;
            xo[s][i] = j < J ? (j / (3 * CIN)) * F_RS + (px * CIN + j % (3 * CIN)) * 2 : 3 * F_RS;
        }
    float sm[KT][16], sq[KT][16];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) sm[kt][i] = sq[kt][i] = 0.f;

    const long long wid = (long long)blockIdx.x * 4 + wave;
    const long long g0 = wid * gpw, g1 = g0 + gpw < groups ? g0 + gpw : groups;
    uint32_t xr[3] = {0u, 0u, 0u};
    GroupPos c = group_pos(g0 < groups ? g0 : 0, Ho, Wo);   // fetch cursor
    auto fetch = [&]() {
        if (lane < ND) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
                xr[r] = reinterpret_cast<const uint32_t*>(x + ((c.n * Hi + c.p + r) * Wi + c.q) * CIN)[lane];
        }
        group_next(c, Ho, Wo);
    };
    if (g0 < g1) fetch();
    for (long long g = g0; g < g1; ++g) {
        if (lane < ND) {
#pragma unroll
            for (int r = 0; r < 3; ++r) *reinterpret_cast<cf_u32_a*>(xs + r * F_RS + lane * 4) = xr[r];
        }
        h16x8 xb[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) xb[s][i] = *reinterpret_cast<const cf_s16_a*>(xs + xo[s][i]);
        if (g + 1 < g1) fetch();
        f32x16 acc[KT];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[kt][i] = bs[kt][i];
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[kt] = kp_mfma16(wa[kt][s], xb[s], acc[kt], 0, 0, 0);
        }
        // statistics (fp32 values before rounding), bf16 pack, transpose through the wave's LDS tile, 1-KiB stores
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[kt][g4 * 4 + e];
                    sm[kt][g4 * 4 + e] += v[e];
                    sq[kt][g4 * 4 + e] = __builtin_fmaf(v[e], v[e], sq[kt][g4 * 4 + e]);
                }
                uint2 pk;
                pk.x = pack_h16x2(v[0], v[1]);
                pk.y = pack_h16x2(v[2], v[3]);
                const int chunk = kt * 4 + g4;
                *reinterpret_cast<uint2*>(stg + px * (K * 2) + ((chunk ^ (px & 7)) << 4) + h * 8) = pk;
            }
        h16_t* yb = y + g * FG * K;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = i * 64 + lane, pp = id >> 3, c16 = id & 7;
            const uint4 vv = *reinterpret_cast<const uint4*>(stg + pp * (K * 2) + ((c16 ^ (pp & 7)) << 4));
            *reinterpret_cast<uint4*>(yb + (size_t)id * 8) = vv;
        }
    }
    if (partials) {   // one row per wavefront (zeros when it had no group)
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            int r1, r2;
            const float ts = reduce16_over32(sm[kt], lane, r1);
            const float tq = reduce16_over32(sq[kt], lane, r2);
            if ((lane & 1) == 0) {
                const int k = kt * 32 + 8 * (r1 >> 2) + 4 * h + (r1 & 3);
                partials[(wid * 2 + 0) * K + k] = ts;
                partials[(wid * 2 + 1) * K + k] = tq;
            }
        }
    }
}

// partial[b][k][tap][c] = sum over the pixels of workgroup b of dy[pixel][k] * x[pixel + tap][c]; the partial buffers
// are summed in a fixed order by kp_wgrad_unpack (no atomics: bitwise reproducible).  dy tiles (32 px x 128 B) go to
// LDS by DMA as they lie and are read with the transpose read (channel-major fragments), the input window as above.
template <int CIN>
__global__ __launch_bounds__(1024) void conv3_first_wgrad_mfma_k(const h16_t* __restrict__ x, const h16_t* __restrict__ dy,
                                                                 float* __restrict__ partial, int Ho, int Wo, int dy_pad,
                                                                 long long groups, int gpw) {
    constexpr int J = 9 * CIN, K = 64, KT = 2, PK = K * 2, NW = 16;
    constexpr int ND = (FG + 2) * CIN * 2 / 4;
    constexpr int DYB = FG * PK;                           // 4 KiB dy tile
    constexpr int WSZ = 2 * DYB + F_XSZ;                   // per wave: double-buffered dy tile + input strip
    __shared__ __attribute__((aligned(16))) unsigned char smem[NW * WSZ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* dyb = smem + wave * WSZ;
    unsigned char* xs = dyb + 2 * DYB;
    const int Hi = Ho + 2, Wi = Wo + 2, Hd = Ho + 2 * dy_pad, Wd = Wo + 2 * dy_pad;

    // dy DMA: instruction u covers tile rows (pixels) 8u..8u+7; 64-byte blocks swizzled by the row (kp_mfma_util.h)
    int dvoff[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int row = u * 8 + (lane >> 3), slot = lane & 7;
        dvoff[u] = row * PK + (((slot >> 2) ^ blk_swz<PK>(row)) << 6) + (slot & 3) * 16;
    }
    // transpose-read lane roles (see wgrad3x3_patch_k)
    const int t16 = lane & 15, g16 = (lane >> 4) & 1, h = lane >> 5;
    const int lrow = 8 * h + (t16 >> 2);
    int kbase[KT][4];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int v = 0; v < 4; ++v) kbase[kt][v] = tr_base<PK>(v, lrow, (kt * 32 + 16 * g16 + (t16 & 3) * 4) * 2);
    // input fragment: column j = (tap, channel) of this lane, 8 consecutive pixels (stride Cin elements)
    const int jj = (lane & 31) < J ? (lane & 31) : J - 1;   // columns >= J are discarded: any valid address
    const int xoff = (jj / (3 * CIN)) * F_RS + (jj % (3 * CIN)) * 2 + 8 * h * CIN * 2;

    f32x16 acc[KT];
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[kt][i] = 0.f;

    const long long wid = (long long)blockIdx.x * NW + wave;
    const long long g0 = wid * gpw, g1 = g0 + gpw < groups ? g0 + gpw : groups;
    uint32_t xr[3] = {0u, 0u, 0u};
    GroupPos c = group_pos(g0 < groups ? g0 : 0, Ho, Wo);   // fetch cursor
    auto fetch = [&](int buf) {
        const h16_t* db = dy + (((c.n * Hd + c.p + dy_pad) * Wd) + c.q + dy_pad) * K;
        const u32x4 dr = rsrc_words(db);   // asm DMA: the builtin would make every transpose read wait for it (kp_mfma_util.h)
#pragma unroll
        for (int u = 0; u < 4; ++u) dma16_asm(dr, dyb + buf * DYB + u * 1024, dvoff[u]);
        if (lane < ND) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
                xr[r] = reinterpret_cast<const uint32_t*>(x + ((c.n * Hi + c.p + r) * Wi + c.q) * CIN)[lane];
        }
        group_next(c, Ho, Wo);
    };
    if (g0 < g1) fetch(0);
    for (long long g = g0; g < g1; ++g) {
        const int buf = (int)(g - g0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this group's dy tile and input rows have arrived
        if (lane < ND) {
#pragma unroll
            for (int r = 0; r < 3; ++r) *reinterpret_cast<cf_u32_a*>(xs + r * F_RS + lane * 4) = xr[r];
        }
        if (g + 1 < g1) fetch(buf ^ 1);                       // in flight during this group's reads and MFMAs
        h16x8 xb[2];
#pragma unroll
        for (int R = 0; R < 2; ++R)
#pragma unroll
            for (int i = 0; i < 8; ++i) xb[R][i] = *reinterpret_cast<const cf_s16_a*>(xs + xoff + (R * 16 + i) * CIN * 2);
        const unsigned char* tile = dyb + buf * DYB;
        h16x8 a[KT][2];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            a[kt][0] = tr_frag<PK, 0>(tile, kbase[kt]);
            a[kt][1] = tr_frag<PK, 16>(tile, kbase[kt]);
        }
#pragma unroll
        for (int kt = 0; kt < KT; ++kt)
#pragma unroll
            for (int R = 0; R < 2; ++R) acc[kt] = kp_mfma16(a[kt][R], xb[R], acc[kt], 0, 0, 0);
    }
    // fold the 16 wavefronts in a fixed order; D[k][j]: lane column = j, rows k = 8*(i>>2) + 4*h + (i&3) (+32 kt)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);              // [16 waves][KT*16][64 lanes]
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int i = 0; i < 16; ++i) red[(wave * KT * 16 + kt * 16 + i) * 64 + lane] = acc[kt][i];
    __syncthreads();
    for (int e = tid; e < KT * 16 * 64; e += 1024) {
        float s = 0.f;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) s += red[wv * KT * 16 * 64 + e];
        const int l = e & 63, i = (e >> 6) & 15, kt = e >> 10;
        const int k = kt * 32 + 8 * (i >> 2) + 4 * (l >> 5) + (i & 3), j = l & 31;
        if (j < J) partial[((long long)blockIdx.x * K + k) * J + j] = s;
    }
}

}  // namespace

// pair != 0 (kp_conv_fwd_pair): N images = two half-batches, each launched exactly as a separate call of N / 2 images would be
// (same grid, same statistics rows: bit-identical), in one grid; the VALU kernel only (the Pong-class first layers)
int kp_conv_first_fwd(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi,
                      int C, int K, hipStream_t st, int pair) {
    if (pair && (N & 1)) return KP_E_UNSUPPORTED;
    const int Nh = pair ? N / 2 : N;
    const long long M = (long long)Nh * (Hi - 2) * (Wi - 2);
    const int CP = K / 8;
    if (C < 1 || C > 4 || K % 8 != 0 || K > 128 || (CP & (CP - 1)) != 0 || M % 64 != 0) return KP_E_UNSUPPORTED;
    if (K == 64 && C <= 3 && (Wi - 2) % FG == 0 && M >= 4096) {
        if (pair) return KP_E_UNSUPPORTED;
        // MFMA path: 8 groups of 32 pixels per wavefront, one statistics row per wavefront (<= M / 64 rows)
        const long long groups = M / FG;
        const int gpw = 8;
        const unsigned grid = (unsigned)((groups + 4 * gpw - 1) / (4 * gpw));
        if (stats) kp_set_last_rows((long long)grid * 4);
        KP_QUERY_RETURN();
#define KP_FIRST_FWD_M(CINv)                                                                                       \
    conv3_first_fwd_mfma_k<CINv><<<grid, 256, 0, st>>>((const h16_t*)x, (const h16_t*)w, bias, (h16_t*)y, stats, \
                                                       Hi, Wi, groups, gpw)
        switch (C) {
            case 1: KP_FIRST_FWD_M(1); break;
            case 2: KP_FIRST_FWD_M(2); break;
            default: KP_FIRST_FWD_M(3); break;
        }
#undef KP_FIRST_FWD_M
        return kp_check_launch("conv3_first_fwd_mfma");
    }
    extern kp_knob_t g_kp_first_rows;
    const int rows_knob = g_kp_first_rows;
    if (C == 1 && (K == 16 || K == 32) && (Wi & 1) == 0 && Wi <= 256 && rows_knob) {
        // row kernel: whole output rows per wavefront, 4 wavefronts per workgroup, one statistics row per workgroup
        const long long rows = (long long)Nh * (Hi - 2);
        long long wg = (rows + 3) / 4;
        const long long cap = rows_knob > 1 ? rows_knob : 768;   // measured 256 .. 2048 (profiles/r05l_first_rows.txt): flat between 512 and 1024
        if (wg > cap) wg = cap;
        // one statistics row per workgroup: the caller's buffer (include/kp_hip.h, kp_conv_fwd) holds ceil(M / 64) rows, which four output
        // rows per workgroup exceed on maps narrower than 16 pixels -- the grid-stride loop of the kernel takes any grid
        if (wg > M / 64) wg = M / 64;
        const unsigned grid = (unsigned)wg;
        if (stats) kp_set_last_rows((long long)grid);
        KP_QUERY_RETURN();
        const long long xh = (long long)Nh * Hi * Wi, yh = M * K;
#define KP_FIRST_ROWS(KBv, WIDEv)                                                                                             \
    conv3_first_fwd_rows_k<KBv, WIDEv><<<pair ? 2 * grid : grid, 256, 0, st>>>((const h16_t*)x, (const h16_t*)w, bias, (h16_t*)y, \
                                                                               stats, Hi, Wi, rows, pair ? (int)grid : 0, xh, yh)
        if (K == 16) { if (Wi > 128) KP_FIRST_ROWS(2, true); else KP_FIRST_ROWS(2, false); }
        else { if (Wi > 128) KP_FIRST_ROWS(4, true); else KP_FIRST_ROWS(4, false); }
#undef KP_FIRST_ROWS
        return kp_check_launch("conv3_first_fwd_rows");
    }
    long long wgs = (M + TPB - 1) / TPB;
    if (wgs > 2048) wgs = 2048;                     // (the caller's statistics buffer holds M / 64 rows)
    const unsigned grid = (unsigned)wgs;
    if (stats) kp_set_last_rows((long long)grid);   // one statistics row per workgroup
    KP_QUERY_RETURN();
    const size_t lds = (size_t)9 * C * K * sizeof(float) + (size_t)4 * 64 * K * 2 + (size_t)4 * 2 * K * sizeof(float);
    const long long xh = (long long)Nh * Hi * Wi * C, yh = M * K;
#define KP_FIRST_FWD(CINv)                                                                                          \
    conv3_first_fwd_k<CINv><<<pair ? 2 * grid : grid, TPB, lds, st>>>((const h16_t*)x, (const h16_t*)w, bias, (h16_t*)y, stats, Hi, \
                                                    Wi, K, M, pair ? (int)grid : 0, xh, yh)
    switch (C) {
        case 1: KP_FIRST_FWD(1); break;
        case 2: KP_FIRST_FWD(2); break;
        case 3: KP_FIRST_FWD(3); break;
        default: KP_FIRST_FWD(4); break;
    }
#undef KP_FIRST_FWD
    return kp_check_launch("conv3_first_fwd");
}

// MFMA weight-gradient plan: workgroups of 16 wavefronts, one partial gradient buffer per workgroup.
static bool first_wgrad_plan(int N, int Ho, int Wo, int C, int K, long long& groups, int& gpw, unsigned& grid) {
    const long long M = (long long)N * Ho * Wo;
    if (!(K == 64 && C >= 1 && C <= 3 && Wo % FG == 0 && M >= 65536)) return false;
    groups = M / FG;
    int cus = kp_device_cus();   // (the CU budget of the launching lane: kp_config(KP_CFG_CUS))
    if (cus <= 0) cus = 256;
    long long per_wg = (groups + cus - 1) / cus;            // one workgroup (160 KB of LDS) per CU
    gpw = (int)((per_wg + 15) / 16);
    if (gpw < 1) gpw = 1;
    grid = (unsigned)((groups + 16LL * gpw - 1) / (16LL * gpw));
    return true;
}

// VALU kernel: workgroups (= partial gradients), 0 if the shape is not taken
static unsigned first_wgrad_valu_plan(int N, int Ho, int Wo, int C, int K, int& rpb, size_t& lds, int* nw_out = nullptr) {
    if (C < 1 || C > 4 || !(K == 16 || K == 32 || K == 64)) return 0;
    const long long rows = (long long)N * Ho;
    const int pp = (C == 1 && K < 64) ? 1 : 4;  // floats per staged pixel (the kernel's PP)
    int nw = K < 64 ? 16 : 4;                   // waves per workgroup (see the kernel): as many as the LDS strips admit
    while (nw > 4 && ((size_t)nw * 3 * (Wo + 2) * pp * sizeof(float) > 48 * 1024 || (size_t)nw * K * 9 * C * sizeof(float) > 48 * 1024)) nw >>= 1;
    while (nw > 4 && rows < 4LL * nw) nw >>= 1;   // tiny inputs: keep several workgroups (and the P > 1 partial protocol)
    rpb = (int)((rows + 511) / 512);            // <= 512 workgroups (2048 measured: kernel 58.6 -> 52.6 us, but the sum over 4x the
                                                // partials in kp_wgrad_unpack 9 -> 22 us)
    if (rpb < nw) rpb = nw;
    rpb = (rpb + nw - 1) / nw * nw;
    lds = (size_t)nw * 3 * (Wo + 2) * pp * sizeof(float);
    const size_t red = (size_t)nw * K * 9 * C * sizeof(float);
    if (red > lds) lds = red;
    if (lds > 64 * 1024) return 0;
    if (nw_out) *nw_out = nw;
    return (unsigned)((rows + rpb - 1) / rpb);
}

long long kp_conv_first_wgrad_partials(int N, int Ho, int Wo, int C, int K) {
    long long groups; int gpw; unsigned grid;
    if (first_wgrad_plan(N, Ho, Wo, C, K, groups, gpw, grid)) return (long long)grid;
    int rpb; size_t lds;
    return first_wgrad_valu_plan(N, Ho, Wo, C, K, rpb, lds);
}

int kp_conv_first_wgrad(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K,
                        int dy_pad, hipStream_t st) {
    if (C < 1 || C > 4 || !(K == 16 || K == 32 || K == 64)) return KP_E_UNSUPPORTED;
    {
        long long groups; int gpw; unsigned grid;
        if (first_wgrad_plan(N, Ho, Wo, C, K, groups, gpw, grid)) {
#define KP_FIRST_WG_M(CINv)                                                                                          \
    conv3_first_wgrad_mfma_k<CINv><<<grid, 1024, 0, st>>>((const h16_t*)x, (const h16_t*)dy, scratch, Ho, Wo, dy_pad, \
                                                          groups, gpw)
            switch (C) {
                case 1: KP_FIRST_WG_M(1); break;
                case 2: KP_FIRST_WG_M(2); break;
                default: KP_FIRST_WG_M(3); break;
            }
#undef KP_FIRST_WG_M
            return kp_check_launch("conv3_first_wgrad_mfma");
        }
    }
    const long long rows = (long long)N * Ho;
    int rpb, nw = 4; size_t lds;
    const unsigned grid = first_wgrad_valu_plan(N, Ho, Wo, C, K, rpb, lds, &nw);
    if (!grid) return KP_E_UNSUPPORTED;
#define KP_FIRST_WG(CINv)                                                                                        \
    do {                                                                                                         \
        if (nw == 16) conv3_first_wgrad_k<CINv, 16><<<grid, 1024, lds, st>>>((const h16_t*)x, (const h16_t*)dy, scratch, Ho, Wo, K, dy_pad, rows, rpb); \
        else if (nw == 8) conv3_first_wgrad_k<CINv, 8><<<grid, 512, lds, st>>>((const h16_t*)x, (const h16_t*)dy, scratch, Ho, Wo, K, dy_pad, rows, rpb); \
        else conv3_first_wgrad_k<CINv, 4><<<grid, 256, lds, st>>>((const h16_t*)x, (const h16_t*)dy, scratch, Ho, Wo, K, dy_pad, rows, rpb);     \
    } while (0)
    switch (C) {
        case 1: KP_FIRST_WG(1); break;
        case 2: KP_FIRST_WG(2); break;
        case 3: KP_FIRST_WG(3); break;
        default: KP_FIRST_WG(4); break;
    }
#undef KP_FIRST_WG
    return kp_check_launch("conv3_first_wgrad");
}
