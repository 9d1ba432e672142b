// 3x3 convolution for SMALL channel counts (C, K in {16, 32}: the Pong-class nets, vgg.py:53,66 VGG_PONG_*), 16-bit storage.
// These layers are HBM-bound (16.6 GFLOP against 174 MB per layer at 256 x 84 x 84), and their images (84 x 84) are
// not 16x16-tileable, so neither MFMA patch kernel applies and the general fp32 kernels sat ~30x off the roofline.
//
// Kernels of this file, in the order the dispatch prefers them (DESIGN.md 3.9 has the measurements behind each):
//   conv3x3_small_band_k   forward / data gradient, a band of image rows staged through LDS, software-pipelined items; forms:
//                          BNIN (the producing layer's BatchNorm-apply + activation + replicate pad on the staging path),
//                          STATS (BatchNorm statistics rows), BRED (the layer below's first BatchNorm-backward pass in the epilogue)
//   wgrad3x3_small16_k     weight gradient on the 16 x 16 x 32 MFMA shape: (16, 16) six waves with one 16 x 16 tile each, one 32-channel
//                          side three waves with two tiles each, (32, 32) four waves with two tiles each; two workgroups per CU; also
//                          the 1-4-channel first layer (CIN)
//   wgrad3x3_small_band_k  weight gradient on the 32 x 32 x 16 shape: one wave per SIMD, LDS double buffer (shapes whose band does not
//                          fit the other kernel's 76 KB; KP_WG16=0)
//   conv3x3_small_k / wgrad3x3_small_k   the LDS-free kernels of round 2 (shapes whose band does not fit; described next)
//
// forward / data-gradient (conv3x3_small_k): one wavefront per 32-pixel row segment, no LDS, no barriers.
//   With C = 16 one tap IS one MFMA reduction step (v_mfma_f32_32x32x16_bf16): the pixel operand of tap (r,s) is a
//   straight coalesced 16-byte-per-lane load of 32 consecutive padded-image pixels (1 KiB, L1/L2-hot after the first
//   tap), the 9 (18 for C = 32) weight fragments stay in registers for the whole kernel.  D = [32 k][32 px]; K = 16
//   uses half of it.  Epilogue: bias, running BN partial sums per lane (one partial row per wavefront at the end),
//   v_permlane32_swap pairs the two half-waves' 8-byte pieces into 16-byte stores.
// weight gradient (wgrad3x3_small_k): also MFMA and LDS-free: one 32x32 [k][c] tile per tap (9 accumulators), the
//   pixel-reduction fragments gathered with 2-byte loads at a channel stride (L1-resident 64-byte runs); the 4
//   wavefronts of a workgroup are summed through LDS, then fp32 atomics into the zeroed scratch.
#include <type_traits>
#include "kp_common.h"
extern kp_knob_t g_kp_wg16;   // kp_api.hip: kp_config(KP_CFG_WG16, bits), default 7

namespace {

typedef __attribute__((ext_vector_type(8))) short h16x8;
typedef __attribute__((ext_vector_type(4))) short h16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define LDS_AS __attribute__((address_space(3)))

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
__global__ __launch_bounds__(TPB) void conv3x3_small_k(const h16_t* __restrict__ x, const h16_t* __restrict__ w,
                                                       const float* __restrict__ bias, h16_t* __restrict__ y,
                                                       float* __restrict__ partials, int N, int Hi, int Wi, int K,
                                                       long long nseg, int segs_per_row) {
    constexpr int C = CB * 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int Ho = Hi - 2, Wo = Wi - 2;

    // weight fragments: lane (k = fr, half = fh): 8 input channels of W[k][tap][cb*16 + fh*8 ...]; rows k >= K are zero.
    // They live in LDS (one copy per workgroup, read back lane-linear: conflict-free), not in 36 / 72 registers: 94-96 VGPRs
    // instead of 128 / 164, i.e. 4-5 waves per SIMD instead of 2-3 for a kernel whose pace is load latency x occupancy
    // (config 2: 5.85 -> 5.5 ms per step).
    __shared__ h16x8 swf[9 * CB][64];
    if (wave == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (fr < K) v = *reinterpret_cast<const h16x8*>(w + ((long long)fr * 9 + t) * C + cb * 16 + fh * 8);
                swf[t * CB + cb][lane] = v;
            }
    }
    __syncthreads();
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
        const h16_t* xp = x + ((n * Hi + p) * Wi + qc) * (long long)C + fh * 8;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        h16x8 xf[9 * CB];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
                xf[t * CB + cb] = *reinterpret_cast<const h16x8*>(xp + ((long long)(t / 3) * Wi + (t % 3)) * C + cb * 16);
#pragma unroll
        for (int t = 0; t < 9 * CB; ++t) acc = kp_mfma16(swf[t][lane], xf[t], acc, 0, 0, 0);
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
            pk[2 * g] = pack_h16x2(v[0], v[1]);
            pk[2 * g + 1] = pack_h16x2(v[2], v[3]);
        }
        // pair the half-waves' 8-byte pieces: after the swaps lanes fh=0 hold channels 8g..8g+7 of chunk g (even g),
        // lanes fh=1 those of chunk g+1
        h16_t* yp = y + ((n * Ho + p) * Wo + q) * (long long)K;
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

// ---- BatchNorm-apply + activation of the PREVIOUS layer on the staging path (kp_conv_fwd_bnin / kp_conv_wgrad_bnin) --------
// A 3x3 layer whose input is another 3x3 layer's output reads that layer's conv output y_prev [N][H][W][C] (no border) and
// applies  a = act(bn(y_prev[clamp(i - 1)][clamp(j - 1)]))  -- BatchNorm2d (train: batch statistics), LeakyReLU / ReLU and
// ReplicationPad2d(1), vgg.py:30-38 -- between the global load and the LDS store of the band staging: the VALU is idle in these
// kernels, and the padded activation tensor (write + read of the layer's output, one whole launch) never exists.  The
// arithmetic is kp_bn_act_fwd's expression for expression ((y - mean) * invstd * gamma + beta, act, RNE to 16 bits), so the
// fragments the MFMAs see are bit-identical to what the separate pass would have stored.
// two half-batches in one launch (see conv3x3_small_band_k): workgroups per half, element offsets of the second half's tensors
struct KpPair { int hb = 0; long long x = 0, y = 0, p = 0; };
typedef KpBnIn BnIn;   // kp_common.h: {mean, invstd, gamma, beta, slope}; mean == nullptr: no BatchNorm, activation only
// per-channel constants of the layer in LDS: [4][32] floats (mean, invstd, gamma, beta).  Without BatchNorm the constants are
// the EXACT identity of the expression below: (y - (+0)) * 1 * 1 + (-0) == y for every y, signed zeros included.
__device__ __forceinline__ void bnin_fill(float (*cst)[32], const BnIn& bn, int C, int tid) {
    if (tid < C) {
        const bool aff = bn.mean != nullptr;
        cst[0][tid] = aff ? bn.mean[tid] : 0.f;
        cst[1][tid] = aff ? bn.invstd[tid] : 1.f;
        cst[2][tid] = aff ? bn.gamma[tid] : 1.f;
        cst[3][tid] = aff ? bn.beta[tid] : -0.f;
    }
}
// a thread's 8 channels (c0 .. c0 + 7) of the four constants, LDS -> registers: once per band, issued behind the band's global
// loads (re-reading them per 16-byte chunk put 16 dependent LDS round trips in front of every chunk: +40 us per launch)
struct BnReg {
    float mu[8], is[8], ga[8], be[8];
    __device__ __forceinline__ void load(const float (*cst)[32], int c0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 a = *reinterpret_cast<const float4*>(&cst[0][c0 + 4 * h]);
            const float4 b = *reinterpret_cast<const float4*>(&cst[1][c0 + 4 * h]);
            const float4 c = *reinterpret_cast<const float4*>(&cst[2][c0 + 4 * h]);
            const float4 d = *reinterpret_cast<const float4*>(&cst[3][c0 + 4 * h]);
            mu[4 * h] = a.x; mu[4 * h + 1] = a.y; mu[4 * h + 2] = a.z; mu[4 * h + 3] = a.w;
            is[4 * h] = b.x; is[4 * h + 1] = b.y; is[4 * h + 2] = b.z; is[4 * h + 3] = b.w;
            ga[4 * h] = c.x; ga[4 * h + 1] = c.y; ga[4 * h + 2] = c.z; ga[4 * h + 3] = c.w;
            be[4 * h] = d.x; be[4 * h + 1] = d.y; be[4 * h + 2] = d.z; be[4 * h + 3] = d.w;
        }
    }
};
// one 16-byte chunk = 8 consecutive channels of a pixel.  act(z) = max(z, z * slope) is (z > 0 ? z : z * slope) bit for bit
// for 0 <= slope <= 1 (the launchers check it), zeros and NaNs included, without the compare + select per element.
__device__ __forceinline__ uint4 bnin_apply(const uint4 raw, const BnReg& k, float slope) {
    return kp_bnin8(raw, k.mu, k.is, k.ga, k.be, slope);
}

// ---- forward / data gradient with the input staged through LDS ------------------------------------------------------------
// The kernel above reads every input pixel nine times (once per tap) through the L1: 0.5-1 GB of L2 -> L1 traffic per layer,
// which is what sets its pace (2-2.6 TB/s of algorithmic bytes).  Here a workgroup copies a band of R + 2 padded input rows
// into LDS once (16-byte loads, all in flight before the first LDS store; pixel pitch C * 2 + 16 bytes: the 16 lanes of a
// ds_read_b128 group land on 16 distinct 4-bank blocks for both channel counts) and its four wavefronts take the (row,
// 32-pixel segment) items of the band: 9 (18) fragment reads from LDS per item, weights from LDS as above, same epilogue.
// Statistics: one partial row per wavefront of the grid, accumulated over its items in a fixed order (deterministic).
#ifndef KP_FB_ABL
#define KP_FB_ABL 0   // measurement builds only (tools/probes/small_abl.sh): 1 no global loads, 2 no items, 4 no stores, 8 no MFMAs, 16 no fragment reads
#endif
constexpr int FB_LDS = 61440;   // band bytes (the weights take 9 / 18 KB more): two workgroups per CU
// Workgroup size of the band kernel (a build-time knob for A/B runs: tools/probes/small_variant.sh).  Eight-wave workgroups
// (4 waves per SIMD at <= 128 VGPRs instead of 2) run the kernels 5-8 % faster in isolation, but the BatchNorm-input variants
// then spill, and in the train step -- where these kernels share the chip with side-stream work -- the step was 3 % SLOWER
// (3.88 vs 4.00 ms, one box, interleaved): four waves it is.
#ifndef KP_FB_TPB
#define KP_FB_TPB 256
#endif
constexpr int FTPB = KP_FB_TPB, FNW = FTPB / 64;
// statistics / BatchNorm-backward rows a band-kernel workgroup leaves: 1 (its wavefronts' rows summed through LDS in the kernel's
// tail, in wave order: the dependent finalize launch reads 512 rows instead of 2 048) or FNW (one per wavefront)
#ifndef KP_FB_ROWS
#define KP_FB_ROWS 1
#endif
constexpr int FROWS = KP_FB_ROWS == 1 ? 1 : FNW;
constexpr int FSL = 12 * 256 / FTPB;   // 16-byte chunks per thread in one band (FTPB x FSL x 16 B = 48 KB of input)
// BNIN: x is the previous layer's conv output [N][Hi - 2][Wi - 2][C] and `bn` its BatchNorm / activation (see BnIn above);
// otherwise x is the padded activation [N][Hi][Wi][C] as stored.
// The kernel is bound by VALU ISSUE, not by HBM (rocprofv3 SQ counters, tools/probes/small_pmc.sh: 3 200 VALU instructions per
// wave for 142 MFMAs -- the VALU pipe is busy for > 40 % of the launch and ~70 % of the time waves are resident, HBM moves 3.5
// of ~5 TB/s).  So the body is written to keep work OFF the vector ALU: everything that is uniform over a wavefront (item ->
// row / segment, output row base, LDS row base) is computed on the scalar unit (`wave` comes from readfirstlane), the per-lane
// parts of every address are constants computed once, the staging tables (LDS offset, source offset, row of each of a
// thread's chunks) are built once per kernel instead of once per band, statistics use packed FMAs and are skipped -- like
// the lane mask of ragged segments -- where a wave-uniform test says they are not needed.
// WREG: the 9 x CB weight fragments live in registers (36 / 72 VGPRs) instead of LDS: occupancy is set by the band's LDS here, not by
// registers, and an item then issues 9 x CB LDS reads instead of 18 x CB (the 32-channel BatchNorm-input variant has no room).
// WIN: a wavefront walks its items DOWN a column of segments (items numbered segment-major) and keeps the fragments of the
// last two input rows in registers -- 3 x CB fragment reads per item instead of 9 x CB (the ablation build without
// fragment reads ran 32 -> 21 us: the LDS reads, issue + latency, were the largest single term).  The window is 36 registers
// per 16 input channels: the 16-channel-input variants have room, and the 32-channel ones without statistics.
// BRED (the data gradient of layer L as the producer of layer L - 1's BatchNorm backward; plain input, no statistics): the
// first pass of that backward -- sums over the batch of gh = g * act'(z) and gh * xhat (kp_bn_act_bwd_reduce) -- runs in the
// epilogue, on the stored (rounded) gradient values.  The output of this kernel is the gradient in the PADDED domain and the
// separate pass folds it first; folding is the adjoint of the replicate pad, so sum_ij fold(g)_ij f_ij = sum_pq g_pq f[clamp(p - 1)]
// [clamp(q - 1)]: every padded-domain element is multiplied with the factors of its clamped source pixel, no fold.  `yred` =
// layer L - 1's conv output [N][Ho - 2][Wo - 2][K], `bn` ITS BatchNorm / activation; a lane's 16-byte chunk of yred is requested
// together with the MFMAs of its item and used one item later; `partials`: one row [2][K] per wavefront of the grid, as for
// the forward statistics (kp_bn_bwd_finalize takes them).  Same sums as the separate pass up to the order of the fp32 additions.
template <int CB, int KB, bool BNIN, bool STATS, bool WREG = !(BNIN && CB == 2), bool WIN = (CB == 1), bool BRED = false>
__global__ __launch_bounds__(FTPB, FTPB / 128) void conv3x3_small_band_k(const h16_t* __restrict__ x, const h16_t* __restrict__ w,
                                                            const float* __restrict__ bias, h16_t* __restrict__ y,
                                                            float* __restrict__ partials, int Hi, int Wi, int R,
                                                            int bands_per_img, int total_bands, BnIn bn,
                                                            const h16_t* __restrict__ yred = nullptr, KpPair pair = KpPair{}) {
    static_assert(!BRED || (!BNIN && !STATS && !WIN), "BRED: data-gradient form (plain input, no statistics, item-strided loop)");
    // pair.hb != 0 (kp_conv_fwd_pair / kp_conv_fwd_bnin_pair): the grid is TWO independent launches of hb workgroups each, one per
    // half-batch -- its own input, output, statistics rows and BatchNorm-input constants ([2][C] mean / invstd); total_bands is a half's
    int bid = blockIdx.x, nblk = gridDim.x;
    if (pair.hb) {
        const int hf = bid >= pair.hb ? 1 : 0;
        bid -= hf * pair.hb; nblk = pair.hb;
        x += hf * pair.x; y += hf * pair.y;
        if (STATS) partials += hf * pair.p;
        if (BNIN && hf && bn.mean) { bn.mean += CB * 16; bn.invstd += CB * 16; }
    }
    constexpr int C = CB * 16, K = KB * 16, PITCH = C * 2 + 16, CPP = C / 8;   // chunks of 16 bytes per pixel
    constexpr int JH = KB * 4;                                                   // register pairs of the accumulator that hold channels < K
    __shared__ h16x8 swf[WREG ? 1 : 9 * CB][64];
    __shared__ __attribute__((aligned(16))) unsigned char xl[FB_LDS];
    __shared__ __attribute__((aligned(16))) float cst[BNIN ? 4 : 1][32];
#ifdef KP_FB_TRACE
    const unsigned long long t_entry = wall_clock64();
#endif
    if (BNIN) bnin_fill(cst, bn, C, threadIdx.x);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const int Ho = Hi - 2, Wo = Wi - 2;
    const int nseg = (Wo + 31) / 32, XPW = nseg * 32 + 2, xrow_l = XPW * PITCH;
    // ---- staging tables of this thread's FSL chunks (band-independent).  A wave instruction covers 64 consecutive 16-byte chunks of
    // the band (1 KB of the source, coalesced) whichever lane takes which; the lanes of a 16-lane group take the SAME channel chunk c of
    // 16 consecutive pixels, so that their LDS stores are PITCH apart -- the conflict-free pattern of the fragment reads.  (With
    // chunk id = tid the 16 lanes of a group wrote (pixel, c) pairs 16 bytes apart inside a pixel and PITCH - 16 * (CPP - 1) apart
    // between pixels: two-way conflicts on a fifth of the banks, 18-25 % of the kernel's LDS cycles, profiles/r03m_pmc_summary.json.)
#ifndef KP_FB_STAGE_LINEAR
    const int c = ((tid & 63) >> 4) % CPP;
#else
    const int c = tid % CPP;
#endif
    int lo[FSL], so[FSL], rw[FSL];   // LDS byte offset; source byte offset (BNIN: of the clamped column only); band row (big: never staged)
#pragma unroll
    for (int u = 0; u < FSL; ++u) {
#ifndef KP_FB_STAGE_LINEAR
        // block of 64 chunks = 64 / CPP pixels: lane group g = lane / 16, j = lane % 16 -> pixel 16 * (g / CPP) + j of the block, chunk g % CPP
        const int pix = ((tid & ~63) + u * FTPB) / CPP + (((tid & 63) >> 4) / CPP) * 16 + (tid & 15);
#else
        const int pix = (tid + u * FTPB) / CPP;
#endif
        const int r = pix / XPW, px = pix - r * XPW;
        lo[u] = r * xrow_l + px * PITCH + c * 16;
        rw[u] = px < Wi ? r : (r | 0x4000);   // bit 14: zero fill beyond the image row
        if (BNIN) {
            const int sc = px - 1 < 0 ? 0 : (px - 1 > Wo - 1 ? Wo - 1 : px - 1);   // ReplicationPad2d(1): column clamp
            so[u] = (sc * CPP + c) * 16;
        } else {
            so[u] = ((r * Wi + px) * CPP + c) * 16;
        }
    }
    const int lane_ld = fr * PITCH + fh * 16;          // fragment read: lane part of the LDS address
    // output store, lane part of the byte offset.  16 channels: after the half-wave swaps a lane holds chunk fh (8 channels)
    // of pixel fr -- the wave's 32 pixels x 32 bytes are one contiguous KB.  32 channels: a second exchange between the
    // 16-lane rows leaves a lane with chunk 2 * (row & 1) + (row >> 1) of pixel (lane & 15) [first store] and of pixel
    // 16 + (lane & 15) [second store], so either store writes whole 64-byte pixels, 1 KB contiguous (two stores of 32 bytes
    // per 64-byte pixel each cost the write path twice the requests: the stores were 17 of the kernel's 53 us)
    const int lrow = lane >> 4, px16 = lane & 15;
    const int lane_st = KB == 1 ? fr * 32 + fh * 16 : px16 * 64 + (2 * (lrow & 1) + (lrow >> 1)) * 16;
    constexpr int OOB = (int)0x80000000;               // beyond every band's bytes: the buffer store drops the lane
    // BRED: the lane's stored chunk (8 channels of a pixel) is chunk fh (16 channels) / 2 * (row & 1) + (row >> 1) (32) of either store
    const int rchunk = KB == 1 ? fh : 2 * (lrow & 1) + (lrow >> 1);
    KpBnIn8 rk;
    float rsb[BRED ? 8 : 1], rsg[BRED ? 8 : 1];
    if constexpr (BRED) {
        rk.load(bn, rchunk * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) rsb[e] = rsg[e] = 0.f;
    }

    // ---- staging: the band's R + 2 input rows, global -> registers -> LDS ------------------------------------------------
    // All of a band's loads are in flight before the first LDS store.  Buffer loads: a chunk outside the band or the image
    // row carries an out-of-range offset and comes back as zeros -- no branch per chunk.  (Issuing band b + 1's loads
    // before the items of band b and parking them in registers -- 48 -- was measured twice, with the old and with this
    // item loop: no faster, 31.3 vs 31.9 us, and the 32-channel variants spill.  The clock stamps of
    // tools/probes/small_trace.py say why: in steady state a band takes 4.8 us per workgroup = ~6 TB/s over the chip;
    // what the kernel loses it loses at its ends -- 1.3 us of prologue, 1.5 us until the first band has landed, 4.5 us
    // between the mean and the last workgroup's end, 4.5 us of launch and teardown around a 22 us body.)
    uint4 sb[FSL];
    unsigned live = 0;   // BNIN: chunks that hold image data (the zero fill must stay zero behind the transform)
    auto issue = [&](int band) {
        const int n = band / bands_per_img;
        const int p0 = (band - n * bands_per_img) * R;
        const int rows = Ho - p0 < R ? Ho - p0 : R;
        const unsigned char* xg = BNIN ? (const unsigned char*)x + (((long long)n * Ho * Wo) * (long long)C) * 2
                                       : (const unsigned char*)x + ((((long long)n * Hi + p0) * Wi) * (long long)C) * 2;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xg, 0, (BNIN ? Ho * Wo : (rows + 2) * Wi) * C * 2, 0x00020000);
        live = 0;
#pragma unroll
        for (int u = 0; u < FSL; ++u) {
            const bool on = rw[u] < rows + 2 && !(KP_FB_ABL & 1);   // (rows beyond the band and the zero fill have a large rw)
            int off = so[u];
            if (BNIN) {   // row clamp of the replicate pad: padded row p0 + r <- clamp(p0 + r - 1) of the unpadded map
                const int pr = p0 + (rw[u] & 0x3fff) - 1;
                off += (pr < 0 ? 0 : (pr > Ho - 1 ? Ho - 1 : pr)) * (Wo * C * 2);
                live |= on ? 1u << u : 0u;
            }
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(xr, on ? off : OOB, 0, 0);
            sb[u] = make_uint4(t[0], t[1], t[2], t[3]);
        }
    };
    auto commit = [&]() {
        if (BNIN) {
            BnReg k;
            k.load(cst, c * 8);
#pragma unroll
            for (int u = 0; u < FSL; ++u) {
                const uint4 t = bnin_apply(sb[u], k, bn.slope);
                const bool on = live & (1u << u);
                sb[u] = make_uint4(on ? t.x : 0u, on ? t.y : 0u, on ? t.z : 0u, on ? t.w : 0u);
            }
        }
#pragma unroll
        for (int u = 0; u < FSL; ++u)
            if ((rw[u] & 0x3fff) < R + 2) *reinterpret_cast<uint4*>(xl + lo[u]) = sb[u];   // (rows past a short last band: zeros, never read)
    };

#ifdef KP_FB_TRACE   // probe builds (tools/probes/small_trace.py): clock stamps of wave 0 behind the statistics rows
    unsigned long long* const trc = STATS ? (unsigned long long*)(partials + 8192 * 2 * K) + (long long)blockIdx.x * 32 : nullptr;
    int trn = 0;
    auto stamp = [&]() { if (STATS && tid == 0 && trn < 32) trc[trn++] = wall_clock64(); };
    if (STATS && tid == 0) trc[trn++] = t_entry;
    if (STATS && tid == 0) { unsigned id; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id)); unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); trc[trn++] = ((unsigned long long)xcc << 32) | id; }
    stamp();
#else
    auto stamp = [&]() {};
#endif
    // (the weights and the bias -- cold loads -- are requested here, behind the address tables, and first used behind the first
    // band's loads: one round trip for both)
    h16x8 wreg[WREG ? 9 * CB : 1];
    if (WREG || wave == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                h16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (fr < K) v = *reinterpret_cast<const h16x8*>(w + ((long long)fr * 9 + t) * C + cb * 16 + fh * 8);
                if (WREG) wreg[t * CB + cb] = v;
                else swf[t * CB + cb][lane] = v;
            }
    }
    // D[k][px]: lane: pixel = fr, k = 8 * (reg >> 2) + 4 * fh + (reg & 3); register PAIRS (2j, 2j + 1) are the packed operands
    f32x2 bv[JH], ssum[STATS ? JH : 1], ssq[STATS ? JH : 1];
#pragma unroll
    for (int j = 0; j < JH; ++j) {
        const int k = 8 * (j >> 1) + 4 * fh + 2 * (j & 1);
        bv[j] = f32x2{bias ? bias[k] : 0.f, bias ? bias[k + 1] : 0.f};
        if (STATS) ssum[j] = ssq[j] = f32x2{0.f, 0.f};
    }
    for (int band = bid; band < total_bands; band += nblk) {
        const int n = band / bands_per_img;
        const int p0 = (band - n * bands_per_img) * R;
        const int rows = Ho - p0 < R ? Ho - p0 : R;
        stamp();
        __syncthreads();   // the previous band's fragments have been read (first trip: the weights are in place after the next one)
        stamp();
        issue(band);
        commit();
        __syncthreads();
        stamp();
        // the band's output rows as one buffer: stores carry a lane-constant offset (or OOB for pixels beyond the row: no
        // branch around the store) and the item's scalar offset
        unsigned char* const yband = (unsigned char*)y + (((long long)n * Ho + p0) * Wo) * (long long)K * 2;   // uniform
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)yband, 0, rows * Wo * K * 2, 0x00020000);
        const int nitems = (KP_FB_ABL & 2) ? 0 : rows * nseg;
        // BRED: the clamped source pixels of this band's items in layer L - 1's conv output (image n)
        const int Hr = Ho - 2, Wr = Wo - 2;
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
            BRED ? (void*)((const unsigned char*)yred + ((long long)n * Hr * Wr) * (long long)K * 2) : (void*)y, 0, BRED ? Hr * Wr * K * 2 : 0, 0x00020000);
        struct YPair { u32x4 a, b; };
        auto yload = [&](int item) {
            YPair q{};
            if constexpr (BRED) {
                const int r = item / nseg, sq = item - r * nseg;
                const int pr = p0 + r - 1;
                const int rowoff = (pr < 0 ? 0 : (pr > Hr - 1 ? Hr - 1 : pr)) * Wr;      // scalar
                const int col = sq * 32 + (KB == 1 ? fr : px16);
                const int ca = col - 1 < 0 ? 0 : (col - 1 > Wr - 1 ? Wr - 1 : col - 1);
                q.a = __builtin_amdgcn_raw_buffer_load_b128(rr, col < Wo ? ((rowoff + ca) * K + rchunk * 8) * 2 : OOB, 0, 0);
                if constexpr (KB == 2) {
                    const int cb2 = col + 15 > Wr - 1 ? Wr - 1 : col + 15;               // (col + 16) - 1 >= 0
                    q.b = __builtin_amdgcn_raw_buffer_load_b128(rr, col + 16 < Wo ? ((rowoff + cb2) * K + rchunk * 8) * 2 : OOB, 0, 0);
                }
            }
            return q;
        };
        // gh = g * act'(z), sums of gh and gh * xhat over the lane's pixels; g = the stored chunk, lanes beyond the row: g = 0
        auto bred = [&](const u32x4& gw, const u32x4& yw, bool on) {
            float g[8], yv[8];
            unpack_h16x2(gw[0], g[0], g[1]); unpack_h16x2(gw[1], g[2], g[3]); unpack_h16x2(gw[2], g[4], g[5]); unpack_h16x2(gw[3], g[6], g[7]);
            unpack_h16x2(yw[0], yv[0], yv[1]); unpack_h16x2(yw[1], yv[2], yv[3]); unpack_h16x2(yw[2], yv[4], yv[5]); unpack_h16x2(yw[3], yv[6], yv[7]);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xh = (yv[e] - rk.mu[e]) * rk.is[e];
                const float z = xh * rk.ga[e] + rk.be[e];
                const float gh = (on ? g[e] : 0.f) * act_grad(z, bn.slope);
                rsb[e] += gh;
                rsg[e] += gh * xh;
            }
        };
        // one item = 32 pixels of one row x K channels: 9 x CB fragment reads, 9 x CB MFMAs into one accumulator tile
        auto mma = [&](int item) {                             // item -> row, segment, LDS base: scalar
            const int r = item / nseg, sq = item - r * nseg;
            const unsigned char* xb = xl + (r * xrow_l + sq * 32 * PITCH) + lane_ld;
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) {
                    const h16x8 xf = (KP_FB_ABL & 16) ? h16x8{(short)item, 0, 0, 0, 0, 0, 0, 0}
                                                      : *reinterpret_cast<const h16x8*>(xb + (t / 3) * xrow_l + (t % 3) * PITCH + cb * 32);
                    if (KP_FB_ABL & 8) acc[(t * CB + cb) & 15] += (float)xf[0];
                    else acc = kp_mfma16(WREG ? wreg[t * CB + cb] : swf[t * CB + cb][lane], xf, acc, 0, 0, 0);
                }
            return acc;
        };
        // bias, statistics, 16-bit pack, lane exchange, store: straight-line code (no branch: the scheduler is free to put it
        // under the MFMAs of the NEXT item, see the loop below)
        auto epi_rest = [&](int r, int sq, const f32x2 (&v)[JH], const YPair& yq = YPair{}) {
            const int valid = Wo - sq * 32;                     // pixels of the segment inside the row (>= 32: all)
            if constexpr (STATS) {
                // lanes beyond the row: x 0 (their fragments are the zero fill and halo, finite); x 1 is exact
                const float okf = fr < valid ? 1.f : 0.f;
                f32x2 ok2{okf, okf};
                asm("" : "+v"(ok2));   // a real register pair: packed multiplies below
#pragma unroll
                for (int j = 0; j < JH; ++j) {
                    const f32x2 m = v[j] * ok2;
                    ssum[j] += m; ssq[j] = __builtin_elementwise_fma(m, m, ssq[j]);
                }
            }
            uint32_t p[JH];
#pragma unroll
            for (int j = 0; j < JH; ++j) p[j] = pack_h16x2(v[j][0], v[j][1]);
            // pair the half-waves' 8-byte pieces: lanes fh = 0 then hold channels 8g .. 8g + 7 (chunk g, even g), lanes fh = 1 chunk g + 1
#pragma unroll
            for (int g = 0; g < JH; g += 4) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(p[g], p[g + 2], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(p[g + 1], p[g + 3], false, false);
                p[g] = s0[0]; p[g + 2] = s0[1]; p[g + 1] = s1[0]; p[g + 3] = s1[1];
            }
            const int soff = (r * Wo + sq * 32) * K * 2;
            if constexpr (KB == 1) {
                const int vo = (fr < valid && !(KP_FB_ABL & 4)) ? lane_st : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{p[0], p[1], p[2], p[3]}, yr, vo, soff, 0);
                if constexpr (BRED) bred(u32x4{p[0], p[1], p[2], p[3]}, yq.a, fr < valid);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {                   // 16-lane rows: (p[i], p[4 + i]) -> (pixels 0-15, pixels 16-31)
                    const auto s = __builtin_amdgcn_permlane16_swap(p[i], p[4 + i], false, false);
                    p[i] = s[0]; p[4 + i] = s[1];
                }
                const int va = (px16 < valid && !(KP_FB_ABL & 4)) ? lane_st : OOB;
                const int vb = (px16 + 16 < valid && !(KP_FB_ABL & 4)) ? lane_st + 1024 : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{p[0], p[1], p[2], p[3]}, yr, va, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{p[4], p[5], p[6], p[7]}, yr, vb, soff, 0);
                if constexpr (BRED) {
                    bred(u32x4{p[0], p[1], p[2], p[3]}, yq.a, px16 < valid);
                    bred(u32x4{p[4], p[5], p[6], p[7]}, yq.b, px16 + 16 < valid);
                }
            }
        };
        auto epi = [&](int item, const f32x16& acc, const YPair& yq = YPair{}) {
            const int r = item / nseg, sq = item - r * nseg;
            f32x2 v[JH];
#pragma unroll
            for (int j = 0; j < JH; ++j) v[j] = f32x2{acc[2 * j], acc[2 * j + 1]} + bv[j];
            epi_rest(r, sq, v, yq);
        };
        // a wavefront's items wave, wave + FNW, ...: the MFMAs of item i + 1 are issued BEFORE the epilogue of item i, in
        // one basic block -- a lone wave otherwise runs a 9-deep dependent MFMA chain and then ~60 dependent VALU
        // instructions back to back (~1 200 cycles per item, with two waves per SIMD to hide it)
        // (two named accumulators and the loop unrolled by two: no register copies between them -- where the registers
        // allow: the 32-channel variants with statistics would spill 4-25 and keep the rolled loop and its 8 copies)
        if constexpr (WIN) {
            // items segment-major: idx = sq * rows + r; the wavefront's contiguous quarter of them
            const int beg = nitems * wave / FNW, end = nitems * (wave + 1) / FNW;
            if (beg < end) {
                int sq = beg / rows, r = beg - sq * rows, idx = beg;
                h16x8 w0[3 * CB], w1[3 * CB], w2[3 * CB];       // the window: three input rows x 3 taps x CB
                auto row_load = [&](h16x8 (&d)[3 * CB], int prow) {
                    const unsigned char* xb = xl + (prow * xrow_l + sq * 32 * PITCH) + lane_ld;
#pragma unroll
                    for (int t = 0; t < 3; ++t)
#pragma unroll
                        for (int cb = 0; cb < CB; ++cb)
                            d[t * CB + cb] = (KP_FB_ABL & 16) ? h16x8{(short)prow, 0, 0, 0, 0, 0, 0, 0}
                                                              : *reinterpret_cast<const h16x8*>(xb + t * PITCH + cb * 32);
                };
                f32x2 vp[JH];                                   // the pending item: accumulators + bias, its row and segment
                int rp = 0, sqp = 0;
                // one step: read the new bottom row, multiply (taps in the order of the rolled kernel: bit-identical
                // outputs), finish the PENDING item under those MFMAs, then take over this item's accumulators
                auto step = [&](const h16x8 (&a)[3 * CB], const h16x8 (&b)[3 * CB], h16x8 (&cn)[3 * CB], auto first) {
                    row_load(cn, r + 2);
                    f32x16 acc;
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
                    for (int t = 0; t < 9 * CB; ++t) {
                        const h16x8& xf = t < 3 * CB ? a[t] : (t < 6 * CB ? b[t - 3 * CB] : cn[t - 6 * CB]);
                        if (KP_FB_ABL & 8) acc[t & 15] += (float)xf[0];
                        else acc = kp_mfma16(wreg[t], xf, acc, 0, 0, 0);
                    }
                    if constexpr (!decltype(first)::value) epi_rest(rp, sqp, vp);
#pragma unroll
                    for (int j = 0; j < JH; ++j) vp[j] = f32x2{acc[2 * j], acc[2 * j + 1]} + bv[j];
                    rp = r; sqp = sq;
                    ++idx; ++r;
                };
                // the three roles of (w0, w1, w2) as straight-line code, each step in a block of its own (a common tail
                // would take the epilogue out from under the MFMAs); at the bottom of a column: next segment, refill
                auto wrap = [&]() {
                    if (r < rows) return false;
                    r = 0; ++sq;
                    row_load(w0, 0); row_load(w1, 1);
                    return true;
                };
                row_load(w0, r); row_load(w1, r + 1);
                step(w0, w1, w2, std::true_type{});
                while (idx < end) {
                    if (!wrap()) {
                        step(w1, w2, w0, std::false_type{});
                        if (idx >= end) break;
                        if (!wrap()) {
                            step(w2, w0, w1, std::false_type{});
                            if (idx >= end) break;
                            wrap();
                        }
                    }
                    // here the two newest rows are (w0, w1) -- after a refill, or after three steps
                    if (idx >= end) break;
                    step(w0, w1, w2, std::false_type{});
                }
                epi_rest(rp, sqp, vp);
            }
        } else if (wave < nitems) {
            int item = wave;
            f32x16 a0 = mma(item);
            if constexpr ((STATS || BRED) && KB == 2) {
                YPair y0 = yload(item);
                for (; item + FNW < nitems; item += FNW) {
                    const f32x16 nxt = mma(item + FNW);
                    const YPair yn = yload(item + FNW);
                    epi(item, a0, y0);
                    a0 = nxt; y0 = yn;
                }
                epi(item, a0, y0);
            } else {
                f32x16 a1;
                YPair y0 = yload(item), y1{};
                for (;;) {
                    if (item + FNW >= nitems) { epi(item, a0, y0); break; }
                    a1 = mma(item + FNW); y1 = yload(item + FNW); epi(item, a0, y0); item += FNW;
                    if (item + FNW >= nitems) { epi(item, a1, y1); break; }
                    a0 = mma(item + FNW); y0 = yload(item + FNW); epi(item, a1, y1); item += FNW;
                }
            }
        }
    }
    stamp();
    // FROWS == 1: the wavefronts' rows go to LDS (the band buffers, once every wavefront has left them) and are summed below
    float* const red = reinterpret_cast<float*>(xl);
    if constexpr ((STATS || BRED) && FROWS == 1) __syncthreads();
    float* const rowp = FROWS == 1 ? red + wave * 2 * K : partials + ((long long)bid * FNW + wave) * 2 * K;
    if constexpr (BRED) {
        // lanes that hold the same chunk: the 32 of a half-wave (16 channels) / the 16 of a row (32 channels)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = 1; o < (KB == 1 ? 32 : 16); o <<= 1) {
                rsb[e] += __shfl_xor(rsb[e], o, 64);
                rsg[e] += __shfl_xor(rsg[e], o, 64);
            }
        }
        if ((lane & (KB == 1 ? 31 : 15)) == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                rowp[0 * K + rchunk * 8 + e] = rsb[e];
                rowp[1 * K + rchunk * 8 + e] = rsg[e];
            }
        }
    }
    if constexpr (STATS) {
        float s1[16], s2[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool on = j < JH;
            s1[2 * j] = on ? ssum[j % JH][0] : 0.f; s1[2 * j + 1] = on ? ssum[j % JH][1] : 0.f;
            s2[2 * j] = on ? ssq[j % JH][0] : 0.f; s2[2 * j + 1] = on ? ssq[j % JH][1] : 0.f;
        }
        int r1, r2;
        const float ts = reduce16_over32(s1, lane, r1);
        const float tq = reduce16_over32(s2, lane, r2);
        const int k = 8 * (r1 >> 2) + 4 * fh + (r1 & 3);
        if ((lane & 1) == 0 && k < K) {
            rowp[0 * K + k] = ts;
            rowp[1 * K + k] = tq;
        }
    }
    if constexpr ((STATS || BRED) && FROWS == 1) {
        __syncthreads();
        for (int i = tid; i < 2 * K; i += FTPB) {
            float t = red[i];
#pragma unroll
            for (int wv = 1; wv < FNW; ++wv) t += red[wv * 2 * K + i];
            partials[(long long)bid * 2 * K + i] = t;
        }
    }
}

// scratch[k][tap][c] += sum_pixels dy[pixel][k] * x[pixel + tap][c] on the matrix cores, no LDS staging:
// D[k][c] (one 32x32 tile per tap, 9 accumulators) += A[k][16 px] B[16 px][c].  Both operands are pixel-major in HBM, so
// a lane gathers its fragment (8 consecutive pixels of ONE channel) with 2-byte loads at a channel stride -- 64-byte
// runs per pixel across the wave, L1-resident; the 10 loads of an input row serve its 3 column taps.
// A wavefront walks whole image rows in 16-pixel segments; tail pixels get dy = 0 (and a clamped, finite x).
template <int K, int C>
__global__ __launch_bounds__(TPB) void wgrad3x3_small_k(const h16_t* __restrict__ x, const h16_t* __restrict__ dy,
                                                           float* __restrict__ scratch, int Ho, int Wo, int dy_pad,
                                                           long long total_rows, int rows_per_wave, long long pstride) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 31, h = lane >> 5;
    const int Hi = Ho + 2, Wi = Wo + 2, Hd = Ho + 2 * dy_pad, Wd = Wo + 2 * dy_pad;
    const int kk = fr < K ? fr : K - 1, cc = fr < C ? fr : C - 1;   // lanes beyond the channel count: clamped, zeroed below
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const long long gw = (long long)blockIdx.x * 4 + wave;
    const long long r_beg = gw * rows_per_wave;
    const long long r_end = r_beg + rows_per_wave < total_rows ? r_beg + rows_per_wave : total_rows;
    for (long long row = r_beg; row < r_end; ++row) {
        const int p = (int)(row % Ho);
        const long long n = row / Ho;
        const unsigned short* dr = dy + (((n * Hd + p + dy_pad) * Wd) + dy_pad) * (long long)K + kk;
        const unsigned short* xr = x + ((n * Hi + p) * Wi) * (long long)C + cc;
        for (int q0 = 0; q0 < Wo; q0 += 16) {
            const int qb = q0 + 8 * h;
            unsigned a16[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = qb + j;
                const unsigned v = dr[(long long)(q < Wo ? q : Wo - 1) * K];
                a16[j] = (q < Wo && fr < K) ? v : 0u;
            }
            h16x8 af;
#pragma unroll
            for (int j = 0; j < 8; ++j) af[j] = (short)a16[j];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                unsigned b16[10];
#pragma unroll
                for (int j = 0; j < 10; ++j) {
                    const int col = qb + j;
                    const unsigned v = xr[((long long)r * Wi + (col < Wi ? col : Wi - 1)) * C];
                    b16[j] = fr < C ? v : 0u;
                }
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) {
                    h16x8 bfv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bfv[j] = (short)b16[j + s_];
                    acc[r * 3 + s_] = kp_mfma16(af, bfv, acc[r * 3 + s_], 0, 0, 0);
                }
            }
        }
    }
    // D[k][c]: lane col = c (fr), rows k = (i&3) + 8*(i>>2) + 4*h.  Sum the 4 wavefronts per tap through LDS; the workgroup's
    // sum is ITS partial gradient (plain stores into scratch[blockIdx.x]; kp_wgrad_unpack adds the partials in order).
    __shared__ float red[4][16][64];
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[t][i];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = (i & 3) + 8 * (i >> 2) + 4 * h;
                if (k < K && fr < C) {
                    const float s = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
                    scratch[(long long)blockIdx.x * pstride + ((long long)k * 9 + t) * C + fr] = s;
                }
            }
        }
    }
}

#ifdef KP_FB_TRACE   // probe builds (tools/probes/wgrad_trace.py): 100 MHz clock stamps of wave 0 of every workgroup of the LAST weight-gradient launch
static __device__ unsigned long long g_wg_trace[512 * 64];
#endif
// The same weight gradient with the operands staged through LDS.  The kernel above gathers every MFMA fragment with 2-byte
// GLOBAL loads (38 per 9 MFMAs, ~64 texture-unit cycles each: 169 us per 16 -> 32 layer of the Pong decoder at 256 x 84 x 84,
// where the data takes 50 us).  Here a workgroup copies a band of R image rows -- R + 2 input rows, R gradient rows, 16-byte
// loads, zero-filled to whole 16-pixel segments -- into LDS once, and its four wavefronts take the (row, segment) items of
// the band, every fragment = two hardware transpose reads (ds_read_b64_tr_b16) of the pixel-major rows: 20 LDS reads per
// 9 MFMAs and no packing arithmetic.  <= 76 KB per workgroup: two per CU, one stages while the other multiplies.  Same per-workgroup partial
// gradients, same order of the reductions inside a workgroup as above is NOT kept (items go round-robin over the waves): results
// differ from the kernel above in the last bits, and are bitwise repeatable.
constexpr int WG_LDS = 77824;
// 16-byte chunks per thread and tensor in one band: 12 at most (256 threads x 12 x 16 B = 48 KB of each); the narrower tensor of a (K, C) pair never gets its half of the LDS, and every chunk is 4 registers that live
// across the multiply phase (the 32 x 16 variant spilled with 12 + 12)
constexpr int sl_x(int K, int C) { return C == 32 ? 12 : (K == 32 ? 8 : 10); }
constexpr int sl_d(int K, int C) { return K == 32 ? 12 : (C == 32 ? 8 : 10); }
// BNIN: x is the previous layer's conv output [N][Ho][Wo][C]; its BatchNorm-apply + activation + replicate pad run between the
// global load and the LDS store (BnIn above), so the weight gradient sees the very fragments the forward multiplied.
// Occupancy and pipeline.  The 9 x 16 accumulators, the fragments and the staging registers take 320-470 registers: ONE wave
// per SIMD, i.e. one workgroup per CU whatever the LDS admits (capping the allocation at 256 spills 60-300 registers and is
// 1.5-2x slower, measured).  So nothing else on the CU can cover a workgroup's staging latency -- the kernel has to do it
// itself: the band is double buffered in LDS (2 x 76 KB), and ALL global loads of band b + 1 (both tensors, 96 registers:
// there is room, the file is this wave's alone) are issued before the multiply phase of band b and written to the other
// buffer behind it; one barrier per band.  One workgroup per CU in the grid, each walking bands blockIdx, + grid, ...
template <int K, int C, bool BNIN>
__global__ __launch_bounds__(TPB, 1) void wgrad3x3_small_band_k(const h16_t* __restrict__ x, const h16_t* __restrict__ dy,
                                                             float* __restrict__ scratch, int Ho, int Wo, int dy_pad, int R,
                                                             int bands_per_img, long long total_bands, long long pstride, BnIn bn) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * WG_LDS];
    __shared__ __attribute__((aligned(16))) float cst[BNIN ? 4 : 1][32];
    if (BNIN) bnin_fill(cst, bn, C, threadIdx.x);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, h = lane >> 5;
    const int Hi = Ho + 2, Wi = Wo + 2, Hd = Ho + 2 * dy_pad, Wd = Wo + 2 * dy_pad;
    const int Wp = (Wo + 15) & ~15, XP = Wp + 2, nseg = Wp / 16;
    const int xrow_b = XP * C * 2, drow_b = Wp * K * 2;            // LDS bytes per staged row (multiples of 16)
    const int dl_off = (R + 2) * xrow_b;                           // a buffer = [(R + 2)][XP][C] input rows, then [R][Wp][K] gradient rows
    // transpose-read lane roles: t = lane & 15 -> (pixel t / 4, chunk t % 4); g = bit 4 -> 16-channel half; lane / 32 -> pixels 8..15
    const int t16 = lane & 15, g16 = (lane >> 4) & 1;
    const int trow = 8 * h + (t16 >> 2);
    const int chunkb = (16 * g16 + (t16 & 3) * 4) * 2;
    const bool kval = 16 * g16 < K, cval = 16 * g16 < C;      // the upper half does not exist in a 16-channel tensor: zero fragment
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // ---- staging: input rows p0 .. p0 + rows + 1 (Wi pixels each, zero-filled to XP), gradient rows (Wo pixels, zero-filled to
    //      Wp); chunk -> (row, column) by carries, no divisions.  issue(): every 16-byte load of a band in flight at once;
    //      commit(): (BNIN: BatchNorm + activation of the input chunks,) LDS stores.
    const int xc = xrow_b / 16, xvalid = Wi * C * 2 / 16, dc = drow_b / 16, dvalid = Wo * K * 2 / 16;
    const int xrow_g = Wi * C * 2, drow_g = Wd * K * 2;
    const int xdr = TPB / xc, xdc = TPB % xc, ddr = TPB / dc, ddc = TPB % dc;
    constexpr int SLX = sl_x(K, C), SLD = sl_d(K, C);
    uint4 sbx[SLX], sbd[SLD];
    unsigned livex = 0;      // BNIN: input chunks that hold image data
    int rows_staged = 0;     // rows of the band the staging registers hold
    auto issue = [&](long long band) {
        const long long n = band / bands_per_img;
        const int p0 = (int)(band % bands_per_img) * R;
        const int rows = Ho - p0 < R ? Ho - p0 : R;
        rows_staged = rows;
        const int xtot = (rows + 2) * xc, dtot = rows * dc;
        const unsigned char* xg = BNIN ? (const unsigned char*)x + ((n * Ho * Wo) * (long long)C) * 2
                                       : (const unsigned char*)x + (((n * Hi + p0) * Wi) * (long long)C) * 2;
        const unsigned char* dg = (const unsigned char*)dy + ((((n * Hd + p0 + dy_pad) * Wd) + dy_pad) * (long long)K) * 2;
        livex = 0;
        int r = tid / xc, c16 = tid % xc;
#pragma unroll
        for (int u = 0; u < SLX; ++u) {
            sbx[u] = make_uint4(0, 0, 0, 0);
            if (tid + u * TPB < xtot && c16 < xvalid) {
                if (BNIN) {   // chunk c16 of padded row p0 + r = (pixel c16 / CPP, channels 8 * (c16 % CPP) ..) <- clamped source
                    constexpr int CPP = C / 8;
                    const int px = c16 / CPP, cg = c16 % CPP;
                    const int sr = p0 + r - 1 < 0 ? 0 : (p0 + r - 1 > Ho - 1 ? Ho - 1 : p0 + r - 1);
                    const int sc = px - 1 < 0 ? 0 : (px - 1 > Wo - 1 ? Wo - 1 : px - 1);
                    sbx[u] = *reinterpret_cast<const uint4*>(xg + (((long long)sr * Wo + sc) * CPP + cg) * 16);
                    livex |= 1u << u;
                } else {
                    sbx[u] = *reinterpret_cast<const uint4*>(xg + (long long)r * xrow_g + c16 * 16);
                }
            }
            r += xdr; c16 += xdc;
            if (c16 >= xc) { c16 -= xc; ++r; }
        }
        int r2 = tid / dc, d16 = tid % dc;
#pragma unroll
        for (int u = 0; u < SLD; ++u) {
            sbd[u] = make_uint4(0, 0, 0, 0);
            if (tid + u * TPB < dtot && d16 < dvalid) sbd[u] = *reinterpret_cast<const uint4*>(dg + (long long)r2 * drow_g + d16 * 16);
            r2 += ddr; d16 += ddc;
            if (d16 >= dc) { d16 -= dc; ++r2; }
        }
    };
    auto commit = [&](unsigned char* buf) {
        const int xtot = (rows_staged + 2) * xc, dtot = rows_staged * dc;
        if (BNIN) {
            BnReg k;
            k.load(cst, (tid % (C / 8)) * 8);   // TPB and xc are multiples of C / 8: a thread's chunks all hold the same channels
#pragma unroll
            for (int u = 0; u < SLX; ++u) {
                const uint4 t = bnin_apply(sbx[u], k, bn.slope);
                const bool on = livex & (1u << u);
                sbx[u] = make_uint4(on ? t.x : 0u, on ? t.y : 0u, on ? t.z : 0u, on ? t.w : 0u);
            }
        }
#pragma unroll
        for (int u = 0; u < SLX; ++u)
            if (tid + u * TPB < xtot) *reinterpret_cast<uint4*>(buf + (tid + u * TPB) * 16) = sbx[u];
#pragma unroll
        for (int u = 0; u < SLD; ++u)
            if (tid + u * TPB < dtot) *reinterpret_cast<uint4*>(buf + dl_off + (tid + u * TPB) * 16) = sbd[u];
    };

#ifdef KP_FB_TRACE
    int trn = 0;
    auto stamp = [&]() { if (tid == 0 && trn < 64 && blockIdx.x < 512) g_wg_trace[blockIdx.x * 64 + trn++] = wall_clock64(); };
    if (tid == 0 && blockIdx.x < 512) for (int i = 0; i < 64; ++i) g_wg_trace[blockIdx.x * 64 + i] = 0;
#else
    auto stamp = [&]() {};
#endif
    // (Clock stamps of the 16 -> 32 layer at 256 x 84 x 84, 12 bands per workgroup, 67 us: issue 1.0 us per band -- a CU's
    //  vector-memory path takes a 1 KB wave-load per ~16 cycles and the 20-24 loads x 4 waves arrive in one burst --, multiply
    //  3.4, commit 0.8.  Spreading the loads over the items of the multiply phase was tried twice: through a switch on a slot
    //  counter the compiler must assume a register is reloaded while its previous load is in flight and waits vmcnt(0) per slot,
    //  111 us; reading the next item's fragments ahead of the MFMAs (a four-row window ring, no copies, unconditional reads
    //  through a zero pad so that the waits sit at the MFMAs): 16-channel inputs 66 -> 62 us, 32-channel ones 72 -> 81: the
    //  multiply phase is bound by the rate of the transpose reads, not by their latency.  The weight gradient costs the
    //  config-2 step its full kernel time, side stream or not: 3.38 -> 2.85 ms with the launches skipped, tools/probes/ablate.sh.)
    stamp();
    long long band = blockIdx.x;
    __syncthreads();                     // (BNIN: the constants are in place)
    if (band < total_bands) { issue(band); stamp(); commit(smem); }
    __syncthreads();
    stamp();
    int cur = 0, rows = rows_staged;
    for (; band < total_bands; band += gridDim.x) {
        const bool more = band + gridDim.x < total_bands;
        if (more) issue(band + gridDim.x);                    // in flight during the multiply phase below
        stamp();
        const unsigned char* xl = smem + cur * WG_LDS;
        const unsigned char* dl = xl + dl_off;
        // ---- multiply: items (row, 16-pixel segment) round-robin over the four wavefronts.  A fragment (lane: channel
        //      lane % 32, 8 consecutive pixels 8 * (lane / 32) ..) is two ds_read_b64_tr_b16: per 16-lane group the
        //      instruction hands lane i the 4 rows of column i of a [4 pixels][16 channels] block, lane t supplying the
        //      address of (pixel t / 4, 4-channel chunk t % 4).  Pixel rows are C * 2 (K * 2) bytes apart: the four rows of
        //      a block fall on disjoint banks.
        // The multiply phase runs at the rate of ds_read_b64_tr_b16 (~16 LDS cycles per wave-instruction, one LDS per CU): an
        // item on its own is 2 (gradient) + 3 x 3 (input rows) = 11 reads for 9 MFMAs.  Vertically adjacent items share two of
        // their three input rows, so a wavefront takes a COLUMN STRIP of items (one 16-pixel segment, half of the band's rows)
        // and slides a three-row window of raw input dwords down it: 2 + 3 reads per item (+ 6 to fill the window).
        const int rh = (rows + 1) >> 1;                          // rows per half
        for (int unit = wave; unit < 2 * nseg; unit += 4) {      // scalar: strip and half
            const int sg = unit % nseg, half = unit / nseg;
            const int rbeg = half * rh, rend = rbeg + rh < rows ? rbeg + rh : rows;
            if (rbeg >= rend) continue;
            const int prow = sg * 16 + trow;
            // raw dwords of an input row at this strip: (p0,p1) (p2,p3) (p4,p5) (p6,p7) (p8,p9), p = pixel 8h + ..
            auto row_load = [&](int r, unsigned (&Rg)[5]) {
#pragma unroll
                for (int i = 0; i < 5; ++i) Rg[i] = 0u;
                if (cval) {
                    const unsigned char* b = xl + r * xrow_b + prow * (C * 2) + chunkb;
                    const uint2 q0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS h16x4*)(b)));
                    const uint2 q1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS h16x4*)(b + 4 * (C * 2))));
                    const uint2 q2 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS h16x4*)(b + 8 * (C * 2))));
                    Rg[0] = q0.x; Rg[1] = q0.y; Rg[2] = q1.x; Rg[3] = q1.y; Rg[4] = q2.x;
                }
            };
            // the three column taps of a row from its raw dwords: tap 2 is a renaming of the dword pairs, tap 1 four v_alignbit
            auto row_mma = [&](const h16x8& af, const unsigned (&Rg)[5], int rr) {
                const u32x4 f0 = {Rg[0], Rg[1], Rg[2], Rg[3]};
                const u32x4 f1 = {__builtin_amdgcn_alignbit(Rg[1], Rg[0], 16), __builtin_amdgcn_alignbit(Rg[2], Rg[1], 16),
                                  __builtin_amdgcn_alignbit(Rg[3], Rg[2], 16), __builtin_amdgcn_alignbit(Rg[4], Rg[3], 16)};
                const u32x4 f2 = {Rg[1], Rg[2], Rg[3], Rg[4]};
                acc[rr * 3 + 0] = kp_mfma16(af, __builtin_bit_cast(h16x8, f0), acc[rr * 3 + 0], 0, 0, 0);
                acc[rr * 3 + 1] = kp_mfma16(af, __builtin_bit_cast(h16x8, f1), acc[rr * 3 + 1], 0, 0, 0);
                acc[rr * 3 + 2] = kp_mfma16(af, __builtin_bit_cast(h16x8, f2), acc[rr * 3 + 2], 0, 0, 0);
            };
            unsigned W0[5], W1[5], W2[5];
            row_load(rbeg, W0);
            row_load(rbeg + 1, W1);
            for (int r = rbeg; r < rend; ++r) {
                h16x8 af = {0, 0, 0, 0, 0, 0, 0, 0};
                if (kval) {
                    const unsigned char* a = dl + r * drow_b + prow * (K * 2) + chunkb;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS h16x4*)(a + 4 * u * (K * 2)));
                        af[4 * u + 0] = v[0]; af[4 * u + 1] = v[1]; af[4 * u + 2] = v[2]; af[4 * u + 3] = v[3];
                    }
                }
                row_load(r + 2, W2);
                row_mma(af, W0, 0);
                row_mma(af, W1, 1);
                row_mma(af, W2, 2);
#pragma unroll
                for (int i = 0; i < 5; ++i) { W0[i] = W1[i]; W1[i] = W2[i]; }
            }
        }
        stamp();
        if (more) commit(smem + (cur ^ 1) * WG_LDS);          // the other buffer: its readers passed the previous barrier
        stamp();
        __syncthreads();
        stamp();
        cur ^= 1;
        rows = rows_staged;
    }
    // D[k][c]: lane col = c (fr), rows k = (i&3) + 8*(i>>2) + 4*h.  The four wavefronts are summed per tap through LDS (fixed
    // order); the workgroup's sum is ITS partial gradient (kp_wgrad_unpack adds the partials in order).
    float (*red)[16][64] = reinterpret_cast<float (*)[16][64]>(smem);
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave][i][lane] = acc[t][i];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int k = (i & 3) + 8 * (i >> 2) + 4 * h;
                if (k < K && fr < C) {
                    const float s = ((red[0][i][lane] + red[1][i][lane]) + red[2][i][lane]) + red[3][i][lane];
                    scratch[(long long)blockIdx.x * pstride + ((long long)k * 9 + t) * C + fr] = s;
                }
            }
        }
    }
}

// ---- weight gradient of the channel pairs with a 16-channel side, on the 16 x 16 x 32 matrix-core shape --------------------------
// wgrad3x3_small_band_k above multiplies 32 x 32 x 16 tiles: with 16 channels on either side half of every MFMA and of every
// transpose read is empty, its 9 x 16 accumulators hold it at ONE wave per SIMD and one workgroup per CU, and nothing covers
// its staging (clock stamps: 1.0 + 0.8 us of 5.3 us per band) -- the kernel costs the config-2 step its full 0.53 ms.  Here
// D[k][c] (16 x 16 per tile, 4 registers) += A[k][32 px] B[32 px][c]: an item is 32 pixels of a row, a lane's fragment 8
// consecutive pixels of one channel (two ds_read_b64_tr_b16 of the pixel-major rows, every lane useful); per item 3 (6) reads
// for the new bottom row of the input window + 2 per 16 gradient channels: 5-8 where the kernel above needs 10 per 32 pixels.
// 9 x KT x CT x 4 = 36-72 accumulator registers: three waves per SIMD, TWO workgroups of six wavefronts per CU -- one stages its
// band (single LDS buffer, loads straight through registers) while the other multiplies.  Six wavefronts: 84-pixel rows are three
// segments, x two halves of the band's rows = six column strips.  Same LDS image, staging code and partial-gradient protocol.
constexpr int W16_TPB = 384, W16_LDS = 77824;
#ifndef W16_SWZ
#define W16_SWZ 0   // 1: bank-swizzled rows.  Measured (round 4): the swizzle takes SQ_LDS_BANK_CONFLICT from 25-30 % of the LDS cycles to
                    // zero (tools/probes/small_pmc.sh) and the kernels do not move (49.5 -> 49.3, 66.1 -> 65.7 us: the transpose reads were
                    // never what the multiply phase waits for), while the BatchNorm-input variants -- at their register cap already --
                    // spill 4-22 registers for the extra lane constants: the config-2 step 2.83 -> 2.91 ms.  Off.
#endif
// chunk index c (16 bytes) within a staged row of CH-channel pixels -> its place under the bank swizzle (see the kernel): chunks of odd
// 8-pixel groups move by +-D chunks (16 channels: D = 8, the group's two 4-pixel halves swap; 32 channels: D = 2, a pixel's two
// 16-channel halves swap).  w16_swz_bits: bit 0 = moves, bit 1 = towards lower addresses; w16_delta: the move in chunks
template <int CH> __device__ __forceinline__ unsigned w16_swz_bits(int c) {
    if (!W16_SWZ) return 0u;
    const unsigned mv = CH == 16 ? (c >> 4) & 1 : (c >> 5) & 1, dn = CH == 16 ? (c >> 3) & 1 : (c >> 1) & 1;
    return mv | ((mv & dn) << 1);
}
template <int CH> __device__ __forceinline__ int w16_delta(unsigned bits) {
    constexpr int D = CH == 16 ? 8 : 2;
    return (bits & 1) ? ((bits & 2) ? -D : D) : 0;
}
// (32, 32): FOUR wavefronts per workgroup (two per SIMD with two workgroups per CU: 256 registers), each owning the two
// gradient-channel tiles of one input-channel half -- they share the input window: 3 + 4 transpose reads per item and wavefront
// W3 (the pairs with ONE 32-channel side): THREE wavefronts per workgroup (256 registers each), every one owning both tiles -- which share
// the input window (K = 32) or the gradient fragment (C = 32): 7 / 8 transpose reads per item instead of 10 -- and two of the six strips
constexpr int w16_waves(int K, int C, bool W3 = false) { return (K == 32 && C == 32) ? 4 : (W3 ? 3 : 6); }
constexpr int w16_slx(int K, int C, bool W3 = false) { return (K == 32 && C == 32) ? 11 : (W3 ? (C == 32 ? 17 : 10) : (C == 32 ? 9 : (K == 32 ? 6 : 7))); }
constexpr int w16_sld(int K, int C, bool W3 = false) { return (K == 32 && C == 32) ? 8 : (W3 ? (C == 32 ? 6 : 14) : (C == 32 ? 4 : (K == 32 ? 8 : 6))); }
typedef float f32x4v __attribute__((ext_vector_type(4)));
// CIN > 0 (C = 16, no BNIN): the FIRST layer of a Unit -- x is the padded input [N][Ho + 2][Wo + 2][CIN] with CIN <= 4 channels
// (grey / RGB frames); its pixels are widened to 16 zero-padded channels on the way into LDS and only the CIN real columns of the
// tiles are written: scratch [partials][K][9][CIN].  (The VALU kernel of kp_conv_first.hip walks image rows as a latency chain:
// 58 us for the 58 MB gradient of the Pong-class first layer, 1 TB/s.)
template <int K, int C, bool BNIN, int CIN = 0, bool W3 = false>
__global__ __launch_bounds__(w16_waves(K, C, W3) * 64, w16_waves(K, C, W3) == 6 ? 3 : 2) void wgrad3x3_small16_k(const h16_t* __restrict__ x, const h16_t* __restrict__ dy,
                                                               float* __restrict__ scratch, int Ho, int Wo, int dy_pad, int R,
                                                               int bands_per_img, long long total_bands, long long pstride, BnIn bn) {
    constexpr int NWV = w16_waves(K, C, W3), TPBW = NWV * 64, KT = K / 16, CT = C / 16;
    static_assert(!W3 || KT * CT == 2, "W3: exactly one 32-channel side");
    static_assert(CIN == 0 || (C == 16 && !BNIN && CIN <= 4), "CIN: first-layer form");
    __shared__ __attribute__((aligned(16))) unsigned char smem[W16_LDS];
    __shared__ __attribute__((aligned(16))) float cst[BNIN ? 4 : 1][32];
    if (BNIN) bnin_fill(cst, bn, C, threadIdx.x);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hi = Ho + 2, Wi = Wo + 2, Hd = Ho + 2 * dy_pad, Wd = Wo + 2 * dy_pad;
    const int Wp = (Wo + 31) & ~31, XP = Wp + 2, nseg = Wp / 32;
    const int xrow_b = XP * C * 2, drow_b = Wp * K * 2;            // LDS bytes per staged row (multiples of 16)
    const int dl_off = (R + 2) * xrow_b;                           // [(R + 2)][XP][C] input rows, then [R][Wp][K] gradient rows
    // transpose-read lane roles: t = lane & 15 -> (pixel t / 4, 4-channel chunk t % 4) of the lane group's 8-pixel run 8 * (lane >> 4)
    const int t16 = lane & 15, grp = lane >> 4;
    // LDS bank swizzle of the staged rows (W16_SWZ).  A transpose read takes 8 bytes per lane, 512 bytes per wavefront = two passes of
    // the 64 banks at best; with dense rows the four 16-lane groups start 8 pixels = 256 / 512 bytes apart and all land on the SAME 32
    // banks (four passes: 25-30 % of the kernel's LDS cycles were bank conflicts, tools/probes/small_pmc.sh).  So the rows are stored
    // with odd 8-pixel groups permuted -- 32-byte pixels (16 channels): their two 4-pixel halves swapped (byte bit 7); 64-byte pixels
    // (32 channels): the two 16-channel halves of every pixel swapped (byte bit 5) -- which puts odd groups on the other 32 banks.
    // w16_swz is the permutation on a row's 16-byte chunks (the staging stores), xoff / doff are the lanes' read offsets under it
    // (lane constants: the three reads of an input row take pixels lp, lp + 4, lp + 8 of the strip, lp + 8 is in the next group).
    const int lp = 8 * grp + (t16 >> 2), q8 = (t16 & 3) * 8, odd = W16_SWZ ? (grp & 1) : 0, nodd = W16_SWZ ? 1 - odd : 0;
    // ONE 16 x 16 tile per wavefront and tap (36 accumulator registers): with two tiles (32 channels on one side) the six wavefronts
    // are three pairs, wavefront w owns tile w % 2 for the whole kernel and takes two of the six column strips of a band -- both
    // tiles in one wavefront want 183-216 registers, and at the 168 of three waves per SIMD the spills made it slower than the old kernel
    constexpr int KW = (KT * CT == 4 || (W3 && KT == 2)) ? 2 : 1;   // gradient-channel tiles per wavefront (they share the input window)
    constexpr int CW = (W3 && CT == 2) ? 2 : 1;                      // input-channel tiles per wavefront (they share the gradient fragment)
    constexpr int TILES = KT * CT / (KW * CW);                       // tile owners among the wavefronts
    static_assert(TILES <= 2 && NWV % TILES == 0, "16 x 16 x 32 weight gradient: wavefronts pair up over at most two tile groups");
    const int tw = wave % TILES, kt = (KT == 2 && KW == 1) ? tw : 0, ct = (CT == 2 && CW == 1) ? tw : 0;
    // input row reads of tile jc: pixels lp, lp + 4, lp + 8 at xa(jc), xa(jc) + x1i, xb(jc) + x2i; gradient reads of tile j: da(j), da(j) + d1i.
    // 32-byte pixels: xa / xb = the lane's pixel with pixel bit 2 = odd / not odd (an odd group's halves are swapped), the third read's
    // group has the other parity: xb + 256.  64-byte pixels: xa / xb = the lane's pixel with the channel half (ct + jc) ^ odd / ^ !odd;
    // with both halves in one wavefront (CW == 2) tile 1's pair is tile 0's exchanged.  Two lane constants per tensor.
#if W16_SWZ
    const int xA = C == 16 ? (lp ^ (4 * odd)) * 32 + q8 : lp * 64 + q8 + (ct ^ odd) * 32;
    const int xB = C == 16 ? (lp ^ (4 * nodd)) * 32 + q8 : lp * 64 + q8 + (ct ^ nodd) * 32;
    const int dA = K == 16 ? (lp ^ (4 * odd)) * 32 + q8 : lp * 64 + q8 + (kt ^ odd) * 32;
    const int dB = K == 16 ? (lp ^ (4 * nodd)) * 32 + q8 : lp * 64 + q8 + (kt ^ nodd) * 32;
#else   // dense rows: the same reads written in the swizzled form's terms (xB = the next 4 pixels / the other channel half)
    const int xA = lp * (C * 2) + q8 + ct * 32, xB = C == 16 ? xA + 128 : xA + 32;
    const int dA = lp * (K * 2) + q8 + kt * 32, dB = K == 16 ? dA + 128 : dA + 32;
    (void)odd; (void)nodd;
#endif
    f32x4v acc[9][KW * CW];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < KW * CW; ++j) acc[t][j] = f32x4v{0.f, 0.f, 0.f, 0.f};

    // ---- staging (as above): chunk -> (row, column) by carries; all loads of a band in flight, then the LDS stores
    const int xc = xrow_b / 16, xvalid = Wi * C * 2 / 16, dc = drow_b / 16, dvalid = Wo * K * 2 / 16;
    const int xrow_g = Wi * C * 2, drow_g = Wd * K * 2;
    const int xdr = TPBW / xc, xdc = TPBW % xc, ddr = TPBW / dc, ddc = TPBW % dc;
    constexpr int SLX = w16_slx(K, C, W3), SLD = w16_sld(K, C, W3);
    auto stage = [&](long long band, int& rows_out) {
        const long long n = band / bands_per_img;
        const int p0 = (int)(band % bands_per_img) * R;
        const int rows = Ho - p0 < R ? Ho - p0 : R;
        rows_out = rows;
        const int xtot = (rows + 2) * xc, dtot = rows * dc;
        const unsigned char* xg = BNIN ? (const unsigned char*)x + ((n * Ho * Wo) * (long long)C) * 2
                                       : (const unsigned char*)x + (((n * Hi + p0) * Wi) * (long long)C) * 2;
        const unsigned char* dg = (const unsigned char*)dy + ((((n * Hd + p0 + dy_pad) * Wd) + dy_pad) * (long long)K) * 2;
        // (in batches of 3-8 chunks, the two tensors one after the other: 168 registers per wave)
        constexpr int XB = (W3 && BNIN) ? 3 : (NWV != 6 ? 6 : (BNIN ? 3 : 5)), DB = (W3 && BNIN) ? 6 : 8;
        int r = tid / xc, c16 = tid % xc;
#pragma unroll
        for (int u0 = 0; u0 < SLX; u0 += XB) {
            uint4 sbx[XB];
            unsigned xsw = 0;    // two bits per slot: the chunk's place in LDS is its linear one + w16_delta (0, +D, -D chunks: w16_swz)
            unsigned livex = 0;
#pragma unroll
            for (int v = 0; v < XB; ++v) {
                const int u = u0 + v;
                sbx[v] = make_uint4(0, 0, 0, 0);
                if (u < SLX) {
                    xsw |= w16_swz_bits<C>(c16) << (2 * v);
                    if (tid + u * TPBW < xtot && c16 < xvalid) {
                        if (BNIN) {   // chunk c16 of padded row p0 + r = (pixel c16 / CPP, channels 8 * (c16 % CPP) ..) <- clamped source
                            constexpr int CPP = C / 8;
                            const int px = c16 / CPP, cg = c16 % CPP;
                            const int sr = p0 + r - 1 < 0 ? 0 : (p0 + r - 1 > Ho - 1 ? Ho - 1 : p0 + r - 1);
                            const int sc = px - 1 < 0 ? 0 : (px - 1 > Wo - 1 ? Wo - 1 : px - 1);
                            sbx[v] = *reinterpret_cast<const uint4*>(xg + (((long long)sr * Wo + sc) * CPP + cg) * 16);
                            livex |= 1u << v;
                        } else if (CIN > 0) {   // chunk 0 of a pixel: its CIN elements, zeros behind them; chunk 1: zeros
                            if ((c16 & 1) == 0) {
                                const h16_t* e = x + (((n * Hi + p0 + r) * Wi) + (c16 >> 1)) * (long long)CIN;
                                unsigned w0 = e[0], w1 = 0;
                                if (CIN > 1) w0 |= (unsigned)e[1] << 16;
                                if (CIN > 2) w1 = e[2];
                                if (CIN > 3) w1 |= (unsigned)e[3] << 16;
                                sbx[v] = make_uint4(w0, w1, 0, 0);
                            }
                        } else {
                            sbx[v] = *reinterpret_cast<const uint4*>(xg + (long long)r * xrow_g + c16 * 16);
                        }
                    }
                    r += xdr; c16 += xdc;
                    if (c16 >= xc) { c16 -= xc; ++r; }
                }
            }
            if (BNIN) {
                BnReg k;
                k.load(cst, (tid % (C / 8)) * 8);   // TPBW and xc are multiples of C / 8: a thread's chunks all hold the same channels
#pragma unroll
                for (int v = 0; v < XB; ++v) {
                    const uint4 t = bnin_apply(sbx[v], k, bn.slope);
                    const bool on = livex & (1u << v);
                    sbx[v] = make_uint4(on ? t.x : 0u, on ? t.y : 0u, on ? t.z : 0u, on ? t.w : 0u);
                }
            }
#pragma unroll
            for (int v = 0; v < XB; ++v)
                if (u0 + v < SLX && tid + (u0 + v) * TPBW < xtot) *reinterpret_cast<uint4*>(smem + (tid + (u0 + v) * TPBW + w16_delta<C>(xsw >> (2 * v))) * 16) = sbx[v];
        }
        int r2 = tid / dc, d16 = tid % dc;
#pragma unroll
        for (int u0 = 0; u0 < SLD; u0 += DB) {
            uint4 sbd[DB];
            unsigned dsw = 0;
#pragma unroll
            for (int v = 0; v < DB; ++v) {
                const int u = u0 + v;
                sbd[v] = make_uint4(0, 0, 0, 0);
                if (u < SLD) {
                    dsw |= w16_swz_bits<K>(d16) << (2 * v);
                    if (tid + u * TPBW < dtot && d16 < dvalid) sbd[v] = *reinterpret_cast<const uint4*>(dg + (long long)r2 * drow_g + d16 * 16);
                    r2 += ddr; d16 += ddc;
                    if (d16 >= dc) { d16 -= dc; ++r2; }
                }
            }
#pragma unroll
            for (int v = 0; v < DB; ++v)
                if (u0 + v < SLD && tid + (u0 + v) * TPBW < dtot) *reinterpret_cast<uint4*>(smem + dl_off + (tid + (u0 + v) * TPBW + w16_delta<K>(dsw >> (2 * v))) * 16) = sbd[v];
        }
    };

    for (long long band = blockIdx.x; band < total_bands; band += gridDim.x) {
        int rows;
        __syncthreads();                 // the previous band's fragments have been read (first trip, BNIN: the constants are in place)
        stage(band, rows);
        __syncthreads();
        const unsigned char* xl = smem;
        const unsigned char* dl = smem + dl_off;
        // ---- multiply: a wavefront takes column strips (32-pixel segment, half of the band's rows) and slides a three-row window
        //      of raw input dwords down each: per item the new bottom row (3 reads per 16 channels) + the gradient fragments
        const int rh = (rows + 1) >> 1;
        for (int unit = wave / TILES; unit < 2 * nseg; unit += NWV / TILES) {
            const int sg = unit % nseg, half = unit / nseg;
            const int rbeg = half * rh, rend = rbeg + rh < rows ? rbeg + rh : rows;
            if (rbeg >= rend) continue;
            // raw dwords of an input row at this strip: (p0,p1) (p2,p3) (p4,p5) (p6,p7) (p8,p9), p = pixel 8 * grp + ..
            auto row_load = [&](int r, unsigned (&Rg)[CW][5]) {
#pragma unroll
                for (int jc = 0; jc < CW; ++jc) {
                    const unsigned char* b = xl + r * xrow_b + sg * 32 * (C * 2);
                    // (tile jc = 1 -- CW == 2, ct == 0 -- is the other channel half: the lane's pair exchanged)
                    const int xa = jc == 0 ? xA : xB, xb = jc == 0 ? xB : xA;
                    const uint2 q0 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS h16x4*)(b + xa)));
                    const uint2 q1 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS h16x4*)(b + (C == 16 ? xb : xa + 4 * 64))));
                    const uint2 q2 = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS h16x4*)(b + (W16_SWZ ? xb + 8 * (C * 2) : xa + 8 * (C * 2)))));
                    Rg[jc][0] = q0.x; Rg[jc][1] = q0.y; Rg[jc][2] = q1.x; Rg[jc][3] = q1.y; Rg[jc][4] = q2.x;
                }
            };
            // the three column taps of a row from its raw dwords: tap 2 is a renaming of the dword pairs, tap 1 four v_alignbit
            auto row_mma = [&](const h16x8 (&af)[KW], const unsigned (&Rg)[CW][5], int rr) {
#pragma unroll
                for (int jc = 0; jc < CW; ++jc) {
                    const u32x4 f0 = {Rg[jc][0], Rg[jc][1], Rg[jc][2], Rg[jc][3]};
                    const u32x4 f1 = {__builtin_amdgcn_alignbit(Rg[jc][1], Rg[jc][0], 16), __builtin_amdgcn_alignbit(Rg[jc][2], Rg[jc][1], 16),
                                      __builtin_amdgcn_alignbit(Rg[jc][3], Rg[jc][2], 16), __builtin_amdgcn_alignbit(Rg[jc][4], Rg[jc][3], 16)};
                    const u32x4 f2 = {Rg[jc][1], Rg[jc][2], Rg[jc][3], Rg[jc][4]};
#pragma unroll
                    for (int j = 0; j < KW; ++j) {
                        acc[rr * 3 + 0][j * CW + jc] = kp_mfma16x16(af[j], __builtin_bit_cast(h16x8, f0), acc[rr * 3 + 0][j * CW + jc]);
                        acc[rr * 3 + 1][j * CW + jc] = kp_mfma16x16(af[j], __builtin_bit_cast(h16x8, f1), acc[rr * 3 + 1][j * CW + jc]);
                        acc[rr * 3 + 2][j * CW + jc] = kp_mfma16x16(af[j], __builtin_bit_cast(h16x8, f2), acc[rr * 3 + 2][j * CW + jc]);
                    }
                }
            };
            unsigned W0[CW][5], W1[CW][5], W2[CW][5];
            row_load(rbeg, W0);
            row_load(rbeg + 1, W1);
            for (int r = rbeg; r < rend; ++r) {
                const unsigned char* a = dl + r * drow_b + sg * 32 * (K * 2);
                h16x8 af[KW];
#pragma unroll
                for (int j = 0; j < KW; ++j) {
                    const int da = j == 0 ? dA : dB, db = j == 0 ? dB : dA;
                    const h16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS h16x4*)(a + da));
                    const h16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS h16x4*)(a + (K == 16 ? db : da + 4 * 64)));
                    af[j] = h16x8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                }
                row_load(r + 2, W2);
                row_mma(af, W0, 0);
                row_mma(af, W1, 1);
                row_mma(af, W2, 2);
#pragma unroll
                for (int jc = 0; jc < CW; ++jc)
#pragma unroll
                    for (int i = 0; i < 5; ++i) { W0[jc][i] = W1[jc][i]; W1[jc][i] = W2[jc][i]; }
            }
        }
    }
    // D tile of this wavefront: lane column c = ct * 16 + (lane & 15), rows k = kt * 16 + 4 * (lane >> 4) + i.  The wavefronts that own
    // the same tile are summed per tap through LDS in a fixed order; the workgroup's sum is ITS partial gradient (kp_wgrad_unpack adds
    // the partials in order).
    constexpr int TW = KW * CW;
    float (*red)[4 * TW][64] = reinterpret_cast<float (*)[4 * TW][64]>(smem);
    for (int t = 0; t < 9; ++t) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TW; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) red[wave][j * 4 + i][lane] = acc[t][j][i];
        __syncthreads();
        if (wave < TILES) {   // wave == tw
#pragma unroll
            for (int j = 0; j < TW; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = (kt + j / CW) * 16 + 4 * grp + i, c = (ct + j % CW) * 16 + t16;
                    float s = red[wave][j * 4 + i][lane];
#pragma unroll
                    for (int wv = TILES; wv < NWV; wv += TILES) s += red[wave + wv][j * 4 + i][lane];
                    if (CIN == 0) scratch[(long long)blockIdx.x * pstride + ((long long)k * 9 + t) * C + c] = s;
                    else if (c < CIN) scratch[(long long)blockIdx.x * pstride + ((long long)k * 9 + t) * CIN + c] = s;
                }
        }
    }
}

bool small_ch(int v) { return v == 16 || v == 32; }
// measurement: KP_WG16 bits -- 1: the 16 x 16 x 32 kernel at all (0: the 32 x 32 x 16 kernel for every channel pair), 2: for the (32, 32) pair too,
// 4: three wavefronts with both tiles each for the pairs with one 32-channel side (else six wavefronts with one tile each)

// band plan of the 16 x 16 x 32 weight gradient: rows per band (0: not its shape / the rows do not fit), grid = min(bands, 2 per CU)
// `allow_w3`: the three-wave variant's staging capacities may be planned for (the launch site must then launch that variant);
// the first-layer launcher always runs the plain six-wave kernel and plans with its capacities
static unsigned small16_plan(int N, int Ho, int Wo, int C, int K, int& R, int& bpi, bool allow_w3 = true) {
    if (!g_kp_wg16 || !small_ch(C) || !small_ch(K) || (C == 32 && K == 32 && !(g_kp_wg16 & 2))) return 0;
    const int Wp = (Wo + 31) & ~31;
    const long long xrow = (long long)(Wp + 2) * C * 2, drow = (long long)Wp * K * 2;
    long long r = (W16_LDS - 2 * xrow) / (xrow + drow);
    if (r > Ho) r = Ho;
    if (r > 16) r = 16;
    const bool both = C == 32 && K == 32, w3 = allow_w3 && (g_kp_wg16 & 4) && (C == 32) != (K == 32);
    const int slx = both ? w16_slx(32, 32) : (C == 32 ? w16_slx(16, 32, w3) : (K == 32 ? w16_slx(32, 16, w3) : w16_slx(16, 16)));
    const int sld = both ? w16_sld(32, 32) : (C == 32 ? w16_sld(16, 32, w3) : (K == 32 ? w16_sld(32, 16, w3) : w16_sld(16, 16)));
    const long long tpb = both ? w16_waves(32, 32) * 64 : (w3 ? 192 : W16_TPB);
    while (r >= 1 && ((r + 2) * xrow > (long long)slx * tpb * 16 || r * drow > (long long)sld * tpb * 16)) --r;
    if (r < 1) return 0;
    R = (int)r;
    bpi = (Ho + R - 1) / R;
    const long long bands = (long long)N * bpi;
    int cus = kp_device_cus();
    if (cus < 1) cus = 256;
    return (unsigned)(bands < 2LL * cus ? bands : 2LL * cus);
}
template <bool BNIN>
static void small16_launch(int C, int K, unsigned grid, hipStream_t st, const void* x, const void* dy, float* scratch, int Ho, int Wo,
                           int dy_pad, int R, int bpi, long long bands, const BnIn& bn) {
#define KP_W16(Kv, Cv) wgrad3x3_small16_k<Kv, Cv, BNIN><<<grid, w16_waves(Kv, Cv) * 64, 0, st>>>((const h16_t*)x, (const h16_t*)dy, scratch, Ho, Wo, dy_pad, R, bpi, bands, (long long)Kv * 9 * Cv, bn)
#define KP_W16_3(Kv, Cv) wgrad3x3_small16_k<Kv, Cv, BNIN, 0, true><<<grid, 192, 0, st>>>((const h16_t*)x, (const h16_t*)dy, scratch, Ho, Wo, dy_pad, R, bpi, bands, (long long)Kv * 9 * Cv, bn)
    const bool w3 = (g_kp_wg16 & 4) != 0;
    if (K == 16 && C == 16) KP_W16(16, 16);
    else if (K == 16) { if (w3) KP_W16_3(16, 32); else KP_W16(16, 32); }
    else if (C == 16) { if (w3) KP_W16_3(32, 16); else KP_W16(32, 16); }
    else KP_W16(32, 32);
#undef KP_W16_3
#undef KP_W16
}

// band plan of the LDS-staged weight gradient: rows per band (0: the rows do not fit -- use the gather kernel), grid
static int small_band_rows(int Ho, int Wo, int C, int K) {
    const int Wp = (Wo + 15) & ~15;
    const long long xrow = (long long)(Wp + 2) * C * 2, drow = (long long)Wp * K * 2;
    long long R = (WG_LDS - 2 * xrow) / (xrow + drow);
    if (R > Ho) R = Ho;
    if (R > 16) R = 16;
    const long long capx = (long long)(C == 32 ? sl_x(32, 32) : (K == 32 ? sl_x(32, 16) : sl_x(16, 16))) * TPB * 16;
    const long long capd = (long long)(K == 32 ? sl_d(32, 32) : (C == 32 ? sl_d(16, 32) : sl_d(16, 16))) * TPB * 16;
    while (R >= 1 && ((R + 2) * xrow > capx || R * drow > capd)) --R;
    return R < 1 ? 0 : (int)R;
}

}  // namespace

extern kp_knob_t g_kp_small_band;
// band plan of the LDS-staged forward: rows per band (0: a band does not fit), bands per image
static int small_fwd_band_rows(int Ho, int Wo, int C) {
    const int nseg = (Wo + 31) / 32, XPW = nseg * 32 + 2;
    const long long row_l = (long long)XPW * (C * 2 + 16), row_chunks = (long long)XPW * (C / 8);
    long long R = FB_LDS / row_l - 2;
    while (R >= 1 && (R + 2) * row_chunks > (long long)FSL * FTPB) --R;
    if (R > Ho) R = Ho;
    if (R > 16) R = 16;
    return R < 1 ? 0 : (int)R;
}

// the 16 instances of the band kernel: input / output channels 16 | 32, BatchNorm input or stored activation, with / without statistics
template <bool BNIN, int CB, int KB>
static void small_band_launch_ck(unsigned grid, hipStream_t st, const void* x, const void* w, const float* bias, void* y, float* stats,
                                 int Hi, int Wi, int R, int bpi, int bands, const BnIn& bn, const KpPair& pr) {
    if (stats)
        conv3x3_small_band_k<CB, KB, BNIN, true><<<grid, FTPB, 0, st>>>((const h16_t*)x, (const h16_t*)w, bias, (h16_t*)y, stats, Hi, Wi, R, bpi, bands, bn, nullptr, pr);
    else
        conv3x3_small_band_k<CB, KB, BNIN, false><<<grid, FTPB, 0, st>>>((const h16_t*)x, (const h16_t*)w, bias, (h16_t*)y, stats, Hi, Wi, R, bpi, bands, bn, nullptr, pr);
}
template <bool BNIN>
static void small_band_launch(int C, int K, unsigned grid, hipStream_t st, const void* x, const void* w, const float* bias, void* y,
                              float* stats, int Hi, int Wi, int R, int bpi, int bands, const BnIn& bn, const KpPair& pr = KpPair{}) {
    if (C == 16 && K == 16) small_band_launch_ck<BNIN, 1, 1>(grid, st, x, w, bias, y, stats, Hi, Wi, R, bpi, bands, bn, pr);
    else if (C == 16) small_band_launch_ck<BNIN, 1, 2>(grid, st, x, w, bias, y, stats, Hi, Wi, R, bpi, bands, bn, pr);
    else if (K == 16) small_band_launch_ck<BNIN, 2, 1>(grid, st, x, w, bias, y, stats, Hi, Wi, R, bpi, bands, bn, pr);
    else small_band_launch_ck<BNIN, 2, 2>(grid, st, x, w, bias, y, stats, Hi, Wi, R, bpi, bands, bn, pr);
}

// band plan of the LDS-staged forward: rows per band, bands per image, grid (0: no band kernel for this shape)
static long long small_fwd_band_plan(int N, int Ho, int Wo, int C, bool stats, int& R, int& bpi, long long& bands) {
    R = small_fwd_band_rows(Ho, Wo, C);
    if (!R) return 0;
    bpi = (Ho + R - 1) / R;
    bands = (long long)N * bpi;
    if (bands > 0x7fffffffLL) return 0;   // the kernel counts bands in 32 bits
    long long grid = bands < 512 ? bands : 512;
    const long long max_rows = ((long long)N * Ho * Wo + 63) / 64;   // the caller's statistics buffer
    if (stats && grid * FROWS > max_rows) grid = max_rows / FROWS;
    return grid >= 1 ? grid : 0;
}

int kp_conv_small_fwd(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi,
                      int C, int K, hipStream_t st) {
    if (!small_ch(C) || !small_ch(K) || Hi < 3 || Wi < 3) return KP_E_UNSUPPORTED;
    const int Ho = Hi - 2, Wo = Wi - 2;
    if (g_kp_small_band) {
        int R, bpi;
        long long bands;
        if (const long long grid = small_fwd_band_plan(N, Ho, Wo, C, stats != nullptr, R, bpi, bands)) {
            if (stats) kp_set_last_rows(grid * FROWS);
            KP_QUERY_RETURN();
            small_band_launch<false>(C, K, (unsigned)grid, st, x, w, bias, y, stats, Hi, Wi, R, bpi, (int)bands, BnIn{});
            return kp_check_launch("conv3x3_small_band");
        }
    }
    const int spr = (Wo + 31) / 32;
    const long long nseg = (long long)N * Ho * spr;
    // stats rows = waves of the grid; the caller's buffer holds ceil(N*Ho*Wo/64) rows
    long long max_rows = ((long long)N * Ho * Wo + 63) / 64;
    long long blocks = (nseg + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    if (stats && blocks * 4 > max_rows) blocks = max_rows / 4;
    if (blocks < 1) { if (stats) return KP_E_UNSUPPORTED; blocks = 1; }
    if (stats) kp_set_last_rows(blocks * 4);
    KP_QUERY_RETURN();
    if (C == 16)
        conv3x3_small_k<1><<<(unsigned)blocks, TPB, 0, st>>>((const h16_t*)x, (const h16_t*)w, bias, (h16_t*)y, stats, N, Hi, Wi, K, nseg, spr);
    else
        conv3x3_small_k<2><<<(unsigned)blocks, TPB, 0, st>>>((const h16_t*)x, (const h16_t*)w, bias, (h16_t*)y, stats, N, Hi, Wi, K, nseg, spr);
    return kp_check_launch("conv3x3_small");
}

// The data gradient of a 3x3 layer (dy with its zero border of 2, weights in the data-gradient layout) that also leaves the first pass
// of the PRODUCING layer's BatchNorm backward (BRED above): gx [N][H + 2][W + 2][C] as from kp_conv_small_fwd(dy, ..., H + 4, W + 4,
// K, C), partials = rows x [2][C] sums of gh and gh * xhat over yred [N][H][W][C]; rows = kp_last_rows().
int kp_conv_small_dgrad_bnred_ok(int N, int H, int W, int K, int C) {
    int R, bpi;
    long long bands;
    return small_ch(C) && small_ch(K) && H >= 1 && W >= 1 && g_kp_small_band && small_fwd_band_plan(N, H + 2, W + 2, K, true, R, bpi, bands) > 0;
}
int kp_conv_small_dgrad_bnred(const void* dy, const void* w_dgrad, void* gx, const void* yred, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, float slope, float* partials, int N, int H, int W, int K, int C,
                              hipStream_t st) {
    if (!kp_conv_small_dgrad_bnred_ok(N, H, W, K, C)) return KP_E_UNSUPPORTED;
    int R, bpi;
    long long bands;
    const long long grid = small_fwd_band_plan(N, H + 2, W + 2, K, true, R, bpi, bands);   // (statistics-style rows: one per wavefront)
    if (!grid) return KP_E_UNSUPPORTED;
    kp_set_last_rows(grid * FROWS);
    KP_QUERY_RETURN();
    const BnIn bn{mean, invstd, gamma, beta, slope};
    // kernel roles: input channels = K (the gradient's), output channels = C
    // (32 output channels: two chunks of constants / sums per lane -- the weight fragments go to LDS to make room)
#define KP_BRED(CBv, KBv) conv3x3_small_band_k<CBv, KBv, false, false, KBv == 1, false, true><<<(unsigned)grid, FTPB, 0, st>>>((const h16_t*)dy, (const h16_t*)w_dgrad, nullptr, (h16_t*)gx, partials, H + 4, W + 4, R, bpi, (int)bands, bn, (const h16_t*)yred)
    if (K == 16 && C == 16) KP_BRED(1, 1);
    else if (K == 16) KP_BRED(1, 2);
    else if (C == 16) KP_BRED(2, 1);
    else KP_BRED(2, 2);
#undef KP_BRED
    return kp_check_launch("conv3x3_small_band_bnred");
}

// The same forward with the previous layer's BatchNorm-apply + activation + replicate pad on the staging path (BnIn above):
// yprev [N][Ho][Wo][C].  Only the band kernel has this form; the plan (grid, statistics rows) is kp_conv_small_fwd's for
// (N, Ho + 2, Wo + 2, C, K), so kp_conv_fwd_stats_rows of that shape tells the rows.
int kp_conv_small_fwd_bnin_ok(int N, int Ho, int Wo, int C, int K) {
    int R, bpi;
    long long bands;
    return small_ch(C) && small_ch(K) && Ho >= 1 && Wo >= 1 && g_kp_small_band && small_fwd_band_plan(N, Ho, Wo, C, true, R, bpi, bands) > 0;
}
int kp_conv_small_fwd_bnin(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           float slope, const void* w, const float* bias, void* y, float* stats, int N, int Ho, int Wo, int C,
                           int K, hipStream_t st) {
    if (!kp_conv_small_fwd_bnin_ok(N, Ho, Wo, C, K) || !(slope >= 0.f && slope <= 1.f)) return KP_E_UNSUPPORTED;
    int R, bpi;
    long long bands;
    const long long grid = small_fwd_band_plan(N, Ho, Wo, C, stats != nullptr, R, bpi, bands);
    if (!grid) return KP_E_UNSUPPORTED;
    if (stats) kp_set_last_rows(grid * FROWS);
    KP_QUERY_RETURN();
    const BnIn bn{mean, invstd, gamma, beta, slope};
    small_band_launch<true>(C, K, (unsigned)grid, st, yprev, w, bias, y, stats, Ho + 2, Wo + 2, R, bpi, (int)bands, bn);
    return kp_check_launch("conv3x3_small_band_bnin");
}

// kp_conv_fwd_bnin_pair: the launch above for TWO half-batches of N / 2 images at once -- each half exactly the launch of a separate
// call (same grid, same bands per workgroup, same statistics rows: bit-identical results), the second half's workgroups behind the
// first's in one grid.  mean / invstd: [2][C]; stats: [2][rows of a half].
int kp_conv_small_fwd_bnin_pair(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                float slope, const void* w, const float* bias, void* y, float* stats, int N, int Ho, int Wo, int C,
                                int K, hipStream_t st) {
    if ((N & 1) || !kp_conv_small_fwd_bnin_ok(N / 2, Ho, Wo, C, K) || !(slope >= 0.f && slope <= 1.f)) return KP_E_UNSUPPORTED;
    int R, bpi;
    long long bands;
    const long long grid = small_fwd_band_plan(N / 2, Ho, Wo, C, stats != nullptr, R, bpi, bands);
    if (!grid) return KP_E_UNSUPPORTED;
    if (stats) kp_set_last_rows(grid * FROWS);   // rows of ONE half
    KP_QUERY_RETURN();
    const BnIn bn{mean, invstd, gamma, beta, slope};
    KpPair pr;
    pr.hb = (int)grid;
    pr.x = (long long)(N / 2) * Ho * Wo * C;
    pr.y = (long long)(N / 2) * Ho * Wo * K;
    pr.p = grid * FROWS * 2 * K;
    small_band_launch<true>(C, K, (unsigned)(2 * grid), st, yprev, w, bias, y, stats, Ho + 2, Wo + 2, R, bpi, (int)bands, bn, pr);
    return kp_check_launch("conv3x3_small_band_bnin_pair");
}

// workgroups of the weight-gradient kernel = partial gradients in its scratch (0: not a small-channel shape)
extern kp_knob_t g_kp_small_band;   // kp_config(KP_CFG_SMALL_BAND, v): 1 (default) = LDS-staged band kernel where the rows fit
static unsigned small_band_plan(int N, int Ho, int Wo, int C, int K, int& R, int& bpi) {
    if (!small_ch(C) || !small_ch(K) || !g_kp_small_band) return 0;
    R = small_band_rows(Ho, Wo, C, K);
    if (!R) return 0;
    bpi = (Ho + R - 1) / R;
    const long long bands = (long long)N * bpi;
    int cus = kp_device_cus();
    if (cus < 1) cus = 256;
    return (unsigned)(bands < cus ? bands : cus);   // one workgroup per CU (registers, 2 x 76 KB of LDS); it walks bands blockIdx, + grid, ...
}

static unsigned small_wgrad_plan(int N, int Ho, int C, int K, int& rpw) {
    if (!small_ch(C) || !small_ch(K)) return 0;
    const long long rows = (long long)N * Ho;
    rpw = (int)((rows + 2047) / 2048);          // ~2048 wavefronts (512 workgroups), each a contiguous run of rows
    if (rpw < 1) rpw = 1;
    const long long waves = (rows + rpw - 1) / rpw;
    return (unsigned)((waves + 3) / 4);
}
// first layer (CIN <= 4 input channels, K in {16, 32}) on the same kernel: partials / launch
long long kp_conv_small_first_wgrad_partials(int N, int Ho, int Wo, int Cin, int K) {
    int R, bpi;
    if (Cin < 1 || Cin > 4 || !g_kp_small_band) return 0;
    return small16_plan(N, Ho, Wo, 16, K, R, bpi, false);
}
int kp_conv_small_first_wgrad(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int Cin, int K, int dy_pad,
                              hipStream_t st) {
    int R, bpi;
    if (Cin < 1 || Cin > 4 || !g_kp_small_band) return KP_E_UNSUPPORTED;
    const unsigned grid = small16_plan(N, Ho, Wo, 16, K, R, bpi, false);   // (W16_TPB threads, the plain variant's staging: same as launched below)
    if (!grid) return KP_E_UNSUPPORTED;
    const long long bands = (long long)N * bpi;
#define KP_W16F(Kv, CINv) wgrad3x3_small16_k<Kv, 16, false, CINv><<<grid, W16_TPB, 0, st>>>((const h16_t*)x, (const h16_t*)dy, scratch, Ho, Wo, dy_pad, R, bpi, bands, (long long)Kv * 9 * CINv, BnIn{})
#define KP_W16FK(CINv) do { if (K == 16) KP_W16F(16, CINv); else KP_W16F(32, CINv); } while (0)
    switch (Cin) {
        case 1: KP_W16FK(1); break;
        case 2: KP_W16FK(2); break;
        case 3: KP_W16FK(3); break;
        default: KP_W16FK(4); break;
    }
#undef KP_W16FK
#undef KP_W16F
    return kp_check_launch("wgrad3x3_small16_first");
}

long long kp_conv_small_wgrad_partials(int N, int Ho, int Wo, int C, int K) {
    int R, bpi, rpw;
    if (g_kp_small_band)
        if (const unsigned g16 = small16_plan(N, Ho, Wo, C, K, R, bpi)) return g16;
    const unsigned g = small_band_plan(N, Ho, Wo, C, K, R, bpi);
    return g ? g : small_wgrad_plan(N, Ho, C, K, rpw);
}

int kp_conv_small_wgrad(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int dy_pad,
                        hipStream_t st) {
    int R, bpi;
    if (g_kp_small_band)
        if (const unsigned g16 = small16_plan(N, Ho, Wo, C, K, R, bpi)) {
            small16_launch<false>(C, K, g16, st, x, dy, scratch, Ho, Wo, dy_pad, R, bpi, (long long)N * bpi, BnIn{});
            return kp_check_launch("wgrad3x3_small16");
        }
    if (const unsigned bgrid = small_band_plan(N, Ho, Wo, C, K, R, bpi)) {
#define KP_SWB(Kv, Cv) wgrad3x3_small_band_k<Kv, Cv, false><<<bgrid, TPB, 0, st>>>((const h16_t*)x, (const h16_t*)dy, scratch, Ho, Wo, dy_pad, R, bpi, (long long)N * bpi, (long long)Kv * 9 * Cv, BnIn{})
        if (K == 16 && C == 16) KP_SWB(16, 16);
        else if (K == 16) KP_SWB(16, 32);
        else if (C == 16) KP_SWB(32, 16);
        else KP_SWB(32, 32);
#undef KP_SWB
        return kp_check_launch("wgrad3x3_small_band");
    }
    int rpw;
    const unsigned grid = small_wgrad_plan(N, Ho, C, K, rpw);
    if (!grid) return KP_E_UNSUPPORTED;
    const long long rows = (long long)N * Ho;
#define KP_SWG(Kv, Cv) wgrad3x3_small_k<Kv, Cv><<<grid, TPB, 0, st>>>((const h16_t*)x, (const h16_t*)dy, scratch, Ho, Wo, dy_pad, rows, rpw, (long long)Kv * 9 * Cv)
    if (K == 16 && C == 16) KP_SWG(16, 16);
    else if (K == 16) KP_SWG(16, 32);
    else if (C == 16) KP_SWG(32, 16);
    else KP_SWG(32, 32);
#undef KP_SWG
    return kp_check_launch("wgrad3x3_small");
}

// The weight gradient with the input side read from the previous layer's conv output (BnIn): band kernel only; partial
// protocol = kp_conv_small_wgrad_partials of the same shape.
int kp_conv_small_wgrad_bnin_ok(int N, int Ho, int Wo, int C, int K) {
    int R, bpi;
    return small_band_plan(N, Ho, Wo, C, K, R, bpi) > 0;
}
int kp_conv_small_wgrad_bnin(const void* yprev, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             float slope, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int dy_pad,
                             hipStream_t st) {
    int R, bpi;
    if (!(slope >= 0.f && slope <= 1.f)) return KP_E_UNSUPPORTED;
    const BnIn bn{mean, invstd, gamma, beta, slope};
    if (g_kp_small_band)
        if (const unsigned g16 = small16_plan(N, Ho, Wo, C, K, R, bpi)) {
            small16_launch<true>(C, K, g16, st, yprev, dy, scratch, Ho, Wo, dy_pad, R, bpi, (long long)N * bpi, bn);
            return kp_check_launch("wgrad3x3_small16_bnin");
        }
    const unsigned bgrid = small_band_plan(N, Ho, Wo, C, K, R, bpi);
    if (!bgrid) return KP_E_UNSUPPORTED;
#define KP_SWB(Kv, Cv) wgrad3x3_small_band_k<Kv, Cv, true><<<bgrid, TPB, 0, st>>>((const h16_t*)yprev, (const h16_t*)dy, scratch, Ho, Wo, dy_pad, R, bpi, (long long)N * bpi, (long long)Kv * 9 * Cv, bn)
    if (K == 16 && C == 16) KP_SWB(16, 16);
    else if (K == 16) KP_SWB(16, 32);
    else if (C == 16) KP_SWB(32, 16);
    else KP_SWB(32, 32);
#undef KP_SWB
    return kp_check_launch("wgrad3x3_small_band_bnin");
}

#if defined(KP_FB_TRACE) && !defined(KP_F16_BUILD)
extern "C" int kp_debug_wgrad_trace(unsigned long long* dst) {   // host buffer of 512 * 64 stamps
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_wg_trace), sizeof(unsigned long long) * 512 * 64);
}
#endif
