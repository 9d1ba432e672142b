// The image-sized first layer of a Unit (Cin <= 3 -> 64 channels: knn.py:115-119, ReplicationPad2d(1) + Conv2d(3x3) + BatchNorm2d +
// LeakyReLU) with its conv output RECOMPUTED instead of stored.
//
// The layer costs 0.03 GMAC per 128 x 128 image and produces 2 MB: it is bound by streaming its [N][H][W][64] output, and the
// unfused path streams that tensor five times per call -- the convolution writes y, the BatchNorm pass reads y and writes the next
// layer's padded input a, and in the backward the two BatchNorm passes read y again and the weight gradient reads dy.  The input
// image is 48x smaller than y, and 27 x 64 MACs per pixel are four MFMA instructions per 32 pixels, so y is cheaper to recompute
// than to read.  Four kernels, one body (first_fused_k<CIN, MODE>), each streaming at most ONE 64-channel tensor:
//   FM_STATS  conv(x) -> BatchNorm batch statistics only (one partial row per workgroup)            reads x
//   FM_FWD    conv(x) -> BatchNorm-apply + activation -> the next layer's padded input a              reads x, writes a
//   FM_BRED   g = dL/da (padded domain), conv(x) again -> the two sums of the BatchNorm backward      reads g, x
//   FM_BWG    g, conv(x) again -> dy in registers -> the weight gradient (one partial per workgroup)  reads g, x
// Arithmetic, expression for expression, is that of the kernels it replaces on the same values: y is rounded to the storage type
// where the unfused path stores it (kp_conv_first_fwd), the activation is kp_bn_act_fwd's kp_bnin1, gh / xhat / dy are
// kp_bn_act_bwd_reduce's and _apply's (dy rounded to the storage type as kp_bn_act_bwd_apply stores it); what differs is the
// partition of the sums (statistics rows, backward rows, weight-gradient partials): equal up to the order of fp32 additions.
// Reference: keypoints/models/knn.py:115-119, vgg.py:30-38 and their autograd.
#include "kp_common.h"
#include "kp_mfma_util.h"
#include "kp_prof.h"

namespace {

constexpr int FG = 32;            // pixels per group (one MFMA column block)
constexpr int F_RS = 256;         // bytes per staged input row: (32 + 2) pixels * Cin * 2 B <= 204
constexpr int F_XSZ = 1024;       // 3 rows + a zero slot (offset 3 * F_RS) for the padded reduction indices
constexpr int F_TILE = FG * 128;  // a 32-pixel x 64-channel 16-bit tile

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
__device__ __forceinline__ void group_next(GroupPos& c, int Ho, int Wo) {
    c.q += FG;
    if (c.q >= Wo) { c.q = 0; if (++c.p >= Ho) { c.p = 0; ++c.n; } }
}

// The wave-private LDS strip / tile is written as dwords or 16-byte vectors and read back as 16-bit elements or 8-byte pieces:
// every access goes through a may_alias type, or type-based alias analysis treats the differently typed reads as independent of
// the writes (hipcc hoisted the fragment reads out of the group loop: they then saw the first group's pixels for every group).
typedef uint32_t u32_a __attribute__((may_alias));
typedef short s16_a __attribute__((may_alias));
typedef unsigned int u32x2_v __attribute__((ext_vector_type(2)));
typedef u32x2_v u32x2_a __attribute__((may_alias));
typedef u32x4 u32x4_a __attribute__((may_alias));

enum { FM_STATS = 0, FM_FWD = 1, FM_BRED = 2, FM_BWG = 3 };
#ifndef KP_FIRST_GBAR
#define KP_FIRST_GBAR 1   // backward modes: a compiler barrier per group (see the loop)
#endif
#ifndef KP_FIRST_OCC
#define KP_FIRST_OCC 2   // workgroups per CU the KTN = 1 kernels are compiled for (N wavefronts per SIMD: <= 512 / N VGPRs)
#endif

struct FirstArgs {
    const h16_t* x;        // [N][Hi][Wi][CIN] padded input image, storage type
    const h16_t* w;        // [64][3][3][CIN] forward pack
    const float* bias;     // [64] or NULL
    const float *mean, *invstd, *gamma, *beta;   // FWD / BRED / BWG: the layer's BatchNorm (batch statistics), [64] each
    const float* sums;     // BWG: [2][64] sums of gh and gh * xhat (kp_bn_bwd_finalize)
    float slope, invM;
    h16_t* aout;           // FWD: [N][Ho + 2][Wo + 2][64] with its replicate border
    const h16_t* g;        // BRED / BWG: dL/da over the padded domain [N][Ho + 2][Wo + 2][64]
    float* out;            // STATS / BRED: [grid][2][64] partial rows; BWG: [grid][64][9 * CIN] partial weight gradients
    int Hi, Wi;
    long long groups;
    int gpw;
};

// KTN = 2: a wavefront computes all 64 output channels of its groups (two 32-channel MFMA blocks); KTN = 1: a PAIR of wavefronts
// shares a stream of groups, each taking one 32-channel block (kt0 = wave & 1) -- half the accumulators, sums and gradient values per
// lane, twice the wavefronts per byte streamed: the backward modes are latency-bound at two wavefronts per SIMD (BRED: 170 us for
// 272 MB at KTN = 2 with 256 VGPRs), and the two halves of a 128-byte pixel are requested by sibling wavefronts at the same time.
template <int CIN, int MODE, int KTN>
__global__ __launch_bounds__(256, KTN == 1 ? KP_FIRST_OCC : 1) void first_fused_k(FirstArgs A) {
    constexpr int J = 9 * CIN, NS = (J + 15) / 16, K = 64, PK = K * 2;
    constexpr int ND = (FG + 2) * CIN * 2 / 4;            // dwords per staged input row
    constexpr int WSZ = F_XSZ + F_TILE;                   // per wave: input strip + one 32 px x 64 ch tile (KTN = 1: its half of it)
    constexpr bool BWD = MODE == FM_BRED || MODE == FM_BWG;
    constexpr int SPW = KTN == 1 ? 2 : 4;                 // group streams per workgroup
    constexpr int NCH = 4 * KTN;                          // 16-byte chunks of a pixel this wavefront handles
    constexpr int NLD = KTN == 1 ? 2 : 4;                 // 16-byte loads / stores per lane that cover its part of a 32-pixel tile
    static_assert(MODE != FM_BWG || KTN == 1, "the weight-gradient mode reduces two streams of two channel blocks");
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * WSZ];
    __shared__ __attribute__((aligned(16))) float cst[7][K];   // mean, invstd, gamma, beta, (BWG) sum_gh / M, sum_gh_xhat / M; [6] bias
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kt0 = KTN == 1 ? (wave & 1) : 0;            // first 32-channel block of this wavefront
    const int sid = KTN == 1 ? (wave >> 1) : wave;        // its group stream within the workgroup
    unsigned char* xs = smem + wave * WSZ;
    unsigned char* tile = xs + F_XSZ;
    const int px = lane & 31, h = lane >> 5;
    const int Hi = A.Hi, Wi = A.Wi, Ho = Hi - 2, Wo = Wi - 2;
    if (lane < 4) *reinterpret_cast<u32_a*>(xs + 3 * F_RS + lane * 4) = 0u;
    if (tid < K) {
        if (MODE != FM_STATS) { cst[0][tid] = A.mean[tid]; cst[1][tid] = A.invstd[tid]; cst[2][tid] = A.gamma[tid]; cst[3][tid] = A.beta[tid]; }
        if (MODE == FM_BWG) { cst[4][tid] = A.sums[tid] * A.invM; cst[5][tid] = A.sums[K + tid] * A.invM; }
        cst[6][tid] = A.bias ? A.bias[tid] : 0.f;
    }
    __syncthreads();

    // weights [K][J] -> A fragments (row = output channel, 8 consecutive reduction indices per lane), once per wave
    h16x8 wa[KTN][NS];
#pragma unroll
    for (int kk = 0; kk < KTN; ++kk)
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int j = s * 16 + h * 8 + i;
                wa[kk][s][i] = j < J ? (short)A.w[((kt0 + kk) * 32 + px) * J + j] : (short)0;
            }
    // accumulator layout: lane = pixel px, register i of block kt = channel kt * 32 + 8 * (i >> 2) + 4 * h + (i & 3)
    // (the accumulators start from the bias, re-read from LDS per group: 32 registers the backward modes do not have)
    int xo[NS][8];   // LDS byte offsets of this lane's 8 * NS reduction elements
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = s * 16 + h * 8 + i;
            xo[s][i] = j < J ? (j / (3 * CIN)) * F_RS + (px * CIN + j % (3 * CIN)) * 2 : 3 * F_RS;
        }
    // running sums of the wave (STATS: sum y, sum y^2; BRED: sum gh, sum gh * xhat), per accumulator register
    constexpr bool SUMS = MODE == FM_STATS || MODE == FM_BRED;
    float s0[SUMS ? KTN : 1][16], s1[SUMS ? KTN : 1][16];
    if (SUMS) {
#pragma unroll
        for (int kk = 0; kk < KTN; ++kk)
#pragma unroll
            for (int i = 0; i < 16; ++i) s0[kk][i] = s1[kk][i] = 0.f;
    }
    // BWG: the weight gradient D[k][j] of the wave's channel block, and the lane roles of its fragments (conv3_first_wgrad_mfma_k)
    f32x16 dw;
    int kbase[4] = {0, 0, 0, 0};
    int xoff = 0;
    if (MODE == FM_BWG) {
        const int t16 = lane & 15, g16 = (lane >> 4) & 1;
        const int lrow = 8 * h + (t16 >> 2);
#pragma unroll
        for (int i = 0; i < 16; ++i) dw[i] = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) kbase[v] = tr_base<PK>(v, lrow, (kt0 * 32 + 16 * g16 + (t16 & 3) * 4) * 2);
        const int jj = (lane & 31) < J ? (lane & 31) : J - 1;   // columns >= J are discarded: any valid address
        xoff = (jj / (3 * CIN)) * F_RS + (jj % (3 * CIN)) * 2 + 8 * h * CIN * 2;
    }

    const long long wid = (long long)blockIdx.x * SPW + sid;
    const long long g0 = wid * A.gpw, g1 = g0 + A.gpw < A.groups ? g0 + A.gpw : A.groups;
    const int Wp = Wo + 2;                                // pitch (pixels) of the padded 64-channel tensors a / g
    uint32_t xr[3] = {0u, 0u, 0u};
    u32x4 gt[NLD];   // (BWD only: dead otherwise; clang vectors, not HIP's uint4 struct: arrays of those stay in scratch memory)
    GroupPos c = group_pos(g0 < A.groups ? g0 : 0, Ho, Wo);   // fetch cursor
    GroupPos cur = c;                                         // the group being worked on
    // the lane's NLD 16-byte pieces of a 32-pixel tile in memory: piece id -> (pixel, byte offset inside the pixel's 128 bytes)
    auto piece = [&](int i, int& pp, int& c16) __attribute__((always_inline)) {
        const int id = i * 64 + lane;
        if (KTN == 1) { pp = id >> 2; c16 = kt0 * 4 + (id & 3); } else { pp = id >> 3; c16 = id & 7; }
    };
    auto load_tile = [&](const h16_t* base, u32x4 (&t)[NLD]) __attribute__((always_inline)) {   // base: the tile's first pixel
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int pp, c16;
            piece(i, pp, c16);
            t[i] = *reinterpret_cast<const u32x4*>(base + pp * K + c16 * 8);
        }
    };
    auto fetch = [&]() __attribute__((always_inline)) {
        if (lane < ND) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
                xr[r] = reinterpret_cast<const uint32_t*>(A.x + ((c.n * Hi + c.p + r) * Wi + c.q) * CIN)[lane];
        }
        if constexpr (BWD)   // the group's 32 pixels of g: padded row p + 1, columns q + 1 ..
            load_tile(A.g + (((c.n * (Ho + 2) + c.p + 1) * Wp) + c.q + 1) * K, gt);
        group_next(c, Ho, Wo);
    };
    // tile <-> registers.  Memory side: piece (pixel pp, chunk c16) at pp * 128 + ((c16 ^ (pp & 7)) << 4); accumulator side: the lane's
    // 8 bytes (channels 4 h .. 4 h + 3) of chunk ch = kt * 4 + g4.
    auto tile_put = [&](const u32x4 (&t)[NLD]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            int pp, c16;
            piece(i, pp, c16);
            *reinterpret_cast<u32x4_a*>(tile + pp * PK + ((c16 ^ (pp & 7)) << 4)) = t[i];
        }
    };
    auto tile_get = [&](float (&gv)[KTN][16], bool add) __attribute__((always_inline)) {
#pragma unroll
        for (int cc = 0; cc < NCH; ++cc) {
            const int ch = kt0 * 4 + cc;
            const u32x2_v v = *reinterpret_cast<const u32x2_a*>(tile + px * PK + ((ch ^ (px & 7)) << 4) + h * 8);
            float f[4];
            unpack_h16x2(v[0], f[0], f[1]); unpack_h16x2(v[1], f[2], f[3]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float& d = gv[cc >> 2][(cc & 3) * 4 + e];
                d = add ? d + f[e] : f[e];
            }
        }
    };
    // the replicate pad's adjoint along the row: pixel 0 of the image row also takes padded column 0, pixel Wo - 1 column Wo + 1
    auto col_extra = [&](const h16_t* grow, float (&gv)[KTN][16]) __attribute__((always_inline)) {   // grow: padded row, column 0
        const bool left = cur.q == 0 && px == 0, right = cur.q + FG == Wo && px == FG - 1;
        if (left || right) {
            const h16_t* src = grow + (long long)(left ? 0 : Wo + 1) * K + kt0 * 32 + 4 * h;
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) {
                const uint2 v = *reinterpret_cast<const uint2*>(src + cc * 8);
                float f[4];
                unpack_h16x2(v.x, f[0], f[1]); unpack_h16x2(v.y, f[2], f[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[cc >> 2][(cc & 3) * 4 + e] += f[e];
            }
        }
    };

    if (g0 < g1) fetch();
    for (long long g = g0; g < g1; ++g) {
        // (a compiler barrier per group: without it the per-channel constants -- loop invariants in LDS -- are hoisted into 64-128
        //  registers for the whole loop, and the kernel runs at two wavefronts per SIMD instead of four)
        if (!BWD || KP_FIRST_GBAR) asm volatile("" ::: "memory");
        if (lane < ND) {
#pragma unroll
            for (int r = 0; r < 3; ++r) *reinterpret_cast<u32_a*>(xs + r * F_RS + lane * 4) = xr[r];
        }
        if constexpr (BWD) tile_put(gt);
        h16x8 xb[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i) xb[s][i] = *reinterpret_cast<const s16_a*>(xs + xo[s][i]);
        h16x8 xb2[2];
        if (MODE == FM_BWG) {
#pragma unroll
            for (int R = 0; R < 2; ++R)
#pragma unroll
                for (int i = 0; i < 8; ++i) xb2[R][i] = *reinterpret_cast<const s16_a*>(xs + xoff + (R * 16 + i) * CIN * 2);
        }
        float gv[KTN][16];   // (BWD only)
        if constexpr (BWD) tile_get(gv, false);
        if (g + 1 < g1) fetch();                              // the next group's loads fly during this group's arithmetic
        if constexpr (BWD) {
            const h16_t* grow = A.g + ((cur.n * (Ho + 2) + cur.p + 1) * Wp) * K;
            col_extra(grow, gv);
            if (cur.p == 0 || cur.p == Ho - 1) {              // top / bottom image row: the border row of the padded domain folds onto it
                const h16_t* brow = cur.p == 0 ? grow - (long long)Wp * K : grow + (long long)Wp * K;
                u32x4 bt[NLD];
                load_tile(brow + (long long)(cur.q + 1) * K, bt);
                tile_put(bt);
                tile_get(gv, true);
                col_extra(brow, gv);
                if (Ho == 1) {                                // a one-row image: top AND bottom
                    const h16_t* brow2 = grow + (long long)Wp * K;
                    load_tile(brow2 + (long long)(cur.q + 1) * K, bt);
                    tile_put(bt);
                    tile_get(gv, true);
                    col_extra(brow2, gv);
                }
            }
        }
        f32x16 acc[KTN];
#pragma unroll
        for (int kk = 0; kk < KTN; ++kk) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 b4 = *reinterpret_cast<const float4*>(&cst[6][(kt0 + kk) * 32 + 8 * g4 + 4 * h]);
                acc[kk][g4 * 4 + 0] = b4.x; acc[kk][g4 * 4 + 1] = b4.y; acc[kk][g4 * 4 + 2] = b4.z; acc[kk][g4 * 4 + 3] = b4.w;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[kk] = kp_mfma16(wa[kk][s], xb[s], acc[kk], 0, 0, 0);
        }

        if constexpr (MODE == FM_STATS) {
#pragma unroll
            for (int kk = 0; kk < KTN; ++kk)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float v = acc[kk][i];
                    s0[kk][i] += v;
                    s1[kk][i] = __builtin_fmaf(v, v, s1[kk][i]);
                }
        } else {
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) {
                const int kk = cc >> 2, g4 = cc & 3, ch = kt0 * 4 + cc, c0 = ch * 8 + 4 * h;
                // y as the unfused path stores and re-reads it: rounded once to the storage type
                float yv[4];
                {
                    const uint32_t p0 = pack_h16x2(acc[kk][g4 * 4 + 0], acc[kk][g4 * 4 + 1]);
                    const uint32_t p1 = pack_h16x2(acc[kk][g4 * 4 + 2], acc[kk][g4 * 4 + 3]);
                    unpack_h16x2(p0, yv[0], yv[1]); unpack_h16x2(p1, yv[2], yv[3]);
                }
                const float4 mu4 = *reinterpret_cast<const float4*>(&cst[0][c0]), is4 = *reinterpret_cast<const float4*>(&cst[1][c0]);
                const float4 ga4 = *reinterpret_cast<const float4*>(&cst[2][c0]), be4 = *reinterpret_cast<const float4*>(&cst[3][c0]);
                const float mu[4] = {mu4.x, mu4.y, mu4.z, mu4.w}, is[4] = {is4.x, is4.y, is4.z, is4.w};
                const float ga[4] = {ga4.x, ga4.y, ga4.z, ga4.w}, be[4] = {be4.x, be4.y, be4.z, be4.w};
                if constexpr (MODE == FM_FWD) {
                    float a4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) a4[e] = kp_bnin1(yv[e], mu[e], is[e], ga[e], be[e], A.slope);
                    u32x2_v pk;
                    pk[0] = pack_h16x2(a4[0], a4[1]);
                    pk[1] = pack_h16x2(a4[2], a4[3]);
                    *reinterpret_cast<u32x2_a*>(tile + px * PK + ((ch ^ (px & 7)) << 4) + h * 8) = pk;
                } else {
                    float db[4] = {0.f, 0.f, 0.f, 0.f}, dg[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (MODE == FM_BWG) {
                        const float4 b4 = *reinterpret_cast<const float4*>(&cst[4][c0]), g4v = *reinterpret_cast<const float4*>(&cst[5][c0]);
                        db[0] = b4.x; db[1] = b4.y; db[2] = b4.z; db[3] = b4.w;
                        dg[0] = g4v.x; dg[1] = g4v.y; dg[2] = g4v.z; dg[3] = g4v.w;
                    }
                    float res[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xh = (yv[e] - mu[e]) * is[e];
                        const float z = xh * ga[e] + be[e];
                        const float gh = gv[kk][g4 * 4 + e] * act_grad(z, A.slope);
                        if constexpr (MODE == FM_BRED) {
                            s0[kk][g4 * 4 + e] += gh;
                            s1[kk][g4 * 4 + e] += gh * xh;
                        } else {
                            res[e] = ga[e] * is[e] * (gh - db[e] - xh * dg[e]);
                        }
                    }
                    if constexpr (MODE == FM_BWG) {   // dy, rounded as kp_bn_act_bwd_apply stores it, into the transpose-read tile
                        u32x2_v pk;
                        pk[0] = pack_h16x2(res[0], res[1]);
                        pk[1] = pack_h16x2(res[2], res[3]);
                        const int cb = c0 * 2;
                        *reinterpret_cast<u32x2_a*>(tile + px * PK + ((((cb >> 6) ^ blk_swz<PK>(px)) << 6) | (cb & 63))) = pk;
                    }
                }
            }
            if constexpr (MODE == FM_FWD) {
                // the 32 pixels as 4 KB contiguous: padded row p + 1, columns q + 1 ..; border copies of the replicate pad
                h16_t* arow = A.aout + (((cur.n * (Ho + 2) + cur.p + 1) * Wp) + cur.q + 1) * K;
                const bool top = cur.p == 0, bot = cur.p == Ho - 1, lft = cur.q == 0, rgt = cur.q + FG == Wo;
#pragma unroll
                for (int i = 0; i < NLD; ++i) {
                    int pp, c16;
                    piece(i, pp, c16);
                    const u32x4 vv = *reinterpret_cast<const u32x4_a*>(tile + pp * PK + ((c16 ^ (pp & 7)) << 4));
                    h16_t* dst = arow + pp * K + c16 * 8;
                    *reinterpret_cast<u32x4*>(dst) = vv;
                    if (top) *reinterpret_cast<u32x4*>(dst - (long long)Wp * K) = vv;
                    if (bot) *reinterpret_cast<u32x4*>(dst + (long long)Wp * K) = vv;
                    const bool le = lft && pp == 0, re = rgt && pp == FG - 1;
                    if (le || re) {
                        h16_t* d2 = le ? dst - K : dst + K;
                        *reinterpret_cast<u32x4*>(d2) = vv;
                        if (top) *reinterpret_cast<u32x4*>(d2 - (long long)Wp * K) = vv;
                        if (bot) *reinterpret_cast<u32x4*>(d2 + (long long)Wp * K) = vv;
                    }
                }
            }
            if constexpr (MODE == FM_BWG) {
                asm volatile("" ::: "memory");   // (the transpose reads are builtins: keep them behind the dy stores above)
                const h16x8 af0 = tr_frag<PK, 0>(tile, kbase), af1 = tr_frag<PK, 16>(tile, kbase);
                dw = kp_mfma16(af0, xb2[0], dw, 0, 0, 0);
                dw = kp_mfma16(af1, xb2[1], dw, 0, 0, 0);
            }
        }
        group_next(cur, Ho, Wo);
    }

    // ---- the workgroup's result: its group streams added in a fixed order ---------------------------------------------------
    __syncthreads();                                          // everyone is done with its strip and tile
    float* red = reinterpret_cast<float*>(smem);
    if constexpr (SUMS) {                                     // one row [2][64] per workgroup
#pragma unroll
        for (int kk = 0; kk < KTN; ++kk) {
            int r1, r2;
            const float t0 = reduce16_over32(s0[kk], lane, r1);
            const float t1 = reduce16_over32(s1[kk], lane, r2);
            if ((lane & 1) == 0) {
                const int k = (kt0 + kk) * 32 + 8 * (r1 >> 2) + 4 * h + (r1 & 3);
                red[(sid * 2 + 0) * K + k] = t0;              // (KTN = 1: the stream's two wavefronts fill the two halves of its row)
                red[(sid * 2 + 1) * K + k] = t1;
            }
        }
        __syncthreads();
        if (tid < 2 * K) {
            float s = red[tid] + red[2 * K + tid];
            if (SPW == 4) s = (s + red[4 * K + tid]) + red[6 * K + tid];
            A.out[(long long)blockIdx.x * 2 * K + tid] = s;
        }
    } else if constexpr (MODE == FM_BWG) {                    // one partial gradient [64][J] per workgroup; D: column = j, rows as above
#pragma unroll
        for (int i = 0; i < 16; ++i) red[((sid * 2 + kt0) * 16 + i) * 64 + lane] = dw[i];
        __syncthreads();
        for (int e = tid; e < 2 * 16 * 64; e += 256) {
            const float s = red[e] + red[2 * 16 * 64 + e];    // stream 0 + stream 1
            const int l = e & 63, i = (e >> 6) & 15, kt = e >> 10;
            const int k = kt * 32 + 8 * (i >> 2) + 4 * (l >> 5) + (i & 3), j = l & 31;
            if (j < J) A.out[((long long)blockIdx.x * K + k) * J + j] = s;
        }
    }
}

// groups per stream / workgroups: ~4 workgroups of 4 wavefronts per CU over the device; the grid -- = statistics rows = backward rows =
// weight-gradient partials, <= 1024 -- is the same for every mode: a KTN = 1 kernel has 2 streams of 2 * per groups per workgroup
// where a KTN = 2 kernel has 4 of `per`
static bool fused_plan(int N, int H, int W, int C, int K, long long& groups, int& per, unsigned& grid) {
    if (!(K == 64 && C >= 1 && C <= 3 && W % FG == 0 && H >= 1 && N >= 1)) return false;
    groups = (long long)N * H * W / FG;
    long long p = (groups + 4095) / 4096;
    if (p < 4) p = 4;
    if (p > 0x3fffffffLL) return false;
    per = (int)p;
    const long long wgs = (groups + 4LL * per - 1) / (4LL * per);
    if (wgs > 0x7fffffffLL) return false;
    grid = (unsigned)wgs;
    return true;
}

template <int MODE, int KTN>
static int fused_launch(FirstArgs a, int per, int C, unsigned grid, hipStream_t st, const char* what) {
    a.gpw = KTN == 1 ? 2 * per : per;
    switch (C) {
        case 1: first_fused_k<1, MODE, KTN><<<grid, 256, 0, st>>>(a); break;
        case 2: first_fused_k<2, MODE, KTN><<<grid, 256, 0, st>>>(a); break;
        default: first_fused_k<3, MODE, KTN><<<grid, 256, 0, st>>>(a); break;
    }
    return kp_check_launch(what);
}

}  // namespace

extern "C" {

int kp_conv_first_fused_supported(int N, int H, int W, int C, int K, int dtype) {
    long long groups; int gpw; unsigned grid;
    return (dtype == KP_H16 && fused_plan(N, H, W, C, K, groups, gpw, grid)) ? 1 : 0;
}

long long kp_conv_first_fused_rows(int N, int H, int W, int C, int K, int dtype) {
    long long groups; int gpw; unsigned grid;
    if (dtype != KP_H16 || !fused_plan(N, H, W, C, K, groups, gpw, grid)) return 0;
    return (long long)grid;
}

int kp_conv_first_stats(const void* x, const void* w, const float* bias, float* stats, int N, int H, int W, int C, int K, int dtype,
                        void* stream) {
    KP_REQUIRE(x && w && stats && N > 0 && H > 0 && W > 0, "kp_conv_first_stats: bad args");
    long long groups; int gpw; unsigned grid;
    if (dtype != KP_H16 || !fused_plan(N, H, W, C, K, groups, gpw, grid)) return KP_E_UNSUPPORTED;
    ProfScope ps(KP_PROF_CONV_FIRST, 2.0 * N * (double)H * W * C * K * 9, (hipStream_t)stream, 2.0 * N * (double)(H + 2) * (W + 2) * C);
    FirstArgs a{};
    a.x = (const h16_t*)x; a.w = (const h16_t*)w; a.bias = bias; a.out = stats; a.Hi = H + 2; a.Wi = W + 2; a.groups = groups;
    const int rc = fused_launch<FM_STATS, 2>(a, gpw, C, grid, (hipStream_t)stream, "conv_first_stats");
    ps.done(KP_PROF_CONV_FIRST);
    return rc;
}

int kp_conv_first_fwd_bnact(const void* x, const void* w, const float* bias, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, float slope, void* a_out, int N, int H, int W, int C, int K, int dtype, void* stream) {
    KP_REQUIRE(x && w && mean && invstd && gamma && beta && a_out && N > 0 && H > 0 && W > 0, "kp_conv_first_fwd_bnact: bad args");
    KP_REQUIRE(slope >= 0.f && slope <= 1.f, "kp_conv_first_fwd_bnact: activation slope must be in [0, 1]");
    long long groups; int gpw; unsigned grid;
    if (dtype != KP_H16 || !fused_plan(N, H, W, C, K, groups, gpw, grid) || ((size_t)a_out & 15)) return KP_E_UNSUPPORTED;
    ProfScope ps(KP_PROF_CONV_FIRST, 2.0 * N * (double)H * W * C * K * 9, (hipStream_t)stream,
                 2.0 * (N * (double)(H + 2) * (W + 2) * C + N * (double)(H + 2) * (W + 2) * K));
    FirstArgs a{};
    a.x = (const h16_t*)x; a.w = (const h16_t*)w; a.bias = bias; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta;
    a.slope = slope; a.aout = (h16_t*)a_out; a.Hi = H + 2; a.Wi = W + 2; a.groups = groups;
    extern kp_knob_t g_kp_first_ktn;
    const int rc = g_kp_first_ktn != 1 ? fused_launch<FM_FWD, 2>(a, gpw, C, grid, (hipStream_t)stream, "conv_first_fwd_bnact")
                                       : fused_launch<FM_FWD, 1>(a, gpw, C, grid, (hipStream_t)stream, "conv_first_fwd_bnact");
    ps.done(KP_PROF_CONV_FIRST);
    return rc;
}

int kp_conv_first_bwd_reduce(const void* g, const void* x, const void* w, const float* bias, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, float slope, float* partials, int N, int H, int W, int C, int K,
                             int dtype, void* stream) {
    KP_REQUIRE(g && x && w && mean && invstd && gamma && beta && partials && N > 0 && H > 0 && W > 0, "kp_conv_first_bwd_reduce: bad args");
    long long groups; int gpw; unsigned grid;
    if (dtype != KP_H16 || !fused_plan(N, H, W, C, K, groups, gpw, grid) || ((size_t)g & 15)) return KP_E_UNSUPPORTED;
    ProfScope ps(KP_PROF_CONV_FIRST, 2.0 * N * (double)H * W * C * K * 9, (hipStream_t)stream,
                 2.0 * (N * (double)(H + 2) * (W + 2) * C + N * (double)(H + 2) * (W + 2) * K));
    FirstArgs a{};
    a.x = (const h16_t*)x; a.w = (const h16_t*)w; a.bias = bias; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta;
    a.slope = slope; a.g = (const h16_t*)g; a.out = partials; a.Hi = H + 2; a.Wi = W + 2; a.groups = groups;
    const int rc = fused_launch<FM_BRED, 1>(a, gpw, C, grid, (hipStream_t)stream, "conv_first_bwd_reduce");
    ps.done(KP_PROF_CONV_FIRST);
    return rc;
}

int kp_conv_first_bwd_wgrad(const void* g, const void* x, const void* w, const float* bias, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, float slope, const float* sums, float* scratch, int N, int H, int W,
                            int C, int K, int dtype, void* stream) {
    KP_REQUIRE(g && x && w && mean && invstd && gamma && beta && sums && scratch && N > 0 && H > 0 && W > 0, "kp_conv_first_bwd_wgrad: bad args");
    long long groups; int gpw; unsigned grid;
    if (dtype != KP_H16 || !fused_plan(N, H, W, C, K, groups, gpw, grid) || ((size_t)g & 15)) return KP_E_UNSUPPORTED;
    ProfScope ps(KP_PROF_CONV_FIRST, 4.0 * N * (double)H * W * C * K * 9, (hipStream_t)stream,
                 2.0 * (N * (double)(H + 2) * (W + 2) * C + N * (double)(H + 2) * (W + 2) * K) + 4.0 * grid * K * 9 * C);
    FirstArgs a{};
    a.x = (const h16_t*)x; a.w = (const h16_t*)w; a.bias = bias; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta;
    a.slope = slope; a.sums = sums; a.invM = 1.0f / (float)((long long)N * H * W); a.g = (const h16_t*)g; a.out = scratch;
    a.Hi = H + 2; a.Wi = W + 2; a.groups = groups;
    const int rc = fused_launch<FM_BWG, 1>(a, gpw, C, grid, (hipStream_t)stream, "conv_first_bwd_wgrad");
    ps.done(KP_PROF_CONV_FIRST);
    return rc;
}

}  // extern "C"
