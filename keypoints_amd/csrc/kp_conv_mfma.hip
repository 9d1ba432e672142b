// bf16 MFMA implicit-GEMM convolution for gfx950 (CDNA4): forward / data-gradient ("valid" 3x3
// correlation over a physically padded NHWC image) and weight-gradient.
//
// Forward / dgrad GEMM view:   Y[m][k] = sum_{tap,c} X[pix(m)+tap][c] * W[k][tap][c]
//   M = N*Ho*Wo pixels, N = K output channels, reduction = 9*C.
// Workgroup tile 128 pixels x BN channels (BN = 128 or 64), reduction step BK = 64 channels of one tap,
// 4 wavefronts in a 2x2 layout (wave tile 64 pixels x BN/2 channels), v_mfma_f32_32x32x16_bf16.
// Operands are swapped (MFMA "A" = weights, "B" = pixels) so that a lane's accumulator registers hold
// 4 *consecutive output channels* of one pixel -> 8-byte packed bf16 stores into the NHWC output.
// Both tiles go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered.
// LDS rows are 128 B (64 bf16); the 16-B slot index is XOR-swizzled with (row>>1)&7, applied on the
// per-lane *source* address (the LDS-DMA destination is lane-linear) and on the ds_read_b128 address,
// which makes every ds_read_b128 lane group hit 16 distinct 16-B bank slots (conflict-free).
#include <stdlib.h>
#include "kp_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short h16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int BM = 128, BK = 64, TPB = 256;
constexpr int ROWB = BK * 2;   // bytes per LDS row

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    // 16 B per lane: LDS destination = wave-uniform base + lane*16
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int swz(int row, int slot) { return slot ^ ((row >> 1) & 7); }

// halving butterfly: 16 per-lane values -> each lane ends with the sum over its 32-lane half of ONE value.
// Returns that sum; `reg` receives which of the 16 inputs it belongs to.
__device__ __forceinline__ float reduce16_over32(float (&v)[16], int lane, int& reg) {
    float a[8];
    const bool b4 = lane & 16;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float keep = b4 ? v[i + 8] : v[i], send = b4 ? v[i] : v[i + 8];
        a[i] = keep + __shfl_xor(send, 16, 64);
    }
    float b[4];
    const bool b3 = lane & 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float keep = b3 ? a[i + 4] : a[i], send = b3 ? a[i] : a[i + 4];
        b[i] = keep + __shfl_xor(send, 8, 64);
    }
    float c[2];
    const bool b2 = lane & 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float keep = b2 ? b[i + 2] : b[i], send = b2 ? b[i] : b[i + 2];
        c[i] = keep + __shfl_xor(send, 4, 64);
    }
    const bool b1 = lane & 2;
    float d = (b1 ? c[1] : c[0]) + __shfl_xor(b1 ? c[0] : c[1], 2, 64);
    d += __shfl_xor(d, 1, 64);
    reg = (b4 ? 8 : 0) | (b3 ? 4 : 0) | (b2 ? 2 : 0) | (b1 ? 1 : 0);
    return d;
}

// Output pixel m -> (image n, row p, col q).
// MODE 0: all Ho x Wo pixels in row-major order.
// MODE 1: only the 1-pixel border ring of the output (2*Wo + 2*(Ho-2) pixels per image), all 9 taps: used for
//         outputs that are a 16x16 multiple plus a ring (interior: LDS-patch kernel, kp_conv_patch.hip).
// MODE 2: the same ring for the data-gradient, whose input (dy) is known to carry a ZERO border of 2: ring pixels
//         are grouped by edge (grp 0 top row, 1 bottom row, 2 left column, 3 right column; m indexes within the
//         group) and only the 3 taps that can touch non-zero input are computed.
template <int MODE>
__device__ __forceinline__ void pix_decode(long long m, int Ho, int Wo, int grp, long long& n, int& p, int& q) {
    if (MODE == 0) {
        q = (int)(m % Wo);
        const long long t = m / Wo;
        p = (int)(t % Ho);
        n = t / Ho;
    } else if (MODE == 1) {
        const int RP = 2 * Wo + 2 * (Ho - 2);
        n = m / RP;
        const int r = (int)(m % RP);
        if (r < Wo) { p = 0; q = r; }
        else if (r < 2 * Wo) { p = Ho - 1; q = r - Wo; }
        else { const int r2 = r - 2 * Wo; p = 1 + (r2 >> 1); q = (r2 & 1) ? Wo - 1 : 0; }
    } else {
        if (grp < 2) { n = m / Wo; q = (int)(m % Wo); p = grp ? Ho - 1 : 0; }
        else { n = m / (Ho - 2); p = 1 + (int)(m % (Ho - 2)); q = (grp == 3) ? Wo - 1 : 0; }
    }
}

// BRED (MODE 2 only): the ring's share of the first pass of the PRODUCING layer's BatchNorm backward (see kp_conv_patch.hip):
// every ring pixel (p, q) of the padded domain meets the factors of its clamped source pixel (clamp(p-1), clamp(q-1)) of
// `yred` [N][Ho-2][Wo-2][K]; rows [2][K] per 64 ring pixels (tile-global numbering; rows of a group's empty tail stay as the
// launcher zeroed them).
struct RingBred { const h16_t* yred; const float *mean, *invstd, *gamma, *beta; float slope; };

template <int BN, int MODE, bool BRED = false>
__global__ __launch_bounds__(TPB, 2) void conv3x3_mfma_k(const h16_t* __restrict__ x, const h16_t* __restrict__ w,
                                                         const float* __restrict__ bias, h16_t* __restrict__ y,
                                                         float* __restrict__ partials, int N, int Hi, int Wi, int C,
                                                         int K, int mtiles, int ktiles, const RingBred rb = RingBred{}) {
    static_assert(!BRED || MODE == 2, "BRED: the data gradient's ring");
    constexpr int WN = BN / 2;            // channels per wave
    constexpr int TN = WN / 32;           // 32-channel MFMA tiles per wave (2 or 1)
    constexpr int XBYTES = BM * ROWB, WBYTES = BN * ROWB, STAGE = XBYTES + WBYTES;
    constexpr int XI = BM / 32, WI = BN / 32;   // glds instructions per wave per stage
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int Ho = Hi - 2, Wo = Wi - 2;
    long long M = MODE == 1 ? (long long)N * (2 * Wo + 2 * (Ho - 2)) : (long long)N * Ho * Wo;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile order: the dispatcher round-robins workgroups over the 8 XCDs; give each XCD a
    // contiguous run of (m-tile, k-tile) pairs with the k-tile fastest so the X rows stay in its L2.
    const int nwg = gridDim.x;
    int tile;
    {
        const int b = blockIdx.x, q = nwg / 8, r = nwg % 8, xcd = b % 8, i = b / 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    int mt = tile / ktiles;
    const int mt_global = mt;
    const int kt = tile % ktiles;
    int grp = 0;
    if (MODE == 2) {   // m-tiles are laid out group after group; every group is padded to whole tiles
        const long long s01 = (long long)N * Wo, s23 = (long long)N * (Ho - 2);
        const int t01 = (int)((s01 + BM - 1) / BM), t23 = (int)((s23 + BM - 1) / BM);
        if (mt < t01) { grp = 0; }
        else if (mt < 2 * t01) { grp = 1; mt -= t01; }
        else if (mt < 2 * t01 + t23) { grp = 2; mt -= 2 * t01; }
        else { grp = 3; mt -= 2 * t01 + t23; }
        M = grp < 2 ? s01 : s23;
    }
    const long long m0 = (long long)mt * BM;
    const int k0 = kt * BN;

    // ---- loader state: per instruction j the lane's global row pointer (swizzled slot folded in)
    const int lrow8 = lane >> 3, lslot = lane & 7;
    const unsigned char* xsrc[XI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int row = (j * 4 + wave) * 8 + lrow8;
        long long m = m0 + row;
        if (m >= M) m = M - 1;   // tail rows: load something valid, masked at the store
        long long n; int p, q;
        pix_decode<MODE>(m, Ho, Wo, grp, n, p, q);
        xsrc[j] = (const unsigned char*)x + (((n * Hi + p) * Wi + q) * (long long)C + swz(row, lslot) * 8) * 2;
    }
    const unsigned char* wsrc[WI];
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int row = (j * 4 + wave) * 8 + lrow8;
        wsrc[j] = (const unsigned char*)w + (((long long)(k0 + row) * 9) * C + swz(row, lslot) * 8) * 2;
    }

    auto stage = [&](int buf, int tapj, int c0) {
        unsigned char* sx = smem + buf * STAGE;
        unsigned char* sw = sx + XBYTES;
        // MODE 2: j-th live tap of this edge group: top (2,j), bottom (0,j), left (j,2), right (j,0)
        const int tap = MODE != 2 ? tapj : (grp == 0 ? 6 + tapj : (grp == 1 ? tapj : (grp == 2 ? 3 * tapj + 2 : 3 * tapj)));
        const long long xo = (((long long)(tap / 3) * Wi + (tap % 3)) * C + c0) * 2;
        const long long wo = ((long long)tap * C + c0) * 2;
#pragma unroll
        for (int j = 0; j < XI; ++j) glds16(xsrc[j] + xo, sx + (j * 4 + wave) * 8 * ROWB);
#pragma unroll
        for (int j = 0; j < WI; ++j) glds16(wsrc[j] + wo, sw + (j * 4 + wave) * 8 * ROWB);
    };

    f32x16 acc[TN][2];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // fragment read offsets (bytes within a stage buffer), one per 32-row tile; kk adds 2 slots
    const int fr = lane & 31, fh = lane >> 5;
    int xoff[2], woff[TN];
#pragma unroll
    for (int b = 0; b < 2; ++b) xoff[b] = (wm * 64 + b * 32 + fr) * ROWB;
#pragma unroll
    for (int a = 0; a < TN; ++a) woff[a] = XBYTES + (wn * WN + a * 32 + fr) * ROWB;
    const int xsw[2] = {((wm * 64 + fr) >> 1) & 7, ((wm * 64 + 32 + fr) >> 1) & 7};
    int wsw[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a) wsw[a] = ((wn * WN + a * 32 + fr) >> 1) & 7;

    const int csteps = C / BK, T = (MODE == 2 ? 3 : 9) * csteps;
    stage(0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int tap = 0, cs = 0;
    for (int t = 0; t < T; ++t) {
        const int buf = t & 1;
        int ntap = tap, ncs = cs + 1;
        if (ncs == csteps) { ncs = 0; ntap = tap + 1; }
        if (t + 1 < T) stage(buf ^ 1, ntap, ncs * BK);
        const unsigned char* sb = smem + buf * STAGE;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            h16x8 xf[2], wf[TN];
            const int ls = kk * 2 + fh;
#pragma unroll
            for (int b = 0; b < 2; ++b) xf[b] = *(const h16x8*)(sb + xoff[b] + ((ls ^ xsw[b]) << 4));
#pragma unroll
            for (int a = 0; a < TN; ++a) wf[a] = *(const h16x8*)(sb + woff[a] + ((ls ^ wsw[a]) << 4));
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = kp_mfma16(wf[a], xf[b], acc[a][b], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        tap = ntap; cs = ncs;
    }

    // ---- epilogue: D[k_out][pixel]; lane: pixel = fr, k_out = 8*(reg>>2) + 4*fh + (reg&3)
    const long long prow = BRED ? (long long)mt_global * 2 + wm : m0 / 64 + wm;   // statistics row of this wave (64 pixels)
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int kb = k0 + wn * WN + a * 32;
        float s[16], sq[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = sq[i] = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const long long m = m0 + wm * 64 + b * 32 + fr;
            const bool ok = m < M;
            long long opix = m, rpix = 0;
            if (MODE != 0 && ok) {
                long long n; int p, q;
                pix_decode<MODE>(m, Ho, Wo, grp, n, p, q);
                opix = (n * Ho + p) * Wo + q;
                if constexpr (BRED) {   // clamped source pixel in the unpadded map [Ho - 2][Wo - 2]
                    const int sp = p < 1 ? 0 : (p > Ho - 2 ? Ho - 3 : p - 1), sqx = q < 1 ? 0 : (q > Wo - 2 ? Wo - 3 : q - 1);
                    rpix = (n * (Ho - 2) + sp) * (Wo - 2) + sqx;
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kq = kb + 8 * g + 4 * fh;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[a][b][g * 4 + e] + (bias ? bias[kq + e] : 0.f);
                    if (!BRED && ok) { s[g * 4 + e] += v[e]; sq[g * 4 + e] += v[e] * v[e]; }
                }
                if (ok) {
                    uint2 pk;
                    pk.x = pack_h16x2(v[0], v[1]);
                    pk.y = pack_h16x2(v[2], v[3]);
                    *reinterpret_cast<uint2*>(y + opix * K + kq) = pk;
                    if constexpr (BRED) {   // on the stored (rounded) values, the expressions of kp_bn_act_bwd_reduce
                        float gr[4], yv[4];
                        unpack_h16x2(pk.x, gr[0], gr[1]); unpack_h16x2(pk.y, gr[2], gr[3]);
                        const uint2 yw = *reinterpret_cast<const uint2*>(rb.yred + rpix * K + kq);
                        unpack_h16x2(yw.x, yv[0], yv[1]); unpack_h16x2(yw.y, yv[2], yv[3]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool aff = rb.mean != nullptr;
                            const float mu = aff ? rb.mean[kq + e] : 0.f, is = aff ? rb.invstd[kq + e] : 1.f;
                            const float ga = aff ? rb.gamma[kq + e] : 1.f, be = aff ? rb.beta[kq + e] : -0.f;
                            const float xh = (yv[e] - mu) * is;
                            const float z = xh * ga + be;
                            const float gh = gr[e] * act_grad(z, rb.slope);
                            s[g * 4 + e] += gh;
                            sq[g * 4 + e] += gh * xh;
                        }
                    }
                }
            }
        }
        if (partials && m0 + wm * 64 < M) {
            int r1, r2;
            const float ts = reduce16_over32(s, lane, r1);
            const float tq = reduce16_over32(sq, lane, r2);
            if ((lane & 1) == 0) {
                const int k = kb + 8 * (r1 >> 2) + 4 * fh + (r1 & 3);
                partials[(prow * 2 + 0) * K + k] = ts;
                partials[(prow * 2 + 1) * K + k] = tq;
            }
        }
    }
}


}  // namespace

int kp_conv_patch_launch(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hib,
                         int Wib, int ioff, int Hob, int Wob, int ooff, int H, int W, int C, int K, hipStream_t st);
int kp_conv_pair_launch(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hib,
                        int Wib, int ioff, int Hob, int Wob, int ooff, int H, int W, int C, int K, hipStream_t st);

template <int MODE>
static int launch_linear(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi,
                         int C, int K, hipStream_t st) {
    KP_QUERY_RETURN();   // statistics rows of the linear kernel: the per-64-pixel default set by kp_conv_fwd
    const int Ho = Hi - 2, Wo = Wi - 2;
    long long M, mtiles;
    if (MODE == 2) {
        const long long s01 = (long long)N * Wo, s23 = (long long)N * (Ho - 2);
        M = 2 * s01 + 2 * s23;
        mtiles = 2 * ((s01 + BM - 1) / BM) + 2 * ((s23 + BM - 1) / BM);
    } else {
        M = MODE == 1 ? (long long)N * (2 * Wo + 2 * (Ho - 2)) : (long long)N * Ho * Wo;
        mtiles = (M + BM - 1) / BM;
    }
    // statistics rows are per 64 pixels: a half-empty last tile would leave the odd row unwritten
    if (stats && (MODE != 0 || (M % 64) != 0)) return KP_E_UNSUPPORTED;
    const int bn = (K % 128 == 0) ? 128 : 64;
    const int ktiles = K / bn;
    if (mtiles * ktiles > 0x7fffffffLL) return KP_E_UNSUPPORTED;
    const unsigned grid = (unsigned)(mtiles * ktiles);
    if (bn == 128)
        conv3x3_mfma_k<128, MODE><<<grid, TPB, 0, st>>>((const h16_t*)x, (const h16_t*)w, bias, (h16_t*)y, stats, N, Hi,
                                                       Wi, C, K, (int)mtiles, ktiles);
    else
        conv3x3_mfma_k<64, MODE><<<grid, TPB, 0, st>>>((const h16_t*)x, (const h16_t*)w, bias, (h16_t*)y, stats, N, Hi,
                                                      Wi, C, K, (int)mtiles, ktiles);
    return kp_check_launch("conv3x3_mfma");
}

extern kp_knob_t g_kp_conv_patch;   // kp_api.hip: kp_config(KP_CFG_CONV_PATCH, v); 0 the linear implicit-GEMM kernel, 1 (default) the LDS-patch kernel,
static int patch_mode() {            // 2 its pair-tile generation first (measured equal: opt-in, docs/experiments/notebook_r01-r04.md 3.10)
    const int v = g_kp_conv_patch;
    return v < 0 ? 1 : v;
}

// LDS-patch kernels over a 16x16-tileable window: the pair kernel (kp_conv_pair.hip: two pixel tiles per workgroup, 128 x 64 wave
// tile) where its tile pairing and statistics layout apply, else the first-generation kernel (kp_conv_patch.hip)
static int patch_launch(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hib, int Wib, int ioff,
                        int Hob, int Wob, int ooff, int H, int W, int C, int K, hipStream_t st) {
    if (patch_mode() >= 2) {
        const int rc = kp_conv_pair_launch(x, w, bias, y, stats, N, Hib, Wib, ioff, Hob, Wob, ooff, H, W, C, K, st);
        if (rc != KP_E_UNSUPPORTED) return rc;
    }
    return kp_conv_patch_launch(x, w, bias, y, stats, N, Hib, Wib, ioff, Hob, Wob, ooff, H, W, C, K, st);
}

// Dispatch: (a) output a multiple of 16x16 -> LDS-patch kernel; (b) output = 16x16 multiple + a 1-pixel ring
// (the data-gradient over the padded domain) -> patch kernel on the interior + linear kernel on the ring;
// (c) anything else -> linear implicit-GEMM kernel.
int kp_conv_fwd_mfma(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hi, int Wi,
                     int C, int K, int R, hipStream_t st) {
    if (R != 3 || C % BK != 0 || K % 64 != 0) return KP_E_UNSUPPORTED;
    const int Ho = Hi - 2, Wo = Wi - 2;
    if (patch_mode()) {
        if (Ho % 16 == 0 && Wo % 16 == 0) {
            int rc = patch_launch(x, w, bias, y, stats, N, Hi, Wi, 0, Ho, Wo, 0, Ho, Wo, C, K, st);
            if (rc != KP_E_UNSUPPORTED) return rc;
        } else if (!stats && Ho > 2 && Wo > 2 && (Ho - 2) % 16 == 0 && (Wo - 2) % 16 == 0) {
            int rc = patch_launch(x, w, bias, y, nullptr, N, Hi, Wi, 1, Ho, Wo, 1, Ho - 2, Wo - 2, C, K, st);
            if (rc == KP_OK) return launch_linear<1>(x, w, bias, y, nullptr, N, Hi, Wi, C, K, st);
            if (rc != KP_E_UNSUPPORTED) return rc;
        }
    }
    return launch_linear<0>(x, w, bias, y, stats, N, Hi, Wi, C, K, st);
}

// Data-gradient of a 3x3 layer: gx[N][H+2][W+2][C] = valid correlation of dy[N][H+4][W+4][K] with the flipped pack,
// where dy is KNOWN to carry a zero border of 2 pixels (written by kp_bn_act_bwd_apply).  Interior by the LDS-patch
// kernel, border ring with 3 taps per pixel.  KP_E_UNSUPPORTED -> the caller falls back to kp_conv_fwd semantics.
int kp_conv_dgrad_mfma(const void* dy, const void* w, void* gx, int N, int H, int W, int K, int C, hipStream_t st) {
    if (K % BK != 0 || C % 64 != 0 || H % 16 != 0 || W % 16 != 0 || H < 16 || W < 16 || !patch_mode()) return KP_E_UNSUPPORTED;
    int rc = patch_launch(dy, w, nullptr, gx, nullptr, N, H + 4, W + 4, 1, H + 2, W + 2, 1, H, W, K, C, st);
    if (rc != KP_OK) return rc;
    return launch_linear<2>(dy, w, nullptr, gx, nullptr, N, H + 4, W + 4, K, C, st);
}

// The ring with the BatchNorm-backward sums of the layer below (BRED): rows = kp_conv_ring_bred_rows, written into `rpart`
// (zeroed here first: a group's last tile may leave its second row untouched).
long long kp_conv_ring_bred_rows(int N, int H, int W) {
    const long long s01 = (long long)N * (W + 2), s23 = (long long)N * H;
    return 2 * (2 * ((s01 + BM - 1) / BM) + 2 * ((s23 + BM - 1) / BM));
}
int kp_conv_ring_dgrad_bred(const void* dy, const void* w, void* gx, const void* yred, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, float slope, float* rpart, int N, int H, int W, int K, int C,
                            hipStream_t st) {
    if (K % BK != 0 || C % 64 != 0) return KP_E_UNSUPPORTED;
    const int Hi = H + 4, Wi = W + 4;
    const long long mtiles = kp_conv_ring_bred_rows(N, H, W) / 2;
    const int bn = (C % 128 == 0) ? 128 : 64;
    const int ktiles = C / bn;
    if (mtiles * ktiles > 0x7fffffffLL) return KP_E_UNSUPPORTED;
    if (hipMemsetAsync(rpart, 0, (size_t)mtiles * 2 * 2 * C * sizeof(float), st) != hipSuccess) return KP_E_LAUNCH;
    const unsigned grid = (unsigned)(mtiles * ktiles);
    const RingBred rb{(const h16_t*)yred, mean, invstd, gamma, beta, slope};
    if (bn == 128)
        conv3x3_mfma_k<128, 2, true><<<grid, TPB, 0, st>>>((const h16_t*)dy, (const h16_t*)w, nullptr, (h16_t*)gx, rpart, N, Hi, Wi, K, C, (int)mtiles, ktiles, rb);
    else
        conv3x3_mfma_k<64, 2, true><<<grid, TPB, 0, st>>>((const h16_t*)dy, (const h16_t*)w, nullptr, (h16_t*)gx, rpart, N, Hi, Wi, K, C, (int)mtiles, ktiles, rb);
    return kp_check_launch("conv3x3_ring_bred");
}
