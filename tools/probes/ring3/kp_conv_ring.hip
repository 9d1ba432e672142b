// Border ring of the 3x3 data gradient (gfx950).  gx[N][H+2][W+2][C] is the gradient over the PADDED input domain; the
// LDS-patch kernel (kp_conv_patch.hip) writes its H x W interior, this kernel the one-pixel ring around it.  dy carries a
// zero border of 2, so a ring pixel has three live taps (the row / column of the filter that faces the image): per edge a
// 1-D three-tap correlation along the edge, as a GEMM of (ring pixels) x (C outputs) over 3 * K.
//
// The ring is 3 % of the layer's pixels and a third of their taps: one workgroup's reduction is only 3 * K / 64 steps long.
// The two-stage linear kernel (kp_conv_mfma.hip, MODE 2) spent it waiting for memory (every step drains vmcnt to 0), and
// its 64 KB of LDS allow 512 resident workgroups where the 128-image layers have 516 / 528 / 1032 tiles -- a second,
// almost empty round.  Here: reduction steps of 32 channels (64-byte LDS rows, 16-byte slot ^ ((row >> 2) & 3):
// conflict-free for every ds_read_b128 lane group), a ring of three 16 KB stages with a counted vmcnt (two stages in
// flight behind the one being multiplied, one barrier per step; LDS-DMA as inline asm so that hipcc does not put its own
// vmcnt(0) in front of the barrier), 48 KB of LDS, 100 VGPRs: three workgroups per CU, 768 slots.
#include "kp_mfma_util.h"

namespace {

constexpr int BM = 128, TPB = 256;
constexpr int RK = 32, RROW = RK * 2;   // reduction step (channels), bytes per LDS row
constexpr int NS = 3;

// ring pixel m of edge group grp (0 top row, 1 bottom row, 2 left column, 3 right column; corners belong to the rows)
__device__ __forceinline__ void ring_decode(int m, int Ho, int Wo, int grp, int& n, int& p, int& q) {
    if (grp < 2) { n = m / Wo; q = m - n * Wo; p = grp ? Ho - 1 : 0; }
    else { n = m / (Ho - 2); p = 1 + m - n * (Ho - 2); q = (grp == 3) ? Wo - 1 : 0; }
}

template <int BN>
__global__ __launch_bounds__(TPB, 3) void dgrad_ring_k(const h16_t* __restrict__ x, const h16_t* __restrict__ w,
                                                       h16_t* __restrict__ y, int N, int Hi, int Wi, int C, int K,
                                                       int ktiles) {
    constexpr int WN = BN / 2, TN = WN / 32;
    constexpr int XB = BM * RROW, WB = BN * RROW, STG = XB + WB;
    constexpr int XI = BM / 64, WI = BN / 64;   // DMA instructions per wave and stage (one covers 16 rows)
    constexpr int L = XI + WI;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NS * STG];

    const int Ho = Hi - 2, Wo = Wi - 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware order: each XCD takes a contiguous run of (m-tile, k-tile) pairs, k-tile fastest (the rows stay in its L2)
    const int nwg = gridDim.x;
    int tile;
    {
        const int b = blockIdx.x, q = nwg / 8, r = nwg % 8, xcd = b % 8, i = b / 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    int mt = tile / ktiles;
    const int kt = tile % ktiles;
    int grp;
    const int s01 = N * Wo, s23 = N * (Ho - 2);   // < 2^31: checked by the launcher
    {
        const int t01 = (int)((s01 + BM - 1) / BM), t23 = (int)((s23 + BM - 1) / BM);
        if (mt < t01) { grp = 0; }
        else if (mt < 2 * t01) { grp = 1; mt -= t01; }
        else if (mt < 2 * t01 + t23) { grp = 2; mt -= 2 * t01; }
        else { grp = 3; mt -= 2 * t01 + t23; }
    }
    const int M = grp < 2 ? s01 : s23;
    const int m0 = mt * BM;
    const int k0 = kt * BN;
    // 32-bit DMA offsets from the first image of the tile (a tile spans a few images at most)
    const int n0 = m0 / (grp < 2 ? Wo : Ho - 2);
    const u32x4 xr = rsrc_words(x + (long long)n0 * Hi * Wi * C);
    const u32x4 wr = rsrc_words(w + (long long)k0 * 9 * C);

    // loader: one instruction = 16 rows x 64 B; lane -> row lane >> 2, physical slot lane & 3 (logical slot in the source)
    const int lrow = lane >> 2, lslot = (lane & 3) ^ ((lane >> 4) & 3);
    int xv[XI], wv[WI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int row = (j * 4 + wave) * 16 + lrow;
        int m = m0 + row;
        if (m >= M) m = M - 1;   // tail rows: load something valid, masked at the store
        int n, p, q;
        ring_decode(m, Ho, Wo, grp, n, p, q);
        xv[j] = ((((n - n0) * Hi + p) * Wi + q) * C + lslot * 8) * 2;   // < 2^31: checked by the launcher
    }
#pragma unroll
    for (int j = 0; j < WI; ++j) {
        const int row = (j * 4 + wave) * 16 + lrow;
        wv[j] = (row * 9 * C + lslot * 8) * 2;
    }
    const int csteps = C / RK, T = 3 * csteps;
    int itap = 0, ics = 0;   // cursor of the next stage to issue
    auto issue = [&](int buf) {
        unsigned char* sx = smem + buf * STG;
        unsigned char* sw = sx + XB;
        // j-th live tap of the edge group: top (2,j), bottom (0,j), left (j,2), right (j,0)
        const int tap = grp == 0 ? 6 + itap : (grp == 1 ? itap : (grp == 2 ? 3 * itap + 2 : 3 * itap));
        const int xo = (((tap / 3) * Wi + (tap % 3)) * C + ics * RK) * 2;
        const int wo = (tap * C + ics * RK) * 2;
#pragma unroll
        for (int j = 0; j < XI; ++j) dma16_asm(xr, sx + (j * 4 + wave) * 16 * RROW, xv[j] + xo);
#pragma unroll
        for (int j = 0; j < WI; ++j) dma16_asm(wr, sw + (j * 4 + wave) * 16 * RROW, wv[j] + wo);
        if (++ics == csteps) { ics = 0; ++itap; }
    };

    f32x16 acc[TN][2];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;
    int xoff[2], woff[TN];
#pragma unroll
    for (int b = 0; b < 2; ++b) xoff[b] = (wm * 64 + b * 32 + fr) * RROW;
#pragma unroll
    for (int a = 0; a < TN; ++a) woff[a] = XB + (wn * WN + a * 32 + fr) * RROW;
    const int fsw = (fr >> 2) & 3;   // fragment rows start at multiples of 32: the swizzle depends on fr only

    issue(0);
    if (T > 1) issue(1);
    for (int t = 0; t < T; ++t) {
        // own DMAs of stage t landed (those of stage t + 1 may still be in flight), then everybody's
        if (t + 1 < T) wait_vm<L>();
        else wait_vm<0>();
        __syncthreads();
        if (t + 2 < T) issue((t + 2) % NS);   // that buffer was read in step t - 1: every wave is past the barrier
        const unsigned char* sb = smem + (t % NS) * STG;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            h16x8 xf[2], wf[TN];
            const int ls = ((kk * 2 + fh) ^ fsw) << 4;
#pragma unroll
            for (int b = 0; b < 2; ++b) xf[b] = *(const h16x8*)(sb + xoff[b] + ls);
#pragma unroll
            for (int a = 0; a < TN; ++a) wf[a] = *(const h16x8*)(sb + woff[a] + ls);
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][b] = kp_mfma16(wf[a], xf[b], acc[a][b], 0, 0, 0);
        }
    }

    // epilogue: D[k_out][pixel]; lane: pixel = fr, k_out = 8*(reg>>2) + 4*fh + (reg&3)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int m = m0 + wm * 64 + b * 32 + fr;
        if (m < M) {
            int n, p, q;
            ring_decode(m, Ho, Wo, grp, n, p, q);
            h16_t* yp = y + (((long long)n * Ho + p) * Wo + q) * K + k0 + wn * WN + 4 * fh;
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 pk;
                    pk.x = pack_h16x2(acc[a][b][g * 4 + 0], acc[a][b][g * 4 + 1]);
                    pk.y = pack_h16x2(acc[a][b][g * 4 + 2], acc[a][b][g * 4 + 3]);
                    *reinterpret_cast<uint2*>(yp + a * 32 + 8 * g) = pk;
                }
        }
    }
}

}  // namespace

// x = dy [N][Hi][Wi][C] with a zero border of 2, w = the flipped pack [K][9][C], y = gx [N][Hi-2][Wi-2][K] (ring only).
// KP_E_UNSUPPORTED -> the caller keeps the linear kernel.
int kp_conv_dgrad_ring(const void* x, const void* w, void* y, int N, int Hi, int Wi, int C, int K, hipStream_t st) {
    const int Ho = Hi - 2, Wo = Wi - 2;
    if (C % RK != 0 || K % 64 != 0 || Ho < 3 || Wo < 3) return KP_E_UNSUPPORTED;
    // 32-bit DMA offsets: a tile's rows lie within ceil(128 / edge) + 1 images of its first one
    const long long img = (long long)Hi * Wi * C * 2, edge = Ho - 2 < Wo ? Ho - 2 : Wo;
    if (img * (BM / edge + 2) >= 0x7fffffffLL || 9LL * C * K * 2 >= 0x7fffffffLL) return KP_E_UNSUPPORTED;
    KP_QUERY_RETURN();
    const long long s01 = (long long)N * Wo, s23 = (long long)N * (Ho - 2);
    const long long mtiles = 2 * ((s01 + BM - 1) / BM) + 2 * ((s23 + BM - 1) / BM);
    const int bn = (K % 128 == 0) ? 128 : 64;
    const int ktiles = K / bn;
    if (mtiles * ktiles > 0x7fffffffLL || (mtiles + 1) * BM > 0x7fffffffLL) return KP_E_UNSUPPORTED;
    const unsigned grid = (unsigned)(mtiles * ktiles);
    if (bn == 128)
        dgrad_ring_k<128><<<grid, TPB, 0, st>>>((const h16_t*)x, (const h16_t*)w, (h16_t*)y, N, Hi, Wi, C, K, ktiles);
    else
        dgrad_ring_k<64><<<grid, TPB, 0, st>>>((const h16_t*)x, (const h16_t*)w, (h16_t*)y, N, Hi, Wi, C, K, ktiles);
    return kp_check_launch("dgrad_ring");
}
