// bf16 MFMA weight-gradient of the 3x3 convolution with tap reuse -- gfx950.
//     scratch[k][tap][c] += sum_{pixels m} dY[m][k] * X[pix(m)+tap][c]           (fp32 accumulate)
//
// The per-tap kernel (kp_wgrad_mfma.hip) streams a dY tile and an X tile per tap: 32 KB of L2->LDS traffic per 64
// MFMAs (64 FLOP/B).  Here one workgroup owns a (TK x TC) block of the gradient for ALL 9 taps (9 x 16 fp32
// accumulators per lane) and walks 8 x 16 pixel blocks: per block it stages the dY tile (128 px x TK) and ONE
// 10 x 18 x TC input patch, and every tap reads the pixel operand from that patch at its (r,s) shift:
// 288 MFMAs per 55 KB staged (~310 FLOP/B).  Patch rows are shared between taps, so a wavefront loads the three
// column-shifted B fragments of a patch row once and uses them against the dY fragments of the three output rows
// that touch it: 8 transpose reads per 9 MFMAs.
// Both operands are pixel-major in HBM (reduction index slow), so tiles are DMA'd to LDS as they lie
// (buffer_load ... lds) and MFMA fragments (8 consecutive pixels of one channel per lane) come from the hardware
// transpose read ds_read_b64_tr_b16; 64-byte blocks of every LDS row are XOR-swizzled by the row index so the 4
// rows of a transpose read hit 4 different bank groups for every tap shift.
// Split over pixel blocks across workgroups; partial tiles are combined with hardware fp32 atomics.
#include "kp_mfma_util.h"

namespace {

constexpr int TPB = 512, NWAVE = 8;
constexpr int BH = 8, BW = 16;                 // pixel block
constexpr int PH = BH + 2, PWC = BW + 2;       // input patch
constexpr int PROWS = PH * PWC;                // 180 patch pixels

struct WgArgs {
    const bf16_t* x; const bf16_t* dy; float* scratch;
    int Ho, Wo, C, K, dy_pad;
    int ctiles, ktiles, stages_per_split, total_stages, bx, by, dbg;
    long long partial_stride;   // floats between the partial gradients of consecutive pixel splits (0: one buffer, atomics)
};

template <int TK, int TC>
__global__ __launch_bounds__(TPB, 2) void wgrad3x3_patch_k(const WgArgs A) {
    constexpr int PK = TK * 2, PC = TC * 2;                       // LDS row pitches (bytes)
    constexpr int RPI_K = 1024 / PK, RPI_C = 1024 / PC;           // LDS rows per DMA instruction
    constexpr int NI_DY = BH * BW / RPI_K;                        // DMA instructions for the dY tile
    constexpr int NI_X = (PROWS + RPI_C - 1) / RPI_C;             // ... for the input patch (last one partial)
    constexpr int NW_DY = NI_DY / NWAVE, NW_X = (NI_X + NWAVE - 1) / NWAVE;   // per wave
    constexpr int DYB = NI_DY * 1024, XB = NW_X * NWAVE * 1024, STAGE = DYB + XB;
    constexpr int LPR_K = PK / 16, LPR_C = PC / 16;               // lanes per LDS row
    constexpr int WK = TK / 32, WC = TC / 32;                     // wave grid (WK * WC == 8)
    static_assert(WK * WC == NWAVE, "one 32x32 gradient tile per wavefront");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave / WC, wc = wave % WC;
    const int Ho = A.Ho, Wo = A.Wo, C = A.C, K = A.K, dp = A.dy_pad;
    const int Hi = Ho + 2, Wi = Wo + 2, Hd = Ho + 2 * dp, Wd = Wo + 2 * dp;

    // XCD-aware order: (c-tile, k-tile) fastest, pixel split slowest, contiguous per XCD -> the workgroups that
    // run together on an XCD read the same pixel blocks (L2 reuse of both operands)
    int id;
    {
        const int nwg = gridDim.x, b = blockIdx.x, q = nwg / 8, r = nwg % 8, xcd = b % 8, i = b / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int ct = id % A.ctiles;
    const int kt = (id / A.ctiles) % A.ktiles;
    const int sp = id / (A.ctiles * A.ktiles);
    const int k0 = kt * TK, c0 = ct * TC;
    const int s_beg = sp * A.stages_per_split;
    const int s_end = (s_beg + A.stages_per_split < A.total_stages) ? s_beg + A.stages_per_split : A.total_stages;
    if (s_beg >= s_end) return;

    // ---- per-lane DMA offsets (relative to the pixel block's origin) -----------------------------------------
    int dvoff[NW_DY], xvoff[NW_X];
#pragma unroll
    for (int j = 0; j < NW_DY; ++j) {
        const int row = (j * NWAVE + wave) * RPI_K + lane / LPR_K;   // pixel of the 8x16 block, row-major
        const int slot = lane % LPR_K;
        const int lblk = (slot >> 2) ^ blk_swz<PK>(row);
        dvoff[j] = ((row / BW) * Wd + (row % BW)) * K * 2 + lblk * 64 + (slot & 3) * 16;
    }
#pragma unroll
    for (int j = 0; j < NW_X; ++j) {
        int row = (j * NWAVE + wave) * RPI_C + lane / LPR_C;         // patch pixel, row-major over 10 x 18
        const int slot = lane % LPR_C;
        const int lblk = (slot >> 2) ^ blk_swz<PC>(row);
        if (row > PROWS - 1) row = PROWS - 1;                         // tail lanes: harmless duplicate
        xvoff[j] = ((row / PWC) * Wi + (row % PWC)) * C * 2 + lblk * 64 + (slot & 3) * 16;
    }

    auto stage = [&](int buf, int sg) {
        const int bxi = sg % A.bx, byi = (sg / A.bx) % A.by, n = sg / (A.bx * A.by);
        const int y0 = byi * BH, x0 = bxi * BW;
        const unsigned char* db = (const unsigned char*)A.dy + ((((long long)n * Hd + y0 + dp) * Wd + x0 + dp) * K + k0) * 2;
        const unsigned char* xb = (const unsigned char*)A.x + ((((long long)n * Hi + y0) * Wi + x0) * C + c0) * 2;
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)db, 0, 0x7ffffffe, 0x00020000);
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, 0x7ffffffe, 0x00020000);
        unsigned char* sd = smem + buf * STAGE;
        unsigned char* sx = sd + DYB;
#pragma unroll
        for (int j = 0; j < NW_DY; ++j) dma16(dr, sd + (j * NWAVE + wave) * 1024, dvoff[j]);
#pragma unroll
        for (int j = 0; j < NW_X; ++j) dma16(xr, sx + (j * NWAVE + wave) * 1024, xvoff[j]);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // transpose-read lane roles: t16 = lane&15 -> (row t16>>2, 8-byte chunk t16&3); g16 -> 16-channel half;
    // h = lane>>5 -> pixels 8h..8h+7 of a 16-pixel row
    const int t16 = lane & 15, g16 = (lane >> 4) & 1, h = lane >> 5;
    const int lrow = 8 * h + (t16 >> 2);
    const int kcol = (wk * 32 + 16 * g16 + (t16 & 3) * 4) * 2;
    const int ccol = (wc * 32 + 16 * g16 + (t16 & 3) * 4) * 2;
    int kbase[4], cbase[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) { kbase[v] = tr_base<PK>(v, lrow, kcol); cbase[v] = tr_base<PC>(v, lrow, ccol); }

    stage(0, s_beg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    for (int sg = s_beg; sg < s_end; ++sg) {
        const int buf = (sg - s_beg) & 1;
        if (sg + 1 < s_end) stage(buf ^ 1, sg + 1);
        const unsigned char* sd = smem + buf * STAGE;
        const unsigned char* sx = sd + DYB;
        // walk the 10 patch rows; patch row p meets output rows p, p-1, p-2 through taps r = 0, 1, 2
        bf16x8 a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0, a2 = a0;   // dY fragments of output rows p, p-1, p-2 (rotating)
        static_for<0, PH>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            a2 = a1; a1 = a0;
            if constexpr (p < BH) a0 = tr_frag<PK, p * BW>(sd, kbase);
            bf16x8 b[3];
            b[0] = tr_frag<PC, p * PWC + 0>(sx, cbase);
            b[1] = tr_frag<PC, p * PWC + 1>(sx, cbase);
            b[2] = tr_frag<PC, p * PWC + 2>(sx, cbase);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if constexpr (p < BH) acc[0 * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b[s], acc[0 * 3 + s], 0, 0, 0);
                if constexpr (p >= 1 && p - 1 < BH) acc[1 * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b[s], acc[1 * 3 + s], 0, 0, 0);
                if constexpr (p >= 2) acc[2 * 3 + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b[s], acc[2 * 3 + s], 0, 0, 0);
            }
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

    // D[k][c]: lane col = c (lane&31), rows k = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const int cc = c0 + wc * 32 + (lane & 31);
    const int kb = k0 + wk * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = kb + (i & 3) + 8 * (i >> 2);
            float* dst = A.scratch + (long long)sp * A.partial_stride + ((long long)kk * 9 + t) * C + cc;
            if (A.dbg & 4) { if (acc[t][i] == 12345.678f) A.scratch[0] = 1.f; }   // ablation: keep the accumulators live
            else if (A.partial_stride) *dst = acc[t][i];     // this split's own buffer: plain coalesced store
            else unsafeAtomicAdd(dst, acc[t][i]);
        }
}

}  // namespace

// Launch plan shared by the query (kp_conv_wgrad_partials) and the launch.  KP_E_UNSUPPORTED when the shape is not
// eligible (then the per-tap kernel takes it).
static int plan(int N, int Ho, int Wo, int C, int K, int dy_pad, WgArgs& a, unsigned& grid, long long& splits, bool& k128) {
    if (C % 64 != 0 || K % 64 != 0 || Ho % BH != 0 || Wo % BW != 0) return KP_E_UNSUPPORTED;
    k128 = (K % 128 == 0);
    if (!k128 && C % 128 != 0) return KP_E_UNSUPPORTED;            // (64 x 64) blocks: per-tap kernel
    // 32-bit DMA offsets inside one pixel block
    if ((long long)(PH + 1) * (Wo + 2 * (dy_pad > 1 ? dy_pad : 1)) * (C > K ? C : K) * 2 >= 0x7fffffffLL)
        return KP_E_UNSUPPORTED;
    a.Ho = Ho; a.Wo = Wo; a.C = C; a.K = K; a.dy_pad = dy_pad;
    a.bx = Wo / BW; a.by = Ho / BH;
    const long long total = (long long)N * a.bx * a.by;
    if (total >= (1LL << 31)) return KP_E_UNSUPPORTED;
    a.total_stages = (int)total;
    const int TKs = k128 ? 128 : 64, TCs = k128 ? 64 : 128;
    a.ktiles = K / TKs; a.ctiles = C / TCs;
    const long long tiles = (long long)a.ktiles * a.ctiles;
    static int cus = 0;
    if (!cus) { cus = kp_device_cus(); if (cus <= 0) cus = 256; }
    // one workgroup per CU (>= 112 KB of LDS each); split the pixel blocks so that the grid fills the chip
    splits = (cus + tiles - 1) / tiles;
    if (splits > total / 4) splits = total / 4;
    if (splits < 1) splits = 1;
    a.stages_per_split = (int)((total + splits - 1) / splits);
    splits = (total + a.stages_per_split - 1) / a.stages_per_split;
    grid = (unsigned)(tiles * splits);
    return KP_OK;
}

// Number of partial gradient buffers kp_conv_wgrad_patch writes for this shape (0: shape not eligible).
long long kp_conv_wgrad_patch_partials(int N, int Ho, int Wo, int C, int K, int dy_pad) {
    WgArgs a; unsigned grid; long long splits; bool k128;
    return plan(N, Ho, Wo, C, K, dy_pad, a, grid, splits, k128) == KP_OK ? splits : 0;
}

// scratch: `splits` buffers of [K][9][C] fp32, each fully written by the workgroups of one pixel split (no zeroing,
// no atomics: the sum over splits is taken in a fixed order by kp_wgrad_unpack -> bitwise reproducible gradients).
int kp_conv_wgrad_patch(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K,
                        int dy_pad, hipStream_t st) {
    WgArgs a; unsigned grid; long long splits; bool k128;
    int rc = plan(N, Ho, Wo, C, K, dy_pad, a, grid, splits, k128);
    if (rc != KP_OK) return rc;
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.scratch = scratch;
    a.partial_stride = (long long)K * 9 * C;
    { extern int g_kp_debug; a.dbg = g_kp_debug; }
    if (k128) wgrad3x3_patch_k<128, 64><<<grid, TPB, 0, st>>>(a);
    else wgrad3x3_patch_k<64, 128><<<grid, TPB, 0, st>>>(a);
    return kp_check_launch("wgrad3x3_patch");
}
