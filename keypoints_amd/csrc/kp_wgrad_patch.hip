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
    const h16_t* x; const h16_t* dy; float* scratch;
    int Ho, Wo, C, K, dy_pad;
    int ctiles, ktiles, stages_per_split, total_stages, bx, by, dbg;
    long long partial_stride;   // floats between the partial gradients of consecutive pixel splits (0: one buffer, atomics)
    long long* trace;           // measurement build: s_memtime stamps of workgroup 0 / wave 0 (512 entries)
};

template <int TK, int TC, bool TRACE>
__global__ __launch_bounds__(TPB, 2) void wgrad3x3_patch_k(const WgArgs A) {
    constexpr int PK = TK * 2, PC = TC * 2;                       // LDS row pitches (bytes)
    constexpr int RPI_K = 1024 / PK, RPI_C = 1024 / PC;           // LDS rows per DMA instruction
    constexpr int NI_DY = BH * BW / RPI_K;                        // DMA instructions for the dY tile
    constexpr int NI_X = (PROWS + RPI_C - 1) / RPI_C;             // ... for the input patch (last one partial)
    constexpr int NW_X = (NI_X + NWAVE - 1) / NWAVE;   // input patch instructions rounded to 8 (LDS layout)
    // Only wavefronts 0..3 issue DMA: a SIMD holds waves w and w+4 and gives the older one (w) priority; with the loads on
    // all eight, wave w+4 finished every stage ~2000 cycles after wave w, running alone at half rate meanwhile.  With
    // the whole non-MFMA work on the priority wave, the other one has nothing but MFMAs to fill its gaps with.
    constexpr int LW = 4, ND2 = NI_DY / LW, NX2 = (NI_X + LW - 1) / LW, NT = ND2 + NX2;   // per loading wave
    constexpr int DYB = NI_DY * 1024, XB = NW_X * NWAVE * 1024, STAGE = DYB + XB;
    constexpr int LPR_K = PK / 16, LPR_C = PC / 16;               // lanes per LDS row
    constexpr int WK = TK / 32, WC = TC / 32;                     // wave grid (WK * WC == 8)
    static_assert(WK * WC == NWAVE, "one 32x32 gradient tile per wavefront");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE + 1024];   // + dummy DMA target

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
    const int lw = wave & (LW - 1);
    const bool loader = wave < LW;
    int dvoff[ND2], xvoff[NX2];
#pragma unroll
    for (int j = 0; j < ND2; ++j) {
        const int row = (j * LW + lw) * RPI_K + lane / LPR_K;   // pixel of the 8x16 block, row-major
        const int slot = lane % LPR_K;
        const int lblk = (slot >> 2) ^ blk_swz<PK>(row);
        dvoff[j] = ((row / BW) * Wd + (row % BW)) * K * 2 + lblk * 64 + (slot & 3) * 16;
    }
#pragma unroll
    for (int j = 0; j < NX2; ++j) {
        int row = (j * LW + lw) * RPI_C + lane / LPR_C;              // patch pixel, row-major over 10 x 18
        const int slot = lane % LPR_C;
        const int lblk = (slot >> 2) ^ blk_swz<PC>(row);
        if (row > PROWS - 1) row = PROWS - 1;                         // tail lanes: harmless duplicate
        xvoff[j] = ((row / PWC) * Wi + (row % PWC)) * C * 2 + lblk * 64 + (slot & 3) * 16;
    }

    // A stage's DMA = NT instructions per loading wave.  The first stage issues them in one go; afterwards the
    // instructions of stage s+1 are spread over the patch rows of stage s (one or two after each row's MFMAs), so the
    // ~100-cycle DMA issues hide behind queued MFMAs instead of stopping the waves at the top of a stage.
    u32x4 dr, xr;   // buffer descriptors of the stage being fetched (asm DMA: see dma16_asm)
    // fetch cursor: pixel block (bxi, byi) of image cn -- divisions only once per workgroup, then carries
    int cbx = s_beg % A.bx, cby = (s_beg / A.bx) % A.by, cn = s_beg / (A.bx * A.by);
    u32x4 dr2, xr2;   // ... and of the stage after it, computed late in a stage behind its MFMAs
    auto stage_addr = [&]() {   // descriptors of the cursor's block -> (dr2, xr2), then advance the cursor
        const int bxi = cbx, byi = cby, n = cn;
        if (++cbx == A.bx) { cbx = 0; if (++cby == A.by) { cby = 0; ++cn; } }
        const int y0 = byi * BH, x0 = bxi * BW;
        const unsigned char* db = (const unsigned char*)A.dy + ((((long long)n * Hd + y0 + dp) * Wd + x0 + dp) * K + k0) * 2;
        const unsigned char* xb = (const unsigned char*)A.x + ((((long long)n * Hi + y0) * Wi + x0) * C + c0) * 2;
        dr2 = rsrc_words(db);
        xr2 = rsrc_words(xb);
    };
    // (`real` = false sends the instruction to a dummy LDS target: the row loop stays branch-free, which the
    //  compiler needs to keep its software pipeline of transpose reads across rows)
    auto stage_issue = [&](int buf, auto jc, bool real) {   // instruction jc of the stage whose addresses are set
        constexpr int j = decltype(jc)::value;
        unsigned char* sd = smem + buf * STAGE;
        if constexpr (j < ND2) dma16_asm(dr, real ? sd + (j * LW + lw) * 1024 : smem + 2 * STAGE, dvoff[j]);
        else if constexpr (j < NT) {
            constexpr int jx = j - ND2;
            // (instructions past the patch's last one exist only for some lw: they land in the buffer's padding)
            dma16_asm(xr, real ? sd + DYB + (jx * LW + lw) * 1024 : smem + 2 * STAGE, xvoff[jx]);
        }
    };
    static_assert(NT <= 2 * PH, "at most two DMA instructions per patch row");
    static_assert(NX2 * LW * 1024 <= XB, "input patch instructions fit the staged buffer");

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

    int tri = 0;
    const bool tracing = TRACE && A.trace != nullptr && blockIdx.x == 0 && wave == 0;
    auto stamp = [&]() {
        if constexpr (TRACE) if (tracing) {
            const long long now = __builtin_amdgcn_s_memtime();
            if (lane == 0 && tri < 512) A.trace[tri] = now;
            ++tri;
        }
    };
    stamp();
    stage_addr();
    dr = dr2; xr = xr2;
    if (loader) static_for<0, NT>([&](auto jc) { stage_issue(0, jc, true); });
    stage_addr();   // block s_beg + 1 (fetched during the first stage)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    for (int sg = s_beg; sg < s_end; ++sg) {
        const int buf = (sg - s_beg) & 1;
        stamp();
        const bool more = sg + 1 < s_end;
        if (more) { dr = dr2; xr = xr2; }   // block sg + 1; the last stage re-reads its own block into the dummy target
                                            // (dr2 may then describe a block past the end of the tensors: never used)
        stamp();
        const unsigned char* sd = smem + buf * STAGE;
        const unsigned char* sx = sd + DYB;
        // walk the 10 patch rows; patch row p meets output rows p, p-1, p-2 through taps r = 0, 1, 2
        h16x8 a0 = {0, 0, 0, 0, 0, 0, 0, 0}, a1 = a0, a2 = a0;   // dY fragments of output rows p, p-1, p-2 (rotating)
        // fragments are read one patch row ahead of their MFMAs, so a wave keeps the matrix pipe fed on its own
        // (the second wave of a SIMD finishes a stage ~2000 cycles after the first and runs alone meanwhile)
        h16x8 an = tr_frag<PK, 0>(sd, kbase), bn[3];
        bn[0] = tr_frag<PC, 0>(sx, cbase);
        bn[1] = tr_frag<PC, 1>(sx, cbase);
        bn[2] = tr_frag<PC, 2>(sx, cbase);
        static_for<0, PH>([&](auto pc) {
            constexpr int p = decltype(pc)::value;
            a2 = a1; a1 = a0;
            if constexpr (p < BH) a0 = an;
            h16x8 b[3] = {bn[0], bn[1], bn[2]};
            if constexpr (p + 1 < PH) {
                if constexpr (p + 1 < BH) an = tr_frag<PK, (p + 1) * BW>(sd, kbase);
                bn[0] = tr_frag<PC, (p + 1) * PWC + 0>(sx, cbase);
                bn[1] = tr_frag<PC, (p + 1) * PWC + 1>(sx, cbase);
                bn[2] = tr_frag<PC, (p + 1) * PWC + 2>(sx, cbase);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                if constexpr (p < BH) acc[0 * 3 + s] = kp_mfma16(a0, b[s], acc[0 * 3 + s], 0, 0, 0);
                if constexpr (p >= 1 && p - 1 < BH) acc[1 * 3 + s] = kp_mfma16(a1, b[s], acc[1 * 3 + s], 0, 0, 0);
                if constexpr (p >= 2) acc[2 * 3 + s] = kp_mfma16(a2, b[s], acc[2 * 3 + s], 0, 0, 0);
            }
            if (loader) {   // two instructions per row from the top: the last one is issued ~3 rows before the stage ends
                stage_issue(buf ^ 1, std::integral_constant<int, 2 * p>{}, more);
                stage_issue(buf ^ 1, std::integral_constant<int, 2 * p + 1>{}, more);
                if constexpr (p == PH - 2) stage_addr();   // block sg + 2, behind this row's MFMAs
            }
        });
        stamp();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
    stamp();

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
    int cus = kp_device_cus();   // (the CU budget of the launching lane: kp_config(KP_CFG_CUS))
    if (cus <= 0) cus = 256;
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
    a.x = (const h16_t*)x; a.dy = (const h16_t*)dy; a.scratch = scratch;
    a.partial_stride = (long long)K * 9 * C;
    { extern kp_knob_t g_kp_debug; extern kp_knob_ptr_t g_kp_trace; a.dbg = g_kp_debug; a.trace = g_kp_trace; }
    if (k128 && a.trace) wgrad3x3_patch_k<128, 64, true><<<grid, TPB, 0, st>>>(a);
    else if (k128) wgrad3x3_patch_k<128, 64, false><<<grid, TPB, 0, st>>>(a);
    else wgrad3x3_patch_k<64, 128, false><<<grid, TPB, 0, st>>>(a);
    return kp_check_launch("wgrad3x3_patch");
}
