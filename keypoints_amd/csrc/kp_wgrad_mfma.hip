// bf16 MFMA weight-gradient of the 3x3 convolution (gfx950):
//     scratch[k][tap][c] += sum_{pixels m} dY[m][k] * X[pix(m)+tap][c]           (fp32 accumulate)
// GEMM view per tap: rows = output channels k, cols = input channels c, reduction = pixels.
// Both operands are stored pixel-major in HBM ([pixel][channel], channel contiguous), i.e. the reduction
// index is the SLOW one.  Tiles are DMA'd to LDS as they lie (global_load_lds_dwordx4) and the MFMA
// fragments (8 consecutive pixels of one channel per lane) are produced by the hardware transpose read
// ds_read_b64_tr_b16: per 16-lane group it returns to lane i the 4 rows of column i of a [4 rows][16 cols]
// block, lane t supplying the address of (row t/4, 8-byte chunk t%4).
// Workgroup tile TK x TC channels (64 or 128 each), 64 pixels per stage, 2x2 wavefronts, double-buffered
// LDS; 64-byte blocks of every LDS row are XOR-swizzled by the pixel row so the 4 rows of a transpose
// read fall on 4 different bank groups (conflict-free).  Split over pixels across workgroups; every split
// stores its tile into ITS OWN partial gradient (scratch[split][K][9][C], plain stores, nothing to zero), which
// kp_wgrad_unpack sums in a fixed order: deterministic like the tap-reuse kernel (this one used fp32 atomics).
#include "kp_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short h16x8;
typedef __attribute__((ext_vector_type(4))) short h16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int TPB = 256, PIX = 64;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gsrc, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

template <int P>   // row pitch in bytes (128 or 256): swizzle of the 64-B block index by the pixel row
__device__ __forceinline__ int blk_swz(int row) { return P == 256 ? (row & 3) : ((row >> 1) & 1); }

template <int P>
__device__ __forceinline__ h16x8 tr_frag(const unsigned char* tile, int prow, int colbyte) {
    // 8 consecutive pixels (prow..prow+7 handled as two 4-row transpose reads) of this lane's channel
    h16x8 r;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int row = prow + 4 * u;   // row of THIS lane's address (already includes t>>2)
        const int off = row * P + ((((colbyte >> 6) ^ blk_swz<P>(row)) << 6) | (colbyte & 63));
        h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS h16x4*)(tile + off));
        r[4 * u + 0] = v[0]; r[4 * u + 1] = v[1]; r[4 * u + 2] = v[2]; r[4 * u + 3] = v[3];
    }
    return r;
}

template <int TK, int TC>
__global__ __launch_bounds__(TPB, 2) void wgrad3x3_mfma_k(const h16_t* __restrict__ x, const h16_t* __restrict__ dy,
                                                          float* __restrict__ scratch, int Ho, int Wo, int C, int K,
                                                          int dy_pad, int lgWo, int lgHo, int ctiles, int ktiles,
                                                          int stages_per_split, int total_stages) {
    constexpr int PK = TK * 2, PC = TC * 2;                  // LDS row pitches (bytes)
    constexpr int DYB = PIX * PK, XB = PIX * PC, STAGE = DYB + XB;
    constexpr int RPI_K = 1024 / PK, RPI_C = 1024 / PC;      // pixel rows per glds instruction
    constexpr int NI_K = PIX / RPI_K / 4, NI_C = PIX / RPI_C / 4;   // instructions per wave per stage
    constexpr int LPR_K = PK / 16, LPR_C = PC / 16;          // lanes per row
    constexpr int AK = TK / 64, AC = TC / 64;                // 32x32 MFMA tiles per wave along k / c
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wc = wave & 1;
    const int Hi = Ho + 2, Wi = Wo + 2, Hd = Ho + 2 * dy_pad, Wd = Wo + 2 * dy_pad;

    // XCD-aware order: tap fastest, then c-tile, then k-tile, split slowest, contiguous per XCD
    const int nwg = gridDim.x;
    int id;
    {
        const int b = blockIdx.x, q = nwg / 8, r = nwg % 8, xcd = b % 8, i = b / 8;
        id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int tap = id % 9;
    const int ct = (id / 9) % ctiles;
    const int kt = (id / (9 * ctiles)) % ktiles;
    const int sp = id / (9 * ctiles * ktiles);
    const int k0 = kt * TK, c0 = ct * TC;
    const int tr_ = tap / 3, ts_ = tap % 3;
    const int s_beg = sp * stages_per_split;
    const int s_end = (s_beg + stages_per_split < total_stages) ? s_beg + stages_per_split : total_stages;
    if (s_beg >= s_end) return;

    // loader lane roles
    const int krow = lane / LPR_K, kslot = lane % LPR_K;     // row within instruction, 16-B slot within row
    const int crow = lane / LPR_C, cslot = lane % LPR_C;

    auto stage = [&](int buf, int s) {
        unsigned char* sd = smem + buf * STAGE;
        unsigned char* sx = sd + DYB;
        const int mbase = s * PIX;
#pragma unroll
        for (int j = 0; j < NI_K; ++j) {
            const int prow = (j * 4 + wave) * RPI_K + krow;
            const int m = mbase + prow;
            const int q = m & (Wo - 1), t = m >> lgWo, p = t & (Ho - 1), n = t >> lgHo;
            const int lblk = (kslot >> 2) ^ blk_swz<PK>(prow);
            const long long off = ((((long long)n * Hd + p + dy_pad) * Wd + q + dy_pad) * K + k0) * 2 + lblk * 64 + (kslot & 3) * 16;
            glds16((const unsigned char*)dy + off, sd + (j * 4 + wave) * 1024);
        }
#pragma unroll
        for (int j = 0; j < NI_C; ++j) {
            const int prow = (j * 4 + wave) * RPI_C + crow;
            const int m = mbase + prow;
            const int q = m & (Wo - 1), t = m >> lgWo, p = t & (Ho - 1), n = t >> lgHo;
            const int lblk = (cslot >> 2) ^ blk_swz<PC>(prow);
            const long long off = ((((long long)n * Hi + p + tr_) * Wi + q + ts_) * C + c0) * 2 + lblk * 64 + (cslot & 3) * 16;
            glds16((const unsigned char*)x + off, sx + (j * 4 + wave) * 1024);
        }
    };

    f32x16 acc[AK][AC];
#pragma unroll
    for (int a = 0; a < AK; ++a)
#pragma unroll
        for (int b = 0; b < AC; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    // transpose-read lane roles: t = lane&15 -> (row t>>2, chunk t&3); g = (lane>>4)&1 -> 16-channel half;
    // h = lane>>5 -> pixels 8h..8h+7 of the 16-pixel sub-step
    const int t16 = lane & 15, g16 = (lane >> 4) & 1, h = lane >> 5;
    const int prow0 = 8 * h + (t16 >> 2);
    int kcol[AK], ccol[AC];
#pragma unroll
    for (int a = 0; a < AK; ++a) kcol[a] = (wk * (TK / 2) + a * 32 + 16 * g16 + (t16 & 3) * 4) * 2;
#pragma unroll
    for (int b = 0; b < AC; ++b) ccol[b] = (wc * (TC / 2) + b * 32 + 16 * g16 + (t16 & 3) * 4) * 2;

    stage(0, s_beg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = s_beg; s < s_end; ++s) {
        const int buf = (s - s_beg) & 1;
        if (s + 1 < s_end) stage(buf ^ 1, s + 1);
        const unsigned char* sd = smem + buf * STAGE;
        const unsigned char* sx = sd + DYB;
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            h16x8 af[AK], bfr[AC];
#pragma unroll
            for (int a = 0; a < AK; ++a) af[a] = tr_frag<PK>(sd, ss * 16 + prow0, kcol[a]);
#pragma unroll
            for (int b = 0; b < AC; ++b) bfr[b] = tr_frag<PC>(sx, ss * 16 + prow0, ccol[b]);
#pragma unroll
            for (int a = 0; a < AK; ++a)
#pragma unroll
                for (int b = 0; b < AC; ++b)
                    acc[a][b] = kp_mfma16(af[a], bfr[b], acc[a][b], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // D[k][c]: lane col = c (lane&31), rows k = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    float* part = scratch + (long long)sp * K * 9 * C;
#pragma unroll
    for (int a = 0; a < AK; ++a)
#pragma unroll
        for (int b = 0; b < AC; ++b) {
            const int cc = c0 + wc * (TC / 2) + b * 32 + (lane & 31);
            const int kb = k0 + wk * (TK / 2) + a * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int kk = kb + (i & 3) + 8 * (i >> 2);
                part[((long long)kk * 9 + tap) * C + cc] = acc[a][b][i];
            }
        }
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

}  // namespace

// shapes this per-tap kernel takes; the pixel splits (= partial gradient buffers) of a shape: also asked by
// kp_conv_wgrad_partials
namespace {
struct WgPlan { int TKs, TCs, ktiles, ctiles, total_stages, sps; long long tiles, splits; };
WgPlan wg_plan(int N, int Ho, int Wo, int C, int K) {
    WgPlan p;
    const long long M = (long long)N * Ho * Wo;
    p.TKs = (K % 128 == 0) ? 128 : 64; p.TCs = (C % 128 == 0) ? 128 : 64;
    p.ktiles = K / p.TKs; p.ctiles = C / p.TCs;
    p.total_stages = (int)(M / PIX);
    p.tiles = 9LL * p.ktiles * p.ctiles;
    long long splits = (1536 + p.tiles - 1) / p.tiles;
    if (splits > p.total_stages / 8) splits = p.total_stages / 8;
    if (splits < 1) splits = 1;
    p.sps = (int)((p.total_stages + splits - 1) / splits);
    p.splits = (p.total_stages + p.sps - 1) / p.sps;   // every split has at least one stage: every partial is written
    return p;
}
}  // namespace

int kp_conv_wgrad_mfma_accepts(int N, int Ho, int Wo, int C, int K, int R) {
    if (R != 3 || C % 64 != 0 || K % 64 != 0) return 0;
    const long long M = (long long)N * Ho * Wo;
    if (!(ilog2_exact(Wo) >= 0 && ilog2_exact(Ho) >= 0 && M % PIX == 0 && M < (1LL << 31))) return 0;
    return (int)wg_plan(N, Ho, Wo, C, K).splits;
}

int kp_conv_wgrad_mfma(const void* x, const void* dy, float* scratch, int N, int Ho, int Wo, int C, int K, int R,
                       int dy_pad, hipStream_t st) {
    if (!kp_conv_wgrad_mfma_accepts(N, Ho, Wo, C, K, R)) return KP_E_UNSUPPORTED;
    const int lgWo = ilog2_exact(Wo), lgHo = ilog2_exact(Ho);
    const WgPlan pl = wg_plan(N, Ho, Wo, C, K);
    const int TKs = pl.TKs, TCs = pl.TCs, ktiles = pl.ktiles, ctiles = pl.ctiles, total_stages = pl.total_stages, sps = pl.sps;
    const long long tiles = pl.tiles, splits = pl.splits;
    const unsigned grid = (unsigned)(tiles * splits);
#define KP_WG_LAUNCH(TKv, TCv)                                                                                      \
    wgrad3x3_mfma_k<TKv, TCv><<<grid, TPB, 0, st>>>((const h16_t*)x, (const h16_t*)dy, scratch, Ho, Wo, C, K,     \
                                                   dy_pad, lgWo, lgHo, ctiles, ktiles, sps, total_stages)
    if (TKs == 128 && TCs == 128) KP_WG_LAUNCH(128, 128);
    else if (TKs == 128) KP_WG_LAUNCH(128, 64);
    else if (TCs == 128) KP_WG_LAUNCH(64, 128);
    else KP_WG_LAUNCH(64, 64);
#undef KP_WG_LAUNCH
    return kp_check_launch("wgrad3x3_mfma");
}
