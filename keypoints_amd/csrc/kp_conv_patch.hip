// bf16 MFMA 3x3 convolution (forward / data-gradient) with an LDS-resident input patch -- gfx950.
//
// The generic implicit-GEMM kernel (kp_conv_mfma.hip) re-fetches the pixel operand for every one of the 9
// taps: 32 KB of L2->LDS traffic per 32 MFMAs = 64 FLOP/B, more than the per-CU L2 bandwidth can feed at
// full MFMA rate.  This kernel uses the structure of the convolution instead of treating it as a GEMM:
//
//   workgroup tile = 16 x 16 output pixels (one image) x BN output channels, 8 wavefronts (4 pixel x 2 channel)
//   for each 64-channel chunk of the input:                      (C/64 chunks)
//       the 18 x 18 x 64ch input patch is in LDS once (41 KB, double buffered, prefetched one chunk ahead)
//       for each of the 9 taps:                                   (one "step": 16 or 8 MFMA 32x32x16 / wave)
//           the pixel operand is read from the patch at the tap's (r,s) shift -- no new HBM/L2 traffic,
//           the weight tile W[k0..k0+BN)[tap][c0..c0+64) streams through a 4-deep LDS ring (16 KB / step).
//
// L2->LDS traffic per step drops from 32 KB to ~21 KB for twice the MFMAs (256-pixel tile): ~200 FLOP/B.
// All HBM->LDS movement is LDS-DMA (buffer_load_dwordx4 ... lds); loads stay in flight across the single
// s_barrier of a step, retired by a counted s_waitcnt vmcnt(N) (N = DMA instructions issued in this step).
// LDS rows are 128 B (64 bf16 of one pixel / one output channel); the 16-B slot index is XOR-swizzled
// (pixel: by its patch column, weights: by the row) on the DMA *source* address and on the ds_read_b128
// address, which makes every ds_read_b128 lane group hit 16 distinct bank slots for all 9 tap shifts.
// MFMA operands are swapped ("A" = weights, "B" = pixels): a lane's accumulator registers hold 4 consecutive
// output channels of one pixel -> packed 8-byte bf16 stores into the NHWC output.
// BatchNorm batch statistics (vgg.py:35) fall out of the epilogue as per-64-pixel partial rows (no atomics).
#include <stdlib.h>
#include "kp_mfma_util.h"

namespace {

constexpr int TPB = 512, NWAVE = 8;
constexpr int TS = 16;                  // tile side (pixels)
constexpr int PW = TS + 2;              // patch side
constexpr int PPIX = PW * PW;           // 324 patch pixels
constexpr int PINSTR = (PPIX + 7) / 8;  // 41 DMA instructions of 8 pixels (1 KiB)
constexpr int PBYTES = PINSTR * 1024;
constexpr int PSLICES = (PINSTR + NWAVE - 1) / NWAVE;   // patch DMA instructions per wave per chunk (6)
constexpr int ROWB = 128;               // bytes per LDS row (64 bf16)
constexpr int NWBUF = 4;               // weight ring slots (power of two: the slot of a step is a run-time counter)
#ifndef KP_PATCH_CROWS
#define KP_PATCH_CROWS 1
#endif
constexpr int CROWS = KP_PATCH_CROWS;   // statistics rows a persistent workgroup leaves in compact mode: 1, or 4 (one per pixel-row group: rounds 1-3)

struct PatchArgs {
    const h16_t* x; const h16_t* w; const float* bias; h16_t* y; float* partials;
    int Hib, Wib, ioff;   // input buffer rows / cols, offset of the tile domain's patch origin in it
    int Hob, Wob, ooff;   // output buffer rows / cols, offset of the tile domain in it
    int H, W;             // tile domain (multiples of 16)
    int C, K, ktiles, ntiles, dbg;
    int linear;           // 1: tile ids in time order (step i of the grid covers ids [i*grid, (i+1)*grid)); 0: 8 XCD ranges
    int compact;          // 1: BN partial sums are accumulated per workgroup (rows = 4 * grid); 0: 4 rows per pixel tile
    long long* trace;     // optional: s_memtime stamps of workgroup 0 / wave 0 (512 entries), measurement only
    // BRED (data gradient of layer L that also leaves the first pass of layer L - 1's BatchNorm backward, see below)
    const h16_t* yred; const float *mean, *invstd, *gamma, *beta; float slope; float* rpart;
    // FIN (compact STATS launches, kp_conv_fwd_fin): the BatchNorm finalize in this kernel's tail.  fin_tickets != NULL: one counter per
    // k-tile; the LAST of the grid / ktiles workgroups of a k-tile to have written its statistics row sums that tile's rows (fixed
    // order, fp64) and writes mean / invstd / running statistics of the tile's channels -- no finalize launch behind this one.
    unsigned* fin_tickets; float *fin_mean, *fin_invstd, *fin_rmean, *fin_rvar; long long* fin_nbt; float fin_momentum, fin_eps;
    long long fin_count;
    // EPI (inference, kp_conv_fwd_act): out = act(acc * ep_scale[k] + ep_shift[k]) -- an eval-mode BatchNorm (scale = gamma / sqrt(var + eps),
    // shift = beta + (bias - mean) * scale) + LeakyReLU(ep_slope) applied to the fp32 accumulators before the one rounding to the storage type
    const float *ep_scale, *ep_shift; float ep_slope;
};

// Persistent workgroups: the (tile, 64-channel chunk, tap) steps of all tiles of a workgroup form ONE software
// pipeline, so the HBM latency of a tile's first patch / weights and the epilogue stores of the previous tile
// overlap.  Step t (8 or 16 MFMA per wave, reduction slices kk = 0..3 of 16 channels):
//     MFMA kk=0,1 (fragments already in registers)  ||  ds_read fragments kk=2,3 of step t
//     s_waitcnt vmcnt(#DMA issued after the previous barrier); s_barrier   <- weights of step t+1 landed;
//                                                                 no LDS wait in front of the barrier
//     issue DMA: weight tile of step t+3 -> ring slot (t-1)%4 (its readers fed MFMAs issued before the barrier),
//                one slice of the next chunk's patch
//     MFMA kk=2,3                                    ||  ds_read fragments kk=0,1 of step t+1
// STATS: the epilogue also accumulates the BatchNorm partial sums (forward); the data-gradient launches leave them out.
// TRACE: s_memtime stamps for tools/probes/trace.py (measurement build of the same kernel).
// BRED: this launch is the data gradient of layer L (interior of the padded domain, offset (1, 1)) and its output gx is the incoming
//       gradient of layer L - 1's BatchNorm + activation: the FIRST pass of that backward (kp_bn_act_bwd_reduce: sums over the batch
//       of gh = g act'(z) and gh xhat, with z / xhat from layer L - 1's conv output `yred` [N][H][W][K]) runs in the epilogue, on the
//       stored (rounded) gradient values, at the point where a lane holds 16 bytes = 8 consecutive channels of a pixel for its
//       coalesced store: the lane's channels are the same for every store of the launch, so their constants and the two sums
//       per channel live in registers; the matching 16 bytes of yred are requested at the start of the epilogue.  Folding (the
//       adjoint of the replicate pad) is not needed: sum_ij fold(g)_ij f_ij = sum_pq g_pq f[clamp(p-1)][clamp(q-1)], and an
//       interior pixel (p, q) of the padded domain IS source pixel (p - 1, q - 1); the border ring adds its share in its own
//       kernel (kp_conv_mfma.hip).  Rows: 4 per workgroup (one per pixel-row group), zero outside the workgroup's k-tile.
// SST (measurement, kp_config(KP_CFG_STATS_STORE, 1); compact STATS launches only): the BatchNorm sums are taken at the STORE stage -- where a
//       lane holds 8 consecutive channels of a pixel, the same 8 for every store of the launch -- from the values as STORED (rounded to the
//       storage type: what apex O2's fp32 BatchNorm sees of an fp16 convolution output), in 16 registers instead of 64.  The kernel then
//       leaves ~100 registers per SIMD free: one wave of a BatchNorm pass / finalize of the OTHER encoder chain fits beside it.  NOT the
//       default: the statistics differ from the fp32-accumulator ones by the storage rounding (docs/experiments/r06.md section 10).
template <int BN, bool STATS, bool TRACE, bool BRED = false, bool EPI = false, bool SST = false>
__global__ __launch_bounds__(TPB, 2) void conv3x3_patch_k(const PatchArgs A) {
    static_assert(!BRED || (!STATS && !TRACE), "BRED: data-gradient form");
    static_assert(!EPI || (!STATS && !TRACE && !BRED), "EPI: inference form");
    static_assert(!SST || (STATS && !TRACE && !BRED && !EPI), "SST: a form of the forward statistics");
    constexpr int WBYTES = BN * ROWB;
    constexpr int NWI = BN / 64;          // weight DMA instructions per wave per step
    constexpr int WN = BN / 2;            // channels per wave
    constexpr int TN = WN / 32;           // 32-channel MFMA tiles per wave
    constexpr int OFF_W = 2 * PBYTES, OFF_DUMMY = OFF_W + NWBUF * WBYTES;
    constexpr int OFF_BIAS = OFF_DUMMY + 1024;            // [8 waves][64] fp32: each wave's own output-channel biases
    constexpr int OFF_TRACE = OFF_BIAS + NWAVE * 64 * 4;
    constexpr int OFF_RK = OFF_TRACE + 16;                 // BRED: [2 channel halves][8 chunks][mean, invstd, gamma, beta][8] fp32 (2 KB)
    constexpr int OFF_EP = OFF_TRACE + 16;                 // EPI: [scale | shift][8 waves][64] fp32 (4 KB), each wave's own output channels
    __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_TRACE + (TRACE ? 4096 : (BRED ? 16 + 2048 : (EPI ? 16 + 4096 : 16)))];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int C = A.C, K = A.K;
    const int dbg = TRACE ? A.dbg : 0;   // ablation switches exist in the measurement build only

    // ---- persistent tile list: XCD x owns tile ids [x*Q, (x+1)*Q) (k-tile fastest), its workgroups take them
    //      round-robin, so the workgroups of an XCD work on neighbouring tiles at the same time (L2 reuse)
    const int per_xcd = gridDim.x >> 3, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int Q = (A.ntiles + 7) >> 3;
    const int tile_hi = A.linear ? A.ntiles : ((xcd + 1) * Q < A.ntiles ? (xcd + 1) * Q : A.ntiles);
    const int tstride = A.linear ? (int)gridDim.x : per_xcd;   // ids between two consecutive tiles of a workgroup
    int tile = A.linear ? xcd * per_xcd + jb : xcd * Q + jb;
    if (STATS && A.compact) {   // this workgroup's row(s): zero everywhere, its own k-tile columns are set at the end
        float* rowp = A.partials + (long long)blockIdx.x * CROWS * 2 * K;
        for (int i = tid; i < CROWS * 2 * K; i += TPB) rowp[i] = 0.f;
    }
    if (BRED) {   // this workgroup's 4 rows of the BatchNorm-backward sums: zero everywhere, its own k-tile columns are set at the end
        float* rowp = A.rpart + (long long)blockIdx.x * 4 * 2 * K;
        for (int i = tid; i < 4 * 2 * K; i += TPB) rowp[i] = 0.f;
    }
    // FIN (linear tile walk only: the launcher checks): the k-tile of workgroup b is (b >> 3) % ktiles -- its first tile id
    // (b & 7) * per_xcd + (b >> 3) modulo the k-tiles, and (grid / 8) % ktiles == 0 in the compact launches, so every tile of a workgroup
    // has it and each k-tile has exactly E = grid / ktiles workgroups (idle ones included): match m of k-tile kt is workgroup
    // (((m >> 3) * ktiles + kt) << 3) | (m & 7), ascending in m.
    // Called by every workgroup of a FIN launch once its statistics row is in memory (all threads; the row's stores are this thread's
    // own, released by the fence).  LDS: NP * 2 * BN doubles at offset 0 (the patch buffers: nobody reads them any more).
    auto fin_fold = [&](int kt) {
        if constexpr (STATS) {
            // the row's stores (all threads) have reached this XCD's L2 when the barrier releases; ONE device-scope release behind it
            // (thread 0: an L2 write-back -- 512 of them per workgroup cost the launch ~25 us, profiles/r06_fin_fold_ab.txt) makes
            // them visible to the other XCDs before the ticket is taken
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* flag = reinterpret_cast<int*>(smem + OFF_TRACE);
            const int E = (int)gridDim.x / A.ktiles;
            if (tid == 0) {
                __threadfence();
                const unsigned old = __hip_atomic_fetch_add(A.fin_tickets + kt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
                *flag = old == (unsigned)E - 1u;
            }
            __syncthreads();
            if (!*flag) return;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the other workgroups' rows (written back by their releases) are read from memory
            // 2 * BN columns (sum | sum of squares of the tile's channels) x NP row parts: part p adds matches p, p + NP, ... in ascending
            // order (loads of 8 matches in flight), the parts are added in index order through LDS -> one fixed summation order
            constexpr int NP = TPB / (2 * BN);
            const int col = tid % (2 * BN), part = tid / (2 * BN);
            const int which = col / BN, kl = col - which * BN;
            const float* colp = A.partials + (long long)which * K + kt * BN + kl;
            const long long rstride = 2LL * K;
            auto row_of = [&](int m) { return (long long)(((((m >> 3) * A.ktiles) + kt) << 3) | (m & 7)) * rstride; };
            double acc64 = 0.0;
            int m = part;
            for (; m + 7 * NP < E; m += 8 * NP) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = colp[row_of(m + u * NP)];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc64 += (double)v[u];
            }
            for (; m < E; m += NP) acc64 += (double)colp[row_of(m)];
            double* red64 = reinterpret_cast<double*>(smem);
            red64[part * 2 * BN + col] = acc64;
            __syncthreads();
            if (tid < BN) {
                double sum = 0.0, sq = 0.0;
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) { sum += red64[pp * 2 * BN + tid]; sq += red64[pp * 2 * BN + BN + tid]; }
                const int c = kt * BN + tid;
                const double count = (double)A.fin_count;
                const double mu = sum / count;              // (the expressions of bn_reduce_finalize_wide_k, kp_bn.hip)
                double var = sq / count - mu * mu;
                if (var < 0.0) var = 0.0;
                A.fin_mean[c] = (float)mu;
                A.fin_invstd[c] = (float)(1.0 / sqrt(var + (double)A.fin_eps));
                if (A.fin_rmean) A.fin_rmean[c] = (1.f - A.fin_momentum) * A.fin_rmean[c] + A.fin_momentum * (float)mu;
                if (A.fin_rvar) {
                    const double unbiased = A.fin_count > 1 ? var * count / (double)(A.fin_count - 1) : var;
                    A.fin_rvar[c] = (1.f - A.fin_momentum) * A.fin_rvar[c] + A.fin_momentum * (float)unbiased;
                }
            }
            if (tid == 0) {
                if (A.fin_nbt && kt == 0) *A.fin_nbt += 1;
                __hip_atomic_store(A.fin_tickets + kt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero again for the next launch on this stream
            }
        }
    };
    if (tile >= tile_hi) {
        if constexpr (STATS) if (A.compact && A.fin_tickets) { __syncthreads(); fin_fold((int)(blockIdx.x >> 3) % A.ktiles); }   // (an idle workgroup's row: the zeros above)
        return;
    }
    const int tiles_x = A.W / TS, tiles_y = A.H / TS;

    struct TileRef { int n, oy0, ox0, k0, pt, kt, tx, ty; };
    auto decode = [&](int id) {   // divisions: only for a workgroup's first tile
        TileRef t;
        t.kt = id % A.ktiles;
        t.pt = id / A.ktiles;
        t.tx = t.pt % tiles_x; t.ty = (t.pt / tiles_x) % tiles_y;
        t.n = t.pt / (tiles_x * tiles_y);
        t.oy0 = t.ty * TS; t.ox0 = t.tx * TS; t.k0 = t.kt * BN;
        return t;
    };
    // a workgroup's next tile is per_xcd ids further: add the mixed-radix digits of that stride with carries
    const int d_kt = tstride % A.ktiles, d_pt = tstride / A.ktiles;
    const int d_tx = d_pt % tiles_x, d_ty = (d_pt / tiles_x) % tiles_y, d_n = d_pt / (tiles_x * tiles_y);
    auto advance = [&](const TileRef& c) {
        TileRef t;
        int carry;
        t.kt = c.kt + d_kt; carry = t.kt >= A.ktiles; t.kt -= carry ? A.ktiles : 0;
        t.pt = c.pt + d_pt + carry;
        t.tx = c.tx + d_tx + carry; carry = t.tx >= tiles_x; t.tx -= carry ? tiles_x : 0;
        t.ty = c.ty + d_ty + carry; carry = t.ty >= tiles_y; t.ty -= carry ? tiles_y : 0;
        t.n = c.n + d_n + carry;
        t.oy0 = t.ty * TS; t.ox0 = t.tx * TS; t.k0 = t.kt * BN;
        return t;
    };
    auto x_rsrc = [&](const TileRef& t) {
        const unsigned char* xbase = (const unsigned char*)A.x +
            ((((long long)t.n * A.Hib + t.oy0 + A.ioff) * A.Wib) + t.ox0 + A.ioff) * (long long)C * 2;
        return __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, 0x7ffffffe, 0x00020000);
    };
    auto w_rsrc = [&](const TileRef& t) {
        const unsigned char* wbase = (const unsigned char*)A.w + (long long)t.k0 * 9 * C * 2;
        return __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7ffffffe, 0x00020000);
    };

    // ---- per-lane DMA offsets (tile independent) --------------------------------------------------------
    const int lrow8 = lane >> 3, lslot = lane & 7;
    int pvoff[PSLICES];   // patch: instruction ii = j*8 + wave covers patch pixels ii*8 .. ii*8+7
#pragma unroll
    for (int j = 0; j < PSLICES; ++j) {
        int pi = (j * NWAVE + wave) * 8 + lrow8;
        if (pi > PPIX - 1) pi = PPIX - 1;
        const int py = pi / PW, px = pi - py * PW;
        pvoff[j] = (py * A.Wib + px) * C * 2 + ((lslot ^ ((px >> 1) & 7)) << 4);
    }
    int wvoff[NWI];       // weights: instruction ii = j*8 + wave covers rows ii*8 .. ii*8+7 of the k-tile
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
        const int row = (j * NWAVE + wave) * 8 + lrow8;
        wvoff[j] = row * 9 * C * 2 + ((lslot ^ ((row >> 1) & 7)) << 4);
    }
    auto issue_w = [&](const __amdgpu_buffer_rsrc_t& wr, int buf, int tap, int c0) {
        unsigned char* dst = smem + OFF_W + buf * WBYTES;
        const int soff = (tap * C + c0) * 2;
#pragma unroll
        for (int j = 0; j < NWI; ++j) dma16(wr, dst + (j * NWAVE + wave) * 1024, wvoff[j], soff);
    };
    auto issue_p = [&](const __amdgpu_buffer_rsrc_t& xr, int pbuf, int j, int c0) {   // j compile-time
        const int ii = j * NWAVE + wave;
        unsigned char* dst = (ii < PINSTR) ? smem + pbuf * PBYTES + ii * 1024 : smem + OFF_DUMMY;
        dma16(xr, dst, pvoff[j], c0 * 2);
    };

    // ---- fragment read offsets --------------------------------------------------------------------------
    const int fr = lane & 31, fh = lane >> 5;
    const int ftx = fr & 15;
    int xoff[2];      // byte offset of this lane's pixel (tap 0,0) inside the patch
#pragma unroll
    for (int b = 0; b < 2; ++b) xoff[b] = ((wm * 4 + b * 2 + (fr >> 4)) * PW + ftx) * ROWB;
    int xs16[3];      // swizzled 16-B slot of kk = 0 for column shift s
#pragma unroll
    for (int s = 0; s < 3; ++s) xs16[s] = (fh ^ (((ftx + s) >> 1) & 7)) << 4;
    int woff[TN], ws16[TN];
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int row = wn * WN + a * 32 + fr;
        woff[a] = row * ROWB;
        ws16[a] = (fh ^ ((row >> 1) & 7)) << 4;
    }

    struct Frag { h16x8 x[2]; h16x8 w[TN]; };
    auto load_frag = [&](Frag& f, const unsigned char* xb, const unsigned char* wb, int s, int kk) {
        if (TRACE && (dbg & 64) && (kk & 1)) return;   // ablation: half the LDS fragment reads
#pragma unroll
        for (int b = 0; b < 2; ++b) f.x[b] = *(const h16x8*)(xb + xoff[b] + (xs16[s] ^ (kk << 5)));
#pragma unroll
        for (int a = 0; a < TN; ++a) f.w[a] = *(const h16x8*)(wb + woff[a] + (ws16[a] ^ (kk << 5)));
    };

    f32x16 acc[TN][2];
    // accumulators start at the bias of the lane's output channels (wave-private LDS copy, see load_bias): the
    // epilogue then has no bias add, and no global load that would queue behind its own stores
    auto init_acc = [&]() {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = *reinterpret_cast<const float4*>(smem + OFF_BIAS + (wave * 64 + a * 32 + 8 * g + 4 * fh) * 4);
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b][g * 4 + 0] = bq.x; acc[a][b][g * 4 + 1] = bq.y;
                    acc[a][b][g * 4 + 2] = bq.z; acc[a][b][g * 4 + 3] = bq.w;
                }
            }
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[a][b] = kp_mfma16(f.w[a], f.x[b], acc[a][b], 0, 0, 0);
    };

    // epilogue: D[k_out][pixel]; lane: pixel = fr of fragment b, k_out = 8*(reg>>2) + 4*fh + (reg&3).
    // The wave's 32-pixel x WN-channel fragments go through a wave-private, XOR-swizzled staging tile in the
    // patch buffer that the finished chunk no longer needs, and leave as 16-byte lanes over whole 128-byte
    // (64-byte for BN = 64) runs per pixel: 8x fewer cache lines per store instruction than storing straight
    // from the MFMA layout.
    constexpr int CPP = WN / 8;            // 16-byte chunks per pixel (this wave's channels)
    // BN partial sums per lane (pixel) and accumulator register (channel).  A persistent workgroup (compact mode) keeps
    // them across its tiles and runs the cross-lane reduction once at its end: no cross-lane work in the epilogue.
    float sm[(STATS && !SST) ? TN : 1][16], sq[(STATS && !SST) ? TN : 1][16];
    float ss8[SST ? 8 : 1], sq8[SST ? 8 : 1];
    if constexpr (SST) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ss8[e] = sq8[e] = 0.f;
    }
    if constexpr (STATS && !SST) {
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) sm[a][i] = sq[a][i] = 0.f;
    }
    constexpr int PPI = 64 / CPP;          // pixels per store instruction
    const int st_voff = ((lane / CPP) * K + (lane % CPP) * 8) * 2;   // lane part of the output address (bytes)
    constexpr int NST = 32 * CPP / 64;     // store instructions per 32-pixel block
    // BRED: the lane's 8 channels (k-tile of the workgroup + wn * WN + (lane % CPP) * 8): their running sums in registers, their
    // constants in LDS (32 more persistent registers beside the 64 accumulators do not fit; read back once per tile)
    float rsb[BRED ? 8 : 1], rsg[BRED ? 8 : 1];
    auto bred8 = [&](const KpBnIn8& rk, const u32x4& gw, const u32x4& yw) {
        float g[8], yv[8];
        unpack_h16x2(gw[0], g[0], g[1]); unpack_h16x2(gw[1], g[2], g[3]); unpack_h16x2(gw[2], g[4], g[5]); unpack_h16x2(gw[3], g[6], g[7]);
        unpack_h16x2(yw[0], yv[0], yv[1]); unpack_h16x2(yw[1], yv[2], yv[3]); unpack_h16x2(yw[2], yv[4], yv[5]); unpack_h16x2(yw[3], yv[6], yv[7]);
#pragma unroll
        for (int e = 0; e < 8; ++e) {   // the expressions of kp_bn_act_bwd_reduce (and of kp_conv_small.hip's BRED)
            const float xh = (yv[e] - rk.mu[e]) * rk.is[e];
            const float z = xh * rk.ga[e] + rk.be[e];
            const float gh = g[e] * act_grad(z, A.slope);
            rsb[e] += gh;
            rsg[e] += gh * xh;
        }
    };
    // BRED: the tile's window of layer L - 1's conv output (unpadded [H][W]) in the store's lane layout, requested during the tile's
    // LAST chunk (two steps before they could stall a barrier) so that the epilogue finds them in registers: requested in the
    // epilogue itself, all workgroups ask for them at once and the chip waits ~3 us per tile round for 17 MB
    u32x4 yq[BRED ? 2 : 1][BRED ? NST : 1];
    constexpr int YTAP = 2;   // the tap of the last chunk behind whose DMA issue the requests go
    auto issue_y = [&](const TileRef& t) {
        if constexpr (BRED) {
            const h16_t* rbase = A.yred + ((((long long)t.n * A.H + t.oy0 + wm * 4) * A.W) + t.ox0) * (long long)K + t.k0 + wn * WN;
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc((void*)rbase, 0, 0x7ffffffe, 0x00020000);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < NST; ++i)
                    yq[b][i] = __builtin_amdgcn_raw_buffer_load_b128(rr, st_voff, ((b * 2 + ((i * PPI) >> 4)) * A.W + ((i * PPI) & 15)) * K * 2, 0);
        }
    };
    auto epilogue = [&](const TileRef& t, int free_pb) {
        unsigned char* stg = smem + free_pb * PBYTES + wave * (32 * WN * 2);
        const long long prow = (long long)t.pt * 4 + wm;   // statistics row of this wave (64 pixels)
        // this wave's 4 x 16 pixel x WN channel output window from one scalar base: stores carry a lane-constant
        // VGPR offset and a scalar (row, column) offset -- no per-store vector address arithmetic
        h16_t* ybase = A.y + ((((long long)t.n * A.Hob + t.oy0 + wm * 4 + A.ooff) * A.Wob) + t.ox0 + A.ooff) * (long long)K +
                        t.k0 + wn * WN;
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, 0x7ffffffe, 0x00020000);
        u32x4 gq[BRED ? 2 : 1][BRED ? NST : 1];
        if constexpr (STATS && !SST) if (!A.compact) {   // per-tile rows: start from zero
#pragma unroll
            for (int a = 0; a < TN; ++a)
#pragma unroll
                for (int i = 0; i < 16; ++i) sm[a][i] = sq[a][i] = 0.f;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int a = 0; a < TN; ++a) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
                    float4 esc, esh;
                    if constexpr (EPI) {
                        esc = *reinterpret_cast<const float4*>(smem + OFF_EP + (wave * 64 + a * 32 + 8 * g + 4 * fh) * 4);
                        esh = *reinterpret_cast<const float4*>(smem + OFF_EP + 2048 + (wave * 64 + a * 32 + 8 * g + 4 * fh) * 4);
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = acc[a][b][g * 4 + e];
                        if constexpr (EPI) {
                            const float z = __builtin_fmaf(v[e], e == 0 ? esc.x : e == 1 ? esc.y : e == 2 ? esc.z : esc.w,
                                                           e == 0 ? esh.x : e == 1 ? esh.y : e == 2 ? esh.z : esh.w);
                            v[e] = fmaxf(z, z * A.ep_slope);   // (= LeakyReLU for 0 <= slope <= 1; slope 0: ReLU)
                        }
                        if constexpr (STATS && !SST) {
                            sm[a][g * 4 + e] += v[e];
                            sq[a][g * 4 + e] = __builtin_fmaf(v[e], v[e], sq[a][g * 4 + e]);
                        }
                    }
                    uint2 pk;
                    pk.x = pack_h16x2(v[0], v[1]);
                    pk.y = pack_h16x2(v[2], v[3]);
                    const int chunk = a * 4 + g;
                    *reinterpret_cast<uint2*>(stg + fr * (WN * 2) + ((chunk ^ (fr & (CPP - 1))) << 4) + fh * 8) = pk;
                }
            }
            if (!(dbg & 1)) {
#pragma unroll
                for (int i = 0; i < 32 * CPP / 64; ++i) {
                    const int id = i * 64 + lane, px = id / CPP, c16 = id % CPP;
                    const u32x4 vv = *reinterpret_cast<const u32x4*>(stg + px * (WN * 2) + ((c16 ^ (px & (CPP - 1))) << 4));
                    const int soff = ((b * 2 + ((i * PPI) >> 4)) * A.Wob + ((i * PPI) & 15)) * K * 2;
                    __builtin_amdgcn_raw_buffer_store_b128(vv, yr, st_voff, soff, 0);
                    if constexpr (BRED) gq[b][i] = vv;
                    if constexpr (SST) {
                        float f[8];
                        unpack_h16x2(vv[0], f[0], f[1]); unpack_h16x2(vv[1], f[2], f[3]); unpack_h16x2(vv[2], f[4], f[5]); unpack_h16x2(vv[3], f[6], f[7]);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { ss8[e] += f[e]; sq8[e] = __builtin_fmaf(f[e], f[e], sq8[e]); }
                    }
                }
            }
        }
        if constexpr (BRED) {   // behind all stores of the tile (the accumulator registers are free, the requests are ~1000 cycles old)
            __builtin_amdgcn_sched_barrier(0);
            KpBnIn8 rk;
            const float* tab = reinterpret_cast<const float*>(smem + OFF_RK) + (wn * 8 + (lane % CPP)) * 32;
#pragma unroll
            for (int e = 0; e < 8; ++e) { rk.mu[e] = tab[e]; rk.is[e] = tab[8 + e]; rk.ga[e] = tab[16 + e]; rk.be[e] = tab[24 + e]; }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < NST; ++i) {
                    bred8(rk, gq[b][i], yq[b][i]);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        if constexpr (STATS && !SST) if (!A.compact && !(dbg & 2)) {
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                const int kb = t.k0 + wn * WN + a * 32;
                int r1, r2;
                const float ts = reduce16_over32(sm[a], lane, r1);
                const float tq = reduce16_over32(sq[a], lane, r2);
                if ((lane & 1) == 0) {
                    const int k = kb + 8 * (r1 >> 2) + 4 * fh + (r1 & 3);
                    A.partials[(prow * 2 + 0) * K + k] = ts;
                    A.partials[(prow * 2 + 1) * K + k] = tq;
                }
            }
        }
    };

    const bool tracing = TRACE && A.trace != nullptr && blockIdx.x == 0 && wave == 0;
    int tri = 0;
    // dbg & 32: one (s_memtime, s_memrealtime) pair per tile instead of the per-step stamps
    auto stamp_tile = [&]() {
        if constexpr (TRACE) if (tracing && (dbg & 32)) {
            const long long now = __builtin_amdgcn_s_memtime(), real = __builtin_amdgcn_s_memrealtime();
            if (lane == 0 && tri + 1 < 512) {
                reinterpret_cast<long long*>(smem + OFF_TRACE)[tri] = now;
                reinterpret_cast<long long*>(smem + OFF_TRACE)[tri + 1] = real;
            }
            tri += 2;
        }
    };
    auto stamp = [&]() {
        if constexpr (TRACE) if (tracing && !(dbg & 32)) {
            const long long now = __builtin_amdgcn_s_memtime();
            if (lane == 0 && tri < 512) reinterpret_cast<long long*>(smem + OFF_TRACE)[tri] = now;
            ++tri;
        }
    };
    float dummy[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};   // TRACE build only (see dbg & 0x700)
    const int nchunks = C / 64;
    TileRef cur = decode(tile);
    __amdgpu_buffer_rsrc_t xr = x_rsrc(cur), wr = w_rsrc(cur);
    // this wave's WN output-channel biases -> its private LDS slot (read back in the epilogue with lgkmcnt only,
    // so the epilogue never waits on vmcnt behind its own stores)
    auto load_bias = [&](const TileRef& t) {
        if (lane < WN) {
            const float bvl = A.bias ? A.bias[t.k0 + wn * WN + lane] : 0.f;
            reinterpret_cast<float*>(smem + OFF_BIAS)[wave * 64 + lane] = bvl;
            if constexpr (EPI) {
                reinterpret_cast<float*>(smem + OFF_EP)[wave * 64 + lane] = A.ep_scale[t.k0 + wn * WN + lane];
                reinterpret_cast<float*>(smem + OFF_EP + 2048)[wave * 64 + lane] = A.ep_shift[t.k0 + wn * WN + lane];
            }
        }
    };
    load_bias(cur);
    if constexpr (BRED) {   // (one k-tile per workgroup: the launcher checks it; the prologue's barrier publishes the table)
        if (wm == 0 && lane < CPP) {
            KpBnIn8 rk;
            const KpBnIn bnr{A.mean, A.invstd, A.gamma, A.beta, A.slope};
            rk.load(bnr, cur.k0 + wn * WN + lane * 8);
            float* tab = reinterpret_cast<float*>(smem + OFF_RK) + (wn * 8 + lane) * 32;
#pragma unroll
            for (int e = 0; e < 8; ++e) { tab[e] = rk.mu[e]; tab[8 + e] = rk.is[e]; tab[16 + e] = rk.ga[e]; tab[24 + e] = rk.be[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) rsb[e] = rsg[e] = 0.f;
    }

    // ---- prologue: patch of chunk 0, weight tiles of steps 0..2 of the first tile -------------------------
#pragma unroll
    for (int j = 0; j < PSLICES; ++j) issue_p(xr, 0, j, 0);
    issue_w(wr, 0, 0, 0);
    issue_w(wr, 1, 1, 0);
    issue_w(wr, 2, 2, 0);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    Frag f0, f1, f2, f3;
    load_frag(f0, smem, smem + OFF_W, 0, 0);
    load_frag(f1, smem, smem + OFF_W, 0, 1);
    init_acc();
    int pbi = 0;       // patch buffer of the current chunk
    int wslot = 0;     // weight ring slot of the current step
    bool after_epi = false;

    while (true) {
        const int ntile_id = tile + tstride;
        const bool has_next = ntile_id < tile_hi;
        TileRef nxt = cur;
        __amdgpu_buffer_rsrc_t xrn = xr, wrn = wr;
        if (has_next) { nxt = advance(cur); xrn = x_rsrc(nxt); wrn = w_rsrc(nxt); }

        for (int ch = 0; ch < nchunks; ++ch) {
            const unsigned char* pb = smem + pbi * PBYTES;
            const unsigned char* npb = smem + (pbi ^ 1) * PBYTES;
            const bool last_ch = ch + 1 == nchunks;
            const bool more_p = !last_ch || has_next;          // is there a "next chunk" to prefetch a patch for
            static_for<0, 9>([&](auto tapc) {
                constexpr int tap = decltype(tapc)::value;
                constexpr int r = tap / 3, s = tap % 3;
                const unsigned char* xb = pb + (r * PW + s) * ROWB;
                const unsigned char* wb = smem + OFF_W + wslot * WBYTES;
                constexpr int dummy_c = 0; (void)dummy_c;
                const int ntap = (tap + 1) % 9, nr = ntap / 3, ns = ntap % 3;
                const unsigned char* nxb = (tap == 8 ? npb : pb) + (nr * PW + ns) * ROWB;
                const unsigned char* nwb = smem + OFF_W + ((wslot + 1) & (NWBUF - 1)) * WBYTES;

                // -- first half: kk = 0,1 from registers, interleaved with the fragment reads of kk = 2,3
                //    (issuing all 8 reads ahead of the MFMAs measured 4 % slower: longer barrier skew)
                mma(f0);
                load_frag(f2, xb, wb, s, 2);
                mma(f1);
                load_frag(f3, xb, wb, s, 3);
                __builtin_amdgcn_sched_barrier(0);
                stamp();
                // No LDS wait here: the ring slot that is overwritten after this barrier is the one of step t-1,
                // and every wave consumed its reads of it (MFMA operands) before it can arrive.  Only the last step
                // of a tile waits, because the epilogue reuses the patch buffer that other waves may still be reading.
                if (tap == 8 && last_ch) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                stamp();
                // Everything issued before the previous barrier has landed (weights of step t+1, older patch
                // slices): the DMA count after a barrier is a compile-time constant per tap (dummy targets keep it
                // so).  Right after an epilogue everything older was already drained and its stores must not be
                // waited for here.
                if (!(tap == 0 && after_epi) && !(dbg & 16)) {
                    constexpr int ND = NWI + (((tap + 8) % 9) < PSLICES ? 1 : 0);
                    // BRED: the 2 NST requests for layer L - 1's conv output sit in the queue between the DMAs of tap YTAP and YTAP + 1
                    if (BRED && last_ch && (tap == YTAP + 1 || tap == YTAP + 2)) wait_vm<ND + (BRED ? 2 * NST : 0)>();
                    else wait_vm<ND>();
                }
                stamp();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                stamp();
                __builtin_amdgcn_sched_barrier(0);

                // (measurement build: dbg & 0x100 / 0x200 / 0x400 add 24 / 48 / 96 dependent-free fp32 FMAs per step and wave --
                //  what a BatchNorm-apply of the patch fused into this loop would issue -- to see whether the power-limited loop
                //  has the issue slots for them: tools/probes/valu_in_loop.py)
                if constexpr (TRACE) {
                    const int reps = ((dbg >> 8) & 1) * 1 + ((dbg >> 9) & 1) * 2 + ((dbg >> 10) & 1) * 4;
                    for (int rr = 0; rr < reps; ++rr) {
#pragma unroll
                        for (int q = 0; q < 24; ++q) dummy[q & 7] = __builtin_fmaf(dummy[q & 7], 1.0001f, dummy[(q + 3) & 7]);
                    }
                }
                // -- second half: kk = 2 MFMAs first (keeps the matrix pipe fed while the DMA instructions issue)
                mma(f2);
                __builtin_amdgcn_sched_barrier(0);
                // weight tile of step t+3 into the ring slot of step t-1 (the last step of a tile defers it until
                // after the epilogue), one slice of the next patch in taps 0..5; no-op cases go to a dummy LDS target
                if (!(tap == 8 && last_ch)) {
                    constexpr int tap3 = (tap + 3) % 9;
                    constexpr bool wrap = tap + 3 >= 9;
                    const bool to_next = wrap && last_ch;
                    const bool real = !to_next || has_next;
                    unsigned char* dst = real ? smem + OFF_W + ((wslot + 3) & (NWBUF - 1)) * WBYTES : smem + OFF_DUMMY;
                    const int cw = to_next ? 0 : (wrap ? ch + 1 : ch) * 64;
                    const int soff = (tap3 * C + cw) * 2;
#pragma unroll
                    for (int j = 0; j < NWI; ++j)
                        if (!(TRACE && (dbg & 128)))   // ablation: no DMA in the main loop
                            dma16(to_next ? wrn : wr, real ? dst + (j * NWAVE + wave) * 1024 : dst, wvoff[j], soff);
                }
                if constexpr (tap < PSLICES) {
                    const int ii = tap * NWAVE + wave;
                    unsigned char* dst = (more_p && ii < PINSTR) ? smem + (pbi ^ 1) * PBYTES + ii * 1024 : smem + OFF_DUMMY;
                    if (!(TRACE && (dbg & 128))) dma16(last_ch ? xrn : xr, dst, pvoff[tap], last_ch ? 0 : (ch + 1) * 128);
                }
                if constexpr (BRED && tap == YTAP) if (last_ch) issue_y(cur);
                __builtin_amdgcn_sched_barrier(0);
                load_frag(f0, nxb, nwb, ns, 0);
                mma(f3);
                load_frag(f1, nxb, nwb, ns, 1);
                __builtin_amdgcn_sched_barrier(0);
                wslot = (wslot + 1) & (NWBUF - 1);
            });
            pbi ^= 1;
            after_epi = false;
        }
        // tile switch: drain the DMA queue (everything in it was issued >= half a step ago), store the tile,
        // then issue the weight tile that the last step deferred.  The next barrier does not wait (after_epi);
        // the one after it retires these stores together with the DMA.
        stamp();
        stamp_tile();
        wait_vm<0>();
        stamp();
        epilogue(cur, pbi ^ 1);   // pbi = buffer of the NEXT tile's first patch; the other one is free
        stamp();
        if (!has_next) break;
        if (nxt.k0 != cur.k0) load_bias(nxt);   // read again only at the next epilogue (many barriers later)
        issue_w(wrn, (wslot + 2) & (NWBUF - 1), 2, 0);   // wslot = slot of the next tile's step 0
        after_epi = true;
        init_acc();
        tile = ntile_id; cur = nxt; xr = xrn; wr = wrn;
    }
    if constexpr (STATS) if (A.compact && !(dbg & 2)) {
        // CROWS == 1: the four pixel-row groups' sums are added through LDS (fixed order wm = 0..3) and the workgroup leaves ONE row:
        // the finalize behind this launch -- a latency chain on the critical path of the layer -- reads a quarter of the rows
        float* red = reinterpret_cast<float*>(smem);   // [4 wm][2][BN] (the patch buffers are free: the last epilogue is behind a barrier-free tail, see below)
        if (CROWS == 1) __syncthreads();               // every wave is past its last epilogue's LDS reads
        if constexpr (SST) {   // lanes l, l + CPP, ... hold the same 8 channels: fold them (fixed tree), lanes 0 .. CPP-1 write
            static_assert(!SST || CROWS == 1, "SST: one statistics row per workgroup");
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int o = CPP; o < 64; o <<= 1) {
                    ss8[e] += __shfl_xor(ss8[e], o, 64);
                    sq8[e] += __shfl_xor(sq8[e], o, 64);
                }
            }
            if (lane < CPP) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[(wm * 2 + 0) * BN + wn * WN + lane * 8 + e] = ss8[e];
                    red[(wm * 2 + 1) * BN + wn * WN + lane * 8 + e] = sq8[e];
                }
            }
        }
#pragma unroll
        for (int a = 0; a < (SST ? 0 : TN); ++a) {
            int rg, r2;
            const float ts = reduce16_over32(sm[a], lane, rg);
            const float tq = reduce16_over32(sq[a], lane, r2);
            if ((lane & 1) == 0) {
                const int kl = wn * WN + a * 32 + 8 * (rg >> 2) + 4 * fh + (rg & 3);
                if (CROWS == 1) {
                    red[(wm * 2 + 0) * BN + kl] = ts;
                    red[(wm * 2 + 1) * BN + kl] = tq;
                } else {
                    const long long prow = (long long)blockIdx.x * 4 + wm;
                    A.partials[(prow * 2 + 0) * K + cur.k0 + kl] = ts;
                    A.partials[(prow * 2 + 1) * K + cur.k0 + kl] = tq;
                }
            }
        }
        if (CROWS == 1) {
            __syncthreads();
            for (int i = tid; i < 2 * BN; i += TPB) {
                const int which = i / BN, kl = i - which * BN;
                const float t = ((red[(0 * 2 + which) * BN + kl] + red[(1 * 2 + which) * BN + kl]) + red[(2 * 2 + which) * BN + kl]) + red[(3 * 2 + which) * BN + kl];
                A.partials[((long long)blockIdx.x * 2 + which) * K + cur.k0 + kl] = t;
            }
            if (A.fin_tickets) fin_fold(cur.kt);
        }
    }
    if constexpr (BRED) {   // lanes l, l + CPP, l + 2 CPP, ... hold the same 8 channels: fold them, lanes 0 .. CPP-1 write
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int o = CPP; o < 64; o <<= 1) {
                rsb[e] += __shfl_xor(rsb[e], o, 64);
                rsg[e] += __shfl_xor(rsg[e], o, 64);
            }
        }
        if (lane < CPP) {
            const long long prow = (long long)blockIdx.x * 4 + wm;
            const int kc = cur.k0 + wn * WN + lane * 8;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                A.rpart[(prow * 2 + 0) * K + kc + e] = rsb[e];
                A.rpart[(prow * 2 + 1) * K + kc + e] = rsg[e];
            }
        }
    }
    if constexpr (TRACE) if ((dbg & 0x700) && dummy[0] + dummy[1] + dummy[2] + dummy[3] + dummy[4] + dummy[5] + dummy[6] + dummy[7] == 12345.678f)
        A.partials[0] = dummy[0];   // keeps the measurement FMAs alive
    if constexpr (TRACE) if (tracing && lane == 0)
        for (int i = 0; i < 512; ++i) A.trace[i] = i < tri ? reinterpret_cast<long long*>(smem + OFF_TRACE)[i] : 0;
}

}  // namespace

// The data gradient's interior with the BatchNorm-backward sums of the layer below in its epilogue (BRED above).  gx window as in
// kp_conv_patch_launch(ioff = ooff = 1); `yred` [N][H][W][C]; rows written into `rpart`: 4 per workgroup = the return value of
// kp_conv_patch_bred_rows.  KP_E_UNSUPPORTED where a workgroup would change its k-tile between tiles (its constants and sums are
// per k-tile) or the shape has no patch tiling.
static bool patch_bred_plan(int N, int H, int W, int K, int C, int& grid, int& ktiles, int& ntiles) {
    if (K % 64 != 0 || C % 64 != 0 || H % TS != 0 || W % TS != 0 || H <= 0 || W <= 0) return false;
    if ((long long)(PW + 1) * (W + 4) * K * 2 >= 0x7fffffffLL || 128LL * 9 * K * 2 + 9LL * K * 2 >= 0x7fffffffLL) return false;
    const long long ptiles = (long long)N * (H / TS) * (W / TS);
    const int bn = (C % 128 == 0) ? 128 : 64;
    // 128-channel output tiles: the kernel stands at 179 registers; 16 running sums + 32 registers of prefetched yred + the epilogue's
    // temporaries push it over the 256 of two waves per SIMD, and ONE spill reload inside the (tap, chunk) loop -- a VMEM
    // instruction in front of the counted vmcnt waits -- costs 35 % (tools/probes/bred_layers.py: 545 + 117 us separate, 863 us
    // fused at 256 -> 256 @ 64 x 64; with the requests issued in the epilogue instead: no spill in the loop, 698 us: the chip waits
    // for 17 MB per tile round).  Only the 64-channel tile (131 registers) takes the fused form: 380 + 118 -> 446 us at
    // 64 -> 128 @ 128 x 128.  kp_config(KP_CFG_BRED_PATCH, 2) offers the 128-channel form too (tests, measurements).
    { extern kp_knob_t g_kp_bred_patch; if (bn == 128 && g_kp_bred_patch < 2) return false; }
    ktiles = C / bn;
    if (ptiles * ktiles > 0x7fffffffLL) return false;
    ntiles = (int)(ptiles * ktiles);
    int cus = kp_device_cus();
    if (cus <= 0) cus = 256;
    grid = (cus / 8) * 8;
    if (grid < 8) grid = 8;
    const int need = ((ntiles + 7) / 8) * 8;
    if (grid > need) grid = need;
    return ntiles <= grid || (((grid / 8) % ktiles) == 0 && (grid % ktiles) == 0);
}
long long kp_conv_patch_bred_rows(int N, int H, int W, int K, int C) {
    int grid, ktiles, ntiles;
    return patch_bred_plan(N, H, W, K, C, grid, ktiles, ntiles) ? (long long)grid * 4 : 0;
}
int kp_conv_patch_launch_bred(const void* dy, const void* w, void* gx, const void* yred, const float* mean, const float* invstd,
                              const float* gamma, const float* beta, float slope, float* rpart, int N, int H, int W, int K, int C,
                              hipStream_t st) {
    int grid, ktiles, ntiles;
    if (!patch_bred_plan(N, H, W, K, C, grid, ktiles, ntiles)) return KP_E_UNSUPPORTED;
    PatchArgs a;
    a.x = (const h16_t*)dy; a.w = (const h16_t*)w; a.bias = nullptr; a.y = (h16_t*)gx; a.partials = nullptr;
    a.Hib = H + 4; a.Wib = W + 4; a.ioff = 1; a.Hob = H + 2; a.Wob = W + 2; a.ooff = 1; a.H = H; a.W = W;
    a.C = K; a.K = C;   // kernel roles: input channels = the gradient's K, output channels = C
    a.dbg = 0; a.trace = nullptr;
    { extern kp_knob_t g_kp_walk; a.linear = (g_kp_walk & KP_WALK_CONV_LINEAR) ? 1 : 0; }
    a.ktiles = ktiles; a.ntiles = ntiles; a.compact = 0;
    a.yred = (const h16_t*)yred; a.mean = mean; a.invstd = invstd; a.gamma = gamma; a.beta = beta; a.slope = slope; a.rpart = rpart;
    a.fin_tickets = nullptr; a.fin_mean = a.fin_invstd = a.fin_rmean = a.fin_rvar = nullptr; a.fin_nbt = nullptr; a.fin_momentum = a.fin_eps = 0.f; a.fin_count = 1;
    a.ep_scale = a.ep_shift = nullptr; a.ep_slope = 1.f;
    if (C % 128 == 0) conv3x3_patch_k<128, false, false, true><<<grid, TPB, 0, st>>>(a);
    else conv3x3_patch_k<64, false, false, true><<<grid, TPB, 0, st>>>(a);
    return kp_check_launch("conv3x3_patch_bred");
}

// Valid 3x3 correlation restricted to a (H x W) window of the output whose top-left output pixel is
// (ooff, ooff) of y[N][Hob][Wob][K]; it reads x[N][Hib][Wib][C] starting at (ioff, ioff).
// Returns KP_E_UNSUPPORTED when the shape is not eligible.
int kp_conv_patch_launch(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hib,
                         int Wib, int ioff, int Hob, int Wob, int ooff, int H, int W, int C, int K, hipStream_t st) {
    if (C % 64 != 0 || K % 64 != 0 || H % TS != 0 || W % TS != 0 || H <= 0 || W <= 0) return KP_E_UNSUPPORTED;
    // 32-bit DMA offsets: one patch + one k-tile of weights must be addressable from their bases
    if ((long long)(PW + 1) * Wib * C * 2 >= 0x7fffffffLL || 128LL * 9 * C * 2 + 9LL * C * 2 >= 0x7fffffffLL)
        return KP_E_UNSUPPORTED;
    PatchArgs a;
    a.x = (const h16_t*)x; a.w = (const h16_t*)w; a.bias = bias; a.y = (h16_t*)y; a.partials = stats;
    a.Hib = Hib; a.Wib = Wib; a.ioff = ioff; a.Hob = Hob; a.Wob = Wob; a.ooff = ooff; a.H = H; a.W = W;
    a.C = C; a.K = K;
    a.yred = nullptr; a.mean = a.invstd = a.gamma = a.beta = nullptr; a.slope = 0.f; a.rpart = nullptr;
    { extern kp_knob_t g_kp_debug; extern kp_knob_ptr_t g_kp_trace; a.dbg = g_kp_debug; a.trace = g_kp_trace; }
    { extern kp_knob_t g_kp_walk; a.linear = (g_kp_walk & KP_WALK_CONV_LINEAR) ? 1 : 0; }
    const long long ptiles = (long long)N * (H / TS) * (W / TS);
    const int bn = (K % 128 == 0) ? 128 : 64;
    a.ktiles = K / bn;
    if (ptiles * a.ktiles > 0x7fffffffLL) return KP_E_UNSUPPORTED;
    a.ntiles = (int)(ptiles * a.ktiles);
    int cus = kp_device_cus();   // (the CU budget of the launching lane: kp_config(KP_CFG_CUS))
    if (cus <= 0) cus = 256;
    // one persistent workgroup per CU (134 KB of LDS each), a multiple of the 8 XCDs
    int grid = (cus / 8) * 8;
    if (grid < 8) grid = 8;
    const int need = ((a.ntiles + 7) / 8) * 8;
    if (grid > need) grid = need;
    a.compact = (stats && ptiles >= grid && ((grid / 8) % a.ktiles) == 0) ? 1 : 0;
    if (stats) kp_set_last_rows(a.compact ? (long long)grid * CROWS : ptiles * 4);
    a.fin_tickets = nullptr; a.fin_mean = a.fin_invstd = a.fin_rmean = a.fin_rvar = nullptr; a.fin_nbt = nullptr; a.fin_momentum = a.fin_eps = 0.f; a.fin_count = 1;
    a.ep_scale = a.ep_shift = nullptr; a.ep_slope = 1.f;
    if (KpFinRequest* rq = kp_fin_request()) {   // kp_conv_fwd_fin: the finalize in this launch's tail (one row per workgroup, <= KP_FIN_TICKETS k-tiles)
        if (stats && a.compact && a.linear && CROWS == 1 && a.ktiles <= KP_FIN_TICKETS && !a.trace && !a.dbg) {
            a.fin_tickets = rq->tickets; a.fin_mean = rq->mean; a.fin_invstd = rq->invstd; a.fin_rmean = rq->running_mean; a.fin_rvar = rq->running_var;
            a.fin_nbt = rq->nbt; a.fin_momentum = rq->momentum; a.fin_eps = rq->eps; a.fin_count = rq->count;
            rq->taken = 1;
        }
    }
    KP_QUERY_RETURN();
    extern kp_knob_t g_kp_stats_store;
    if ((a.trace || a.dbg) && bn == 128 && stats) conv3x3_patch_k<128, true, true><<<grid, TPB, 0, st>>>(a);
    else if (bn == 128 && stats && a.compact && CROWS == 1 && g_kp_stats_store) conv3x3_patch_k<128, true, false, false, false, true><<<grid, TPB, 0, st>>>(a);
    else if (bn == 128 && stats) conv3x3_patch_k<128, true, false><<<grid, TPB, 0, st>>>(a);
    else if (bn == 128) conv3x3_patch_k<128, false, false><<<grid, TPB, 0, st>>>(a);
    else if (stats) conv3x3_patch_k<64, true, false><<<grid, TPB, 0, st>>>(a);
    else conv3x3_patch_k<64, false, false><<<grid, TPB, 0, st>>>(a);
    return kp_check_launch("conv3x3_patch");
}

// Inference form (kp_conv_fwd_act): the window convolution of kp_conv_patch_launch with the per-channel affine + LeakyReLU epilogue (EPI);
// no bias operand (folded into `shift`), no statistics.  KP_E_UNSUPPORTED where the shape has no patch tiling.
int kp_conv_patch_launch_act(const void* x, const void* w, const float* scale, const float* shift, float slope, void* y, int N, int Hib,
                             int Wib, int ioff, int Hob, int Wob, int ooff, int H, int W, int C, int K, hipStream_t st) {
    if (C % 64 != 0 || K % 64 != 0 || H % TS != 0 || W % TS != 0 || H <= 0 || W <= 0) return KP_E_UNSUPPORTED;
    if ((long long)(PW + 1) * Wib * C * 2 >= 0x7fffffffLL || 128LL * 9 * C * 2 + 9LL * C * 2 >= 0x7fffffffLL) return KP_E_UNSUPPORTED;
    PatchArgs a;
    a.x = (const h16_t*)x; a.w = (const h16_t*)w; a.bias = nullptr; a.y = (h16_t*)y; a.partials = nullptr;
    a.Hib = Hib; a.Wib = Wib; a.ioff = ioff; a.Hob = Hob; a.Wob = Wob; a.ooff = ooff; a.H = H; a.W = W;
    a.C = C; a.K = K; a.dbg = 0; a.trace = nullptr; a.compact = 0;
    a.yred = nullptr; a.mean = a.invstd = a.gamma = a.beta = nullptr; a.slope = 0.f; a.rpart = nullptr;
    a.fin_tickets = nullptr; a.fin_mean = a.fin_invstd = a.fin_rmean = a.fin_rvar = nullptr; a.fin_nbt = nullptr; a.fin_momentum = a.fin_eps = 0.f; a.fin_count = 1;
    a.ep_scale = scale; a.ep_shift = shift; a.ep_slope = slope;
    { extern kp_knob_t g_kp_walk; a.linear = (g_kp_walk & KP_WALK_CONV_LINEAR) ? 1 : 0; }
    const long long ptiles = (long long)N * (H / TS) * (W / TS);
    const int bn = (K % 128 == 0) ? 128 : 64;
    a.ktiles = K / bn;
    if (ptiles * a.ktiles > 0x7fffffffLL) return KP_E_UNSUPPORTED;
    a.ntiles = (int)(ptiles * a.ktiles);
    int cus = kp_device_cus();
    if (cus <= 0) cus = 256;
    int grid = (cus / 8) * 8;
    if (grid < 8) grid = 8;
    const int need = ((a.ntiles + 7) / 8) * 8;
    if (grid > need) grid = need;
    KP_QUERY_RETURN();
    if (bn == 128) conv3x3_patch_k<128, false, false, false, true><<<grid, TPB, 0, st>>>(a);
    else conv3x3_patch_k<64, false, false, false, true><<<grid, TPB, 0, st>>>(a);
    return kp_check_launch("conv3x3_patch_act");
}
