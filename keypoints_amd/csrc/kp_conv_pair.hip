// 16-bit MFMA 3x3 convolution (forward / data-gradient), second generation of the LDS-patch kernel -- gfx950.
//
// kp_conv_patch.hip runs at the package power cap: what moves it is energy per FLOP, and its
// ablations say where the energy goes besides the MFMAs -- LDS fragment reads (half of them gone: +9 %) and L2 -> LDS
// DMA (none: +10 %).  This kernel keeps that kernel's structure (persistent workgroups, one software pipeline over
// (tile, chunk, tap) steps, LDS-DMA with counted vmcnt, a 4-slot weight ring, swizzled conflict-free fragment reads)
// and changes the tiling so that both go down per MFMA:
//
//   workgroup tile = TWO 16 x 16 pixel tiles (consecutive pixel-tile ids: the right-hand neighbour, or -- at the end of
//                    a tile row / for 16 x 16 maps -- the next row's / next image's first tile) x BN output channels
//   wave tile      = 128 pixels (4 rows x 16 columns of BOTH pixel tiles) x 64 channels: per 16-channel reduction slice
//                    4 pixel fragments + 2 weight fragments feed 8 MFMAs (0.75 ds_read_b128 per MFMA; was 1.0)
//   reduction      = 32-channel chunks (64-byte LDS rows): both 18 x 18 patches of a chunk are 42 KB, double buffered;
//                    a step = (tap, chunk) = 2 reduction slices = 16 MFMAs per wave, as before
//   weights        = W[k0 .. k0+BN)[tap][c0 .. c0+32): 8 KB per step for 512 pixels (was 16 KB per step for 256 pixels)
//
// Per MFMA: fragment reads x 0.75, weight DMA x 0.5, patch DMA unchanged.  The fragment sets shrink from four to two
// (a step has two slices: one is multiplied while the other is read), which is what pays for the 128 accumulator
// registers: 128 + 48 + addresses.
// LDS rows are 64 B (32 channels of one pixel / one output channel): the 16-byte slot index is XOR-swizzled by
// (patch column >> 1) & 3 (pixels) / (row >> 2) & 3 (weights), on the DMA source address and on the ds_read_b128
// address: every ds_read_b128 lane group hits 16 distinct slots for all 9 tap shifts (exhaustive check:
// tools/probes/swizzle64.py).
// BatchNorm statistics (forward): the per-lane sums of a tile are folded over the 32 pixels of a lane half by the
// halving butterfly once per TILE and accumulated in two registers per 32-channel block -- the 64 per-lane sum
// registers of the first-generation kernel do not fit beside 128 accumulators.
#include "kp_mfma_util.h"

namespace {

constexpr int TPB = 512, NWAVE = 8;
constexpr int TS = 16;                  // tile side (pixels)
constexpr int PW = TS + 2;              // patch side
constexpr int PPIX = PW * PW;           // 324 pixels per patch
constexpr int ROWB = 64;                // bytes per LDS row (32 x 16 bit)
constexpr int CH = 32;                  // channels per chunk
constexpr int SUBI = (PPIX + 15) / 16;  // 21 DMA instructions of 16 pixels (1 KiB) per patch
constexpr int SUBB = SUBI * 1024;       // bytes per patch in LDS
constexpr int PINSTR = 2 * SUBI;        // 42 per chunk
constexpr int PBYTES = 2 * SUBB;        // both patches of a chunk
constexpr int PSLICES = (PINSTR + NWAVE - 1) / NWAVE;   // patch DMA instructions per wave per chunk (6)
constexpr int NWBUF = 4;                // weight ring slots

struct PairArgs {
    const h16_t* x; const h16_t* w; const float* bias; h16_t* y; float* partials;
    int Hib, Wib, ioff;   // input buffer rows / cols, offset of the tile domain's patch origin in it
    int Hob, Wob, ooff;   // output buffer rows / cols, offset of the tile domain in it
    int H, W;             // tile domain (multiples of 16)
    int C, K, ktiles, ntiles;
    int linear;           // 1: tile ids in time order (step i of the grid covers ids [i*grid, (i+1)*grid)); 0: 8 XCD ranges
};

// Step t of the pipeline (16 MFMA per wave: slices kk = 0, 1 of 16 channels; fragment sets A = kk 0, B = kk 1):
//     read B(t)                      ||  MFMA A(t)          (A(t) was read during step t-1)
//     s_waitcnt vmcnt(#DMA issued after the previous barrier); s_barrier     <- weights of step t+1 landed
//     MFMA B(t), first weight block
//     issue DMA: weight tile of step t+3 -> ring slot (t-1) % 4, one slice of the next chunk's patches
//     read A(t+1)                    ||  MFMA B(t), second weight block
template <int BN, bool STATS>
__global__ __launch_bounds__(TPB, 2) void conv3x3_pair_k(const PairArgs A) {
    constexpr int WBYTES = BN * ROWB;
    constexpr int WINSTR = WBYTES / 1024;   // weight DMA instructions per step (8 or 4): wave i issues instruction i
    constexpr int WN = BN / 2;              // channels per wave
    constexpr int TN = WN / 32;             // 32-channel MFMA blocks per wave
    constexpr int OFF_W = 2 * PBYTES, OFF_DUMMY = OFF_W + NWBUF * WBYTES;
    constexpr int OFF_BIAS = OFF_DUMMY + 1024;            // [8 waves][64] fp32: each wave's own output-channel biases
    constexpr int SMEM = OFF_BIAS + NWAVE * 64 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int C = A.C, K = A.K;

    const int per_xcd = gridDim.x >> 3, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int Q = (A.ntiles + 7) >> 3;
    const int tile_hi = A.linear ? A.ntiles : ((xcd + 1) * Q < A.ntiles ? (xcd + 1) * Q : A.ntiles);
    const int tstride = A.linear ? (int)gridDim.x : per_xcd;
    int tile = A.linear ? xcd * per_xcd + jb : xcd * Q + jb;
    if (STATS) {   // this workgroup's 4 rows: zero everywhere, its own k-tile columns are set at the end
        float* rowp = A.partials + (long long)blockIdx.x * 4 * 2 * K;
        for (int i = tid; i < 4 * 2 * K; i += TPB) rowp[i] = 0.f;
    }
    if (tile >= tile_hi) return;
    const int tiles_x = A.W / TS, tiles_y = A.H / TS;

    struct Sub { int n, ty, tx; };
    struct TileRef { Sub s[2]; int kt, k0; };
    auto inc = [&](Sub c, int dx, int dy, int dn) {   // (tx, ty, n) + (dx, dy, dn) with carries; dx < tiles_x + 3, dy < tiles_y
        Sub t;
        int cy = 0;
        t.tx = c.tx + dx;
        while (t.tx >= tiles_x) { t.tx -= tiles_x; ++cy; }
        t.ty = c.ty + dy + cy; cy = 0;
        while (t.ty >= tiles_y) { t.ty -= tiles_y; ++cy; }
        t.n = c.n + dn + cy;
        return t;
    };
    auto decode = [&](int id) {   // divisions: only for a workgroup's first tile
        TileRef t;
        t.kt = id % A.ktiles;
        const int pt = (id / A.ktiles) * 2;
        t.s[0].tx = pt % tiles_x; t.s[0].ty = (pt / tiles_x) % tiles_y; t.s[0].n = pt / (tiles_x * tiles_y);
        t.s[1] = inc(t.s[0], 1, 0, 0);
        t.k0 = t.kt * BN;
        return t;
    };
    // a workgroup's next tile is `tstride` ids further: the mixed-radix digits of that stride, added with carries
    const int d_kt = tstride % A.ktiles, d_pt = (tstride / A.ktiles) * 2;
    const int d_tx = d_pt % tiles_x, d_ty = (d_pt / tiles_x) % tiles_y, d_n = d_pt / (tiles_x * tiles_y);
    auto advance = [&](const TileRef& c) {
        TileRef t;
        t.kt = c.kt + d_kt;
        const int carry = t.kt >= A.ktiles;
        t.kt -= carry ? A.ktiles : 0;
        t.s[0] = inc(c.s[0], d_tx + 2 * carry, d_ty, d_n);
        t.s[1] = inc(t.s[0], 1, 0, 0);
        t.k0 = t.kt * BN;
        return t;
    };
    auto x_rsrc = [&](const Sub& s) {
        const unsigned char* xbase = (const unsigned char*)A.x +
            ((((long long)s.n * A.Hib + s.ty * TS + A.ioff) * A.Wib) + s.tx * TS + A.ioff) * (long long)C * 2;
        return __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, 0x7ffffffe, 0x00020000);
    };
    auto w_rsrc = [&](const TileRef& t) {
        const unsigned char* wbase = (const unsigned char*)A.w + (long long)t.k0 * 9 * C * 2;
        return __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7ffffffe, 0x00020000);
    };

    // ---- per-lane DMA offsets (tile independent) --------------------------------------------------------
    // one instruction = 16 LDS rows of 64 B: lane -> (row lane >> 2, 16-byte slot lane & 3)
    const int lrow = lane >> 2, lslot = lane & 3;
    int pvoff[PSLICES];   // patches: instruction ii = j*8 + wave -> patch ii / 21, its pixels (ii % 21) * 16 .. + 15
#pragma unroll
    for (int j = 0; j < PSLICES; ++j) {
        const int ii = j * NWAVE + wave;
        int pi = (ii % SUBI) * 16 + lrow;
        if (pi > PPIX - 1) pi = PPIX - 1;
        const int py = pi / PW, px = pi - py * PW;
        pvoff[j] = (py * A.Wib + px) * C * 2 + ((lslot ^ ((px >> 1) & 3)) << 4);
    }
    int wvoff;            // weights: instruction `wave` covers rows wave*16 .. + 15 of the k-tile (waves >= WINSTR: a dummy target)
    {
        const int row = (wave < WINSTR ? wave : 0) * 16 + lrow;
        wvoff = row * 9 * C * 2 + ((lslot ^ ((row >> 2) & 3)) << 4);
    }
    auto issue_w = [&](const __amdgpu_buffer_rsrc_t& wr, int buf, int tap, int c0, bool real) {
        unsigned char* dst = (real && wave < WINSTR) ? smem + OFF_W + buf * WBYTES + wave * 1024 : smem + OFF_DUMMY;
        dma16(wr, dst, wvoff, (tap * C + c0) * 2);
    };
    auto issue_p = [&](const __amdgpu_buffer_rsrc_t& x0, const __amdgpu_buffer_rsrc_t& x1, int pbuf, int j, int c0, bool real) {
        const int ii = j * NWAVE + wave;
        unsigned char* dst = (real && ii < PINSTR) ? smem + pbuf * PBYTES + ii * 1024 : smem + OFF_DUMMY;
        dma16(ii < SUBI ? x0 : x1, dst, pvoff[j], c0 * 2);
    };

    // ---- fragment read offsets --------------------------------------------------------------------------
    const int fr = lane & 31, fh = lane >> 5;
    const int ftx = fr & 15;
    int xoff[4];      // byte offset of this lane's pixel (tap 0,0) inside the chunk's patch pair; b = 2 * patch + row pair
#pragma unroll
    for (int b = 0; b < 4; ++b) xoff[b] = (b >> 1) * SUBB + ((wm * 4 + (b & 1) * 2 + (fr >> 4)) * PW + ftx) * ROWB;
    int xs16[3];      // swizzled 16-B slot of kk = 0 for column shift s
#pragma unroll
    for (int s = 0; s < 3; ++s) xs16[s] = (fh ^ (((ftx + s) >> 1) & 3)) << 4;
    int woff[TN];     // row offset + swizzled slot of kk = 0
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int row = wn * WN + a * 32 + fr;
        woff[a] = row * ROWB + ((fh ^ ((row >> 2) & 3)) << 4);
    }

    struct Frag { h16x8 x[4]; h16x8 w[TN]; };
    auto load_frag = [&](Frag& f, const unsigned char* xb, const unsigned char* wb, int s, int kk) {
        // in the order the MFMAs want them: (w0, x0), x1, x2, x3, w1
        f.w[0] = *(const h16x8*)(wb + (woff[0] ^ (kk << 5)));
#pragma unroll
        for (int b = 0; b < 4; ++b) f.x[b] = *(const h16x8*)(xb + xoff[b] + (xs16[s] ^ (kk << 5)));
#pragma unroll
        for (int a = 1; a < TN; ++a) f.w[a] = *(const h16x8*)(wb + (woff[a] ^ (kk << 5)));
    };

    f32x16 acc[TN][4];
    auto init_acc = [&]() {   // accumulators start at the bias of the lane's output channels (wave-private LDS copy)
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = *reinterpret_cast<const float4*>(smem + OFF_BIAS + (wave * 64 + a * 32 + 8 * g + 4 * fh) * 4);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    acc[a][b][g * 4 + 0] = bq.x; acc[a][b][g * 4 + 1] = bq.y;
                    acc[a][b][g * 4 + 2] = bq.z; acc[a][b][g * 4 + 3] = bq.w;
                }
            }
    };
    auto mma_a = [&](const Frag& f, int a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = kp_mfma16(f.w[a], f.x[b], acc[a][b], 0, 0, 0);
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int a = 0; a < TN; ++a) mma_a(f, a);
    };

    // epilogue: D[k_out][pixel]; lane: pixel = fr of fragment b, k_out = 8*(reg>>2) + 4*fh + (reg&3).  A wave's 32-pixel x
    // WN-channel fragments go through a wave-private XOR-swizzled staging tile in the patch buffer the finished chunk no
    // longer needs and leave as 16-byte lanes over whole runs per pixel.
    constexpr int CPP = WN / 8;            // 16-byte chunks per pixel (this wave's channels)
    constexpr int PPI = 64 / CPP;          // pixels per store instruction
    float psum[STATS ? TN : 1], psq[STATS ? TN : 1];   // this lane pair's channel (see reduce16_over32), summed over the workgroup's tiles
    if constexpr (STATS) {
#pragma unroll
        for (int a = 0; a < TN; ++a) psum[a] = psq[a] = 0.f;
    }
    const int st_voff = ((lane / CPP) * K + (lane % CPP) * 8) * 2;   // lane part of the output address (bytes)
    auto epilogue = [&](const TileRef& t, int free_pb) {
        unsigned char* stg = smem + free_pb * PBYTES + wave * (32 * WN * 2);
        __amdgpu_buffer_rsrc_t yr[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            h16_t* ybase = A.y + ((((long long)t.s[p].n * A.Hob + t.s[p].ty * TS + wm * 4 + A.ooff) * A.Wob) + t.s[p].tx * TS + A.ooff) * (long long)K +
                           t.k0 + wn * WN;
            yr[p] = __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, 0x7ffffffe, 0x00020000);
        }
        if constexpr (STATS) {
            // sums over the lane's 4 pixels per register, then over the 32 lanes of its half (VALU permutes, no LDS)
#pragma unroll
            for (int a = 0; a < TN; ++a) {
                float s16[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) s16[i] = (acc[a][0][i] + acc[a][1][i]) + (acc[a][2][i] + acc[a][3][i]);
                int r1;
                psum[a] += reduce16_over32(s16, lane, r1);
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    s16[i] = __builtin_fmaf(acc[a][0][i], acc[a][0][i], acc[a][1][i] * acc[a][1][i]) +
                             __builtin_fmaf(acc[a][2][i], acc[a][2][i], acc[a][3][i] * acc[a][3][i]);
                psq[a] += reduce16_over32(s16, lane, r1);
            }
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
#pragma unroll
            for (int a = 0; a < TN; ++a) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 pk;
                    pk.x = pack_h16x2(acc[a][b][g * 4 + 0], acc[a][b][g * 4 + 1]);
                    pk.y = pack_h16x2(acc[a][b][g * 4 + 2], acc[a][b][g * 4 + 3]);
                    const int chunk = a * 4 + g;
                    *reinterpret_cast<uint2*>(stg + fr * (WN * 2) + ((chunk ^ (fr & (CPP - 1))) << 4) + fh * 8) = pk;
                }
            }
#pragma unroll
            for (int i = 0; i < 32 * CPP / 64; ++i) {
                const int id = i * 64 + lane, px = id / CPP, c16 = id % CPP;
                const u32x4 vv = *reinterpret_cast<const u32x4*>(stg + px * (WN * 2) + ((c16 ^ (px & (CPP - 1))) << 4));
                const int soff = (((b & 1) * 2 + ((i * PPI) >> 4)) * A.Wob + ((i * PPI) & 15)) * K * 2;
                __builtin_amdgcn_raw_buffer_store_b128(vv, yr[b >> 1], st_voff, soff, 0);
            }
        }
    };

    const int nchunks = C / CH;
    TileRef cur = decode(tile);
    __amdgpu_buffer_rsrc_t xr0 = x_rsrc(cur.s[0]), xr1 = x_rsrc(cur.s[1]), wr = w_rsrc(cur);
    auto load_bias = [&](const TileRef& t) {
        if (lane < WN) {
            const float bvl = A.bias ? A.bias[t.k0 + wn * WN + lane] : 0.f;
            reinterpret_cast<float*>(smem + OFF_BIAS)[wave * 64 + lane] = bvl;
        }
    };
    load_bias(cur);

    // ---- prologue: patches of chunk 0, weight tiles of steps 0..2 of the first tile ------------------------
#pragma unroll
    for (int j = 0; j < PSLICES; ++j) issue_p(xr0, xr1, 0, j, 0, true);
    issue_w(wr, 0, 0, 0, true);
    issue_w(wr, 1, 1, 0, true);
    issue_w(wr, 2, 2, 0, true);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    Frag fA, fB;
    load_frag(fA, smem, smem + OFF_W, 0, 0);
    init_acc();
    int pbi = 0;       // patch buffer of the current chunk
    int wslot = 0;     // weight ring slot of the current step
    bool after_epi = false;

    while (true) {
        const int ntile_id = tile + tstride;
        const bool has_next = ntile_id < tile_hi;
        TileRef nxt = cur;
        __amdgpu_buffer_rsrc_t xrn0 = xr0, xrn1 = xr1, wrn = wr;
        if (has_next) { nxt = advance(cur); xrn0 = x_rsrc(nxt.s[0]); xrn1 = x_rsrc(nxt.s[1]); wrn = w_rsrc(nxt); }

        for (int ch = 0; ch < nchunks; ++ch) {
            const unsigned char* pb = smem + pbi * PBYTES;
            const unsigned char* npb = smem + (pbi ^ 1) * PBYTES;
            const bool last_ch = ch + 1 == nchunks;
            const bool more_p = !last_ch || has_next;          // is there a "next chunk" to prefetch patches for
            static_for<0, 9>([&](auto tapc) {
                constexpr int tap = decltype(tapc)::value;
                constexpr int r = tap / 3, s = tap % 3;
                const unsigned char* xb = pb + (r * PW + s) * ROWB;
                const unsigned char* wb = smem + OFF_W + wslot * WBYTES;
                constexpr int ntap = (tap + 1) % 9, nr = ntap / 3, ns = ntap % 3;
                const unsigned char* nxb = (tap == 8 ? npb : pb) + (nr * PW + ns) * ROWB;
                const unsigned char* nwb = smem + OFF_W + ((wslot + 1) & (NWBUF - 1)) * WBYTES;

                // -- first half: slice 0 from registers, the reads of slice 1 under it
                // (the reads are fenced in front of the MFMAs: left to itself the scheduler sinks them behind the MFMAs that free
                //  their registers -- one fragment set instead of two -- and every half step then starts with an LDS round trip)
                load_frag(fB, xb, wb, s, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(fA);
                __builtin_amdgcn_sched_barrier(0);
                // No LDS wait in front of the barrier: the ring slot that is overwritten after it is the one of step t-1, and
                // every wave consumed its reads of it (MFMA operands) before it can arrive.  Only the last step of a tile
                // waits, because the epilogue reuses the patch buffer that other waves may still be reading.
                if (tap == 8 && last_ch) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // Everything issued before the previous barrier has landed (weights of step t+1, older patch slices): the
                // DMA count after a barrier is a compile-time constant per tap (dummy targets keep it so).  Right after an
                // epilogue everything older was already drained and its stores must not be waited for here.
                if (!(tap == 0 && after_epi)) wait_vm<1 + (((tap + 8) % 9) < PSLICES ? 1 : 0)>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);

                // -- second half: the first weight block's MFMAs keep the matrix pipe fed while the DMA instructions issue
                mma_a(fB, 0);
                __builtin_amdgcn_sched_barrier(0);
                // weight tile of step t+3 into the ring slot of step t-1 (the last step of a tile defers it until after
                // the epilogue), one slice of the next patches in taps 0..5; no-op cases go to a dummy LDS target
                if (!(tap == 8 && last_ch)) {
                    constexpr int tap3 = (tap + 3) % 9;
                    constexpr bool wrap = tap + 3 >= 9;
                    const bool to_next = wrap && last_ch;
                    const bool real = !to_next || has_next;
                    const int cw = to_next ? 0 : (wrap ? ch + 1 : ch) * CH;
                    issue_w(to_next ? wrn : wr, (wslot + 3) & (NWBUF - 1), tap3, cw, real);
                }
                if constexpr (tap < PSLICES)
                    issue_p(last_ch ? xrn0 : xr0, last_ch ? xrn1 : xr1, pbi ^ 1, tap, last_ch ? 0 : (ch + 1) * CH, more_p);
                __builtin_amdgcn_sched_barrier(0);
                load_frag(fA, nxb, nwb, ns, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (TN > 1) mma_a(fB, 1);
                __builtin_amdgcn_sched_barrier(0);
                wslot = (wslot + 1) & (NWBUF - 1);
            });
            pbi ^= 1;
            after_epi = false;
        }
        // tile switch: drain the DMA queue (everything in it was issued >= half a step ago), store the tile, then issue
        // the weight tile that the last step deferred.  The next barrier does not wait (after_epi); the one after it
        // retires these stores together with the DMA.
        wait_vm<0>();
        epilogue(cur, pbi ^ 1);   // pbi = buffer of the NEXT tile's first patches; the other one is free
        if (!has_next) break;
        if (nxt.k0 != cur.k0) load_bias(nxt);   // read again only at the next epilogue (many barriers later)
        issue_w(wrn, (wslot + 2) & (NWBUF - 1), 2, 0, true);   // wslot = slot of the next tile's step 0
        after_epi = true;
        init_acc();
        tile = ntile_id; cur = nxt; xr0 = xrn0; xr1 = xrn1; wr = wrn;
    }
    if constexpr (STATS) {
#pragma unroll
        for (int a = 0; a < TN; ++a) {
            if ((lane & 1) == 0) {
                const int rg = ((lane & 16) ? 8 : 0) | ((lane & 8) ? 4 : 0) | ((lane & 4) ? 2 : 0) | ((lane & 2) ? 1 : 0);
                const int k = cur.k0 + wn * WN + a * 32 + 8 * (rg >> 2) + 4 * fh + (rg & 3);
                const long long prow = (long long)blockIdx.x * 4 + wm;
                A.partials[(prow * 2 + 0) * K + k] = psum[a];
                A.partials[(prow * 2 + 1) * K + k] = psq[a];
            }
        }
    }
}

}  // namespace

// Eligibility + grid of the pair kernel for a tile domain (shared by the launcher and the row-count query).
static bool pair_plan(const float* stats, int N, int Hib, int Wib, int H, int W, int C, int K, int& ktiles, int& ntiles, int& grid) {
    if (C % CH != 0 || C < 64 || K % 64 != 0 || H % TS != 0 || W % TS != 0 || H <= 0 || W <= 0) return false;
    if ((long long)(PW + 1) * Wib * C * 2 >= 0x7fffffffLL || 128LL * 9 * C * 2 + 9LL * C * 2 >= 0x7fffffffLL) return false;
    const long long ptiles = (long long)N * (H / TS) * (W / TS);
    if (ptiles % 2 != 0) return false;
    const int bn = (K % 128 == 0) ? 128 : 64;
    ktiles = K / bn;
    if ((ptiles / 2) * ktiles > 0x7fffffffLL) return false;
    ntiles = (int)((ptiles / 2) * ktiles);
    int cus = kp_device_cus();
    if (cus <= 0) cus = 256;
    grid = (cus / 8) * 8;
    if (grid < 8) grid = 8;
    const int need = ((ntiles + 7) / 8) * 8;
    if (grid > need) grid = need;
    // statistics: one set of 4 rows per workgroup, which must then keep ONE k-tile over all of its tiles
    // (more than one tile per workgroup: its stride through the ids must be a multiple of the k-tile count)
    if (stats && ntiles > grid && !(((grid / 8) % ktiles) == 0 && (grid % ktiles) == 0)) return false;
    // the caller's statistics buffer holds ceil(N H W / 64) rows (include/kp_hip.h): 4 rows per workgroup must fit
    if (stats && (long long)grid * 4 > ((long long)N * H * W + 63) / 64) return false;
    return true;
}

// Same contract as kp_conv_patch_launch (kp_conv_patch.hip).  KP_E_UNSUPPORTED: shape not eligible (odd number of pixel
// tiles, too few tiles for per-workgroup statistics rows, ...): the caller then takes the first-generation kernel.
int kp_conv_pair_launch(const void* x, const void* w, const float* bias, void* y, float* stats, int N, int Hib,
                        int Wib, int ioff, int Hob, int Wob, int ooff, int H, int W, int C, int K, hipStream_t st) {
    PairArgs a;
    int grid = 0;
    if (!pair_plan(stats, N, Hib, Wib, H, W, C, K, a.ktiles, a.ntiles, grid)) return KP_E_UNSUPPORTED;
    a.x = (const h16_t*)x; a.w = (const h16_t*)w; a.bias = bias; a.y = (h16_t*)y; a.partials = stats;
    a.Hib = Hib; a.Wib = Wib; a.ioff = ioff; a.Hob = Hob; a.Wob = Wob; a.ooff = ooff; a.H = H; a.W = W;
    a.C = C; a.K = K;
    { extern kp_knob_t g_kp_walk; a.linear = (g_kp_walk & KP_WALK_CONV_LINEAR) ? 1 : 0; }
    if (stats) kp_set_last_rows((long long)grid * 4);
    KP_QUERY_RETURN();
    const int bn = (K % 128 == 0) ? 128 : 64;
    if (bn == 128 && stats) conv3x3_pair_k<128, true><<<grid, TPB, 0, st>>>(a);
    else if (bn == 128) conv3x3_pair_k<128, false><<<grid, TPB, 0, st>>>(a);
    else if (stats) conv3x3_pair_k<64, true><<<grid, TPB, 0, st>>>(a);
    else conv3x3_pair_k<64, false><<<grid, TPB, 0, st>>>(a);
    return kp_check_launch("conv3x3_pair");
}
