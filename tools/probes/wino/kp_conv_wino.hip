// fp16 MFMA 3x3 convolution (forward / data-gradient) with 1.5x fewer multiplications: Winograd F(2, 3) along the image
// columns' ROW direction (output rows in pairs), on the structure of kp_conv_pair.hip -- gfx950.
//
// The matrix-core kernels of the step run at the package power cap: time = joules, and three quarters
// of the joules are MACs.  F(2, 3) computes two output rows from four input rows with 4 products per filter column
// instead of 6:
//     V_0 = d_0 - d_2,  V_1 = d_1 + d_2,  V_2 = d_2 - d_1,  V_3 = d_1 - d_3            (d_j: input row 2 t + j)
//     U_0 = g_0,  U_1 = (g_0 + g_1 + g_2) / 2,  U_2 = (g_0 - g_1 + g_2) / 2,  U_3 = g_2  (g_r: filter row r; packed once per step)
//     M_p[t][x][k] = sum_{s, c} V_p[t][x + s][c] U_p[s][k][c]                             (p = 0..3: the 16-bit MFMA operands)
//     y[2 t][x] = M_0 + M_1 + M_2,   y[2 t + 1][x] = M_1 - M_2 - M_3                       (fp32, in registers)
// The transform runs along the patch ROWS so that the column shift s of a tap stays what it is in the direct kernels (the
// same conflict-free column swizzle), and a lane's four input rows are four ds_read_b128 at row offsets.  V is formed in
// registers from the patch fragments (v_pk_add_f16: one rounding, exactly "round(d_a +- d_b)"); it never exists in LDS.
//
//   workgroup tile = two 16 x 16 pixel tiles (as kp_conv_pair.hip) x 64 output channels, 8 wavefronts = 4 row groups x 2 tiles
//   wave tile      = 32 row pairs (2 pairs x 16 columns) x 64 channels x 4 products = 8 accumulator blocks of 32 x 32
//   step           = (filter column s, 32-channel chunk): 16 MFMAs per wave (2 slices x 4 p x 2 channel blocks)
//   weights        = U[k0 .. k0+64)[p][s][c0 .. c0+32) = 16 KB per step through a 4-slot LDS ring
//   patches        = 2 x 18 x 18 x 32 channels (42 KB), double buffered
// Numerics (tools/probes/winograd_parity.py, conditioned model F, fp16 storage, 128 pairs): keypoints 3.6e-3 - 4.2e-3 from the
// fp64 oracle against 3.5e-3 for the direct kernels (stated tolerance 5e-3); fp16 only (the bf16 build has no packed adds and
// no headroom).  Reference statement replaced: nn.Conv2d(3x3) after ReplicationPad2d(1), keypoints/models/vgg.py:30-33.
#include "kp_mfma_util.h"

#ifdef KP_F16_BUILD

namespace {

constexpr int TPB = 512, NWAVE = 8;
constexpr int TS = 16, PW = TS + 2, PPIX = PW * PW;
constexpr int ROWB = 64, CH = 32;
constexpr int SUBI = (PPIX + 15) / 16, SUBB = SUBI * 1024;
constexpr int PINSTR = 2 * SUBI, PBYTES = 2 * SUBB;
constexpr int PSL = 6;                  // patch DMA instructions per wave per chunk (8 waves x 6 = 48 >= 42), all issued in the chunk's
                                        // first step: the next chunk's patches are first read in the chunk's LAST step, and a DMA is
                                        // known to have landed two barriers after the one it was issued behind
constexpr int NWBUF = 4;
constexpr int BN = 64;                  // output channels per workgroup tile
constexpr int WROWS = 4 * BN;           // LDS rows of a weight tile: p * 64 + k
constexpr int WBYTES = WROWS * ROWB;    // 16 KB
constexpr int NWI = WBYTES / 1024 / NWAVE;   // weight DMA instructions per wave per step (2)

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

struct WinoArgs {
    const h16_t* x; const h16_t* u; const float* bias; h16_t* y; float* partials;
    int Hib, Wib, ioff, Hob, Wob, ooff, H, W;
    int C, K, ktiles, ntiles, linear;
};

__device__ __forceinline__ h16x8 hadd8(h16x8 a, h16x8 b) {
    return __builtin_bit_cast(h16x8, __builtin_bit_cast(half8_t, a) + __builtin_bit_cast(half8_t, b));
}
__device__ __forceinline__ h16x8 hsub8(h16x8 a, h16x8 b) {
    return __builtin_bit_cast(h16x8, __builtin_bit_cast(half8_t, a) - __builtin_bit_cast(half8_t, b));
}

template <bool STATS>
__global__ __launch_bounds__(TPB, 2) void conv3x3_wino_k(const WinoArgs A) {
    constexpr int OFF_W = 2 * PBYTES, OFF_DUMMY = OFF_W + NWBUF * WBYTES;
    constexpr int OFF_BIAS = OFF_DUMMY + 1024;            // [8 waves][64] fp32: every wave's own copy of the k-tile's biases
    constexpr int SMEM = OFF_BIAS + NWAVE * 64 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, sub = wave & 1;
    const int C = A.C, K = A.K;

    const int per_xcd = gridDim.x >> 3, xcd = blockIdx.x & 7, jb = blockIdx.x >> 3;
    const int Q = (A.ntiles + 7) >> 3;
    const int tile_hi = A.linear ? A.ntiles : ((xcd + 1) * Q < A.ntiles ? (xcd + 1) * Q : A.ntiles);
    const int tstride = A.linear ? (int)gridDim.x : per_xcd;
    int tile = A.linear ? xcd * per_xcd + jb : xcd * Q + jb;
    if (STATS) {   // this workgroup's 8 rows (one per wave): zero everywhere, its own k-tile columns are set at the end
        float* rowp = A.partials + (long long)blockIdx.x * NWAVE * 2 * K;
        for (int i = tid; i < NWAVE * 2 * K; i += TPB) rowp[i] = 0.f;
    }
    if (tile >= tile_hi) return;
    const int tiles_x = A.W / TS, tiles_y = A.H / TS;

    struct Sub { int n, ty, tx; };
    struct TileRef { Sub s[2]; int kt, k0; };
    auto inc = [&](Sub c, int dx, int dy, int dn) {
        Sub t;
        int cy = 0;
        t.tx = c.tx + dx;
        while (t.tx >= tiles_x) { t.tx -= tiles_x; ++cy; }
        t.ty = c.ty + dy + cy; cy = 0;
        while (t.ty >= tiles_y) { t.ty -= tiles_y; ++cy; }
        t.n = c.n + dn + cy;
        return t;
    };
    auto decode = [&](int id) {
        TileRef t;
        t.kt = id % A.ktiles;
        const int pt = (id / A.ktiles) * 2;
        t.s[0].tx = pt % tiles_x; t.s[0].ty = (pt / tiles_x) % tiles_y; t.s[0].n = pt / (tiles_x * tiles_y);
        t.s[1] = inc(t.s[0], 1, 0, 0);
        t.k0 = t.kt * BN;
        return t;
    };
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
    auto w_rsrc = [&](const TileRef& t) {   // U[K][4][3][C]: 12 C elements per output channel
        const unsigned char* wbase = (const unsigned char*)A.u + (long long)t.k0 * 12 * C * 2;
        return __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7ffffffe, 0x00020000);
    };

    // ---- per-lane DMA offsets: one instruction = 16 LDS rows of 64 B, lane -> (row lane >> 2, 16-byte slot lane & 3)
    const int lrow = lane >> 2, lslot = lane & 3;
    int pvoff[PSL];   // patches: instruction ii = j*8 + wave -> patch ii / 21, its pixels (ii % 21) * 16 .. + 15
#pragma unroll
    for (int j = 0; j < PSL; ++j) {
        const int ii = j * NWAVE + wave;
        int pi = (ii % SUBI) * 16 + lrow;
        if (pi > PPIX - 1) pi = PPIX - 1;
        const int py = pi / PW, px = pi - py * PW;
        pvoff[j] = (py * A.Wib + px) * C * 2 + ((lslot ^ ((px >> 1) & 3)) << 4);
    }
    int wvoff[NWI];            // weights: instruction ii = j*8 + wave covers LDS rows ii*16 .. + 15; row = p * 64 + k
#pragma unroll
    for (int j = 0; j < NWI; ++j) {
        const int row = (j * NWAVE + wave) * 16 + lrow;
        const int p = row >> 6, k = row & 63;
        wvoff[j] = (k * 12 + p * 3) * C * 2 + ((lslot ^ ((row >> 2) & 3)) << 4);
    }
    auto issue_w = [&](const __amdgpu_buffer_rsrc_t& wr, int buf, int s, int c0, bool real) {
#pragma unroll
        for (int j = 0; j < NWI; ++j) {
            unsigned char* dst = real ? smem + OFF_W + buf * WBYTES + (j * NWAVE + wave) * 1024 : smem + OFF_DUMMY;
            dma16(wr, dst, wvoff[j], (s * C + c0) * 2);
        }
    };
    auto issue_p = [&](const __amdgpu_buffer_rsrc_t& x0, const __amdgpu_buffer_rsrc_t& x1, int pbuf, int j, int c0, bool real) {
        const int ii = j * NWAVE + wave;
        unsigned char* dst = (real && ii < PINSTR) ? smem + pbuf * PBYTES + ii * 1024 : smem + OFF_DUMMY;
        dma16(ii < SUBI ? x0 : x1, dst, pvoff[j], c0 * 2);
    };

    // ---- fragment read offsets
    const int fr = lane & 31, fh = lane >> 5;
    const int ftx = fr & 15;
    // input row 2 t of this lane's row pair t = 2 wm + (fr >> 4), column ftx, in the wave's patch
    const int xoff = sub * SUBB + ((4 * wm + 2 * (fr >> 4)) * PW + ftx) * ROWB;
    int xs16[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) xs16[s] = (fh ^ (((ftx + s) >> 1) & 3)) << 4;
    int woff[2];     // row a*32 + fr of product p = 0 (p adds 64 rows), swizzled slot of kk = 0
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int row = a * 32 + fr;
        woff[a] = row * ROWB + ((fh ^ ((row >> 2) & 3)) << 4);
    }

    struct XF { h16x8 v[4]; };   // d_0..d_3 when read, V_0..V_3 after the transform
    struct WF { h16x8 u[4]; };   // U_p fragments of one 32-channel block
    auto load_d = [&](XF& f, const unsigned char* xb, int s, int kk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) f.v[j] = *(const h16x8*)(xb + xoff + j * (PW * ROWB) + (xs16[s] ^ (kk << 5)));
    };
    auto transform = [&](XF& f) {
        const h16x8 d0 = f.v[0], d1 = f.v[1], d2 = f.v[2], d3 = f.v[3];
        f.v[0] = hsub8(d0, d2); f.v[1] = hadd8(d1, d2); f.v[2] = hsub8(d2, d1); f.v[3] = hsub8(d1, d3);
    };
    auto load_w = [&](WF& f, const unsigned char* wb, int a, int kk) {
#pragma unroll
        for (int p = 0; p < 4; ++p) f.u[p] = *(const h16x8*)(wb + p * (BN * ROWB) + (woff[a] ^ (kk << 5)));
    };

    f32x16 acc[2][4];   // [channel block a][product p]
    // y[2t] = M0 + M1 + M2 and y[2t+1] = M1 - M2 - M3: M0 starts at the bias, M3 at minus the bias
    auto init_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bq = *reinterpret_cast<const float4*>(smem + OFF_BIAS + (wave * 64 + a * 32 + 8 * g + 4 * fh) * 4);
                acc[a][0][g * 4 + 0] = bq.x; acc[a][0][g * 4 + 1] = bq.y; acc[a][0][g * 4 + 2] = bq.z; acc[a][0][g * 4 + 3] = bq.w;
                acc[a][3][g * 4 + 0] = -bq.x; acc[a][3][g * 4 + 1] = -bq.y; acc[a][3][g * 4 + 2] = -bq.z; acc[a][3][g * 4 + 3] = -bq.w;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[a][1][g * 4 + e] = acc[a][2][g * 4 + e] = 0.f;
            }
    };
    auto mma = [&](const WF& w, const XF& x, int a) {
#pragma unroll
        for (int p = 0; p < 4; ++p) acc[a][p] = kp_mfma16(w.u[p], x.v[p], acc[a][p], 0, 0, 0);
    };

    // epilogue: output transform in registers, then the pair kernel's staging: lane: row pair / column = fr, channel
    // 8*(reg>>2) + 4*fh + (reg&3) of block a; block e = output row 2 t + e.
    constexpr int CPP = BN / 8, PPI = 64 / CPP;   // 8 chunks of 16 B per pixel, 8 pixels per store instruction
    float psum[STATS ? 2 : 1], psq[STATS ? 2 : 1];
    if constexpr (STATS) { psum[0] = psum[1] = psq[0] = psq[1] = 0.f; }
    const int st_voff = ((lane / CPP) * K + (lane % CPP) * 8) * 2;
    auto epilogue = [&](const TileRef& t, int free_pb) {
        unsigned char* stg = smem + free_pb * PBYTES + wave * (32 * BN * 2);
        h16_t* ybase = A.y + ((((long long)t.s[sub].n * A.Hob + t.s[sub].ty * TS + wm * 4 + A.ooff) * A.Wob) + t.s[sub].tx * TS + A.ooff) * (long long)K + t.k0;
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)ybase, 0, 0x7ffffffe, 0x00020000);
        // y0 -> acc[a][0], y1 -> acc[a][3] (in place)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float m1 = acc[a][1][i], m2 = acc[a][2][i];
                acc[a][0][i] = acc[a][0][i] + (m1 + m2);
                acc[a][3][i] = (m1 - m2) - acc[a][3][i];
            }
        if constexpr (STATS) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                float s16[16];
                int r1;
#pragma unroll
                for (int i = 0; i < 16; ++i) s16[i] = acc[a][0][i] + acc[a][3][i];
                psum[a] += reduce16_over32(s16, lane, r1);
#pragma unroll
                for (int i = 0; i < 16; ++i) s16[i] = __builtin_fmaf(acc[a][0][i], acc[a][0][i], acc[a][3][i] * acc[a][3][i]);
                psq[a] += reduce16_over32(s16, lane, r1);
            }
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    uint2 pk;
                    pk.x = pack_h16x2(acc[a][3 * e][g * 4 + 0], acc[a][3 * e][g * 4 + 1]);
                    pk.y = pack_h16x2(acc[a][3 * e][g * 4 + 2], acc[a][3 * e][g * 4 + 3]);
                    const int chunk = a * 4 + g;
                    *reinterpret_cast<uint2*>(stg + fr * (BN * 2) + ((chunk ^ (fr & (CPP - 1))) << 4) + fh * 8) = pk;
                }
            }
#pragma unroll
            for (int i = 0; i < 32 * CPP / 64; ++i) {
                const int id = i * 64 + lane, px = id / CPP, c16 = id % CPP;
                const u32x4 vv = *reinterpret_cast<const u32x4*>(stg + px * (BN * 2) + ((c16 ^ (px & (CPP - 1))) << 4));
                // pixels i*8 .. i*8+7 of the block: row pair (i*8) >> 4 -> output row 2 * pair + e, column (i*8) & 15
                const int soff = ((2 * ((i * PPI) >> 4) + e) * A.Wob + ((i * PPI) & 15)) * K * 2;
                __builtin_amdgcn_raw_buffer_store_b128(vv, yr, st_voff, soff, 0);
            }
        }
    };

    const int nchunks = C / CH;
    TileRef cur = decode(tile);
    __amdgpu_buffer_rsrc_t xr0 = x_rsrc(cur.s[0]), xr1 = x_rsrc(cur.s[1]), wr = w_rsrc(cur);
    auto load_bias = [&](const TileRef& t) {
        reinterpret_cast<float*>(smem + OFF_BIAS)[wave * 64 + lane] = A.bias ? A.bias[t.k0 + lane] : 0.f;
    };
    load_bias(cur);

    // ---- prologue: patches of chunk 0, weight tiles of steps 0..2 (= the three filter columns of chunk 0)
#pragma unroll
    for (int j = 0; j < PSL; ++j) issue_p(xr0, xr1, 0, j, 0, true);
    issue_w(wr, 0, 0, 0, true);
    issue_w(wr, 1, 1, 0, true);
    issue_w(wr, 2, 2, 0, true);
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    XF xA, xB;
    WF wC, wN;
    load_d(xA, smem, 0, 0);
    load_w(wC, smem + OFF_W, 0, 0);
    transform(xA);
    init_acc();
    int pbi = 0, wslot = 0;
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
            const bool more_p = !last_ch || has_next;
            static_for<0, 3>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int ns = (s + 1) % 3;
                const unsigned char* xb = pb + s * ROWB;
                const unsigned char* wb = smem + OFF_W + wslot * WBYTES;
                const unsigned char* nxb = (s == 2 ? npb : pb) + ns * ROWB;
                const unsigned char* nwb = smem + OFF_W + ((wslot + 1) & (NWBUF - 1)) * WBYTES;

                // -- quarter 0: slice 0, channel block 0; the reads of block 1's weights under it
                load_w(wN, wb, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma(wC, xA, 0);
                __builtin_amdgcn_sched_barrier(0);
                // -- quarter 1: slice 0, block 1; reads of slice 1 (input rows, block 0's weights); V of slice 1
                load_d(xB, xb, s, 1);
                load_w(wC, wb, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(wN, xA, 1);
                __builtin_amdgcn_sched_barrier(0);
                transform(xB);
                __builtin_amdgcn_sched_barrier(0);
                if (s == 2 && last_ch) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                // in flight at most what the previous step issued: 2 weight instructions, + the 6 patch instructions of a chunk's first step
                if (!(s == 0 && after_epi)) wait_vm<NWI + (s == 1 ? PSL : 0)>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // -- quarter 2: slice 1, block 0; DMA of step t+3 and of the next patches; reads of block 1's weights
                load_w(wN, wb, 1, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(wC, xB, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s == 0) {       // the next chunk's patches (ahead of the weights: they are retired first)
#pragma unroll
                    for (int j = 0; j < PSL; ++j)
                        issue_p(last_ch ? xrn0 : xr0, last_ch ? xrn1 : xr1, pbi ^ 1, j, last_ch ? 0 : (ch + 1) * CH, more_p);
                }
                if (!(s == 2 && last_ch)) {   // step t+3 = the same filter column of the next chunk (the last step of a tile defers it)
                    const bool to_next = last_ch;
                    const bool real = !to_next || has_next;
                    issue_w(to_next ? wrn : wr, (wslot + 3) & (NWBUF - 1), s, to_next ? 0 : (ch + 1) * CH, real);
                }
                __builtin_amdgcn_sched_barrier(0);
                // -- quarter 3: slice 1, block 1; reads of the next step's slice 0
                load_d(xA, nxb, ns, 0);
                load_w(wC, nwb, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                mma(wN, xB, 1);
                __builtin_amdgcn_sched_barrier(0);
                transform(xA);
                __builtin_amdgcn_sched_barrier(0);
                wslot = (wslot + 1) & (NWBUF - 1);
            });
            pbi ^= 1;
            after_epi = false;
        }
        wait_vm<0>();
        epilogue(cur, pbi ^ 1);
        if (!has_next) break;
        if (nxt.k0 != cur.k0) load_bias(nxt);
        issue_w(wrn, (wslot + 2) & (NWBUF - 1), 2, 0, true);
        after_epi = true;
        init_acc();
        tile = ntile_id; cur = nxt; xr0 = xrn0; xr1 = xrn1; wr = wrn;
    }
    if constexpr (STATS) {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if ((lane & 1) == 0) {
                const int rg = ((lane & 16) ? 8 : 0) | ((lane & 8) ? 4 : 0) | ((lane & 4) ? 2 : 0) | ((lane & 2) ? 1 : 0);
                const int k = cur.k0 + a * 32 + 8 * (rg >> 2) + 4 * fh + (rg & 3);
                const long long prow = (long long)blockIdx.x * NWAVE + wave;
                A.partials[(prow * 2 + 0) * K + k] = psum[a];
                A.partials[(prow * 2 + 1) * K + k] = psq[a];
            }
        }
    }
}

// U[K][4][3][C] (16-bit) from the packed direct-kernel weights wf[K][3 r][3 s][C] (16-bit): U_p[s] = sum_r G[p][r] wf[r][s],
// G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], formed in fp32 from the 16-bit values and rounded once.
__global__ void wino_pack_k(const h16_t* __restrict__ wf, h16_t* __restrict__ u, long long n, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over K * 3 * C
    if (i >= n) return;
    const int c = (int)(i % C);
    const int s = (int)((i / C) % 3);
    const long long k = i / (3LL * C);
    const float g0 = h2f(wf[((k * 3 + 0) * 3 + s) * C + c]), g1 = h2f(wf[((k * 3 + 1) * 3 + s) * C + c]), g2 = h2f(wf[((k * 3 + 2) * 3 + s) * C + c]);
    h16_t* o = u + (k * 12 + s) * C + c;
    o[0] = f2h(g0);
    o[3LL * C] = f2h(0.5f * (g0 + g1 + g2));
    o[6LL * C] = f2h(0.5f * (g0 - g1 + g2));
    o[9LL * C] = f2h(g2);
}

}  // namespace

static bool wino_plan(const float* stats, int N, int Hib, int Wib, int H, int W, int C, int K, int& ktiles, int& ntiles, int& grid) {
    if (C % CH != 0 || C < 64 || K % BN != 0 || H % TS != 0 || W % TS != 0 || H <= 0 || W <= 0) return false;
    if ((long long)(PW + 1) * Wib * C * 2 >= 0x7fffffffLL || 64LL * 12 * C * 2 + 12LL * C * 2 >= 0x7fffffffLL) return false;
    const long long ptiles = (long long)N * (H / TS) * (W / TS);
    if (ptiles % 2 != 0) return false;
    ktiles = K / BN;
    if ((ptiles / 2) * ktiles > 0x7fffffffLL) return false;
    ntiles = (int)((ptiles / 2) * ktiles);
    int cus = kp_device_cus();
    if (cus <= 0) cus = 256;
    grid = (cus / 8) * 8;
    if (grid < 8) grid = 8;
    const int need = ((ntiles + 7) / 8) * 8;
    if (grid > need) grid = need;
    if (stats && ntiles > grid && !(((grid / 8) % ktiles) == 0 && (grid % ktiles) == 0)) return false;
    return true;
}

// Winograd weights of a layer from its packed 16-bit weights ([K][3][3][C], the forward or the flipped data-gradient pack)
int kp_conv_wino_pack_impl(const void* wf, void* u, int K, int C, hipStream_t st) {
    const long long n = (long long)K * 3 * C;
    wino_pack_k<<<cdiv(n, 256), 256, 0, st>>>((const h16_t*)wf, (h16_t*)u, n, C);
    return kp_check_launch("wino_pack");
}

// Same contract as kp_conv_patch_launch with `u` = the Winograd weights (kp_conv_wino_pack) in place of the packed weights.
int kp_conv_wino_launch(const void* x, const void* u, const float* bias, void* y, float* stats, int N, int Hib,
                        int Wib, int ioff, int Hob, int Wob, int ooff, int H, int W, int C, int K, hipStream_t st) {
    WinoArgs a;
    int grid = 0;
    if (!wino_plan(stats, N, Hib, Wib, H, W, C, K, a.ktiles, a.ntiles, grid)) return KP_E_UNSUPPORTED;
    a.x = (const h16_t*)x; a.u = (const h16_t*)u; a.bias = bias; a.y = (h16_t*)y; a.partials = stats;
    a.Hib = Hib; a.Wib = Wib; a.ioff = ioff; a.Hob = Hob; a.Wob = Wob; a.ooff = ooff; a.H = H; a.W = W;
    a.C = C; a.K = K;
    { extern kp_knob_t g_kp_walk; a.linear = (g_kp_walk & KP_WALK_CONV_LINEAR) ? 1 : 0; }
    if (stats) kp_set_last_rows((long long)grid * NWAVE);
    KP_QUERY_RETURN();
    if (stats) conv3x3_wino_k<true><<<grid, TPB, 0, st>>>(a);
    else conv3x3_wino_k<false><<<grid, TPB, 0, st>>>(a);
    return kp_check_launch("conv3x3_wino");
}

#else   // bf16 build: no Winograd kernel (no packed bf16 adds, and bf16 storage has no parity headroom to spend)

int kp_conv_wino_pack_impl(const void*, void*, int, int, hipStream_t) { return KP_E_UNSUPPORTED; }
int kp_conv_wino_launch(const void*, const void*, const float*, void*, float*, int, int, int, int, int, int, int, int, int, int, int,
                        hipStream_t) { return KP_E_UNSUPPORTED; }

#endif
