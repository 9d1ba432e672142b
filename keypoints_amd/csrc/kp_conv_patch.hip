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
//           the weight tile W[k0..k0+BN)[tap][c0..c0+64) streams through a 3-deep LDS ring (16 KB / step).
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
#include "kp_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_AS __attribute__((address_space(3)))

constexpr int TPB = 512, NWAVE = 8;
constexpr int TS = 16;                  // tile side (pixels)
constexpr int PW = TS + 2;              // patch side
constexpr int PPIX = PW * PW;           // 324 patch pixels
constexpr int PINSTR = (PPIX + 7) / 8;  // 41 DMA instructions of 8 pixels (1 KiB)
constexpr int PBYTES = PINSTR * 1024;
constexpr int PSLICES = (PINSTR + NWAVE - 1) / NWAVE;   // patch DMA instructions per wave per chunk (6)
constexpr int ROWB = 128;               // bytes per LDS row (64 bf16)
constexpr int NWBUF = 3;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_n(int n) {   // n is wave-uniform
    switch (n) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<1>(); break;
        case 2: wait_vm<2>(); break;
        default: wait_vm<3>(); break;
    }
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_base, int voff, int soff) {
    // 16 B per lane; LDS destination = wave-uniform base + lane*16
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// halving butterfly: 16 per-lane values -> each lane ends with the sum over its 32-lane half of ONE value.
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

struct PatchArgs {
    const bf16_t* x; const bf16_t* w; const float* bias; bf16_t* y; float* partials;
    int Hib, Wib, ioff;   // input buffer rows / cols, offset of the tile domain's patch origin in it
    int Hob, Wob, ooff;   // output buffer rows / cols, offset of the tile domain in it
    int H, W;             // tile domain (multiples of 16)
    int C, K, ktiles;
};

template <int BN>
__global__ __launch_bounds__(TPB, 2) void conv3x3_patch_k(const PatchArgs A) {
    constexpr int WBYTES = BN * ROWB;
    constexpr int NWI = BN / 64;          // weight DMA instructions per wave per step
    constexpr int WN = BN / 2;            // channels per wave
    constexpr int TN = WN / 32;           // 32-channel MFMA tiles per wave
    constexpr int OFF_W = 2 * PBYTES, OFF_DUMMY = OFF_W + NWBUF * WBYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[OFF_DUMMY + 1024];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int C = A.C, K = A.K;

    // XCD-aware tile order: consecutive tile ids (k-tile fastest) stay on one XCD so its L2 keeps the patch
    int tile;
    {
        const int nwg = gridDim.x, b = blockIdx.x, q = nwg / 8, r = nwg % 8, xcd = b % 8, i = b / 8;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int kt = tile % A.ktiles, pt = tile / A.ktiles;
    const int tiles_x = A.W / TS, tiles_y = A.H / TS;
    const int txi = pt % tiles_x, tyi = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
    const int oy0 = tyi * TS, ox0 = txi * TS, k0 = kt * BN;

    // ---- DMA descriptors and per-lane offsets -----------------------------------------------------------
    const unsigned char* xbase = (const unsigned char*)A.x +
        ((((long long)n * A.Hib + oy0 + A.ioff) * A.Wib) + ox0 + A.ioff) * (long long)C * 2;
    const unsigned char* wbase = (const unsigned char*)A.w + (long long)k0 * 9 * C * 2;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)xbase, 0, 0x7ffffffe, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wbase, 0, 0x7ffffffe, 0x00020000);

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

    auto issue_w = [&](int buf, int tap, int c0) {
        unsigned char* dst = smem + OFF_W + buf * WBYTES;
        const int soff = (tap * C + c0) * 2;
#pragma unroll
        for (int j = 0; j < NWI; ++j) dma16(wr, dst + (j * NWAVE + wave) * 1024, wvoff[j], soff);
    };
    auto issue_p = [&](int pbuf, int j, int c0) {   // j compile-time after unrolling
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

    f32x16 acc[TN][2];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;

    const int nchunks = C / 64, T = nchunks * 9;

    // ---- prologue: patch of chunk 0, weight tiles of steps 0 and 1 ------------------------------------------
#pragma unroll
    for (int j = 0; j < PSLICES; ++j) issue_p(0, j, 0);
    issue_w(0, 0, 0);
    issue_w(1, 1, 0);   // T >= 9
    wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    for (int ch = 0; ch < nchunks; ++ch) {
        const unsigned char* pb = smem + (ch & 1) * PBYTES;
        const int npb = (ch + 1) & 1;
        const bool more_p = ch + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int t = ch * 9 + tap;
            // -- issue: weight tile of step t+2, one slice of the next chunk's patch
            int issued = 0;
            if (t + 2 < T) {
                const int tap2 = (tap + 2) % 9;
                const int c2 = (tap + 2 >= 9 ? ch + 1 : ch) * 64;
                issue_w((tap + 2) % NWBUF, tap2, c2);
                issued += NWI;
            }
            if (tap < PSLICES && more_p) {
                issue_p(npb, tap, (ch + 1) * 64);
                issued += 1;
            }
            // -- compute step t from patch pb (shift r,s) and weight buffer tap % 3
            const int r = tap / 3, s = tap % 3;
            const unsigned char* xb = pb + (r * PW + s) * ROWB;
            const unsigned char* wb = smem + OFF_W + (tap % NWBUF) * WBYTES;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                bf16x8 xf[2], wf[TN];
#pragma unroll
                for (int b = 0; b < 2; ++b) xf[b] = *(const bf16x8*)(xb + xoff[b] + (xs16[s] ^ (kk << 5)));
#pragma unroll
                for (int a = 0; a < TN; ++a) wf[a] = *(const bf16x8*)(wb + woff[a] + (ws16[a] ^ (kk << 5)));
#pragma unroll
                for (int a = 0; a < TN; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[a], xf[b], acc[a][b], 0, 0, 0);
            }
            // -- everything older than this step's DMA has landed (weights of step t+1, earlier patch slices)
            wait_vm_n(issued);
            __builtin_amdgcn_s_barrier();
        }
    }

    // ---- epilogue: D[k_out][pixel]; lane: pixel = fr of fragment b, k_out = 8*(reg>>2) + 4*fh + (reg&3)
    const long long prow = (long long)pt * 4 + wm;   // statistics row of this wave (64 pixels)
#pragma unroll
    for (int a = 0; a < TN; ++a) {
        const int kb = k0 + wn * WN + a * 32;
        float sm[16], sq[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) sm[i] = sq[i] = 0.f;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int oy = oy0 + wm * 4 + b * 2 + (fr >> 4) + A.ooff, ox = ox0 + ftx + A.ooff;
            bf16_t* yp = A.y + (((long long)n * A.Hob + oy) * A.Wob + ox) * K;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kq = kb + 8 * g + 4 * fh;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[a][b][g * 4 + e] + (A.bias ? A.bias[kq + e] : 0.f);
                    sm[g * 4 + e] += v[e];
                    sq[g * 4 + e] += v[e] * v[e];
                }
                uint2 pk;
                pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
                pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
                *reinterpret_cast<uint2*>(yp + kq) = pk;
            }
        }
        if (A.partials) {
            int r1, r2;
            const float ts = reduce16_over32(sm, lane, r1);
            const float tq = reduce16_over32(sq, lane, r2);
            if ((lane & 1) == 0) {
                const int k = kb + 8 * (r1 >> 2) + 4 * fh + (r1 & 3);
                A.partials[(prow * 2 + 0) * K + k] = ts;
                A.partials[(prow * 2 + 1) * K + k] = tq;
            }
        }
    }
}

}  // namespace

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
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y; a.partials = stats;
    a.Hib = Hib; a.Wib = Wib; a.ioff = ioff; a.Hob = Hob; a.Wob = Wob; a.ooff = ooff; a.H = H; a.W = W;
    a.C = C; a.K = K;
    const long long ptiles = (long long)N * (H / TS) * (W / TS);
    if (K % 128 == 0) {
        a.ktiles = K / 128;
        if (ptiles * a.ktiles > 0x7fffffffLL) return KP_E_UNSUPPORTED;
        conv3x3_patch_k<128><<<(unsigned)(ptiles * a.ktiles), TPB, 0, st>>>(a);
    } else {
        a.ktiles = K / 64;
        if (ptiles * a.ktiles > 0x7fffffffLL) return KP_E_UNSUPPORTED;
        conv3x3_patch_k<64><<<(unsigned)(ptiles * a.ktiles), TPB, 0, st>>>(a);
    }
    return kp_check_launch("conv3x3_patch");
}
