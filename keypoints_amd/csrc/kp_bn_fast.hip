// Fast bf16 paths of the BatchNorm(train) + activation passes (kp_bn.hip holds the general kernels and the
// entry points; this file is their HBM-roofline version for the shapes the networks actually run).
//
// Same mapping idea as kp_bn.hip -- a thread owns ONE 16-byte channel vector (8 bf16) and walks pixels, so the
// per-channel constants live in registers -- but shaped for bandwidth:
//   * a workgroup takes several image rows (>= 16 vectors per thread), not one;
//   * the pixel loop is unrolled by 4 with all loads issued before the first use (4-8 x 16 B in flight per lane);
//   * tail / border-width are template parameters; the replicate-pad adjoint ("fold") of an incoming gradient
//     is one unconditional centre load plus a rare edge path;
//   * the backward reduction leaves ONE partial row per workgroup (kp_last_rows() = number of workgroups),
//     which also shortens the fp64 column reduction that follows.
// Arithmetic is expression-for-expression the same as the general kernels (only the summation order of the
// fold at image edges differs).
#include <stdlib.h>
#include "kp_common.h"

namespace {

#ifndef KP_UNR
#define KP_UNR 4   // vectors in flight per thread in the forward no-tail pass (build-time knob for tools/probes)
#endif
constexpr int TPB = 256, UNR = KP_UNR;
constexpr int UNB = 4;   // pixels in flight per thread in the backward kernels (2 and 1 measured slower)
constexpr int UNB4 = 4;  // the same for the 4-channel variants

struct BnP {
    const float *mean, *invstd, *gamma, *beta;
};

// ---- register images of V consecutive 16-bit channels: V = 8 -> uint4, V = 4 -> uint2 ------------------------------------
__device__ __forceinline__ void unpackv(const uint4 v, float (&o)[8]) {
    unpack_h16x2(v.x, o[0], o[1]);
    unpack_h16x2(v.y, o[2], o[3]);
    unpack_h16x2(v.z, o[4], o[5]);
    unpack_h16x2(v.w, o[6], o[7]);
}
__device__ __forceinline__ void unpackv(const uint2 v, float (&o)[4]) {
    unpack_h16x2(v.x, o[0], o[1]);
    unpack_h16x2(v.y, o[2], o[3]);
}
__device__ __forceinline__ uint4 packv(const float (&o)[8]) {
    uint4 v;
    v.x = pack_h16x2(o[0], o[1]); v.y = pack_h16x2(o[2], o[3]);
    v.z = pack_h16x2(o[4], o[5]); v.w = pack_h16x2(o[6], o[7]);
    return v;
}
__device__ __forceinline__ uint2 packv(const float (&o)[4]) {
    uint2 v;
    v.x = pack_h16x2(o[0], o[1]); v.y = pack_h16x2(o[2], o[3]);
    return v;
}
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
// nt: streaming load -- the line is not kept in the caches behind it (last use of the data for a long time)
template <typename R> __device__ __forceinline__ R ldraw(const h16_t* p, int nt);
template <> __device__ __forceinline__ uint4 ldraw<uint4>(const h16_t* p, int nt) {
    if (!nt) return *reinterpret_cast<const uint4*>(p);
    const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <> __device__ __forceinline__ uint2 ldraw<uint2>(const h16_t* p, int nt) {
    if (!nt) return *reinterpret_cast<const uint2*>(p);
    const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
    return make_uint2(v.x, v.y);
}
__device__ __forceinline__ void straw(h16_t* p, const uint4 v) { *reinterpret_cast<uint4*>(p) = v; }
__device__ __forceinline__ void straw(h16_t* p, const uint2 v) { *reinterpret_cast<uint2*>(p) = v; }
template <typename R> __device__ __forceinline__ R raw_zero();
template <> __device__ __forceinline__ uint4 raw_zero<uint4>() { return make_uint4(0, 0, 0, 0); }
template <> __device__ __forceinline__ uint2 raw_zero<uint2>() { return make_uint2(0, 0); }
// V one-byte argmax codes (window position 0..3 of MaxPool2d(2,2))
__device__ __forceinline__ unsigned am_code(const uint2 a, int e) { return ((e < 4 ? a.x : a.y) >> (8 * (e & 3))) & 0xffu; }
__device__ __forceinline__ unsigned am_code(const unsigned a, int e) { return (a >> (8 * e)) & 0xffu; }
__device__ __forceinline__ uint2 am_pack(const int (&b)[8]) {
    uint2 pk;
    pk.x = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    pk.y = b[4] | (b[5] << 8) | (b[6] << 16) | (b[7] << 24);
    return pk;
}
__device__ __forceinline__ unsigned am_pack(const int (&b)[4]) { return b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24); }

namespace v8 {
constexpr int V = 8;
typedef uint4 raw_t;
typedef uint2 am_t;
#include "kp_bn_fast_body.inc"
}  // namespace v8
namespace v4 {
constexpr int V = 4;
typedef uint2 raw_t;
typedef unsigned am_t;
#include "kp_bn_fast_body.inc"
}  // namespace v4
using v8::clampi;
using v8::bilin;
constexpr int V = 8;   // (the bilinear adjoint below works on 16-byte lanes)
__device__ __forceinline__ void unpack8(const uint4 v, float (&o)[8]) { unpackv(v, o); }
__device__ __forceinline__ uint4 pack8(const float (&o)[8]) { return packv(o); }
__device__ __forceinline__ uint4 ldv(const h16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ void stv(h16_t* p, const uint4 v) { *reinterpret_cast<uint4*>(p) = v; }

bool fast_shape(int C, int v = 8) { return C % v == 0 && (C / v) <= TPB && TPB % (C / v) == 0; }

int rows_per_block(long long total_rows, int vec_per_row, int cus, int max_blocks = 0) {
    // >= 16 vectors per thread, but keep >= ~4 workgroups per CU
    int rpb = (16 * TPB + vec_per_row - 1) / vec_per_row;
    if (rpb < 1) rpb = 1;
    const long long max_rpb = total_rows / ((long long)cus * 4);
    if (rpb > max_rpb) rpb = (int)(max_rpb > 1 ? max_rpb : 1);
    // reduce passes: every workgroup leaves one partial row for the finalize launch, which is a handful of workgroups streaming
    // ALL of them (8 192 rows of a 128-channel layer: 35 us for 8 MB) -- cap the grid (KP_BN_REDUCE_BLOCKS rows at most)
    if (max_blocks > 0 && (total_rows + rpb - 1) / rpb > max_blocks) rpb = (int)((total_rows + max_blocks - 1) / max_blocks);
    return rpb;
}
#ifndef KP_BN_REDUCE_BLOCKS
#define KP_BN_REDUCE_BLOCKS 1024   // (2048 -> 1024: the reduce passes themselves 5-17 % faster at model F shapes, tools/probes/bn_ab3.sh)
#endif
// (grid caps of the three pass classes: compile-time constants; -DKP_BN_REDUCE_BLOCKS / -DKP_BN_APPLY_BLOCKS for sweeps)
static int bn_blocks_cap(int, int dflt) { return dflt; }
#ifndef KP_BN_APPLY_BLOCKS
#define KP_BN_APPLY_BLOCKS 0   // (apply / forward passes: no finalize behind them; A/B knob)
#endif


// ---- adjoint of the bilinear x2 tail (align_corners=True) with the replicate-pad border folded in ---------------
// Separable: workgroup = (image, coarse row h, 64-channel chunk).  Phase 1 folds the <= 5 fine rows that touch h
// (weights Ah) and the padded border into T[fine column][64 ch] (fp32, LDS), every fine pixel read as a coalesced
// 16-byte vector; phase 2 applies the column weights Aw from LDS.  The general kernel gathers up to 25 fine pixels
// per output vector straight from memory with the index logic in the inner loop.
__global__ __launch_bounds__(256) void upsample2_bwd_fast_k(const h16_t* __restrict__ gsrc, int gp, h16_t* __restrict__ g,
                                                            int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) float T[];   // [Wo][64]
    const int Ho = 2 * H, Wo = 2 * W, Hp = Ho + 2 * gp, Wp = Wo + 2 * gp, chunks = C / 64;
    // XCD x (workgroups with blockIdx % 8 == x) walks a contiguous range of (image, row, chunk) items, so that the
    // workgroups of neighbouring coarse rows, which share 3 of their 5 fine rows, meet in one L2
    int b;
    {
        const int nwg = gridDim.x, q = nwg / 8, r = nwg % 8, xcd = blockIdx.x % 8, i = blockIdx.x / 8;
        b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
    }
    const int ch = b % chunks; b /= chunks;
    const int h = b % H;
    const long long n = b / H;
    const int cv = threadIdx.x & 7, pl = threadIdx.x >> 3;
    const int c = ch * 64 + cv * 8;
    const int i_lo = 2 * h - 2 < 0 ? 0 : 2 * h - 2, i_hi = 2 * h + 2 > Ho - 1 ? Ho - 1 : 2 * h + 2;
    // rows away from the top / bottom border: exactly the 5 fine rows 2h-2 .. 2h+2, no vertical fold -> their loads
    // are issued together (the general loop below waits for every load before it decides on the next one)
    const bool vint = i_lo > 0 && i_hi < Ho - 1 && i_hi - i_lo == 4;
    float rw[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) {
        int h0, hp; float lh0, lh1;
        bilin(i_lo + r, H, Ho, h0, hp, lh0, lh1);
        rw[r] = (h0 == h ? lh0 : 0.f) + (h0 + hp == h ? lh1 : 0.f);
    }
    if (vint) {
        const h16_t* rowp = gsrc + ((n * Hp + i_lo + gp) * Wp) * C + c;
        for (int j = pl; j < Wo; j += 32) {
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = 0.f;
            const int jlo = (j == 0) ? 0 : j + gp, jhi = (j == Wo - 1) ? Wo + 2 * gp - 1 : j + gp;
            for (int bb = jlo; bb <= jhi; ++bb) {
                uint4 raw[5];
#pragma unroll
                for (int r = 0; r < 5; ++r) raw[r] = ldv(rowp + ((long long)r * Wp + bb) * C);
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    float v[V];
                    unpack8(raw[r], v);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] = fmaf(rw[r], v[e], acc[e]);
                }
            }
            float4* t = reinterpret_cast<float4*>(T + j * 64 + cv * 8);
            t[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
            t[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
        }
    } else
    for (int j = pl; j < Wo; j += 32) {
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        const int jlo = (j == 0) ? 0 : j + gp, jhi = (j == Wo - 1) ? Wo + 2 * gp - 1 : j + gp;
        for (int i = i_lo; i <= i_hi; ++i) {
            int h0, hp; float lh0, lh1;
            bilin(i, H, Ho, h0, hp, lh0, lh1);
            const float wh = (h0 == h ? lh0 : 0.f) + (h0 + hp == h ? lh1 : 0.f);
            if (wh == 0.f) continue;
            const int ilo = (i == 0) ? 0 : i + gp, ihi = (i == Ho - 1) ? Ho + 2 * gp - 1 : i + gp;
            for (int a = ilo; a <= ihi; ++a)
                for (int bb = jlo; bb <= jhi; ++bb) {
                    float v[V];
                    unpack8(ldv(gsrc + ((n * Hp + a) * Wp + bb) * C + c), v);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] = fmaf(wh, v[e], acc[e]);
                }
        }
        float4* t = reinterpret_cast<float4*>(T + j * 64 + cv * 8);
        t[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
        t[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    __syncthreads();
    for (int w = pl; w < W; w += 32) {
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        const int j_lo = 2 * w - 2 < 0 ? 0 : 2 * w - 2, j_hi = 2 * w + 2 > Wo - 1 ? Wo - 1 : 2 * w + 2;
        for (int j = j_lo; j <= j_hi; ++j) {
            int w0, wp; float lw0, lw1;
            bilin(j, W, Wo, w0, wp, lw0, lw1);
            const float ww = (w0 == w ? lw0 : 0.f) + (w0 + wp == w ? lw1 : 0.f);
            if (ww == 0.f) continue;
            const float4* t = reinterpret_cast<const float4*>(T + j * 64 + cv * 8);
            const float4 t0 = t[0], t1 = t[1];
            acc[0] = fmaf(ww, t0.x, acc[0]); acc[1] = fmaf(ww, t0.y, acc[1]); acc[2] = fmaf(ww, t0.z, acc[2]); acc[3] = fmaf(ww, t0.w, acc[3]);
            acc[4] = fmaf(ww, t1.x, acc[4]); acc[5] = fmaf(ww, t1.y, acc[5]); acc[6] = fmaf(ww, t1.z, acc[6]); acc[7] = fmaf(ww, t1.w, acc[7]);
        }
        stv(g + ((n * H + h) * W + w) * C + c, pack8(acc));
    }
}

}  // namespace

// Channels per thread of a pass (experiment mask, kp_config(KP_CFG_WALK): KP_WALK_V4_* bits select the 4-channel variants)
static int pick_v(int bit, int C) {
    extern kp_knob_t g_kp_walk;
    return ((g_kp_walk & bit) && fast_shape(C, 4)) ? 4 : 8;
}

int kp_bn_act_fwd_fast(const void* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                       float slope, int tail, void* out, int out_pad, unsigned char* argmax, int N, int H, int W, int C,
                       hipStream_t st) {
    if (!fast_shape(C)) return KP_E_UNSUPPORTED;
    if (tail == KP_TAIL_MAXPOOL2 && ((H | W) & 1)) return KP_E_UNSUPPORTED;
    extern kp_knob_t g_kp_walk;
    const int walk = ((g_kp_walk & KP_WALK_FWD_REVERSE) ? 1 : 0) | ((g_kp_walk & KP_WALK_STREAM_LOADS) ? 2 : 0);
    BnP bn{mean, invstd, gamma, beta};
    const int Ho = tail == KP_TAIL_MAXPOOL2 ? H / 2 : (tail == KP_TAIL_UPSAMPLE2 ? 2 * H : H);
    const int Wo = tail == KP_TAIL_MAXPOOL2 ? W / 2 : (tail == KP_TAIL_UPSAMPLE2 ? 2 * W : W);
    const long long rows = (long long)N * (Ho + 2 * out_pad);
    const int work = tail == KP_TAIL_NONE ? 1 : 3;   // 4 source vectors per output vector
    const int vv = pick_v(tail == KP_TAIL_NONE ? KP_WALK_V4_FWD_NONE : (tail == KP_TAIL_MAXPOOL2 ? KP_WALK_V4_FWD_POOL : KP_WALK_V4_FWD_UP), C);
    const int rpb = rows_per_block(rows, (Wo + 2 * out_pad) * (C / 8) * work, kp_device_cus(), bn_blocks_cap(0, 0));
    const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
    if (tail == KP_TAIL_NONE)
    {
#define KP_ARGS (const h16_t*)y, bn, slope, (h16_t*)out, out_pad, H, W, C, rows, rpb, walk
        if (vv == 4) v4::bn_act_fwd_none_fast_k<<<grid, TPB, 0, st>>>(KP_ARGS);
        else v8::bn_act_fwd_none_fast_k<<<grid, TPB, 0, st>>>(KP_ARGS);
#undef KP_ARGS
    }
    else if (tail == KP_TAIL_MAXPOOL2)
    {
#define KP_ARGS (const h16_t*)y, bn, slope, (h16_t*)out, out_pad, argmax, H, W, C, rows, rpb, walk
        if (vv == 4) v4::bn_act_fwd_tail_fast_k<1><<<grid, TPB, 0, st>>>(KP_ARGS);
        else v8::bn_act_fwd_tail_fast_k<1><<<grid, TPB, 0, st>>>(KP_ARGS);
#undef KP_ARGS
    }
    else {
        const int HB = (Ho + 2 * out_pad + 1) / 2, WB = (Wo + 2 * out_pad + 1) / 2;
        const long long brows = (long long)N * HB;
        const int brpb = rows_per_block(brows, WB * (C / 8) * 4, kp_device_cus(), bn_blocks_cap(0, 0));
        const unsigned bgrid = (unsigned)((brows + brpb - 1) / brpb);
#define KP_ARGS (const h16_t*)y, bn, slope, (h16_t*)out, out_pad, H, W, C, brows, brpb, walk
        if (vv == 4) v4::bn_act_fwd_up_fast_k<<<bgrid, TPB, 0, st>>>(KP_ARGS);
        else v8::bn_act_fwd_up_fast_k<<<bgrid, TPB, 0, st>>>(KP_ARGS);
#undef KP_ARGS
    }
    return kp_check_launch("bn_act_fwd_fast");
}

// mode 0: reduce (partials [N*H][2][C], fully written); mode 1: apply
int kp_bn_bwd_fast(int mode, const void* gsrc, int gp, int tail, const unsigned char* argmax, const void* y,
                   const float* mean, const float* invstd, const float* gamma, const float* beta, float slope,
                   float* partials, const float* sums, void* dy, int dp, int N, int H, int W, int C, hipStream_t st) {
    if (!(tail == KP_TAIL_NONE || tail == KP_TAIL_MAXPOOL2) || gp < 0 || gp > 1 || !fast_shape(C)) return KP_E_UNSUPPORTED;
    if (tail == KP_TAIL_MAXPOOL2 && ((H | W) & 1)) return KP_E_UNSUPPORTED;
    BnP bn{mean, invstd, gamma, beta};
    extern kp_knob_t g_kp_walk;
    const int walk = ((g_kp_walk & (mode ? KP_WALK_APPLY_REVERSE : KP_WALK_REDUCE_REVERSE)) ? 1 : 0) | ((g_kp_walk & KP_WALK_STREAM_LOADS) ? 2 : 0);
    const int pool = tail == KP_TAIL_MAXPOOL2;
    if (pool) {   // walk pooled rows
        const int vv = pick_v(mode ? KP_WALK_V4_POOL_APPLY : KP_WALK_V4_POOL_REDUCE, C);
        const long long prows = (long long)N * (H / 2);
        // (apply over pooled rows: fewer, longer workgroups measured faster at every model F shape -- 277 -> 264, 162 -> 143, 90 -> 77 us)
        const int prpb = rows_per_block(prows, (W / 2) * (C / 8) * 4, kp_device_cus(), mode == 0 ? bn_blocks_cap(1, KP_BN_REDUCE_BLOCKS) : bn_blocks_cap(2, 512));
        const unsigned pgrid = (unsigned)((prows + prpb - 1) / prpb);
        if (mode == 0) kp_set_last_rows(pgrid);
        KP_QUERY_RETURN();
#define KP_ARGS (const h16_t*)gsrc, argmax, (const h16_t*)y, bn, slope, partials, sums, (h16_t*)dy, dp, N, H, W, C, prows, prpb, walk
#define KP_BWDP(GPv, MODEv)                                                                       \
    do {                                                                                          \
        if (vv == 4) v4::bn_bwd_pool_fast_k<GPv, MODEv><<<pgrid, TPB, 0, st>>>(KP_ARGS);          \
        else v8::bn_bwd_pool_fast_k<GPv, MODEv><<<pgrid, TPB, 0, st>>>(KP_ARGS);                  \
    } while (0)
        if (mode == 0) { if (gp) KP_BWDP(1, 0); else KP_BWDP(0, 0); }
        else { if (gp) KP_BWDP(1, 1); else KP_BWDP(0, 1); }
#undef KP_BWDP
#undef KP_ARGS
        return kp_check_launch("bn_bwd_pool_fast");
    }
    const int vv = pick_v(mode ? KP_WALK_V4_APPLY : KP_WALK_V4_REDUCE, C);
    extern kp_knob_t g_kp_bn_flat;
    // narrow maps with few channels (Pong-class nets): the row kernel's UNB x (TPB / CV) pixel slots are mostly empty -> the
    // kernel whose slots run over all rows of the workgroup (bn_bwd_flat_k).  Wide layers (model F) keep the row kernel.
    const int unb = vv == 4 ? UNB4 : UNB;
    const long long frows = (long long)N * H;
    const int frpb = rows_per_block(frows, W * (C / 8), kp_device_cus(), mode == 0 ? bn_blocks_cap(1, KP_BN_REDUCE_BLOCKS) : bn_blocks_cap(2, KP_BN_APPLY_BLOCKS));
    // (the flat kernel splits a pixel slot into row / column with a float multiply that is exact for slot indices below 2^20:
    //  a workgroup with more pixels than that -- a capped grid over a very large batch -- takes the row kernel)
    if (g_kp_bn_flat && W < unb * (TPB / (C / vv)) && H >= 2 && (long long)frpb * W < (1LL << 20)) {
        const unsigned fgrid = (unsigned)((frows + frpb - 1) / frpb);
        if (mode == 0) kp_set_last_rows(fgrid);
        KP_QUERY_RETURN();
#define KP_ARGS (const h16_t*)gsrc, (const h16_t*)y, bn, slope, partials, sums, (h16_t*)dy, dp, N, H, W, C, frows, frpb, walk
#define KP_BWDF(GPv, MODEv)                                                                       \
    do {                                                                                          \
        if (vv == 4) v4::bn_bwd_flat_k<GPv, MODEv><<<fgrid, TPB, 0, st>>>(KP_ARGS);               \
        else v8::bn_bwd_flat_k<GPv, MODEv><<<fgrid, TPB, 0, st>>>(KP_ARGS);                       \
    } while (0)
        if (mode == 0) { if (gp) KP_BWDF(1, 0); else KP_BWDF(0, 0); }
        else { if (gp) KP_BWDF(1, 1); else KP_BWDF(0, 1); }
#undef KP_BWDF
#undef KP_ARGS
        return kp_check_launch("bn_bwd_flat");
    }
    const long long rows = mode ? (long long)N * (H + 2 * dp) : (long long)N * H;
    // (apply: the 512-channel layers like 512 workgroups -- 108 -> 98, 29 -> 26 us --, the wide shallow ones the uncapped grid)
    const int rpb = rows_per_block(rows, W * (C / 8), kp_device_cus(), mode == 0 ? bn_blocks_cap(1, KP_BN_REDUCE_BLOCKS) : bn_blocks_cap(2, C >= 512 ? 512 : KP_BN_APPLY_BLOCKS));
    const unsigned grid = (unsigned)((rows + rpb - 1) / rpb);
    if (mode == 0) kp_set_last_rows(grid);
    KP_QUERY_RETURN();
#define KP_ARGS (const h16_t*)gsrc, argmax, (const h16_t*)y, bn, slope, partials, sums, (h16_t*)dy, dp, N, H, W, C, rows, rpb, walk
#define KP_BWD(GPv, MODEv)                                                                        \
    do {                                                                                          \
        if (vv == 4) v4::bn_bwd_fast_k<0, GPv, MODEv><<<grid, TPB, 0, st>>>(KP_ARGS);             \
        else v8::bn_bwd_fast_k<0, GPv, MODEv><<<grid, TPB, 0, st>>>(KP_ARGS);                     \
    } while (0)
    if (mode == 0) { if (gp) KP_BWD(1, 0); else KP_BWD(0, 0); }
    else { if (gp) KP_BWD(1, 1); else KP_BWD(0, 1); }
#undef KP_BWD
#undef KP_ARGS
    return kp_check_launch("bn_bwd_fast");
}

int kp_upsample2_bwd_fast(const void* gsrc, int gp, void* g, int N, int H, int W, int C, hipStream_t st) {
    const long long blocks = (long long)N * H * (C / 64);
    if (C % 64 != 0 || W > 128 || blocks > 0x7fffffffLL) return KP_E_UNSUPPORTED;
    upsample2_bwd_fast_k<<<(unsigned)blocks, 256, (size_t)2 * W * 64 * sizeof(float), st>>>((const h16_t*)gsrc, gp, (h16_t*)g, H, W, C);
    return kp_check_launch("upsample2_bwd_fast");
}
