// Fast bf16 paths of the 1x1 head (out_block = Conv2d(core_out, out, 1) + LeakyReLU, knn.py:123-125) for a
// SMALL number of output channels (K <= 16: the decoder's image channels, the keypoint net's K maps).
// At the decoder head (C = 64 -> K = 3 over N*128*128 pixels) the op is a pure stream over the [pixels][C] bf16
// activation (268 MB at 128 pairs), so all three kernels read / write it as 16-byte lanes on consecutive addresses:
// a thread owns one 8-channel vector of a pixel; the C/8 lanes of a pixel are adjacent and combine by shuffles.
//   fwd : out[n][k][hw] = act(bias[k] + sum_c x[m][c] w[k][c])                 (fp32 NCHW output)
//   dx  : dx[m][c]      = sum_k dz[m][k] w[k][c],  dz = gout * act'(out)       (bf16 NHWC output)
//   dw  : dw[k][c]     += sum_m dz[m][k] x[m][c];  db[k] += sum_m dz[m][k]     (fp32, atomics per workgroup)
// kp_elem.hip keeps the general kernels (any K, fp32 storage) and the entry points.
#include "kp_common.h"

namespace {

constexpr int TPB = 256, V = 8;

__device__ __forceinline__ void unpack8(const uint4 v, float (&o)[V]) {
    unpack_h16x2(v.x, o[0], o[1]);
    unpack_h16x2(v.y, o[2], o[3]);
    unpack_h16x2(v.z, o[4], o[5]);
    unpack_h16x2(v.w, o[6], o[7]);
}

// Sum over the CV adjacent lanes of a pixel (CV a power of two <= 64): DPP for the steps inside a 16-lane row
// (quad_perm xor 1 / xor 2, row_half_mirror, row_ror:8 -- no LDS round trip), ds_bpermute only across rows.
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float group_sum(float v, int CV) {
    if (CV >= 2) v += dpp_mov<0xB1>(v);    // lane ^ 1
    if (CV >= 4) v += dpp_mov<0x4E>(v);    // lane ^ 2
    if (CV >= 8) v += dpp_mov<0x141>(v);   // lane ^ 7: the other quad of the 8-lane group (quads are uniform by now)
    if (CV >= 16) v += dpp_mov<0x128>(v);  // lane ^ 8
    if (CV >= 32) v += __shfl_xor(v, 16, 64);
    if (CV >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

// KB = compile-time bound on K (1, 4 or 16); CV = C/8 is a power of two <= 64 (checked by the launcher).
// The backward kernels keep the thread's weight columns (or accumulators) in registers, are branch-free over k (rows
// k >= K are zero-padded: a `k < K` test inside the unrolled loop put every strided gout / out load behind its own
// s_waitcnt, 16 serialised round trips per pixel), walk TWO pixels per iteration and, for KB <= 4, issue the loads of the
// NEXT iteration before the arithmetic of the current one.  Pixel indices fit 32 bits (checked by the launcher), so the
// per-pixel (image, offset) split is a 32-bit division.
constexpr int PIX = 2;   // pixels in flight per thread

// forward, general form (K > 4 or C not 16 / 32 / 64; the image heads take head_fwd_k4_k below): weights in LDS (the
// register-resident variant measured slower at KB = 16: 173 VGPRs)
// BNIN (all three kernels that read x): x is the producing layer's conv output and `bn` its BatchNorm / activation, applied
// to each 16-byte chunk as it is loaded (KpBnIn, kp_common.h) -- a thread's 8 channels are fixed, their constants live in registers
template <int KB, bool BNIN>
__global__ __launch_bounds__(TPB) void head_fwd_small_k(const h16_t* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float slope,
                                                        float* __restrict__ out, long long M, int HW, int C, int K,
                                                        int ppb, KpBnIn bn) {
    extern __shared__ __attribute__((aligned(16))) float wl[];   // [K][C]
    for (int i = threadIdx.x; i < K * C; i += TPB) wl[i] = w[i];
    __syncthreads();
    const int CV = C / V, cv = threadIdx.x % CV, pl = threadIdx.x / CV, npl = TPB / CV;
    KpBnIn8 k8;
    if (BNIN) k8.load(bn, cv * V);
    const long long m0 = (long long)blockIdx.x * ppb;
    for (int i = pl; i < ppb; i += npl) {
        const long long m = m0 + i;
        if (m >= M) break;   // uniform within a pixel's lane group
        float xv[V];
        uint4 xraw = *reinterpret_cast<const uint4*>(x + m * C + cv * V);
        if (BNIN) xraw = k8.apply(xraw, bn.slope);
        unpack8(xraw, xv);
        float acc[KB];
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            acc[k] = 0.f;
            if (k < K) {
                const float4 w0 = *reinterpret_cast<const float4*>(wl + k * C + cv * V);
                const float4 w1 = *reinterpret_cast<const float4*>(wl + k * C + cv * V + 4);
                acc[k] = xv[0] * w0.x + xv[1] * w0.y + xv[2] * w0.z + xv[3] * w0.w + xv[4] * w1.x + xv[5] * w1.y +
                         xv[6] * w1.z + xv[7] * w1.w;
            }
        }
#pragma unroll
        for (int k = 0; k < KB; ++k)
            if (k < K) acc[k] = group_sum(acc[k], CV);
        if (cv == 0) {
            const long long n = m / HW, hw = m - n * HW;
#pragma unroll
            for (int k = 0; k < KB; ++k)
                if (k < K) out[(n * K + k) * HW + hw] = act_fwd(acc[k] + (bias ? bias[k] : 0.f), slope);
        }
    }
}

// forward for K <= 4 and C = 16 / 32 / 64 (the image heads): the kernel above is one load in flight per thread and a `k < K` branch
// around every weight read (121 us for the 268 MB of model F's decoder head, 2.4 TB/s).  Here the lanes-per-pixel count is a
// compile-time constant, the thread's weight columns (zero rows for k >= K) and bias live in registers, the loads of the next two
// pixels are issued before the arithmetic of the current two, the dot products are FMA chains, and lane cv of a pixel stores output
// channel cv (one store instruction per pixel group instead of K).  No division in the loop: (image, offset) advance with the
// pixel index.
template <int CV, int KK, bool BNIN>   // KK = 1 (grey-scale images) or 4 (K = 2 .. 4, rows k >= K zero)
__global__ __launch_bounds__(TPB) void head_fwd_k4_k(const h16_t* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float slope, float* __restrict__ out,
                                                     long long M, int HW, int K, int ppb, KpBnIn bn) {
    constexpr int C = CV * V, NPL = TPB / CV, PF = 2;
    const int cv = threadIdx.x % CV, pl = threadIdx.x / CV;
    float wr[KK][V], bs[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) {
        bs[k] = (bias && k < K) ? bias[k] : 0.f;
#pragma unroll
        for (int e = 0; e < V; ++e) wr[k][e] = k < K ? w[k * C + cv * V + e] : 0.f;
    }
    KpBnIn8 k8;
    if (BNIN) k8.load(bn, cv * V);
    const int m0 = (int)((long long)blockIdx.x * ppb), Mi = (int)M;
    int n = (m0 + pl) / HW, hw = (m0 + pl) - n * HW;          // of pixel m0 + i (u = 0); u > 0 adds u * NPL
    uint4 raw[PF] = {}, rawn[PF] = {};   // (zero: the last rotation copies a `rawn` nothing was fetched into)
    int m[PF] = {}, mn[PF] = {};
    auto issue = [&](int i, uint4 (&rw)[PF], int (&mm)[PF]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            mm[u] = m0 + i + u * NPL;
            if (!(i + u * NPL < ppb && mm[u] < Mi)) mm[u] = -1;
            rw[u] = mm[u] >= 0 ? *reinterpret_cast<const uint4*>(x + (long long)mm[u] * C + cv * V) : make_uint4(0, 0, 0, 0);
        }
    };
    if (pl < ppb) issue(pl, raw, m);
    for (int i = pl; i < ppb; i += PF * NPL) {
        if (i + PF * NPL < ppb) issue(i + PF * NPL, rawn, mn);   // in flight during this iteration's arithmetic
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const uint4 r = BNIN ? k8.apply(raw[u], bn.slope) : raw[u];
            float xv[V], acc[KK];
            unpack8(r, xv);
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                acc[k] = 0.f;
#pragma unroll
                for (int e = 0; e < V; ++e) acc[k] = fmaf(xv[e], wr[k][e], acc[k]);
                acc[k] = group_sum(acc[k], CV);
            }
            int hh = hw + u * NPL;                            // (PF * NPL <= HW, the launcher's condition: at most one wrap)
            const bool wrap = hh >= HW;
            hh -= wrap ? HW : 0;
            const int nn = n + (wrap ? 1 : 0);
#pragma unroll
            for (int k0 = 0; k0 < KK; k0 += CV) {              // lane cv: channels cv, cv + CV, ...
                const int k = k0 + cv;
                float v = acc[0] + bs[0];
#pragma unroll
                for (int j = 1; j < KK; ++j) v = k == j ? acc[j] + bs[j] : v;
                if (m[u] >= 0 && k < K) out[((long long)nn * K + k) * HW + hh] = act_fwd(v, slope);
            }
        }
        hw += PF * NPL;
        if (hw >= HW) { hw -= HW; ++n; }
#pragma unroll
        for (int u = 0; u < PF; ++u) { raw[u] = rawn[u]; m[u] = mn[u]; }
    }
}

// dz[k] of one pixel: all 2*KB strided loads issued back to back (rows k >= K re-read row K-1 and are zeroed).
// KB = 4 with at least four lanes per pixel (C >= 32): the lanes of a pixel would all load the same 2 x 4 values -- eight
// address-unit passes per pixel for 32 bytes, the bound of the K <= 4 backward kernels (142 us for the 268 MB activation of model
// F's decoder head).  Lane cv loads row cv & 3 only (two loads), computes ITS dz, and the four values go round the quad by DPP
// (quad_perm broadcasts: no LDS, no extra rounding -- every dz[k] is the same product as before).
// QUAD (KB = 4 with at least four lanes per pixel, C >= 32): the lanes of a pixel would all load the same 2 x 4 values -- eight
// address-unit passes per pixel for 32 bytes.  Lane cv loads row cv & 3 only (two loads), computes ITS dz, and the four values go
// round the quad by DPP (quad_perm broadcasts: no LDS, no extra rounding -- every dz[k] is the same product as before).
template <int KB, bool QUAD> struct DzLoad { float g[QUAD ? 1 : KB], o[QUAD ? 1 : KB]; };   // what dz_issue fetched for one pixel

template <int KB, bool QUAD>
__device__ __forceinline__ void dz_issue(const float* __restrict__ gout, const float* __restrict__ out, int m, int HW, int K,
                                         int cv, DzLoad<KB, QUAD>& r) {
    const int mm = m < 0 ? 0 : m;                    // (pixels past the end read pixel 0; dz_finish zeroes them)
    const int n = mm / HW, hw = mm - n * HW;
    if (QUAD) {
        const int k = cv & 3;
        const long long idx = ((long long)n * K + (k < K ? k : K - 1)) * HW + hw;
        r.g[0] = gout[idx];
        r.o[0] = out[idx];
    } else {
        const long long base = ((long long)n * K) * HW + hw;
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            const long long idx = base + (long long)(k < K ? k : K - 1) * HW;
            r.g[QUAD ? 0 : k] = gout[idx];
            r.o[QUAD ? 0 : k] = out[idx];
        }
    }
}
template <int KB, bool QUAD>
__device__ __forceinline__ void dz_finish(const DzLoad<KB, QUAD>& r, int m, int K, float slope, int cv, float (&dz)[KB]) {
    if (QUAD) {
        const int k = cv & 3;
        const float d = (m >= 0 && k < K) ? r.g[0] * act_grad(r.o[0], slope) : 0.f;
        dz[0] = dpp_mov<0x00>(d);
        dz[1 % KB] = dpp_mov<0x55>(d);
        dz[2 % KB] = dpp_mov<0xAA>(d);
        dz[3 % KB] = dpp_mov<0xFF>(d);
    } else {
#pragma unroll
        for (int k = 0; k < KB; ++k) dz[k] = (m >= 0 && k < K) ? r.g[QUAD ? 0 : k] * act_grad(r.o[QUAD ? 0 : k], slope) : 0.f;
    }
}
// (the fused head + BatchNorm backward kernels: one call, the quad form chosen at run time)
template <int KB>
__device__ __forceinline__ void load_dz(const float* __restrict__ gout, const float* __restrict__ out, int m, int HW, int K,
                                        float slope, float (&dz)[KB], int cv, int CV) {
    if (KB == 4 && CV >= 4) {
        DzLoad<KB, KB == 4> r;
        dz_issue<KB, KB == 4>(gout, out, m, HW, K, cv, r);
        dz_finish<KB, KB == 4>(r, m, K, slope, cv, dz);
    } else {
        DzLoad<KB, false> r;
        dz_issue<KB, false>(gout, out, m, HW, K, cv, r);
        dz_finish<KB, false>(r, m, K, slope, cv, dz);
    }
}

// The two backward kernels below are software-pipelined for KB <= 4: the loads of iteration i + 1 are issued before the arithmetic
// of iteration i (the straight loop alternated "all waves wait" with "all waves compute": ~50 us of VALU work + ~55 us of HBM time
// took 124 us at model F's decoder head).
template <int KB, bool QUAD>
__global__ __launch_bounds__(TPB) void head_dx_small_k(const float* __restrict__ gout, const float* __restrict__ out,
                                                       const float* __restrict__ w, float slope,
                                                       h16_t* __restrict__ dx, long long M, int HW, int C, int K,
                                                       int ppb) {
    const int CV = C / V, cv = threadIdx.x % CV, pl = threadIdx.x / CV, npl = TPB / CV;
    float wr[KB][V];
#pragma unroll
    for (int k = 0; k < KB; ++k)
#pragma unroll
        for (int e = 0; e < V; e += 4) {
            const float4 q = k < K ? *reinterpret_cast<const float4*>(w + k * C + cv * V + e) : make_float4(0.f, 0.f, 0.f, 0.f);
            wr[k][e] = q.x; wr[k][e + 1] = q.y; wr[k][e + 2] = q.z; wr[k][e + 3] = q.w;
        }
    const int m0 = (int)((long long)blockIdx.x * ppb), Mi = (int)M;
    constexpr bool PIPE = KB <= 4;
    DzLoad<KB, QUAD> cur[PIX] = {}, nxt[PIX] = {};   // (zero: the last rotation copies a `nxt` nothing was fetched into)
    int m[PIX] = {}, mn[PIX] = {};
    auto issue = [&](int i, DzLoad<KB, QUAD> (&r)[PIX], int (&mm)[PIX]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PIX; ++u) {
            mm[u] = m0 + i + u * npl;
            if (!(i + u * npl < ppb && mm[u] < Mi)) mm[u] = -1;
            dz_issue<KB, QUAD>(gout, out, mm[u], HW, K, cv, r[u]);
        }
    };
    if (PIPE && pl < ppb) issue(pl, cur, m);
    for (int i = pl; i < ppb; i += PIX * npl) {
        if (PIPE) {
            if (i + PIX * npl < ppb) issue(i + PIX * npl, nxt, mn);
        } else {
            issue(i, cur, m);
        }
#pragma unroll
        for (int u = 0; u < PIX; ++u) {
            float dz[KB];
            dz_finish<KB, QUAD>(cur[u], m[u], K, slope, cv, dz);
            if (m[u] < 0) continue;
            float r[V];
#pragma unroll
            for (int e = 0; e < V; ++e) r[e] = 0.f;
#pragma unroll
            for (int k = 0; k < KB; ++k)
#pragma unroll
                for (int e = 0; e < V; ++e) r[e] = fmaf(dz[k], wr[k][e], r[e]);
            uint4 pk;
            pk.x = pack_h16x2(r[0], r[1]); pk.y = pack_h16x2(r[2], r[3]);
            pk.z = pack_h16x2(r[4], r[5]); pk.w = pack_h16x2(r[6], r[7]);
            *reinterpret_cast<uint4*>(dx + (long long)m[u] * C + cv * V) = pk;
        }
        if (PIPE) {
#pragma unroll
            for (int u = 0; u < PIX; ++u) { cur[u] = nxt[u]; m[u] = mn[u]; }
        }
    }
}

// out(i) = sum over the rows r of src[r * stride + i], i < width, in a FIXED order: with width <= TPB / 2 the rows are cut into S
// segments (S the largest power of two with S * width <= TPB, S <= rows) summed by different threads, then the S partial sums are
// added in order -- a 16-float gradient (C = 16, K = 1) over 128 pixel lanes was 128 dependent LDS reads on 16 threads, 6 us per
// workgroup.  S = 1 (model F's 64 -> 3 head: width 192): one thread per column, rows in order, as before.  `part`: TPB floats.
template <class F>
__device__ __forceinline__ void block_sum_rows(const float* src, int rows, int stride, int width, float* part, F&& emit) {
    int S = 1;
    while (S * 2 * width <= TPB && S * 2 <= rows) S *= 2;
    if (S == 1) {
        for (int i = threadIdx.x; i < width; i += TPB) {
            float s = 0.f;
            for (int r = 0; r < rows; ++r) s += src[(size_t)r * stride + i];
            emit(i, s);
        }
        return;
    }
    const int per = rows / S, t = threadIdx.x;      // (rows = pixel lanes per workgroup: a power of two)
    if (t < S * width) {
        const int seg = t / width, i = t - seg * width;
        float s = 0.f;
        for (int r = seg * per; r < (seg + 1) * per; ++r) s += src[(size_t)r * stride + i];
        part[t] = s;
    }
    __syncthreads();
    if (t < width) {
        float s = 0.f;
        for (int seg = 0; seg < S; ++seg) s += part[seg * width + t];
        emit(t, s);
    }
    __syncthreads();
}

template <int KB, bool QUAD, bool BNIN>
__global__ __launch_bounds__(TPB) void head_dw_small_k(const float* __restrict__ gout, const float* __restrict__ out,
                                                       const h16_t* __restrict__ x, float slope,
                                                       float* __restrict__ dw, float* __restrict__ db,
                                                       float* __restrict__ ws, long long M, int HW, int C, int K,
                                                       int ppb, KpBnIn bn) {
    extern __shared__ __attribute__((aligned(16))) float red[];   // [npl][KC][C] (+ [npl][KC] for db, + TPB for block_sum_rows)
    const int CV = C / V, cv = threadIdx.x % CV, pl = threadIdx.x / CV, npl = TPB / CV;
    KpBnIn8 k8;
    if (BNIN) k8.load(bn, cv * V);
    float acc[KB][V], dbs[KB];
#pragma unroll
    for (int k = 0; k < KB; ++k) {
        dbs[k] = 0.f;
#pragma unroll
        for (int e = 0; e < V; ++e) acc[k][e] = 0.f;
    }
    const int m0 = (int)((long long)blockIdx.x * ppb), Mi = (int)M;
    constexpr bool PIPE = KB <= 4;
    DzLoad<KB, QUAD> cur[PIX] = {}, nxt[PIX] = {};   // (zero: the last rotation copies a `nxt` nothing was fetched into)
    uint4 raw[PIX] = {}, rawn[PIX] = {};
    int m[PIX] = {}, mn[PIX] = {};
    auto issue = [&](int i, DzLoad<KB, QUAD> (&r)[PIX], uint4 (&rw)[PIX], int (&mm)[PIX]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < PIX; ++u) {
            mm[u] = m0 + i + u * npl;
            if (!(i + u * npl < ppb && mm[u] < Mi)) mm[u] = -1;
            rw[u] = mm[u] >= 0 ? *reinterpret_cast<const uint4*>(x + (long long)mm[u] * C + cv * V) : make_uint4(0, 0, 0, 0);
            dz_issue<KB, QUAD>(gout, out, mm[u], HW, K, cv, r[u]);
        }
    };
    if (PIPE && pl < ppb) issue(pl, cur, raw, m);
    for (int i = pl; i < ppb; i += PIX * npl) {
        if (PIPE) {
            if (i + PIX * npl < ppb) issue(i + PIX * npl, nxt, rawn, mn);
        } else {
            issue(i, cur, raw, m);
        }
#pragma unroll
        for (int u = 0; u < PIX; ++u) {
            float dz[KB], xv[V];
            dz_finish<KB, QUAD>(cur[u], m[u], K, slope, cv, dz);
            // (pixels past the end: dz = 0, whatever the transform makes of the zero chunk)
            unpack8(BNIN ? k8.apply(raw[u], bn.slope) : raw[u], xv);
#pragma unroll
            for (int k = 0; k < KB; ++k) {
                dbs[k] += dz[k];
#pragma unroll
                for (int e = 0; e < V; ++e) acc[k][e] = fmaf(dz[k], xv[e], acc[k][e]);
            }
        }
        if (PIPE) {
#pragma unroll
            for (int u = 0; u < PIX; ++u) { cur[u] = nxt[u]; raw[u] = rawn[u]; m[u] = mn[u]; }
        }
    }
    // block reduction over the pixel lanes, KC output channels at a time (LDS = npl * KC * C floats, <= 32 KB)
    constexpr int KC = 4;
    float* rdb = red + (size_t)npl * KC * C;
    for (int k0 = 0; k0 < K; k0 += KC) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KB; ++k) {
            if (k >= k0 && k < k0 + KC && k < K) {
#pragma unroll
                for (int e = 0; e < V; ++e) red[((size_t)pl * KC + (k - k0)) * C + cv * V + e] = acc[k][e];
                if (cv == 0) rdb[pl * KC + (k - k0)] = dbs[k];
            }
        }
        __syncthreads();
        const int kn = K - k0 < KC ? K - k0 : KC;
        float* part = rdb + (size_t)npl * KC;
        float* wsb = ws ? ws + (size_t)blockIdx.x * ((size_t)K * C + K) : nullptr;   // this workgroup's partial
        block_sum_rows(red, npl, KC * C, kn * C, part, [&](int i, float s) {
            if (wsb) wsb[(size_t)k0 * C + i] = s;
            else unsafeAtomicAdd(dw + (size_t)k0 * C + i, s);
        });
        if (db || ws)
            block_sum_rows(rdb, npl, KC, kn, part, [&](int i, float s) {
                if (wsb) wsb[(size_t)K * C + k0 + i] = s;
                else unsafeAtomicAdd(db + k0 + i, s);
            });
    }
}

bool small_shape(int C, int K) {
    const int CV = C / V;
    return C % V == 0 && CV >= 1 && CV <= 64 && (CV & (CV - 1)) == 0 && K >= 1 && K <= 16;
}

}  // namespace

#define KP_HEAD_KB(K, CALL)                          \
    do {                                             \
        if ((K) == 1) { constexpr int KB = 1; CALL; } \
        else if ((K) <= 4) { constexpr int KB = 4; CALL; } \
        else { constexpr int KB = 16; CALL; }        \
    } while (0)
// ... and the quad form of the dz loads (K = 2 .. 4 with at least four lanes per pixel)
#define KP_HEAD_KBQ(K, C, CALL)                                                          \
    do {                                                                                 \
        if ((K) == 1) { constexpr int KB = 1; constexpr bool QUAD = false; CALL; }       \
        else if ((K) <= 4 && (C) / V >= 4) { constexpr int KB = 4; constexpr bool QUAD = true; CALL; } \
        else if ((K) <= 4) { constexpr int KB = 4; constexpr bool QUAD = false; CALL; }  \
        else { constexpr int KB = 16; constexpr bool QUAD = false; CALL; }               \
    } while (0)

// bn != nullptr: x is the producing layer's conv output, *bn its BatchNorm / activation (kp_head_fwd_bnin)
int kp_head_fwd_fast(const void* x, const float* w, const float* bias, float slope, float* out, long long M, int HW,
                     int C, int K, hipStream_t st, const KpBnIn* bn) {
    if (!small_shape(C, K) || M >= (1LL << 31)) return KP_E_UNSUPPORTED;
    const int npl = TPB / (C / V);
    if (K <= 4 && (C == 16 || C == 32 || C == 64) && HW >= 2 * npl) {
        // two pixels computed while the next two load, 16 per thread and workgroup (the weights are loaded once per thread)
        const int ppb = npl * 16;
        const unsigned grid = (unsigned)((M + ppb - 1) / ppb);
#define KP_HEAD_F4(CVv, KKv)                                                                                                         \
    do {                                                                                                                             \
        if (bn) head_fwd_k4_k<CVv, KKv, true><<<grid, TPB, 0, st>>>((const h16_t*)x, w, bias, slope, out, M, HW, K, ppb, *bn);       \
        else head_fwd_k4_k<CVv, KKv, false><<<grid, TPB, 0, st>>>((const h16_t*)x, w, bias, slope, out, M, HW, K, ppb, KpBnIn{});    \
    } while (0)
#define KP_HEAD_F4K(CVv) do { if (K == 1) KP_HEAD_F4(CVv, 1); else KP_HEAD_F4(CVv, 4); } while (0)
        if (C == 16) KP_HEAD_F4K(2); else if (C == 32) KP_HEAD_F4K(4); else KP_HEAD_F4K(8);
#undef KP_HEAD_F4K
#undef KP_HEAD_F4
        return kp_check_launch("head_fwd_k4");
    }
    const int ppb = npl * 8;
    const unsigned grid = (unsigned)((M + ppb - 1) / ppb);
    const size_t lds = (size_t)K * C * sizeof(float);
    if (bn) KP_HEAD_KB(K, (head_fwd_small_k<KB, true><<<grid, TPB, lds, st>>>((const h16_t*)x, w, bias, slope, out, M, HW, C, K, ppb, *bn)));
    else KP_HEAD_KB(K, (head_fwd_small_k<KB, false><<<grid, TPB, lds, st>>>((const h16_t*)x, w, bias, slope, out, M, HW, C, K, ppb, KpBnIn{})));
    return kp_check_launch("head_fwd_fast");
}
bool kp_head_small_shape(long long M, int C, int K) { return small_shape(C, K) && M < (1LL << 31); }

int kp_head_dx_fast(const float* gout, const float* out, const float* w, float slope, void* dx, long long M, int HW,
                    int C, int K, hipStream_t st) {
    if (!small_shape(C, K) || M >= (1LL << 31)) return KP_E_UNSUPPORTED;
    // the register-resident weights are loaded once per thread: 8..32 pixels per thread, ~4096 workgroups where the map allows
    const int npl = TPB / (C / V);
    long long per = M / ((long long)npl * 4096);   // (measured at model F's head, ~1024 / 2048 / 4096 workgroups: 96 / 90 / 85 us)
    per = per < 8 ? 8 : (per > 32 ? 32 : per);
    const int ppb = npl * (int)per;
    const unsigned grid = (unsigned)((M + ppb - 1) / ppb);
    KP_HEAD_KBQ(K, C, (head_dx_small_k<KB, QUAD><<<grid, TPB, 0, st>>>(gout, out, w, slope, (h16_t*)dx, M, HW, C, K, ppb)));
    return kp_check_launch("head_dx_fast");
}

// workgroups (= partial gradients in deterministic mode) of the small-shape weight-gradient kernel; 0: not its shape
static long long dw_fast_plan(long long M, int C, int K, long long& ppb) {
    if (!small_shape(C, K) || M >= (1LL << 31)) return 0;
    const int npl = TPB / (C / V);
    const size_t lds = ((size_t)npl * 4 * C + (size_t)npl * 4 + TPB) * sizeof(float);   // KC = 4 channels per reduction pass
    if (lds > 64 * 1024) return 0;
    ppb = (M + 1023) / 1024;                    // ~1024 workgroups (four per CU: one resident round at the kernels' 4 waves per SIMD), each a
                                                // contiguous run of pixels; measured 512 / 768 / 1024 / 2048 / 4096: 114 / 99 / 93 / 99 / 110 us at model F's head
    if (ppb < npl) ppb = npl;
    ppb = (ppb + npl - 1) / npl * npl;
    return (M + ppb - 1) / ppb;
}
long long kp_head_dw_parts_fast(long long M, int C, int K) { long long ppb; return dw_fast_plan(M, C, K, ppb); }

int kp_head_dw_fast(const float* gout, const float* out, const void* x, float slope, float* dw, float* db, float* ws,
                    long long M, int HW, int C, int K, hipStream_t st, const KpBnIn* bn) {
    long long ppb;
    const long long parts = dw_fast_plan(M, C, K, ppb);
    if (parts <= 0) return KP_E_UNSUPPORTED;
    const int npl = TPB / (C / V);
    const size_t lds = ((size_t)npl * 4 * C + (size_t)npl * 4 + TPB) * sizeof(float);
    const unsigned grid = (unsigned)parts;
    if (bn) KP_HEAD_KBQ(K, C, (head_dw_small_k<KB, QUAD, true><<<grid, TPB, lds, st>>>(gout, out, (const h16_t*)x, slope, dw, db, ws, M, HW, C, K, (int)ppb, *bn)));
    else KP_HEAD_KBQ(K, C, (head_dw_small_k<KB, QUAD, false><<<grid, TPB, lds, st>>>(gout, out, (const h16_t*)x, slope, dw, db, ws, M, HW, C, K, (int)ppb, KpBnIn{})));
    return kp_check_launch("head_dw_fast");
}

