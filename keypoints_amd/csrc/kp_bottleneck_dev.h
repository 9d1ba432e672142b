// Device pieces shared by the two forms of the fused keypoint bottleneck: kp_bottleneck_fused.hip (fp32 NCHW feature maps, the
// reference's own layout) and kp_bottleneck_t.hip (16-bit NHWC feature maps straight from / to the neighbouring Units).
// Phase A (heat-maps -> keypoints: functional.py:11-14,22-24,37-44) is the same code in both, so the keypoints, marginals and
// masks are bit-identical whichever form runs.
#pragma once
#include "kp_common.h"
#include "kp_rowsum.h"

namespace {

constexpr int NT = 1024;            // threads per workgroup
constexpr int NW = NT / 64;
constexpr int TILE_FLOATS = 0;      // (the per-wave LDS staging tile of the first version: wave_keypoint reads the map straight from global memory now)
constexpr int MAXDIM = 256;         // H, W <= MAXDIM (per-wave marginal vectors)
constexpr int WAVE_FLOATS = TILE_FLOATS + 2 * MAXDIM;
constexpr int BWD_LDS_FLOATS = 16384;

// LDS traffic inside ONE wavefront: order the wave's own writes before its reads (no workgroup barrier involved)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// marginal (log-)softmax of one heat-map by one wavefront; returns the keypoint coordinate pair in (ky, kx) on all lanes.
// The marginals straight from global memory, no LDS tile: a lane owns columns lane, lane + 64, lane + 128, lane + 192 (W <= 256), the
// wavefront walks the rows -- coalesced 4-byte loads, two batches of 8 rows in flight.  Column sums are the lane's running sums
// over the rows in order (the order they always had); a row sum is the lane's partial over its <= 4 columns, then the halving
// butterfly of kp_rows8_over64 (kp_rowsum.h), eight rows per pass.  (The first version staged 12-row chunks through an LDS tile and let <= 12 lanes walk a row each: 84
// dependent LDS reads per chunk and lane -- the phase took 56 of the kernel's 128 us at 256 pairs x 8 maps of 84 x 84, 1 TB/s
// of heat-map reads, tools/probes/bneck_t_abl.py.  The row-sum order changed with it, in kp_spatial_softmax_fwd too: the
// fused and the piecewise kernels stay bit-identical to each other.)
__device__ void wave_keypoint(const float* __restrict__ src, int H, int W, float* tile, float* rowv, float* colv,
                              float* __restrict__ pdst, float* __restrict__ qdst, float& ky, float& kx) {
    (void)tile;
    const int lane = threadIdx.x & 63;
    constexpr int RB = 8;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    const bool on1 = lane + 64 < W, on2 = lane + 128 < W, on3 = lane + 192 < W, on0 = lane < W;
    auto load_batch = [&](float (&buf)[RB][4], int r0) {
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = r0 + r;
            const float* sp = src + (long long)(row < H ? row : 0) * W + lane;
            const bool rv = row < H;
            buf[r][0] = (rv && on0) ? sp[0] : 0.f;
            buf[r][1] = (rv && on1) ? sp[64] : 0.f;
            buf[r][2] = (rv && on2) ? sp[128] : 0.f;
            buf[r][3] = (rv && on3) ? sp[192] : 0.f;
        }
    };
    auto reduce_batch = [&](const float (&buf)[RB][4], int r0) {
        float pp[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {   // (rows beyond H were loaded as zeros)
            pp[r] = buf[r][0];
            cs[0] += buf[r][0];
#pragma unroll
            for (int q = 1; q < 4; ++q) { pp[r] += buf[r][q]; cs[q] += buf[r][q]; }
        }
        int idx;
        const float tot = kp_rows8_over64(pp, lane, idx);
        if ((lane & 7) == 0 && r0 + idx < H) rowv[r0 + idx] = tot / (float)W;
    };
    float A[RB][4], B[RB][4];
    load_batch(A, 0);
    for (int r0 = 0; r0 < H; r0 += 2 * RB) {
        load_batch(B, r0 + RB);
        reduce_batch(A, r0);
        load_batch(A, r0 + 2 * RB);
        reduce_batch(B, r0 + RB);
    }
    if (on0) colv[lane] = cs[0];
    if (on1) colv[lane + 64] = cs[1];
    if (on2) colv[lane + 128] = cs[2];
    if (on3) colv[lane + 192] = cs[3];
    wave_lds_sync();
    for (int j = lane; j < W; j += 64) colv[j] = colv[j] / (float)H;
    wave_lds_sync();
    float key[2];
#pragma unroll
    for (int axis = 0; axis < 2; ++axis) {
        const int L = axis == 0 ? H : W;
        const float* a = axis == 0 ? rowv : colv;
        float* dst = axis == 0 ? pdst : qdst;
        float mx = -INFINITY;
        for (int i = lane; i < L; i += 64) mx = fmaxf(mx, a[i]);
        mx = wave_max(mx);
        float se = 0.f;
        for (int i = lane; i < L; i += 64) se += expf(a[i] - mx);
        se = wave_sum(se);
        const float lse = logf(se);
        float kk = 0.f;
        for (int i = lane; i < L; i += 64) {
            const float lp = (a[i] - mx) - lse;
            const float r = ruler_at(i, L);
            if (r > 0.f) kk += expf(lp + logf(r));   // ruler[0] = 0: log = -inf, the term is exactly 0 (functional.py:23)
            if (dst) dst[i] = expf(lp);
        }
        key[axis] = wave_sum(kk);
    }
    ky = key[0];
    kx = key[1];
}

template <int V> struct Vec;
template <> struct Vec<4> {
    static __device__ __forceinline__ void ld(const float* p, float (&o)[4]) {
        const float4 v = *reinterpret_cast<const float4*>(p);
        o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
    }
    static __device__ __forceinline__ void st(float* p, const float (&o)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
    }
    static __device__ __forceinline__ void st8(unsigned char* p, const int (&o)[4]) {
        *reinterpret_cast<uint32_t*>(p) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
    }
    static __device__ __forceinline__ void ld8(const unsigned char* p, int (&o)[4]) {
        const uint32_t v = *reinterpret_cast<const uint32_t*>(p);
        o[0] = v & 255; o[1] = (v >> 8) & 255; o[2] = (v >> 16) & 255; o[3] = v >> 24;
    }
};
template <> struct Vec<1> {
    static __device__ __forceinline__ void ld(const float* p, float (&o)[1]) { o[0] = *p; }
    static __device__ __forceinline__ void st(float* p, const float (&o)[1]) { *p = o[0]; }
    static __device__ __forceinline__ void st8(unsigned char* p, const int (&o)[1]) { *p = (unsigned char)o[0]; }
    static __device__ __forceinline__ void ld8(const unsigned char* p, int (&o)[1]) { o[0] = *p; }
};

}  // namespace
