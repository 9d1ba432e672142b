// Device pieces shared by the two forms of the fused keypoint bottleneck: kp_bottleneck_fused.hip (fp32 NCHW feature maps, the
// reference's own layout) and kp_bottleneck_t.hip (16-bit NHWC feature maps straight from / to the neighbouring Units).
// Phase A (heat-maps -> keypoints: functional.py:11-14,22-24,37-44) is the same code in both, so the keypoints, marginals and
// masks are bit-identical whichever form runs.
#pragma once
#include "kp_common.h"

namespace {

constexpr int NT = 1024;            // threads per workgroup
constexpr int NW = NT / 64;
constexpr int TILE_FLOATS = 1024;   // per-wave LDS staging tile (4 KiB)
constexpr int MAXDIM = 256;         // H, W <= MAXDIM (per-wave marginal vectors)
constexpr int WAVE_FLOATS = TILE_FLOATS + 2 * MAXDIM;
constexpr int BWD_LDS_FLOATS = 16384;

// LDS traffic inside ONE wavefront: order the wave's own writes before its reads (no workgroup barrier involved)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// marginal (log-)softmax of one heat-map by one wavefront; returns the keypoint coordinate pair in (ky, kx) on all lanes
__device__ void wave_keypoint(const float* __restrict__ src, int H, int W, float* tile, float* rowv, float* colv,
                              float* __restrict__ pdst, float* __restrict__ qdst, float& ky, float& kx) {
    const int lane = threadIdx.x & 63;
    const int pitch = W | 1;
    int rc = TILE_FLOATS / pitch;
    if (rc > H) rc = H;
    for (int j = lane; j < W; j += 64) colv[j] = 0.f;
    const bool vec = (W & 3) == 0 && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    for (int r0 = 0; r0 < H; r0 += rc) {
        const int rows = (H - r0 < rc) ? H - r0 : rc;
        const float* s = src + (long long)r0 * W;
        if (vec) {
            for (int e = lane * 4; e < rows * W; e += 256) {
                const float4 v = *reinterpret_cast<const float4*>(s + e);
                float* d = tile + (e / W) * pitch + (e % W);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int e = lane; e < rows * W; e += 64) tile[(e / W) * pitch + (e % W)] = s[e];
        }
        wave_lds_sync();
        for (int i = lane; i < rows; i += 64) {   // lanes walk different rows; odd pitch => conflict-free
            float a = 0.f;
            for (int j = 0; j < W; ++j) a += tile[i * pitch + j];
            rowv[r0 + i] = a / (float)W;
        }
        for (int j = lane; j < W; j += 64) {
            float a = colv[j];
            for (int i = 0; i < rows; ++i) a += tile[i * pitch + j];
            colv[j] = a;
        }
        wave_lds_sync();
    }
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
