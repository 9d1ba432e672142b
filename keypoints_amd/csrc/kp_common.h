// Shared device helpers for libkp_hip.so (gfx950 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// ---- the 16-bit storage type ("T" of the C ABI when dtype != KP_F32) ------------------------------------------------
// Every kernel source that touches 16-bit activations / packed weights is compiled TWICE (Makefile): once with bfloat16
// as the storage type (dtype KP_BF16) and once, with -DKP_F16_BUILD, with IEEE half (dtype KP_F16: 3 more mantissa bits,
// the reference's apex-O2 storage type; needs loss scaling, keypoints_amd/amp.py).  The kernels are written against
// h16_t / h2f / f2h / pack_h16x2 / unpack_h16x2 / kp_mfma16 and never look at the bits themselves.  External symbols
// of those sources carry a per-build suffix (kp_h16_names.h); kp_api.hip holds the dtype dispatch.
#ifdef KP_F16_BUILD
#define KP_H16 KP_F16
#define KP_SFX(n) n##__f16
typedef _Float16 kp_h16_native;
#else
#define KP_H16 KP_BF16
#define KP_SFX(n) n##__bf16
typedef __bf16 kp_h16_native;
#endif
#ifdef KP_H16_SOURCE
#include "kp_h16_names.h"   // must precede kp_hip.h: the dtype-taking entry points are declared under their build's name
#endif
#include "../../include/kp_hip.h"

typedef uint16_t h16_t;   // raw bits of the 16-bit storage type

typedef kp_h16_native kp_h16x2 __attribute__((ext_vector_type(2)));
typedef float kp_f32x2 __attribute__((ext_vector_type(2)));
// conversions are the hardware ones (round to nearest even): v_cvt_pk_bf16_f32 / v_cvt_f16_f32, v_cvt_f32_f16
__device__ __forceinline__ float h2f(h16_t v) {
#ifdef KP_F16_BUILD
    return (float)__builtin_bit_cast(_Float16, v);
#else
    return __uint_as_float(((uint32_t)v) << 16);
#endif
}
__device__ __forceinline__ h16_t f2h(float f) { return __builtin_bit_cast(h16_t, (kp_h16_native)f); }
__device__ __forceinline__ uint32_t pack_h16x2(float lo, float hi) {
    const kp_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, kp_h16x2));
}
__device__ __forceinline__ void unpack_h16x2(uint32_t w, float& lo, float& hi) {
#ifdef KP_F16_BUILD
    const kp_h16x2 h = __builtin_bit_cast(kp_h16x2, w);
    lo = (float)h[0];
    hi = (float)h[1];
#else
    lo = __uint_as_float(w << 16);
    hi = __uint_as_float(w & 0xffff0000u);
#endif
}

// v_mfma_f32_32x32x16_{bf16,f16}: same shape, same rate; which one depends on the build's 16-bit storage type.
// A8: any 16-byte vector of eight 16-bit values (the kernels keep fragments as raw shorts); C16: 16 x float.
template <typename A8, typename C16> __device__ __forceinline__ C16 kp_mfma16(A8 a, A8 b, C16 c, int, int, int) {
    typedef kp_h16_native v8 __attribute__((ext_vector_type(8)));
    static_assert(sizeof(A8) == 16 && sizeof(C16) == 64, "kp_mfma16: 8 x 16-bit operands, 16 x fp32 accumulator");
#ifdef KP_F16_BUILD
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
#endif
}

// v_mfma_f32_16x16x32_{bf16,f16}: D[16][16] += A[16][32] B[32][16]; a lane holds 8 consecutive reduction elements
// 8 * (lane >> 4) .. of row / column lane & 15; D: column lane & 15, rows 4 * (lane >> 4) + reg.  C4: 4 x float.
template <typename A8, typename C4> __device__ __forceinline__ C4 kp_mfma16x16(A8 a, A8 b, C4 c) {
    typedef kp_h16_native v8 __attribute__((ext_vector_type(8)));
    static_assert(sizeof(A8) == 16 && sizeof(C4) == 16, "kp_mfma16x16: 8 x 16-bit operands, 4 x fp32 accumulator");
#ifdef KP_F16_BUILD
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8, a), __builtin_bit_cast(v8, b), c, 0, 0, 0);
#endif
}

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<h16_t>(const h16_t* p) { return h2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<h16_t>(h16_t* p, float v) { *p = f2h(v); }

// 4-wide vector access (16 B fp32 / 8 B bf16); pointer must be aligned accordingly.
template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&o)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void ld4<h16_t>(const h16_t* p, float (&o)[4]) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    unpack_h16x2(v.x, o[0], o[1]);
    unpack_h16x2(v.y, o[2], o[3]);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&o)[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
template <> __device__ __forceinline__ void st4<h16_t>(h16_t* p, const float (&o)[4]) {
    uint2 v;
    v.x = pack_h16x2(o[0], o[1]);
    v.y = pack_h16x2(o[2], o[3]);
    *reinterpret_cast<uint2*>(p) = v;
}

// wave64 reductions
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__device__ __forceinline__ float act_fwd(float z, float slope) { return z > 0.f ? z : z * slope; }
__device__ __forceinline__ float act_grad(float z, float slope) { return z > 0.f ? 1.f : slope; }

// ---- BatchNorm-apply + activation of the PRODUCING layer applied by the consumer while it loads its input ------------------
// (kp_conv_fwd_bnin / kp_conv_wgrad_bnin: kp_conv_small.hip; kp_head_fwd_bnin / kp_head_bwd_bnin: kp_head_fast.hip, kp_head_mfma.hip).
// The arithmetic is kp_bn_act_fwd's, expression for expression: (y - mean) * invstd * gamma + beta, the activation, RNE to the
// 16-bit storage type -- what the consumer multiplies is bit for bit what the separate pass would have stored.
struct KpBnIn {
    const float *mean, *invstd, *gamma, *beta;   // mean == nullptr: no BatchNorm (batch_norm=False), activation only
    float slope;                                 // 0 <= slope <= 1 (checked by the launchers)
};
// act(z) = max(z, z * slope) is (z > 0 ? z : z * slope) bit for bit for 0 <= slope <= 1, zeros and NaNs included
__device__ __forceinline__ float kp_bnin1(float y, float mu, float is, float ga, float be, float slope) {
    const float z = (y - mu) * is * ga + be;
    return fmaxf(z, z * slope);
}
// one 16-byte chunk = 8 consecutive channels of a pixel, raw conv output -> stored activation: kp_bnin1 on the eight elements,
// written on channel PAIRS so that the four arithmetic steps are packed instructions (v_pk_add_f32 / v_pk_mul_f32: 28
// instead of 48 VALU per chunk; the same IEEE operation per element, no contraction: -ffp-contract=off)
__device__ __forceinline__ uint4 kp_bnin8(const uint4 raw, const float (&mu)[8], const float (&is)[8], const float (&ga)[8],
                                          const float (&be)[8], float slope) {
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
    uint32_t o[4];
    const kp_f32x2 sl = {slope, slope};
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float lo, hi;
        unpack_h16x2(w[p], lo, hi);
        kp_f32x2 v = {lo, hi};
        v = v - kp_f32x2{mu[2 * p], mu[2 * p + 1]};
        v = v * kp_f32x2{is[2 * p], is[2 * p + 1]};
        v = v * kp_f32x2{ga[2 * p], ga[2 * p + 1]};
        v = v + kp_f32x2{be[2 * p], be[2 * p + 1]};
        const kp_f32x2 t = v * sl;
        o[p] = pack_h16x2(fmaxf(v[0], t[0]), fmaxf(v[1], t[1]));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}
// the four constants of channels c0 .. c0 + 7 (without BatchNorm the EXACT identity (y - (+0)) * 1 * 1 + (-0))
struct KpBnIn8 {
    float mu[8], is[8], ga[8], be[8];
    __device__ __forceinline__ void load(const KpBnIn& bn, int c0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool aff = bn.mean != nullptr;
            mu[e] = aff ? bn.mean[c0 + e] : 0.f;
            is[e] = aff ? bn.invstd[c0 + e] : 1.f;
            ga[e] = aff ? bn.gamma[c0 + e] : 1.f;
            be[e] = aff ? bn.beta[c0 + e] : -0.f;
        }
    }
    // one 16-byte chunk = 8 consecutive channels of a pixel, raw conv output -> stored activation
    __device__ __forceinline__ uint4 apply(const uint4 raw, float slope) const {
        return kp_bnin8(raw, mu, is, ga, be, slope);
    }
};

// torch.linspace(0, 1, n)[i] in fp32, built the way ATen builds it (symmetric halves).
// exp(-sqrt(dd) * inv2s2): one Gaussian-like map value from dd = dy^2 + dx^2 + eps (functional.py:56-63), on the hardware's
// v_sqrt_f32 / v_exp_f32 (1 ulp each; the exponent's rounding adds |x| * 2^-24 relative, |x| <= 71 here: the value is within 1e-5
// relative and 1e-7 absolute of the IEEE expression, SURVEY 8c asks atol 2.5e-4).  The correctly rounded sqrtf + expf are ~40 VALU
// instructions per value and were most of the fused bottleneck's second phase; every kernel that renders the maps calls THIS, so the
// fused, the T and the piecewise forms stay bit-identical.
__device__ __forceinline__ float kp_fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float kp_fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ float kp_gauss(float dd, float inv2s2) { return kp_fast_exp(-kp_fast_sqrt(dd) * inv2s2); }
__device__ __forceinline__ float ruler_at(int i, int n) {
    if (n <= 1) return 0.f;
    float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

// ---- host side -------------------------------------------------------------
void kp_set_error(const char* fmt, ...);
// Process-wide measurement knobs (kp_config / kp_config_ptr, kp_api.hip): atomics, so that a thread that sets one while another
// launches is a defined program (relaxed cost on the host: a plain load at every read).  They select kernels for A/B
// measurements; no result depends on them.
typedef std::atomic<int> kp_knob_t;
typedef std::atomic<long long*> kp_knob_ptr_t;

int kp_check_launch(const char* what);
// Step-level byte accounting (kp_prof_enable(2), bench.py's step HBM fraction): every entry point adds the ALGORITHMIC HBM bytes of
// the launch it is about to make (operands + results, each once); `launch` = 0 adds bytes to the launch counted before.
void kp_count_bytes(double bytes, int launch = 1);
void kp_set_last_rows(long long rows);   // see kp_last_rows() in kp_hip.h
// kp_conv_fwd_fin (kp_api.hip): the BatchNorm finalize that follows a statistics-producing convolution, to be run in the TAIL of that
// convolution where its launcher can (the persistent patch kernel with one statistics row per workgroup: the last workgroup of each
// output-channel tile to arrive finalizes that tile's channels).  Thread-local request, set and cleared inside ONE public call; a
// launcher that takes it sets `taken` (the caller then launches kp_bn_finalize itself when nobody did).
struct KpFinRequest {
    float *mean, *invstd, *running_mean, *running_var;
    long long* nbt;
    float momentum, eps;
    long long count;
    unsigned* tickets;   // >= KP_FIN_TICKETS zero-initialised counters (zero again when the kernel ends)
    int taken;
};
KpFinRequest* kp_fin_request();
void kp_set_fin_request(KpFinRequest* r);
// Query mode (kp_conv_fwd_stats_rows / kp_bn_bwd_rows): the dispatch runs as usual, every launcher records its row count
// and returns before launching anything.  Thread-local, set and cleared inside ONE public call.
int kp_query_mode();
void kp_set_query_mode(int on);
#define KP_QUERY_RETURN()            \
    do {                             \
        if (kp_query_mode()) return KP_OK; \
    } while (0)

#define KP_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            kp_set_error(__VA_ARGS__);        \
            return KP_E_ARG;                  \
        }                                     \
    } while (0)

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
