// Shared device helpers for libkp_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/kp_hip.h"

typedef uint16_t bf16_t;   // raw bfloat16 bits

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32), which hipcc
// selects for __bf16 casts -- one instruction per PAIR instead of ~8 VALU ops per value.
typedef __bf16 kp_bf16x2 __attribute__((ext_vector_type(2)));
typedef float kp_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const kp_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, kp_bf16x2));
}

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<bf16_t>(const bf16_t* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }

// 4-wide vector access (16 B fp32 / 8 B bf16); pointer must be aligned accordingly.
template <typename T> __device__ __forceinline__ void ld4(const T* p, float (&o)[4]);
template <> __device__ __forceinline__ void ld4<float>(const float* p, float (&o)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void ld4<bf16_t>(const bf16_t* p, float (&o)[4]) {
    uint2 v = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
}
template <typename T> __device__ __forceinline__ void st4(T* p, const float (&o)[4]);
template <> __device__ __forceinline__ void st4<float>(float* p, const float (&o)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, const float (&o)[4]) {
    uint2 v;
    v.x = pack_bf16x2(o[0], o[1]);
    v.y = pack_bf16x2(o[2], o[3]);
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

// torch.linspace(0, 1, n)[i] in fp32, built the way ATen builds it (symmetric halves).
__device__ __forceinline__ float ruler_at(int i, int n) {
    if (n <= 1) return 0.f;
    float step = 1.0f / (float)(n - 1);
    return (i < n / 2) ? step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

// ---- host side -------------------------------------------------------------
void kp_set_error(const char* fmt, ...);
int kp_check_launch(const char* what);
void kp_set_last_rows(long long rows);   // see kp_last_rows() in kp_hip.h

#define KP_REQUIRE(cond, ...)                 \
    do {                                      \
        if (!(cond)) {                        \
            kp_set_error(__VA_ARGS__);        \
            return KP_E_ARG;                  \
        }                                     \
    } while (0)

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }
