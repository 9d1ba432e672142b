// Device helpers shared by the MFMA kernels (gfx950): LDS-DMA, cross-lane reductions on VALU permutes, and the
// transpose-read fragment addressing.  Everything is internal to a translation unit (anonymous namespace).
#pragma once
#include <type_traits>
#include "kp_common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) short h16x8;
typedef __attribute__((ext_vector_type(4))) short h16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define LDS_AS __attribute__((address_space(3)))

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// 16 B per lane, global/buffer -> LDS without passing through registers; LDS destination = wave-uniform base + lane*16
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, void* lds_wave_base, int voff, int soff = 0) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// The same instruction as inline asm, for kernels that interleave DMA issues with ds_read_b64_tr_b16: hipcc orders
// every transpose read after any builtin LDS-DMA that might alias it (s_waitcnt vmcnt(0) after each DMA), which
// serialises the DMA latencies; the asm form is invisible to that analysis, the kernel keeps its own vmcnt wait + barrier
// before it reads a DMA'd buffer.  A kernel that uses this must not use the builtin form (M0 is written behind the
// compiler's back).
__device__ __forceinline__ u32x4 rsrc_words(const void* base) {
    const unsigned long long b = (unsigned long long)base;
    u32x4 d;
    d.x = __builtin_amdgcn_readfirstlane((unsigned)b);
    d.y = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    d.z = 0x7ffffffeu;
    d.w = 0x00020000u;
    return d;
}
__device__ __forceinline__ void dma16_asm(u32x4 rsrc, const void* lds_wave_base, int voff) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(LDS_AS const unsigned char*)lds_wave_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc));
}

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// Halving butterfly: 16 per-lane values -> each lane ends with the sum over its 32-lane half of ONE value
// (index returned in `reg`).  All exchanges are VALU cross-lane ops (v_permlane16_swap, DPP row_ror / half-mirror /
// quad_perm) -- no ds_bpermute round trips through the LDS.
// First stage on its own (16 -> 8 values per lane, summed over the two 16-lane rows of each 32-lane half): what
// persistent workgroups accumulate per tile; the remaining stages then run once per workgroup.
__device__ __forceinline__ void reduce16_stage1(const float (&v)[16], float (&a)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // rows of 16 lanes: odd rows of v[i] <-> even rows of v[i+8]
        // (inline asm: the builtin's second result comes back equal to the first with this hipcc; the s_nop covers
        //  the VALU-write -> permlane-read hazard that the compiler does not pad inside asm)
        float lo = v[i], hi = v[i + 8];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        a[i] = lo + hi;
    }
}
__device__ __forceinline__ float reduce8_rest(const float (&a)[8], int lane, int& reg) {
    const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
    float b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // partner = lane ^ 8 (row_ror:8)
        const float keep = b3 ? a[i + 4] : a[i], send = b3 ? a[i] : a[i + 4];
        b[i] = keep + dpp_mov<0x128>(send);
    }
    float c[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // partner = lane ^ 7 (row_half_mirror): opposite bit 2
        const float keep = b2 ? b[i + 2] : b[i], send = b2 ? b[i] : b[i + 2];
        c[i] = keep + dpp_mov<0x141>(send);
    }
    float d = (b1 ? c[1] : c[0]) + dpp_mov<0x4E>(b1 ? c[0] : c[1]);   // partner = lane ^ 2
    d += dpp_mov<0xB1>(d);                                              // partner = lane ^ 1
    reg = (b4 ? 8 : 0) | (b3 ? 4 : 0) | (b2 ? 2 : 0) | (b1 ? 1 : 0);
    return d;
}
__device__ __forceinline__ float reduce16_over32(const float (&v)[16], int lane, int& reg) {
    float a[8];
    reduce16_stage1(v, a);
    return reduce8_rest(a, lane, reg);
}

// ---- transpose-read (ds_read_b64_tr_b16) fragments of a pixel-major LDS tile --------------------------------
template <int P>   // row pitch in bytes (128 or 256): swizzle of the 64-B block index by the LDS row
__device__ __forceinline__ int blk_swz(int row) { return P == 256 ? (row & 3) : ((row >> 1) & 1); }

// A fragment = 8 consecutive LDS rows (pixels) R..R+7 of this lane's channel, read as two 4-row
// ds_read_b64_tr_b16.  The lane's row is R + lrow (+4 for the second read) with R a compile-time constant after
// unrolling; the XOR swizzle of the 64-byte block depends on the row only through (R mod 4) and lrow, so the per-lane
// part is precomputed once for the 4 residues (tr_base) and R*P + 4*P*u folds into the instruction's immediate
// offset: no address arithmetic next to the MFMAs.
template <int P>
__device__ __forceinline__ int tr_base(int v, int lrow, int colbyte) {
    const int row = v + lrow;
    return lrow * P + ((((colbyte >> 6) ^ blk_swz<P>(row)) << 6) | (colbyte & 63));
}
template <int P, int R>
__device__ __forceinline__ h16x8 tr_frag(const unsigned char* tile, const int (&base)[4]) {
    h16x8 r;
    const unsigned char* a = tile + base[R & 3];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        h16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((LDS_AS h16x4*)(a + (R + 4 * u) * P));
        r[4 * u + 0] = v[0]; r[4 * u + 1] = v[1]; r[4 * u + 2] = v[2]; r[4 * u + 3] = v[3];
    }
    return r;
}

}  // namespace
