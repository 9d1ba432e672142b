// Row sums of a heat-map by a wavefront (spacial_softmax / spacial_logsoftmax marginals, keypoints/models/functional.py:27-44):
// eight rows at a time, the lane's partials of the eight rows folded over the 64 lanes by a halving butterfly on VALU cross-lane
// operations only (v_permlane32_swap, v_permlane16_swap, DPP) -- a ds_bpermute butterfly per row (wave_sum) costs six dependent
// LDS-crossbar round trips per row.  Every row goes through the same pairing tree (lane ^ 32, ^ 16, ^ 8, ^ 7, ^ 2, ^ 1) whatever its
// position in the batch, and the result is taken from the lanes with (lane & 7) == 0: the fused kernels (kp_bottleneck_dev.h) and
// kp_spatial_softmax_fwd (kp_bottleneck.hip) share this function and stay bit-identical to each other.
#pragma once
#include "kp_common.h"

namespace {

template <int CTRL> __device__ __forceinline__ float kp_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// 8 per-lane values -> each lane gets the sum over all 64 lanes of ONE of them (index returned in idx).  All lanes active.
__device__ __forceinline__ float kp_rows8_over64(const float (&v)[8], int lane, int& idx) {
    const bool b5 = lane & 32, b4 = lane & 16, b3 = lane & 8;
    float a[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // halves of 32 lanes: upper half of v[i] <-> lower half of v[i + 4]
        float lo = v[i], hi = v[i + 4];
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        a[i] = lo + hi;
    }
    float b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {   // rows of 16 lanes: odd rows of a[i] <-> even rows of a[i + 2]
        float lo = a[i], hi = a[i + 2];
        asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        b[i] = lo + hi;
    }
    float d = (b3 ? b[1] : b[0]) + kp_dpp<0x128>(b3 ? b[0] : b[1]);   // partner = lane ^ 8 (row_ror:8)
    d += kp_dpp<0x141>(d);    // + lane ^ 7 (row_half_mirror)
    d += kp_dpp<0x4E>(d);     // + lane ^ 2
    d += kp_dpp<0xB1>(d);     // + lane ^ 1
    idx = (b5 ? 4 : 0) | (b4 ? 2 : 0) | (b3 ? 1 : 0);
    return d;
}

}  // namespace
