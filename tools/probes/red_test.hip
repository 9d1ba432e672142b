#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    float v = (float)lane;
    out[0 * 64 + lane] = dpp_mov<0x128>(v);
    out[1 * 64 + lane] = dpp_mov<0x141>(v);
    out[2 * 64 + lane] = dpp_mov<0x4E>(v);
    out[3 * 64 + lane] = dpp_mov<0xB1>(v);
    float a = (float)lane, b = 100.f + lane;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    out[4 * 64 + lane] = a;
    out[5 * 64 + lane] = b;
}
int main() {
    float* d; hipMalloc(&d, 6 * 64 * 4);
    k<<<1, 64>>>(d);
    float h[6 * 64]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[6] = {"ror8", "hmirror", "qp4E", "qpB1", "swap r0", "swap r1"};
    for (int i = 0; i < 6; ++i) { printf("%-8s:", names[i]); for (int l = 0; l < 64; ++l) printf(" %g", h[i * 64 + l]); printf("\n"); }
    return 0;
}
