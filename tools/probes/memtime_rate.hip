// Probe: how many s_memtime ticks does a back-to-back v_mfma_f32_32x32x16_bf16 stream take?  (32 shader cycles per
// MFMA per SIMD is the known issue rate; the ratio calibrates the s_memtime stamps of tools/probes/trace.py.)
//   hipcc --offload-arch=gfx950 -O3 -o memtime_rate memtime_rate.hip && ./memtime_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__global__ void k(long long* out, float* sink, int waves_per_simd) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(threadIdx.x + i); b[i] = (short)(threadIdx.x * 3 + i); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 256; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    long long* d; float* s; hipMalloc(&d, 4096 * 8); hipMalloc(&s, 4096 * 1024 * 4);
    for (int tpb : {64, 256, 512}) {
        for (int grid : {1, 256}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<grid, tpb>>>(d, s, 0); hipDeviceSynchronize();
            hipEventRecord(e0); k<<<grid, tpb>>>(d, s, 0); hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[4]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
            printf("tpb %d grid %d: %lld ticks for 1024 MFMA/wave = %.1f ticks/MFMA/wave; kernel %.1f us\n", tpb, grid, h[0], h[0] / 1024.0, ms * 1e3);
        }
    }
    return 0;
}
