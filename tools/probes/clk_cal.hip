// which clock does __builtin_readcyclecounter() (s_memtime) count on this part?  hipcc --offload-arch=gfx950 clk_cal.hip -o clk_cal && ./clk_cal
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    unsigned long long t;
    float x = threadIdx.x;
    do { for (int i = 0; i < 64; ++i) x = x * 1.0001f + 0.5f; t = __builtin_readcyclecounter(); } while (t - t0 < ticks);
    if (threadIdx.x == 0) { out[0] = t - t0; out[1] = wall_clock64() - w0; out[2] = (unsigned long long)x; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int busy = 0; busy < 2; ++busy) {
        spin<<<busy ? 2048 : 1, 256>>>(1000000, d); hipDeviceSynchronize();
        hipEventRecord(a); spin<<<busy ? 2048 : 1, 256>>>(10000000, d); hipEventRecord(b); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, a, b);
        unsigned long long h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("grid %d: %llu cycle-counter ticks, %llu wall_clock64 ticks in %.3f ms -> %.1f MHz, wall %.1f MHz\n", busy ? 2048 : 1, h[0], h[1], ms, h[0] / ms / 1e3, h[1] / ms / 1e3);
    }
}
