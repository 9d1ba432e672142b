// Probe: what do plain streaming kernels reach on this box?  (the ceiling the BatchNorm family is compared with)
//   hipcc --offload-arch=gfx950 -O3 -o hbm_ceiling hbm_ceiling.hip && ./hbm_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NR, int NW, int U>
__global__ __launch_bounds__(256) void stream_k(const uint4* __restrict__ a, const uint4* __restrict__ b, uint4* __restrict__ c,
                                                uint4* __restrict__ d, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i0 = (size_t)blockIdx.x * 256 * U + threadIdx.x; i0 < n; i0 += stride) {
        uint4 va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + (size_t)u * 256;
            if (i < n) { va[u] = a[i]; if (NR > 1) vb[u] = b[i]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t i = i0 + (size_t)u * 256;
            if (i < n) {
                uint4 r = va[u];
                if (NR > 1) { r.x ^= vb[u].x; r.y += vb[u].y; r.z ^= vb[u].z; r.w += vb[u].w; }
                if (NW >= 1) c[i] = r;
                if (NW >= 2) d[i] = r;
                if (NW == 0 && r.x == 0x12345678u) c[0] = r;
            }
        }
    }
}
template <int NR, int NW, int U> void run(const char* name, uint4* a, uint4* b, uint4* c, uint4* d, size_t n, int grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    stream_k<NR, NW, U><<<grid, 256>>>(a, b, c, d, n);
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) stream_k<NR, NW, U><<<grid, 256>>>(a, b, c, d, n);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-26s grid %6d: %7.1f us  %6.0f GB/s\n", name, grid, ms * 1e3, (double)(NR + NW) * n * 16 / ms / 1e6);
}
int main() {
    const size_t n = (size_t)128 * 128 * 128 * 128 * 2 / 16;   // one 128-channel 128x128 bf16 activation of 128 images (537 MB)
    uint4 *a, *b, *c, *d;
    hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16); hipMalloc(&d, n * 16);
    hipMemset(a, 1, n * 16); hipMemset(b, 2, n * 16);
    for (int grid : {1024, 2048, 4096, 16384}) {
        run<1, 0, 4>("read", a, b, c, d, n, grid);
        run<1, 1, 4>("copy (1r 1w)", a, b, c, d, n, grid);
        run<2, 0, 4>("2 reads", a, b, c, d, n, grid);
        run<2, 1, 4>("triad (2r 1w)", a, b, c, d, n, grid);
        run<2, 1, 2>("triad (2r 1w) U2", a, b, c, d, n, grid);
    }
    return 0;
}
