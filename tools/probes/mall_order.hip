// Probe: does the 256 MiB Infinity Cache serve the tail of a just-written (or just-read) buffer to the next
// streaming kernel?  A writer fills X (S bytes) front to back; a reader then sweeps it front-to-back or
// back-to-front.  If the back-to-front sweep is faster for S > 256 MiB, HBM-bound passes that follow a producer
// should walk its output in the reverse of the producer's order.
//   hipcc --offload-arch=gfx950 -O3 -o mall_order mall_order.hip && ./mall_order
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int CH = 4096;   // uint4 per block chunk (64 KiB)
__global__ __launch_bounds__(256) void write_k(uint4* x, size_t nchunks) {
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        uint4* p = x + c * CH;
#pragma unroll 4
        for (int i = threadIdx.x; i < CH; i += 256) p[i] = make_uint4((unsigned)c, i, 3, 4);
    }
}
template <bool REV, bool WR>
__global__ __launch_bounds__(256) void read_k(const uint4* x, uint4* out, size_t nchunks) {
    unsigned acc = 0;
    for (size_t c0 = blockIdx.x; c0 < nchunks; c0 += gridDim.x) {
        const size_t c = REV ? nchunks - 1 - c0 : c0;
        const uint4* p = x + c * CH;
        uint4 v[4];
        for (int i = threadIdx.x; i < CH; i += 1024) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p[i + u * 256];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc += v[u].x ^ v[u].w;
                if (WR) out[c * CH + i + u * 256] = v[u];
            }
        }
    }
    if (acc == 0x12345679u) out[0] = make_uint4(acc, 0, 0, 0);
}
int main() {
    uint4 *x, *y, *junk;
    const size_t maxb = (size_t)1 << 30;
    hipMalloc(&x, maxb); hipMalloc(&y, maxb); hipMalloc(&junk, maxb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 2048;
    for (size_t mb : {64, 128, 192, 256, 384, 512, 768, 1024}) {
        const size_t nch = mb * 1024 * 1024 / (CH * 16);
        float t[6] = {0, 0, 0, 0, 0, 0};
        const int reps = 5;
        for (int r = 0; r < reps; ++r) {
            float ms;
            // 0: write then forward read; 1: write then reverse read
            write_k<<<grid, 256>>>(x, nch);
            hipEventRecord(e0); read_k<false, false><<<grid, 256>>>(x, y, nch); hipEventRecord(e1); hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1); t[0] += ms;
            write_k<<<grid, 256>>>(x, nch);
            hipEventRecord(e0); read_k<true, false><<<grid, 256>>>(x, y, nch); hipEventRecord(e1); hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1); t[1] += ms;
            // 2: cold (1 GiB of other traffic in between) forward read
            write_k<<<grid, 256>>>(junk, maxb / (CH * 16));
            hipEventRecord(e0); read_k<false, false><<<grid, 256>>>(x, y, nch); hipEventRecord(e1); hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1); t[2] += ms;
            // 3: forward read then forward read again; 4: forward read then reverse read
            read_k<false, false><<<grid, 256>>>(x, y, nch);
            hipEventRecord(e0); read_k<false, false><<<grid, 256>>>(x, y, nch); hipEventRecord(e1); hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1); t[3] += ms;
            read_k<false, false><<<grid, 256>>>(x, y, nch);
            hipEventRecord(e0); read_k<true, false><<<grid, 256>>>(x, y, nch); hipEventRecord(e1); hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1); t[4] += ms;
            // 5: write x, then reverse copy x -> y (read + write)
            write_k<<<grid, 256>>>(x, nch);
            hipEventRecord(e0); read_k<true, true><<<grid, 256>>>(x, y, nch); hipEventRecord(e1); hipDeviceSynchronize();
            hipEventElapsedTime(&ms, e0, e1); t[5] += ms;
        }
        const double gb = (double)mb * 1.048576e-3;
        printf("%5zu MiB  w->fwd %6.0f  w->rev %6.0f  cold %6.0f  r->fwd %6.0f  r->rev %6.0f GB/s   w->rev copy %6.0f GB/s (r+w)\n", mb,
               gb / (t[0] / reps) * 1e3, gb / (t[1] / reps) * 1e3, gb / (t[2] / reps) * 1e3, gb / (t[3] / reps) * 1e3,
               gb / (t[4] / reps) * 1e3, 2 * gb / (t[5] / reps) * 1e3);
    }
    return 0;
}
