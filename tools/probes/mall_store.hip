// Probe: which store flavour leaves NO dirty lines behind in the Infinity Cache?  A reader that follows a large plain
// write runs at ~3.5 TB/s instead of ~5.9 TB/s (tools/probes/mall_order.hip: it pays for the write-back of up to 256 MiB of
// dirty lines).  Here the writer uses plain / nt / sc1 / sc0 sc1 / nt sc0 sc1 stores and the reader then sweeps ANOTHER
// (cold) buffer: a fast read means the writer's data already reached HBM while the writer ran.
//   hipcc --offload-arch=gfx950 -O3 -o mall_store mall_store.hip && ./mall_store
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int CH = 4096;   // uint4 per block chunk (64 KiB)
template <int F>
__global__ __launch_bounds__(256) void write_k(uint4* x, size_t nchunks) {
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        uint4* p = x + c * CH;
#pragma unroll 4
        for (int i = threadIdx.x; i < CH; i += 256) {
            const uint4 v0 = make_uint4((unsigned)c, i, 3, 4);
            u32x4 v; v.x = v0.x; v.y = v0.y; v.z = v0.z; v.w = v0.w;
            uint4* q = p + i;
            if (F == 0) *q = v0;
            else if (F == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(q), "v"(v) : "memory");
            else if (F == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
            else if (F == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(q), "v"(v) : "memory");
            else if (F == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(q), "v"(v) : "memory");
            else if (F == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(q), "v"(v) : "memory");
            else if (F == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(q), "v"(v) : "memory");
        }
    }
}
template <int F>
__global__ __launch_bounds__(256) void read_k(const uint4* x, uint4* out, size_t nchunks) {
    unsigned acc = 0;
    for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint4* p = x + c * CH;
        u32x4 v[4];
        for (int i = threadIdx.x; i < CH; i += 1024) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint4* q = p + i + u * 256;
                if (F == 0) { const uint4 t = *q; v[u].x = t.x; v[u].y = t.y; v[u].z = t.z; v[u].w = t.w; }
                else asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[u]) : "v"(q) : "memory");
            }
            if (F) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].w;
        }
    }
    if (acc == 0x12345679u) out[0] = make_uint4(acc, 0, 0, 0);
}
int main() {
    uint4 *x, *y, *junk;
    const size_t maxb = (size_t)1 << 30;
    hipMalloc(&x, maxb); hipMalloc(&y, maxb); hipMalloc(&junk, maxb);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    const int grid = 2048;
    const size_t nj = maxb / (CH * 16);
    const char* names[7] = {"plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc0 nt", "sc1 nt"};
    for (size_t mb : {256, 512}) {
        const size_t nch = mb * 1024 * 1024 / (CH * 16);
        read_k<0><<<grid, 256>>>(x, y, nch);
        for (int f = 0; f < 7; ++f) {
            float tw = 0, tr = 0, trr = 0;
            const int reps = 4;
            for (int r = 0; r < reps; ++r) {
                float ms;
                hipEventRecord(e0);
                switch (f) {
                    case 0: write_k<0><<<grid, 256>>>(junk, nj); break;
                    case 1: write_k<1><<<grid, 256>>>(junk, nj); break;
                    case 2: write_k<2><<<grid, 256>>>(junk, nj); break;
                    case 3: write_k<3><<<grid, 256>>>(junk, nj); break;
                    case 4: write_k<4><<<grid, 256>>>(junk, nj); break;
                    case 5: write_k<5><<<grid, 256>>>(junk, nj); break;
                    case 6: write_k<6><<<grid, 256>>>(junk, nj); break;
                }
                hipEventRecord(e1);
                read_k<0><<<grid, 256>>>(x, y, nch);
                hipEventRecord(e2); hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1); tw += ms;
                hipEventElapsedTime(&ms, e1, e2); tr += ms;
                // and: does the written data itself still hit?  read the END of junk (last mb MiB)
                hipEventRecord(e0);
                read_k<0><<<grid, 256>>>(junk + (nj - nch) * CH, y, nch);
                hipEventRecord(e1); hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1); trr += ms;
            }
            const double gb = (double)mb * 1.048576e-3;
            printf("%4zu MiB reader | writer %-11s: write 1 GiB at %6.0f GB/s, then cold read %6.0f GB/s, then read of the written tail %6.0f GB/s\n",
                   mb, names[f], 1.0737 / (tw / reps) * 1e3, gb / (tr / reps) * 1e3, gb / (trr / reps) * 1e3);
        }
    }
    // nt loads: do they keep the reader from displacing what is cached?  x (256 MiB) resident, sweep junk with plain / nt, re-read x
    {
        const size_t nch = (size_t)192 * 1024 * 1024 / (CH * 16);
        for (int f = 0; f < 2; ++f) {
            float t = 0; const int reps = 4;
            for (int r = 0; r < reps; ++r) {
                float ms;
                read_k<0><<<grid, 256>>>(x, y, nch);
                read_k<0><<<grid, 256>>>(x, y, nch);
                if (f == 0) read_k<0><<<grid, 256>>>(junk, y, nj); else read_k<1><<<grid, 256>>>(junk, y, nj);
                hipEventRecord(e0);
                read_k<0><<<grid, 256>>>(x, y, nch);
                hipEventRecord(e1); hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1); t += ms;
            }
            printf("192 MiB resident, 1 GiB swept with %-5s loads, re-read: %6.0f GB/s\n", f ? "nt" : "plain", 192 * 1.048576e-3 / (t / reps) * 1e3);
        }
    }
    return 0;
}
