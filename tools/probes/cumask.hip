// Probe: CU-masked streams (hipExtStreamCreateWithCUMask) on MI355X.
//   1. which physical CUs (XCC, SE, CU) does mask bit i select?  (census kernel reading HW_ID / XCC_ID)
//   2. what does a streaming copy reach from a subset of the CUs?
//   3. does a copy on 64 CUs run next to an MFMA-heavy kernel on the other 192 without slowing it?
//   hipcc --offload-arch=gfx950 -O3 -o cumask cumask.hip && ./cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <set>
#include <map>

__global__ void fill_k(uint4* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u + 12345u;
        uint4 v; x ^= x >> 13; x *= 0x5bd1e995u; v.x = (x & 0x807f807fu) | 0x3f003f00u; x ^= x >> 15; x *= 0x27d4eb2du;
        v.y = (x & 0x807f807fu) | 0x3f003f00u; x ^= x >> 13; x *= 0x5bd1e995u; v.z = (x & 0x807f807fu) | 0x3e803f00u; x ^= x >> 15;
        v.w = (x & 0x807f807fu) | 0x3f003e80u;
        a[i] = v;
    }
}

__global__ void census_k(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = hw; out[blockIdx.x * 2 + 1] = xcc; }
    // stay resident for a while so that the blocks spread over every allowed CU
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}

__global__ __launch_bounds__(256) void copy_k(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i0 = (size_t)blockIdx.x * 1024 + threadIdx.x; i0 < n; i0 += stride) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u * 256 < n) v[u] = a[i0 + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u * 256 < n) b[i0 + u * 256] = v[u];
    }
}

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
// MFMA-bound stand-in for the persistent conv kernels: 8 waves per workgroup, one workgroup per CU (LDS-limited),
// operands from random register contents
__global__ __launch_bounds__(512, 2) void mfma_k(const uint4* __restrict__ seed, float* out, int iters) {
    __shared__ unsigned char pad[140 * 1024];
    uint4 s0 = seed[threadIdx.x], s1 = seed[threadIdx.x + 512];
    bf16x8 a0, b0, a1, b1;
    __builtin_memcpy(&a0, &s0, 16); __builtin_memcpy(&b0, &s1, 16);
    s0.x ^= 0x55aa1234u; s1.y += 77;
    __builtin_memcpy(&a1, &s0, 16); __builtin_memcpy(&b1, &s1, 16);
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c3, 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += c0[i] + c1[i] + c2[i] + c3[i];
    if (r == 123.456f) { out[0] = r; pad[threadIdx.x] = 1; }
}

static hipStream_t masked(const std::vector<int>& bits) {
    uint32_t m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b : bits) m[b >> 5] |= 1u << (b & 31);
    hipStream_t s;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m);
    if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e)); exit(1); }
    return s;
}

static void census(const char* name, hipStream_t s, unsigned* dbuf) {
    const int nb = 4096;
    census_k<<<nb, 64, 0, s>>>(dbuf, 200000);
    hipStreamSynchronize(s);
    std::vector<unsigned> h(nb * 2);
    hipMemcpy(h.data(), dbuf, nb * 8, hipMemcpyDeviceToHost);
    std::map<int, std::set<int>> per_xcc;
    for (int i = 0; i < nb; ++i) {
        const unsigned hw = h[i * 2], xcc = h[i * 2 + 1] & 15;
        const int cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc].insert((se << 5) | (sh << 4) | cu);
    }
    int tot = 0;
    printf("%-28s:", name);
    for (auto& kv : per_xcc) { printf(" xcc%d=%zu", kv.first, kv.second.size()); tot += (int)kv.second.size(); }
    printf("  total %d CUs\n", tot);
}

int main() {
    unsigned* dbuf; hipMalloc(&dbuf, 1 << 20);
    hipStream_t full; hipStreamCreate(&full);
    census("unmasked", full, dbuf);
    { std::vector<int> b; for (int i = 0; i < 64; ++i) b.push_back(i); census("bits 0..63", masked(b), dbuf); }
    { std::vector<int> b; for (int i = 0; i < 256; i += 4) b.push_back(i); census("every 4th bit", masked(b), dbuf); }
    { std::vector<int> b; for (int i = 0; i < 8; ++i) b.push_back(i); census("bits 0..7", masked(b), dbuf); }
    { std::vector<int> b; for (int i = 0; i < 32; ++i) b.push_back(i); census("bits 0..31", masked(b), dbuf); }
    { std::vector<int> b; for (int i = 192; i < 256; ++i) b.push_back(i); census("bits 192..255", masked(b), dbuf); }

    // ---- 2. copy bandwidth from subsets
    const size_t n = (size_t)512 * 1024 * 1024 / 16;
    uint4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); fill_k<<<4096, 256>>>(a, n); hipDeviceSynchronize();
    hipEvent_t e0, e1, f0, f1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
    auto time_copy = [&](hipStream_t s, int grid) {
        copy_k<<<grid, 256, 0, s>>>(a, b, n);
        hipEventRecord(e0, s);
        for (int r = 0; r < 5; ++r) copy_k<<<grid, 256, 0, s>>>(a, b, n);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 5;
    };
    std::vector<int> lo, hi, lo32, hi224;
    // (bit i selects CU i / 8 of XCC i % 8; a mask that leaves an XCC without CUs is ignored altogether)
    for (int i = 0; i < 256; ++i) { (i < 64 ? lo : hi).push_back(i); (i < 32 ? lo32 : hi224).push_back(i); }
    hipStream_t s64 = masked(lo), s192 = masked(hi), s32 = masked(lo32), s224 = masked(hi224);
    census("mask bits 0..63 [64]", s64, dbuf);
    census("mask bits 64..255 [192]", s192, dbuf);
    census("mask bits 0..31 [32]", s32, dbuf);
    for (int grid : {2048, 8192}) {
        printf("copy 512 MiB grid %5d: full %6.0f  192 CUs %6.0f  64 CUs %6.0f  32 CUs %6.0f GB/s\n", grid,
               2.0 * n * 16 / time_copy(full, grid) / 1e6, 2.0 * n * 16 / time_copy(s192, grid) / 1e6,
               2.0 * n * 16 / time_copy(s64, grid) / 1e6, 2.0 * n * 16 / time_copy(s32, grid) / 1e6);
    }
    // ---- 3. MFMA kernel alone (256 / 224 / 192 CUs) and next to a copy on the complementary CUs
    float* out; hipMalloc(&out, 64);
    const int iters = 40000;
    auto time_mfma = [&](hipStream_t s, int grid) {
        mfma_k<<<grid, 512, 0, s>>>(a, out, 2000);
        hipEventRecord(e0, s);
        mfma_k<<<grid, 512, 0, s>>>(a, out, iters);
        hipEventRecord(e1, s); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
    };
    const double fl = 4.0 * 2 * 32 * 32 * 16 * 8 * (double)iters;   // per workgroup
    float t256 = time_mfma(full, 256), t192 = time_mfma(s192, 192), t224 = time_mfma(s224, 224);
    printf("mfma alone: 256 CUs %.3f ms (%.0f TF/s)   224 CUs %.3f ms (%.0f TF/s)   192 CUs %.3f ms (%.0f TF/s)\n", t256,
           fl * 256 / t256 / 1e9, t224, fl * 224 / t224 / 1e9, t192, fl * 192 / t192 / 1e9);
    auto both = [&](hipStream_t sm, int gm, hipStream_t sc, const char* name) {
        hipDeviceSynchronize();
        hipEventRecord(e0, sm); hipEventRecord(f0, sc);
        mfma_k<<<gm, 512, 0, sm>>>(a, out, iters);
        int ncopy = 0;
        // keep the copy stream busy for about as long as the MFMA kernel runs alone
        for (; ncopy < 12; ++ncopy) copy_k<<<4096, 256, 0, sc>>>(a, b, n);
        hipEventRecord(e1, sm); hipEventRecord(f1, sc);
        hipDeviceSynchronize();
        float tm, tc; hipEventElapsedTime(&tm, e0, e1); hipEventElapsedTime(&tc, f0, f1);
        printf("%-34s mfma %.3f ms (%.0f TF/s)   %d copies %.3f ms (%.0f GB/s while running)\n", name, tm, fl * gm / tm / 1e9, ncopy,
               tc, 2.0 * n * 16 * ncopy / tc / 1e6);
    };
    both(s192, 192, s64, "mfma@192 || copy@64:");
    both(s224, 224, s32, "mfma@224 || copy@32:");
    both(full, 256, full == nullptr ? full : s64, "mfma@256(unmasked) || copy@64:");
    return 0;
}
