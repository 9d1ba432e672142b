// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  LDS holds u16 value == its element index.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint32_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    int l = threadIdx.x;
    // mode 0: lane-linear addresses (lane l -> elements 4l..4l+3)
    // mode 1: row pitch 64 elements: lane t in 16-group: row t/4, col chunk t%4  (+ group*16 cols)
    uint32_t addr;
    if (mode == 0) addr = l * 8;
    else { int t = l & 15, g = l >> 4; addr = ((t >> 2) * 64 + g * 16 + (t & 3) * 4) * 2; }
    addr += (uint32_t)(uintptr_t)lds;   // LDS aperture low bits: generic->local
    uint32_t lo, hi;
    uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane(0) + (uint32_t)((size_t)((__attribute__((address_space(3))) uint16_t*)lds)) ;
    if (mode == 0) a += l * 8; else { int t = l & 15, g = l >> 4; a += ((t >> 2) * 64 + g * 16 + (t & 3) * 4) * 2; }
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(*(uint64_t*)&lo) : "v"(a) : "memory");
    (void)hi; (void)addr;
}
__global__ void probe2(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    int l = threadIdx.x;
    uint32_t a = (uint32_t)((size_t)((__attribute__((address_space(3))) uint16_t*)lds));
    if (mode == 0) a += l * 8; else { int t = l & 15, g = l >> 4; a += ((t >> 2) * 64 + g * 16 + (t & 3) * 4) * 2; }
    uint64_t v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)(v >> (16 * j));
}
int main() {
    uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
    uint16_t h[256];
    for (int mode = 0; mode < 2; ++mode) {
        probe2<<<1, 64>>>(d, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
