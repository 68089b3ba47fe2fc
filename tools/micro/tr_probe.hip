// tr_probe.hip -- empirical semantics of ds_read_b64_tr_b16 (gfx950): LDS holds u16 element indices (value = byte address / 2);
// every lane supplies a byte address, the instruction returns 4 x u16 per lane.  Prints, per address pattern, what each lane got.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__global__ void probe(uint32_t* out, int pattern) {
    __shared__ uint16_t lds[16384];
    for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int lane = threadIdx.x;
    uint32_t addr;
    if (pattern == 0) addr = 0;                                          // uniform
    else if (pattern == 1) addr = lane * 8;                              // lane-linear, 8 B per lane
    else if (pattern == 2) addr = (lane & 15) * 256 + (lane >> 4) * 8;   // 16 rows of 256 B, 4 column groups of 8 B
    else addr = (lane & 15) * 64 + (lane >> 4) * 1024;                   // rows of 64 B; lane groups 1 KiB apart
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds;
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr + base) : "memory");
    out[lane * 2] = v.x;
    out[lane * 2 + 1] = v.y;
}
int main() {
    uint32_t* d;
    hipMalloc(&d, 64 * 2 * 4);
    std::vector<uint32_t> h(128);
    for (int p = 0; p < 4; ++p) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, p);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; ++l)
            printf("  lane %2d: %5u %5u %5u %5u\n", l, h[2 * l] & 0xffff, h[2 * l] >> 16, h[2 * l + 1] & 0xffff, h[2 * l + 1] >> 16);
    }
    return 0;
}
