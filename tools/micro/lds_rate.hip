// lds_rate.hip -- ds_read_b128 / ds_read_b64 throughput per CU on gfx950 (conflict-free 64-B-row image as in the bf16 tile loops).
// Build: hipcc --offload-arch=gfx950 -O3 lds_rate.hip -o lds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t* out, int iters) {
    __shared__ __attribute__((aligned(16))) char sm[48 * 1024];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
    for (int i = tid; i < 48 * 1024 / 4; i += 256) reinterpret_cast<uint32_t*>(sm)[i] = i * 2654435761u;
    __syncthreads();
    const int r = (tid >> 6) * 32 + l32;
    const uint32_t base = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)sm + r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[j]) : "v"(base), "i"(j * 4096));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j];
        } else {
            u32x2 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v[j]) : "v"(base), "i"(j * 2048));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < 16; ++j) { acc.x ^= v[j].x; acc.y ^= v[j].y; }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[tid] = acc.x;
}
int main() {
    uint32_t* out;
    hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode)
        for (int wgs_per_cu = 1; wgs_per_cu <= 3; ++wgs_per_cu) {
            const int grid = 256 * wgs_per_cu;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, out, iters);
                else hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    const double bytes = (double)grid * 256 * iters * 128.0;   // 8 x 16 B or 16 x 8 B per thread and iteration
                    printf("%s  %d WG/CU: %.3f ms  %.1f TB/s chip = %.1f B/clk/CU at 2.4 GHz\n", mode == 0 ? "ds_read_b128" : "ds_read_b64 ",
                           wgs_per_cu, ms, bytes / ms / 1e9, bytes / ms / 1e-3 / 256 / 2.4e9);
                }
            }
        }
    return 0;
}
