// What the HBM of THIS box sustains for the access mixes of the step's bandwidth-bound passes: read-only, write-only, copy (1:1),
// 2 reads : 1 write (LayerNorm backward), float4 grid-stride with 8 loads in flight per lane, 2-GiB buffers.
// hipcc --offload-arch=gfx950 -O3 hbm_rate.hip -o hbm_rate && ./hbm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>   // 0 read, 1 write, 2 copy, 3 two reads + one write
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ o, long n4, float* sink) {
    const long stride = (long)gridDim.x * 256;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += 8 * stride) {
        f32x4 v[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long j = i + u * stride;
            if (MODE != 1) v[u] = j < n4 ? __builtin_nontemporal_load(a + j) : acc;
            if (MODE == 3) w[u] = j < n4 ? __builtin_nontemporal_load(b + j) : acc;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long j = i + u * stride;
            if (MODE == 0) acc += v[u];
            else if (j < n4) o[j] = MODE == 1 ? acc : (MODE == 2 ? v[u] : v[u] + w[u]);
        }
    }
    if (MODE == 0 && acc.x == 12345.678f) sink[0] = acc.y;
}
int main() {
    const long bytes = 2L << 30, n4 = bytes / 16;
    f32x4 *a, *b, *o; float* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, bytes); hipMalloc(&sink, 64);
    hipMemset(a, 0, bytes); hipMemset(b, 0, bytes); hipMemset(o, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[4] = {"read only", "write only", "copy (1 read : 1 write)", "2 reads : 1 write"};
    const double moved[4] = {1.0, 1.0, 2.0, 3.0};
    for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k<2>, dim3(256 * 8), dim3(256), 0, 0, a, b, o, n4, sink);
    for (int mode = 0; mode < 4; ++mode)
        for (int blocks : {256 * 4, 256 * 8, 256 * 16}) {
            float best = 1e9f;
            for (int rep = 0; rep < 5; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, a, b, o, n4, sink);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, a, b, o, n4, sink);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, a, b, o, n4, sink);
                else hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, a, b, o, n4, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%-26s %5d blocks: %.3f ms  %.2f TB/s\n", names[mode], blocks, best, moved[mode] * bytes / best / 1e9);
        }
    return 0;
}
