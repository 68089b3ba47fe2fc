// What the fp16 matrix cores SUSTAIN under the package power cap, by operand content (round 4).  Register-only operands, no memory
// traffic, every CU busy with 8 waves (two per SIMD) for ~50-100 ms per case:
//   structured : the operands of tools/micro/mfma_peak.hip (small integers / smooth ramps: few toggling bits)      -> the nominal peak
//   random     : uniformly random fp16 values, a different operand register set for every instruction               -> real data
//   split      : operand pairs as the split engine feeds them: hi planes random in [2^13, 2^14) magnitude, lo planes = the
//                rounding residues (|lo| <= ulp(hi) / 2), the three terms ah bh, ah bl, al bh in the loop's order
// hipcc --offload-arch=gfx950 -O3 mfma_power.hip -o mfma_power && ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ _Float16 rnd16(uint32_t& s, float scale) {
    s = mix(s + 0x9e3779b9u);
    return (_Float16)(((int)(s >> 8) - (1 << 23)) * (scale / (1 << 23)));
}

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 ah[4], al[4], bh[4], bl[4];
    uint32_t s = blockIdx.x * 512 + threadIdx.x + 1;
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {
                ah[q][i] = (_Float16)(threadIdx.x * 1e-3f + i); bh[q][i] = (_Float16)(1.f + i * 0.5f);
                al[q][i] = ah[q][i]; bl[q][i] = bh[q][i];
            } else if (MODE == 1) {
                ah[q][i] = rnd16(s, 1.f); bh[q][i] = rnd16(s, 1.f); al[q][i] = rnd16(s, 1.f); bl[q][i] = rnd16(s, 1.f);
            } else {
                s = mix(s + 0x9e3779b9u);
                const float x = ((int)(s >> 8) - (1 << 23)) * (16384.f / (1 << 23));
                const _Float16 h = (_Float16)x; ah[q][i] = h; al[q][i] = (_Float16)(x - (float)h);
                s = mix(s + 0x9e3779b9u);
                const float y = ((int)(s >> 8) - (1 << 23)) * (16384.f / (1 << 23));
                const _Float16 g = (_Float16)y; bh[q][i] = g; bl[q][i] = (_Float16)(y - (float)g);
            }
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const f16x8& a = (MODE == 2 && ((q * 8 + i) % 3 == 2)) ? al[q] : ah[(q + i) & 3];
                const f16x8& b = (MODE == 2 && ((q * 8 + i) % 3 == 1)) ? bl[q] : bh[(q + 2 * i) & 3];
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            }
        }
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// the other two engines' instructions on random operands: bf16 32x32x16 and fp32 32x32x2
__global__ __launch_bounds__(512) void k_bf16(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[4], b[4];
    uint32_t s = blockIdx.x * 512 + threadIdx.x + 1;
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i) { a[q][i] = (__bf16)(float)rnd16(s, 1.f); b[q][i] = (__bf16)(float)rnd16(s, 1.f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(q + i) & 3], b[(q + 2 * i) & 3], acc[i], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}
__global__ __launch_bounds__(512) void k_f32(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a[4], b[4];
    uint32_t s = blockIdx.x * 512 + threadIdx.x + 1;
    for (int q = 0; q < 4; ++q) { a[q] = (float)rnd16(s, 1.f) + 1e-4f * (float)rnd16(s, 1.f); b[q] = (float)rnd16(s, 1.f) + 1e-4f * (float)rnd16(s, 1.f); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(q + i) & 3], b[(q + 2 * i) & 3], acc[i], 0, 0, 0);
    }
    float t = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) t += acc[i][r];
    out[blockIdx.x * 512 + threadIdx.x] = t;
}

int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256, iters = 30000;   // 256 CUs x 8 waves x 32 MFMAs x iters
    const char* names[3] = {"structured", "random", "split (hi/lo planes, 3 terms)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(512), 0, 0, out, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(512), 0, 0, out, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(512), 0, 0, out, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * 8 * 32 * iters * 32768.0;
            printf("%-32s %8.2f ms  %7.1f TFLOP/s  (= %.2f GHz x 1024 FLOP/clk/SIMD x 1024 SIMDs)\n", names[mode], ms, flop / ms / 1e9,
                   flop / ms / 1e9 / (1024.0 * 1024.0) * 1e3);
        }
    for (int rep = 0; rep < 2; ++rep) {
        float ms;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_bf16, dim3(grid), dim3(512), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("%-32s %8.2f ms  %7.1f TFLOP/s\n", "bf16 32x32x16, random", ms, (double)grid * 8 * 32 * iters * 32768.0 / ms / 1e9);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_f32, dim3(grid), dim3(512), 0, 0, out, iters / 2);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        printf("%-32s %8.2f ms  %7.1f TFLOP/s\n", "fp32 32x32x2, random", ms, (double)grid * 8 * 32 * (iters / 2) * 4096.0 / ms / 1e9);
    }
    return 0;
}
