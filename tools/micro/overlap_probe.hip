// Do the matrix pipe and the VALU of ONE SIMD run concurrently when the MFMAs and the VALU work come from DIFFERENT waves on that SIMD?
// One 512-thread workgroup per CU (2 waves per SIMD).  mode 0: waves 0-3 MFMA chain, waves 4-7 idle; mode 1: waves 4-7 VALU chain
// (fma / exp / rcp mix like the gate epilogue), waves 0-3 idle; mode 2: both; mode 3: every wave alternates phases (MFMA block, VALU
// block) in lockstep -- the shape of the gate forward today; mode 4: same, the two waves of a SIMD in opposite phase.
// hipcc --offload-arch=gfx950 -O3 overlap_probe.hip -o overlap_probe && ./overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void mfma_block(f32x16 (&acc)[8], const f16x8& a, const f16x8& b, int n) {
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
    }
}
__device__ __forceinline__ void valu_block(float (&v)[16], int n) {
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float x = v[i];
            x = __builtin_amdgcn_rcpf(1.f + __expf(-x));          // sigmoid: exp + rcp + 2 plain
            x = fmaf(x, 1.0001f, 0.25f);
            x = fmaf(x, 0.9999f, -0.125f);
            x = fmaf(x, x, 0.01f);
            v[i] = x;
        }
    }
}
__device__ __forceinline__ void fma_block(float (&v)[16], int n) {
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float x = v[i];
            x = fmaf(x, 1.0001f, 0.25f);
            x = fmaf(x, 0.9999f, -0.125f);
            x = fmaf(x, 0.5f, 0.01f);
            x = fmaf(x, 1.0001f, 0.25f);
            x = fmaf(x, 0.9999f, -0.125f);
            x = fmaf(x, 0.5f, 0.01f);
            v[i] = x;
        }
    }
}
// one wave: every MFMA followed by K independent fma (the "VALU in the shadow of the MFMA" pattern)
template <int K>
__device__ __forceinline__ void interleaved_block(f32x16 (&acc)[8], const f16x8& a, const f16x8& b, float (&v)[16], int n) {
    for (int it = 0; it < n; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < K; ++k) v[(i * K + k) & 15] = fmaf(v[(i * K + k) & 15], 1.0001f, 0.25f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <int mode>
__global__ __launch_bounds__(512) void probe(float* out, int n_mfma, int n_valu, int rounds) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 1e-3f + i); b[i] = (_Float16)(1.f + i * 0.5f); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.1f;
    const bool lower = wave < 4;
    for (int r = 0; r < rounds; ++r) {
        if (mode == 0) { if (lower) mfma_block(acc, a, b, n_mfma); }
        else if (mode == 1) { if (!lower) valu_block(v, n_valu); }
        else if (mode == 2) { if (lower) mfma_block(acc, a, b, n_mfma); else valu_block(v, n_valu); }
        else if (mode == 3) { mfma_block(acc, a, b, n_mfma / 2); valu_block(v, n_valu / 2); }
        else if (mode == 4) { if (lower) { mfma_block(acc, a, b, n_mfma / 2); valu_block(v, n_valu / 2); } else { valu_block(v, n_valu / 2); mfma_block(acc, a, b, n_mfma / 2); } }
        else if (mode == 5) { if (!lower) fma_block(v, n_valu); }                                              // plain-fma waves only
        else if (mode == 6) { if (lower) mfma_block(acc, a, b, n_mfma); else fma_block(v, n_valu); }           // MFMA waves + plain-fma waves
        else if (mode == 7) { if (lower) interleaved_block<0>(acc, a, b, v, n_mfma); }                         // one wave, MFMA only (same loop shape)
        else if (mode == 8) { if (lower) interleaved_block<4>(acc, a, b, v, n_mfma); }                         // one wave, 4 fma behind every MFMA
        else if (mode == 9) { if (lower) interleaved_block<8>(acc, a, b, v, n_mfma); }                         // 8 fma behind every MFMA
        else { if (lower) interleaved_block<12>(acc, a, b, v, n_mfma); }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[11] = {"MFMA waves only (1 per SIMD)", "VALU waves only (1 per SIMD)", "MFMA waves + VALU waves (different waves, same SIMD)",
                            "all 8 waves: MFMA block then VALU block, lockstep", "all 8 waves alternate, the two waves of a SIMD in opposite phase", "plain-fma waves only (1 per SIMD)",
                            "MFMA waves + plain-fma waves (different waves, same SIMD)", "one wave per SIMD: MFMA only (interleave loop shape)",
                            "one wave per SIMD: 4 fma behind every MFMA", "one wave per SIMD: 8 fma behind every MFMA", "one wave per SIMD: 12 fma behind every MFMA"};
    for (int w = 0; w < 40; ++w) hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, out, 32, 32, 2000);   // clocks up (~1 s)
    hipDeviceSynchronize();
    for (int nm = 32; nm <= 32; nm *= 2)
    for (int nv : {16, 32}) {
        printf("-- per round: MFMA block = %d x 8 MFMAs (%d cycles of matrix pipe), VALU block = %d x 16 x 6 instructions\n", nm, nm * 8 * 32, nv);
        for (int mode = 0; mode < 11; ++mode) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                switch (mode) {
                    case 0: hipLaunchKernelGGL(probe<0>, dim3(256), dim3(512), 0, 0, out, nm, nv, 2000); break;
                    case 1: hipLaunchKernelGGL(probe<1>, dim3(256), dim3(512), 0, 0, out, nm, nv, 2000); break;
                    case 2: hipLaunchKernelGGL(probe<2>, dim3(256), dim3(512), 0, 0, out, nm, nv, 2000); break;
                    case 3: hipLaunchKernelGGL(probe<3>, dim3(256), dim3(512), 0, 0, out, nm, nv, 2000); break;
                    case 4: hipLaunchKernelGGL(probe<4>, dim3(256), dim3(512), 0, 0, out, nm, nv, 2000); break;
                    case 5: hipLaunchKernelGGL(probe<5>, dim3(256), dim3(512), 0, 0, out, nm, nv, 2000); break;
                    case 6: hipLaunchKernelGGL(probe<6>, dim3(256), dim3(512), 0, 0, out, nm, nv, 2000); break;
                    case 7: hipLaunchKernelGGL(probe<7>, dim3(256), dim3(512), 0, 0, out, nm, nv, 2000); break;
                    case 8: hipLaunchKernelGGL(probe<8>, dim3(256), dim3(512), 0, 0, out, nm, nv, 2000); break;
                    case 9: hipLaunchKernelGGL(probe<9>, dim3(256), dim3(512), 0, 0, out, nm, nv, 2000); break;
                    default: hipLaunchKernelGGL(probe<10>, dim3(256), dim3(512), 0, 0, out, nm, nv, 2000); break;
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("   mode %d %-66s %.3f ms\n", mode, names[mode], best);
        }
    }
    return 0;
}
