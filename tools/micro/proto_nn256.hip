// Experiment: C = A B with a 256x256 workgroup tile, 4 waves of 128x128 (256 accumulators in AGPRs, 1 wave per SIMD),
// fp32 MFMA + LDS-DMA + in-wave fragment prefetch.  Steady state 131 TF at T=262144, N=2048, K=512 -- the same as the shipped 128x256
// engine (3 waves per SIMD), at full sclk (2.39 GHz, rocm-smi), so neither LDS traffic per MFMA nor occupancy is the limiter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int PM = 256, PN = 256, PK = 16;
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__global__ __launch_bounds__(256) void nn256(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, float* __restrict__ C,
                                             int64_t ldc, int64_t T, int Nc, int Kc) {
    __shared__ __attribute__((aligned(16))) struct { float A[2][PM * PK]; float B[2][PK][PN]; } sm;   // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wm = (tid >> 6) >> 1, wn = (tid >> 6) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncol = Nc / PN;
    const int nt = blockIdx.x % ncol;
    const int64_t t0 = (int64_t)(blockIdx.x / ncol) * PM;
    const int n0 = nt * PN;
    const float* srcA[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sl = (wave * 4 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        int64_t t = t0 + row; if (t > T - 1) t = T - 1;
        srcA[q] = A + t * lda + kq * 4;
    }
    const float* __restrict__ srcB = B + n0 + lane * 4;
    auto issue = [&](int st, int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(srcA[q] + k0, &sm.A[st][(wave * 4 + q) * 256]);
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(srcB + (int64_t)(k0 + wave * 4 + q) * Nc, &sm.B[st][wave * 4 + q][0]);
    };
    const int l32 = lane & 31, kh = lane >> 5;
    int offA[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) { const int r = wm * 128 + rt * 32 + l32; offA[rt] = r * PK + ((kh ^ ((r >> 2) & 3)) << 2); }
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nch = Kc / PK;
    issue(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int st = ch & 1;
        if (ch + 1 < nch) issue(st ^ 1, (ch + 1) * PK);
        // software pipeline inside the wave (1 wave / SIMD: nobody else hides the LDS latency): the fragments of step s+1
        // are requested before the 16 MFMAs of step s and pinned there with a scheduling barrier
        f32x4 fa[2][4];
        float fb[2][4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) fa[0][rt] = *reinterpret_cast<const f32x4*>(&sm.A[st][offA[rt]]);
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fb[0][ct] = sm.B[st][4 * kh][wn * 128 + ct * 32 + l32];
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            const int g = s8 >> 2, e = s8 & 3, cur = s8 & 1, nxt = cur ^ 1;
            if (s8 + 1 < 8) {
                const int g2 = (s8 + 1) >> 2, e2 = (s8 + 1) & 3;
                if (e2 == 0) {
#pragma unroll
                    for (int rt = 0; rt < 4; ++rt) fa[g2][rt] = *reinterpret_cast<const f32x4*>(&sm.A[st][offA[rt] ^ (g2 << 3)]);
                }
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) fb[nxt][ct] = sm.B[st][8 * g2 + 4 * kh + e2][wn * 128 + ct * 32 + l32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int rt = 0; rt < 4; ++rt)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g][rt][e], fb[cur][ct], acc[rt][ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t t = t0 + wm * 128 + rt * 32 + acc_row(r, lane);
            if (t < T) {
                float* __restrict__ o = C + t * ldc + n0 + wn * 128 + l32;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) o[ct * 32] = acc[rt][ct][r];
            }
        }
}
int main() {
    const int64_t T = 262144; const int N = 2048, K = 512;
    float *A, *B, *C; hipMalloc(&A, T * K * 4); hipMalloc(&B, (size_t)K * N * 4); hipMalloc(&C, T * N * 4);
    std::vector<float> hA(1024 * K), hB((size_t)K * N);
    for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
    for (int64_t r = 0; r < T; r += 1024) hipMemcpy(A + r * K, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = (int)((T / PM) * (N / PN));
    for (int rep = 0; rep < 14; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(nn256, dim3(grid), dim3(256), 0, 0, A, (int64_t)K, B, C, (int64_t)N, T, N, K);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("nn256: %.3f ms  %.1f TF\n", ms, 2.0 * T * N * K / ms / 1e9);
    }
    std::vector<float> hC(4 * N); hipMemcpy(hC.data(), C + (int64_t)777 * N, 4 * N * 4, hipMemcpyDeviceToHost);
    double maxerr = 0;
    for (int r = 0; r < 4; ++r) for (int n = 0; n < N; n += 97) {
        double s = 0; for (int k = 0; k < K; ++k) s += (double)hA[(size_t)((777 + r) % 1024) * K + k] * hB[(size_t)k * N + n];
        maxerr = fmax(maxerr, fabs(s - hC[(size_t)r * N + n]));
    }
    printf("max abs err vs fp64: %.3e\n", maxerr);
    return 0;
}
