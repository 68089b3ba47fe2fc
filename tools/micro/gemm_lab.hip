// gemm_lab.hip -- within-process A/B of fp32-MFMA tile-engine variants on the pre_attn / gate GEMM shape
// (C[T,N] = A[T,K] B[K,N], T = 262144, N = 2048, K = 512), without torch.  Round-2 experiment behind DESIGN.md section 6.1:
//   PIPE = 0  the round-1 loop: fragments requested right before use (compiler order), DMA issue block at the chunk top
//   PIPE = 1  software-pipelined inside the wave: fragments of step s+1 requested before the 8 MFMAs of step s, the chunk
//             barrier sits before the LAST step (its 8 MFMAs cover the post-barrier DMA issue + first fragment reads)
//   AGPR = 0  __launch_bounds__(256, 2): hipcc selects the VGPR form of the MFMAs (accumulators in arch VGPRs)
//   AGPR = 1  __launch_bounds__(256):    accumulators in AGPRs (what the register-only peak test uses)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_lab.hip -o gemm_lab ; run: ./gemm_lab [rounds]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 256, BK = 16;
constexpr int STAGGER_SLEEPS = 16;   // x 8128 cycles: one third of a 393k-cycle tile generation

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
// LDS-DMA with the saddr form: uniform 64-bit base in SGPRs + 32-bit per-lane byte offset; M0 = LDS destination of the wave.
// Not counted by hipcc's vmcnt bookkeeping: the caller waits (s_waitcnt vmcnt) itself before the barrier that publishes it.
__device__ __forceinline__ void glds16_s(uint32_t voff, const void* sbase, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n / 8, r = n % 8, xcd = bid % 8, pos = bid / 8;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + pos;
}

struct Smem {
    float A[2][BM * BK];
    float B[2][BK][BN];
};

template <int PIPE, int EPI, int PRIO>
__device__ __forceinline__ void nn_body(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, float* __restrict__ C,
                                        int64_t ldc, int64_t T, int Nc, int Kc, int n_tiles, Smem& sm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: SGPR addressing downstream
    const int wm = wave >> 1, wn = wave & 1;
    const int ncol = Nc / BN;
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int nt = lid % ncol;
    const int64_t t0 = (int64_t)(lid / ncol) * BM;
    const int n0 = nt * BN;
    if constexpr (PRIO >= 2) {   // de-synchronise the co-resident workgroups: the first generation starts staggered
        if (blockIdx.x < 768) {
            const int ph = (PRIO == 2) ? (int)(blockIdx.x % 3) : (int)((blockIdx.x / 8) % 3);
            for (int i = 0; i < ph * STAGGER_SLEEPS; ++i) __builtin_amdgcn_s_sleep(127);   // 127 x 64 cycles each
        }
    }
    if constexpr (PRIO == 1) {   // de-synchronise the co-resident workgroups: static issue priority from the tile index
        const int pr = lid % 3;
        if (pr == 1) __builtin_amdgcn_s_setprio(1);
        else if (pr == 2) __builtin_amdgcn_s_setprio(2);
    }
    const float* srcA[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sl = (wave * 2 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        int64_t t = t0 + row;
        if (t > T - 1) t = T - 1;
        srcA[q] = A + t * lda + kq * 4;
    }
    const float* srcB = B + n0 + lane * 4 + (int64_t)(wave * 4) * Nc;
    const int l32 = lane & 31, kh = lane >> 5;
    const int colb[4] = {wn * 128, wn * 128 + 32, wn * 128 + 64, wn * 128 + 96};
    int offA[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * BK + ((kh ^ ((r >> 2) & 3)) << 2);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nch = Kc / BK;

    if constexpr (PIPE == 0) {
        auto issue = [&](int st, int k0) {
#pragma unroll
            for (int q = 0; q < 2; ++q) glds16(srcA[q] + k0, &sm.A[st][(wave * 2 + q) * 256]);
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16(srcB + (int64_t)(k0 + q) * Nc, &sm.B[st][wave * 4 + q][0]);
        };
        issue(0, 0);
        __syncthreads();
        for (int ch = 0; ch < nch; ++ch) {
            const int st = ch & 1;
            if (ch + 1 < nch) issue(st ^ 1, (ch + 1) * BK);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                f32x4 fa[2];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const f32x4*>(&sm.A[st][offA[rt] ^ (g << 3)]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float fb[4];
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) fb[ct] = sm.B[st][8 * g + 4 * kh + e][colb[ct] + l32];
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const int rt = m & 1, ct = m >> 1;
                        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rt][e], fb[ct], acc[rt][ct], 0, 0, 0);
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // DMA sources: uniform 64-bit bases (SGPRs, advanced by scalar adds) + per-lane 32-bit byte offsets inside the tile
        const char* baseA = reinterpret_cast<const char*>(A + t0 * lda);
        const char* baseB = reinterpret_cast<const char*>(B + n0 + (int64_t)(wave * 4) * Nc);
        uint32_t voA[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int sl = (wave * 2 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
            int64_t rr = row;
            if (t0 + rr > T - 1) rr = T - 1 - t0;
            voA[q] = (uint32_t)(rr * lda * 4 + kq * 16);
        }
        const uint32_t voB = lane * 16;
        const int64_t rowB = (int64_t)Nc * 4;
        // chunk f of the K range (f is clamped by the callers: the last two iterations re-fetch the last chunk into a
        // stage nobody reads again, which keeps the loop body branch-free)
        auto dmaA = [&](int st, int f) {
            const char* a = baseA + (int64_t)f * (BK * 4);
            glds16_s(voA[0], a, lds_addr_of(&sm.A[st][(wave * 2 + 0) * 256]));
            glds16_s(voA[1], a, lds_addr_of(&sm.A[st][(wave * 2 + 1) * 256]));
        };
        auto dmaB = [&](int st, int f, int q) {
            glds16_s(voB, baseB + ((int64_t)f * BK + q) * rowB, lds_addr_of(&sm.B[st][wave * 4 + q][0]));
        };
#define DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
        f32x4 fa0[2], fa1[2];   // A fragments of k-group 0 / 1 of the current chunk (rt = 0, 1)
        float fb0[4], fb1[4];   // B fragments of even / odd steps
        auto ldA = [&](f32x4 (&fa)[2], int st, int g) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const f32x4*>(&sm.A[st][offA[rt] ^ (g << 3)]);
        };
        auto ldB = [&](float (&fb)[4], int st, int s) {   // step s = 4 g + e
            const int g = s >> 2, e = s & 3;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fb[ct] = sm.B[st][8 * g + 4 * kh + e][colb[ct] + l32];
        };
        auto mma1 = [&](const f32x4 (&fa)[2], int e, const float (&fb)[4], int m) {
            const int rt = m & 1, ct = m >> 1;
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rt][e], fb[ct], acc[rt][ct], 0, 0, 0);
        };
#define SB() __builtin_amdgcn_sched_barrier(0)
        // one step: first MFMA, then the fragment requests of the NEXT step (their registers were last read by the previous
        // step, fully issued by now), then the other 7 MFMAs: the requests have 7 MFMAs (~450 cycles) to land
#define STEP(FA, E, FB, LOADS)                          \
        mma1(FA, E, FB, 0);                             \
        SB();                                           \
        LOADS;                                          \
        SB();                                           \
        _Pragma("unroll") for (int m = 1; m < 8; ++m) mma1(FA, E, FB, m); \
        SB();
        dmaA(0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) dmaB(0, 0, q);
        DMA_WAIT();
        __syncthreads();
        {
            const int f = nch > 1 ? 1 : 0;
            dmaA(1, f);
#pragma unroll
            for (int q = 0; q < 4; ++q) dmaB(1, f, q);
        }
        ldA(fa0, 0, 0);
        ldB(fb0, 0, 0);
        for (int ch = 0; ch < nch; ++ch) {
            const int st = ch & 1;
            STEP(fa0, 0, fb0, ldB(fb1, st, 1); ldA(fa1, st, 1))
            STEP(fa0, 1, fb1, ldB(fb0, st, 2))
            STEP(fa0, 2, fb0, ldB(fb1, st, 3))
            STEP(fa0, 3, fb1, ldB(fb0, st, 4))
            STEP(fa1, 0, fb0, ldB(fb1, st, 5))
            STEP(fa1, 1, fb1, ldB(fb0, st, 6))
            STEP(fa1, 2, fb0, ldB(fb1, st, 7))
            // last step: every read of stage `st` has been requested; the barrier waits for them (lgkmcnt) and for this wave's
            // DMA pieces of chunk ch+1 (vmcnt); then stage st is free for chunk ch+2 and stage st^1 is readable.
            DMA_WAIT();
            __syncthreads();
            ldA(fa0, st ^ 1, 0);
            ldB(fb0, st ^ 1, 0);
            SB();
            const int f = (ch + 2 < nch) ? ch + 2 : nch - 1;   // the DMA of chunk ch+2 rides between the MFMAs of this step
            mma1(fa1, 3, fb1, 0);
            SB();
            dmaA(st, f);
            SB();
            mma1(fa1, 3, fb1, 1);
            SB();
            dmaB(st, f, 0);
            dmaB(st, f, 1);
            SB();
            mma1(fa1, 3, fb1, 2);
            SB();
            dmaB(st, f, 2);
            dmaB(st, f, 3);
            SB();
#pragma unroll
            for (int m = 3; m < 8; ++m) mma1(fa1, 3, fb1, m);
            SB();
        }
#undef STEP
#undef SB
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the two redundant tail fetches
    }
    if constexpr (EPI == 0) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t t = t0 + wm * 64 + rt * 32 + acc_row(r, lane);
                if (t < T) {
                    float* __restrict__ o = C + t * ldc + n0 + l32;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) o[colb[ct]] = acc[rt][ct][r];
                }
            }
    } else if constexpr (EPI == 1) {
        // wave-private LDS transpose: 4 passes of a 32 x 64 sub-tile (8 KiB per wave), then 16-B row-contiguous stores
        // (16 lanes = 256 B of one row): 32 store instructions per wave instead of 128
        __syncthreads();   // every wave is done with the staging buffers
        float* tile = reinterpret_cast<float*>(&sm) + wave * (32 * 64);
        const int rl = lane >> 4, c4 = lane & 15;
        const uint32_t ldc32 = (uint32_t)ldc;
        const uint32_t voff = (rl * ldc32 + c4 * 4) * 4;          // byte offset of this lane inside a 32-row block
        const bool full = t0 + BM <= T;                             // block-uniform
        auto run = [&](auto full_c) {
            constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const int64_t tb = t0 + wm * 64 + rt * 32;
                const int nvalid = FULL ? 32 : (int)((T - tb) < 0 ? 0 : ((T - tb) > 32 ? 32 : (T - tb)));
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) tile[acc_row(r, lane) * 64 + c2 * 32 + l32] = acc[rt][cp * 2 + c2][r];
                    char* cb = reinterpret_cast<char*>(C + tb * ldc + n0 + wn * 128 + cp * 64);   // uniform base of the pass
#pragma unroll
                    for (int h = 0; h < 2; ++h) {   // 4 reads in flight, then their 4 stores
                        f32x4 v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const f32x4*>(&tile[((h * 4 + j) * 4 + rl) * 64 + c4 * 4]);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (FULL || (h * 4 + j) * 4 + rl < nvalid)
                                *reinterpret_cast<f32x4*>(cb + (voff + (uint32_t)(h * 4 + j) * 16u * ldc32)) = v[j];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };
        if (full) run(std::true_type{});
        else run(std::false_type{});
    } else {
        if (T < 0) {   // never true at run time: keeps the accumulators alive without an epilogue
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int r = 0; r < 16; ++r)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) C[(int64_t)r * ldc + colb[ct] + l32 + rt] = acc[rt][ct][r];
        }
    }
}

#define KERNEL2(NAME, PIPE, EPI, PRIO, PADKB)                                                                              \
    __global__ __launch_bounds__(256) void NAME(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,        \
                                                float* __restrict__ C, int64_t ldc, int64_t T, int Nc, int Kc, int n_tiles) { \
        __shared__ __attribute__((aligned(16))) struct { Smem s; float pad[PADKB * 256]; } smx;                               \
        if (T < 0) smx.pad[threadIdx.x] = 1.f;  /* keeps the padding allocated: fewer workgroups per CU */                    \
        nn_body<PIPE, EPI, PRIO>(A, lda, B, C, ldc, T, Nc, Kc, n_tiles, smx.s);                                                \
    }
#define KERNEL(NAME, PIPE, EPI, PRIO, ...)                                                                                             \
    __global__ __launch_bounds__(__VA_ARGS__) void NAME(const float* __restrict__ A, int64_t lda, const float* __restrict__ B, \
                                                        float* __restrict__ C, int64_t ldc, int64_t T, int Nc, int Kc,       \
                                                        int n_tiles) {                                                      \
        __shared__ __attribute__((aligned(16))) Smem sm;                                                                    \
        nn_body<PIPE, EPI, PRIO>(A, lda, B, C, ldc, T, Nc, Kc, n_tiles, sm);                                                            \
    }
KERNEL(k_base_v, 0, 0, 0, 256, 2)
KERNEL(k_pipe_v, 1, 0, 0, 256, 2)
KERNEL(k_pipe_a, 1, 0, 0, 256)
KERNEL(k_pipe_prio, 1, 0, 1, 256)
KERNEL(k_pipe_lds, 1, 1, 0, 256)
KERNEL(k_pipe_lds_prio, 1, 1, 1, 256)
KERNEL(k_pipe_noepi, 1, 2, 0, 256)
KERNEL(k_pipe_lds_stag, 1, 1, 2, 256)
KERNEL(k_pipe_lds_stag8, 1, 1, 3, 256)
KERNEL(k_pipe_dir_stag, 1, 0, 2, 256)
KERNEL2(k_pipe_lds_2wg, 1, 1, 0, 24)   /* 72 KiB: 2 workgroups per CU */
KERNEL2(k_pipe_lds_1wg, 1, 1, 0, 48)   /* 96 KiB: 1 workgroup per CU */
KERNEL(k_base_noepi, 0, 2, 0, 256, 2)

// register-only MFMA rate with the accumulators forced into arch VGPRs / AGPRs (3 waves per SIMD like the engine)
__global__ __launch_bounds__(256, 2) void peak_v(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void peak_a(float* out, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef void (*kern_t)(const float*, int64_t, const float*, float*, int64_t, int64_t, int, int, int);

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 6;
    const int64_t T = 262144;
    const int N = 2048, K = 512;
    float *A, *B, *C;
    hipMalloc(&A, T * K * 4);
    hipMalloc(&B, (size_t)K * N * 4);
    hipMalloc(&C, T * N * 4);
    std::vector<float> hA(1024 * K), hB((size_t)K * N);
    srand(1);
    for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
    for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
    for (int64_t r = 0; r < T; r += 1024) hipMemcpy(A + r * K, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct V { const char* name; kern_t k; };
    const V vs[] = {{"base_vgpr", k_base_v}, {"pipe_vgpr", k_pipe_v}, {"pipe_agpr", k_pipe_a}, {"pipe_prio", k_pipe_prio},
                    {"pipe_lds", k_pipe_lds}, {"pipe_lds_prio", k_pipe_lds_prio}, {"pipe_noepi", k_pipe_noepi},
                    {"pipe_lds_2wg", k_pipe_lds_2wg}, {"pipe_lds_1wg", k_pipe_lds_1wg}, {"pipe_lds_stag", k_pipe_lds_stag},
                    {"pipe_lds_stag8", k_pipe_lds_stag8}, {"pipe_dir_stag", k_pipe_dir_stag},
                    {"base_noepi", k_base_noepi}};
    const int nv = sizeof(vs) / sizeof(vs[0]);
    const int64_t Ts[2] = {T, 1000};   // full size (timing) and a ragged small problem (tails)
    // correctness first: rows 777.. of the big problem and the whole ragged problem against fp64 on the host
    for (int v = 0; v < nv; ++v) {
        if (strstr(vs[v].name, "noepi")) continue;
        for (int which = 0; which < 2; ++which) {
            const int64_t Tc = Ts[which];
            const int tiles = (int)(((Tc + BM - 1) / BM) * (N / BN));
            hipMemset(C, 0xff, (size_t)Tc * N * 4);
            hipLaunchKernelGGL(vs[v].k, dim3(tiles), dim3(256), 0, 0, A, (int64_t)K, B, C, (int64_t)N, Tc, N, K, tiles);
            hipDeviceSynchronize();
            const int64_t r0 = which ? 0 : 777;
            const int nr = which ? 1000 : 8;
            std::vector<float> hC((size_t)nr * N);
            hipMemcpy(hC.data(), C + r0 * N, hC.size() * 4, hipMemcpyDeviceToHost);
            double maxerr = 0;
            for (int r = 0; r < nr; r += (which ? 37 : 1))
                for (int n = 0; n < N; n += 13) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += (double)hA[(size_t)((r0 + r) % 1024) * K + k] * hB[(size_t)k * N + n];
                    maxerr = fmax(maxerr, fabs(s - hC[(size_t)r * N + n]));
                }
            printf("check %-10s T=%-7lld max abs err vs fp64 %.3e %s\n", vs[v].name, (long long)Tc, maxerr, maxerr < 1e-3 ? "OK" : "FAIL");
        }
    }
    const int tiles = (int)((T / BM) * (N / BN));
    std::vector<double> best(nv, 1e9), sum(nv, 0);
    for (int rd = 0; rd < rounds; ++rd)
        for (int v = 0; v < nv; ++v) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(vs[v].k, dim3(tiles), dim3(256), 0, 0, A, (int64_t)K, B, C, (int64_t)N, T, N, K, tiles);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rd > 0) { sum[v] += ms; if (ms < best[v]) best[v] = ms; }
        }
    for (int v = 0; v < nv; ++v)
        printf("%-10s mean %.3f ms (%.1f TF)  best %.3f ms (%.1f TF)\n", vs[v].name, sum[v] / (rounds - 1),
               2.0 * T * N * K / (sum[v] / (rounds - 1)) / 1e9, best[v], 2.0 * T * N * K / best[v] / 1e9);
    return 0;
}
