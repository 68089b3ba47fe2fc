// gemm_lab_bf16.hip -- standalone A/B of the bf16 "NT" tile engine:  C[m, n] (fp32) = sum_k A[m][k] B[n][k], bf16 operands,
// v_mfma_f32_32x32x16_bf16.  Gate-forward shape: M = 262144 tokens, N = 1024 ([Wa;Wb] rows of one head), K = 512.
//   v0  the round-1 engine shape: 128 x 256 tile, BK = 32, 4 waves, barrier + DMA issue block per 16 MFMAs
//   v1  256 x 256 tile, BK = 64, 8 waves (2 x 4, 128 x 64 per wave), 2 LDS stages of 64 KiB, software-pipelined like the
//       fp32 engine (tile_engine.hpp): fragments of k-step s+1 requested after the first MFMA of step s, the chunk barrier
//       before the last k-step, the next chunk's 8 LDS-DMA pieces between that step's MFMAs, saddr-form DMA.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 gemm_lab_bf16.hip -o gemm_lab_bf16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

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
#define DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define SB() __builtin_amdgcn_sched_barrier(0)

// ------------------------------------------------------------------------------------------------------------------
// v1: 256 x 256 x 64
// LDS image of one operand stage: [256 rows][64 k] bf16 = 128 B per row, 16-B chunk c of row r stored at chunk position
// c ^ ((r >> 1) & 7): with the 128-B row stride a ds_read_b128 lane group (16 lanes, rows r..) then touches 16 distinct
// 16-B slots of the 256-B bank row -> conflict-free.
constexpr int PM = 256, PN = 256, PK = 64;
constexpr int STAGE_BYTES = PM * PK * 2;   // 32 KiB per operand per stage
struct SmemP {
    char A[2][STAGE_BYTES];
    char B[2][STAGE_BYTES];
};

template <int EPI>
__device__ __forceinline__ void nt256_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                           float* __restrict__ C, int64_t ldc, int64_t M, int N, int K, SmemP& sm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;   // 2 x 4: rows wm*128 + rt*32 (rt < 4), columns wn*64 + ct*32 (ct < 2)
    const int ncol = N / PN;
    const int nt = blockIdx.x % ncol;
    const int64_t m0 = (int64_t)(blockIdx.x / ncol) * PM;
    const int n0 = nt * PN;
    const int l32 = lane & 31, kh = lane >> 5;

    // DMA: one instruction = 8 rows x 128 B; wave w issues row blocks 4w .. 4w+3 of each operand (256 rows = 32 blocks)
    const char* baseA = reinterpret_cast<const char*>(A + m0 * lda);
    const char* baseB = reinterpret_cast<const char*>(B + (int64_t)n0 * ldb);
    uint32_t voA[4], voB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t ra = row;
        if (m0 + ra > M - 1) ra = M - 1 - m0;
        voA[i] = (uint32_t)(ra * lda * 2 + c * 16);
        voB[i] = (uint32_t)((int64_t)row * ldb * 2 + c * 16);
    }
    auto dma = [&](int st, int f, int piece) {   // piece 0..7: 0-3 = A row blocks, 4-7 = B row blocks
        const int i = piece & 3;
        if (piece < 4) glds16_s(voA[i], baseA + (int64_t)f * (PK * 2), lds_addr_of(&sm.A[st][(wave * 4 + i) * 1024]));
        else glds16_s(voB[i], baseB + (int64_t)f * (PK * 2), lds_addr_of(&sm.B[st][(wave * 4 + i) * 1024]));
    };
    // fragment addresses (stage 0, k-step 0); k-step ks: ^ (ks << 5); stage: + st * STAGE_BYTES
    uint32_t offA[4], offB[2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int r = wm * 128 + rt * 32 + l32;
        offA[rt] = r * 128 + ((kh ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        const int r = wn * 64 + ct * 32 + l32;
        offB[ct] = r * 128 + ((kh ^ ((r >> 1) & 7)) << 4);
    }
    bf16x8 fa0[4], fb0[2], fa1[4], fb1[2];
    auto ld = [&](bf16x8 (&fa)[4], bf16x8 (&fb)[2], int st, int ks) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) fa[rt] = *reinterpret_cast<const bf16x8*>(&sm.A[st][offA[rt] ^ (ks << 5)]);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(&sm.B[st][offB[ct] ^ (ks << 5)]);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma1 = [&](const bf16x8 (&fa)[4], const bf16x8 (&fb)[2], int m) {
        const int rt = m >> 1, ct = m & 1;
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
    };
#define KSTEP(FA, FB, LOADS)                                                    \
    mma1(FA, FB, 0);                                                            \
    SB();                                                                       \
    LOADS;                                                                      \
    SB();                                                                       \
    _Pragma("unroll") for (int m = 1; m < 8; ++m) mma1(FA, FB, m);              \
    SB();
    const int nch = K / PK;
#pragma unroll
    for (int p = 0; p < 8; ++p) dma(0, 0, p);
    DMA_WAIT();
    __syncthreads();
    {
        const int f = nch > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) dma(1, f, p);
    }
    ld(fa0, fb0, 0, 0);
    for (int ch = 0; ch < nch; ++ch) {
        const int st = ch & 1;
        KSTEP(fa0, fb0, ld(fa1, fb1, st, 1))
        KSTEP(fa1, fb1, ld(fa0, fb0, st, 2))
        KSTEP(fa0, fb0, ld(fa1, fb1, st, 3))
        DMA_WAIT();
        __syncthreads();
        ld(fa0, fb0, st ^ 1, 0);
        SB();
        const int f = (ch + 2 < nch) ? ch + 2 : nch - 1;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            mma1(fa1, fb1, m);
            SB();
            dma(st, f, m);
            SB();
        }
    }
    DMA_WAIT();
    __syncthreads();
    // epilogue: plain (lab): row-per-lane dword stores; EPI = 1: none (main-loop rate)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 128 + rt * 32 + acc_row(r, lane);
            if (EPI == 1 ? (M < 0) : (m < M)) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) C[m * ldc + n0 + wn * 64 + ct * 32 + l32] = acc[rt][ct][r];
            }
        }
}
__global__ __launch_bounds__(512) void nt256(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                             float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ __attribute__((aligned(16))) SmemP sm;
    nt256_body<0>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}
__global__ __launch_bounds__(512) void nt256_noepi(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                   float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ __attribute__((aligned(16))) SmemP sm;
    nt256_body<1>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}

// ------------------------------------------------------------------------------------------------------------------
// v0: the round-1 engine shape (128 x 256 x 32, 4 waves, 2 LDS stages of 24 KiB, compiler-ordered loop)
constexpr int BBM = 128, BBN = 256, BBK = 32;
struct __attribute__((aligned(16))) SmemNT {
    bf16_t A[2][BBM * BBK];
    bf16_t B[2][BBN * BBK];
};
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}
template <int EPI>
__device__ __forceinline__ void nt128_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                           float* __restrict__ C, int64_t ldc, int64_t M, int N, int K, SmemNT& sm) {
    const int tid = threadIdx.x, lane = tid & 63, wm = (tid >> 6) >> 1, wn = (tid >> 6) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncol = N / BBN;
    const int nt = blockIdx.x % ncol;
    const int64_t m0 = (int64_t)(blockIdx.x / ncol) * BBM;
    const int n0 = nt * BBN;
    const bf16_t* srcA[2];
    const bf16_t* srcB[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sl = (wave * 2 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        int64_t t = m0 + row;
        if (t > M - 1) t = M - 1;
        srcA[q] = A + t * lda + kq * 8;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sl = (wave * 4 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        srcB[q] = B + (int64_t)(n0 + row) * ldb + kq * 8;
    }
    auto issue = [&](int st, int ch) {
        const int k0 = ch * BBK;
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(srcA[q] + k0, &sm.A[st][(wave * 2 + q) * 512]);
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(srcB[q] + k0, &sm.B[st][(wave * 4 + q) * 512]);
    };
    const int l32 = lane & 31, kh = lane >> 5;
    int offA[2], offB[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int r = wn * 128 + ct * 32 + l32;
        offB[ct] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nch = K / BBK;
    issue(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int st = ch & 1;
        if (ch + 1 < nch) issue(st ^ 1, ch + 1);
        const char* Ab = reinterpret_cast<const char*>(sm.A[st]);
        const char* Bb = reinterpret_cast<const char*>(sm.B[st]);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const bf16x8*>(Ab + (offA[rt] ^ (g << 5)));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(Bb + (offB[ct] ^ (g << 5)));
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int rt = m & 1, ct = m >> 1;
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 64 + rt * 32 + acc_row(r, lane);
            if (EPI == 1 ? (M < 0) : (m < M)) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) C[m * ldc + n0 + wn * 128 + ct * 32 + l32] = acc[rt][ct][r];
            }
        }
}
__global__ __launch_bounds__(256, 2) void nt128(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNT sm;
    nt128_body<0>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}
__global__ __launch_bounds__(256, 2) void nt128_noepi(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                      float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNT sm;
    nt128_body<1>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}

// v2: 128 x 256 x 32, 4 waves, the round-1 LDS image, but the in-wave pipeline of the fp32 engine (2 k-steps per chunk: the
// barrier sits before the second one, whose 8 MFMAs carry the next fragment requests and the 6 DMA pieces) and saddr-form DMA.
template <int EPI>
__device__ __forceinline__ void nt128p_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                            float* __restrict__ C, int64_t ldc, int64_t M, int N, int K, SmemNT& sm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ncol = N / BBN;
    const int nt = blockIdx.x % ncol;
    const int64_t m0 = (int64_t)(blockIdx.x / ncol) * BBM;
    const int n0 = nt * BBN;
    const char* baseA = reinterpret_cast<const char*>(A + m0 * lda);
    const char* baseB = reinterpret_cast<const char*>(B + (int64_t)n0 * ldb);
    uint32_t voA[2], voB[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sl = (wave * 2 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        int64_t r = row;
        if (m0 + r > M - 1) r = M - 1 - m0;
        voA[q] = (uint32_t)(r * lda * 2 + kq * 16);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sl = (wave * 4 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        voB[q] = (uint32_t)((int64_t)row * ldb * 2 + kq * 16);
    }
    auto dma = [&](int st, int f, int piece) {   // 0,1: A; 2..5: B
        if (piece < 2) glds16_s(voA[piece], baseA + (int64_t)f * (BBK * 2), lds_addr_of(&sm.A[st][(wave * 2 + piece) * 512]));
        else glds16_s(voB[piece - 2], baseB + (int64_t)f * (BBK * 2), lds_addr_of(&sm.B[st][(wave * 4 + piece - 2) * 512]));
    };
    const int l32 = lane & 31, kh = lane >> 5;
    int offA[2], offB[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int r = wn * 128 + ct * 32 + l32;
        offB[ct] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
    bf16x8 fa0[2], fb0[4], fa1[2], fb1[4];
    auto ld = [&](bf16x8 (&fa)[2], bf16x8 (&fb)[4], int st, int g) {
        const char* Ab = reinterpret_cast<const char*>(sm.A[st]);
        const char* Bb = reinterpret_cast<const char*>(sm.B[st]);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const bf16x8*>(Ab + (offA[rt] ^ (g << 5)));
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(Bb + (offB[ct] ^ (g << 5)));
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma1 = [&](const bf16x8 (&fa)[2], const bf16x8 (&fb)[4], int m) {
        const int rt = m & 1, ct = m >> 1;
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
    };
    const int nch = K / BBK;
#pragma unroll
    for (int p = 0; p < 6; ++p) dma(0, 0, p);
    DMA_WAIT();
    __syncthreads();
    {
        const int f = nch > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 6; ++p) dma(1, f, p);
    }
    ld(fa0, fb0, 0, 0);
    for (int ch = 0; ch < nch; ++ch) {
        const int st = ch & 1;
        mma1(fa0, fb0, 0);
        SB();
        ld(fa1, fb1, st, 1);
        SB();
#pragma unroll
        for (int m = 1; m < 8; ++m) mma1(fa0, fb0, m);
        SB();
        if (EPI != 3) {
            DMA_WAIT();
            __syncthreads();
        }
        ld(fa0, fb0, st ^ 1, 0);
        SB();
        const int f = (ch + 2 < nch) ? ch + 2 : nch - 1;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            mma1(fa1, fb1, m);
            SB();
            if (EPI != 3 && m < 6) dma(st, f, m);
            SB();
        }
    }
    DMA_WAIT();
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 64 + rt * 32 + acc_row(r, lane);
            if (EPI >= 1 ? (M < 0) : (m < M)) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) C[m * ldc + n0 + wn * 128 + ct * 32 + l32] = acc[rt][ct][r];
            }
        }
}
template <int NS>
struct __attribute__((aligned(16))) SmemNTS {
    bf16_t A[NS][BBM * BBK];
    bf16_t B[NS][BBN * BBK];
};
// v6: v2 (in-wave pipelined) on a 3-stage ring: the DMA of chunk ch+3 is issued behind the barrier of chunk ch (stage of chunk ch is
// free once its fragments are in registers), so ~2.5 stages per workgroup are in flight instead of ~1.
template <int EPI>
__device__ __forceinline__ void nt128pr_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                            float* __restrict__ C, int64_t ldc, int64_t M, int N, int K, SmemNTS<3>& sm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ncol = N / BBN;
    const int nt = blockIdx.x % ncol;
    const int64_t m0 = (int64_t)(blockIdx.x / ncol) * BBM;
    const int n0 = nt * BBN;
    const char* baseA = reinterpret_cast<const char*>(A + m0 * lda);
    const char* baseB = reinterpret_cast<const char*>(B + (int64_t)n0 * ldb);
    uint32_t voA[2], voB[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sl = (wave * 2 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        int64_t r = row;
        if (m0 + r > M - 1) r = M - 1 - m0;
        voA[q] = (uint32_t)(r * lda * 2 + kq * 16);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sl = (wave * 4 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        voB[q] = (uint32_t)((int64_t)row * ldb * 2 + kq * 16);
    }
    auto dma = [&](int st, int f, int piece) {   // 0,1: A; 2..5: B
        if (piece < 2) glds16_s(voA[piece], baseA + (int64_t)f * (BBK * 2), lds_addr_of(&sm.A[st][(wave * 2 + piece) * 512]));
        else glds16_s(voB[piece - 2], baseB + (int64_t)f * (BBK * 2), lds_addr_of(&sm.B[st][(wave * 4 + piece - 2) * 512]));
    };
    const int l32 = lane & 31, kh = lane >> 5;
    int offA[2], offB[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int r = wn * 128 + ct * 32 + l32;
        offB[ct] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
    bf16x8 fa0[2], fb0[4], fa1[2], fb1[4];
    auto ld = [&](bf16x8 (&fa)[2], bf16x8 (&fb)[4], int st, int g) {
        const char* Ab = reinterpret_cast<const char*>(sm.A[st]);
        const char* Bb = reinterpret_cast<const char*>(sm.B[st]);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const bf16x8*>(Ab + (offA[rt] ^ (g << 5)));
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(Bb + (offB[ct] ^ (g << 5)));
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma1 = [&](const bf16x8 (&fa)[2], const bf16x8 (&fb)[4], int m) {
        const int rt = m & 1, ct = m >> 1;
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
    };
    const int nch = K / BBK;
    auto clampf = [&](int f) { return f < nch ? f : nch - 1; };
#pragma unroll
    for (int c3 = 0; c3 < 3; ++c3)
#pragma unroll
        for (int p = 0; p < 6; ++p) dma(c3, clampf(c3), p);
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    __syncthreads();
    ld(fa0, fb0, 0, 0);
    int st = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const int stn = st == 2 ? 0 : st + 1;
        mma1(fa0, fb0, 0);
        SB();
        ld(fa1, fb1, st, 1);
        SB();
#pragma unroll
        for (int m = 1; m < 8; ++m) mma1(fa0, fb0, m);
        SB();
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // chunk ch+1 has landed (chunk ch+2 may still be in flight)
        __syncthreads();                                   // every read of stage st has completed in every wave
        ld(fa0, fb0, stn, 0);
        SB();
        const int f = clampf(ch + 3);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            mma1(fa1, fb1, m);
            SB();
            if (m < 6) dma(st, f, m);
            SB();
        }
        st = stn;
    }
    DMA_WAIT();
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 64 + rt * 32 + acc_row(r, lane);
            if (EPI >= 1 ? (M < 0) : (m < M)) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) C[m * ldc + n0 + wn * 128 + ct * 32 + l32] = acc[rt][ct][r];
            }
        }
}
__global__ __launch_bounds__(256) void nt128p(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                              float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNT sm;
    nt128p_body<0>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}
__global__ __launch_bounds__(256) void nt128p_mmaonly(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                      float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNT sm;
    nt128p_body<3>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}
__global__ __launch_bounds__(256) void nt128p_noepi(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                    float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNT sm;
    nt128p_body<1>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}
__global__ __launch_bounds__(256) void nt128pr(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                               float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNTS<3> sm;
    nt128pr_body<0>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}
__global__ __launch_bounds__(256) void nt128pr_noepi(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                     float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNTS<3> sm;
    nt128pr_body<1>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}

// ------------------------------------------------------------------------------------------------------------------
// v3: 128 x 256 x 32, 4 waves, NS-stage LDS ring (NS x 24 KiB): the LDS-DMA of chunk ch + NS - 1 is issued when chunk ch starts,
// so NS - 1 chunks (not one) cover the global->LDS latency.  With v_mfma_f32_32x32x16_bf16 a 32-deep chunk is only 16 x 32 =
// 512 MFMA cycles (~0.2 us) of work per wave: one chunk of lookahead (v0) leaves every chunk waiting ~1 us for its operands.
template <int EPI, int NS>
__device__ __forceinline__ void nt128s_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                            float* __restrict__ C, int64_t ldc, int64_t M, int N, int K, SmemNTS<NS>& sm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ncol = N / BBN;
    const int nt = blockIdx.x % ncol;
    const int64_t m0 = (int64_t)(blockIdx.x / ncol) * BBM;
    const int n0 = nt * BBN;
    const char* baseA = reinterpret_cast<const char*>(A + m0 * lda);
    const char* baseB = reinterpret_cast<const char*>(B + (int64_t)n0 * ldb);
    uint32_t voA[2], voB[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sl = (wave * 2 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        int64_t r = row;
        if (m0 + r > M - 1) r = M - 1 - m0;
        voA[q] = (uint32_t)(r * lda * 2 + kq * 16);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sl = (wave * 4 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        voB[q] = (uint32_t)((int64_t)row * ldb * 2 + kq * 16);
    }
    const uint32_t ldsA = lds_addr_of(&sm.A[0][0]) + wave * 2 * 1024, ldsB = lds_addr_of(&sm.B[0][0]) + wave * 4 * 1024;
    auto issue = [&](int st, int f) {
        const char* a = baseA + (int64_t)f * (BBK * 2);
        const char* b = baseB + (int64_t)f * (BBK * 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16_s(voA[q], a, ldsA + st * (BBM * BBK * 2) + q * 1024);
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16_s(voB[q], b, ldsB + st * (BBN * BBK * 2) + q * 1024);
    };
    const int l32 = lane & 31, kh = lane >> 5;
    int offA[2], offB[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int r = wn * 128 + ct * 32 + l32;
        offB[ct] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nch = K / BBK;
    // prologue: chunks 0 .. NS-2 in flight (indices past the end re-fetch the last chunk: harmless, keeps vmcnt uniform)
#pragma unroll
    for (int p = 0; p < NS - 1; ++p) issue(p, p < nch ? p : nch - 1);
    int st = 0;
    for (int ch = 0; ch < nch; ++ch) {
        // this wave's pieces of chunk ch have landed once at most (NS - 2) x 6 younger DMA instructions are outstanding
        if (NS == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NS == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if (NS == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        if (EPI != 4) __syncthreads();   // every wave's pieces of chunk ch are in LDS; every wave is done reading the stage of chunk ch - 1
        if (EPI != 3 && EPI != 4) {   // EPI 3: no LDS-DMA in the loop (operands stale): ds_read + MFMA rate alone
            const int f = ch + NS - 1 < nch ? ch + NS - 1 : nch - 1;
            int sn = st + NS - 1;
            if (sn >= NS) sn -= NS;
            issue(sn, f);
        }
        const char* Ab = reinterpret_cast<const char*>(sm.A[0]) + st * (BBM * BBK * 2);
        const char* Bb = reinterpret_cast<const char*>(sm.B[0]) + st * (BBN * BBK * 2);
        if (EPI != 2)     // EPI 2: no ds_read / MFMA: the global -> LDS feed rate alone
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const bf16x8*>(Ab + (offA[rt] ^ (g << 5)));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(Bb + (offB[ct] ^ (g << 5)));
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int rt = m & 1, ct = m >> 1;
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
            }
        }
        st = st + 1 == NS ? 0 : st + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 64 + rt * 32 + acc_row(r, lane);
            if (EPI == 1 ? (M < 0) : (m < M)) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) C[m * ldc + n0 + wn * 128 + ct * 32 + l32] = acc[rt][ct][r];
            }
        }
}
// v4: warp-specialised 128 x 256 x 32: 8 waves = 4 consumer waves (ds_read + MFMA only) + 4 producer waves (LDS-DMA only), 3-stage
// ring, one workgroup barrier per chunk.  The ~6 x (60..100)-cycle issue cost of a chunk's LDS-DMA pieces leaves the MFMA waves'
// instruction streams; producer p issues exactly the pieces consumer p issued in v0/v3.
template <int EPI>
__device__ __forceinline__ void nt128ws_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                             float* __restrict__ C, int64_t ldc, int64_t M, int N, int K, SmemNTS<3>& sm) {
    constexpr int NS = 3;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave8 >= 4;
    const int wave = wave8 & 3;
    const int wm = wave >> 1, wn = wave & 1;
    const int ncol = N / BBN;
    const int nt = blockIdx.x % ncol;
    const int64_t m0 = (int64_t)(blockIdx.x / ncol) * BBM;
    const int n0 = nt * BBN;
    const int nch = K / BBK;
    if (producer) {
        const char* baseA = reinterpret_cast<const char*>(A + m0 * lda);
        const char* baseB = reinterpret_cast<const char*>(B + (int64_t)n0 * ldb);
        uint32_t voA[2], voB[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int sl = (wave * 2 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
            int64_t r = row;
            if (m0 + r > M - 1) r = M - 1 - m0;
            voA[q] = (uint32_t)(r * lda * 2 + kq * 16);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int sl = (wave * 4 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
            voB[q] = (uint32_t)((int64_t)row * ldb * 2 + kq * 16);
        }
        const uint32_t ldsA = lds_addr_of(&sm.A[0][0]) + wave * 2 * 1024, ldsB = lds_addr_of(&sm.B[0][0]) + wave * 4 * 1024;
        auto issue = [&](int st, int f) {
            const char* a = baseA + (int64_t)f * (BBK * 2);
            const char* b = baseB + (int64_t)f * (BBK * 2);
#pragma unroll
            for (int q = 0; q < 2; ++q) glds16_s(voA[q], a, ldsA + st * (BBM * BBK * 2) + q * 1024);
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16_s(voB[q], b, ldsB + st * (BBN * BBK * 2) + q * 1024);
        };
        issue(0, 0);
        issue(1, 1 < nch ? 1 : nch - 1);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int sn = 2;
        for (int ch = 0; ch < nch; ++ch) {
            issue(sn, ch + 2 < nch ? ch + 2 : nch - 1);
            sn = sn + 1 == NS ? 0 : sn + 1;
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const int l32 = lane & 31, kh = lane >> 5;
    int offA[2], offB[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int r = wn * 128 + ct * 32 + l32;
        offB[ct] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    int st = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const char* Ab = reinterpret_cast<const char*>(sm.A[0]) + st * (BBM * BBK * 2);
        const char* Bb = reinterpret_cast<const char*>(sm.B[0]) + st * (BBN * BBK * 2);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const bf16x8*>(Ab + (offA[rt] ^ (g << 5)));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(Bb + (offB[ct] ^ (g << 5)));
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int rt = m & 1, ct = m >> 1;
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
            }
        }
        st = st + 1 == NS ? 0 : st + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 64 + rt * 32 + acc_row(r, lane);
            if (EPI == 1 ? (M < 0) : (m < M)) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) C[m * ldc + n0 + wn * 128 + ct * 32 + l32] = acc[rt][ct][r];
            }
        }
}
__global__ __launch_bounds__(512) void nt128ws(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                               float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNTS<3> sm;
    nt128ws_body<0>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}
__global__ __launch_bounds__(512) void nt128ws_noepi(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                     float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNTS<3> sm;
    nt128ws_body<1>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}

// v5: v4 + software-pipelined consumers + a 4-stage ring that runs one chunk further ahead: chunk ch+1 has landed before the
// barrier that ENDS chunk ch-1, so a consumer requests the first fragments of the next chunk while the current chunk's MFMAs run
// and never waits for LDS behind a barrier.
template <int EPI>
__device__ __forceinline__ void nt128wsp_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                              float* __restrict__ C, int64_t ldc, int64_t M, int N, int K, SmemNTS<4>& sm) {
    constexpr int NS = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave8 >= 4;
    const int wave = wave8 & 3;
    const int wm = wave >> 1, wn = wave & 1;
    const int ncol = N / BBN;
    const int nt = blockIdx.x % ncol;
    const int64_t m0 = (int64_t)(blockIdx.x / ncol) * BBM;
    const int n0 = nt * BBN;
    const int nch = K / BBK;
    if (producer) {
        const char* baseA = reinterpret_cast<const char*>(A + m0 * lda);
        const char* baseB = reinterpret_cast<const char*>(B + (int64_t)n0 * ldb);
        uint32_t voA[2], voB[4];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int sl = (wave * 2 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
            int64_t r = row;
            if (m0 + r > M - 1) r = M - 1 - m0;
            voA[q] = (uint32_t)(r * lda * 2 + kq * 16);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int sl = (wave * 4 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
            voB[q] = (uint32_t)((int64_t)row * ldb * 2 + kq * 16);
        }
        const uint32_t ldsA = lds_addr_of(&sm.A[0][0]) + wave * 2 * 1024, ldsB = lds_addr_of(&sm.B[0][0]) + wave * 4 * 1024;
        auto issue = [&](int st, int f) {
            const char* a = baseA + (int64_t)f * (BBK * 2);
            const char* b = baseB + (int64_t)f * (BBK * 2);
#pragma unroll
            for (int q = 0; q < 2; ++q) glds16_s(voA[q], a, ldsA + st * (BBM * BBK * 2) + q * 1024);
#pragma unroll
            for (int q = 0; q < 4; ++q) glds16_s(voB[q], b, ldsB + st * (BBN * BBK * 2) + q * 1024);
        };
        auto clampf = [&](int f) { return f < nch ? f : nch - 1; };
        issue(0, 0);
        issue(1, clampf(1));
        issue(2, clampf(2));
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int sn = 3;
        for (int ch = 0; ch < nch; ++ch) {
            issue(sn, clampf(ch + 3));
            sn = sn + 1 == NS ? 0 : sn + 1;
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    const int l32 = lane & 31, kh = lane >> 5;
    int offA[2], offB[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int r = wn * 128 + ct * 32 + l32;
        offB[ct] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
    bf16x8 fa0[2], fb0[4], fa1[2], fb1[4];
    auto ld = [&](bf16x8 (&fa)[2], bf16x8 (&fb)[4], int st, int g) {
        const char* Ab = reinterpret_cast<const char*>(sm.A[0]) + st * (BBM * BBK * 2);
        const char* Bb = reinterpret_cast<const char*>(sm.B[0]) + st * (BBN * BBK * 2);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const bf16x8*>(Ab + (offA[rt] ^ (g << 5)));
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(Bb + (offB[ct] ^ (g << 5)));
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma1 = [&](const bf16x8 (&fa)[2], const bf16x8 (&fb)[4], int m) {
        const int rt = m & 1, ct = m >> 1;
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
    };
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ld(fa0, fb0, 0, 0);
    int st = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const int stn = st + 1 == NS ? 0 : st + 1;
        mma1(fa0, fb0, 0);
        SB();
        ld(fa1, fb1, st, 1);
        SB();
#pragma unroll
        for (int m = 1; m < 8; ++m) mma1(fa0, fb0, m);
        SB();
        mma1(fa1, fb1, 0);
        SB();
        ld(fa0, fb0, stn, 0);   // chunk ch+1 landed before the previous barrier (past the end: a stale stage, never used)
        SB();
#pragma unroll
        for (int m = 1; m < 8; ++m) mma1(fa1, fb1, m);
        SB();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        st = stn;
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 64 + rt * 32 + acc_row(r, lane);
            if (EPI == 1 ? (M < 0) : (m < M)) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) C[m * ldc + n0 + wn * 128 + ct * 32 + l32] = acc[rt][ct][r];
            }
        }
}
__global__ __launch_bounds__(512) void nt128wsp(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNTS<4> sm;
    nt128wsp_body<0>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}
__global__ __launch_bounds__(512) void nt128wsp_noepi(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
                                                      int64_t ldb, float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNTS<4> sm;
    nt128wsp_body<1>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}

// v7: v3 with NS = 4 but the LDS-DMA issued in PAIRS of chunks: chunk 2k and 2k+1 are the two 64-B halves of the same 128-B cache
// lines; issued back to back from the same wave their requests can merge in the L1 (one line crossing of the L2->CU fabric instead
// of two, cf. tools/micro/feed_rate.hip).  Even iterations: wait for the pair in flight, barrier, issue the next pair; odd: barrier.
template <int EPI>
__device__ __forceinline__ void nt128pair_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                               float* __restrict__ C, int64_t ldc, int64_t M, int N, int K, SmemNTS<4>& sm) {
    constexpr int NS = 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ncol = N / BBN;
    const int nt = blockIdx.x % ncol;
    const int64_t m0 = (int64_t)(blockIdx.x / ncol) * BBM;
    const int n0 = nt * BBN;
    const char* baseA = reinterpret_cast<const char*>(A + m0 * lda);
    const char* baseB = reinterpret_cast<const char*>(B + (int64_t)n0 * ldb);
    uint32_t voA[2], voB[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sl = (wave * 2 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        int64_t r = row;
        if (m0 + r > M - 1) r = M - 1 - m0;
        voA[q] = (uint32_t)(r * lda * 2 + kq * 16);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sl = (wave * 4 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        voB[q] = (uint32_t)((int64_t)row * ldb * 2 + kq * 16);
    }
    const uint32_t ldsA = lds_addr_of(&sm.A[0][0]) + wave * 2 * 1024, ldsB = lds_addr_of(&sm.B[0][0]) + wave * 4 * 1024;
    const int nch = K / BBK;   // even
    auto issue_pair = [&](int st0, int f0) {   // chunks f0, f0+1 -> stages st0, st0+1 (f0 even, st0 in {0, 2})
        const int f = f0 < nch ? f0 : nch - 2;
        const char* a = baseA + (int64_t)f * (BBK * 2);
        const char* b = baseB + (int64_t)f * (BBK * 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            glds16_s(voA[q], a, ldsA + st0 * (BBM * BBK * 2) + q * 1024);
            glds16_s(voA[q], a + BBK * 2, ldsA + (st0 + 1) * (BBM * BBK * 2) + q * 1024);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            glds16_s(voB[q], b, ldsB + st0 * (BBN * BBK * 2) + q * 1024);
            glds16_s(voB[q], b + BBK * 2, ldsB + (st0 + 1) * (BBN * BBK * 2) + q * 1024);
        }
    };
    const int l32 = lane & 31, kh = lane >> 5;
    int offA[2], offB[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int r = wn * 128 + ct * 32 + l32;
        offB[ct] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    issue_pair(0, 0);
    int st = 0;
    for (int ch = 0; ch < nch; ++ch) {
        if ((ch & 1) == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pair (ch, ch+1) has landed
        __syncthreads();
        if ((ch & 1) == 0) issue_pair(st ^ 2, ch + 2);                        // stages of the pair consumed two iterations ago
        const char* Ab = reinterpret_cast<const char*>(sm.A[0]) + st * (BBM * BBK * 2);
        const char* Bb = reinterpret_cast<const char*>(sm.B[0]) + st * (BBN * BBK * 2);
        if (EPI != 2)
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            bf16x8 fa[2], fb[4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const bf16x8*>(Ab + (offA[rt] ^ (g << 5)));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(Bb + (offB[ct] ^ (g << 5)));
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int rt = m & 1, ct = m >> 1;
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
            }
        }
        st = (st + 1) & 3;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 64 + rt * 32 + acc_row(r, lane);
            if (EPI >= 1 ? (M < 0) : (m < M)) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) C[m * ldc + n0 + wn * 128 + ct * 32 + l32] = acc[rt][ct][r];
            }
        }
}
__global__ __launch_bounds__(256) void nt128pair(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                 float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNTS<4> sm;
    nt128pair_body<0>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}
__global__ __launch_bounds__(256) void nt128pair_noepi(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                                       float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNTS<4> sm;
    nt128pair_body<1>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}
__global__ __launch_bounds__(256) void nt128pair_dmaonly(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
                                                         int64_t ldb, float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) {
    __shared__ SmemNTS<4> sm;
    nt128pair_body<2>(A, lda, B, ldb, C, ldc, M, N, K, sm);
}

#define NT128S(NAME, EPI, NS, OCC)                                                                                              \
    __global__ __launch_bounds__(256, OCC) void NAME(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,   \
                                                     int64_t ldb, float* __restrict__ C, int64_t ldc, int64_t M, int N, int K) { \
        __shared__ SmemNTS<NS> sm;                                                                                              \
        nt128s_body<EPI, NS>(A, lda, B, ldb, C, ldc, M, N, K, sm);                                                              \
    }
NT128S(nt128s3, 0, 3, 2)
NT128S(nt128s3_noepi, 1, 3, 2)
NT128S(nt128s4_noepi, 1, 4, 1)
NT128S(nt128s6_noepi, 1, 6, 1)
NT128S(nt128s2_noepi, 1, 2, 2)
NT128S(nt128s2_dmaonly, 2, 2, 2)
NT128S(nt128s2_mmaonly, 3, 2, 2)
NT128S(nt128s3_dmaonly, 2, 3, 2)
NT128S(nt128s2_mmaonly_nobar, 4, 2, 2)

// ------------------------------------------------------------------------------------------------------------------
// TN: C[i, n] = sum_k A[k][i] B[k][n] with BOTH operands K-major in memory (token-major activations: the dW products) --
// no transposed copies: the MFMA fragments (8 consecutive k per lane) are gathered from the K-major LDS image by
// ds_read_b64_tr_b16.  Measured semantics (tools/micro/tr_probe.hip): within a 16-lane group every lane r supplies the
// address of 4 consecutive bf16 D[r][0..3]; lane l receives D[4j + (l >> 2)][l & 3], j = 0..3.  With lane r pointing at
// tile[k0 + (r >> 2)][i0 + 4 (r & 3)] lane l gets tile[k0 + j][i0 + l]: 4 consecutive k of "its" row.
// LDS image: [32 k][W] bf16 (W = 128 for A, 256 for B); 64-B unit u of k-row k is stored at unit u ^ (k & 3) (source-side
// swizzle of the LDS-DMA) so that the 4 k-rows of a read hit 4 different bank windows.
constexpr int TK = 32;
struct __attribute__((aligned(16))) SmemTN {
    bf16_t A[2][TK * 128];   // 8 KiB per stage
    bf16_t B[2][TK * 256];   // 16 KiB per stage
};
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u32x2 ds_tr16(uint32_t lds_byte_addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_byte_addr));
    return v;
}
template <int EPI, int SWZ>
__device__ __forceinline__ void tn128_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                           float* __restrict__ C, int64_t ldc, int64_t Kt, int Mi, int N, int64_t k_per_split,
                                           SmemTN& sm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nit = Mi / 128, nnt = N / 256;
    const int it = blockIdx.x % nit, nt = (blockIdx.x / nit) % nnt, sp = blockIdx.x / (nit * nnt);
    const int i0 = it * 128, n0 = nt * 256;
    const int64_t ks = (int64_t)sp * k_per_split;
    int64_t ke = ks + k_per_split;
    if (ke > Kt) ke = Kt;
    const int nch = (int)((ke - ks + TK - 1) / TK);
    // DMA: A: instruction q of wave w = k-rows 4(2w+q) .. +3, 256 B each (16 lanes per row); B: instruction q = k-rows 2(4w+q), +1
    const char* baseA = reinterpret_cast<const char*>(A + ks * lda + i0);
    const char* baseB = reinterpret_cast<const char*>(B + ks * ldb + n0);
    uint32_t voA[2], voB[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int kr = (wave * 2 + q) * 4 + (lane >> 4), c = lane & 15;          // 16-B chunk c' of k-row kr
        const int src = SWZ ? (c ^ ((kr & 3) << 2)) : c;
        voA[q] = (uint32_t)((int64_t)kr * lda * 2 + src * 16);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int kr = (wave * 4 + q) * 2 + (lane >> 5), c = lane & 31;
        const int src = SWZ ? (c ^ ((kr & 3) << 2)) : c;
        voB[q] = (uint32_t)((int64_t)kr * ldb * 2 + src * 16);
    }
    auto dma = [&](int st, int f, int piece) {
        if (piece < 2) glds16_s(voA[piece], baseA + (int64_t)f * TK * lda * 2, lds_addr_of(&sm.A[st][(wave * 2 + piece) * 512]));
        else glds16_s(voB[piece - 2], baseB + (int64_t)f * TK * ldb * 2, lds_addr_of(&sm.B[st][(wave * 4 + piece - 2) * 512]));
    };
    // fragment addresses: lane = 16 g + r; rows/cols of its group: col block (g & 1) * 16, k half g >> 1
    const int g = lane >> 4, r = lane & 15;
    const uint32_t ldsA = lds_addr_of(&sm.A[0][0]), ldsB = lds_addr_of(&sm.B[0][0]);
    auto addrA = [&](int st, int rt, int ks16, int h) {   // rows wm*64 + rt*32 .. ; k = ks16*16 + (g>>1)*8 + h*4 + (r>>2)
        const int k = ks16 * 16 + (g >> 1) * 8 + h * 4 + (r >> 2);
        int col = wm * 64 + rt * 32 + (g & 1) * 16 + (r & 3) * 4;           // element index within the 128-wide row
        int byte = col * 2;
        if (SWZ) byte ^= (k & 3) << 6;
        return ldsA + st * (TK * 128 * 2) + k * 256 + byte;
    };
    auto addrB = [&](int st, int ct, int ks16, int h) {
        const int k = ks16 * 16 + (g >> 1) * 8 + h * 4 + (r >> 2);
        int col = wn * 128 + ct * 32 + (g & 1) * 16 + (r & 3) * 4;
        int byte = col * 2;
        if (SWZ) byte ^= (k & 3) << 6;
        return ldsB + st * (TK * 256 * 2) + k * 512 + byte;
    };
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (nch > 0) {
#pragma unroll
        for (int p = 0; p < 6; ++p) dma(0, 0, p);
    }
    DMA_WAIT();
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int st = ch & 1;
        if (ch + 1 < nch) {
#pragma unroll
            for (int p = 0; p < 6; ++p) dma(st ^ 1, ch + 1, p);
        }
#pragma unroll
        for (int ks16 = 0; ks16 < 2; ++ks16) {
            u32x2 a[2][2], b[4][2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int h = 0; h < 2; ++h) a[rt][h] = ds_tr16(addrA(st, rt, ks16, h));
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int h = 0; h < 2; ++h) b[ct][h] = ds_tr16(addrB(st, ct, ks16, h));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SB();
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int rt = m & 1, ct = m >> 1;
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 av = {a[rt][0].x, a[rt][0].y, a[rt][1].x, a[rt][1].y};
                const u32x4 bv = {b[ct][0].x, b[ct][0].y, b[ct][1].x, b[ct][1].y};
                acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv),
                                                                      acc[rt][ct], 0, 0, 0);
            }
        }
        DMA_WAIT();
        __syncthreads();
    }
    const int l32 = lane & 31;
    float* __restrict__ so = C + (int64_t)sp * Mi * ldc;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wm * 64 + rt * 32 + acc_row(e, lane);
            if (EPI == 1 ? (Kt < 0) : true) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) so[(int64_t)i * ldc + n0 + wn * 128 + ct * 32 + l32] = acc[rt][ct][e];
            }
        }
}
__global__ __launch_bounds__(256, 2) void tn128(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, float* C, int64_t ldc,
                                                int64_t Kt, int Mi, int N, int64_t kps) {
    __shared__ SmemTN sm;
    tn128_body<0, 1>(A, lda, B, ldb, C, ldc, Kt, Mi, N, kps, sm);
}
__global__ __launch_bounds__(256, 2) void tn128_noswz(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, float* C, int64_t ldc,
                                                      int64_t Kt, int Mi, int N, int64_t kps) {
    __shared__ SmemTN sm;
    tn128_body<0, 0>(A, lda, B, ldb, C, ldc, Kt, Mi, N, kps, sm);
}

// TN on the 256 x 256 x 64 tile: 8 waves (2 x 4, 128 x 64 each), K-major stage images [64 k][256] bf16 (512-B k-rows, 64-B unit u of
// k-row t at unit u ^ (t & 3)), fragments by ds_read_b64_tr_b16, in-wave pipelined like nt256 (asm reads: lgkmcnt waits by hand).
constexpr int UK = 64;
struct __attribute__((aligned(16))) SmemTQ {
    char A[2][UK * 256 * 2];   // 32 KiB per stage
    char B[2][UK * 256 * 2];
};
template <int OFF>
__device__ __forceinline__ u32x2 ds_tr16o(uint32_t a) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(OFF));
    return v;
}
struct TqFrag {
    u32x2 a[4][2], b[2][2];
};
template <int KS>   // k-step 0..3 of the staged chunk: + KS * 16 k-rows = KS * 8192 B; second k half: + 4 rows = 2048 B
__device__ __forceinline__ void tq_load(TqFrag& f, const uint32_t (&aA)[4], const uint32_t (&aB)[2]) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        f.a[rt][0] = ds_tr16o<KS * 8192>(aA[rt]);
        f.a[rt][1] = ds_tr16o<KS * 8192 + 2048>(aA[rt]);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        f.b[ct][0] = ds_tr16o<KS * 8192>(aB[ct]);
        f.b[ct][1] = ds_tr16o<KS * 8192 + 2048>(aB[ct]);
    }
}
__device__ __forceinline__ void tq_mma(f32x16 (&acc)[4][2], const TqFrag& f, int m) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int rt = m >> 1, ct = m & 1;
    const u32x4 av = {f.a[rt][0].x, f.a[rt][0].y, f.a[rt][1].x, f.a[rt][1].y};
    const u32x4 bv = {f.b[ct][0].x, f.b[ct][0].y, f.b[ct][1].x, f.b[ct][1].y};
    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[rt][ct], 0, 0, 0);
}
template <int EPI>
__device__ __forceinline__ void tn256_body(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B, int64_t ldb,
                                           float* __restrict__ C, int64_t ldc, int64_t Kt, int Mi, int N, int64_t k_per_split,
                                           SmemTQ& sm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int nit = Mi / 256, nnt = N / 256;
    const int it = blockIdx.x % nit, nt = (blockIdx.x / nit) % nnt, sp = blockIdx.x / (nit * nnt);
    const int i0 = it * 256, n0 = nt * 256;
    const int64_t ks = (int64_t)sp * k_per_split;
    int64_t ke = ks + k_per_split;
    if (ke > Kt) ke = Kt;
    const int nch = (int)((ke - ks + UK - 1) / UK);
    // DMA: one instruction = 2 k-rows x 512 B; wave w issues k-row pairs 4w .. 4w+3 of each operand (32 pairs per stage)
    const char* baseA = reinterpret_cast<const char*>(A + ks * lda + i0);
    const char* baseB = reinterpret_cast<const char*>(B + ks * ldb + n0);
    uint32_t voA[4], voB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int kr = (wave * 4 + q) * 2 + (lane >> 5), c = lane & 31;   // 16-B chunk position c of k-row kr holds global chunk c ^ ((kr&3)<<2)
        const int src = c ^ ((kr & 3) << 2);
        voA[q] = (uint32_t)((int64_t)kr * lda * 2 + src * 16);
        voB[q] = (uint32_t)((int64_t)kr * ldb * 2 + src * 16);
    }
    auto dma = [&](int st, int f, int piece) {
        const int q = piece & 3;
        if (piece < 4) glds16_s(voA[q], baseA + (int64_t)f * UK * lda * 2, lds_addr_of(&sm.A[st][(wave * 4 + q) * 1024]));
        else glds16_s(voB[q], baseB + (int64_t)f * UK * ldb * 2, lds_addr_of(&sm.B[st][(wave * 4 + q) * 1024]));
    };
    const int g = lane >> 4, r = lane & 15;
    const int kb = (g >> 1) * 8 + (r >> 2);
    uint32_t a0[4], b0[2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
        a0[rt] = lds_addr_of(&sm.A[0][0]) + kb * 512 + (((wm * 128 + rt * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        b0[ct] = lds_addr_of(&sm.B[0][0]) + kb * 512 + (((wn * 64 + ct * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (nch > 0) {
#pragma unroll
        for (int p = 0; p < 8; ++p) dma(0, 0, p);
        DMA_WAIT();
        __syncthreads();
        {
            const int f = nch > 1 ? 1 : 0;
#pragma unroll
            for (int p = 0; p < 8; ++p) dma(1, f, p);
        }
        TqFrag f0, f1;
        tq_load<0>(f0, a0, b0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SB();
        for (int ch = 0; ch < nch; ++ch) {
            const int st = ch & 1;
            uint32_t aA[4], aB[2], nA[4], nB[2];
#pragma unroll
            for (int rt = 0; rt < 4; ++rt) {
                aA[rt] = a0[rt] + st * (UK * 512);
                nA[rt] = a0[rt] + (st ^ 1) * (UK * 512);
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                aB[ct] = b0[ct] + st * (UK * 512);
                nB[ct] = b0[ct] + (st ^ 1) * (UK * 512);
            }
#define TQ_STEP(CUR, NXT, KS)                                                   \
    tq_mma(acc, CUR, 0);                                                        \
    SB();                                                                       \
    tq_load<KS>(NXT, aA, aB);                                                   \
    SB();                                                                       \
    _Pragma("unroll") for (int m = 1; m < 8; ++m) tq_mma(acc, CUR, m);          \
    SB();                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                         \
    SB();
            TQ_STEP(f0, f1, 1)
            TQ_STEP(f1, f0, 2)
            TQ_STEP(f0, f1, 3)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            SB();
            __syncthreads();
            tq_load<0>(f0, nA, nB);
            SB();
            const int f = (ch + 2 < nch) ? ch + 2 : nch - 1;
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                tq_mma(acc, f1, m);
                SB();
                dma(st, f, m);
                SB();
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            SB();
        }
#undef TQ_STEP
        DMA_WAIT();
        __syncthreads();
    }
    const int l32 = lane & 31;
    float* __restrict__ so = C + (int64_t)sp * Mi * ldc;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wm * 128 + rt * 32 + acc_row(e, lane);
            if (EPI == 1 ? (Kt < 0) : true) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) so[(int64_t)i * ldc + n0 + wn * 64 + ct * 32 + l32] = acc[rt][ct][e];
            }
        }
}
__global__ __launch_bounds__(512) void tn256(const bf16_t* A, int64_t lda, const bf16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t Kt,
                                             int Mi, int N, int64_t kps) {
    __shared__ SmemTQ sm;
    tn256_body<0>(A, lda, B, ldb, C, ldc, Kt, Mi, N, kps, sm);
}

typedef void (*kern_t)(const bf16_t*, int64_t, const bf16_t*, int64_t, float*, int64_t, int64_t, int, int);
static uint16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 6;
    const int64_t M = 262144;
    const int N = 1024, K = 512;
    uint16_t *A, *B;
    float* C;
    hipMalloc(&A, M * K * 2);
    hipMalloc(&B, (size_t)N * K * 2);
    hipMalloc(&C, M * N * 4);
    std::vector<uint16_t> hA(1024 * K), hB((size_t)N * K);
    srand(1);
    for (auto& v : hA) v = f2bf((rand() % 2001 - 1000) * 1e-3f);
    for (auto& v : hB) v = f2bf((rand() % 2001 - 1000) * 1e-3f);
    for (int64_t r = 0; r < M; r += 1024) hipMemcpy(A + r * K, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct V { const char* name; kern_t k; int bm, bn, threads; };
    const V vs[] = {{"nt128x256x32", nt128, BBM, BBN, 256}, {"nt256x256x64_pipe", nt256, PM, PN, 512},
                    {"nt128_pipe", nt128p, BBM, BBN, 256}, {"nt128_ring3", nt128s3, BBM, BBN, 256}, {"nt128_wspec", nt128ws, BBM, BBN, 512}, {"nt128_wspec_pipe", nt128wsp, BBM, BBN, 512}, {"nt128_pipe_ring3", nt128pr, BBM, BBN, 256}, {"nt128_pair4", nt128pair, BBM, BBN, 256},
                    {"nt128_noepi", nt128_noepi, BBM, BBN, 256}, {"nt256_noepi", nt256_noepi, PM, PN, 512},
                    {"nt128_pipe_noepi", nt128p_noepi, BBM, BBN, 256}, {"nt128_ring2_noepi", nt128s2_noepi, BBM, BBN, 256},
                    {"nt128_ring3_noepi", nt128s3_noepi, BBM, BBN, 256}, {"nt128_ring4_noepi", nt128s4_noepi, BBM, BBN, 256},
                    {"nt128_ring6_noepi", nt128s6_noepi, BBM, BBN, 256}, {"nt128_wspec_noepi", nt128ws_noepi, BBM, BBN, 512}, {"nt128_wspec_pipe_noepi", nt128wsp_noepi, BBM, BBN, 512}, {"nt128_ring2_DMA_ONLY", nt128s2_dmaonly, BBM, BBN, 256},
                    {"nt128_ring3_DMA_ONLY", nt128s3_dmaonly, BBM, BBN, 256}, {"nt128_ring2_MMA_ONLY", nt128s2_mmaonly, BBM, BBN, 256}, {"nt128_MMA_ONLY_NOBAR", nt128s2_mmaonly_nobar, BBM, BBN, 256}, {"nt128_pipe_MMA_ONLY", nt128p_mmaonly, BBM, BBN, 256}, {"nt128_pipe_ring3_noepi", nt128pr_noepi, BBM, BBN, 256}, {"nt128_pair4_noepi", nt128pair_noepi, BBM, BBN, 256},
                    {"nt128_pair4_DMA_ONLY", nt128pair_dmaonly, BBM, BBN, 256}};
    const int nv = 25;
    for (int v = 0; v < 8; ++v)
        for (int which = 0; which < 2; ++which) {
            const int64_t Mc = which ? 1000 : M;
            const int tiles = (int)(((Mc + vs[v].bm - 1) / vs[v].bm) * (N / vs[v].bn));
            hipMemset(C, 0xff, (size_t)Mc * N * 4);
            hipLaunchKernelGGL(vs[v].k, dim3(tiles), dim3(vs[v].threads), 0, 0, (const bf16_t*)A, (int64_t)K, (const bf16_t*)B, (int64_t)K, C,
                               (int64_t)N, Mc, N, K);
            hipDeviceSynchronize();
            const int64_t r0 = which ? 0 : 777;
            const int nr = which ? 1000 : 8;
            std::vector<float> hC((size_t)nr * N);
            hipMemcpy(hC.data(), C + r0 * N, hC.size() * 4, hipMemcpyDeviceToHost);
            double maxerr = 0;
            for (int r = 0; r < nr; r += (which ? 37 : 1))
                for (int n = 0; n < N; n += 13) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += (double)bf2f(hA[(size_t)((r0 + r) % 1024) * K + k]) * bf2f(hB[(size_t)n * K + k]);
                    maxerr = fmax(maxerr, fabs(s - hC[(size_t)r * N + n]));
                }
            printf("check %-22s M=%-7lld max abs err vs fp64 %.3e %s\n", vs[v].name, (long long)Mc, maxerr, maxerr < 1e-3 ? "OK" : "FAIL");
        }
    std::vector<double> best(nv, 1e9), sum(nv, 0);
    for (int rd = 0; rd < rounds; ++rd)
        for (int v = 0; v < nv; ++v) {
            const int tiles = (int)((M / vs[v].bm) * (N / vs[v].bn));
            hipEventRecord(e0);
            hipLaunchKernelGGL(vs[v].k, dim3(tiles), dim3(vs[v].threads), 0, 0, (const bf16_t*)A, (int64_t)K, (const bf16_t*)B, (int64_t)K, C,
                               (int64_t)N, M, N, K);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rd > 0) { sum[v] += ms; if (ms < best[v]) best[v] = ms; }
        }
    for (int v = 0; v < nv; ++v)
        printf("%-22s mean %.3f ms (%.0f TF)  best %.3f ms (%.0f TF)\n", vs[v].name, sum[v] / (rounds - 1),
               2.0 * M * N * K / (sum[v] / (rounds - 1)) / 1e9, best[v], 2.0 * M * N * K / best[v] / 1e9);
    // ---- TN: the gate dW shape of one head: C[512, 1024] = A[Kt, 512]^T B[Kt, 1024], Kt = 262144 tokens in 72 splits ----
    {
        const int64_t Kt = 262144;
        const int Mi = 512, Nn = 1024, S = 72;
        const int64_t kps = ((Kt + S - 1) / S + TK - 1) / TK * TK;
        uint16_t *At, *Bt;
        float* Ct;
        hipMalloc(&At, Kt * Mi * 2);
        hipMalloc(&Bt, Kt * Nn * 2);
        hipMalloc(&Ct, (size_t)S * Mi * Nn * 4);
        std::vector<uint16_t> hAt(4096 * Mi), hBt((size_t)4096 * Nn);
        for (auto& v : hAt) v = f2bf((rand() % 2001 - 1000) * 1e-3f);
        for (auto& v : hBt) v = f2bf((rand() % 2001 - 1000) * 1e-3f);
        for (int64_t r = 0; r < Kt; r += 4096) {
            hipMemcpy(At + r * Mi, hAt.data(), hAt.size() * 2, hipMemcpyHostToDevice);
            hipMemcpy(Bt + r * Nn, hBt.data(), hBt.size() * 2, hipMemcpyHostToDevice);
        }
        typedef void (*tn_t)(const bf16_t*, int64_t, const bf16_t*, int64_t, float*, int64_t, int64_t, int, int, int64_t);
        struct TV { const char* name; tn_t k; };
        const TV tv[] = {{"tn128_swz", tn128}, {"tn128_noswz", tn128_noswz}, {"tn256_pipe", tn256}};
        for (int v = 0; v < 3; ++v) {
            const int bm = v == 2 ? 256 : 128, thr = v == 2 ? 512 : 256;
            const int grid = (Mi / bm) * (Nn / 256) * S;
            // correctness on a small K (one split of 96 tokens incl. a partial... K multiple of 32 here)
            hipMemset(Ct, 0xff, (size_t)Mi * Nn * 4);
            hipLaunchKernelGGL(tv[v].k, dim3((Mi / bm) * (Nn / 256)), dim3(thr), 0, 0, (const bf16_t*)At, (int64_t)Mi, (const bf16_t*)Bt,
                               (int64_t)Nn, Ct, (int64_t)Nn, (int64_t)(v == 2 ? 192 : 96), Mi, Nn, (int64_t)(v == 2 ? 192 : 96));
            hipDeviceSynchronize();
            std::vector<float> hC((size_t)Mi * Nn);
            hipMemcpy(hC.data(), Ct, hC.size() * 4, hipMemcpyDeviceToHost);
            double maxerr = 0;
            for (int i = 0; i < Mi; i += 7)
                for (int n = 0; n < Nn; n += 11) {
                    double s2 = 0;
                    for (int k = 0; k < (v == 2 ? 192 : 96); ++k) s2 += (double)bf2f(hAt[(size_t)k * Mi + i]) * bf2f(hBt[(size_t)k * Nn + n]);
                    maxerr = fmax(maxerr, fabs(s2 - hC[(size_t)i * Nn + n]));
                }
            printf("check %-12s K=96 max abs err vs fp64 %.3e %s\n", tv[v].name, maxerr, maxerr < 1e-3 ? "OK" : "FAIL");
            double sum2 = 0, best2 = 1e9;
            for (int rd = 0; rd < rounds; ++rd) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(tv[v].k, dim3(grid), dim3(thr), 0, 0, (const bf16_t*)At, (int64_t)Mi, (const bf16_t*)Bt, (int64_t)Nn, Ct,
                                   (int64_t)Nn, Kt, Mi, Nn, (v == 2 ? (kps + 63) / 64 * 64 : kps));
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rd > 0) { sum2 += ms; if (ms < best2) best2 = ms; }
            }
            printf("%-18s mean %.3f ms (%.0f TF)  best %.3f ms (%.0f TF)\n", tv[v].name, sum2 / (rounds - 1),
                   2.0 * Kt * Mi * Nn / (sum2 / (rounds - 1)) / 1e9, best2, 2.0 * Kt * Mi * Nn / best2 / 1e9);
        }
    }
    return 0;
}
