// split_lab.hip -- round-3 lab: fp32-ACCURATE contractions on the 16-bit matrix cores by operand splitting.
//   x = hi + lo (two fp16 planes, 22-23 significand bits)  ->  a.b ~= ah.bh + ah.bl + al.bh            3 MFMAs (v_mfma_f32_32x32x16_f16)
//   x = hi + mid + lo (three bf16 planes, 24 bits)         ->  ah.bh + ah.bm + am.bh + ah.bl + am.bm + al.bh   6 MFMAs (..._bf16)
// against 8 v_mfma_f32_32x32x2_f32 per 16 k (64 cycles each = 512 cycles; the 16-bit MFMAs are 32 cycles each: 96 / 192).
// "NT" product C[m, n] = sum_k A[m][k] B[n][k] on the 256 x 256 tile of tile_engine_bf16.hpp (8 waves, 128 x 64 per wave), operands as
// SPLIT IMAGES  [row][K / KB][plane slot 0..3][KB 16-bit values]  (KB = 32 for fp16x2: 2 x 64 B = one 128-B line per row and block;
// KB = 16 for bf16x3: 3 x 32 B + 32 B pad), so that one LDS stage row is 128 B and the k-step index of the bf16 engine doubles as the
// plane index.  Prints accuracy against fp64 (and the error of a plain fp32 fmaf chain for scale) and the sustained rate.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 split_lab.hip -o split_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

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

constexpr int PM = 256, PN = 256;
constexpr int STAGE_BYTES = PM * 128;   // 32 KiB per operand per stage (128-B rows)
struct SmemP {
    char A[2][STAGE_BYTES];
    char B[2][STAGE_BYTES];
};

template <int NP>
__device__ __forceinline__ f32x16 mfma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    if (NP == 2) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One chunk = one K block (KB logical k).  LDS row = 8 chunks of 16 B: chunk index = ks * 2 + kh with
//   NP = 2: ks = plane * 2 + s   (s = 16-k half of the 32-k block)      sets: (s | hi,hi) (s | hi,lo) (s | lo,hi), s = 0, 1
//   NP = 3: ks = plane (0..2; 3 = pad)                                  sets: (0,0) (0,1) (1,0) (0,2) (1,1) (2,0)
template <int NP, int EPI, int PF>
__device__ __forceinline__ void split_body(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, float* __restrict__ C,
                                           int64_t ldc, int64_t M, int N, int nblk, SmemP& sm) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int ncol = N / PN;
    const int nt = blockIdx.x % ncol;
    const int64_t m0 = (int64_t)(blockIdx.x / ncol) * PM;
    const int n0 = nt * PN;
    const int l32 = lane & 31, kh = lane >> 5;
    const int64_t row_bytes = (int64_t)nblk * 128;   // image row: nblk blocks of 128 B

    const char* baseA = reinterpret_cast<const char*>(A) + m0 * row_bytes;
    const char* baseB = reinterpret_cast<const char*>(B) + (int64_t)n0 * row_bytes;
    uint32_t voA[4], voB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (wave * 4 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t ra = row;
        if (m0 + ra > M - 1) ra = M - 1 - m0;
        voA[i] = (uint32_t)(ra * row_bytes + c * 16);
        voB[i] = (uint32_t)((int64_t)row * row_bytes + c * 16);
    }
    auto dma = [&](int st, int f, int piece) {
        const int i = piece & 3;
        if (piece < 4) glds16_s(voA[i], baseA + (int64_t)f * 128, lds_addr_of(&sm.A[st][(wave * 4 + i) * 1024]));
        else glds16_s(voB[i], baseB + (int64_t)f * 128, lds_addr_of(&sm.B[st][(wave * 4 + i) * 1024]));
    };
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
    auto ldA = [&](u32x4 (&fa)[4], int st, int ks) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) fa[rt] = *reinterpret_cast<const u32x4*>(&sm.A[st][offA[rt] ^ (ks << 5)]);
    };
    auto ldB = [&](u32x4 (&fb)[2], int st, int ks) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) fb[ct] = *reinterpret_cast<const u32x4*>(&sm.B[st][offB[ct] ^ (ks << 5)]);
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma1 = [&](const u32x4 (&fa)[4], const u32x4 (&fb)[2], int m) {
        const int rt = m >> 1, ct = m & 1;
        acc[rt][ct] = mfma16<NP>(fa[rt], fb[ct], acc[rt][ct]);
    };
#define SET(FA, FB, LOADS)                                                      \
    mma1(FA, FB, 0);                                                            \
    SB();                                                                       \
    LOADS;                                                                      \
    SB();                                                                       \
    _Pragma("unroll") for (int m = 1; m < 8; ++m) mma1(FA, FB, m);              \
    SB();
    // the set whose stage reads are all requested: barrier first, then the next chunk's first fragments + DMA between its MFMAs
#define LASTSET(FA, FB, NEXTLOADS)                                              \
    if (PF) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else DMA_WAIT();   \
    __syncthreads();                                                            \
    NEXTLOADS;                                                                  \
    SB();                                                                       \
    {                                                                           \
        const int f = (ch + 2 < nblk) ? ch + 2 : nblk - 1;                      \
        _Pragma("unroll") for (int m = 0; m < 8; ++m) {                         \
            mma1(FA, FB, m);                                                    \
            SB();                                                               \
            dma(st, f, m);                                                      \
            SB();                                                               \
        }                                                                       \
        if (PF) {   /* touch the lines of block ch + PF (one 128-B line per thread: 256 A rows, 256 B rows) -> L2 */ \
            const int fp = (ch + PF < nblk) ? ch + PF : nblk - 1;               \
            asm volatile("global_load_dword %0, %1, %2" : "+v"(pfreg) : "v"(pfoff), "s"(pfbase + (int64_t)fp * 128) : "memory"); \
        }                                                                       \
    }
    u32x4 a0[4], a1[4], a2[4], b0[2], b1[2], b2[2];
    // prefetch addressing: thread t < 256 touches A row t, t >= 256 touches B row t - 256 (uniform base = A image start)
    uint32_t pfreg = 0;
    // waves 0-3 touch the 256 A rows of the block, waves 4-7 the 256 B rows (wave-uniform base, one 128-B line per thread)
    const char* pfbase;
    uint32_t pfoff;
    {
        int64_t ra = (tid & 255);
        if (wave < 4) { if (m0 + ra > M - 1) ra = M - 1 - m0; pfbase = baseA; pfoff = (uint32_t)(ra * row_bytes); }
        else { pfbase = baseB; pfoff = (uint32_t)(ra * row_bytes); }
        const uint64_t v = reinterpret_cast<uint64_t>(pfbase);
        const uint32_t plo = __builtin_amdgcn_readfirstlane((uint32_t)v), phi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        pfbase = reinterpret_cast<const char*>(((uint64_t)phi << 32) | plo);
    }
    if (PF) asm volatile("global_load_dword %0, %1, %2" : "+v"(pfreg) : "v"(pfoff), "s"(pfbase) : "memory");
#pragma unroll
    for (int p = 0; p < 8; ++p) dma(0, 0, p);
    DMA_WAIT();
    __syncthreads();
    {
        const int f = nblk > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) dma(1, f, p);
    }
    ldA(a0, 0, 0);
    ldB(b0, 0, 0);
    for (int ch = 0; ch < nblk; ++ch) {
        const int st = ch & 1;
        if (NP == 2) {
            // ks: 0 = hi k 0-15, 1 = hi k 16-31, 2 = lo k 0-15, 3 = lo k 16-31.   a0 = A hi s0, b0 = B hi s0 on entry
            SET(a0, b0, ldB(b1, st, 2))                    // hh s0        | B lo s0
            SET(a0, b1, ldA(a1, st, 2))                    // hl s0        | A lo s0
            SET(a1, b0, ldA(a2, st, 1); ldB(b2, st, 1))    // lh s0        | A hi s1, B hi s1
            SET(a2, b2, ldB(b1, st, 3))                    // hh s1        | B lo s1
            SET(a2, b1, ldA(a1, st, 3))                    // hl s1        | A lo s1
            LASTSET(a1, b2, ldA(a0, st ^ 1, 0); ldB(b0, st ^ 1, 0))   // lh s1
        } else {
            // ks = plane: 0 hi, 1 mid, 2 lo.   a0 = A hi, b0 = B hi on entry
            SET(a0, b0, ldB(b1, st, 1))                    // hh           | B mid
            SET(a0, b1, ldA(a1, st, 1))                    // hm           | A mid
            SET(a1, b0, ldB(b2, st, 2))                    // mh           | B lo
            SET(a0, b2, ldA(a2, st, 2))                    // hl           | A lo
            SET(a1, b1, )                                  // mm
            LASTSET(a2, b0, ldA(a0, st ^ 1, 0); ldB(b0, st ^ 1, 0))   // lh  (b0 is re-loaded AFTER the barrier; its last use is this set)
        }
    }
    DMA_WAIT();
    __syncthreads();
    if (PF && pfreg == 0x12345678u && M < 0) C[0] = 1.f;   // keeps the prefetch register allocated to the end
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
template <int NP, int EPI, int PF = 0>
__global__ __launch_bounds__(512) void split_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, float* __restrict__ C,
                                                    int64_t ldc, int64_t M, int N, int nblk) {
    __shared__ __attribute__((aligned(16))) SmemP sm;
    split_body<NP, EPI, PF>(A, B, C, ldc, M, N, nblk, sm);
}

// ------------------------------------------------------------------------------------------------------------------
// Variant W4: the same 256 x 256 x 32 chunk loop with FOUR waves (2 x 2), 128 x 128 = 4 x 4 MFMA tiles per wave (256 accumulator
// registers in AGPRs, one wave per SIMD): 8 fragment reads per 16 MFMAs instead of 6 per 8 -- a third less LDS read traffic per MFMA.
// ------------------------------------------------------------------------------------------------------------------
template <int EPI>
__global__ __launch_bounds__(256) void split_w4_kernel(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, float* __restrict__ C,
                                                       int64_t ldc, int64_t M, int N, int nblk) {
    __shared__ __attribute__((aligned(16))) SmemP sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int ncol = N / PN;
    const int nt = blockIdx.x % ncol;
    const int64_t m0 = (int64_t)(blockIdx.x / ncol) * PM;
    const int n0 = nt * PN;
    const int l32 = lane & 31, kh = lane >> 5;
    const int64_t row_bytes = (int64_t)nblk * 128;
    const char* baseA = reinterpret_cast<const char*>(A) + m0 * row_bytes;
    const char* baseB = reinterpret_cast<const char*>(B) + (int64_t)n0 * row_bytes;
    uint32_t voA[8], voB[8];   // wave w issues row groups 8w .. 8w+7 (8 rows each) of each operand
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (wave * 8 + i) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        int64_t ra = row;
        if (m0 + ra > M - 1) ra = M - 1 - m0;
        voA[i] = (uint32_t)(ra * row_bytes + c * 16);
        voB[i] = (uint32_t)((int64_t)row * row_bytes + c * 16);
    }
    auto dma = [&](int st, int f, int piece) {   // 16 pieces: 0-7 A, 8-15 B
        const int i = piece & 7;
        if (piece < 8) glds16_s(voA[i], baseA + (int64_t)f * 128, lds_addr_of(&sm.A[st][(wave * 8 + i) * 1024]));
        else glds16_s(voB[i], baseB + (int64_t)f * 128, lds_addr_of(&sm.B[st][(wave * 8 + i) * 1024]));
    };
    uint32_t offA[4], offB[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int ra = wm * 128 + t * 32 + l32, rb = wn * 128 + t * 32 + l32;
        offA[t] = ra * 128 + ((kh ^ ((ra >> 1) & 7)) << 4);
        offB[t] = rb * 128 + ((kh ^ ((rb >> 1) & 7)) << 4);
    }
    auto ldA = [&](u32x4 (&fa)[4], int st, int ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fa[t] = *reinterpret_cast<const u32x4*>(&sm.A[st][offA[t] ^ (ks << 5)]);
    };
    auto ldB = [&](u32x4 (&fb)[4], int st, int ks) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[t] = *reinterpret_cast<const u32x4*>(&sm.B[st][offB[t] ^ (ks << 5)]);
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma1 = [&](const u32x4 (&fa)[4], const u32x4 (&fb)[4], int m) {
        const int rt = m >> 2, ct = m & 3;
        acc[rt][ct] = mfma16<2>(fa[rt], fb[ct], acc[rt][ct]);
    };
#define SET4(FA, FB, LOADS)                                                     \
    mma1(FA, FB, 0);                                                            \
    SB();                                                                       \
    LOADS;                                                                      \
    SB();                                                                       \
    _Pragma("unroll") for (int m = 1; m < 16; ++m) mma1(FA, FB, m);             \
    SB();
    u32x4 a0[4], a1[4], a2[4], b0[4], b1[4], b2[4];
#pragma unroll
    for (int p = 0; p < 16; ++p) dma(0, 0, p);
    DMA_WAIT();
    __syncthreads();
    {
        const int f = nblk > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 16; ++p) dma(1, f, p);
    }
    ldA(a0, 0, 0);
    ldB(b0, 0, 0);
    for (int ch = 0; ch < nblk; ++ch) {
        const int st = ch & 1;
        SET4(a0, b0, ldB(b1, st, 2))
        SET4(a0, b1, ldA(a1, st, 2))
        SET4(a1, b0, ldA(a2, st, 1); ldB(b2, st, 1))
        SET4(a2, b2, ldB(b1, st, 3))
        SET4(a2, b1, ldA(a1, st, 3))
        DMA_WAIT();
        __syncthreads();
        ldA(a0, st ^ 1, 0);
        ldB(b0, st ^ 1, 0);
        SB();
        const int f = (ch + 2 < nblk) ? ch + 2 : nblk - 1;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            mma1(a1, b2, m);
            SB();
            dma(st, f, m);
            SB();
        }
    }
    DMA_WAIT();
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * 128 + rt * 32 + acc_row(r, lane);
            if (EPI == 1 ? (M < 0) : (m < M)) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) C[m * ldc + n0 + wn * 128 + ct * 32 + l32] = acc[rt][ct][r];
            }
        }
}

// ---- host-side splitting -------------------------------------------------------------------------------------------------------
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
static uint16_t f2h(float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

// image [rows][nblk][4 slots][KB] with KB = 32 (NP = 2) / 16 (NP = 3)
static void make_image(const std::vector<float>& X, int rows, int K, int NP, float scale, std::vector<uint16_t>& img) {
    const int KB = NP == 2 ? 32 : 16, nblk = K / KB;
    img.assign((size_t)rows * nblk * 64, 0);
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < K; ++k) {
            float x = X[(size_t)r * K + k] * scale;
            uint16_t* blk = &img[((size_t)r * nblk + k / KB) * 64];
            if (NP == 2) {
                const uint16_t h = f2h(x);
                const uint16_t l = f2h(x - h2f(h));
                blk[0 * 32 + k % 32] = h;
                blk[1 * 32 + k % 32] = l;
            } else {
                const uint16_t h = f2bf(x);
                const float r1 = x - bf2f(h);
                const uint16_t m = f2bf(r1);
                const uint16_t l = f2bf(r1 - bf2f(m));
                blk[0 * 16 + k % 16] = h;
                blk[1 * 16 + k % 16] = m;
                blk[2 * 16 + k % 16] = l;
            }
        }
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 6;
    const int64_t M = 262144;
    const int N = argc > 2 ? atoi(argv[2]) : 1024, K = argc > 3 ? atoi(argv[3]) : 512;
    const int RP = 1024;   // distinct A rows (repeated down the matrix)
    std::vector<float> hA((size_t)RP * K), hB((size_t)N * K);
    srand(1);
    // gaussian-ish activations (sum of uniforms) x a few large values; weights ~ U(-1,1)/sqrt(K)
    for (auto& v : hA) { float s = 0; for (int i = 0; i < 4; ++i) s += (rand() % 20001 - 10000) * 1e-4f; v = s * (rand() % 97 == 0 ? 8.f : 1.f); }
    for (auto& v : hB) v = (rand() % 20001 - 10000) * 1e-4f / sqrtf((float)K);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float* C;
    hipMalloc(&C, M * N * 4);
    for (int NP = 2; NP <= 3; ++NP) {
        const int KB = NP == 2 ? 32 : 16, nblk = K / KB;
        std::vector<uint16_t> iA, iB;
        make_image(hA, RP, K, NP, NP == 2 ? 256.f : 1.f, iA);    // fp16: per-tensor power-of-two scales (undone below)
        make_image(hB, N, K, NP, NP == 2 ? 4096.f : 1.f, iB);
        const float unscale = NP == 2 ? 1.f / (256.f * 4096.f) : 1.f;
        uint16_t *A, *B;
        hipMalloc(&A, (size_t)M * nblk * 128);
        hipMalloc(&B, (size_t)N * nblk * 128);
        for (int64_t r = 0; r < M; r += RP) hipMemcpy((char*)A + r * nblk * 128, iA.data(), iA.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(B, iB.data(), iB.size() * 2, hipMemcpyHostToDevice);
        const int tiles = (int)((M / PM) * (N / PN));
        hipMemset(C, 0xff, (size_t)M * N * 4);
        if (NP == 2) hipLaunchKernelGGL((split_kernel<2, 0>), dim3(tiles), dim3(512), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
        else hipLaunchKernelGGL((split_kernel<3, 0>), dim3(tiles), dim3(512), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
        hipDeviceSynchronize();
        const int64_t r0 = 777 + 1024 * 100;
        const int nr = 64;
        std::vector<float> hC((size_t)nr * N);
        hipMemcpy(hC.data(), C + r0 * N, hC.size() * 4, hipMemcpyDeviceToHost);
        double max_rel = 0, sum_rel2 = 0, max_rel32 = 0, sum_rel32 = 0;
        int cnt = 0;
        for (int r = 0; r < nr; ++r)
            for (int n = 0; n < N; n += 7) {
                double s = 0, sabs = 0;
                float f = 0.f;
                const float* a = &hA[(size_t)((r0 + r) % RP) * K];
                const float* b = &hB[(size_t)n * K];
                for (int k = 0; k < K; ++k) { s += (double)a[k] * b[k]; sabs += fabs((double)a[k] * b[k]); f = fmaf(a[k], b[k], f); }
                const double e = fabs((double)hC[(size_t)r * N + n] * unscale - s) / sabs, e32 = fabs((double)f - s) / sabs;
                max_rel = fmax(max_rel, e); sum_rel2 += e * e; max_rel32 = fmax(max_rel32, e32); sum_rel32 += e32 * e32; ++cnt;
            }
        printf("%s: |err| / sum|a_k b_k|  max %.3e rms %.3e    (fp32 fmaf chain: max %.3e rms %.3e; 2^-24 = 5.96e-08)\n",
               NP == 2 ? "fp16x2 (3 MFMAs)" : "bf16x3 (6 MFMAs)", max_rel, sqrt(sum_rel2 / cnt), max_rel32, sqrt(sum_rel32 / cnt));
        double sum = 0, best = 1e9, sum_ne = 0;
        for (int rd = 0; rd < rounds; ++rd)
            for (int epi = 0; epi < 2; ++epi) {
                hipEventRecord(e0);
                if (NP == 2 && epi == 0) hipLaunchKernelGGL((split_kernel<2, 0>), dim3(tiles), dim3(512), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
                if (NP == 2 && epi == 1) hipLaunchKernelGGL((split_kernel<2, 1>), dim3(tiles), dim3(512), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
                if (NP == 3 && epi == 0) hipLaunchKernelGGL((split_kernel<3, 0>), dim3(tiles), dim3(512), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
                if (NP == 3 && epi == 1) hipLaunchKernelGGL((split_kernel<3, 1>), dim3(tiles), dim3(512), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rd > 0) { if (epi == 0) { sum += ms; if (ms < best) best = ms; } else sum_ne += ms; }
            }
        if (NP == 2) {
            for (int pf = 2; pf <= 4; ++pf) {
                double sp = 0;
                for (int rd = 0; rd < rounds; ++rd) {
                    hipEventRecord(e0);
                    if (pf == 2) hipLaunchKernelGGL((split_kernel<2, 0, 2>), dim3(tiles), dim3(512), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
                    if (pf == 3) hipLaunchKernelGGL((split_kernel<2, 0, 3>), dim3(tiles), dim3(512), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
                    if (pf == 4) hipLaunchKernelGGL((split_kernel<2, 0, 4>), dim3(tiles), dim3(512), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rd > 0) sp += ms;
                }
                printf("   L2 touch-prefetch %d blocks ahead: mean %.3f ms = %.0f TF\n", pf, sp / (rounds - 1), 2.0 * M * N * K / (sp / (rounds - 1)) / 1e9);
            }
        }
        if (NP == 2) {
            hipMemset(C, 0xff, (size_t)M * N * 4);
            hipLaunchKernelGGL((split_w4_kernel<0>), dim3(tiles), dim3(256), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
            hipDeviceSynchronize();
            std::vector<float> hC2((size_t)nr * N);
            hipMemcpy(hC2.data(), C + r0 * N, hC2.size() * 4, hipMemcpyDeviceToHost);
            double md = 0;
            for (size_t i = 0; i < hC2.size(); ++i) md = fmax(md, fabs((double)hC2[i] - (double)hC[i]));
            for (int epi = 0; epi < 2; ++epi) {
                double sp = 0;
                for (int rd = 0; rd < rounds; ++rd) {
                    hipEventRecord(e0);
                    if (epi == 0) hipLaunchKernelGGL((split_w4_kernel<0>), dim3(tiles), dim3(256), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
                    else hipLaunchKernelGGL((split_w4_kernel<1>), dim3(tiles), dim3(256), 0, 0, A, B, C, (int64_t)N, M, N, nblk);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    if (rd > 0) sp += ms;
                }
                printf("   W4 (4 waves x 128x128)%s: mean %.3f ms = %.0f TF   (max |diff| vs the 8-wave kernel %.2e)\n", epi ? " no C store" : "",
                       sp / (rounds - 1), 2.0 * M * N * K / (sp / (rounds - 1)) / 1e9, md);
            }
        }
        const double fl = 2.0 * M * N * K, mean = sum / (rounds - 1), mean_ne = sum_ne / (rounds - 1);
        printf("   M=%lld N=%d K=%d: mean %.3f ms = %.0f TF fp32-equivalent (raw MFMA %.2f PF); best %.3f ms (%.0f TF);  no C store: %.3f ms (%.0f TF)\n",
               (long long)M, N, K, mean, fl / mean / 1e9, fl * (NP == 2 ? 3 : 6) / mean / 1e12, best, fl / best / 1e9, mean_ne, fl / mean_ne / 1e9);
        hipFree(A);
        hipFree(B);
    }
    return 0;
}
