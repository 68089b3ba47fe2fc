// tile_engine_bf16.hpp -- the two bf16 tile loops (v_mfma_f32_32x32x16_bf16, fp32 accumulate) shared by the bf16 gate kernels
// (abmil_gate_bf16.hip) and the bf16 Linear kernels (linear_bf16.hip).  Workgroup = 4 waves (2 x 2), tile 128 x 256, each wave
// 64 x 128 = 2 x 4 MFMA tiles (128 accumulator registers); the NT loop also comes 128 x 128 (NCT = 2 column tiles per wave).
//   NT: C[m][n] = sum_k A[m][k] B[n][k]  both operands K-contiguous rows; LDS image = XOR-swizzled 64-B rows, ds_read_b128.
//   TN: C[m][n] = sum_k A[k][m] B[k][n]  both operands K-major; fragments gathered by ds_read_b64_tr_b16.
#pragma once
#include <cstdlib>
#include "gate_common.hpp"

namespace mdl {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// ================================================================================================
// NT
// ================================================================================================
constexpr int BBM = 128, BBN = 256, BBK = 32;

struct __attribute__((aligned(16))) SmemNT {
    bf16_t A[2][BBM * BBK];  // 8 KiB per stage
    bf16_t B[2][BBN * BBK];  // 16 KiB per stage
};

template <int NCT>
__device__ __forceinline__ void zero_acc8(f32x16 (&acc)[2][NCT]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NCT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// byte offsets (within one stage) of the 16-B fragments this lane reads for k-step g = 0; g = 1 is `^ 32`
template <int NCT>
__device__ __forceinline__ void nt_offsets(int wm, const int (&colb)[NCT], int lane, int (&offA)[2], int (&offB)[NCT]) {
    const int l32 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int r = colb[ct] + l32;
        offB[ct] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
}

// the 16 MFMAs of one staged chunk (two k-steps of 16)
template <int NCT>
__device__ __forceinline__ void nt_mma_chunk(const SmemNT& sm, int st, f32x16 (&acc)[2][NCT], const int (&offA)[2],
                                             const int (&offB)[NCT]) {
    const char* Ab = reinterpret_cast<const char*>(sm.A[st]);
    const char* Bb = reinterpret_cast<const char*>(sm.B[st]);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        bf16x8 fa[2], fb[NCT];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const bf16x8*>(Ab + (offA[rt] ^ (g << 5)));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(Bb + (offB[ct] ^ (g << 5)));
#pragma unroll
        for (int m = 0; m < 2 * NCT; ++m) {
            const int rt = m & 1, ct = m >> 1;
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
        }
    }
}

// LDS-DMA slot geometry: instruction q of wave w deposits slots [(nq*w + q)*64, +64); slot s = (row s>>2, stored chunk s&3)
// holds the global chunk (s&3) ^ ((row>>2)&3) of that row.
__device__ __forceinline__ void nt_slot(int instr, int lane, int& row, int& kq) {
    const int sl = instr * 64 + lane;
    row = sl >> 2;
    kq = (sl & 3) ^ ((row >> 2) & 3);
}

template <int NCT, class Issue>
__device__ __forceinline__ void nt_mainloop(SmemNT& sm, f32x16 (&acc)[2][NCT], int64_t nch, Issue&& issue, const int (&offA)[2],
                                            const int (&offB)[NCT]) {
    if (nch > 0) issue(0, (int64_t)0);
    __syncthreads();
    for (int64_t ch = 0; ch < nch; ++ch) {
        const int st = (int)(ch & 1);
        if (ch + 1 < nch) issue(st ^ 1, ch + 1);  // lands in the stage last read before the previous barrier
        nt_mma_chunk(sm, st, acc, offA, offB);
        __syncthreads();  // drains the LDS-DMA of chunk ch+1 (vmcnt) and fences this chunk's reads
    }
}

// ---- NS-stage ring form of the NT loop ----------------------------------------------------------------------------------------
// Round-2 finding (tools/micro/gemm_lab_bf16.hip): what bounds the two-stage loop is the number of operand
// BYTES IN FLIGHT per CU, not issue slots or LDS bandwidth: the global -> LDS latency under load is ~1.7 us, a 32-deep bf16 chunk
// is ~0.2 us of MFMA work, so a workgroup that has one 24-KiB stage in flight is fed at (24 KiB x workgroups per CU) / 1.7 us.
// With two workgroups per CU (the 170-190 VGPR kernels) that is ~7 TB/s = 0.65-0.75 PF.  The ring keeps NS - 1 stages in flight
// per workgroup: the LDS-DMA of chunk ch + NS - 1 is issued when chunk ch starts and each wave waits only for its pieces of chunk
// ch (s_waitcnt vmcnt((NS-2) x pieces): the DMA is inline asm, hipcc would wait for vmcnt(0) at the barrier).
template <int NS>
struct __attribute__((aligned(16))) SmemNTR {
    bf16_t A[NS][BBM * BBK];
    bf16_t B[NS][BBN * BBK];
};
template <int NS, int NCT>
__device__ __forceinline__ void nt_mma_chunk_r(const SmemNTR<NS>& sm, int st, f32x16 (&acc)[2][NCT], const int (&offA)[2],
                                               const int (&offB)[NCT]) {
    const char* Ab = reinterpret_cast<const char*>(sm.A[0]) + st * (BBM * BBK * 2);
    const char* Bb = reinterpret_cast<const char*>(sm.B[0]) + st * (BBN * BBK * 2);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        bf16x8 fa[2], fb[NCT];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const bf16x8*>(Ab + (offA[rt] ^ (g << 5)));
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(Bb + (offB[ct] ^ (g << 5)));
#pragma unroll
        for (int m = 0; m < 2 * NCT; ++m) {
            const int rt = m & 1, ct = m >> 1;
            acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
        }
    }
}
template <int N>
__device__ __forceinline__ void dma_wait_outstanding() {   // s_waitcnt vmcnt(N) for the inline-asm LDS-DMA
    if (N <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
// issue(stage, chunk): this wave's 2 + NCT LDS-DMA pieces (glds16_s) of `chunk` into `stage`; chunk indices past the end are
// clamped by the loop (the re-fetch of the last chunk into a free stage keeps the outstanding count uniform).
template <int NS, int NCT, class Issue>
__device__ __forceinline__ void nt_mainloop_ring(SmemNTR<NS>& sm, f32x16 (&acc)[2][NCT], int64_t nch, Issue&& issue,
                                                 const int (&offA)[2], const int (&offB)[NCT]) {
    if (nch <= 0) return;
#pragma unroll
    for (int p = 0; p < NS - 1; ++p) issue(p, p < nch ? (int64_t)p : nch - 1);
    int st = 0;
    for (int64_t ch = 0; ch < nch; ++ch) {
        dma_wait_outstanding<(NS - 2) * (2 + NCT)>();   // this wave's pieces of chunk ch have landed
        __syncthreads();                                // ... every wave's; and every wave is done reading the stage of chunk ch - 1
        int sn = st + NS - 1;
        if (sn >= NS) sn -= NS;
        issue(sn, ch + NS - 1 < nch ? ch + NS - 1 : nch - 1);
        nt_mma_chunk_r<NS, NCT>(sm, st, acc, offA, offB);
        st = st + 1 == NS ? 0 : st + 1;
    }
    dma_wait_outstanding<0>();   // the clamped re-fetches of the tail must land before the epilogue reuses the staging memory
    __syncthreads();
}

// ---- NT on the 256 x 256 x 64 tile (8 waves) --------------------------------------------------------------------------------------
// Round-2 lab result (tools/micro/gemm_lab_bf16.hip): the only loop variant that moves the ~1 PF ceiling of the
// bf16 engine is the one with twice the MFMA work per staged byte AND whole 128-B cache lines per row and chunk (BK = 64): 1.2 PF
// for the main loop against 1.0.  Workgroup = 8 waves (2 x 4), each 128 x 64 = 4 x 2 MFMA tiles; LDS stage = 256 rows x 128 B per
// operand, 16-B chunk c of row r stored at chunk c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 on 128-B rows); two stages =
// 128 KiB, one workgroup per CU; the in-wave pipeline of the fp32 engine (fragments of k-step s+1 requested behind the first MFMA
// of step s, one barrier per chunk before its last step, the next-but-one chunk's 8 LDS-DMA pieces between that step's MFMAs).
// It wins on long contractions (K >= 1024); on K = 512 one workgroup per CU exposes the tile prologue and epilogue.
// a wave-uniform pointer the compiler must keep in SGPRs (saddr operand of the LDS-DMA; loop-variant bases otherwise land in VGPRs)
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}
constexpr int QM = 256, QN = 256, QK = 64, Q_STAGE = QM * QK * 2;
struct __attribute__((aligned(16))) SmemQ {
    char A[2][Q_STAGE];
    char B[2][Q_STAGE];
};
// Round 6: A ring three stages deep = 160 KiB, the LDS-DMA pieces of a chunk spread over its k-steps instead of issued
// together behind the chunk barrier (nt256_mainloop3 / tn256_mainloop3; the one-tile-per-workgroup kernels).  MADELEINE_BF16_STAGES = 2 | 3.
struct __attribute__((aligned(16))) SmemQ3 {
    char A[3][Q_STAGE];
    char B[2][Q_STAGE];
};
#ifndef MDL_BF16_STAGES
#define MDL_BF16_STAGES 3
#endif
// the Linears' 256-tile kernels separately (MADELEINE_BF16_LIN_STAGES = 2 | 3 | 31 (NT only) | 32 (TN only)): measured at config 2
// (profiles/r06r_*, r06s_*) the gate's dX / dW gain 5 %; of the Linears the dX (NT) gains 1 %, the dW (TN) loses 4 % -- default 31
#ifndef MDL_BF16_LIN_STAGES
#define MDL_BF16_LIN_STAGES 31
#endif
static inline int bf16_lin_stages() {
    static const int v = getenv("MADELEINE_BF16_LIN_STAGES") ? atoi(getenv("MADELEINE_BF16_LIN_STAGES")) : MDL_BF16_LIN_STAGES;
    return v;
}
static inline int bf16_stages() {
    static const int v = getenv("MADELEINE_BF16_STAGES") ? atoi(getenv("MADELEINE_BF16_STAGES")) : MDL_BF16_STAGES;
    return v == 2 ? 2 : 3;
}
// LDS-DMA piece i (0..3) of wave w for either operand: 8 rows x 128 B; this lane fetches global 16-B chunk c of row `row` (tile-relative)
// and the DMA deposits it at LDS slot (wave*4 + i)*1024 + lane*16 = row*128 + (c ^ ((row >> 1) & 7))*16.
__device__ __forceinline__ void nt256_slot(int wave, int i, int lane, int& row, int& c) {
    row = (wave * 4 + i) * 8 + (lane >> 3);
    c = (lane & 7) ^ ((row >> 1) & 7);
}
// dma(stage, chunk, piece): piece 0..3 = this wave's A row blocks, 4..7 = its B row blocks (glds16_s).  acc is zeroed here.
template <class Dma>
// chunk0_in_flight: the caller already issued this wave's 8 pieces of chunk 0 into stage 0 (a persistent workgroup requests the next
// tile's first chunk before the epilogue of the current one, see gate_fwd256_bf16_kernel).
__device__ __forceinline__ void nt256_mainloop(SmemQ& sm, f32x16 (&acc)[4][2], int64_t nch, int wm, int wn, int lane, Dma&& dma,
                                               bool chunk0_in_flight = false) {
    const int l32 = lane & 31, kh = lane >> 5;
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
#define NT256_SB() __builtin_amdgcn_sched_barrier(0)
#define NT256_KSTEP(FA, FB, LOADS)                                              \
    mma1(FA, FB, 0);                                                            \
    NT256_SB();                                                                 \
    LOADS;                                                                      \
    NT256_SB();                                                                 \
    _Pragma("unroll") for (int m = 1; m < 8; ++m) mma1(FA, FB, m);              \
    NT256_SB();
    if (nch <= 0) return;
    if (!chunk0_in_flight) {
#pragma unroll
        for (int p = 0; p < 8; ++p) dma(0, (int64_t)0, p);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int64_t f = nch > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) dma(1, f, p);
    }
    ld(fa0, fb0, 0, 0);
    for (int64_t ch = 0; ch < nch; ++ch) {
        const int st = (int)(ch & 1);
        NT256_KSTEP(fa0, fb0, ld(fa1, fb1, st, 1))
        NT256_KSTEP(fa1, fb1, ld(fa0, fb0, st, 2))
        NT256_KSTEP(fa0, fb0, ld(fa1, fb1, st, 3))
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ld(fa0, fb0, st ^ 1, 0);
        NT256_SB();
        const int64_t f = (ch + 2 < nch) ? ch + 2 : nch - 1;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            mma1(fa1, fb1, m);
            NT256_SB();
            dma(st, f, m);
            NT256_SB();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-fetches of the tail
    __syncthreads();                                    // staging memory is free for the epilogue
#undef NT256_KSTEP
#undef NT256_SB
}
// The same loop on SmemQ3: iteration ch issues B of chunk ch + 1 (into the B stage chunk ch - 1 freed; B is a weight image: L2 / MALL
// resident) and A of chunk ch + 2 (into the A stage chunk ch - 1 freed) behind MFMAs 1, 3 and 5 of its first three k-steps -- B first --
// and waits with vmcnt(4): everything but this iteration's A pieces has landed.  No chunk0_in_flight form (one tile per workgroup).
template <class Dma>
__device__ __forceinline__ void nt256_mainloop3(SmemQ3& sm, f32x16 (&acc)[4][2], int64_t nch, int wm, int wn, int lane, Dma&& dma) {
    const int l32 = lane & 31, kh = lane >> 5;
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
    auto ld = [&](bf16x8 (&fa)[4], bf16x8 (&fb)[2], int sa, int sb, int ks) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) fa[rt] = *reinterpret_cast<const bf16x8*>(&sm.A[sa][offA[rt] ^ (ks << 5)]);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) fb[ct] = *reinterpret_cast<const bf16x8*>(&sm.B[sb][offB[ct] ^ (ks << 5)]);
    };
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
#define NT256_SB() __builtin_amdgcn_sched_barrier(0)
#define NT256_KSTEPD(FA, FB, LOADS, D1, D2, D3)                                 \
    mma1(FA, FB, 0);                                                            \
    NT256_SB();                                                                 \
    LOADS;                                                                      \
    NT256_SB();                                                                 \
    mma1(FA, FB, 1);                                                            \
    NT256_SB();                                                                 \
    D1;                                                                         \
    NT256_SB();                                                                 \
    mma1(FA, FB, 2);                                                            \
    mma1(FA, FB, 3);                                                            \
    NT256_SB();                                                                 \
    D2;                                                                         \
    NT256_SB();                                                                 \
    mma1(FA, FB, 4);                                                            \
    mma1(FA, FB, 5);                                                            \
    NT256_SB();                                                                 \
    D3;                                                                         \
    NT256_SB();                                                                 \
    mma1(FA, FB, 6);                                                            \
    mma1(FA, FB, 7);                                                            \
    NT256_SB();
    if (nch <= 0) return;
#pragma unroll
    for (int p = 0; p < 8; ++p) dma(0, (int64_t)0, p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int64_t f = nch > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 4; ++p) dma(1, f, p);   // A of chunk 1 (B of chunk 1 follows inside iteration 0)
    }
    ld(fa0, fb0, 0, 0, 0);
    int sa = 0;
    for (int64_t ch = 0; ch < nch; ++ch) {
        const int st = (int)(ch & 1);
        const int san = (sa == 2) ? 0 : sa + 1, saf = (san == 2) ? 0 : san + 1;
        const int64_t fb = (ch + 1 < nch) ? ch + 1 : nch - 1, fa = (ch + 2 < nch) ? ch + 2 : nch - 1;
#define NT256_DB(i) dma(st ^ 1, fb, 4 + (i))
#define NT256_DA(i) dma(saf, fa, (i))
        NT256_KSTEPD(fa0, fb0, ld(fa1, fb1, sa, st, 1), NT256_DB(0), NT256_DB(1), NT256_DB(2))
        NT256_KSTEPD(fa1, fb1, ld(fa0, fb0, sa, st, 2), NT256_DB(3), NT256_DA(0), NT256_DA(1))
        NT256_KSTEPD(fa0, fb0, ld(fa1, fb1, sa, st, 3), NT256_DA(2), NT256_DA(3), (void)0)
#undef NT256_DB
#undef NT256_DA
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __syncthreads();
        ld(fa0, fb0, san, st ^ 1, 0);
        NT256_SB();
#pragma unroll
        for (int m = 0; m < 8; ++m) mma1(fa1, fb1, m);
        NT256_SB();
        sa = san;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-fetches of the tail
    __syncthreads();                                    // staging memory is free for the epilogue
#undef NT256_KSTEPD
#undef NT256_SB
}
// epilogue of the 256-tile: the wave's 128 x 64 sub-tile through its private 32 x 64 LDS transpose tile, as two 64-row halves of
// epilogue_rows8 (emit sees tile-relative rows / columns)
template <class Emit>
// stage1 = true: the transpose tiles live in stage 1's memory only (waves 0-3: A[1], 4-7: B[1]) -- for persistent workgroups whose stage 0
// already receives the next tile's first chunk during this epilogue
__device__ __forceinline__ void nt256_epilogue(const f32x16 (&acc)[4][2], SmemQ& sm, int wave, int wm, int wn, int lane, int rows_valid,
                                               Emit&& emit, bool stage1 = false) {
    float* tile = stage1 ? reinterpret_cast<float*>(wave < 4 ? &sm.A[1][wave * 8192] : &sm.B[1][(wave - 4) * 8192])
                         : reinterpret_cast<float*>(&sm) + wave * (32 * 64);
    const int colb[2] = {wn * 64, wn * 64 + 32};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const f32x16 (&half)[2][2] = reinterpret_cast<const f32x16 (&)[2][2]>(acc[2 * h]);
        if (rows_valid >= QM) epilogue_rows8<true>(half, tile, wm * 2 + h, colb, lane, QM, emit);
        else epilogue_rows8<false>(half, tile, wm * 2 + h, colb, lane, rows_valid, emit);
    }
}

// ================================================================================================
// TN.  Measured semantics of ds_read_b64_tr_b16 (tools/micro/tr_probe.hip): within a 16-lane group every lane r supplies the
// address of 4 consecutive bf16 D[r][0..3] and lane l receives D[4j + (l >> 2)][l & 3], j = 0..3; with lane r pointing at
// tile[k0 + (r >> 2)][i0 + 4 (r & 3)] lane l gets tile[k0 + j][i0 + l] -- 4 consecutive k of "its" row/column.  LDS stage:
// A image [32 k][128 m] (8 KiB) + B image [32 k][256 n] (16 KiB), 64-B unit u of k-row t stored at unit u ^ (t & 3)
// (source-side swizzle of the LDS-DMA), two stages, fragments of k-step s+1 requested behind the first MFMA of step s
// (inline-asm reads: the s_waitcnt lgkmcnt is placed by hand).
// ================================================================================================
constexpr int TNK = 32;
struct __attribute__((aligned(16))) SmemTN {
    bf16_t A[2][TNK * 128];
    bf16_t B[2][TNK * 256];
};
template <int OFF>
__device__ __forceinline__ u32x2 ds_tr16(uint32_t lds_byte_addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_addr), "i"(OFF));
    return v;
}
struct TnFrag {
    u32x2 a[2][2], b[4][2];   // [row / column tile][k half]
};
template <int KS>   // k-step 0 / 1 of the staged chunk
__device__ __forceinline__ void tn_load(TnFrag& f, const uint32_t (&aA)[2], const uint32_t (&aB)[4]) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        f.a[rt][0] = ds_tr16<KS * 4096>(aA[rt]);
        f.a[rt][1] = ds_tr16<KS * 4096 + 1024>(aA[rt]);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        f.b[ct][0] = ds_tr16<KS * 8192>(aB[ct]);
        f.b[ct][1] = ds_tr16<KS * 8192 + 2048>(aB[ct]);
    }
}
__device__ __forceinline__ void tn_mma(f32x16 (&acc)[2][4], const TnFrag& f, int m) {
    const int rt = m & 1, ct = m >> 1;
    const u32x4 av = {f.a[rt][0].x, f.a[rt][0].y, f.a[rt][1].x, f.a[rt][1].y};
    const u32x4 bv = {f.b[ct][0].x, f.b[ct][0].y, f.b[ct][1].x, f.b[ct][1].y};
    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[rt][ct], 0,
                                                          0, 0);
}

// Generic TN tile loop: acc[rt][ct] += sum over nch chunks of 32 k of A[k][wm*64 + rt*32 ..] B[k][wn*128 + ct*32 ..].
// dma(stage, chunk, piece): piece 0..5 = this wave's LDS-DMA instructions (0,1: A k-rows 4(2w+p)..+3; 2..5: B k-rows 2(4w+p-2), +1).
template <class Dma>
__device__ __forceinline__ void tn_mainloop(SmemTN& sm, f32x16 (&acc)[2][4], int64_t nch, int wm, int wn, int lane, Dma&& dma) {
    const int g = lane >> 4, r = lane & 15;
    // lane r of group g points at token row (g >> 1) * 8 + (r >> 2) (+ 16 KS + 4 h through the immediate offset), columns
    // tile0 + (g & 1) * 16 + (r & 3) * 4; the 64-B unit XOR depends on (token & 3) = (r >> 2) only
    uint32_t a0[2], b0[4];
    const int kb = (g >> 1) * 8 + (r >> 2);
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
        a0[rt] = lds_addr_of(&sm.A[0][0]) + kb * 256 + (((wm * 64 + rt * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
        b0[ct] = lds_addr_of(&sm.B[0][0]) + kb * 512 + (((wn * 128 + ct * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
    if (nch <= 0) return;
#pragma unroll
    for (int p = 0; p < 6; ++p) dma(0, (int64_t)0, p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int64_t f = nch > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 6; ++p) dma(1, f, p);
    }
    TnFrag f0, f1;
    tn_load<0>(f0, a0, b0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    for (int64_t ch = 0; ch < nch; ++ch) {
        const int st = (int)(ch & 1);
        uint32_t aA[2], aB[4], nA[2], nB[4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            aA[rt] = a0[rt] + st * (TNK * 128 * 2);
            nA[rt] = a0[rt] + (st ^ 1) * (TNK * 128 * 2);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            aB[ct] = b0[ct] + st * (TNK * 256 * 2);
            nB[ct] = b0[ct] + (st ^ 1) * (TNK * 256 * 2);
        }
        // k-step 0: its first MFMA, then the requests of k-step 1, then the other 7 MFMAs
        tn_mma(acc, f0, 0);
        __builtin_amdgcn_sched_barrier(0);
        tn_load<1>(f1, aA, aB);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 1; m < 8; ++m) tn_mma(acc, f0, m);
        __builtin_amdgcn_sched_barrier(0);
        // k-step 1 (last of the chunk): all reads of this stage are requested -> wait for them and for this wave's DMA of the next
        // chunk, barrier, then the next chunk's first fragments and the DMA of chunk ch+2 ride between the MFMAs
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        tn_load<0>(f0, nA, nB);
        __builtin_amdgcn_sched_barrier(0);
        const int64_t f = (ch + 2 < nch) ? ch + 2 : nch - 1;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            tn_mma(acc, f1, m);
            __builtin_amdgcn_sched_barrier(0);
            if (m < 6) dma(st, f, m);
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);   // no MFMA of the next iteration above the wait (inline-asm reads are not tracked)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}


// ---- TN on the 256 x 256 tile (round 5) -------------------------------------------------------------------------------------------------
// The 128 x 256 x 32 loop above stages 24.5 KB per 2.1 MFLOP (85 flop per byte through L2 -> LDS) and takes a barrier every 16 MFMAs per
// wave: the Linear / gate dW products ran at 320-455 TF where the 256-tile NT products reach 780-890.  Here: 8 waves (2 x 4), each 128 x 64
// = 4 x 2 MFMA tiles, chunks of TQK = 64 tokens, stage = [64 tokens][256 columns] bf16 per operand (512-B rows, 64-B unit u of token row
// t stored at unit u ^ (t & 3)), two stages = SmemQ's 128 KiB -- the image and the fragment reads of the split engine's TN loop
// (split_engine.hpp sp_tn_mainloop) with one bf16 plane of 64 tokens instead of two fp16 planes of 32: 128 flop per staged byte, a barrier
// every 32 MFMAs.  k-step KS (16 tokens) = + KS * 8192 B; the second half of a k-step's fragment = + 4 rows = 2048 B.
// DMA piece q (< 4) of wave w for one operand: token rows (4 w + q) * 2 + (lane >> 5); the stored 16-B chunk position lane & 31 holds
// the global chunk (lane & 31) ^ ((row & 3) << 2) (8 columns each)  ->  LDS bytes (4 w + q) * 1024 + lane * 16.
constexpr int TQK = 64;
struct Tn256Frag {
    u32x2 a[4][2], b[2][2];
};
template <int KS>
__device__ __forceinline__ void tn256_load(Tn256Frag& f, const uint32_t (&aA)[4], const uint32_t (&aB)[2]) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        f.a[rt][0] = ds_tr16<KS * 8192>(aA[rt]);
        f.a[rt][1] = ds_tr16<KS * 8192 + 2048>(aA[rt]);
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        f.b[ct][0] = ds_tr16<KS * 8192>(aB[ct]);
        f.b[ct][1] = ds_tr16<KS * 8192 + 2048>(aB[ct]);
    }
}
__device__ __forceinline__ void tn256_mma(f32x16 (&acc)[4][2], const Tn256Frag& f, int m) {
    const int rt = m >> 1, ct = m & 1;
    const u32x4 av = {f.a[rt][0].x, f.a[rt][0].y, f.a[rt][1].x, f.a[rt][1].y};
    const u32x4 bv = {f.b[ct][0].x, f.b[ct][0].y, f.b[ct][1].x, f.b[ct][1].y};
    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[rt][ct], 0,
                                                          0, 0);
}
// this lane's token row and source chunk (x 16 B) of DMA piece q
__device__ __forceinline__ void tn256_slot(int wave, int q, int lane, int& row, int& src_chunk) {
    row = (wave * 4 + q) * 2 + (lane >> 5);
    src_chunk = (lane & 31) ^ ((row & 3) << 2);
}
// acc[rt][ct] += sum over nch chunks of 64 tokens of A[t][wm*128 + rt*32 ..] B[t][wn*64 + ct*32 ..];  dma(stage, chunk, piece < 8):
// pieces 0-3 = this wave's A token-row pairs, 4-7 = its B pairs (glds16_s).  acc is zeroed here.
template <class Dma>
__device__ __forceinline__ void tn256_mainloop(SmemQ& sm, f32x16 (&acc)[4][2], int64_t nch, int wm, int wn, int lane, Dma&& dma) {
    const int g = lane >> 4, r = lane & 15;
    const int kb = (g >> 1) * 8 + (r >> 2);
    uint32_t a_0[4], b_0[2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
        a_0[rt] = lds_addr_of(&sm.A[0][0]) + kb * 512 + (((wm * 128 + rt * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        b_0[ct] = lds_addr_of(&sm.B[0][0]) + kb * 512 + (((wn * 64 + ct * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (nch <= 0) return;
#define TQ_SB() __builtin_amdgcn_sched_barrier(0)
#define TQ_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#pragma unroll
    for (int p = 0; p < 8; ++p) dma(0, (int64_t)0, p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int64_t f = nch > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) dma(1, f, p);
    }
    Tn256Frag f0, f1;
    tn256_load<0>(f0, a_0, b_0);
    TQ_LGKM();
    TQ_SB();
    // one k-step: its first MFMA, the fragment requests of the next k-step behind it, the other 7 MFMAs, then the wait for those requests
    // (inline-asm LDS reads are not tracked by the compiler's s_waitcnt insertion)
#define TQ_SET(F, LOADS)                                                \
    tn256_mma(acc, F, 0);                                               \
    TQ_SB();                                                            \
    LOADS;                                                              \
    TQ_SB();                                                            \
    _Pragma("unroll") for (int m = 1; m < 8; ++m) tn256_mma(acc, F, m); \
    TQ_SB();                                                            \
    TQ_LGKM();                                                          \
    TQ_SB();
    for (int64_t ch = 0; ch < nch; ++ch) {
        const int st = (int)(ch & 1);
        uint32_t aA[4], aB[2], nA[4], nB[2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            aA[rt] = a_0[rt] + st * Q_STAGE;
            nA[rt] = a_0[rt] + (st ^ 1) * Q_STAGE;
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            aB[ct] = b_0[ct] + st * Q_STAGE;
            nB[ct] = b_0[ct] + (st ^ 1) * Q_STAGE;
        }
        TQ_SET(f0, tn256_load<1>(f1, aA, aB))
        TQ_SET(f1, tn256_load<2>(f0, aA, aB))
        TQ_SET(f0, tn256_load<3>(f1, aA, aB))
        // last k-step of the chunk: every read of stage st has been requested and waited for -> this wave's DMA of the next chunk must
        // have landed, barrier, then the next chunk's first fragments and the DMA of chunk ch + 2 between this k-step's MFMAs
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TQ_SB();
        __syncthreads();
        tn256_load<0>(f0, nA, nB);
        TQ_SB();
        const int64_t f = (ch + 2 < nch) ? ch + 2 : nch - 1;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            tn256_mma(acc, f1, m);
            TQ_SB();
            dma(st, f, m);
            TQ_SB();
        }
        TQ_LGKM();
        TQ_SB();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-fetches of the tail
    __syncthreads();
#undef TQ_SET
#undef TQ_LGKM
#undef TQ_SB
}

// The TN loop on SmemQ3: both operands stream from HBM -- A of chunk ch + 2 is spread over the first three k-steps of
// iteration ch (into the A stage chunk ch - 1 freed), B of chunk ch + 2 keeps the slot behind the chunk barrier (four pieces instead of eight
// there); wait vmcnt(4).
template <class Dma>
__device__ __forceinline__ void tn256_mainloop3(SmemQ3& sm, f32x16 (&acc)[4][2], int64_t nch, int wm, int wn, int lane, Dma&& dma) {
    const int g = lane >> 4, r = lane & 15;
    const int kb = (g >> 1) * 8 + (r >> 2);
    uint32_t a_0[4], b_0[2];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
        a_0[rt] = lds_addr_of(&sm.A[0][0]) + kb * 512 + (((wm * 128 + rt * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
        b_0[ct] = lds_addr_of(&sm.B[0][0]) + kb * 512 + (((wn * 64 + ct * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (nch <= 0) return;
#define TQ_SB() __builtin_amdgcn_sched_barrier(0)
#define TQ_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#pragma unroll
    for (int p = 0; p < 8; ++p) dma(0, (int64_t)0, p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const int64_t f = nch > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) dma(1, f, p);   // A and B of chunk 1 (A of chunk 2 follows inside iteration 0)
    }
    Tn256Frag f0, f1;
    tn256_load<0>(f0, a_0, b_0);
    TQ_LGKM();
    TQ_SB();
#define TQ_SETD(F, LOADS, D1, D2)                                       \
    tn256_mma(acc, F, 0);                                               \
    TQ_SB();                                                            \
    LOADS;                                                              \
    TQ_SB();                                                            \
    tn256_mma(acc, F, 1);                                               \
    tn256_mma(acc, F, 2);                                               \
    TQ_SB();                                                            \
    D1;                                                                 \
    TQ_SB();                                                            \
    tn256_mma(acc, F, 3);                                               \
    tn256_mma(acc, F, 4);                                               \
    tn256_mma(acc, F, 5);                                               \
    TQ_SB();                                                            \
    D2;                                                                 \
    TQ_SB();                                                            \
    tn256_mma(acc, F, 6);                                               \
    tn256_mma(acc, F, 7);                                               \
    TQ_SB();                                                            \
    TQ_LGKM();                                                          \
    TQ_SB();
    int sa = 0;
    for (int64_t ch = 0; ch < nch; ++ch) {
        const int st = (int)(ch & 1);
        const int san = (sa == 2) ? 0 : sa + 1, saf = (san == 2) ? 0 : san + 1;
        uint32_t aA[4], aB[2], nA[4], nB[2];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            aA[rt] = a_0[rt] + sa * Q_STAGE;
            nA[rt] = a_0[rt] + san * Q_STAGE;
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            aB[ct] = b_0[ct] + st * Q_STAGE;
            nB[ct] = b_0[ct] + (st ^ 1) * Q_STAGE;
        }
        const int64_t fa = (ch + 2 < nch) ? ch + 2 : nch - 1;
#define TQ_DA(i) dma(saf, fa, (i))
        TQ_SETD(f0, tn256_load<1>(f1, aA, aB), TQ_DA(0), TQ_DA(1))
        TQ_SETD(f1, tn256_load<2>(f0, aA, aB), TQ_DA(2), (void)0)
        TQ_SETD(f0, tn256_load<3>(f1, aA, aB), TQ_DA(3), (void)0)
#undef TQ_DA
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        TQ_SB();
        __syncthreads();
        tn256_load<0>(f0, nA, nB);
        TQ_SB();
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            tn256_mma(acc, f1, m);
            TQ_SB();
            if (m >= 4) dma(st, fa, m);   // B of chunk ch + 2 into the stage this chunk frees
            TQ_SB();
        }
        TQ_LGKM();
        TQ_SB();
        sa = san;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the clamped re-fetches of the tail
    __syncthreads();
#undef TQ_SETD
#undef TQ_LGKM
#undef TQ_SB
}

}  // namespace mdl
