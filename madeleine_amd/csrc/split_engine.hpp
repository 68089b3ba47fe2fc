// split_engine.hpp -- fp32-ACCURATE contractions on the fp16 matrix cores (round 3).
//
// Every fp32 contraction of the step is exact-fp32 work (parity at InfoNCE temperature 0.001 needs it), and
// v_mfma_f32_32x32x2_f32 tops out at 157 TFLOP/s.  Here each fp32 operand value x (times a per-tensor power-of-two scale s) is
// carried as TWO fp16 planes
//         hi = RN16(s x),   lo = RN16(s x - hi)            ->  s x = hi + lo  up to 2^-23 |s x|
// and a product as THREE v_mfma_f32_32x32x16_f16 terms accumulated in fp32:
//         a b ~= ah bh + ah bl + al bh                      (al bl <= 2^-22 |a b| is dropped)
// Each 16-bit product is exact in fp32; the accumulation is the matrix core's fp32 accumulation.  Measured (tools/micro/split_lab.hip,
// K = 512 .. 2048, against fp64): max |err| / sum_k |a_k b_k| = 1.6e-7, rms 2.0e-8 -- BELOW the plain fp32 fmaf chain's 3.4e-7 / 3.0e-8
// (fewer roundings: 16 k per accumulate).  3 x 32 MFMA cycles per 16 k against 8 x 64 for the fp32 instruction: the same contraction
// in 19 % of the matrix-core cycles; the lab kernel sustains 290-330 TFLOP/s fp32-equivalent against the fp32 engine's 141.
// fp16 has 5 exponent bits: s is chosen per tensor such that max |s x| < 2^15 (from an exact absmax or a rigorous bound); relative
// representation error <= max(2^-23, 2^-25 / |s x|): values within 2^-16 of the maximum keep 21+ bits, smaller ones degrade
// gracefully (absolute error <= 2^-38 of the tensor maximum -- far below the fp32 rounding of any sum they enter).
//
// SPLIT IMAGE of a matrix X [rows][K] (K % 32 == 0): uint16 [rows][K / 32][2 planes][32]  -- 128 B per row and 32-k block (one
// cache line: hi plane | lo plane), i.e. exactly the bytes of the fp32 row.  One image serves both operand roles:
//   NT  C[m][n] = sum_k A[m][k] B[n][k]   rows of both images are K-contiguous          (forward products, dX)
//   TN  C[m][n] = sum_t A[t][m] B[t][n]   rows of both images are the contraction index  (dW; fragments by ds_read_b64_tr_b16)
// Tile: 256 x 256 per workgroup (8 waves as 2 x 4, 128 x 64 = 4 x 2 MFMA tiles per wave, 128 accumulators; see SPNCT), one chunk =
// one 32-k block of both operands = 2 x 32 KiB by LDS-DMA, 48 MFMAs per wave and chunk, the in-wave software pipeline of the other engines
// (fragments of the next MFMA set requested behind the first MFMA of the current one, one barrier per chunk before its last set).  Two
// loop families: the two-stage ring (128 KiB; the next-but-one chunk's eight DMA pieces between the MFMAs of the last set) and, since
// round 6 the default of every one-tile-per-workgroup kernel, the three-stage A ring (160 KiB) with the DMA pieces spread over the chunk
// (sp_nt_mainloop3 / sp_tn_mainloop3).
#pragma once
#include "gate_common.hpp"

namespace mdl {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int SPM = 256, SPN = 256, SPK = 32;
constexpr int SP_STAGE = 256 * 128;   // bytes per operand per stage
// Wave layout of the 256 x 256 tile: 2 x SP_WN waves, each 128 rows x (32 SPNCT) columns = 4 x SPNCT MFMA tiles.
//   SPNCT = 2: 8 waves x 128 x 64  (128 accumulator registers, two waves per SIMD)  -- shipped
//   SPNCT = 4: 4 waves x 128 x 128 (256 accumulators in AGPRs, ONE wave per SIMD): 8 fragment reads per 16 MFMAs instead of 6 per 8, a
//              third less LDS read traffic per MFMA.  +12-17 % in tools/micro/split_lab.hip (L2-resident A operand, plain epilogue), but
//              SLOWER in the product (Linears -7 %, gate dX / dW -8 %; the fused gate epilogues spill with 256 live accumulators):
//              with HBM-streamed operands and the LDS-transposed epilogues a single wave per SIMD has nothing to hide its stalls behind.
//              The kernels are written against these constants, so the A/B is this one line.
constexpr int SPNCT = 2;
constexpr int SP_WN = 8 / SPNCT;            // wave columns
constexpr int SP_WAVES = 2 * SP_WN;
constexpr int SP_THREADS = 64 * SP_WAVES;
constexpr int SP_PW = 32 / SP_WAVES;        // LDS-DMA pieces (1 KiB) per wave, operand and stage
constexpr int SP_NP = 2 * SP_PW;            // pieces per wave and chunk = MFMAs of one set = 4 SPNCT
constexpr int SP_WCOLS = 32 * SPNCT;        // columns per wave
typedef f32x16 SpAcc[4][SPNCT];
// NA = stages of the A ring.  2: the symmetric two-stage ring (128 KiB).  3 (round 6, NT products, sp_nt_mainloop3): A three deep, B two
// deep = 160 KiB, the whole LDS.
template <int NA>
struct __attribute__((aligned(16))) SmemSPn {
    char A[NA][SP_STAGE];
    char B[2][SP_STAGE];
};
typedef SmemSPn<2> SmemSP;
typedef SmemSPn<3> SmemSP3;
// which pieces the three-stage loops spread (A/B through tools/ab: -DMDL_SP_NT_BSPREAD=0 / -DMDL_SP_TN_BSPREAD=1)
#ifndef MDL_SP_NT_BSPREAD
#define MDL_SP_NT_BSPREAD 1   // NT: the B operand is a weight (L2 / MALL resident): its pieces inside the chunk too
#endif
#ifndef MDL_SP_TN_BSPREAD
#define MDL_SP_TN_BSPREAD 0   // TN: both operands stream from HBM: B keeps the full chunk of flight time
#endif
// which NT main loop the launchers pick: MADELEINE_SP_NT_STAGES = 2 | 3 (A/B switch), default MDL_SP_NT_STAGES
#ifndef MDL_SP_NT_STAGES
#define MDL_SP_NT_STAGES 3
#endif
// the TN (dW) loops: MADELEINE_SP_TN_STAGES = 2 | 3 (sp_tn_mainloop3), default MDL_SP_TN_STAGES
#ifndef MDL_SP_TN_STAGES
#define MDL_SP_TN_STAGES 3
#endif
static inline int sp_tn_stages() {
    static const int v = getenv("MADELEINE_SP_TN_STAGES") ? atoi(getenv("MADELEINE_SP_TN_STAGES")) : MDL_SP_TN_STAGES;
    return v == 3 ? 3 : 2;
}
static inline int sp_nt_stages() {
    static const int v = getenv("MADELEINE_SP_NT_STAGES") ? atoi(getenv("MADELEINE_SP_NT_STAGES")) : MDL_SP_NT_STAGES;
    return v == 2 ? 2 : 3;
}

// a wave-uniform pointer the compiler can keep in SGPRs (saddr operand of the LDS-DMA)
__device__ __forceinline__ const char* sp_uniform(const char* p) {
    const uint64_t v = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return reinterpret_cast<const char*>(((uint64_t)hi << 32) | lo);
}

#define SP_SB() __builtin_amdgcn_sched_barrier(0)
#define SP_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

__device__ __forceinline__ void sp_zero(SpAcc& acc) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < SPNCT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}
__device__ __forceinline__ f32x16 sp_mfma(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// ---- NT ---------------------------------------------------------------------------------------------------------------------------
// LDS stage row = 128 B = 8 chunks of 16 B; chunk index = ks * 2 + kh with ks = plane * 2 + s (s = 16-k half of the block); 16-B
// chunk c of tile row r is stored at chunk position c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 on 128-B rows).
// DMA piece i (< SP_PW) of wave w for one operand: rows (SP_PW w + i) * 8 .. + 7 -> LDS bytes (SP_PW w + i) * 1024 ..; this lane
// deposits global chunk c of row `row`.
__device__ __forceinline__ void sp_nt_slot(int wave, int i, int lane, int& row, int& c) {   // i < SP_PW
    row = (wave * SP_PW + i) * 8 + (lane >> 3);
    c = (lane & 7) ^ ((row >> 1) & 7);
}
// dma(stage, block, piece): piece < SP_PW = this wave's A row groups, SP_PW .. SP_NP - 1 = its B row groups (glds16_s).
// TERMS = 3: ah bh + ah bl + al bh.  TERMS = 2 drops ah bl, i.e. the B operand enters rounded to its hi plane (11 bits): four MFMA sets
// per chunk instead of six.  For gradient products whose B operand is a weight (dX = dY W); never the default.
// chunk0_in_flight: the caller already issued this wave's pieces of block 0 into stage 0 (a persistent workgroup requests the next
// tile's first block before the epilogue of the current one and keeps its epilogue staging inside stage 1: sp_stage1_tile).
template <int TERMS = 3, class Dma>
__device__ __forceinline__ void sp_nt_mainloop(SmemSP& sm, SpAcc& acc, int nblk, int wm, int wn, int lane, Dma&& dma,
                                               bool chunk0_in_flight = false) {
    const int l32 = lane & 31, kh = lane >> 5;
    uint32_t offA[4], offB[SPNCT];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int r = wm * 128 + rt * 32 + l32;
        offA[rt] = r * 128 + ((kh ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < SPNCT; ++ct) {
        const int r = wn * SP_WCOLS + ct * 32 + l32;
        offB[ct] = r * 128 + ((kh ^ ((r >> 1) & 7)) << 4);
    }
    auto ldA = [&](u32x4 (&fa)[4], int st, int ks) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) fa[rt] = *reinterpret_cast<const u32x4*>(&sm.A[st][offA[rt] ^ (ks << 5)]);
    };
    auto ldB = [&](u32x4 (&fb)[SPNCT], int st, int ks) {
#pragma unroll
        for (int ct = 0; ct < SPNCT; ++ct) fb[ct] = *reinterpret_cast<const u32x4*>(&sm.B[st][offB[ct] ^ (ks << 5)]);
    };
    auto mma1 = [&](const u32x4 (&fa)[4], const u32x4 (&fb)[SPNCT], int m) {
        const int rt = m / SPNCT, ct = m % SPNCT;
        acc[rt][ct] = sp_mfma(fa[rt], fb[ct], acc[rt][ct]);
    };
#define SP_SET(FA, FB, LOADS)                                                   \
    mma1(FA, FB, 0);                                                            \
    SP_SB();                                                                    \
    LOADS;                                                                      \
    SP_SB();                                                                    \
    _Pragma("unroll") for (int m = 1; m < SP_NP; ++m) mma1(FA, FB, m);          \
    SP_SB();
    if (nblk <= 0) return;
    u32x4 a0[4], a1[4], a2[4], b0[SPNCT], b1[SPNCT], b2[SPNCT];
    if (!chunk0_in_flight) {
#pragma unroll
        for (int p = 0; p < SP_NP; ++p) dma(0, 0, p);
    }
    SP_DMA_WAIT();
    __syncthreads();
    {
        const int f = nblk > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < SP_NP; ++p) dma(1, f, p);
    }
    ldA(a0, 0, 0);
    ldB(b0, 0, 0);
    for (int ch = 0; ch < nblk; ++ch) {
        const int st = ch & 1;
        // ks: 0 = hi k 0-15, 1 = hi k 16-31, 2 = lo k 0-15, 3 = lo k 16-31;  a0 = A hi s0, b0 = B hi s0 on entry
        if constexpr (TERMS == 3) {
            SP_SET(a0, b0, ldB(b1, st, 2))                   // hi hi, s0   | B lo s0
            SP_SET(a0, b1, ldA(a1, st, 2))                   // hi lo, s0   | A lo s0
            SP_SET(a1, b0, ldA(a2, st, 1); ldB(b2, st, 1))   // lo hi, s0   | A hi s1, B hi s1
            SP_SET(a2, b2, ldB(b1, st, 3))                   // hi hi, s1   | B lo s1
            SP_SET(a2, b1, ldA(a1, st, 3))                   // hi lo, s1   | A lo s1
        } else {
            SP_SET(a0, b0, ldA(a1, st, 2))                   // hi hi, s0   | A lo s0
            SP_SET(a1, b0, ldA(a2, st, 1); ldB(b2, st, 1))   // lo hi, s0   | A hi s1, B hi s1
            SP_SET(a2, b2, ldA(a1, st, 3))                   // hi hi, s1   | A lo s1
        }
        // last set of the chunk: every read of stage st has been requested -> barrier, then the next chunk's first fragments and
        // the DMA of block ch + 2 between this set's MFMAs (the last two iterations re-fetch the last block: branch-free body)
        SP_DMA_WAIT();
        __syncthreads();
        ldA(a0, st ^ 1, 0);
        ldB(b0, st ^ 1, 0);
        SP_SB();
        const int f = (ch + 2 < nblk) ? ch + 2 : nblk - 1;
#pragma unroll
        for (int m = 0; m < SP_NP; ++m) {
            mma1(a1, b2, m);                             // lo hi, s1
            SP_SB();
            dma(st, f, m);
            SP_SB();
        }
    }
    SP_DMA_WAIT();
    __syncthreads();   // staging memory is free for the epilogue
#undef SP_SET
}

// The same loop on SmemSP3 (round 6, an experiment kept behind MADELEINE_SP_NT_STAGES=3): A ring three stages deep, and the LDS-DMA pieces
// SPREAD over the chunk instead of issued together behind the chunk barrier.  In the two-stage loop all eight pieces of block ch + 2 are
// issued inside the last MFMA set of chunk ch -- the only place where a stage is free -- and an LDS-DMA piece costs 100-185 issue cycles in
// a phase that already carries eight of them (MI355X_MICROARCH.md price list; the NODMA probe of round 4: +19 %).  With a third A stage a
// free stage exists during the whole chunk: iteration ch issues B of block ch + 1 (into the B stage freed by chunk ch - 1) behind MFMAs
// 2 and 4 of its first two sets and A of block ch + 2 (into the A stage freed by chunk ch - 1) in sets 3 .. 5.  Wait at the chunk end:
// vmcnt(SP_PW) -- everything but this iteration's A pieces has landed (B of ch + 1 was issued before them, A of ch + 1 an iteration ago).
// BSPREAD = false: only the A pieces are spread (one per set, sets 1 .. 4); B of block ch + 2 keeps the slot behind the chunk barrier (the
// placement of sp_tn_mainloop3).
template <int TERMS = 3, bool BSPREAD = (MDL_SP_NT_BSPREAD != 0), class Dma>
__device__ __forceinline__ void sp_nt_mainloop3(SmemSP3& sm, SpAcc& acc, int nblk, int wm, int wn, int lane, Dma&& dma) {
    static_assert(SP_PW == 4, "piece placement below is written for four pieces per wave and operand");
    const int l32 = lane & 31, kh = lane >> 5;
    uint32_t offA[4], offB[SPNCT];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        const int r = wm * 128 + rt * 32 + l32;
        offA[rt] = r * 128 + ((kh ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < SPNCT; ++ct) {
        const int r = wn * SP_WCOLS + ct * 32 + l32;
        offB[ct] = r * 128 + ((kh ^ ((r >> 1) & 7)) << 4);
    }
    auto ldA = [&](u32x4 (&fa)[4], int st, int ks) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) fa[rt] = *reinterpret_cast<const u32x4*>(&sm.A[st][offA[rt] ^ (ks << 5)]);
    };
    auto ldB = [&](u32x4 (&fb)[SPNCT], int st, int ks) {
#pragma unroll
        for (int ct = 0; ct < SPNCT; ++ct) fb[ct] = *reinterpret_cast<const u32x4*>(&sm.B[st][offB[ct] ^ (ks << 5)]);
    };
    auto mma1 = [&](const u32x4 (&fa)[4], const u32x4 (&fb)[SPNCT], int m) {
        const int rt = m / SPNCT, ct = m % SPNCT;
        acc[rt][ct] = sp_mfma(fa[rt], fb[ct], acc[rt][ct]);
    };
    // one MFMA set with the fragment loads of the next set behind its first MFMA and DMA pieces behind its third and fifth
#define SP_SETD(FA, FB, LOADS, D1, D2)                                          \
    mma1(FA, FB, 0);                                                            \
    SP_SB();                                                                    \
    LOADS;                                                                      \
    SP_SB();                                                                    \
    mma1(FA, FB, 1);                                                            \
    mma1(FA, FB, 2);                                                            \
    SP_SB();                                                                    \
    D1;                                                                         \
    SP_SB();                                                                    \
    mma1(FA, FB, 3);                                                            \
    mma1(FA, FB, 4);                                                            \
    SP_SB();                                                                    \
    D2;                                                                         \
    SP_SB();                                                                    \
    _Pragma("unroll") for (int m = 5; m < SP_NP; ++m) mma1(FA, FB, m);          \
    SP_SB();
    if (nblk <= 0) return;
    u32x4 a0[4], a1[4], a2[4], b0[SPNCT], b1[SPNCT], b2[SPNCT];
#pragma unroll
    for (int p = 0; p < SP_NP; ++p) dma(0, 0, p);
    SP_DMA_WAIT();
    __syncthreads();
    {
        const int f = nblk > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < (BSPREAD ? SP_PW : SP_NP); ++p) dma(1, f, p);   // A of block 1 (BSPREAD: B of block 1 follows inside iteration 0)
    }
    ldA(a0, 0, 0);
    ldB(b0, 0, 0);
    int sa = 0;   // A stage of chunk ch = ch % 3
    for (int ch = 0; ch < nblk; ++ch) {
        const int st = ch & 1;                               // B stage of chunk ch
        const int san = (sa == 2) ? 0 : sa + 1;             // A stage of chunk ch + 1 (in flight since the previous iteration)
        const int saf = (san == 2) ? 0 : san + 1;           // A stage freed by chunk ch - 1: receives block ch + 2
        const int fb = (ch + 1 < nblk) ? ch + 1 : nblk - 1, fa = (ch + 2 < nblk) ? ch + 2 : nblk - 1;
#define SP_DB(i) dma(st ^ 1, fb, SP_PW + (i))
#define SP_DA(i) dma(saf, fa, (i))
        if constexpr (!BSPREAD && TERMS == 3) {
            SP_SETD(a0, b0, ldB(b1, st, 2), SP_DA(0), (void)0)
            SP_SETD(a0, b1, ldA(a1, sa, 2), SP_DA(1), (void)0)
            SP_SETD(a1, b0, ldA(a2, sa, 1); ldB(b2, st, 1), SP_DA(2), (void)0)
            SP_SETD(a2, b2, ldB(b1, st, 3), SP_DA(3), (void)0)
            SP_SETD(a2, b1, ldA(a1, sa, 3), (void)0, (void)0)
        } else if constexpr (!BSPREAD) {
            SP_SETD(a0, b0, ldA(a1, sa, 2), SP_DA(0), SP_DA(1))
            SP_SETD(a1, b0, ldA(a2, sa, 1); ldB(b2, st, 1), SP_DA(2), SP_DA(3))
            SP_SETD(a2, b2, ldA(a1, sa, 3), (void)0, (void)0)
        } else if constexpr (TERMS == 3) {
            SP_SETD(a0, b0, ldB(b1, st, 2), SP_DB(0), SP_DB(1))                    // hi hi, s0   | B lo s0
            SP_SETD(a0, b1, ldA(a1, sa, 2), SP_DB(2), SP_DB(3))                    // hi lo, s0   | A lo s0
            SP_SETD(a1, b0, ldA(a2, sa, 1); ldB(b2, st, 1), SP_DA(0), SP_DA(1))    // lo hi, s0   | A hi s1, B hi s1
            SP_SETD(a2, b2, ldB(b1, st, 3), SP_DA(2), (void)0)                     // hi hi, s1   | B lo s1
            SP_SETD(a2, b1, ldA(a1, sa, 3), SP_DA(3), (void)0)                     // hi lo, s1   | A lo s1
        } else {
            SP_SETD(a0, b0, ldA(a1, sa, 2), SP_DB(0); SP_DB(1), SP_DB(2); SP_DB(3))              // hi hi, s0   | A lo s0
            SP_SETD(a1, b0, ldA(a2, sa, 1); ldB(b2, st, 1), SP_DA(0), SP_DA(1))                  // lo hi, s0   | A hi s1, B hi s1
            SP_SETD(a2, b2, ldA(a1, sa, 3), SP_DA(2), SP_DA(3))                                  // hi hi, s1   | A lo s1
        }
#undef SP_DB
#undef SP_DA
        // every read of this chunk's stages has been requested; B of block ch + 1 and A of block ch + 1 have landed once at most this
        // iteration's SP_PW A pieces are outstanding -> barrier, the next chunk's first fragments, the last MFMA set
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SP_PW) : "memory");
        __syncthreads();
        ldA(a0, san, 0);
        ldB(b0, st ^ 1, 0);
        SP_SB();
#pragma unroll
        for (int m = 0; m < SP_NP; ++m) {
            mma1(a1, b2, m);   // lo hi, s1
            if constexpr (!BSPREAD) {
                SP_SB();
                if (m >= SP_NP - SP_PW) dma(st, fa, m);   // B of block ch + 2 into the stage this chunk frees
                SP_SB();
            }
        }
        SP_SB();
        sa = san;
    }
    SP_DMA_WAIT();
    __syncthreads();   // staging memory is free for the epilogue
#undef SP_SETD
}

// the wave's [32][64]-float epilogue staging tile inside STAGE 1 of the ring (waves 0 .. SP_WAVES/2 - 1: A[1], the rest: B[1]), for
// persistent workgroups whose stage 0 already receives the next tile's first block during the epilogue
template <int NA>
__device__ __forceinline__ float* sp_stage1_tile(SmemSPn<NA>& sm, int wave) {
    static_assert(SP_WAVES / 2 * 8192 <= SP_STAGE, "staging tiles of half the waves fit one operand stage");
    return reinterpret_cast<float*>(wave < SP_WAVES / 2 ? &sm.A[1][wave * 8192] : &sm.B[1][(wave - SP_WAVES / 2) * 8192]);
}

// ---- NT, tall tile (round 4): 512 x 128 outputs -----------------------------------------------------------------------------------
// For products with at most 128 output columns (the token_projector, Model.py:140: 2048 -> 128): on the 256 x 256 tile half of every
// MFMA is spent on columns that do not exist.  Here the workgroup covers 512 rows x 128 columns: 8 waves stacked over the rows, 64 rows x
// 128 columns = 2 x 4 MFMA tiles each (128 accumulators, 24 fragment reads per 48 MFMAs -- the ratios of the square tile); stage = 64 KiB
// of A + 16 KiB of B, two stages = the whole 160 KiB of LDS.  Same LDS row format / swizzle / six-set schedule as sp_nt_mainloop.
constexpr int SPT_M = 512, SPT_N = 128;
constexpr int SPT_PA = 8, SPT_PB = 2;   // LDS-DMA pieces (8 rows x 128 B) per wave, stage and operand
typedef f32x16 SpAccT[2][4];
struct __attribute__((aligned(16))) SmemSPT {
    char A[2][SPT_M * 128];
    char B[2][SPT_N * 128];
};
__device__ __forceinline__ void sp_zero(SpAccT& acc) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}
// row / stored chunk of DMA piece i of this wave: A pieces i < SPT_PA (rows of the 512), B pieces i < SPT_PB (rows of the 128)
__device__ __forceinline__ void sp_tall_slot(int wave, int i, int per_wave, int lane, int& row, int& c) {
    row = (wave * per_wave + i) * 8 + (lane >> 3);
    c = (lane & 7) ^ ((row >> 1) & 7);
}
// dmaA(stage, block, i < SPT_PA), dmaB(stage, block, j < SPT_PB)
template <int TERMS = 3, class DmaA, class DmaB>
__device__ __forceinline__ void sp_nt_tall_mainloop(SmemSPT& sm, SpAccT& acc, int nblk, int wave, int lane, DmaA&& dmaA, DmaB&& dmaB) {
    const int l32 = lane & 31, kh = lane >> 5;
    uint32_t offA[2], offB[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wave * 64 + rt * 32 + l32;
        offA[rt] = r * 128 + ((kh ^ ((r >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int r = ct * 32 + l32;
        offB[ct] = r * 128 + ((kh ^ ((r >> 1) & 7)) << 4);
    }
    auto ldA = [&](u32x4 (&fa)[2], int st, int ks) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const u32x4*>(&sm.A[st][offA[rt] ^ (ks << 5)]);
    };
    auto ldB = [&](u32x4 (&fb)[4], int st, int ks) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fb[ct] = *reinterpret_cast<const u32x4*>(&sm.B[st][offB[ct] ^ (ks << 5)]);
    };
    auto mma1 = [&](const u32x4 (&fa)[2], const u32x4 (&fb)[4], int m) {
        const int rt = m >> 2, ct = m & 3;
        acc[rt][ct] = sp_mfma(fa[rt], fb[ct], acc[rt][ct]);
    };
#define SPT_SET(FA, FB, LOADS)                                         \
    mma1(FA, FB, 0);                                                   \
    SP_SB();                                                           \
    LOADS;                                                             \
    SP_SB();                                                           \
    _Pragma("unroll") for (int m = 1; m < 8; ++m) mma1(FA, FB, m);     \
    SP_SB();
    if (nblk <= 0) return;
    u32x4 a0[2], a1[2], a2[2], b0[4], b1[4], b2[4];
    auto dma_all = [&](int st, int f) {
#pragma unroll
        for (int i = 0; i < SPT_PA; ++i) dmaA(st, f, i);
#pragma unroll
        for (int j = 0; j < SPT_PB; ++j) dmaB(st, f, j);
    };
    dma_all(0, 0);
    SP_DMA_WAIT();
    __syncthreads();
    dma_all(1, nblk > 1 ? 1 : 0);
    ldA(a0, 0, 0);
    ldB(b0, 0, 0);
    for (int ch = 0; ch < nblk; ++ch) {
        const int st = ch & 1;
        if constexpr (TERMS == 3) {
            SPT_SET(a0, b0, ldB(b1, st, 2))                   // hi hi, s0   | B lo s0
            SPT_SET(a0, b1, ldA(a1, st, 2))                   // hi lo, s0   | A lo s0
            SPT_SET(a1, b0, ldA(a2, st, 1); ldB(b2, st, 1))   // lo hi, s0   | A hi s1, B hi s1
            SPT_SET(a2, b2, ldB(b1, st, 3))                   // hi hi, s1   | B lo s1
            SPT_SET(a2, b1, ldA(a1, st, 3))                   // hi lo, s1   | A lo s1
        } else {
            SPT_SET(a0, b0, ldA(a1, st, 2))
            SPT_SET(a1, b0, ldA(a2, st, 1); ldB(b2, st, 1))
            SPT_SET(a2, b2, ldA(a1, st, 3))
        }
        SP_DMA_WAIT();
        __syncthreads();
        ldA(a0, st ^ 1, 0);
        ldB(b0, st ^ 1, 0);
        SP_SB();
        const int f = (ch + 2 < nblk) ? ch + 2 : nblk - 1;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            mma1(a1, b2, m);                                  // lo hi, s1
            SP_SB();
            dmaA(st, f, m);
            if (m < SPT_PB) dmaB(st, f, m);
            SP_SB();
        }
    }
    SP_DMA_WAIT();
    __syncthreads();   // staging memory is free for the epilogue
#undef SPT_SET
}
// the wave's 64 x 128 sub-tile as row-contiguous float4s: emit(row, col, v) in tile coordinates (sp_epilogue_rows for the tall tile)
template <bool FULL, int INFLIGHT = 4, class Emit>
__device__ __forceinline__ void sp_epilogue_rows_tall(const SpAccT& acc, SmemSPT& sm, int wave, int lane, int rows_valid, Emit&& emit) {
    float* tile = reinterpret_cast<float*>(&sm) + wave * (32 * 64);
    const int l32 = lane & 31, rl = lane >> 4, c4 = lane & 15;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int cp = 0; cp < 2; ++cp) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) tile[acc_row(r, lane) * 64 + c2 * 32 + l32] = acc[rt][cp * 2 + c2][r];
#pragma unroll
            for (int h = 0; h < 8 / INFLIGHT; ++h) {
                f32x4 v[INFLIGHT];
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j)
                    v[j] = *reinterpret_cast<const f32x4*>(&tile[((h * INFLIGHT + j) * 4 + rl) * 64 + c4 * 4]);
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j) {
                    const int row = wave * 64 + rt * 32 + (h * INFLIGHT + j) * 4 + rl;
                    if (FULL || row < rows_valid) emit(row, cp * 64 + c4 * 4, v[j]);
                }
                SP_SB();
            }
        }
}

// ---- TN ---------------------------------------------------------------------------------------------------------------------------
// LDS stage image of one operand: [64 kr][256 columns] fp16, kr = plane * 32 + token of the 32-token chunk, 512 B per kr row, 64-B
// unit u of row kr stored at unit u ^ (kr & 3).  Fragments (8 consecutive tokens of the lane's output row / column) are gathered by
// ds_read_b64_tr_b16 (lane mapping: tile_engine_bf16.hpp).  k-step KS = plane * 2 + s: + KS * 16 kr rows = KS * 8192 B.
// DMA piece q (< SP_PW) of wave w for one operand: kr rows (SP_PW w + q) * 2 + (lane >> 5); stored 16-B chunk position lane & 31 holds the
// global chunk (lane & 31) ^ ((kr & 3) << 2)  (8 columns each).
template <int OFF>
__device__ __forceinline__ u32x2 sp_tr16(uint32_t lds_byte_addr) {
    u32x2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_byte_addr), "i"(OFF));
    return v;
}
struct SpFragA {
    u32x2 v[4][2];
};
struct SpFragB {
    u32x2 v[SPNCT][2];
};
template <int KS>
__device__ __forceinline__ void sp_tn_ldA(SpFragA& f, const uint32_t (&a)[4]) {
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
        f.v[rt][0] = sp_tr16<KS * 8192>(a[rt]);
        f.v[rt][1] = sp_tr16<KS * 8192 + 2048>(a[rt]);
    }
}
template <int KS>
__device__ __forceinline__ void sp_tn_ldB(SpFragB& f, const uint32_t (&b)[SPNCT]) {
#pragma unroll
    for (int ct = 0; ct < SPNCT; ++ct) {
        f.v[ct][0] = sp_tr16<KS * 8192>(b[ct]);
        f.v[ct][1] = sp_tr16<KS * 8192 + 2048>(b[ct]);
    }
}
__device__ __forceinline__ void sp_tn_mma(SpAcc& acc, const SpFragA& fa, const SpFragB& fb, int m) {
    const int rt = m / SPNCT, ct = m % SPNCT;
    const u32x4 av = {fa.v[rt][0].x, fa.v[rt][0].y, fa.v[rt][1].x, fa.v[rt][1].y};
    const u32x4 bv = {fb.v[ct][0].x, fb.v[ct][0].y, fb.v[ct][1].x, fb.v[ct][1].y};
    acc[rt][ct] = sp_mfma(av, bv, acc[rt][ct]);
}
#define SP_LGKM_WAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// acc[rt][ct] += sum over nch chunks of 32 tokens of A[t][wm*128 + rt*32 ..] B[t][wn*SP_WCOLS + ct*32 ..];  dma(stage, chunk, piece < SP_NP)
// TERMS = 2 drops al bh: the A operand (the activations of a dW product; B is the gradient) enters rounded to its hi plane.
template <int TERMS = 3, class Dma>
__device__ __forceinline__ void sp_tn_mainloop(SmemSP& sm, SpAcc& acc, int64_t nch, int wm, int wn, int lane, Dma&& dma) {
    const int g = lane >> 4, r = lane & 15;
    const int kb = (g >> 1) * 8 + (r >> 2);
    uint32_t a_0[4], b_0[SPNCT];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
        a_0[rt] = lds_addr_of(&sm.A[0][0]) + kb * 512 + (((wm * 128 + rt * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
#pragma unroll
    for (int ct = 0; ct < SPNCT; ++ct)
        b_0[ct] = lds_addr_of(&sm.B[0][0]) + kb * 512 + (((wn * SP_WCOLS + ct * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
    if (nch <= 0) return;
#pragma unroll
    for (int p = 0; p < SP_NP; ++p) dma(0, (int64_t)0, p);
    SP_DMA_WAIT();
    __syncthreads();
    {
        const int64_t f = nch > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < SP_NP; ++p) dma(1, f, p);
    }
    SpFragA a0, a1, a2;
    SpFragB b0, b1, b2;
    sp_tn_ldA<0>(a0, a_0);
    sp_tn_ldB<0>(b0, b_0);
    SP_LGKM_WAIT();
    SP_SB();
#define SP_TSET(FA, FB, LOADS)                                                  \
    sp_tn_mma(acc, FA, FB, 0);                                                  \
    SP_SB();                                                                    \
    LOADS;                                                                      \
    SP_SB();                                                                    \
    _Pragma("unroll") for (int m = 1; m < SP_NP; ++m) sp_tn_mma(acc, FA, FB, m); \
    SP_SB();                                                                    \
    SP_LGKM_WAIT();                                                             \
    SP_SB();
    for (int64_t ch = 0; ch < nch; ++ch) {
        const int st = (int)(ch & 1);
        uint32_t aA[4], aB[SPNCT], nA[4], nB[SPNCT];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            aA[rt] = a_0[rt] + st * SP_STAGE;
            nA[rt] = a_0[rt] + (st ^ 1) * SP_STAGE;
        }
#pragma unroll
        for (int ct = 0; ct < SPNCT; ++ct) {
            aB[ct] = b_0[ct] + st * SP_STAGE;
            nB[ct] = b_0[ct] + (st ^ 1) * SP_STAGE;
        }
        if constexpr (TERMS == 3) {
            SP_TSET(a0, b0, sp_tn_ldB<2>(b1, aB))                          // hi hi, s0 | B lo s0
            SP_TSET(a0, b1, sp_tn_ldA<2>(a1, aA))                          // hi lo, s0 | A lo s0
            SP_TSET(a1, b0, sp_tn_ldA<1>(a2, aA); sp_tn_ldB<1>(b2, aB))    // lo hi, s0 | A hi s1, B hi s1
            SP_TSET(a2, b2, sp_tn_ldB<3>(b1, aB))                          // hi hi, s1 | B lo s1
            SP_TSET(a2, b1, sp_tn_ldA<3>(a1, aA))                          // hi lo, s1 | A lo s1
        } else {
            SP_TSET(a0, b0, sp_tn_ldB<2>(b1, aB))                          // hi hi, s0 | B lo s0
            SP_TSET(a0, b1, sp_tn_ldA<1>(a2, aA); sp_tn_ldB<1>(b2, aB))    // hi lo, s0 | A hi s1, B hi s1
            SP_TSET(a2, b2, sp_tn_ldB<3>(b1, aB))                          // hi hi, s1 | B lo s1
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SP_SB();
        __syncthreads();
        sp_tn_ldA<0>(a0, nA);
        sp_tn_ldB<0>(b0, nB);
        SP_SB();
        const int64_t f = (ch + 2 < nch) ? ch + 2 : nch - 1;
#pragma unroll
        for (int m = 0; m < SP_NP; ++m) {
            if constexpr (TERMS == 3) sp_tn_mma(acc, a1, b2, m);       // lo hi, s1
            else sp_tn_mma(acc, a2, b1, m);                            // hi lo, s1
            SP_SB();
            dma(st, f, m);
            SP_SB();
        }
        SP_LGKM_WAIT();
        SP_SB();
    }
    SP_DMA_WAIT();
    __syncthreads();
#undef SP_TSET
}

// The TN loop on SmemSP3 (round 6): both operands of a dW product stream from HBM, so only A gets the third stage and the
// spread -- iteration ch issues A of chunk ch + 2 (into the A stage freed by chunk ch - 1) one piece per set behind MFMA 2 of its first
// four sets; B of chunk ch + 2 still goes into the stage chunk ch frees, inside the last set (four pieces instead of eight there).  Wait at
// the chunk end: vmcnt(SP_PW) -- B of chunk ch + 1 (issued in the previous last set) and A of chunk ch + 1 (an iteration ago) have landed.
template <int TERMS = 3, bool BSPREAD = (MDL_SP_TN_BSPREAD != 0), class Dma>   // BSPREAD = true: B too is issued inside the chunk (the placement of sp_nt_mainloop3)
__device__ __forceinline__ void sp_tn_mainloop3(SmemSP3& sm, SpAcc& acc, int64_t nch, int wm, int wn, int lane, Dma&& dma) {
    static_assert(SP_PW == 4, "piece placement below is written for four pieces per wave and operand");
    const int g = lane >> 4, r = lane & 15;
    const int kb = (g >> 1) * 8 + (r >> 2);
    uint32_t a_0[4], b_0[SPNCT];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
        a_0[rt] = lds_addr_of(&sm.A[0][0]) + kb * 512 + (((wm * 128 + rt * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
#pragma unroll
    for (int ct = 0; ct < SPNCT; ++ct)
        b_0[ct] = lds_addr_of(&sm.B[0][0]) + kb * 512 + (((wn * SP_WCOLS + ct * 32 + (g & 1) * 16 + (r & 3) * 4) * 2) ^ ((kb & 3) << 6));
    if (nch <= 0) return;
#pragma unroll
    for (int p = 0; p < SP_NP; ++p) dma(0, (int64_t)0, p);
    SP_DMA_WAIT();
    __syncthreads();
    {
        const int64_t f = nch > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < (BSPREAD ? SP_PW : SP_NP); ++p) dma(1, f, p);   // A (and B) of chunk 1 (A of chunk 2 follows inside iteration 0)
    }
    SpFragA a0, a1, a2;
    SpFragB b0, b1, b2;
    sp_tn_ldA<0>(a0, a_0);
    sp_tn_ldB<0>(b0, b_0);
    SP_LGKM_WAIT();
    SP_SB();
#define SP_TSETD2(FA, FB, LOADS, D1, D2)                                         \
    sp_tn_mma(acc, FA, FB, 0);                                                  \
    SP_SB();                                                                    \
    LOADS;                                                                      \
    SP_SB();                                                                    \
    sp_tn_mma(acc, FA, FB, 1);                                                  \
    sp_tn_mma(acc, FA, FB, 2);                                                  \
    SP_SB();                                                                    \
    D1;                                                                         \
    SP_SB();                                                                    \
    sp_tn_mma(acc, FA, FB, 3);                                                  \
    sp_tn_mma(acc, FA, FB, 4);                                                  \
    SP_SB();                                                                    \
    D2;                                                                         \
    SP_SB();                                                                    \
    _Pragma("unroll") for (int m = 5; m < SP_NP; ++m) sp_tn_mma(acc, FA, FB, m); \
    SP_SB();                                                                    \
    SP_LGKM_WAIT();                                                             \
    SP_SB();
#define SP_TSETD(FA, FB, LOADS, D1) SP_TSETD2(FA, FB, LOADS, D1, (void)0)
    int sa = 0;
    for (int64_t ch = 0; ch < nch; ++ch) {
        const int st = (int)(ch & 1);
        const int san = (sa == 2) ? 0 : sa + 1, saf = (san == 2) ? 0 : san + 1;
        uint32_t aA[4], aB[SPNCT], nA[4], nB[SPNCT];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            aA[rt] = a_0[rt] + sa * SP_STAGE;
            nA[rt] = a_0[rt] + san * SP_STAGE;
        }
#pragma unroll
        for (int ct = 0; ct < SPNCT; ++ct) {
            aB[ct] = b_0[ct] + st * SP_STAGE;
            nB[ct] = b_0[ct] + (st ^ 1) * SP_STAGE;
        }
        const int64_t fa = (ch + 2 < nch) ? ch + 2 : nch - 1;
        const int64_t fb1 = (ch + 1 < nch) ? ch + 1 : nch - 1;
#define SP_DA(i) dma(saf, fa, (i))
#define SP_DB(i) dma(st ^ 1, fb1, SP_PW + (i))
        if constexpr (BSPREAD && TERMS == 3) {
            SP_TSETD2(a0, b0, sp_tn_ldB<2>(b1, aB), SP_DB(0), SP_DB(1))
            SP_TSETD2(a0, b1, sp_tn_ldA<2>(a1, aA), SP_DB(2), SP_DB(3))
            SP_TSETD2(a1, b0, sp_tn_ldA<1>(a2, aA); sp_tn_ldB<1>(b2, aB), SP_DA(0), SP_DA(1))
            SP_TSETD2(a2, b2, sp_tn_ldB<3>(b1, aB), SP_DA(2), (void)0)
            SP_TSETD2(a2, b1, sp_tn_ldA<3>(a1, aA), SP_DA(3), (void)0)
        } else if constexpr (BSPREAD) {
            SP_TSETD2(a0, b0, sp_tn_ldB<2>(b1, aB), SP_DB(0); SP_DB(1), SP_DB(2); SP_DB(3))
            SP_TSETD2(a0, b1, sp_tn_ldA<1>(a2, aA); sp_tn_ldB<1>(b2, aB), SP_DA(0), SP_DA(1))
            SP_TSETD2(a2, b2, sp_tn_ldB<3>(b1, aB), SP_DA(2), SP_DA(3))
        } else if constexpr (TERMS == 3) {
            SP_TSETD(a0, b0, sp_tn_ldB<2>(b1, aB), SP_DA(0))                          // hi hi, s0 | B lo s0
            SP_TSETD(a0, b1, sp_tn_ldA<2>(a1, aA), SP_DA(1))                          // hi lo, s0 | A lo s0
            SP_TSETD(a1, b0, sp_tn_ldA<1>(a2, aA); sp_tn_ldB<1>(b2, aB), SP_DA(2))    // lo hi, s0 | A hi s1, B hi s1
            SP_TSETD(a2, b2, sp_tn_ldB<3>(b1, aB), SP_DA(3))                          // hi hi, s1 | B lo s1
            SP_TSETD(a2, b1, sp_tn_ldA<3>(a1, aA), (void)0)                           // hi lo, s1 | A lo s1
        } else {
            SP_TSETD(a0, b0, sp_tn_ldB<2>(b1, aB), SP_DA(0); SP_DA(1))                          // hi hi, s0 | B lo s0
            SP_TSETD(a0, b1, sp_tn_ldA<1>(a2, aA); sp_tn_ldB<1>(b2, aB), SP_DA(2); SP_DA(3))    // hi lo, s0 | A hi s1, B hi s1
            SP_TSETD(a2, b2, sp_tn_ldB<3>(b1, aB), (void)0)                                     // hi hi, s1 | B lo s1
        }
#undef SP_DA
#undef SP_DB
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SP_PW) : "memory");
        SP_SB();
        __syncthreads();
        sp_tn_ldA<0>(a0, nA);
        sp_tn_ldB<0>(b0, nB);
        SP_SB();
        const int64_t fb = (ch + 2 < nch) ? ch + 2 : nch - 1;
#pragma unroll
        for (int m = 0; m < SP_NP; ++m) {
            if constexpr (TERMS == 3) sp_tn_mma(acc, a1, b2, m);       // lo hi, s1
            else sp_tn_mma(acc, a2, b1, m);                            // hi lo, s1
            SP_SB();
            if constexpr (!BSPREAD) {
                if (m >= SP_NP - SP_PW) dma(st, fb, m);                // the four B pieces (pieces SP_PW .. SP_NP - 1) behind the last MFMAs
            }
            SP_SB();
        }
        SP_LGKM_WAIT();
        SP_SB();
        sa = san;
    }
    SP_DMA_WAIT();
    __syncthreads();
#undef SP_TSETD
#undef SP_TSETD2
}

// ---- epilogue through LDS ---------------------------------------------------------------------------------------------------------
// Hands the wave's 128 x SP_WCOLS sub-tile to emit(row, col, v) as row-contiguous float4s: columns col .. col + 3 of tile row `row` (tile
// coordinates); 16 lanes cover 256 contiguous bytes of a row.  FULL = false skips rows >= rows_valid.  Call after the main loop
// returned (staging memory free); wave-private LDS regions ([32][64] floats per wave), no block barrier inside.
template <bool FULL, int INFLIGHT = 4, class Emit>
__device__ __forceinline__ void sp_epilogue_rows(const SpAcc& acc, SmemSP& sm, int wave, int wm, int wn, int lane, int rows_valid,
                                                 Emit&& emit) {
    float* tile = reinterpret_cast<float*>(&sm) + wave * (32 * 64);
    const int l32 = lane & 31, rl = lane >> 4, c4 = lane & 15;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int cp = 0; cp < SPNCT / 2; ++cp) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) tile[acc_row(r, lane) * 64 + c2 * 32 + l32] = acc[rt][cp * 2 + c2][r];
#pragma unroll
            for (int h = 0; h < 8 / INFLIGHT; ++h) {   // INFLIGHT reads in flight, then their consumers (bounds the VGPRs)
                f32x4 v[INFLIGHT];
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j)
                    v[j] = *reinterpret_cast<const f32x4*>(&tile[((h * INFLIGHT + j) * 4 + rl) * 64 + c4 * 4]);
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j) {
                    const int row = wm * 128 + rt * 32 + (h * INFLIGHT + j) * 4 + rl;
                    if (FULL || row < rows_valid) emit(row, wn * SP_WCOLS + cp * 64 + c4 * 4, v[j]);
                }
                SP_SB();
            }
        }
}

// absmax bookkeeping: non-negative floats order like their bit patterns
// the same with ONE atomic per workgroup: `lds` = SP_WAVES floats nobody else touches until the workgroup ends.  Kernel-uniform call
// sites only (it synchronises).  The eight waves of a tile -- and the 256 tiles resident at once -- reach this point together.
__device__ __forceinline__ void sp_block_absmax(float* dst, float v, float* lds) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = lds[0];
#pragma unroll
        for (int w = 1; w < SP_WAVES; ++w) m = fmaxf(m, lds[w]);
        atomicMax(reinterpret_cast<unsigned int*>(dst), __float_as_uint(m));
    }
}
__device__ __forceinline__ void sp_atomic_absmax(float* dst, float v) {
    v = wave_max(v);
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned int*>(dst), __float_as_uint(v));
}
// scale 2^e with absmax * 2^e in [2^13, 2^14)  (1 for absmax = 0 / non-finite)
__device__ __forceinline__ float sp_scale_for(float absmax) {
    if (!(absmax > 0.f) || !(absmax < 3.0e38f)) return 1.f;
    int e;
    frexpf(absmax, &e);   // absmax = f 2^e, f in [0.5, 1)
    return ldexpf(1.f, 14 - e);
}
// the two planes of 8 consecutive values (already scaled)
__device__ __forceinline__ void sp_split8(const float (&v)[8], u32x4& hi, u32x4& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 h0 = (_Float16)v[2 * i], h1 = (_Float16)v[2 * i + 1];
        const _Float16 l0 = (_Float16)(v[2 * i] - (float)h0), l1 = (_Float16)(v[2 * i + 1] - (float)h1);
        h[i] = __builtin_bit_cast(uint32_t, h2{h0, h1});
        l[i] = __builtin_bit_cast(uint32_t, h2{l0, l1});
    }
    hi = u32x4{h[0], h[1], h[2], h[3]};
    lo = u32x4{l[0], l[1], l[2], l[3]};
}
// byte offset of the 16-B group holding columns k .. k + 7 (k % 8 == 0) of plane p within an image row
__device__ __forceinline__ int64_t sp_img_off(int k, int p) { return (int64_t)(k >> 5) * 128 + p * 64 + (k & 31) * 2; }

// Split-image store of 4 consecutive channels per lane, WHOLE LINES per store instruction (round 5).  Lane l of a full wave owns channels
// c .. c + 3 of an image row with c = c0 + 4 l (c0 % 256 == 0): lanes 2k and 2k + 1 hold the two halves of channels 8k' .. 8k' + 7.  The
// even lane hands its lo-plane 8 B to the odd lane and receives the odd lane's hi-plane 8 B (DPP quad_perm [1,0,3,2]); then the even lane
// writes 16 B of the hi plane and the odd lane 16 B of the lo plane: one store instruction of the wave = 8 whole 128-B lines, after fully
// coalesced float4 loads.  (8 B + 8 B per lane writes the 64-B halves of 8 lines per instruction, 16 B + 16 B from 32-B loads the halves
// of 16: 5.1-5.3 / 4.8 TB/s against 5.2-5.4 for this order and for a plain float4 copy on the same box, tools/micro/hbm_rate.hip
// patterns.)  The mirror image of the pooling kernels' image loads (abmil_pool.hip PoolLd<img_t>).  Every lane of the wave must call, with
// the same row.  v is scaled by s first.
// MEASURED in the product (profiles/r05q): no gain -- the dz pass gained its 5 % from the coalesced float4 loads alone (1.87 -> 1.77 ms with
// either store), the LayerNorm kernels are 1.5 % faster with the plain 8 B + 8 B stores.  Default MDL_IMG_PAIR=0 = those; =1 is the exchange.
#ifndef MDL_IMG_PAIR
#define MDL_IMG_PAIR 0
#endif
__device__ __forceinline__ void sp_img_store4(char* __restrict__ row, int c, const f32x4& v, float s) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    uint32_t h[2], l[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float a0 = v[2 * i] * s, a1 = v[2 * i + 1] * s;
        const _Float16 h0 = (_Float16)a0, h1 = (_Float16)a1;
        h[i] = __builtin_bit_cast(uint32_t, h2{h0, h1});
        l[i] = __builtin_bit_cast(uint32_t, h2{(_Float16)(a0 - (float)h0), (_Float16)(a1 - (float)h1)});
    }
#if MDL_IMG_PAIR
    const bool odd = (c >> 2) & 1;
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? h[0] : l[0]), 0xB1, 0xF, 0xF, false);
    const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(odd ? h[1] : l[1]), 0xB1, 0xF, 0xF, false);
    char* p = row + (int64_t)(c >> 5) * 128 + ((c & 31) & ~7) * 2 + (odd ? 64 : 0);
    *reinterpret_cast<u32x4*>(p) = odd ? u32x4{r0, r1, l[0], l[1]} : u32x4{h[0], h[1], r0, r1};
#else
    char* p = row + (int64_t)(c >> 5) * 128 + (c & 31) * 2;
    *reinterpret_cast<u32x2*>(p) = u32x2{h[0], h[1]};
    *reinterpret_cast<u32x2*>(p + 64) = u32x2{l[0], l[1]};
#endif
}

}  // namespace mdl
