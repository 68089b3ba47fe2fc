// tile_engine.hpp -- the fp32-MFMA tile engine shared by the gate kernels (abmil_gate.hip) and the pre_attn Linears
// (linear_fp32.hip): 128 x 256 output tile per 256-thread workgroup (4 waves as 2 x 2, 64 x 128 = 2 x 4
// v_mfma_f32_32x32x2_f32 accumulators per wave), BK = 16, every operand by LDS-DMA into two LDS stages of 24 KiB,
// 3 workgroups per CU.
//
// Round-2 structure (tools/micro/gemm_lab.hip holds the within-process A/B: 131 -> 145 TF on the 262144 x 2048 x 512 shape):
//   * SOFTWARE-PIPELINED INSIDE THE WAVE.  hipcc placed every fragment read right in front of the MFMAs that use it
//     (ds_read; s_waitcnt lgkmcnt(0); 4 MFMAs -- an exposed LDS round trip per 256 MFMA-cycles) and the next chunk's DMA
//     issue block (~40 instructions) un-overlapped at the chunk top.  Here the K-chunk is 8 steps of 8 MFMAs; the
//     fragments of step s+1 are requested right after the FIRST MFMA of step s (the registers they overwrite were last
//     read by step s-1, fully issued by then) and are needed 7 MFMAs (~450 cycles) later; __builtin_amdgcn_sched_barrier(0)
//     pins that order, the s_waitcnt placement stays the compiler's.
//   * ONE BARRIER PER CHUNK, BEFORE ITS LAST STEP.  At that point every fragment of the chunk has been requested, so
//     after the barrier the stage is free: the LDS-DMA of chunk ch+2 is issued BETWEEN the last step's MFMAs, together
//     with the first fragment reads of chunk ch+1 -- the post-barrier work hides under 8 MFMAs instead of stalling them.
//   * LDS-DMA in the saddr form (uniform 64-bit base in SGPRs + one 32-bit per-lane byte offset, inline asm): the
//     chunk advance is scalar arithmetic, 3 address VGPRs instead of 8.  hipcc does not count inline-asm loads, so the
//     engine waits (s_waitcnt vmcnt(0)) itself before the barrier that publishes a stage.
//   * Accumulators in AGPRs (__launch_bounds__(256) instead of (256, 2) makes hipcc select the AGPR form): 36-40 arch
//     VGPRs + 128 AGPRs = 3 waves per SIMD.
//   * EPILOGUE THROUGH LDS: the 32x32 MFMA layout gives a lane one column of 16 rows -> 128 dword stores per lane; a
//     wave-private 32 x 64 transpose in the (now free) staging memory turns them into 32 row-contiguous 16-B stores
//     (16 lanes = 256 B of one row).  With three co-resident workgroups in lockstep the store tail was fully exposed.
#pragma once
#include <type_traits>

#include "gate_common.hpp"

namespace mdl {

constexpr int TBM = 128, TBN = 256, TBK = 16;

struct TileSmem {
    float A[2][TBM * TBK];   // NN form: [row][16 k] XOR-swizzled row image; TN form: [16 k][128] K-major
    float B[2][TBK][TBN];    // K-major rows
};

#define TILE_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define TILE_SB() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ void tile_zero(f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// ---- DMA pieces -------------------------------------------------------------------------------------------------
// K-contiguous operand (rows of 16 k floats): wave w fills LDS slots [(2w+q)*64, +64), slot s = (row = s>>2, kq' = s&3)
// holds global 16-B chunk kq = kq' ^ ((row>>2)&3) of that row (bank conflicts removed on the SOURCE side; the fragment
// reads apply the same involution).  Rows >= rows_valid re-read the last valid row (their outputs are discarded).
__device__ __forceinline__ void rows_voff(uint32_t (&vo)[2], int64_t rows_valid, int64_t ld_floats, int wave, int lane) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sl = (wave * 2 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        int64_t rr = row;
        if (rr > rows_valid - 1) rr = rows_valid - 1;
        vo[q] = (uint32_t)(rr * ld_floats * 4 + kq * 16);
    }
}
__device__ __forceinline__ void rows_issue(const char* chunk_base, const uint32_t (&vo)[2], float* stageA, int wave) {
    glds16_s(vo[0], chunk_base, lds_addr_of(stageA + (wave * 2 + 0) * 256));
    glds16_s(vo[1], chunk_base, lds_addr_of(stageA + (wave * 2 + 1) * 256));
}
// K-major operand, 256 floats per k-row: wave w fills rows 4w .. 4w+3, one 1-KiB row per instruction.
// row_base = address of row (4w + q) for q = q0; the next row is row_stride bytes further.
__device__ __forceinline__ void krows_issue2(const char* row_base, int64_t row_stride, uint32_t vo, float (*stageB)[TBN], int wave,
                                             int q0) {
    glds16_s(vo, row_base, lds_addr_of(&stageB[wave * 4 + q0][0]));
    glds16_s(vo, row_base + row_stride, lds_addr_of(&stageB[wave * 4 + q0 + 1][0]));
}

// ---- NN main loop -------------------------------------------------------------------------------------------------
// acc[rt][ct] += sum over nch chunks of A[rows wm*64 + rt*32 ..][k] * B[k][colb[ct] ..].   dma(stage, chunk, piece) issues
// this wave's two LDS-DMA instructions of piece 0..2 for K-chunk `chunk` into stage `stage` (piece 0: the A rows;
// pieces 1, 2: B rows 4w+0..1 / 4w+2..3 by convention).  On return every DMA has landed and all waves have passed a barrier
// after their last fragment read: the staging memory is free.
// Generic form: the A stages start at Ab (a_stage floats apart, rows of 16 k), the B stages at Bb (rows of NB floats); the
// "tall" geometry (256 rows x 128 columns per workgroup, the 4 waves stacked in M: wm = wave) uses a_stage = 4096, NB = 128.
template <int NB, class Dma>
__device__ __forceinline__ void tile_loop_nn_g(f32x16 (&acc)[2][4], float* Ab, int a_stage, float* Bb, int nch, int wm,
                                               const int (&colb)[4], int lane, Dma&& dma) {
    const int l32 = lane & 31, kh = lane >> 5;
    int offA[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * TBK + ((kh ^ ((r >> 2) & 3)) << 2);
    }
    f32x4 fa0[2], fa1[2];   // A fragments of k-group 0 / 1 of the current chunk (rt = 0, 1); k-pair (8g+e | 8g+4+e) per half-wave
    float fb0[4], fb1[4];   // B fragments of even / odd steps
    auto ldA = [&](f32x4 (&fa)[2], int st, int g) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const f32x4*>(&Ab[st * a_stage + (offA[rt] ^ (g << 3))]);
    };
    auto ldB = [&](float (&fb)[4], int st, int s) {   // step s = 4 g + e
        const int g = s >> 2, e = s & 3;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fb[ct] = Bb[(st * TBK + 8 * g + 4 * kh + e) * NB + colb[ct] + l32];
    };
    auto mma1 = [&](const f32x4 (&fa)[2], int e, const float (&fb)[4], int m) {
        const int rt = m & 1, ct = m >> 1;
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rt][e], fb[ct], acc[rt][ct], 0, 0, 0);
    };
#define TILE_STEP(FA, E, FB, LOADS)                                             \
    mma1(FA, E, FB, 0);                                                         \
    TILE_SB();                                                                  \
    LOADS;                                                                      \
    TILE_SB();                                                                  \
    _Pragma("unroll") for (int m = 1; m < 8; ++m) mma1(FA, E, FB, m);           \
    TILE_SB();
    if (nch <= 0) return;
#pragma unroll
    for (int p = 0; p < 3; ++p) dma(0, 0, p);
    TILE_DMA_WAIT();
    __syncthreads();
    {
        const int f = nch > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 3; ++p) dma(1, f, p);
    }
    ldA(fa0, 0, 0);
    ldB(fb0, 0, 0);
    for (int ch = 0; ch < nch; ++ch) {
        const int st = ch & 1;
        TILE_STEP(fa0, 0, fb0, ldB(fb1, st, 1); ldA(fa1, st, 1))
        TILE_STEP(fa0, 1, fb1, ldB(fb0, st, 2))
        TILE_STEP(fa0, 2, fb0, ldB(fb1, st, 3))
        TILE_STEP(fa0, 3, fb1, ldB(fb0, st, 4))
        TILE_STEP(fa1, 0, fb0, ldB(fb1, st, 5))
        TILE_STEP(fa1, 1, fb1, ldB(fb0, st, 6))
        TILE_STEP(fa1, 2, fb0, ldB(fb1, st, 7))
        // last step: every read of stage st has been requested; the barrier waits for them (lgkmcnt) and for this wave's DMA
        // pieces of chunk ch+1 (vmcnt).  Then stage st is free for chunk ch+2 and stage st^1 is readable.  The last two
        // iterations re-fetch the last chunk into a stage nobody reads again (keeps the body branch-free).
        TILE_DMA_WAIT();
        __syncthreads();
        ldA(fa0, st ^ 1, 0);
        ldB(fb0, st ^ 1, 0);
        TILE_SB();
        const int f = (ch + 2 < nch) ? ch + 2 : nch - 1;
        mma1(fa1, 3, fb1, 0);
        TILE_SB();
        dma(st, f, 0);
        TILE_SB();
        mma1(fa1, 3, fb1, 1);
        TILE_SB();
        dma(st, f, 1);
        TILE_SB();
        mma1(fa1, 3, fb1, 2);
        TILE_SB();
        dma(st, f, 2);
        TILE_SB();
#pragma unroll
        for (int m = 3; m < 8; ++m) mma1(fa1, 3, fb1, m);
        TILE_SB();
    }
    TILE_DMA_WAIT();   // the redundant tail fetches
    __syncthreads();   // every wave is done with the staging buffers
}

template <class Dma>
__device__ __forceinline__ void tile_loop_nn(f32x16 (&acc)[2][4], TileSmem& sm, int nch, int wm, const int (&colb)[4], int lane,
                                             Dma&& dma) {
    tile_loop_nn_g<TBN>(acc, &sm.A[0][0], TBM * TBK, &sm.B[0][0][0], nch, wm, colb, lane, static_cast<Dma&&>(dma));
}

// ---- TN main loop -------------------------------------------------------------------------------------------------
// Both operands K-major: A image [16 k][128 rows], B image [16 k][256].  acc[rt][ct] += sum_k A[k][wm*64 + rt*32 ..] B[k][colb[ct] ..].
// dma(stage, chunk, piece): piece 0 = the A rows (2 instructions of two 512-B rows each), pieces 1, 2 = B rows.
// The MFMA's K = 2 takes k = 2*kk + (lane >> 5): 8 steps of one k-pair each.
template <class Dma>
__device__ __forceinline__ void tile_loop_tn(f32x16 (&acc)[2][4], TileSmem& sm, int64_t nch, int wm, const int (&colb)[4], int lane,
                                             Dma&& dma) {
    const int l32 = lane & 31, kh = lane >> 5;
    float (*As)[2][TBK][TBM] = reinterpret_cast<float (*)[2][TBK][TBM]>(&sm.A[0][0]);
    float fa0[2], fa1[2], fb0[4], fb1[4];
    auto ld = [&](float (&fa)[2], float (&fb)[4], int st, int s) {
        const int k = 2 * s + kh;
        fa[0] = (*As)[st][k][wm * 64 + l32];
        fa[1] = (*As)[st][k][wm * 64 + 32 + l32];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) fb[ct] = sm.B[st][k][colb[ct] + l32];
    };
    auto mma1 = [&](const float (&fa)[2], const float (&fb)[4], int m) {
        const int rt = m & 1, ct = m >> 1;
        acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rt], fb[ct], acc[rt][ct], 0, 0, 0);
    };
#define TILE_STEP_TN(FA, FB, LOADS)                                             \
    mma1(FA, FB, 0);                                                            \
    TILE_SB();                                                                  \
    LOADS;                                                                      \
    TILE_SB();                                                                  \
    _Pragma("unroll") for (int m = 1; m < 8; ++m) mma1(FA, FB, m);              \
    TILE_SB();
    if (nch <= 0) return;
#pragma unroll
    for (int p = 0; p < 3; ++p) dma(0, (int64_t)0, p);
    TILE_DMA_WAIT();
    __syncthreads();
    {
        const int64_t f = nch > 1 ? 1 : 0;
#pragma unroll
        for (int p = 0; p < 3; ++p) dma(1, f, p);
    }
    ld(fa0, fb0, 0, 0);
    for (int64_t ch = 0; ch < nch; ++ch) {
        const int st = (int)(ch & 1);
        TILE_STEP_TN(fa0, fb0, ld(fa1, fb1, st, 1))
        TILE_STEP_TN(fa1, fb1, ld(fa0, fb0, st, 2))
        TILE_STEP_TN(fa0, fb0, ld(fa1, fb1, st, 3))
        TILE_STEP_TN(fa1, fb1, ld(fa0, fb0, st, 4))
        TILE_STEP_TN(fa0, fb0, ld(fa1, fb1, st, 5))
        TILE_STEP_TN(fa1, fb1, ld(fa0, fb0, st, 6))
        TILE_STEP_TN(fa0, fb0, ld(fa1, fb1, st, 7))
        TILE_DMA_WAIT();
        __syncthreads();
        ld(fa0, fb0, st ^ 1, 0);
        TILE_SB();
        const int64_t f = (ch + 2 < nch) ? ch + 2 : nch - 1;
        mma1(fa1, fb1, 0);
        TILE_SB();
        dma(st, f, 0);
        TILE_SB();
        mma1(fa1, fb1, 1);
        TILE_SB();
        dma(st, f, 1);
        TILE_SB();
        mma1(fa1, fb1, 2);
        TILE_SB();
        dma(st, f, 2);
        TILE_SB();
#pragma unroll
        for (int m = 3; m < 8; ++m) mma1(fa1, fb1, m);
        TILE_SB();
    }
    TILE_DMA_WAIT();
    __syncthreads();
}

// ---- epilogue through LDS -------------------------------------------------------------------------------------------
// Hands the wave's 64 x 128 sub-tile to `emit` as row-contiguous float4s: emit(row_u, rl, lane_col, v, cp) gets the 4 consecutive
// columns lane_col .. lane_col+3 (tile coordinates, via colb) of tile row row_u + rl, where row_u is wave-uniform and
// rl = lane >> 4; 16 lanes cover 256 contiguous bytes of one row, so the caller's stores are 16-B wide and can use a
// uniform base + 32-bit lane offset.  cp (compile-time after unrolling) = 0 for the accumulator column tiles 0,1 and 1 for 2,3.  FULL = false additionally skips rows >= rows_valid (ragged last tile).
// Must be called after tile_loop_* returned (staging memory free); wave-private LDS regions, no block barrier inside.
template <bool FULL, int INFLIGHT = 4, class Emit>
__device__ __forceinline__ void tile_epilogue_rows(const f32x16 (&acc)[2][4], TileSmem& sm, int wave, int wm, const int (&colb)[4],
                                                   int lane, int rows_valid, Emit&& emit) {
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
            // 16 lanes x 4 columns = the two 32-column accumulator tiles cp*2, cp*2+1
            const int lane_col = ((c4 >> 3) ? colb[cp * 2 + 1] : colb[cp * 2]) + (c4 & 7) * 4;
#pragma unroll
            for (int h = 0; h < 8 / INFLIGHT; ++h) {   // INFLIGHT reads in flight, then their consumers (bounds the VGPRs)
                f32x4 v[INFLIGHT];
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j)
                    v[j] = *reinterpret_cast<const f32x4*>(&tile[((h * INFLIGHT + j) * 4 + rl) * 64 + c4 * 4]);
#pragma unroll
                for (int j = 0; j < INFLIGHT; ++j) {
                    const int row_u = wm * 64 + rt * 32 + (h * INFLIGHT + j) * 4;
                    if (FULL || row_u + rl < rows_valid) emit(row_u, rl, lane_col, v[j], cp);
                }
                TILE_SB();
            }
        }
}

}  // namespace mdl
