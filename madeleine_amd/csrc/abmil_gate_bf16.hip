// abmil_gate_bf16.hip -- A2 in the bf16 mode (the reference's `precision: bfloat16` autocast runs):
// gated attention scores of all heads with bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 2.5 PFLOP/s
// dense peak = 16x the fp32 matrix rate), fp32 epilogues.  Same maths as abmil_gate.hip (reference
// madeleine/models/abmil.py:41-68), same dropout counter hash, same partial-score / finalize scheme.
//
// All three contractions are "NT" products of two K-contiguous row images, so ONE tile engine serves them:
//     C[m, n] = sum_k A[m][k] B[n][k]      tile 128 x 256, BK = 32 bf16 (64 B per tile row), 4 waves (2 x 2) x (2 x 4) MFMA tiles
//   forward : A = E rows (tokens)            B = [Wa;Wb] rows (bf16 copy, K = 512)        fused activation / dropout / wc epilogue
//   dX      : A = dz rows (tokens, K = 1024) B = [Wa;Wb]^T rows (bf16 transposed copy)    dE (bf16) (+)= C
//   dW      : "TN" instead: A = E rows, B = dz rows, both token-major (K-major); the fragments (8 CONSECUTIVE k per lane) are
//             gathered from the K-major LDS image by ds_read_b64_tr_b16 -- no transposed copies (round 1 made E^T and dz^T
//             per backward: 1.3 ms of HBM-bound passes at config 2).  fp32 slabs over token splits, reduced by gate_reduce_w.
// Operands reach LDS by LDS-DMA (global_load_lds_dwordx4) into the same XOR-swizzled 64-B row image the fp32 kernels
// use (16-B chunk kq of row r is stored at slot kq ^ ((r >> 2) & 3)), read back with conflict-free ds_read_b128.
#include <cstdlib>
#include <type_traits>

#include "tile_engine_bf16.hpp"

namespace mdl {

// ------------------------------------------------------------------------------------------------
// weight images (once per call; 4 MiB)
//   WK [H][1024 = a j | b j][512 k]   rows K-contiguous: B operand of the forward
//   WN [H][512 e][1024 = a j | b j]   rows = input channel, K = gate column: B operand of dX
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_wk_bf16_kernel(const float* __restrict__ Wa, const float* __restrict__ Wb,
                                                           bf16_t* __restrict__ WK, int H) {
    const int64_t i4 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i4 >= (int64_t)H * 1024 * HID) return;
    const int k = (int)(i4 % HID), r = (int)((i4 / HID) % 1024), c = (int)(i4 / ((int64_t)1024 * HID));
    const float* src = (r < HID) ? Wa + ((int64_t)c * HID + r) * HID + k : Wb + ((int64_t)c * HID + r - HID) * HID + k;
    st4(WK + i4, ld4(src));
}

__global__ __launch_bounds__(256) void gate_wn_bf16_kernel(const float* __restrict__ Wa, const float* __restrict__ Wb,
                                                           bf16_t* __restrict__ WN) {
    __shared__ float tile[32][33];
    const int c = blockIdx.z, jb = blockIdx.y * 32, kb = blockIdx.x * 32;  // jb over 1024 (a | b), kb over 512
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ W = (jb < HID) ? Wa + ((int64_t)c * HID + jb) * HID : Wb + ((int64_t)c * HID + jb - HID) * HID;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + i * 8][tx] = W[(int64_t)(ty + i * 8) * HID + kb + tx];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) WN[((int64_t)c * HID + kb + ty + i * 8) * 1024 + jb + tx] = (bf16_t)tile[tx][ty + i * 8];
}

// ================================================================================================
// forward
// ================================================================================================
// DM: dropout mode 0 = off, 1 = counter-hash RNG, 2 = explicit uint8 masks; SAVE: store the activations (one code path per
// instantiation keeps the unrolled epilogue inside the register budget).
template <int DM, bool SAVE>
__global__ __launch_bounds__(256, 2) void gate_fwd_bf16_kernel(const bf16_t* __restrict__ E, int64_t ldE,
                                                               const bf16_t* __restrict__ WK, const float* __restrict__ ba,
                                                               const float* __restrict__ bb, const float* __restrict__ wc,
                                                               float* __restrict__ part, bf16_t* __restrict__ act_a,
                                                               bf16_t* __restrict__ act_b, int64_t T, int H, int n_ttiles,
                                                               DropCfg drop) {
    __shared__ SmemNT sm;
    const int tid = threadIdx.x, lane = tid & 63, wm = (tid >> 6) >> 1, wn = (tid >> 6) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int jt = xh.li % GATE_JT, c = xh.c, tt = (xh.li / GATE_JT) * xh.nshare + xh.share;
    if (tt >= n_ttiles) return;  // block-uniform
    const int64_t t0 = (int64_t)tt * BBM;
    const int j0 = jt * 128;

    const bf16_t* srcA[2];
    const bf16_t* srcB[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int row, kq;
        nt_slot(wave * 2 + q, lane, row, kq);
        int64_t t = t0 + row;
        if (t > T - 1) t = T - 1;  // rows past T re-read row T-1 (discarded in the epilogue)
        srcA[q] = E + t * ldE + (int64_t)c * HID + kq * 8;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row, kq;
        nt_slot(wave * 4 + q, lane, row, kq);
        const int wrow = (row < 128) ? (j0 + row) : (HID + j0 + row - 128);  // a columns, then b columns
        srcB[q] = WK + ((int64_t)c * 1024 + wrow) * HID + kq * 8;
    }
    auto issue = [&](int st, int64_t ch) {
        const int k0 = (int)ch * BBK;
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(srcA[q] + k0, &sm.A[st][(wave * 2 + q) * 512]);
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(srcB[q] + k0, &sm.B[st][(wave * 4 + q) * 512]);
    };
    const int colb[4] = {wn * 64, wn * 64 + 32, 128 + wn * 64, 128 + wn * 64 + 32};  // a, a, b, b
    int offA[2], offB[4];
    nt_offsets(wm, colb, lane, offA, offB);
    f32x16 acc[2][4];
    zero_acc8(acc);
    nt_mainloop(sm, acc, HID / BBK, issue, offA, offB);

    // ---- epilogue: 4 passes (rt, ct) of a 32-row x (32 a | 32 b)-column block through a wave-private LDS tile.
    // (1) accumulator layout: a = tanh(za + ba), b = sigmoid(zb + bb), rounded to bf16 like the stored copies the backward
    //     re-reads -> tile;  (2) transposed layout, lane = (row = lane >> 2, 8 columns): 16-B stores of the activations (the
    //     accumulator layout gives 2-byte stores 64 B apart), dropout, wc-weighted partial sum reduced over the row's 4 lanes.
    const int l32 = lane & 31;
    float* tile = reinterpret_cast<float*>(&sm) + wave * (32 * 64);
    float* sred = reinterpret_cast<float*>(&sm) + 4 * (32 * 64) + wn * BBM + wm * 64;   // [2 (wn)][128 rows]
    const int g4 = lane & 3, r16 = lane >> 2;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int jc = j0 + wn * 64 + ct * 32;                 // first gate column of this pass
            const float ta = 2.f * MDL_LOG2E * ba[c * HID + jc + l32], tb = -MDL_LOG2E * bb[c * HID + jc + l32];   // (folded, see common.hpp)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                tile[acc_row(r, lane) * 64 + l32] = (float)(bf16_t)gate_tanh_pre(acc[rt][ct][r], 2.f * MDL_LOG2E, ta);
                tile[acc_row(r, lane) * 64 + 32 + l32] = (float)(bf16_t)gate_sigmoid_pre(acc[rt][2 + ct][r], -MDL_LOG2E, tb);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            const float inv2 = drop.inv * drop.inv;   // both dropout factors folded into wc: one select per (a, b) pair
            const f32x4 wlo = *reinterpret_cast<const f32x4*>(wc + c * HID + jc + g4 * 8) * inv2;
            const f32x4 whi = *reinterpret_cast<const f32x4*>(wc + c * HID + jc + g4 * 8 + 4) * inv2;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = i * 16 + r16;
                const f32x4 alo = *reinterpret_cast<const f32x4*>(&tile[row * 64 + g4 * 8]);
                const f32x4 ahi = *reinterpret_cast<const f32x4*>(&tile[row * 64 + g4 * 8 + 4]);
                const f32x4 blo = *reinterpret_cast<const f32x4*>(&tile[row * 64 + 32 + g4 * 8]);
                const f32x4 bhi = *reinterpret_cast<const f32x4*>(&tile[row * 64 + 32 + g4 * 8 + 4]);
                float sum = 0.f;
                if (t0 + wm * 64 + rt * 32 + row < T) {
                    const int64_t idx = ((t0 + wm * 64 + rt * 32) * H + c) * HID + jc + (uint32_t)(row * H * HID + g4 * 8);
                    const uint32_t rkey = drop_row_key(drop, idx);   // idx % 8 == 0: the 8 elements share the high word
                    if (SAVE) {
                        st8_bf16(act_a + idx, alo, ahi);
                        st8_bf16(act_b + idx, blo, bhi);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bool keep_a, keep_b;
                        gate_keep2_fwd<DM>(drop, idx, e, rkey, keep_a, keep_b);
                        const float a = e < 4 ? alo[e & 3] : ahi[e & 3], b = e < 4 ? blo[e & 3] : bhi[e & 3];
                        const float w = e < 4 ? wlo[e & 3] : whi[e & 3];
                        sum = fmaf((keep_a && keep_b) ? a * b : 0.f, w, sum);
                    }
                }
                sum += __shfl_xor(sum, 1, 64);
                sum += __shfl_xor(sum, 2, 64);
                if (g4 == 0) {
                    if (ct == 0) sred[rt * 32 + row] = sum;
                    else sred[rt * 32 + row] += sum;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    __syncthreads();
    if (tid < BBM) {
        const int64_t t = t0 + tid;
        const float* sr = reinterpret_cast<const float*>(&sm) + 4 * (32 * 64);
        if (t < T) part[(t * H + c) * GATE_JT + jt] = sr[tid] + sr[BBM + tid];
    }
}

// ------------------------------------------------------------------------------------------------
// Round 3: the forward on the 256 x 256 x 64 tile (nt256_mainloop: whole 128-B lines per row and chunk, 8 waves as 2 x 4, 128 x 64 per
// wave) for T >= 4096.  Tile = 256 tokens x (128 a | 128 b) gate columns; tile column n = wn * 64 + ct * 32 + l: ct = 0 -> a column
// j0 + wn * 32 + l, ct = 1 -> the b column of the same j, so a wave holds za and zb of the same (token, j) in acc[rt][0] / acc[rt][1]
// (the layout of abmil_gate_split.hip).  Epilogue as above: 4 passes of 32 rows x (32 a | 32 b) through the wave's LDS tile.
// ------------------------------------------------------------------------------------------------
// Round 5: persistent workgroups.  With one workgroup per CU (128 KiB of stages) nothing overlaps a tile's prologue -- workgroup launch,
// first chunk's memory latency -- with the previous tile, and K = 512 is only 8 chunks.  A persistent workgroup runs several tiles back
// to back and requests the NEXT tile's first chunk into stage 0 before the epilogue of the current one; the epilogue stages through
// stage 1's memory (waves 0-3: A[1], waves 4-7: B[1]) and the row sums have their own 4 KiB.  PMODE as in abmil_gate_split.hip:
// 0 = one tile per workgroup, 1 = the GATE_JT column tiles of a token tile, 2 = GATE_PT16 token tiles of one column tile.
#ifndef MDL_GATE_BF16_PMODE
#define MDL_GATE_BF16_PMODE 1
#endif
constexpr int GATE_PT16 = 4;
template <int DM, bool SAVE, int PMODE>
__global__ __launch_bounds__(512) void gate_fwd256_bf16_kernel(const bf16_t* __restrict__ E, int64_t ldE, const bf16_t* __restrict__ WK,
                                                               const float* __restrict__ ba, const float* __restrict__ bb,
                                                               const float* __restrict__ wc, float* __restrict__ part,
                                                               bf16_t* __restrict__ act_a, bf16_t* __restrict__ act_b, int64_t T, int H,
                                                               int n_ttiles, DropCfg drop) {
    __shared__ SmemQ sm;
    __shared__ float sred_s[4 * QM];   // [4 (wn)][256 rows]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int c = xh.c;
    constexpr int n_k = PMODE == 1 ? GATE_JT : PMODE == 2 ? GATE_PT16 : 1;
    auto tt_of = [&](int k) { return (PMODE == 1 ? xh.li : PMODE == 2 ? (xh.li / GATE_JT) * GATE_PT16 + k : xh.li / GATE_JT) * xh.nshare + xh.share; };
    auto jt_of = [&](int k) { return PMODE == 1 ? k : xh.li % GATE_JT; };
    if (tt_of(0) >= n_ttiles) return;  // block-uniform

    uint32_t voB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row, ch;
        nt256_slot(wave, i, lane, row, ch);
        const int wrow = ((row >> 5) & 1) * HID + (row >> 6) * 32 + (row & 31);   // relative to the tile's first gate column j0
        voB[i] = (uint32_t)(wrow * (HID * 2) + ch * 16);
    }
    auto a_offsets = [&](int64_t t0, uint32_t (&vo)[4]) {   // rows past T re-read the last valid row (discarded)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int row, ch;
            nt256_slot(wave, i, lane, row, ch);
            int64_t ra = row;
            if (t0 + ra > T - 1) ra = T - 1 - t0;
            vo[i] = (uint32_t)(ra * ldE * 2 + ch * 16);
        }
    };
    const int l32 = lane & 31;
    float* tile = reinterpret_cast<float*>(wave < 4 ? &sm.A[1][wave * 8192] : &sm.B[1][(wave - 4) * 8192]);
    float* sred = sred_s + wn * QM + wm * 128;
    const int g4 = lane & 3, r16 = lane >> 2;
    const float inv2 = drop.inv * drop.inv;   // both dropout factors folded into wc: one select per (a, b) pair

    for (int k = 0; k < n_k; ++k) {
    const int tt = tt_of(k), jt = jt_of(k);
    if (tt >= n_ttiles) break;   // block-uniform (PMODE 2: a short last group)
    const int64_t t0 = (int64_t)tt * QM;
    const int j0 = jt * 128;
    const char* baseA = reinterpret_cast<const char*>(E + t0 * ldE + (int64_t)c * HID);
    const char* baseB = reinterpret_cast<const char*>(WK + ((int64_t)c * 1024 + j0) * HID);
    uint32_t voA[4];
    a_offsets(t0, voA);
    auto dma = [&](int st, int64_t f, int piece) {
        const int i = piece & 3;
        if (piece < 4) glds16_s(voA[i], baseA + f * (QK * 2), lds_addr_of(&sm.A[st][(wave * 4 + i) * 1024]));
        else glds16_s(voB[i], baseB + f * (QK * 2), lds_addr_of(&sm.B[st][(wave * 4 + i) * 1024]));
    };
    f32x16 acc[4][2];
    nt256_mainloop(sm, acc, HID / QK, wm, wn, lane, dma, k > 0);
    if (k + 1 < n_k && tt_of(k + 1) < n_ttiles) {   // the next tile's first chunk travels during this epilogue (stage 0 is free, the epilogue is in stage 1)
        const int64_t t0n = (int64_t)tt_of(k + 1) * QM;
        const char* baseAn = reinterpret_cast<const char*>(E + t0n * ldE + (int64_t)c * HID);
        const char* baseBn = reinterpret_cast<const char*>(WK + ((int64_t)c * 1024 + jt_of(k + 1) * 128) * HID);
        uint32_t voAn[4];
        a_offsets(t0n, voAn);
#pragma unroll
        for (int piece = 0; piece < 8; ++piece) {
            const int i = piece & 3;
            if (piece < 4) glds16_s(voAn[i], baseAn, lds_addr_of(&sm.A[0][(wave * 4 + i) * 1024]));
            else glds16_s(voB[i], baseBn, lds_addr_of(&sm.B[0][(wave * 4 + i) * 1024]));
        }
    }

    const int jc = j0 + wn * 32;
    const float ta = 2.f * MDL_LOG2E * ba[c * HID + jc + l32], tb = -MDL_LOG2E * bb[c * HID + jc + l32];   // (folded, see common.hpp)
    const f32x4 wlo = *reinterpret_cast<const f32x4*>(wc + c * HID + jc + g4 * 8) * inv2;
    const f32x4 whi = *reinterpret_cast<const f32x4*>(wc + c * HID + jc + g4 * 8 + 4) * inv2;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            tile[acc_row(r, lane) * 64 + l32] = (float)(bf16_t)gate_tanh_pre(acc[rt][0][r], 2.f * MDL_LOG2E, ta);
            tile[acc_row(r, lane) * 64 + 32 + l32] = (float)(bf16_t)gate_sigmoid_pre(acc[rt][1][r], -MDL_LOG2E, tb);
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = i * 16 + r16;
            const f32x4 alo = *reinterpret_cast<const f32x4*>(&tile[row * 64 + g4 * 8]);
            const f32x4 ahi = *reinterpret_cast<const f32x4*>(&tile[row * 64 + g4 * 8 + 4]);
            const f32x4 blo = *reinterpret_cast<const f32x4*>(&tile[row * 64 + 32 + g4 * 8]);
            const f32x4 bhi = *reinterpret_cast<const f32x4*>(&tile[row * 64 + 32 + g4 * 8 + 4]);
            float sum = 0.f;
            if (t0 + wm * 128 + rt * 32 + row < T) {
                const int64_t idx = ((t0 + wm * 128 + rt * 32) * H + c) * HID + jc + (uint32_t)(row * H * HID + g4 * 8);
                const uint32_t rkey = drop_row_key(drop, idx);
                if (SAVE) {
                    st8_bf16(act_a + idx, alo, ahi);
                    st8_bf16(act_b + idx, blo, bhi);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    bool keep_a, keep_b;
                    gate_keep2_fwd<DM>(drop, idx, e, rkey, keep_a, keep_b);
                    const float a = e < 4 ? alo[e & 3] : ahi[e & 3], b = e < 4 ? blo[e & 3] : bhi[e & 3];
                    const float w = e < 4 ? wlo[e & 3] : whi[e & 3];
                    sum = fmaf((keep_a && keep_b) ? a * b : 0.f, w, sum);
                }
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            if (g4 == 0) sred[rt * 32 + row] = sum;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    if (tid < QM) {
        const int64_t t = t0 + tid;
        const float* sr = sred_s;
        if (t < T) part[(t * H + c) * GATE_JT + jt] = ((sr[tid] + sr[QM + tid]) + sr[2 * QM + tid]) + sr[3 * QM + tid];
    }
    }   // k (the next tile's main loop has barriers between these reads of sred_s and its epilogue's writes)
}

// ================================================================================================
// dX: dE[t, c, n0 + n] (+)= sum_j dz[t, c, j] WN[c][n0 + n][j]
// ================================================================================================
__global__ __launch_bounds__(256, 2) void gate_dx_bf16_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ WN,
                                                              bf16_t* __restrict__ dE, int64_t ldE, int accumulate, int64_t T,
                                                              int H, PoolTerm pt) {
    __shared__ SmemNT sm;
    const int tid = threadIdx.x, lane = tid & 63, wm = (tid >> 6) >> 1, wn = (tid >> 6) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int nt = xh.li % 2, c = xh.c, tt = (xh.li / 2) * xh.nshare + xh.share;
    const int64_t t0 = (int64_t)tt * BBM;
    if (t0 >= T) return;  // block-uniform
    const int n0 = nt * BBN;

    const bf16_t* srcA[2];
    const bf16_t* srcB[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int row, kq;
        nt_slot(wave * 2 + q, lane, row, kq);
        int64_t t = t0 + row;
        if (t > T - 1) t = T - 1;
        srcA[q] = dz + (t * H + c) * 1024 + kq * 8;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row, kq;
        nt_slot(wave * 4 + q, lane, row, kq);
        srcB[q] = WN + ((int64_t)c * HID + n0 + row) * 1024 + kq * 8;
    }
    auto issue = [&](int st, int64_t ch) {
        const int k0 = (int)ch * BBK;
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(srcA[q] + k0, &sm.A[st][(wave * 2 + q) * 512]);
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(srcB[q] + k0, &sm.B[st][(wave * 4 + q) * 512]);
    };
    const int colb[4] = {wn * 128, wn * 128 + 32, wn * 128 + 64, wn * 128 + 96};
    int offA[2], offB[4];
    nt_offsets(wm, colb, lane, offA, offB);
    // fused A3 term: softmax weight and d_pooled row of every tile row, once per row (see sp_gate_dx_kernel)
    float row_w = 0.f;
    int row_off = 0;
    if (pt.scores && tid < BBM && t0 + tid < T) {
        int bag;
        row_w = pool_term_weight(pt, t0 + tid, c, H, bag);
        row_off = (bag * H + c) * HID;
    }
    f32x16 acc[2][4];
    zero_acc8(acc);
    nt_mainloop(sm, acc, 1024 / BBK, issue, offA, offB);

    // dE (bf16) leaves through the LDS transpose: 8 columns = one 16-B store per lane, 128 contiguous bytes per row
    float* tile = reinterpret_cast<float*>(&sm) + wave * (32 * 64);
    float* rw_s = reinterpret_cast<float*>(&sm) + 4 * (32 * 64);   // behind the four waves' transpose areas
    int* ro_s = reinterpret_cast<int*>(rw_s + BBM);
    if (pt.scores) {
        __syncthreads();   // every wave has left the main loop: the staging memory is free
        if (tid < BBM) {
            rw_s[tid] = row_w;
            ro_s[tid] = row_off;
        }
        __syncthreads();
    }
    const float* dpb = pt.d_pooled + n0;
    bf16_t* ob = dE + t0 * ldE + (int64_t)c * HID + n0;
    auto emit = [&](int row_u, int rl, int lane_col, const f32x4& lo, const f32x4& hi, int) {
        bf16_t* o = ob + (int64_t)row_u * ldE + ((uint32_t)rl * (uint32_t)ldE + (uint32_t)lane_col);
        f32x4 a = lo, b = hi;
        if (pt.scores) {  // fused A3 term: + w[t,c] * d_pooled[bag(t), c, :]
            const float w = rw_s[row_u + rl];
            const float* __restrict__ dp = dpb + (ro_s[row_u + rl] + lane_col);
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dp), d1 = *reinterpret_cast<const f32x4*>(dp + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = fmaf(w, d0[i], lo[i]);
                b[i] = fmaf(w, d1[i], hi[i]);
            }
        }
        if (accumulate) {
            const bf16x8 old = *reinterpret_cast<const bf16x8*>(o);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] += (float)old[i];
                b[i] += (float)old[4 + i];
            }
        }
        st8_bf16(o, a, b);
    };
    if (t0 + BBM <= T) epilogue_rows8<true>(acc, tile, wm, colb, lane, BBM, emit);
    else epilogue_rows8<false>(acc, tile, wm, colb, lane, (int)(T - t0), emit);
}

// ------------------------------------------------------------------------------------------------
// dX on the 256 x 256 x 64 tile (tile_engine_bf16.hpp: the long contraction K = 1024 is where that tile wins): same products and
// epilogue as gate_dx_bf16_kernel, 256 token rows x 256 of the head's 512 input channels per workgroup of 8 waves.
// ------------------------------------------------------------------------------------------------
template <int NA>   // NA = 3: nt256_mainloop3 on SmemQ3
__global__ __launch_bounds__(512) void gate_dx256_bf16_kernel(const bf16_t* __restrict__ dz, const bf16_t* __restrict__ WN,
                                                              bf16_t* __restrict__ dE, int64_t ldE, int accumulate, int64_t T, int H,
                                                              PoolTerm pt) {
    __shared__ typename std::conditional<NA == 3, SmemQ3, SmemQ>::type sm3;
    SmemQ& sm = reinterpret_cast<SmemQ&>(sm3);   // (epilogue staging: the first 128 KiB of the drained ring)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int nt = xh.li % 2, c = xh.c, tt = (xh.li / 2) * xh.nshare + xh.share;
    const int64_t t0 = (int64_t)tt * QM;
    if (t0 >= T) return;  // block-uniform
    const int n0 = nt * QN;

    const char* baseA = reinterpret_cast<const char*>(dz + (t0 * H + c) * 1024);
    const char* baseB = reinterpret_cast<const char*>(WN + ((int64_t)c * HID + n0) * 1024);
    const uint32_t rowA = (uint32_t)H * 1024u * 2u;
    uint32_t voA[4], voB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row, ch;
        nt256_slot(wave, i, lane, row, ch);
        int64_t ra = row;
        if (t0 + ra > T - 1) ra = T - 1 - t0;
        voA[i] = (uint32_t)ra * rowA + ch * 16;
        voB[i] = (uint32_t)row * 2048u + ch * 16;
    }
    auto dma = [&](int st, int64_t f, int piece) {
        const int i = piece & 3;
        if (piece < 4) glds16_s(voA[i], baseA + f * (QK * 2), lds_addr_of(&sm3.A[st][(wave * 4 + i) * 1024]));
        else glds16_s(voB[i], baseB + f * (QK * 2), lds_addr_of(&sm3.B[st][(wave * 4 + i) * 1024]));
    };
    // fused A3 term: softmax weight and d_pooled row of every tile row, once per row (see sp_gate_dx_kernel)
    float row_w = 0.f;
    int row_off = 0;
    if (pt.scores && tid < QM && t0 + tid < T) {
        int bag;
        row_w = pool_term_weight(pt, t0 + tid, c, H, bag);
        row_off = (bag * H + c) * HID;
    }
    f32x16 acc[4][2];
    if constexpr (NA == 3) nt256_mainloop3(sm3, acc, 1024 / QK, wm, wn, lane, dma);
    else nt256_mainloop(sm3, acc, 1024 / QK, wm, wn, lane, dma);

    float* rw_s = reinterpret_cast<float*>(&sm) + 8 * (32 * 64);   // behind the eight waves' transpose areas
    int* ro_s = reinterpret_cast<int*>(rw_s + QM);
    if (pt.scores) {
        __syncthreads();   // every wave has left the main loop: the staging memory is free
        if (tid < QM) {
            rw_s[tid] = row_w;
            ro_s[tid] = row_off;
        }
        __syncthreads();
    }
    const float* dpb = pt.d_pooled + n0;
    bf16_t* ob = dE + t0 * ldE + (int64_t)c * HID + n0;
    auto emit = [&](int row_u, int rl, int lane_col, const f32x4& lo, const f32x4& hi, int) {
        bf16_t* o = ob + (int64_t)row_u * ldE + ((uint32_t)rl * (uint32_t)ldE + (uint32_t)lane_col);
        f32x4 a = lo, b = hi;
        if (pt.scores) {  // fused A3 term: + w[t,c] * d_pooled[bag(t), c, :]
            const float w = rw_s[row_u + rl];
            const float* __restrict__ dp = dpb + (ro_s[row_u + rl] + lane_col);
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dp), d1 = *reinterpret_cast<const f32x4*>(dp + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = fmaf(w, d0[i], lo[i]);
                b[i] = fmaf(w, d1[i], hi[i]);
            }
        }
        if (accumulate) {
            const bf16x8 old = *reinterpret_cast<const bf16x8*>(o);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] += (float)old[i];
                b[i] += (float)old[4 + i];
            }
        }
        st8_bf16(o, a, b);
    };
    nt256_epilogue(acc, sm, wave, wm, wn, lane, (int)((T - t0 < QM) ? (T - t0) : QM), emit);
}

// ================================================================================================
// dW: slabW[sp][c][k'][n] = sum_{t in split sp} E[t, c, k'] dz[t, c, n]        ("TN": both operands token-major = K-major)
// Round 2: no transposed copies of E and dz any more.  The MFMA fragments (8 consecutive k = tokens per lane) are gathered
// from the K-major LDS image by ds_read_b64_tr_b16.  Measured semantics (tools/micro/tr_probe.hip): within a 16-lane group
// every lane r supplies the address of 4 consecutive bf16 D[r][0..3] and lane l receives D[4j + (l >> 2)][l & 3], j = 0..3;
// with lane r pointing at tile[k0 + (r >> 2)][i0 + 4 (r & 3)] lane l gets tile[k0 + j][i0 + l] -- 4 consecutive tokens of
// "its" row.  LDS stage: A image [32 t][128 k'] (8 KiB) + B image [32 t][256 n] (16 KiB), 64-B unit u of token row t stored
// at unit u ^ (t & 3) (source-side swizzle of the LDS-DMA), two stages, fragments of k-step s+1 requested behind the first
// MFMA of step s (inline-asm reads: the s_waitcnt lgkmcnt is placed by hand).
// ================================================================================================
__global__ __launch_bounds__(256, 2) void gate_dw_bf16_kernel(const bf16_t* __restrict__ E, int64_t ldE, const bf16_t* __restrict__ dz,
                                                              float* __restrict__ slabW, int64_t T, int H, int64_t tok_per_split,
                                                              int n_splits) {
    __shared__ SmemTN sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int kt = xh.li % 4, ntile = (xh.li / 4) % 4, c = xh.c, sp = (xh.li / 16) * xh.nshare + xh.share;
    if (sp >= n_splits) return;  // block-uniform
    const int k0 = kt * 128, n0 = ntile * 256;
    const int64_t ts = (int64_t)sp * tok_per_split;
    int64_t te = ts + tok_per_split;
    if (te > T) te = T;
    const int64_t nch = (te > ts) ? (te - ts + TNK - 1) / TNK : 0;

    const char* baseA = reinterpret_cast<const char*>(E + ts * ldE + (int64_t)c * HID + k0);
    const char* baseB = reinterpret_cast<const char*>(dz + (ts * H + c) * 1024 + n0);
    const uint32_t ldA2 = (uint32_t)ldE * 2u, ldB2 = (uint32_t)H * 1024u * 2u;
    uint32_t cA, cB, kA[2], kB[4];   // 16-B chunk (swizzled) and token row of this lane in each DMA piece
    cA = lane & 15;
    cB = lane & 31;
#pragma unroll
    for (int q = 0; q < 2; ++q) kA[q] = (wave * 2 + q) * 4 + (lane >> 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) kB[q] = (wave * 4 + q) * 2 + (lane >> 5);
    auto dma = [&](int st, int64_t f, int piece) {
        if (piece < 2) {   // E rows past T-1 re-read row T-1: their dz rows are the zero pad
            uint32_t k = kA[piece];
            const int64_t left = T - 1 - (ts + f * TNK);
            if (left < TNK) k = k < (uint32_t)left ? k : (uint32_t)left;   // uniform branch: only the chunk at the end of E
            const uint32_t vo = k * ldA2 + ((cA ^ ((kA[piece] & 3) << 2)) << 4);
            glds16_s(vo, baseA + f * TNK * (int64_t)ldA2, lds_addr_of(&sm.A[st][(wave * 2 + piece) * 512]));
        } else {
            const int q = piece - 2;
            const uint32_t vo = kB[q] * ldB2 + ((cB ^ ((kB[q] & 3) << 2)) << 4);
            glds16_s(vo, baseB + f * TNK * (int64_t)ldB2, lds_addr_of(&sm.B[st][(wave * 4 + q) * 512]));
        }
    };
    f32x16 acc[2][4];
    zero_acc8(acc);
    tn_mainloop(sm, acc, nch, wm, wn, lane, dma);

    // slabW [split][head][k' 512][1024: a-cols 0..511 | b-cols 512..1023]
    float* __restrict__ so = slabW + (((int64_t)sp * H + c) * HID) * 1024;
    const int l32 = lane & 31;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = k0 + wm * 64 + rt * 32 + acc_row(r, lane);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) so[(int64_t)kr * 1024 + n0 + wn * 128 + ct * 32 + l32] = acc[rt][ct][r];
        }
}

// ---- workspace layouts -------------------------------------------------------------------------------
static inline int64_t up16(int64_t b) { return (b + 15) & ~(int64_t)15; }

// dW on the 256 x 256 tile (round 5; tn256_mainloop, 64-token chunks): tile = 256 of the head's 512 E columns x 256 of its 1024 dz columns,
// 8 tiles per head and split.  E rows past T - 1 re-read row T - 1: their dz rows are the zero pad (TQK rows).
template <int NA>   // NA = 3: tn256_mainloop3 on SmemQ3
__global__ __launch_bounds__(512) void gate_dw256_bf16_kernel(const bf16_t* __restrict__ E, int64_t ldE, const bf16_t* __restrict__ dz,
                                                              float* __restrict__ slabW, int64_t T, int H, int64_t tok_per_split,
                                                              int n_splits) {
    __shared__ typename std::conditional<NA == 3, SmemQ3, SmemQ>::type sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int kt = xh.li % 2, ntile = (xh.li / 2) % 4, c = xh.c, sp = (xh.li / 8) * xh.nshare + xh.share;
    if (sp >= n_splits) return;  // block-uniform
    const int k0 = kt * 256, n0 = ntile * 256;
    const int64_t ts = (int64_t)sp * tok_per_split;
    int64_t te = ts + tok_per_split;
    if (te > T) te = T;
    const int64_t nch = (te > ts) ? (te - ts + TQK - 1) / TQK : 0;

    const char* baseA = reinterpret_cast<const char*>(E + ts * ldE + (int64_t)c * HID + k0);
    const char* baseB = reinterpret_cast<const char*>(dz + (ts * H + c) * 1024 + n0);
    const uint32_t ldA2 = (uint32_t)ldE * 2u, ldB2 = (uint32_t)H * 1024u * 2u;
    uint32_t rowq[4], cs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row, src;
        tn256_slot(wave, q, lane, row, src);
        rowq[q] = (uint32_t)row;
        cs[q] = (uint32_t)src << 4;
    }
    auto dma = [&](int st, int64_t f, int piece) {
        const int q = piece & 3;
        if (piece < 4) {
            uint32_t k = rowq[q];
            const int64_t left = T - 1 - (ts + f * TQK);
            if (left < TQK) k = k < (uint32_t)left ? k : (uint32_t)left;   // uniform branch: only the chunk at the end of E
            glds16_s(k * ldA2 + cs[q], uniform_ptr(baseA + f * TQK * (int64_t)ldA2), lds_addr_of(&sm.A[st][(wave * 4 + q) * 1024]));
        } else {
            glds16_s(rowq[q] * ldB2 + cs[q], uniform_ptr(baseB + f * TQK * (int64_t)ldB2), lds_addr_of(&sm.B[st][(wave * 4 + q) * 1024]));
        }
    };
    f32x16 acc[4][2];
    if constexpr (NA == 3) tn256_mainloop3(sm, acc, nch, wm, wn, lane, dma);
    else tn256_mainloop(sm, acc, nch, wm, wn, lane, dma);

    // slabW [split][head][k' 512][1024: a-cols 0..511 | b-cols 512..1023]
    float* __restrict__ so = slabW + (((int64_t)sp * H + c) * HID) * 1024;
    const int l32 = lane & 31;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = k0 + wm * 128 + rt * 32 + acc_row(r, lane);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) so[(int64_t)kr * 1024 + n0 + wn * 64 + ct * 32 + l32] = acc[rt][ct][r];
        }
}
static inline bool gate_dw_use_q(int64_t T) {
    static const bool off = getenv("MADELEINE_BF16_TN256") && atoi(getenv("MADELEINE_BF16_TN256")) == 0;   // A/B switch
    return !off && T >= 16384;
}

struct BwdWs {
    int S;
    int64_t tps, Tpad, nblk;
    int64_t oWN, odz, oslabW, oslabV, total;  // byte offsets
};
static inline BwdWs bwd_ws(int64_t T, int H) {
    BwdWs w;
    const bool q = gate_dw_use_q(T);
    const int chunk = q ? TQK : TNK;
    w.S = q ? splits_for(T, 8 * H, 256)     // gate_dw256_bf16_kernel: one workgroup per CU
            : splits_for(T, 16 * H, 512);   // gate_dw_bf16_kernel: 200 VGPRs = two workgroups per CU
    int64_t tps = (T + w.S - 1) / w.S;
    w.tps = ((tps + chunk - 1) / chunk) * chunk;  // whole chunks of tokens: a chunk never straddles two splits
    if (w.tps < chunk) w.tps = chunk;
    w.Tpad = w.tps * w.S;
    w.nblk = (T + DZ_ROWS - 1) / DZ_ROWS;
    int64_t o = 0;
    w.oWN = o; o += up16((int64_t)H * HID * 1024 * 2);
    w.odz = o; o += up16((T + TQK) * H * 1024 * 2);   // + TQK zero rows: K-tail of the dW contraction
    w.oslabW = o; o += up16((int64_t)w.S * H * HID * 1024 * 4);
    w.oslabV = o; o += up16(w.nblk * H * 4 * HID * 4);
    w.total = o + 64;
    return w;
}

}  // namespace mdl

using namespace mdl;

extern "C" int64_t mdl_abmil_gate_fwd_bf16_ws_bytes(int64_t T, int H) {
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS) return MDL_E_ARG;
    // WK [H][1024][512] bf16 | score partials [T][H][4] fp32
    return (int64_t)H * 1024 * HID * 2 + T * H * GATE_JT * 4 + 64;
}

extern "C" int mdl_abmil_gate_fwd_bf16(const uint16_t* E, int64_t ldE, const float* Wa, const float* ba, const float* Wb,
                                       const float* bb, const float* wc, const float* bc, float* scores, uint16_t* act_a,
                                       uint16_t* act_b, int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a,
                                       const uint8_t* keep_b, void* ws, void* stream) {
    if (!E || !Wa || !ba || !Wb || !bb || !wc || !bc || !scores || !ws) return MDL_E_ARG;
    if ((act_a == nullptr) != (act_b == nullptr)) return MDL_E_ARG;
    if ((keep_a == nullptr) != (keep_b == nullptr)) return MDL_E_ARG;
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS || ldE < (int64_t)H * HID || (ldE & 7)) return MDL_E_ARG;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return MDL_E_ARG;
    if (!host_aligned16(E) || !host_aligned16(Wa) || !host_aligned16(Wb) || !host_aligned16(ws)) return MDL_E_ALIGN;
    if (T == 0) return MDL_OK;
    if (H != 1 && H != 2 && H != 4 && H != 8) return MDL_E_UNSUPPORTED;
    const int64_t n_tt = (T + BBM - 1) / BBM;
    const int64_t grid = xcd_head_grid(n_tt, GATE_JT, H);
    if (grid > 0x7fffffff) return MDL_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const DropCfg d = make_drop(p_drop, seed, keep_a, keep_b);
    bf16_t* WK = (bf16_t*)ws;
    float* part = (float*)((char*)ws + (int64_t)H * 1024 * HID * 2);
    hipLaunchKernelGGL(gate_wk_bf16_kernel, dim3((unsigned)((int64_t)H * 1024 * HID / 4 / 256)), dim3(256), 0, s, Wa, Wb, WK, H);
    MDL_LAUNCH_CHECK();
    const int dm = gate_drop_mode(d);
#define MDL_GATE_FWD16(DM, SAVE)                                                                                                   \
    hipLaunchKernelGGL((gate_fwd_bf16_kernel<DM, SAVE>), dim3((unsigned)grid), dim3(256), 0, s, (const bf16_t*)E, ldE, (const bf16_t*)WK, \
                       ba, bb, wc, part, (bf16_t*)act_a, (bf16_t*)act_b, T, H, (int)n_tt, d)
    const int64_t n_tt256 = (T + QM - 1) / QM;
    // persistent workgroups (round 5, see gate_fwd256_bf16_kernel): MADELEINE_BF16_GATE_PERSIST = 0 | 1 | 2 picks the mode (A/B switch)
    static const int pmode_env = getenv("MADELEINE_BF16_GATE_PERSIST") ? atoi(getenv("MADELEINE_BF16_GATE_PERSIST")) : MDL_GATE_BF16_PMODE;
    const int64_t grid_tiles = xcd_head_grid(n_tt256, GATE_JT, H);
    int pmode = pmode_env;
    if (pmode == 1 && !gate_persist_pays(grid_tiles, 0.93)) pmode = 0;
    if (pmode == 2 && !gate_persist_pays(grid_tiles, 0.93, GATE_PT16)) pmode = 0;
    const int64_t per_share = (n_tt256 + 8 / H - 1) / (8 / H);
    const int64_t grid256 = pmode == 1 ? grid_tiles / GATE_JT : pmode == 2 ? 8 * ((per_share + GATE_PT16 - 1) / GATE_PT16) * GATE_JT : grid_tiles;
#define MDL_GATE_FWD256_1(DM, SAVE, PM)                                                                                            \
    hipLaunchKernelGGL((gate_fwd256_bf16_kernel<DM, SAVE, PM>), dim3((unsigned)grid256), dim3(512), 0, s, (const bf16_t*)E, ldE,   \
                       (const bf16_t*)WK, ba, bb, wc, part, (bf16_t*)act_a, (bf16_t*)act_b, T, H, (int)n_tt256, d)
#define MDL_GATE_FWD256(DM, SAVE)                                                                                                  \
    do {                                                                                                                           \
        if (pmode == 1) MDL_GATE_FWD256_1(DM, SAVE, 1);                                                                            \
        else if (pmode == 2) MDL_GATE_FWD256_1(DM, SAVE, 2);                                                                       \
        else MDL_GATE_FWD256_1(DM, SAVE, 0);                                                                                       \
    } while (0)
    const bool big = T >= 4096 && !getenv("MADELEINE_BF16_GATE128");
    if (big) {   // 256 x 256 x 64 tile
        if (act_a) {
            if (dm == 0) MDL_GATE_FWD256(0, true);
            else if (dm == 1) MDL_GATE_FWD256(1, true);
            else if (dm == 3) MDL_GATE_FWD256(3, true);
            else MDL_GATE_FWD256(2, true);
        } else {
            if (dm == 0) MDL_GATE_FWD256(0, false);
            else if (dm == 1) MDL_GATE_FWD256(1, false);
            else if (dm == 3) MDL_GATE_FWD256(3, false);
            else MDL_GATE_FWD256(2, false);
        }
    } else if (act_a) {
        if (dm == 0) MDL_GATE_FWD16(0, true);
        else if (dm == 1) MDL_GATE_FWD16(1, true);
        else if (dm == 3) MDL_GATE_FWD16(3, true);
        else MDL_GATE_FWD16(2, true);
    } else {
        if (dm == 0) MDL_GATE_FWD16(0, false);
        else if (dm == 1) MDL_GATE_FWD16(1, false);
        else if (dm == 3) MDL_GATE_FWD16(3, false);
        else MDL_GATE_FWD16(2, false);
    }
#undef MDL_GATE_FWD256
#undef MDL_GATE_FWD256_1
#undef MDL_GATE_FWD16
    MDL_LAUNCH_CHECK();
    return gate_launch_finalize(part, bc, scores, T * H, H, s);
}

extern "C" int64_t mdl_abmil_gate_bwd_bf16_ws_bytes(int64_t T, int H) {
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS) return MDL_E_ARG;
    return bwd_ws(T, H).total;
}

static int gate_bwd_bf16_impl(const uint16_t* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                              const uint16_t* act_a, const uint16_t* act_b, const float* d_scores, uint16_t* dE, int accumulate,
                              float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc, int64_t T, int H, float p_drop,
                              uint64_t seed, const uint8_t* keep_a, const uint8_t* keep_b, void* ws, void* stream,
                              const PoolTerm& pt, int phases = 3) {
    if (!E || !Wa || !Wb || !wc || !act_a || !act_b || !d_scores || !dE || !dWa || !dWb || !dba || !dbb || !dwc || !ws)
        return MDL_E_ARG;
    if (phases < 1 || phases > 3) return MDL_E_ARG;   // bit 0: dz pass + its reduction, bit 1: dX / dW contractions (see abmil_gate.hip)
    if ((keep_a == nullptr) != (keep_b == nullptr)) return MDL_E_ARG;
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS || ldE < (int64_t)H * HID || (ldE & 7)) return MDL_E_ARG;
    if (H != 1 && H != 2 && H != 4 && H != 8) return MDL_E_UNSUPPORTED;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return MDL_E_ARG;
    if (!host_aligned16(E) || !host_aligned16(dE) || !host_aligned16(Wa) || !host_aligned16(Wb) || !host_aligned16(act_a) ||
        !host_aligned16(act_b) || !host_aligned16(wc) || !host_aligned16(ws))
        return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const DropCfg d = make_drop(p_drop, seed, keep_a, keep_b);
    const BwdWs L = bwd_ws(T, H);
    char* base = (char*)ws;
    bf16_t* WN = (bf16_t*)(base + L.oWN);
    bf16_t* dz = (bf16_t*)(base + L.odz);
    float* slabW = (float*)(base + L.oslabW);
    float* slabV = (float*)(base + L.oslabV);
    if (L.nblk * H > 0x7fffffff) return MDL_E_UNSUPPORTED;
    if (phases & 1) {
        {   // zero pad rows of dz (K-tail of the dW contraction)
            const hipError_t e = hipMemsetAsync(dz + T * H * 1024, 0, (size_t)TQK * H * 1024 * 2, s);
            if (e != hipSuccess) return (int)e;
        }
        if (T > 0) {
#define MDL_GATE_DZ16(DM)                                                                                                            \
    hipLaunchKernelGGL((gate_dz_kernel<bf16_t, bf16_t, DM>), dim3((unsigned)L.nblk), dim3(64 * H), 0, s, wc, (const bf16_t*)act_a,        \
                       (const bf16_t*)act_b, d_scores, dz, slabV, T, H, d)
            MDL_DISPATCH_DM(gate_drop_mode(d), MDL_GATE_DZ16);
#undef MDL_GATE_DZ16
            MDL_LAUNCH_CHECK();
        }
        const int rc = gate_launch_reduce_v(slabV, dba, dbb, dwc, dbc, H, (int)L.nblk, s);
        if (rc) return rc;
    }
    if (phases & 2) {
        if (T > 0) {
            hipLaunchKernelGGL(gate_wn_bf16_kernel, dim3(16, 32, H), dim3(256), 0, s, Wa, Wb, WN);
            MDL_LAUNCH_CHECK();
            if (T >= 4096) {   // long contraction (K = 1024): the 256 x 256 x 64 tile (measured: 1.58 -> see DESIGN.md)
                const int64_t n_tt = (T + QM - 1) / QM;
                const int64_t grid = xcd_head_grid(n_tt, 2, H);
                if (grid > 0x7fffffff) return MDL_E_UNSUPPORTED;
                if (bf16_stages() == 3)
                    hipLaunchKernelGGL(gate_dx256_bf16_kernel<3>, dim3((unsigned)grid), dim3(512), 0, s, (const bf16_t*)dz, (const bf16_t*)WN,
                                       (bf16_t*)dE, ldE, accumulate, T, H, pt);
                else
                    hipLaunchKernelGGL(gate_dx256_bf16_kernel<2>, dim3((unsigned)grid), dim3(512), 0, s, (const bf16_t*)dz, (const bf16_t*)WN,
                                       (bf16_t*)dE, ldE, accumulate, T, H, pt);
            } else {
                const int64_t n_tt = (T + BBM - 1) / BBM;
                const int64_t grid = xcd_head_grid(n_tt, 2, H);
                if (grid > 0x7fffffff) return MDL_E_UNSUPPORTED;
                hipLaunchKernelGGL(gate_dx_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)WN,
                                   (bf16_t*)dE, ldE, accumulate, T, H, pt);
            }
            MDL_LAUNCH_CHECK();
        }
        if (gate_dw_use_q(T) && bf16_stages() == 3)
            hipLaunchKernelGGL(gate_dw256_bf16_kernel<3>, dim3((unsigned)xcd_head_grid(L.S, 8, H)), dim3(512), 0, s, (const bf16_t*)E, ldE,
                               (const bf16_t*)dz, slabW, T, H, L.tps, L.S);
        else if (gate_dw_use_q(T))
            hipLaunchKernelGGL(gate_dw256_bf16_kernel<2>, dim3((unsigned)xcd_head_grid(L.S, 8, H)), dim3(512), 0, s, (const bf16_t*)E, ldE,
                               (const bf16_t*)dz, slabW, T, H, L.tps, L.S);
        else
        hipLaunchKernelGGL(gate_dw_bf16_kernel, dim3((unsigned)xcd_head_grid(L.S, 16, H)), dim3(256), 0, s, (const bf16_t*)E, ldE,
                           (const bf16_t*)dz, slabW, T, H, L.tps, L.S);
        MDL_LAUNCH_CHECK();
        const int rc = gate_launch_reduce_w(slabW, dWa, dWb, H, L.S, s);
        if (rc) return rc;
    }
    return MDL_OK;
}

extern "C" int mdl_abmil_gate_bwd_bf16(const uint16_t* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                                       const uint16_t* act_a, const uint16_t* act_b, const float* d_scores, uint16_t* dE,
                                       int accumulate, float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc,
                                       int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a,
                                       const uint8_t* keep_b, void* ws, void* stream) {
    return gate_bwd_bf16_impl(E, ldE, Wa, Wb, wc, act_a, act_b, d_scores, dE, accumulate, dWa, dWb, dba, dbb, dwc, dbc, T, H, p_drop,
                              seed, keep_a, keep_b, ws, stream, PoolTerm{nullptr, nullptr, nullptr, nullptr, nullptr, 1});
}

extern "C" int mdl_abmil_attnpool_bwd_bf16(const uint16_t* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                                           const uint16_t* act_a, const uint16_t* act_b, const float* d_scores, uint16_t* dE,
                                           int accumulate, float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc,
                                           int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a, const uint8_t* keep_b,
                                           const float* scores, const float* stat_m, const float* stat_l, const float* d_pooled,
                                           const int32_t* row_bag, int64_t N, void* ws, void* stream) {
    if (!scores || !stat_m || !stat_l || !d_pooled || (!row_bag && N < 1)) return MDL_E_ARG;
    return gate_bwd_bf16_impl(E, ldE, Wa, Wb, wc, act_a, act_b, d_scores, dE, accumulate, dWa, dWb, dba, dbb, dwc, dbc, T, H, p_drop, seed,
                              keep_a, keep_b, ws, stream, PoolTerm{scores, stat_m, stat_l, d_pooled, row_bag, N});
}

extern "C" int mdl_abmil_attnpool_bwd_phases_bf16(const uint16_t* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                                                  const uint16_t* act_a, const uint16_t* act_b, const float* d_scores, uint16_t* dE,
                                                  int accumulate, float* dWa, float* dWb, float* dba, float* dbb, float* dwc,
                                                  float* dbc, int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a,
                                                  const uint8_t* keep_b, const float* scores, const float* stat_m,
                                                  const float* stat_l, const float* d_pooled, const int32_t* row_bag, int64_t N,
                                                  void* ws, void* stream, int phases) {
    if (!scores || !stat_m || !stat_l || !d_pooled || (!row_bag && N < 1)) return MDL_E_ARG;
    return gate_bwd_bf16_impl(E, ldE, Wa, Wb, wc, act_a, act_b, d_scores, dE, accumulate, dWa, dWb, dba, dbb, dwc, dbc, T, H, p_drop, seed,
                              keep_a, keep_b, ws, stream, PoolTerm{scores, stat_m, stat_l, d_pooled, row_bag, N}, phases);
}
