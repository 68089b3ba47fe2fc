// abmil_gate_bf16.hip -- A2 in the bf16 mode (the reference's `precision: bfloat16` autocast runs):
// gated attention scores of all heads with bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate, 2.5 PFLOP/s
// dense peak = 16x the fp32 matrix rate), fp32 epilogues.  Same maths as abmil_gate.hip (reference
// madeleine/models/abmil.py:41-68), same dropout counter hash, same partial-score / finalize scheme.
//
// All three contractions are "NT" products of two K-contiguous row images, so ONE tile engine serves them:
//     C[m, n] = sum_k A[m][k] B[n][k]      tile 128 x 256, BK = 32 bf16 (64 B per tile row), 4 waves (2 x 2) x (2 x 4) MFMA tiles
//   forward : A = E rows (tokens)            B = [Wa;Wb] rows (bf16 copy, K = 512)        fused activation / dropout / wc epilogue
//   dX      : A = dz rows (tokens, K = 1024) B = [Wa;Wb]^T rows (bf16 transposed copy)    dE (bf16) (+)= C
//   dW      : A = E^T rows (channels)        B = dz^T rows (K = tokens, split over K)     fp32 slabs, reduced by gate_reduce_w
// E^T and dz^T are bf16 transposed copies made once per backward (HBM-bound passes, ~1 ms at config 2): the MFMA
// operand registers hold 8 CONSECUTIVE k per lane, which a token-major tensor cannot feed without a transpose.
// Operands reach LDS by LDS-DMA (global_load_lds_dwordx4) into the same XOR-swizzled 64-B row image the fp32 kernels
// use (16-B chunk kq of row r is stored at slot kq ^ ((r >> 2) & 3)), read back with conflict-free ds_read_b128.
#include "gate_common.hpp"

namespace mdl {

constexpr int BBM = 128, BBN = 256, BBK = 32;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct __attribute__((aligned(16))) SmemNT {
    bf16_t A[2][BBM * BBK];  // 8 KiB per stage
    bf16_t B[2][BBN * BBK];  // 16 KiB per stage
};

__device__ __forceinline__ void zero_acc8(f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

// byte offsets (within one stage) of the 16-B fragments this lane reads for k-step g = 0; g = 1 is `^ 32`
__device__ __forceinline__ void nt_offsets(int wm, const int (&colb)[4], int lane, int (&offA)[2], int (&offB)[4]) {
    const int l32 = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        const int r = colb[ct] + l32;
        offB[ct] = r * 64 + ((kh ^ ((r >> 2) & 3)) << 4);
    }
}

// the 16 MFMAs of one staged chunk (two k-steps of 16)
__device__ __forceinline__ void nt_mma_chunk(const SmemNT& sm, int st, f32x16 (&acc)[2][4], const int (&offA)[2],
                                             const int (&offB)[4]) {
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
}

// LDS-DMA slot geometry: instruction q of wave w deposits slots [(nq*w + q)*64, +64); slot s = (row s>>2, stored chunk s&3)
// holds the global chunk (s&3) ^ ((row>>2)&3) of that row.
__device__ __forceinline__ void nt_slot(int instr, int lane, int& row, int& kq) {
    const int sl = instr * 64 + lane;
    row = sl >> 2;
    kq = (sl & 3) ^ ((row >> 2) & 3);
}

template <class Issue>
__device__ __forceinline__ void nt_mainloop(SmemNT& sm, f32x16 (&acc)[2][4], int64_t nch, Issue&& issue, const int (&offA)[2],
                                            const int (&offB)[4]) {
    if (nch > 0) issue(0, (int64_t)0);
    __syncthreads();
    for (int64_t ch = 0; ch < nch; ++ch) {
        const int st = (int)(ch & 1);
        if (ch + 1 < nch) issue(st ^ 1, ch + 1);  // lands in the stage last read before the previous barrier
        nt_mma_chunk(sm, st, acc, offA, offB);
        __syncthreads();  // drains the LDS-DMA of chunk ch+1 (vmcnt) and fences this chunk's reads
    }
}

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

// out[c][r] = in[r][c] (bf16), r < R; columns R <= r < ld_out are zero-filled.  64 x 64 tiles; grid (ld_out/64, C/64).
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const uint16_t* __restrict__ in, int64_t R, int64_t ld_in,
                                                             uint16_t* __restrict__ out, int64_t ld_out) {
    __shared__ uint16_t tile[64][72];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tid = threadIdx.x, row = tid >> 2, seg = (tid & 3) * 16;
    {
        const int64_t r = r0 + row;
        u32x4 v0 = {0u, 0u, 0u, 0u}, v1 = v0;
        if (r < R) {
            const u32x4* src = reinterpret_cast<const u32x4*>(in + r * ld_in + c0 + seg);
            v0 = __builtin_nontemporal_load(src);
            v1 = __builtin_nontemporal_load(src + 1);
        }
        *reinterpret_cast<u32x4*>(&tile[row][seg]) = v0;
        *reinterpret_cast<u32x4*>(&tile[row][seg + 8]) = v1;
    }
    __syncthreads();
    {
        uint32_t w[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) w[i] = (uint32_t)tile[seg + 2 * i][row] | ((uint32_t)tile[seg + 2 * i + 1][row] << 16);
        u32x4* dst = reinterpret_cast<u32x4*>(out + (int64_t)(c0 + row) * ld_out + r0 + seg);
        dst[0] = u32x4{w[0], w[1], w[2], w[3]};
        dst[1] = u32x4{w[4], w[5], w[6], w[7]};
    }
}

// ================================================================================================
// forward
// ================================================================================================
// DM: dropout mode 0 = off, 1 = counter-hash RNG, 2 = explicit uint8 masks; SAVE: store the activations (one code path per
// instantiation keeps the unrolled epilogue inside the register budget).
template <int DM>
__device__ __forceinline__ void fwd_keep2_bf16(const DropCfg& d, int64_t idx, bool& ka, bool& kb) {
    if (DM == 0) {
        ka = kb = true;
    } else if (DM == 2) {
        ka = d.ka[idx] != 0;
        kb = d.kb[idx] != 0;
    } else {
        const uint32_t h = rng_u32(d.key, (uint64_t)idx);
        ka = (h & 0xFFFFu) >= d.thr;
        kb = (h >> 16) >= d.thr;
    }
}

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
            const float bav = ba[c * HID + jc + l32], bbv = bb[c * HID + jc + l32];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                tile[acc_row(r, lane) * 64 + l32] = (float)(bf16_t)fast_tanh(acc[rt][ct][r] + bav);
                tile[acc_row(r, lane) * 64 + 32 + l32] = (float)(bf16_t)fast_sigmoid(acc[rt][2 + ct][r] + bbv);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            const f32x4 wlo = *reinterpret_cast<const f32x4*>(wc + c * HID + jc + g4 * 8);
            const f32x4 whi = *reinterpret_cast<const f32x4*>(wc + c * HID + jc + g4 * 8 + 4);
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
                    if (SAVE) {
                        st8_bf16(act_a + idx, alo, ahi);
                        st8_bf16(act_b + idx, blo, bhi);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        bool keep_a, keep_b;
                        fwd_keep2_bf16<DM>(drop, idx + e, keep_a, keep_b);
                        const float a = e < 4 ? alo[e & 3] : ahi[e & 3], b = e < 4 ? blo[e & 3] : bhi[e & 3];
                        const float w = e < 4 ? wlo[e & 3] : whi[e & 3];
                        const float ad = keep_a ? a * drop.inv : 0.f;
                        const float bd = keep_b ? b * drop.inv : 0.f;
                        sum += ad * bd * w;
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
    f32x16 acc[2][4];
    zero_acc8(acc);
    nt_mainloop(sm, acc, 1024 / BBK, issue, offA, offB);

    // dE (bf16) leaves through the LDS transpose: 8 columns = one 16-B store per lane, 128 contiguous bytes per row
    float* tile = reinterpret_cast<float*>(&sm) + wave * (32 * 64);
    bf16_t* ob = dE + t0 * ldE + (int64_t)c * HID + n0;
    auto emit = [&](int row_u, int rl, int lane_col, const f32x4& lo, const f32x4& hi, int) {
        bf16_t* o = ob + (int64_t)row_u * ldE + ((uint32_t)rl * (uint32_t)ldE + (uint32_t)lane_col);
        f32x4 a = lo, b = hi;
        if (pt.scores) {  // fused A3 term: + w[t,c] * d_pooled[bag(t), c, :]
            int bag;
            const float w = pool_term_weight(pt, t0 + row_u + rl, c, H, bag);
            const float* __restrict__ dp = pt.d_pooled + ((int64_t)bag * H + c) * HID + n0 + lane_col;
            const f32x4 d0 = *reinterpret_cast<const f32x4*>(dp), d1 = *reinterpret_cast<const f32x4*>(dp + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = fmaf(w, d0[i], lo[i]);
                b[i] = fmaf(w, d1[i], hi[i]);
            }
        } else if (accumulate) {
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

// ================================================================================================
// dW: slabW[sp][c][k'][n] = sum_{t in split sp} ET[c*512 + k'][t] dzT[c*1024 + n][t]
// ================================================================================================
__global__ __launch_bounds__(256, 2) void gate_dw_bf16_kernel(const bf16_t* __restrict__ ET, const bf16_t* __restrict__ dzT,
                                                              int64_t ldT, float* __restrict__ slabW, int H,
                                                              int64_t tok_per_split, int n_splits) {
    __shared__ SmemNT sm;
    const int tid = threadIdx.x, lane = tid & 63, wm = (tid >> 6) >> 1, wn = (tid >> 6) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int kt = xh.li % 4, ntile = (xh.li / 4) % 4, c = xh.c, sp = (xh.li / 16) * xh.nshare + xh.share;
    if (sp >= n_splits) return;  // block-uniform
    const int k0 = kt * BBM, n0 = ntile * BBN;
    const int64_t ts = (int64_t)sp * tok_per_split;  // [ts, ts + tok_per_split) lies inside the zero-padded ldT

    const bf16_t* srcA[2];
    const bf16_t* srcB[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int row, kq;
        nt_slot(wave * 2 + q, lane, row, kq);
        srcA[q] = ET + ((int64_t)c * HID + k0 + row) * ldT + ts + kq * 8;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row, kq;
        nt_slot(wave * 4 + q, lane, row, kq);
        srcB[q] = dzT + ((int64_t)c * 1024 + n0 + row) * ldT + ts + kq * 8;
    }
    auto issue = [&](int st, int64_t ch) {
        const int64_t o = ch * BBK;
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(srcA[q] + o, &sm.A[st][(wave * 2 + q) * 512]);
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(srcB[q] + o, &sm.B[st][(wave * 4 + q) * 512]);
    };
    const int colb[4] = {wn * 128, wn * 128 + 32, wn * 128 + 64, wn * 128 + 96};
    int offA[2], offB[4];
    nt_offsets(wm, colb, lane, offA, offB);
    f32x16 acc[2][4];
    zero_acc8(acc);
    nt_mainloop(sm, acc, tok_per_split / BBK, issue, offA, offB);

    // slabW [split][head][k' 512][1024: a-cols 0..511 | b-cols 512..1023]
    float* __restrict__ so = slabW + (((int64_t)sp * H + c) * HID) * 1024;
    const int l32 = lane & 31;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = k0 + wm * 64 + rt * 32 + acc_row(r, lane);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) so[(int64_t)kr * 1024 + n0 + colb[ct] + l32] = acc[rt][ct][r];
        }
}

// ---- workspace layouts -------------------------------------------------------------------------------
static inline int64_t up16(int64_t b) { return (b + 15) & ~(int64_t)15; }

struct BwdWs {
    int S;
    int64_t tps, Tpad, nblk;
    int64_t oWN, odz, odzT, oET, oslabW, oslabV, total;  // byte offsets
};
static inline BwdWs bwd_ws(int64_t T, int H) {
    BwdWs w;
    w.S = gate_splits(T, H);
    int64_t tps = (T + w.S - 1) / w.S;
    w.tps = ((tps + 63) / 64) * 64;  // a multiple of the 64-row transpose tile (and of BBK)
    if (w.tps < 64) w.tps = 64;
    w.Tpad = w.tps * w.S;
    w.nblk = (T + DZ_ROWS - 1) / DZ_ROWS;
    int64_t o = 0;
    w.oWN = o; o += up16((int64_t)H * HID * 1024 * 2);
    w.odz = o; o += up16(T * H * 1024 * 2);
    w.odzT = o; o += up16((int64_t)H * 1024 * w.Tpad * 2);
    w.oET = o; o += up16((int64_t)H * HID * w.Tpad * 2);
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
    const int dm = !d.on ? 0 : (d.ka ? 2 : 1);
#define MDL_GATE_FWD16(DM, SAVE)                                                                                                   \
    hipLaunchKernelGGL((gate_fwd_bf16_kernel<DM, SAVE>), dim3((unsigned)grid), dim3(256), 0, s, (const bf16_t*)E, ldE, (const bf16_t*)WK, \
                       ba, bb, wc, part, (bf16_t*)act_a, (bf16_t*)act_b, T, H, (int)n_tt, d)
    if (act_a) {
        if (dm == 0) MDL_GATE_FWD16(0, true);
        else if (dm == 1) MDL_GATE_FWD16(1, true);
        else MDL_GATE_FWD16(2, true);
    } else {
        if (dm == 0) MDL_GATE_FWD16(0, false);
        else if (dm == 1) MDL_GATE_FWD16(1, false);
        else MDL_GATE_FWD16(2, false);
    }
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
                              const PoolTerm& pt) {
    if (!E || !Wa || !Wb || !wc || !act_a || !act_b || !d_scores || !dE || !dWa || !dWb || !dba || !dbb || !dwc || !ws)
        return MDL_E_ARG;
    if ((keep_a == nullptr) != (keep_b == nullptr)) return MDL_E_ARG;
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS || ldE < (int64_t)H * HID || (ldE & 7)) return MDL_E_ARG;
    if (H != 1 && H != 2 && H != 4 && H != 8) return MDL_E_UNSUPPORTED;
    if (ldE != (int64_t)H * HID) return MDL_E_UNSUPPORTED;  // the E^T pass transposes the whole [T, H*512] matrix
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
    bf16_t* dzT = (bf16_t*)(base + L.odzT);
    bf16_t* ET = (bf16_t*)(base + L.oET);
    float* slabW = (float*)(base + L.oslabW);
    float* slabV = (float*)(base + L.oslabV);
    if (L.nblk > 0x7fffffff || L.Tpad / 64 > 0x7fffffff) return MDL_E_UNSUPPORTED;
    if (T > 0) {
        hipLaunchKernelGGL(gate_wn_bf16_kernel, dim3(16, 32, H), dim3(256), 0, s, Wa, Wb, WN);
        MDL_LAUNCH_CHECK();
        hipLaunchKernelGGL((gate_dz_kernel<bf16_t, bf16_t>), dim3((unsigned)L.nblk, H), dim3(256), 0, s, wc, (const bf16_t*)act_a,
                           (const bf16_t*)act_b, d_scores, dz, slabV, T, H, d);
        MDL_LAUNCH_CHECK();
        const int64_t n_tt = (T + BBM - 1) / BBM;
        const int64_t grid = xcd_head_grid(n_tt, 2, H);
        if (grid > 0x7fffffff) return MDL_E_UNSUPPORTED;
        hipLaunchKernelGGL(gate_dx_bf16_kernel, dim3((unsigned)grid), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)WN,
                           (bf16_t*)dE, ldE, accumulate, T, H, pt);
        MDL_LAUNCH_CHECK();
    }
    // transposed copies for the token contraction (zero-filled up to Tpad), then dW over S splits of the tokens
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)(L.Tpad / 64), (unsigned)(H * 1024 / 64)), dim3(256), 0, s,
                       (const uint16_t*)dz, T, (int64_t)H * 1024, (uint16_t*)dzT, L.Tpad);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((unsigned)(L.Tpad / 64), (unsigned)(H * HID / 64)), dim3(256), 0, s, E, T, ldE,
                       (uint16_t*)ET, L.Tpad);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(gate_dw_bf16_kernel, dim3((unsigned)xcd_head_grid(L.S, 16, H)), dim3(256), 0, s, (const bf16_t*)ET,
                       (const bf16_t*)dzT, L.Tpad, slabW, H, L.tps, L.S);
    MDL_LAUNCH_CHECK();
    int rc = gate_launch_reduce_w(slabW, dWa, dWb, H, L.S, s);
    if (rc) return rc;
    return gate_launch_reduce_v(slabV, dba, dbb, dwc, dbc, H, (int)L.nblk, s);
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
                                           float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc, int64_t T, int H,
                                           float p_drop, uint64_t seed, const uint8_t* keep_a, const uint8_t* keep_b,
                                           const float* scores, const float* stat_m, const float* stat_l, const float* d_pooled,
                                           const int32_t* row_bag, int64_t N, void* ws, void* stream) {
    if (!scores || !stat_m || !stat_l || !d_pooled || (!row_bag && N < 1)) return MDL_E_ARG;
    return gate_bwd_bf16_impl(E, ldE, Wa, Wb, wc, act_a, act_b, d_scores, dE, 0, dWa, dWb, dba, dbb, dwc, dbc, T, H, p_drop, seed,
                              keep_a, keep_b, ws, stream, PoolTerm{scores, stat_m, stat_l, d_pooled, row_bag, N});
}
