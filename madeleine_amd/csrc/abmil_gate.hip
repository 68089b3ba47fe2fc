// abmil_gate.hip -- A2: gated attention scores of all heads (forward, backward).
//
// Replaces BatchedABMIL.forward (reference madeleine/models/abmil.py:41-68) called once per head from
// ABMILEmbedder.forward (reference madeleine/models/Model.py:406-409):
//     a = Dropout(.25)(tanh(x Wa^T + ba));  b = Dropout(.25)(sigmoid(x Wb^T + bb));  s = (a*b) wc^T + bc
// Three fp32 contractions per head dominate the path (4.19 MFLOP/token fwd); they run on
// v_mfma_f32_32x32x2_f32 (exact fp32 = an fmaf chain, 157 TFLOP/s peak).
//
//   forward : S_part[t, jt] = sum_{j in tile jt} a'_j b'_j wc_j      GEMM [T,512] x [512, 128a|128b], fused epilogue
//   dz      : d(za)|d(zb) [T,H,1024] + column sums dba, dbb, dwc, dbc HBM-bound pass (once per step)
//   dX      : dE[t,c,:]  (+)= dz[t,c,:] . [Wa;Wb]                     GEMM [T,1024] x [1024,512]
//   dW      : [dWa;dWb]^T = X^T dz                                    GEMM [512,T] x [T,128a|128b], split over T
//
// One block tile shape for the three GEMMs: 128 x 256 outputs, BK = 16, 256 threads = 4 waves (2 x 2), each wave
// 64 x 128 = 2 x 4 MFMA 32x32 accumulators (128 AGPRs).  Every operand reaches LDS by LDS-DMA
// (global_load_lds_dwordx4: no staging VGPRs, no ds_write pass), two LDS stages of 24 KiB; the main loop is the
// software-pipelined one of tile_engine.hpp (fragment requests one step ahead, the chunk barrier before the last step, the
// next DMA between that step's MFMAs), 3 workgroups per CU.
#include "tile_engine.hpp"

namespace mdl {

constexpr int GBM = 128, GBN = 256, GBK = 16;

// ================================================================================================
// forward
// ================================================================================================
// Direct-to-LDS staging (global_load_lds_dwordx4).  A = E rows (K-contiguous): a wave instruction deposits 64 x 16 B
// = 16 rows x 64 B (BK = 16) straight into a row-major LDS image [row][16 k] -- no staging VGPRs, no ds_write pass
// (ablation: register staging + transposed ds_write_b32 scatter cost 11 % of the kernel).  The image must be
// lane-linear, so bank conflicts are removed on the SOURCE side: the 16-B chunk a lane fetches is XOR-swizzled,
// slot = kq ^ ((row >> 2) & 3), and the ds_read_b128 fragment reads apply the same involution; the MFMA's K = 2 is
// fed with k-pairs (8g+e | 8g+4+e) from the two half-waves (a permutation of the k order, applied to A and B alike).
// B = the gate weights, transposed once per call into a K-major image WT [H][512 k][a j 0..511 | b j 0..511]
// (gate_wt_kernel, 8 MiB): a chunk is then 16 rows x (128 a | 128 b) columns, one 1-KiB row per wave instruction,
// read back with conflict-free ds_read_b32 -- which keeps the kernel at ~160 VGPRs = 3 waves/SIMD.
// WT[c][k][j] = Wa[c][j][k], WT[c][k][512 + j] = Wb[c][j][k]   (32 x 32 LDS transpose)
__global__ __launch_bounds__(256) void gate_wt_kernel(const float* __restrict__ Wa, const float* __restrict__ Wb,
                                                      float* __restrict__ WT) {
    __shared__ float tile[32][33];
    const int c = blockIdx.z, jb = blockIdx.y * 32, kb = blockIdx.x * 32;  // jb over 1024 (a | b), kb over 512
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ W = (jb < HID) ? Wa + ((int64_t)c * HID + jb) * HID : Wb + ((int64_t)c * HID + jb - HID) * HID;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + i * 8][tx] = W[(int64_t)(ty + i * 8) * HID + kb + tx];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) WT[((int64_t)c * HID + kb + ty + i * 8) * 1024 + jb + tx] = tile[tx][ty + i * 8];
}

// DM: dropout mode 0 = off, 1 = counter-hash RNG, 2 = explicit uint8 masks (one code path per instantiation: the
// three-way runtime choice per element cost 12 arch VGPRs, i.e. the third wave per SIMD).  SAVE: store the activations.
template <int DM, bool SAVE>
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ E, int64_t ldE,
                                                       const float* __restrict__ WT, const float* __restrict__ ba,
                                                       const float* __restrict__ bb, const float* __restrict__ wc,
                                                       float* __restrict__ part, float* __restrict__ act_a,
                                                       float* __restrict__ act_b, int64_t T, int H, int n_ttiles,
                                                       DropCfg drop) {
    __shared__ __attribute__((aligned(16))) TileSmem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform (LDS-DMA bases, SGPR addressing)
    const int wm = wave >> 1, wn = wave & 1;
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int jt = xh.li % GATE_JT, c = xh.c, tt = (xh.li / GATE_JT) * xh.nshare + xh.share;
    if (tt >= n_ttiles) return;  // block-uniform
    const int64_t t0 = (int64_t)tt * GBM;
    const int j0 = jt * 128;

    // A: E rows of head c (swizzled row image).  B: row k of WT, lanes 0-31 -> a columns j0.., lanes 32-63 -> b columns 512 + j0..
    const char* baseA = reinterpret_cast<const char*>(E + t0 * ldE + (int64_t)c * HID);
    const char* baseB = reinterpret_cast<const char*>(WT + ((int64_t)c * HID + wave * 4) * 1024 + j0);
    uint32_t voA[2];
    rows_voff(voA, T - t0, ldE, wave, lane);
    const uint32_t voB = ((lane & 31) * 4 + (lane >> 5) * HID) * 4;
    constexpr int64_t rowB = 1024 * 4;
    const int l32 = lane & 31;
    const int colb[4] = {wn * 64, wn * 64 + 32, 128 + wn * 64, 128 + wn * 64 + 32};  // a, a, b, b

    f32x16 acc[2][4];
    tile_zero(acc);
    tile_loop_nn(acc, sm, HID / GBK, wm, colb, lane, [&](int st, int f, int piece) {
        if (piece == 0) rows_issue(baseA + (int64_t)f * (GBK * 4), voA, sm.A[st], wave);
        else krows_issue2(baseB + ((int64_t)f * GBK + (piece - 1) * 2) * rowB, rowB, voB, sm.B[st], wave, (piece - 1) * 2);
    });

    // ---- epilogue: 4 passes (rt, ct) of a 32-row x (32 a | 32 b)-column block through a wave-private LDS tile.
    // (1) accumulator layout: a = tanh(za + ba), b = sigmoid(zb + bb) -> tile (accumulators are read one at a time:
    //     the arch-VGPR budget of 3 waves per SIMD is 40);  (2) transposed layout, lane = (row = lane>>3, 4 columns):
    //     dropout, wc-weighted partial sum reduced over the row's 8 lanes, and 16-B stores of the activations.
    float* tile = reinterpret_cast<float*>(&sm) + wave * (32 * 64);
    float* sred = reinterpret_cast<float*>(&sm) + 4 * (32 * 64) + wn * GBM + wm * 64;   // [2 (wn)][128 rows]
    const int g8 = lane & 7, r8 = lane >> 3;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int jc = j0 + wn * 64 + ct * 32;                 // first gate column of this pass
            // (bias and the exponent's scaling folded into one FMA: gate_tanh_pre / gate_sigmoid_pre, common.hpp)
            const float ta = 2.f * MDL_LOG2E * ba[c * HID + jc + l32], tb = -MDL_LOG2E * bb[c * HID + jc + l32];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float za, zb;
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(za) : "a"(acc[rt][ct][r]));
                asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(zb) : "a"(acc[rt][2 + ct][r]));
                tile[acc_row(r, lane) * 64 + l32] = gate_tanh_pre(za, 2.f * MDL_LOG2E, ta);
                tile[acc_row(r, lane) * 64 + 32 + l32] = gate_sigmoid_pre(zb, -MDL_LOG2E, tb);
                if ((r & 3) == 3) TILE_SB();
            }
            const f32x4 wc4 = *reinterpret_cast<const f32x4*>(wc + c * HID + jc + g8 * 4) * (drop.inv * drop.inv);   // both dropout factors
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + r8;
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&tile[row * 64 + g8 * 4]);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(&tile[row * 64 + 32 + g8 * 4]);
                float sum = 0.f;
                if (t0 + wm * 64 + rt * 32 + row < T) {
                    // element index = uniform 64-bit base of the pass + 32-bit lane part
                    const int64_t idx = ((t0 + wm * 64 + rt * 32) * H + c) * HID + jc + (uint32_t)(row * H * HID + g8 * 4);
                    const uint32_t rkey = drop_row_key(drop, idx);   // idx % 4 == 0: the 4 elements share the high word
                    if (SAVE) {
                        *reinterpret_cast<f32x4*>(act_a + idx) = a4;
                        *reinterpret_cast<f32x4*>(act_b + idx) = b4;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bool keep_a, keep_b;
                        gate_keep2_fwd<DM>(drop, idx, e, rkey, keep_a, keep_b);
                        const float ab = a4[e] * b4[e];
                        sum = fmaf((keep_a && keep_b) ? ab : 0.f, wc4[e], sum);
                    }
                }
                sum += __shfl_xor(sum, 1, 64);
                sum += __shfl_xor(sum, 2, 64);
                sum += __shfl_xor(sum, 4, 64);
                if (g8 == 0) {
                    if (ct == 0) sred[rt * 32 + row] = sum;
                    else sred[rt * 32 + row] += sum;
                }
                TILE_SB();   // one row group at a time: keeps the epilogue inside the 40 arch VGPRs of 3 waves per SIMD
            }
        }
    __syncthreads();
    if (tid < GBM) {
        const int64_t t = t0 + tid;
        const float* sr = reinterpret_cast<const float*>(&sm) + 4 * (32 * 64);
        if (t < T) part[(t * H + c) * GATE_JT + jt] = sr[tid] + sr[GBM + tid];
    }
}

__global__ void gate_finalize_kernel(const float* __restrict__ part, const float* __restrict__ bc,
                                     float* __restrict__ scores, int64_t n, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // over T*H
    if (i >= n) return;
    const f32x4 p = *reinterpret_cast<const f32x4*>(part + i * GATE_JT);
    scores[i] = bc[i % H] + (((p.x + p.y) + p.z) + p.w);
}

// ================================================================================================
// backward, stage 1: d(za) | d(zb) once per step (+ every column sum), so that dX and dW are pure GEMMs
//   dza = ds * wc_j * b' * (keep_a/(1-p)) * (1 - a^2)        b' = keep_b ? b/(1-p) : 0
//   dzb = ds * wc_j * a' * (keep_b/(1-p)) * b (1 - b)        a' = keep_a ? a/(1-p) : 0
// (ablation: computing this transform inside the staging of both GEMMs, x2 and x4 redundantly, cost 4.2 ms of
//  the 21.9 ms backward at config 2; as one HBM-bound pass it is 8.6 GB ~ 1.8 ms and the column sums ride along.)
// dz layout: [T + GBK][H][1024] = (dza[512] | dzb[512]); the GBK pad rows are zero (K-tail of the dW GEMM).
// ================================================================================================
// ================================================================================================
// backward, stage 2: dE[t, c, :] (+)= dz[t, c, :] . [Wa_c ; Wb_c]     GEMM [T, 1024] x [1024, 512], both operands by LDS-DMA
// A = dz rows (K-contiguous): swizzled row image like the forward; B = 16 weight rows x 256 columns per chunk (K-major
// in global memory already): one 1-KiB row per wave instruction.  k order within a chunk: (8g+e | 8g+4+e) per half-wave.
// ================================================================================================
__global__ __launch_bounds__(256) void gate_bwd_dx_kernel(const float* __restrict__ dz, const float* __restrict__ Wa,
                                                          const float* __restrict__ Wb, float* __restrict__ dE,
                                                          int64_t ldE, int accumulate, int64_t T, int H, PoolTerm pt) {
    __shared__ __attribute__((aligned(16))) TileSmem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int nt = xh.li % 2, c = xh.c, tt = (xh.li / 2) * xh.nshare + xh.share;
    const int64_t t0 = (int64_t)tt * GBM;
    if (t0 >= T) return;  // block-uniform
    const int n0 = nt * GBN;

    const char* baseA = reinterpret_cast<const char*>(dz + (t0 * H + c) * 1024);
    uint32_t voA[2];
    rows_voff(voA, T - t0, (int64_t)H * 1024, wave, lane);
    // B: weight rows j (K-major in memory already): chunks 0..31 = Wa rows, 32..63 = Wb rows (a chunk never straddles)
    const char* baseWa = reinterpret_cast<const char*>(Wa + ((int64_t)c * HID + wave * 4) * HID + n0);
    const char* baseWb = reinterpret_cast<const char*>(Wb + ((int64_t)c * HID + wave * 4) * HID + n0);
    const uint32_t voB = lane * 16;
    constexpr int64_t rowB = HID * 4;
    const int colb[4] = {wn * 128, wn * 128 + 32, wn * 128 + 64, wn * 128 + 96};
    // fused A3 term: softmax weight and d_pooled row of every tile row, once per row (see sp_gate_dx_kernel)
    float row_w = 0.f;
    int row_off = 0;
    if (pt.scores && tid < GBM && t0 + tid < T) {
        int bag;
        row_w = pool_term_weight(pt, t0 + tid, c, H, bag);
        row_off = (bag * H + c) * HID;
    }
    f32x16 acc[2][4];
    tile_zero(acc);
    tile_loop_nn(acc, sm, 2 * HID / GBK, wm, colb, lane, [&](int st, int f, int piece) {
        if (piece == 0) {
            rows_issue(baseA + (int64_t)f * (GBK * 4), voA, sm.A[st], wave);
        } else {
            const char* w = (f < HID / GBK) ? baseWa + (int64_t)f * GBK * rowB : baseWb + (int64_t)(f - HID / GBK) * GBK * rowB;
            krows_issue2(w + (piece - 1) * 2 * rowB, rowB, voB, sm.B[st], wave, (piece - 1) * 2);
        }
    });

    char* ob = reinterpret_cast<char*>(dE + t0 * ldE + (int64_t)c * HID + n0);
    const uint32_t ld4 = (uint32_t)ldE * 4u;
    float* rw_s = reinterpret_cast<float*>(&sm) + 4 * (32 * 64);   // behind the four waves' transpose areas
    int* ro_s = reinterpret_cast<int*>(rw_s + GBM);
    if (pt.scores) {
        __syncthreads();   // every wave has left the main loop: the staging memory is free
        if (tid < GBM) {
            rw_s[tid] = row_w;
            ro_s[tid] = row_off;
        }
        __syncthreads();
    }
    const float* dpb = pt.d_pooled + n0;
    auto emit = [&](int row_u, int rl, int lane_col, const f32x4& v, int) {
        f32x4* o = reinterpret_cast<f32x4*>(ob + (int64_t)row_u * ld4 + ((uint32_t)rl * ld4 + (uint32_t)lane_col * 4u));
        f32x4 r = v;
        if (pt.scores) {  // fused A3 term: + w[t,c] * d_pooled[bag(t), c, :]  (cache-resident row, no dE re-read)
            const float w = rw_s[row_u + rl];
            const f32x4 dp = *reinterpret_cast<const f32x4*>(dpb + (ro_s[row_u + rl] + lane_col));
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = fmaf(w, dp[i], v[i]);
        }
        if (accumulate) r += *o;   // e.g. the token_projector's dX already sits in dE (fused A2 + A3 + token-projector backward)
        *o = r;
    };
    if (t0 + GBM <= T) tile_epilogue_rows<true, 2>(acc, sm, wave, wm, colb, lane, GBM, emit);
    else tile_epilogue_rows<false, 2>(acc, sm, wave, wm, colb, lane, (int)(T - t0), emit);
}

// ================================================================================================
// backward, stage 3: dW^T tile  C[k', n] = sum_t X[t, k'] dz[t, n]   (n < 128: dza column j0+n ; else dzb column j0+n-128)
// over the token range of this split; both operands are K-major in global memory -> LDS-DMA, natural images.
// ================================================================================================
__global__ __launch_bounds__(256) void gate_bwd_dw_kernel(const float* __restrict__ E, int64_t ldE,
                                                          const float* __restrict__ dz, float* __restrict__ slabW,
                                                          int64_t T, int H, int64_t tok_per_split, int n_splits) {
    __shared__ __attribute__((aligned(16))) TileSmem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int kt = xh.li % 4, jt = (xh.li / 4) % GATE_JT, c = xh.c, sp = (xh.li / (4 * GATE_JT)) * xh.nshare + xh.share;
    if (sp >= n_splits) return;  // block-uniform
    const int k0 = kt * 128, j0 = jt * 128;
    const int64_t ts = (int64_t)sp * tok_per_split;
    int64_t te = ts + tok_per_split;
    if (te > T) te = T;

    // A: two 512-B rows of X per wave instruction (rows past T re-read row T-1: their dz rows are the zero pad)
    // B: one row per wave instruction: lanes 0-31 -> dza segment, lanes 32-63 -> dzb segment (rows < T + GBK: zero pad)
    const char* Xb = reinterpret_cast<const char*>(E + (int64_t)c * HID + k0);
    const char* Zb = reinterpret_cast<const char*>(dz + (int64_t)c * 1024 + j0);
    const uint32_t ldE4 = (uint32_t)ldE * 4u;
    const uint32_t colA = (lane & 31) * 16;
    const uint32_t voB = ((lane & 31) * 4 + (lane >> 5) * HID) * 4;
    const int64_t rowZ = (int64_t)H * 1024 * 4;
    float (*As)[TBK][TBM] = reinterpret_cast<float (*)[TBK][TBM]>(&sm.A[0][0]);

    f32x16 acc[2][4];
    tile_zero(acc);
    const int colb[4] = {wn * 128, wn * 128 + 32, wn * 128 + 64, wn * 128 + 96};
    const int64_t nch = (te > ts) ? (te - ts + GBK - 1) / GBK : 0;
    tile_loop_tn(acc, sm, nch, wm, colb, lane, [&](int st, int64_t f, int piece) {
        const int64_t tb = ts + f * GBK;
        if (piece == 0) {
            const int64_t left = T - 1 - tb;   // >= 0
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r0 = (wave * 2 + q) * 2;
                uint32_t r = r0 + (lane >> 5);
                if (left < GBK) r = r < (uint32_t)left ? r : (uint32_t)left;   // uniform branch: only the chunk at the end of E
                glds16_s(r * ldE4 + colA, Xb + tb * (int64_t)ldE4, lds_addr_of(&As[st][r0][0]));
            }
        } else {
            const char* z = Zb + (tb + wave * 4 + (piece - 1) * 2) * rowZ;
            krows_issue2(z, rowZ, voB, sm.B[st], wave, (piece - 1) * 2);
        }
    });

    // slabW [split][head][k' 512][1024: a-cols 0..511 | b-cols 512..1023]
    float* __restrict__ so = slabW + (((int64_t)sp * H + c) * HID) * 1024;
    const int l32 = lane & 31;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = k0 + wm * 64 + rt * 32 + acc_row(r, lane);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int n = colb[ct] + l32;  // [0,256)
                const int col = (n < 128) ? (j0 + n) : (512 + j0 + n - 128);
                so[(int64_t)kr * 1024 + col] = acc[rt][ct][r];
            }
        }
}

// dWa[c][j][k] = sum_s slabW[s][c][k][j] ; dWb[c][j][k] = sum_s slabW[s][c][k][512+j]  (32x32 LDS transpose)
__global__ __launch_bounds__(256) void gate_reduce_w_kernel(const float* __restrict__ slabW, float* __restrict__ dWa,
                                                            float* __restrict__ dWb, int H, int S) {
    __shared__ float tile[32][33];
    const int c = blockIdx.z, kb = blockIdx.y * 32, nb = blockIdx.x * 32;  // nb over 1024 columns
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                // 32 x 8
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = kb + ty + i * 8;
        const float* __restrict__ src = slabW + ((int64_t)c * HID + k) * 1024 + nb + tx;
        const int64_t st = (int64_t)H * HID * 1024;
        float v = 0.f;
        int sp = 0;
        for (; sp + 8 <= S; sp += 8) {   // 8 slab reads in flight per thread, additions in slab order
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)(sp + u) * st];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; sp < S; ++sp) v += src[(int64_t)sp * st];
        tile[ty + i * 8][tx] = v;
    }
    __syncthreads();
    float* __restrict__ dst = (nb < 512) ? dWa : dWb;
    const int jb = (nb < 512) ? nb : nb - 512;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = jb + ty + i * 8;
        dst[((int64_t)c * HID + j) * HID + kb + tx] = tile[tx][ty + i * 8];
    }
}

// dba | dbb | dwc [H,512] and dbc [H] = sum over row blocks of slabV [S][H][4][512]: 32 columns x 8 block-groups per
// workgroup, 8 loads in flight per thread, groups merged through LDS in a fixed order.
__global__ __launch_bounds__(256) void gate_reduce_v_kernel(const float* __restrict__ slabV, float* __restrict__ dba,
                                                            float* __restrict__ dbb, float* __restrict__ dwc,
                                                            float* __restrict__ dbc, int H, int S) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;  // over H*4*512
    const int j = i % HID, which = (i / HID) % 4, c = i / (4 * HID);
    const bool live = i < H * 4 * HID && !(which == 3 && j != 0);
    float v = 0.f;
    if (live) {
        const float* __restrict__ src = slabV + ((int64_t)c * 4 + which) * HID + j;
        const int64_t stride = (int64_t)H * 4 * HID;
        int s = grp;
        for (; s + 56 < S; s += 64) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)(s + 8 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; s < S; s += 8) v += src[(int64_t)s * stride];
    }
    red[grp][cl] = v;
    __syncthreads();
    if (grp == 0 && live) {
        float t = red[0][cl];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += red[g][cl];
        if (which == 3) {
            if (dbc) dbc[c] = t;
        } else {
            (which == 0 ? dba : (which == 1 ? dbb : dwc))[c * HID + j] = t;
        }
    }
}

__global__ void gate_mask_kernel(uint8_t* __restrict__ keep, int64_t n, int which, DropCfg d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        bool ka, kb;
        drop_keep2(d, i, drop_row_key(d, i), ka, kb);     // the very function the forward / backward kernels evaluate (either field width)
        keep[i] = (which ? kb : ka) ? 1 : 0;
    }
}

static inline int64_t gate_tok_per_split(int64_t T, int S) {
    int64_t tps = (T + S - 1) / S;
    return ((tps + GBK - 1) / GBK) * GBK;
}

int gate_launch_finalize(const float* part, const float* bc, float* scores, int64_t n, int H, hipStream_t s) {
    hipLaunchKernelGGL(gate_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, bc, scores, n, H);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}
int gate_launch_reduce_w(const float* slabW, float* dWa, float* dWb, int H, int S, hipStream_t s) {
    hipLaunchKernelGGL(gate_reduce_w_kernel, dim3(32, 16, H), dim3(256), 0, s, slabW, dWa, dWb, H, S);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}
int gate_launch_reduce_v(const float* slabV, float* dba, float* dbb, float* dwc, float* dbc, int H, int S, hipStream_t s) {
    hipLaunchKernelGGL(gate_reduce_v_kernel, dim3((H * 4 * HID + 31) / 32), dim3(256), 0, s, slabV, dba, dbb, dwc, dbc, H, S);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

}  // namespace mdl

using namespace mdl;

extern "C" int64_t mdl_abmil_gate_fwd_ws_bytes(int64_t T, int H) {
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS) return MDL_E_ARG;
    // WT [H][512][1024] (transposed weights) | score partials [T][H][4]
    return ((int64_t)H * HID * 1024 + T * H * GATE_JT) * 4 + 64;
}

extern "C" int mdl_abmil_gate_fwd(const float* E, int64_t ldE, const float* Wa, const float* ba, const float* Wb,
                                  const float* bb, const float* wc, const float* bc, float* scores, float* act_a,
                                  float* act_b, int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a,
                                  const uint8_t* keep_b, void* ws, void* stream) {
    if (!E || !Wa || !ba || !Wb || !bb || !wc || !bc || !scores || !ws) return MDL_E_ARG;
    if ((act_a == nullptr) != (act_b == nullptr)) return MDL_E_ARG;
    if ((keep_a == nullptr) != (keep_b == nullptr)) return MDL_E_ARG;
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS || ldE < (int64_t)H * HID || (ldE & 3)) return MDL_E_ARG;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return MDL_E_ARG;
    if (!host_aligned16(E) || !host_aligned16(Wa) || !host_aligned16(Wb) || !host_aligned16(ws)) return MDL_E_ALIGN;
    if (T == 0) return MDL_OK;
    const int64_t n_tt = (T + GBM - 1) / GBM;
    if (H != 1 && H != 2 && H != 4 && H != 8) return MDL_E_UNSUPPORTED;
    const int64_t grid = xcd_head_grid(n_tt, GATE_JT, H);
    if (grid > 0x7fffffff) return MDL_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const DropCfg d = make_drop(p_drop, seed, keep_a, keep_b);
    float* WT = (float*)ws;
    float* part = WT + (int64_t)H * HID * 1024;
    hipLaunchKernelGGL(gate_wt_kernel, dim3(16, 32, H), dim3(256), 0, s, Wa, Wb, WT);
    MDL_LAUNCH_CHECK();
    const int dm = gate_drop_mode(d);
#define MDL_GATE_FWD(DM, SAVE)                                                                                                  \
    hipLaunchKernelGGL((gate_fwd_kernel<DM, SAVE>), dim3((unsigned)grid), dim3(256), 0, s, E, ldE, (const float*)WT, ba, bb, wc, part, \
                       act_a, act_b, T, H, (int)n_tt, d)
    if (act_a) {
        if (dm == 0) MDL_GATE_FWD(0, true);
        else if (dm == 1) MDL_GATE_FWD(1, true);
        else if (dm == 3) MDL_GATE_FWD(3, true);
        else MDL_GATE_FWD(2, true);
    } else {
        if (dm == 0) MDL_GATE_FWD(0, false);
        else if (dm == 1) MDL_GATE_FWD(1, false);
        else if (dm == 3) MDL_GATE_FWD(3, false);
        else MDL_GATE_FWD(2, false);
    }
#undef MDL_GATE_FWD
    MDL_LAUNCH_CHECK();
    const int64_t n = T * H;
    hipLaunchKernelGGL(gate_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)part, bc, scores,
                       n, H);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

static inline int64_t dz_blocks(int64_t T) { return (T + DZ_ROWS - 1) / DZ_ROWS; }

extern "C" int64_t mdl_abmil_gate_bwd_ws_bytes(int64_t T, int H) {
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS) return MDL_E_ARG;
    const int S = gate_splits(T, H);
    // dz [(T + GBK)][H][1024] | slabW [S][H][512][1024] | slabV [dz_blocks][H][4][512]
    return ((T + GBK) * H * 1024 + (int64_t)S * H * HID * 1024 + dz_blocks(T) * H * 4 * HID) * 4 + 64;
}

static int gate_bwd_impl(const float* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc, const float* act_a,
                         const float* act_b, const float* d_scores, float* dE, int accumulate, float* dWa, float* dWb, float* dba,
                         float* dbb, float* dwc, float* dbc, int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a,
                         const uint8_t* keep_b, void* ws, void* stream, const PoolTerm& pt, int phases = 3) {
    // phases: bit 0 = dz pass (HBM-bound) + its column-sum reduction, bit 1 = the dX / dW contractions (MFMA-bound) + slab reduction.
    // The product path passes 3; the *_phases entry points let a profiler time the two halves separately on the same workspace.
    if (!E || !Wa || !Wb || !wc || !act_a || !act_b || !d_scores || !dE || !dWa || !dWb || !dba || !dbb || !dwc || !ws)
        return MDL_E_ARG;
    if (phases < 1 || phases > 3) return MDL_E_ARG;
    if ((keep_a == nullptr) != (keep_b == nullptr)) return MDL_E_ARG;
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS || ldE < (int64_t)H * HID || (ldE & 3)) return MDL_E_ARG;
    if (H != 1 && H != 2 && H != 4 && H != 8) return MDL_E_UNSUPPORTED;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return MDL_E_ARG;
    if (!host_aligned16(E) || !host_aligned16(Wa) || !host_aligned16(Wb) || !host_aligned16(act_a) ||
        !host_aligned16(act_b) || !host_aligned16(wc) || !host_aligned16(ws))
        return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const DropCfg d = make_drop(p_drop, seed, keep_a, keep_b);
    const int S = gate_splits(T, H);
    const int64_t tps = gate_tok_per_split(T, S);
    float* dz = (float*)ws;
    float* slabW = dz + (T + GBK) * H * 1024;
    float* slabV = slabW + (int64_t)S * H * HID * 1024;
    const int64_t nblk = dz_blocks(T);
    if (phases & 1) {
        {   // zero pad rows of dz (K-tail of the dW GEMM)
            const hipError_t e = hipMemsetAsync(dz + T * H * 1024, 0, (size_t)GBK * H * 1024 * sizeof(float), s);
            if (e != hipSuccess) return (int)e;
        }
        if (T > 0) {
            if (nblk * H > 0x7fffffff) return MDL_E_UNSUPPORTED;
#define MDL_GATE_DZ(DM)                                                                                                              \
    hipLaunchKernelGGL((gate_dz_kernel<float, float, DM>), dim3((unsigned)nblk), dim3(64 * H), 0, s, wc, act_a, act_b, d_scores, dz, slabV, \
                       T, H, d)
            MDL_DISPATCH_DM(gate_drop_mode(d), MDL_GATE_DZ);
#undef MDL_GATE_DZ
            MDL_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(gate_reduce_v_kernel, dim3((H * 4 * HID + 31) / 32), dim3(256), 0, s, (const float*)slabV, dba, dbb,
                           dwc, dbc, H, (int)nblk);
        MDL_LAUNCH_CHECK();
    }
    if (phases & 2) {
        if (T > 0) {
            const int64_t n_tt = (T + GBM - 1) / GBM;
            const int64_t grid = xcd_head_grid(n_tt, 2, H);
            if (grid > 0x7fffffff) return MDL_E_UNSUPPORTED;
            hipLaunchKernelGGL(gate_bwd_dx_kernel, dim3((unsigned)grid), dim3(256), 0, s, (const float*)dz, Wa, Wb, dE, ldE,
                               accumulate, T, H, pt);
            MDL_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL(gate_bwd_dw_kernel, dim3((unsigned)xcd_head_grid(S, 4 * GATE_JT, H)), dim3(256), 0, s, E, ldE,
                           (const float*)dz, slabW, T, H, tps, S);
        MDL_LAUNCH_CHECK();
        hipLaunchKernelGGL(gate_reduce_w_kernel, dim3(32, 16, H), dim3(256), 0, s, (const float*)slabW, dWa, dWb, H, S);
        MDL_LAUNCH_CHECK();
    }
    return MDL_OK;
}

extern "C" int mdl_abmil_gate_bwd(const float* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                                  const float* act_a, const float* act_b, const float* d_scores, float* dE,
                                  int accumulate, float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc,
                                  int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a,
                                  const uint8_t* keep_b, void* ws, void* stream) {
    return gate_bwd_impl(E, ldE, Wa, Wb, wc, act_a, act_b, d_scores, dE, accumulate, dWa, dWb, dba, dbb, dwc, dbc, T, H, p_drop,
                         seed, keep_a, keep_b, ws, stream, PoolTerm{nullptr, nullptr, nullptr, nullptr, nullptr, 1});
}

extern "C" int mdl_abmil_attnpool_bwd(const float* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                                      const float* act_a, const float* act_b, const float* d_scores, float* dE, int accumulate,
                                      float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc, int64_t T, int H,
                                      float p_drop, uint64_t seed, const uint8_t* keep_a, const uint8_t* keep_b,
                                      const float* scores, const float* stat_m, const float* stat_l, const float* d_pooled,
                                      const int32_t* row_bag, int64_t N, void* ws, void* stream) {
    if (!scores || !stat_m || !stat_l || !d_pooled || (!row_bag && N < 1)) return MDL_E_ARG;
    return gate_bwd_impl(E, ldE, Wa, Wb, wc, act_a, act_b, d_scores, dE, accumulate, dWa, dWb, dba, dbb, dwc, dbc, T, H, p_drop, seed, keep_a,
                         keep_b, ws, stream, PoolTerm{scores, stat_m, stat_l, d_pooled, row_bag, N});
}

extern "C" int mdl_abmil_attnpool_bwd_phases(const float* E, int64_t ldE, const float* Wa, const float* Wb, const float* wc,
                                             const float* act_a, const float* act_b, const float* d_scores, float* dE,
                                             int accumulate, float* dWa, float* dWb, float* dba, float* dbb, float* dwc, float* dbc,
                                             int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a,
                                             const uint8_t* keep_b, const float* scores, const float* stat_m, const float* stat_l,
                                             const float* d_pooled, const int32_t* row_bag, int64_t N, void* ws, void* stream,
                                             int phases) {
    if (!scores || !stat_m || !stat_l || !d_pooled || (!row_bag && N < 1)) return MDL_E_ARG;
    return gate_bwd_impl(E, ldE, Wa, Wb, wc, act_a, act_b, d_scores, dE, accumulate, dWa, dWb, dba, dbb, dwc, dbc, T, H, p_drop, seed, keep_a,
                         keep_b, ws, stream, PoolTerm{scores, stat_m, stat_l, d_pooled, row_bag, N}, phases);
}

extern "C" int mdl_abmil_gate_dropout_mask(uint8_t* keep, int64_t T, int H, int which, float p_drop, uint64_t seed,
                                           void* stream) {
    if (!keep || T < 0 || H < 1 || H > MDL_MAX_HEADS || (which != 0 && which != 1)) return MDL_E_ARG;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return MDL_E_ARG;
    const int64_t n = T * H * HID;
    if (n == 0) return MDL_OK;
    DropCfg d = make_drop(p_drop, seed, nullptr, nullptr);
    d.on = 1;   // (p = 0: threshold 0, every element kept)
    hipLaunchKernelGGL(gate_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keep, n, which, d);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}
