// abmil_gate_split.hip -- A2 (gated attention scores, reference madeleine/models/abmil.py:41-68 called per head from
// madeleine/models/Model.py:406-409) forward and backward with the three contractions on the split-fp16 engine (split_engine.hpp):
// fp32 values, fp32-level accuracy, 3 v_mfma_f32_32x32x16_f16 per 16 k instead of 8 v_mfma_f32_32x32x2_f32.  Same maths, same
// dropout counter hash, same partial-score / finalize / slab-reduction scheme as abmil_gate.hip (whose kernels remain the 'fp32'
// GEMM mode); the epilogues are fp32.
//   forward : A = image(E) rows of head c (K = 512)      B = image([Wa;Wb]) rows (a j | b j)     fused tanh / sigmoid / dropout / wc
//   dz      : d(za)|d(zb) written as an IMAGE directly by the HBM-bound pass (scale from the rigorous bound max|ds| max|wc| / (1-p)^2)
//   dX      : A = image(dz) rows (t, c) (K = 1024)       B = image([Wa;Wb]^T) rows e              dE (+)= . + pooling term
//   dW      : TN over tokens: A = image(E) head columns, B = image(dz) head columns              slabs -> gate_reduce_w
#include "split_engine.hpp"

namespace mdl {

// ---- weight images --------------------------------------------------------------------------------------------------------------------
// sc[1] = max(|Wa|, |Wb|) (sp_absmax on both), sc[0] = its scale.
// WK [H][1024 rows: a j 0..511 | b j 0..511][512 k]: the forward's B operand
__global__ __launch_bounds__(256) void sp_gate_wk_kernel(const float* __restrict__ Wa, const float* __restrict__ Wb, char* __restrict__ WK,
                                                         int H, const float* __restrict__ sc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over H * 1024 * 64 groups of 8 k
    if (i >= (int64_t)H * 1024 * 64) return;
    const int k = (int)(i % 64) * 8, r = (int)((i / 64) % 1024), c = (int)(i / (64 * 1024));
    const float* src = (r < HID) ? Wa + ((int64_t)c * HID + r) * HID + k : Wb + ((int64_t)c * HID + r - HID) * HID + k;
    const float s = sc[0];
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
    const float v[8] = {x0.x * s, x0.y * s, x0.z * s, x0.w * s, x1.x * s, x1.y * s, x1.z * s, x1.w * s};
    u32x4 hi, lo;
    sp_split8(v, hi, lo);
    char* row = WK + ((int64_t)c * 1024 + r) * (HID * 4);
    *reinterpret_cast<u32x4*>(row + sp_img_off(k, 0)) = hi;
    *reinterpret_cast<u32x4*>(row + sp_img_off(k, 1)) = lo;
}
// WN [H][512 rows e][1024 columns: a j | b j] = [Wa;Wb]^T: the dX product's B operand   (32 x 32 LDS transpose)
__global__ __launch_bounds__(256) void sp_gate_wn_kernel(const float* __restrict__ Wa, const float* __restrict__ Wb, char* __restrict__ WN,
                                                         const float* __restrict__ sc) {
    __shared__ float tile[32][33];
    const int c = blockIdx.z, jb = blockIdx.y * 32, eb = blockIdx.x * 32;  // jb over 1024 (a | b), eb over 512
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* __restrict__ W = (jb < HID) ? Wa + ((int64_t)c * HID + jb) * HID : Wb + ((int64_t)c * HID + jb - HID) * HID;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + i * 8][tx] = W[(int64_t)(ty + i * 8) * HID + eb + tx];   // tile[j][e]
    __syncthreads();
    if (threadIdx.x < 128) {
        const int e = threadIdx.x >> 2, g = threadIdx.x & 3;   // row e, 8 consecutive j
        const float s = sc[0];
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = tile[g * 8 + u][e] * s;
        u32x4 hi, lo;
        sp_split8(v, hi, lo);
        char* row = WN + ((int64_t)c * HID + eb + e) * (1024 * 4);
        *reinterpret_cast<u32x4*>(row + sp_img_off(jb + g * 8, 0)) = hi;
        *reinterpret_cast<u32x4*>(row + sp_img_off(jb + g * 8, 1)) = lo;
    }
}
// out[0] = scale for a bound in[0] * in[1] * mult (dz: max|ds| * max|wc| / (1-p)^2); out[1] = the bound
__global__ void sp_bound_scale_kernel(const float* __restrict__ in, float mult, float* __restrict__ out) {
    const float b = in[0] * in[1] * mult;
    out[1] = b;
    out[0] = sp_scale_for(b);
}

// ================================================================================================
// forward
// ================================================================================================
// Tile: 256 tokens x (128 a | 128 b) gate columns j0 .. j0 + 127 of head c.  Tile column n = wn * SP_WCOLS + ct * 32 + l with the first
// SPNCT / 2 column tiles of a wave = a columns j0 + wn * (16 SPNCT) + ct * 32 + l and the second half = the b columns of the same j, so that
// a wave holds za and zb of the same (token, j) in acc[rt][cp] / acc[rt][SPNCT / 2 + cp].
// Round 5: persistent workgroups.  With one workgroup per CU (128 KiB of stages) nothing overlaps a tile's prologue -- workgroup launch,
// the first block's memory latency -- with the previous tile.  A persistent workgroup runs PER tiles back to back and requests the NEXT
// tile's first block into stage 0 before the epilogue of the current one; the epilogue stages through stage 1 (sp_stage1_tile) and the
// row sums have their own LDS.  Same arithmetic in the same order: bit-identical to one workgroup per tile.
//   PMODE 0: one tile per workgroup (round 4)
//   PMODE 1: the GATE_JT column tiles of one (token tile, head): the E tile is re-read by the same CU GATE_JT times in a row
//   PMODE 2: GATE_PT consecutive token tiles of one (column tile, head): the weight tile stays, and the GATE_JT workgroups of the same
//            token tiles still run side by side on one XCD (they share each E tile through its L2, as in PMODE 0)
#ifndef MDL_GATE_SP_PMODE
#define MDL_GATE_SP_PMODE 0
#endif
constexpr int GATE_PT = 4;
// NA = 3 (round 6; one tile per workgroup only): sp_nt_mainloop3 on SmemSP3 -- the LDS-DMA pieces spread over the chunk; the epilogue's
// transposes then use A[0] / A[1] and the row sums B[0] of the drained ring (160 KiB leave no room for a separate array).
template <int DM, bool SAVE, int PMODE, int NA = 2>
__global__ __launch_bounds__(SP_THREADS) void sp_gate_fwd_kernel(const char* __restrict__ Ei, int64_t e_rsb, const float* __restrict__ e_sc,
                                                          const char* __restrict__ WK, const float* __restrict__ w_sc,
                                                          const float* __restrict__ ba, const float* __restrict__ bb,
                                                          const float* __restrict__ wc, float* __restrict__ part,
                                                          float* __restrict__ act_a, float* __restrict__ act_b, int64_t T, int H,
                                                          int n_ttiles, DropCfg drop) {
    static_assert(NA == 2 || PMODE == 0, "the three-stage ring serves one tile per workgroup");
    __shared__ SmemSPn<NA> sm;
    __shared__ float sred_own[NA == 2 ? SP_WN * SPM : 1];   // [SP_WN][256 rows]
    float* sred_s = NA == 2 ? sred_own : reinterpret_cast<float*>(&sm.B[0][0]);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / SP_WN, wn = wave % SP_WN;
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int c = xh.c;
    // the workgroup's tiles: (token tile tt_of(k), column tile jt_of(k)), k = 0 .. n_k - 1
    constexpr int n_k = PMODE == 1 ? GATE_JT : PMODE == 2 ? GATE_PT : 1;
    auto tt_of = [&](int k) { return (PMODE == 1 ? xh.li : PMODE == 2 ? (xh.li / GATE_JT) * GATE_PT + k : xh.li / GATE_JT) * xh.nshare + xh.share; };
    auto jt_of = [&](int k) { return PMODE == 1 ? k : xh.li % GATE_JT; };
    if (tt_of(0) >= n_ttiles) return;  // block-uniform

    constexpr int HALF = SPNCT / 2;   // a (= b) column tiles per wave
    uint32_t voB[SP_PW];
#pragma unroll
    for (int i = 0; i < SP_PW; ++i) {
        int row, ch;
        sp_nt_slot(wave, i, lane, row, ch);
        // tile row (= tile column of the product) -> row of the head's [a | b] weight block, relative to the tile's first gate column j0
        const int wv = row / SP_WCOLS, ct = (row >> 5) % SPNCT;
        const int wrow = (ct / HALF) * HID + wv * (32 * HALF) + (ct % HALF) * 32 + (row & 31);
        voB[i] = (uint32_t)(wrow * (HID * 4) + ch * 16);
    }
    auto a_offsets = [&](int64_t t0, uint32_t (&vo)[SP_PW]) {   // rows past T re-read the last valid row (discarded)
#pragma unroll
        for (int i = 0; i < SP_PW; ++i) {
            int row, ch;
            sp_nt_slot(wave, i, lane, row, ch);
            int64_t ra = row;
            if (t0 + ra > T - 1) ra = T - 1 - t0;
            vo[i] = (uint32_t)(ra * e_rsb + ch * 16);
        }
    };
    const float inv = 1.f / (e_sc[0] * w_sc[0]);
    const int l32 = lane & 31;
    float* tile = NA == 2 ? sp_stage1_tile(sm, wave) : reinterpret_cast<float*>(&sm.A[wave / (SP_WAVES / 2)][(wave % (SP_WAVES / 2)) * 8192]);
    float* sred = sred_s + wn * SPM + wm * 128;
    const int g8 = lane & 7, r8 = lane >> 3;

    for (int k = 0; k < n_k; ++k) {
    const int tt = tt_of(k), jt = jt_of(k);
    if (tt >= n_ttiles) break;   // block-uniform (PMODE 2: a short last group)
    const int64_t t0 = (int64_t)tt * SPM;
    const int j0 = jt * 128;
    const char* baseA = Ei + t0 * e_rsb + (int64_t)c * (HID * 4);
    const char* baseB = WK + ((int64_t)c * 1024 + j0) * (HID * 4);
    uint32_t voA[SP_PW];
    a_offsets(t0, voA);
    SpAcc acc;
    sp_zero(acc);
    auto dma = [&](int st, int f, int piece) {
        const int i = piece % SP_PW;
        if (piece < SP_PW) glds16_s(voA[i], sp_uniform(baseA + (int64_t)f * 128), lds_addr_of(&sm.A[st][(wave * SP_PW + i) * 1024]));
        else glds16_s(voB[i], sp_uniform(baseB + (int64_t)f * 128), lds_addr_of(&sm.B[st][(wave * SP_PW + i) * 1024]));
    };
    if constexpr (NA == 3) sp_nt_mainloop3(sm, acc, HID / 32, wm, wn, lane, dma);
    else sp_nt_mainloop(sm, acc, HID / 32, wm, wn, lane, dma, k > 0);
    if (k + 1 < n_k && tt_of(k + 1) < n_ttiles) {   // the next tile's first block travels during this epilogue
        const int64_t t0n = (int64_t)tt_of(k + 1) * SPM;
        const char* baseAn = Ei + t0n * e_rsb + (int64_t)c * (HID * 4);
        const char* baseBn = WK + ((int64_t)c * 1024 + jt_of(k + 1) * 128) * (HID * 4);
        uint32_t voAn[SP_PW];
        a_offsets(t0n, voAn);
#pragma unroll
        for (int piece = 0; piece < SP_NP; ++piece) {
            const int i = piece % SP_PW;
            if (piece < SP_PW) glds16_s(voAn[i], sp_uniform(baseAn), lds_addr_of(&sm.A[0][(wave * SP_PW + i) * 1024]));
            else glds16_s(voB[i], sp_uniform(baseBn), lds_addr_of(&sm.B[0][(wave * SP_PW + i) * 1024]));
        }
    }

    // ---- epilogue: 4 x HALF passes (rt, cp) of a 32-row x (32 a | 32 b)-column block through the wave's LDS tile (as abmil_gate.hip)
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int cp = 0; cp < HALF; ++cp) {
            const int jc = j0 + wn * (32 * HALF) + cp * 32;   // first gate column of this pass
            // pre-activation = acc * inv + bias, folded into the exponent's FMA (gate_tanh_pre / gate_sigmoid_pre, common.hpp)
            const float ta = 2.f * MDL_LOG2E * ba[c * HID + jc + l32], tb = -MDL_LOG2E * bb[c * HID + jc + l32];
            const float sa2 = 2.f * MDL_LOG2E * inv, sb1 = -MDL_LOG2E * inv;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                tile[acc_row(r, lane) * 64 + l32] = gate_tanh_pre(acc[rt][cp][r], sa2, ta);
                tile[acc_row(r, lane) * 64 + 32 + l32] = gate_sigmoid_pre(acc[rt][HALF + cp][r], sb1, tb);
                if ((r & 3) == 3) SP_SB();
            }
            // score term of an element: keep_a keep_b / (1-p)^2 a b wc -- the two dropout factors folded into wc, ONE select per pair
            const f32x4 wc4 = *reinterpret_cast<const f32x4*>(wc + c * HID + jc + g8 * 4) * (drop.inv * drop.inv);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 8 + r8;
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&tile[row * 64 + g8 * 4]);
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(&tile[row * 64 + 32 + g8 * 4]);
                float sum = 0.f;
                if (t0 + wm * 128 + rt * 32 + row < T) {
                    const int64_t idx = ((t0 + wm * 128 + rt * 32) * H + c) * HID + jc + (uint32_t)(row * H * HID + g8 * 4);
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
                    if (cp == 0) sred[rt * 32 + row] = sum;
                    else sred[rt * 32 + row] += sum;
                }
                SP_SB();
            }
        }
    __syncthreads();
    for (int rr = tid; rr < SPM; rr += SP_THREADS) {
        const int64_t t = t0 + rr;
        float v = sred_s[rr];
#pragma unroll
        for (int w = 1; w < SP_WN; ++w) v += sred_s[w * SPM + rr];
        if (t < T) part[(t * H + c) * GATE_JT + jt] = v;
    }
    }   // k (the next tile's main loop has barriers between these reads of sred_s and its epilogue's writes)
}

// ================================================================================================
// backward, stage 1: d(za) | d(zb) as a split image [T + 32][H][1024] (+ the column sums), scale dz_sc[0] from the bound
// ================================================================================================
// Workgroup = DZ_ROWS token rows x ALL heads (64 H threads: thread = (head, 8 columns)): every row is one contiguous 2 KB H read of
// each activation and one contiguous 4 KB H write of the image (a workgroup per head touched 2 KB of every 8 KB).
template <int DM>   // dropout mode, as the forward kernels (gate_keep2_fwd)
__global__ __launch_bounds__(64 * MDL_MAX_HEADS) void sp_gate_dz_kernel(const float* __restrict__ wc, const float* __restrict__ act_a,
                                                         const float* __restrict__ act_b, const float* __restrict__ d_scores,
                                                         char* __restrict__ dzi, const float* __restrict__ dz_sc,
                                                         float* __restrict__ slabV, int64_t T, int H, DropCfg drop, int rows_per_wg) {
    // Round 5: lane q of head c owns columns 4q .. 4q + 3 and 256 + 4q .. 256 + 4q + 3 of its head (fully coalesced float4 loads: one
    // instruction of the wave = 1 KiB), and the image goes out through sp_img_store4's pair exchange (whole 128-B lines per store
    // instruction).  Round 4 had 8 consecutive columns per lane: 32-B loads and 16 B + 16 B stores, the half-line pattern that streams
    // 4.8 TB/s where this order streams 5.2-5.4 (tools/micro/hbm_rate.hip patterns).  Same values, same summation order per column.
    constexpr int NH = 2, VEC = 4;   // column halves per lane x columns per float4
    const int tid = threadIdx.x, q = tid & 63, c = tid >> 6;
    const int64_t bx = blockIdx.x;
    const int64_t r0 = bx * rows_per_wg;
    int64_t r1 = r0 + rows_per_wg;
    if (r1 > T) r1 = T;
    const float s = dz_sc[0];
    f32x4 vw[NH], sa[NH], sb[NH], sw[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        vw[h] = *reinterpret_cast<const f32x4*>(wc + c * HID + h * 256 + q * VEC);
        sa[h] = sb[h] = sw[h] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    float sds = 0.f;
    constexpr int UNR = 2;   // 2 rows x (a, b) x 2 x 16 B = 8 loads in flight per thread
    for (int64_t rb = r0; rb < r1; rb += UNR) {
        f32x4 va[UNR][NH], vb[UNR][NH];
        float ds[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t r = rb + u;
            const bool ok = r < r1;
            const int64_t o = ((ok ? r : rb) * H + c) * HID + q * VEC;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                va[u][h] = ld4_nt(act_a + o + h * 256);
                vb[u][h] = ld4_nt(act_b + o + h * 256);
            }
            ds[u] = ok ? d_scores[r * H + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t r = rb + u;
            if (r < r1) {   // block-uniform
                const int64_t o = (r * H + c) * HID + q * VEC;
                const uint32_t rkey = drop_row_key(drop, o);   // the head's 512 elements share the high word
                char* row = dzi + (r * H + c) * (int64_t)(1024 * 4);
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    f32x4 za, zb;
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        float w, a, b;
                        gate_dz_t<DM>(drop, ds[u], vw[h][i], va[u][h][i], vb[u][h][i], o + h * 256, i, rkey, a, b, w);
                        sw[h][i] += w;
                        sa[h][i] += a;
                        sb[h][i] += b;
                        za[i] = a;
                        zb[i] = b;
                    }
                    sp_img_store4(row, h * 256 + q * VEC, za, s);
                    sp_img_store4(row, HID + h * 256 + q * VEC, zb, s);
                }
                sds += ds[u];
            }
        }
    }
    float* __restrict__ o = slabV + (bx * H + c) * 4 * HID + q * VEC;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        *reinterpret_cast<f32x4*>(o + h * 256) = sa[h];
        *reinterpret_cast<f32x4*>(o + HID + h * 256) = sb[h];
        *reinterpret_cast<f32x4*>(o + 2 * HID + h * 256) = sw[h];
    }
    if (q == 0) o[3 * HID] = sds;
}

// ================================================================================================
// backward, stage 2: dE[t, c, n0 + n] (+)= sum_j dz[t, c, j] WN[c][n0 + n][j]  (+ pooling term), K = 1024
// ================================================================================================
template <int TERMS, int NA = 2>   // NA = 3: sp_nt_mainloop3 on SmemSP3 (the epilogue works in the first 128 KiB of the drained ring)
__global__ __launch_bounds__(SP_THREADS) void sp_gate_dx_kernel(const char* __restrict__ dzi, const float* __restrict__ dz_sc,
                                                         const char* __restrict__ WN, const float* __restrict__ w_sc,
                                                         float* __restrict__ dE, int64_t ldE, int accumulate, int64_t T, int H,
                                                         PoolTerm pt, float* __restrict__ absmax_out) {
    __shared__ SmemSPn<NA> sm3;
    SmemSP& sm = reinterpret_cast<SmemSP&>(sm3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / SP_WN, wn = wave % SP_WN;
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int nt = xh.li % 2, c = xh.c, tt = (xh.li / 2) * xh.nshare + xh.share;
    const int64_t t0 = (int64_t)tt * SPM;
    if (t0 >= T) return;  // block-uniform
    const int n0 = nt * SPN;

    const uint32_t rowA = (uint32_t)H * 4096u;
    const char* baseA = dzi + (t0 * H + c) * (int64_t)4096;
    const char* baseB = WN + ((int64_t)c * HID + n0) * 4096;
    uint32_t voA[SP_PW], voB[SP_PW];
#pragma unroll
    for (int i = 0; i < SP_PW; ++i) {
        int row, ch;
        sp_nt_slot(wave, i, lane, row, ch);
        int64_t ra = row;
        if (t0 + ra > T - 1) ra = T - 1 - t0;
        voA[i] = (uint32_t)ra * rowA + ch * 16;
        voB[i] = (uint32_t)row * 4096u + ch * 16;
    }
    // fused A3 term: the softmax weight and the d_pooled row of every tile row, ONCE per row (thread r < 256 owns row r; the loads
    // fly behind the main loop) -- per output vector this was a 64-bit division, three dependent loads and an exp: 8,400 epilogue
    // instructions per wave against the main loop's 5,000
    float row_w = 0.f;
    int row_off = 0;
    if (pt.scores && tid < SPM && t0 + tid < T) {
        int bag;
        row_w = pool_term_weight(pt, t0 + tid, c, H, bag);
        row_off = (bag * H + c) * HID;
    }
    SpAcc acc;
    sp_zero(acc);
    auto dma = [&](int st, int f, int piece) {
        const int i = piece % SP_PW;
        if (piece < SP_PW) glds16_s(voA[i], sp_uniform(baseA + (int64_t)f * 128), lds_addr_of(&sm3.A[st][(wave * SP_PW + i) * 1024]));
        else glds16_s(voB[i], sp_uniform(baseB + (int64_t)f * 128), lds_addr_of(&sm3.B[st][(wave * SP_PW + i) * 1024]));
    };
    if constexpr (NA == 3) sp_nt_mainloop3<TERMS>(sm3, acc, 1024 / 32, wm, wn, lane, dma);
    else sp_nt_mainloop<TERMS>(sm3, acc, 1024 / 32, wm, wn, lane, dma);
    const float inv = 1.f / (dz_sc[0] * w_sc[0]);
    float amax = 0.f;
    char* ob = reinterpret_cast<char*>(dE + t0 * ldE + (int64_t)c * HID + n0);
    const uint32_t ld4 = (uint32_t)ldE * 4u;
    // row factors -> LDS behind the epilogue's transpose area (the first 64 KiB of the staging memory, free after the main loop)
    float* rw_s = reinterpret_cast<float*>(&sm.B[0][0]);
    int* ro_s = reinterpret_cast<int*>(&sm.B[0][SPM * 4]);
    if (pt.scores) {
        if (tid < SPM) {
            rw_s[tid] = row_w;
            ro_s[tid] = row_off;
        }
        __syncthreads();
    }
    const float* dpb = pt.d_pooled + n0;
    auto emit = [&](int row, int col, const f32x4& v) {
        f32x4* o = reinterpret_cast<f32x4*>(ob + (int64_t)row * ld4 + (uint32_t)col * 4u);
        f32x4 r = v * inv;
        if (pt.scores) {  // + w[t,c] * d_pooled[bag(t), c, :]  (cache-resident row)
            const float w = rw_s[row];
            const f32x4 dp = *reinterpret_cast<const f32x4*>(dpb + (ro_s[row] + col));
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = fmaf(w, dp[i], r[i]);
        }
        if (accumulate) r += *o;
        *o = r;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
    };
    if (t0 + SPM <= T) sp_epilogue_rows<true, 2>(acc, sm, wave, wm, wn, lane, SPM, emit);
    else sp_epilogue_rows<false, 2>(acc, sm, wave, wm, wn, lane, (int)(T - t0), emit);
    if (absmax_out) sp_block_absmax(absmax_out, amax, reinterpret_cast<float*>(&sm.B[1][0]));   // B[0] holds the row factors
}

// ================================================================================================
// backward, stage 3: slabW[sp][c][k' 512][1024: a | b] = sum_{t in split} E[t, c, k'] dz[t, c, n]      (TN over tokens)
// ================================================================================================
template <int TERMS, int NA = 2>   // NA = 3: sp_tn_mainloop3 on SmemSP3
__global__ __launch_bounds__(SP_THREADS) void sp_gate_dw_kernel(const char* __restrict__ Ei, int64_t e_rsb, const float* __restrict__ e_sc,
                                                         const char* __restrict__ dzi, const float* __restrict__ dz_sc,
                                                         float* __restrict__ slabW, int64_t T, int H, int64_t tok_per_split,
                                                         int n_splits) {
    __shared__ SmemSPn<NA> sm3;
    SmemSP& sm = reinterpret_cast<SmemSP&>(sm3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / SP_WN, wn = wave % SP_WN;
    const XcdHead xh = xcd_head(blockIdx.x, H);
    const int kt = xh.li % 2, ntile = (xh.li / 2) % 4, c = xh.c, sp = (xh.li / 8) * xh.nshare + xh.share;
    if (sp >= n_splits) return;  // block-uniform
    const int i0 = kt * SPM, n0 = ntile * SPN;
    const int64_t ts = (int64_t)sp * tok_per_split;
    int64_t te = ts + tok_per_split;
    if (te > T) te = T;
    const int64_t nch = (te > ts) ? (te - ts + SPK - 1) / SPK : 0;

    uint32_t tokq[SP_PW], coA[SP_PW], coB[SP_PW];
#pragma unroll
    for (int q = 0; q < SP_PW; ++q) {
        const int kr = (wave * SP_PW + q) * 2 + (lane >> 5), p = kr >> 5, src = (lane & 31) ^ ((kr & 3) << 2);
        tokq[q] = kr & 31;
        coA[q] = (uint32_t)sp_img_off(i0 + src * 8, p);
        coB[q] = (uint32_t)sp_img_off(n0 + src * 8, p);
    }
    const uint32_t rowB = (uint32_t)H * 4096u;
    const char* baseA = Ei + ts * e_rsb + (int64_t)c * (HID * 4);
    const char* baseB = dzi + (ts * H + c) * (int64_t)4096;
    SpAcc acc;
    sp_zero(acc);
    auto dma = [&](int st, int64_t f, int piece) {
        const int q = piece % SP_PW;
        if (piece < SP_PW) {   // E rows past T - 1 re-read row T - 1: their dz rows are the zero pad
            uint32_t tk = tokq[q];
            const int64_t left = T - 1 - (ts + f * SPK);
            if (left < SPK) tk = tk < (uint32_t)left ? tk : (uint32_t)left;
            glds16_s(tk * (uint32_t)e_rsb + coA[q], sp_uniform(baseA + f * SPK * e_rsb), lds_addr_of(&sm3.A[st][(wave * SP_PW + q) * 1024]));
        } else {
            glds16_s(tokq[q] * rowB + coB[q], sp_uniform(baseB + f * SPK * (int64_t)rowB), lds_addr_of(&sm3.B[st][(wave * SP_PW + q) * 1024]));
        }
    };
    if constexpr (NA == 3) sp_tn_mainloop3<TERMS>(sm3, acc, nch, wm, wn, lane, dma);
    else sp_tn_mainloop<TERMS>(sm3, acc, nch, wm, wn, lane, dma);
    const float inv = 1.f / (e_sc[0] * dz_sc[0]);
    float* so = slabW + (((int64_t)sp * H + c) * HID + i0) * 1024 + n0;
    auto emit = [&](int row, int col, const f32x4& v) { *reinterpret_cast<f32x4*>(so + (int64_t)row * 1024 + col) = v * inv; };
    sp_epilogue_rows<true>(acc, sm, wave, wm, wn, lane, SPM, emit);
}

// split_gemm.hip
int sp_launch_absmax(const float* X, int64_t ldx, int64_t rows, int K, float* out, hipStream_t s);
int sp_launch_absmax_flat(const float* x, int64_t n, float* out, hipStream_t s);
int sp_launch_scale(float* sc, hipStream_t s);

static inline int64_t up16s(int64_t b) { return (b + 15) & ~(int64_t)15; }
// token rows per workgroup of the dz pass: a workgroup walks its rows two at a time (~1.3 us per pair of rows, latency), so 256 rows
// only pay when there are thousands of workgroups to overlap them; a 2,048-token step (config 1) spent 340 us in 8 workgroups
#ifndef MDL_DZ_ROWS_BIG
#define MDL_DZ_ROWS_BIG DZ_ROWS
#endif
static inline int sp_dz_rows(int64_t T) { return T >= 131072 ? MDL_DZ_ROWS_BIG : (T >= 16384 ? 64 : 16); }
struct SpBwdWs {
    int S;
    int64_t tps, nblk;
    int64_t oWN, odz, oslabW, oslabV, osc, total;
};
static inline SpBwdWs sp_bwd_ws(int64_t T, int H) {
    SpBwdWs w;
    w.S = splits_for(T, 8 * H, 256);   // 2 x 4 tiles per head and split; one workgroup per CU
    int64_t tps = (T + w.S - 1) / w.S;
    w.tps = ((tps + SPK - 1) / SPK) * SPK;
    if (w.tps < SPK) w.tps = SPK;
    w.nblk = (T + sp_dz_rows(T) - 1) / sp_dz_rows(T);
    int64_t o = 0;
    w.oWN = o; o += up16s((int64_t)H * HID * 1024 * 4);
    w.odz = o; o += up16s((T + SPK) * H * 1024 * 4);       // + 32 zero rows: token tail of the dW contraction
    w.oslabW = o; o += up16s((int64_t)w.S * H * HID * 1024 * 4);
    w.oslabV = o; o += up16s(w.nblk * H * 4 * HID * 4);
    w.osc = o; o += 64;                                      // floats: [0,1] W scale / absmax | [2,3] max|ds|, max|wc| | [4,5] dz scale / bound
    w.total = o + 64;
    return w;
}

}  // namespace mdl

using namespace mdl;

extern "C" int64_t mdl_abmil_gate_fwd_split_ws_bytes(int64_t T, int H) {
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS) return MDL_E_ARG;
    // WK image [H][1024][512] | score partials [T][H][4] | scale floats
    return (int64_t)H * 1024 * HID * 4 + T * H * GATE_JT * 4 + 128;
}

/* mdl_abmil_gate_fwd on the split engine: E as a split image (rows of e_rsb bytes holding the H*512 head-major channels, scale e_scale);
 * everything else as mdl_abmil_gate_fwd (fp32 parameters, scores, saved activations). */
extern "C" int mdl_abmil_gate_fwd_split(const void* E_img, int64_t e_rsb, const float* e_scale, const float* Wa, const float* ba,
                                        const float* Wb, const float* bb, const float* wc, const float* bc, float* scores, float* act_a,
                                        float* act_b, int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a,
                                        const uint8_t* keep_b, void* ws, void* stream) {
    if (!E_img || !e_scale || !Wa || !ba || !Wb || !bb || !wc || !bc || !scores || !ws) return MDL_E_ARG;
    if ((act_a == nullptr) != (act_b == nullptr)) return MDL_E_ARG;
    if ((keep_a == nullptr) != (keep_b == nullptr)) return MDL_E_ARG;
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS || e_rsb < (int64_t)H * HID * 4 || (e_rsb & 15) || e_rsb * SPM > 0x7fffffff) return MDL_E_ARG;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return MDL_E_ARG;
    if (!host_aligned16(E_img) || !host_aligned16(Wa) || !host_aligned16(Wb) || !host_aligned16(ws)) return MDL_E_ALIGN;
    if (T == 0) return MDL_OK;
    if (H != 1 && H != 2 && H != 4 && H != 8) return MDL_E_UNSUPPORTED;
    const int64_t n_tt = (T + SPM - 1) / SPM;
    const int64_t grid = xcd_head_grid(n_tt, GATE_JT, H);
    if (grid > 0x7fffffff) return MDL_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const DropCfg d = make_drop(p_drop, seed, keep_a, keep_b);
    char* WK = (char*)ws;
    float* part = (float*)(WK + (int64_t)H * 1024 * HID * 4);
    float* sc = part + T * H * GATE_JT;
    {
        const hipError_t e = hipMemsetAsync(sc, 0, 2 * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    int rc = sp_launch_absmax(Wa, HID, (int64_t)H * HID, HID, sc + 1, s);
    if (rc) return rc;
    rc = sp_launch_absmax(Wb, HID, (int64_t)H * HID, HID, sc + 1, s);
    if (rc) return rc;
    rc = sp_launch_scale(sc, s);
    if (rc) return rc;
    hipLaunchKernelGGL(sp_gate_wk_kernel, dim3((unsigned)((int64_t)H * 1024 * 64 / 256)), dim3(256), 0, s, Wa, Wb, WK, H, (const float*)sc);
    MDL_LAUNCH_CHECK();
    const int dm = gate_drop_mode(d);
    // persistent workgroups (round 5, see sp_gate_fwd_kernel): MADELEINE_GATE_PERSIST = 0 | 1 | 2 picks the mode (A/B switch)
    static const int pmode_env = getenv("MADELEINE_GATE_PERSIST") ? atoi(getenv("MADELEINE_GATE_PERSIST")) : MDL_GATE_SP_PMODE;
    const int nshare = 8 / H;
    const int64_t per_share = (n_tt + nshare - 1) / nshare;
    int pmode = pmode_env;
    if (pmode == 1 && !gate_persist_pays(grid, 0.96)) pmode = 0;
    if (pmode == 2 && !gate_persist_pays(grid, 0.96, GATE_PT)) pmode = 0;
    const int64_t pgrid = pmode == 1 ? grid / GATE_JT : pmode == 2 ? 8 * ((per_share + GATE_PT - 1) / GATE_PT) * GATE_JT : grid;
    const bool na3 = sp_nt_stages() == 3;   // sp_nt_mainloop3 (MADELEINE_SP_NT_STAGES, split_engine.hpp)
#define MDL_GATE_FWD_SP1(DM, SAVE, PM)                                                                                                 \
    hipLaunchKernelGGL((sp_gate_fwd_kernel<DM, SAVE, PM>), dim3((unsigned)pgrid), dim3(SP_THREADS), 0, s, (const char*)E_img, e_rsb,   \
                       e_scale, (const char*)WK, (const float*)sc, ba, bb, wc, part, act_a, act_b, T, H, (int)n_tt, d)
#define MDL_GATE_FWD_SP(DM, SAVE)                                                                                                     \
    do {                                                                                                                              \
        if (pmode == 1) MDL_GATE_FWD_SP1(DM, SAVE, 1);                                                                                \
        else if (pmode == 2) MDL_GATE_FWD_SP1(DM, SAVE, 2);                                                                           \
        else if (na3) hipLaunchKernelGGL((sp_gate_fwd_kernel<DM, SAVE, 0, 3>), dim3((unsigned)pgrid), dim3(SP_THREADS), 0, s,             \
                                         (const char*)E_img, e_rsb, e_scale, (const char*)WK, (const float*)sc, ba, bb, wc, part, act_a,  \
                                         act_b, T, H, (int)n_tt, d);                                                                  \
        else MDL_GATE_FWD_SP1(DM, SAVE, 0);                                                                                           \
    } while (0)
    if (act_a) {
        if (dm == 0) MDL_GATE_FWD_SP(0, true);
        else if (dm == 1) MDL_GATE_FWD_SP(1, true);
        else if (dm == 3) MDL_GATE_FWD_SP(3, true);
        else MDL_GATE_FWD_SP(2, true);
    } else {
        if (dm == 0) MDL_GATE_FWD_SP(0, false);
        else if (dm == 1) MDL_GATE_FWD_SP(1, false);
        else if (dm == 3) MDL_GATE_FWD_SP(3, false);
        else MDL_GATE_FWD_SP(2, false);
    }
#undef MDL_GATE_FWD_SP
#undef MDL_GATE_FWD_SP1
    MDL_LAUNCH_CHECK();
    return gate_launch_finalize(part, bc, scores, T * H, H, s);
}

extern "C" int64_t mdl_abmil_gate_bwd_split_ws_bytes(int64_t T, int H) {
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS) return MDL_E_ARG;
    return sp_bwd_ws(T, H).total;
}

/* mdl_abmil_attnpool_bwd (scores == NULL: plain mdl_abmil_gate_bwd) on the split engine; E as a split image.  dE_absmax (device float,
 * may be NULL; zeroed by the caller) is raised to max |dE|.  phases as mdl_abmil_attnpool_bwd_phases. */
extern "C" int mdl_abmil_attnpool_bwd_split(const void* E_img, int64_t e_rsb, const float* e_scale, const float* Wa, const float* Wb,
                                            const float* wc, const float* act_a, const float* act_b, const float* d_scores, float* dE,
                                            int64_t ldE, int accumulate, float* dWa, float* dWb, float* dba, float* dbb, float* dwc,
                                            float* dbc, int64_t T, int H, float p_drop, uint64_t seed, const uint8_t* keep_a,
                                            const uint8_t* keep_b, const float* scores, const float* stat_m, const float* stat_l,
                                            const float* d_pooled, const int32_t* row_bag, int64_t N, float* dE_absmax, void* ws,
                                            void* stream, int phases, int terms) {
    if (terms != 2 && terms != 3) return MDL_E_ARG;
    if (!E_img || !e_scale || !Wa || !Wb || !wc || !act_a || !act_b || !d_scores || !dE || !dWa || !dWb || !dba || !dbb || !dwc || !ws)
        return MDL_E_ARG;
    if (phases < 1 || phases > 15) return MDL_E_ARG;
    const bool do_dx = (phases & (2 | 4)) != 0, do_dw = (phases & (2 | 8)) != 0;   // bit 1 = both contractions; bit 2 = dX only, bit 3 = dW only
    if ((keep_a == nullptr) != (keep_b == nullptr)) return MDL_E_ARG;
    if (scores && (!stat_m || !stat_l || !d_pooled || (!row_bag && N < 1))) return MDL_E_ARG;
    if (T < 0 || H < 1 || H > MDL_MAX_HEADS || ldE < (int64_t)H * HID || (ldE & 3) || e_rsb < (int64_t)H * HID * 4 || (e_rsb & 15) ||
        e_rsb * SPK > 0x7fffffff)
        return MDL_E_ARG;
    if (H != 1 && H != 2 && H != 4 && H != 8) return MDL_E_UNSUPPORTED;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return MDL_E_ARG;
    if (!host_aligned16(E_img) || !host_aligned16(dE) || !host_aligned16(Wa) || !host_aligned16(Wb) || !host_aligned16(act_a) ||
        !host_aligned16(act_b) || !host_aligned16(wc) || !host_aligned16(ws))
        return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const DropCfg d = make_drop(p_drop, seed, keep_a, keep_b);
    const SpBwdWs L = sp_bwd_ws(T, H);
    char* base = (char*)ws;
    char* WN = base + L.oWN;
    char* dzi = base + L.odz;
    float* slabW = (float*)(base + L.oslabW);
    float* slabV = (float*)(base + L.oslabV);
    float* sc = (float*)(base + L.osc);
    if (L.nblk * H > 0x7fffffff) return MDL_E_UNSUPPORTED;
    const PoolTerm pt{scores, stat_m, stat_l, d_pooled, row_bag, N};
    if (phases & 1) {
        hipError_t e = hipMemsetAsync(sc, 0, 8 * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
        e = hipMemsetAsync(dzi + T * H * (int64_t)4096, 0, (size_t)SPK * H * 4096, s);   // zero pad rows of dz
        if (e != hipSuccess) return (int)e;
        // scales: weights from their exact absmax; dz from the bound |dz| <= max|ds| max|wc| / (1-p)^2
        int rc = sp_launch_absmax(Wa, HID, (int64_t)H * HID, HID, sc + 1, s);
        if (rc) return rc;
        rc = sp_launch_absmax(Wb, HID, (int64_t)H * HID, HID, sc + 1, s);
        if (rc) return rc;
        rc = sp_launch_scale(sc, s);
        if (rc) return rc;
        rc = sp_launch_absmax_flat(d_scores, T * H, sc + 2, s);
        if (rc) return rc;
        rc = sp_launch_absmax_flat(wc, (int64_t)H * HID, sc + 3, s);
        if (rc) return rc;
        hipLaunchKernelGGL(sp_bound_scale_kernel, dim3(1), dim3(1), 0, s, (const float*)(sc + 2), d.inv * d.inv, sc + 4);
        MDL_LAUNCH_CHECK();
        if (T > 0) {
            const int dm = gate_drop_mode(d);
#define MDL_GATE_DZ_SP(DM)                                                                                                    \
    hipLaunchKernelGGL(sp_gate_dz_kernel<DM>, dim3((unsigned)L.nblk), dim3(64 * H), 0, s, wc, act_a, act_b, d_scores, dzi,    \
                       (const float*)(sc + 4), slabV, T, H, d, sp_dz_rows(T))
            if (dm == 0) MDL_GATE_DZ_SP(0);
            else if (dm == 1) MDL_GATE_DZ_SP(1);
            else if (dm == 3) MDL_GATE_DZ_SP(3);
            else MDL_GATE_DZ_SP(2);
#undef MDL_GATE_DZ_SP
            MDL_LAUNCH_CHECK();
        }
        rc = gate_launch_reduce_v(slabV, dba, dbb, dwc, dbc, H, (int)L.nblk, s);
        if (rc) return rc;
    }
    if (do_dx && T > 0) {
        hipLaunchKernelGGL(sp_gate_wn_kernel, dim3(16, 32, H), dim3(256), 0, s, Wa, Wb, WN, (const float*)sc);
        MDL_LAUNCH_CHECK();
        const int64_t n_tt = (T + SPM - 1) / SPM;
        const int64_t grid = xcd_head_grid(n_tt, 2, H);
        if (grid > 0x7fffffff) return MDL_E_UNSUPPORTED;
        const bool na3 = sp_nt_stages() == 3;
        hipLaunchKernelGGL(terms == 2 ? (na3 ? sp_gate_dx_kernel<2, 3> : sp_gate_dx_kernel<2, 2>) : (na3 ? sp_gate_dx_kernel<3, 3> : sp_gate_dx_kernel<3, 2>), dim3((unsigned)grid), dim3(SP_THREADS), 0, s, (const char*)dzi, (const float*)(sc + 4),
                           (const char*)WN, (const float*)sc, dE, ldE, accumulate, T, H, pt, dE_absmax);
        MDL_LAUNCH_CHECK();
    }
    if (do_dw) {
        const bool tn3 = sp_tn_stages() == 3;
        hipLaunchKernelGGL(terms == 2 ? (tn3 ? sp_gate_dw_kernel<2, 3> : sp_gate_dw_kernel<2, 2>) : (tn3 ? sp_gate_dw_kernel<3, 3> : sp_gate_dw_kernel<3, 2>), dim3((unsigned)xcd_head_grid(L.S, 8, H)), dim3(SP_THREADS), 0, s, (const char*)E_img, e_rsb, e_scale,
                           (const char*)dzi, (const float*)(sc + 4), slabW, T, H, L.tps, L.S);
        MDL_LAUNCH_CHECK();
        const int rc = gate_launch_reduce_w(slabW, dWa, dWb, H, L.S, s);
        if (rc) return rc;
    }
    return MDL_OK;
}
