// split_gemm.hip -- the generic entry points of the split-fp16 engine (split_engine.hpp): image construction from fp32 tensors and
// the two contraction forms on images.  They carry every Linear of the encoder in the "split" GEMM mode (functional.LinearFn):
//     forward : Y = X W^T            mdl_split_gemm_nt(A = image(X),  B = image(W))          reference Model.py:351, :355, :359
//     dX      : dX = dY W            mdl_split_gemm_nt(A = image(dY), B = image(W^T))
//     dW      : dW = dY^T X          mdl_split_gemm_tn(A = image(X),  B = image(dY)), split over tokens, slabs reduced + transposed
// (the gate kernels with their fused epilogues on the same engine live in abmil_gate_split.hip).
#include "split_engine.hpp"

namespace mdl {

int lin_launch_reduce(const float* slab, float* dW, int K, int N, int S, hipStream_t s);   // linear_fp32.hip

// ---- image construction -------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sp_absmax_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int K,
                                                        float* __restrict__ out) {
    const int64_t n4 = rows * (K / 4);
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (K / 4);
        const int c = (int)(i % (K / 4)) * 4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(X + r * ldx + c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    sp_atomic_absmax(out, m);
}
// sc[0] = scale from the absmax in sc[1]
__global__ void sp_scale_kernel(float* __restrict__ sc) { sc[0] = sp_scale_for(sc[1]); }

__global__ __launch_bounds__(256) void sp_absmax_flat_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(x[i]));
    sp_atomic_absmax(out, m);
}
// launchers shared with abmil_gate_split.hip / preattn_act.hip: *out is raised to max |X| (the caller zeroes it)
int sp_launch_absmax(const float* X, int64_t ldx, int64_t rows, int K, float* out, hipStream_t s) {
    if (rows <= 0) return MDL_OK;
    const int64_t n4 = rows * (K / 4);
    int nb = (int)((n4 + 256 * 8 - 1) / (256 * 8));
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(sp_absmax_kernel, dim3(nb), dim3(256), 0, s, X, ldx, rows, K, out);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}
int sp_launch_absmax_flat(const float* x, int64_t n, float* out, hipStream_t s) {
    if (n <= 0) return MDL_OK;
    int nb = (int)((n + 256 * 8 - 1) / (256 * 8));
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(sp_absmax_flat_kernel, dim3(nb), dim3(256), 0, s, x, n, out);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}
int sp_launch_scale(float* sc, hipStream_t s) {   // sc[0] = scale for the absmax in sc[1]
    hipLaunchKernelGGL(sp_scale_kernel, dim3(1), dim3(1), 0, s, sc);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

// img[r][k / 32][plane][k % 32] = planes of sc[0] * X[r][k]; one thread = 8 consecutive k
__global__ __launch_bounds__(256) void sp_convert_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int K,
                                                         char* __restrict__ img, int64_t rsb, const float* __restrict__ sc) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int g = K / 8;
    if (i >= rows * g) return;
    const int64_t r = i / g;
    const int k = (int)(i % g) * 8;
    const float s = sc[0];
    const f32x4 x0 = *reinterpret_cast<const f32x4*>(X + r * ldx + k), x1 = *reinterpret_cast<const f32x4*>(X + r * ldx + k + 4);
    const float v[8] = {x0.x * s, x0.y * s, x0.z * s, x0.w * s, x1.x * s, x1.y * s, x1.z * s, x1.w * s};
    u32x4 hi, lo;
    sp_split8(v, hi, lo);
    char* row = img + r * rsb;
    *reinterpret_cast<u32x4*>(row + sp_img_off(k, 0)) = hi;
    *reinterpret_cast<u32x4*>(row + sp_img_off(k, 1)) = lo;
}

// Row-scaled image (mdl_split_image_rows): every row r carries its own power-of-two scale s_r with max_k |s_r X[r][k]| in [2^13, 2^14)
// (an all-zero row: row_inv = 0) -- the image of a tensor whose ROWS differ in magnitude by more than the ~2^16 a common scale represents
// at full precision (the patch features a caller hands in: one outlier patch must not cost the other patches their low bits).
// row_inv[r] = 1 / s_r.  One wave per row: pass 1 the row maximum, pass 2 (the row is in L1 / L2) the planes; *absmax is raised to max |X|.
__global__ __launch_bounds__(256) void sp_image_rows_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int K,
                                                            char* __restrict__ img, int64_t rsb, float* __restrict__ row_inv,
                                                            float* __restrict__ absmax) {
    const int lane = threadIdx.x & 63;
    float tot = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (int64_t)gridDim.x * 4) {
        const float* __restrict__ xr = X + r * ldx;
        float m = 0.f;
        for (int k = lane * 8; k < K; k += 512) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(xr + k), x1 = *reinterpret_cast<const f32x4*>(xr + k + 4);
            m = fmaxf(m, fmaxf(fmaxf(fmaxf(fabsf(x0.x), fabsf(x0.y)), fmaxf(fabsf(x0.z), fabsf(x0.w))),
                               fmaxf(fmaxf(fabsf(x1.x), fabsf(x1.y)), fmaxf(fabsf(x1.z), fabsf(x1.w)))));
        }
        m = wave_max(m);
        tot = fmaxf(tot, m);
        const float s = sp_scale_for(m);
        // exact: s is a power of two.  An all-zero row (absent-stain bag, wsi_dataset.py:66) gets factor 0: its products vanish anyway, and
        // as row_mul of the paired gradient image (mdl_ln_gelu_drop_bwd_split) 0 keeps that row -- which contributes x = 0 to dW -- from
        // setting the gradient image's scale
        if (lane == 0) row_inv[r] = m > 0.f ? 1.f / s : 0.f;
        char* row = img + r * rsb;
        for (int k = lane * 8; k < K; k += 512) {
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(xr + k), x1 = *reinterpret_cast<const f32x4*>(xr + k + 4);
            const float v[8] = {x0.x * s, x0.y * s, x0.z * s, x0.w * s, x1.x * s, x1.y * s, x1.z * s, x1.w * s};
            u32x4 hi, lo;
            sp_split8(v, hi, lo);
            *reinterpret_cast<u32x4*>(row + sp_img_off(k, 0)) = hi;
            *reinterpret_cast<u32x4*>(row + sp_img_off(k, 1)) = lo;
        }
    }
    if (lane == 0 && absmax) atomicMax(reinterpret_cast<unsigned int*>(absmax), __float_as_uint(tot));
}
// K <= 512 SEG: SEG x 8 values per lane and row stay in registers, ROWS rows per wave in flight (2 ROWS SEG 16-B loads per lane before the
// first reduction): one HBM read of X, no second pass.  (The two-pass kernel above moved 1.07 GB in 415 us at config 2 = 2.6 TB/s.)
template <int SEG, int ROWS>
__global__ __launch_bounds__(256) void sp_image_rows_reg_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int K,
                                                                char* __restrict__ img, int64_t rsb, float* __restrict__ row_inv,
                                                                float* __restrict__ absmax) {
    const int lane = threadIdx.x & 63;
    float tot = 0.f;
    // -DMDL_ACT_CONTIG: one contiguous range of row groups per workgroup, its four waves interleaved inside it (round-5 experiment: a plain
    // 1 read : 1 write stream gains 15-25 % from that order, tools/micro/hbm_rate.hip; this kernel does not -- 0.207 vs 0.202 ms)
#ifndef MDL_ACT_CONTIG
    const int64_t stride = (int64_t)gridDim.x * 4 * ROWS, r_begin = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS, r_end = rows;
#else
    const int64_t ng = (rows + ROWS - 1) / ROWS, per = ((ng + gridDim.x - 1) / gridDim.x + 3) / 4 * 4;
    const int64_t g0 = (int64_t)blockIdx.x * per, g1 = g0 + per < ng ? g0 + per : ng;
    const int64_t stride = 4 * ROWS, r_begin = (g0 + (threadIdx.x >> 6)) * ROWS, r_end = g1 * ROWS < rows ? g1 * ROWS : rows;
#endif
    for (int64_t r0 = r_begin; r0 < r_end; r0 += stride) {
        f32x4 v[ROWS][SEG][2];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int64_t r = r0 + j < rows ? r0 + j : rows - 1;
            const float* __restrict__ xr = X + r * ldx;
#pragma unroll
            for (int g = 0; g < SEG; ++g) {
                const int k = g * 512 + lane * 8;
                if (k < K) {
                    v[j][g][0] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + k));
                    v[j][g][1] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(xr + k + 4));
                } else {
                    v[j][g][0] = f32x4{0.f, 0.f, 0.f, 0.f};
                    v[j][g][1] = v[j][g][0];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            float m = 0.f;
#pragma unroll
            for (int g = 0; g < SEG; ++g)
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    m = fmaxf(m, fmaxf(fmaxf(fabsf(v[j][g][h].x), fabsf(v[j][g][h].y)), fmaxf(fabsf(v[j][g][h].z), fabsf(v[j][g][h].w))));
            m = wave_max(m);
            if (r0 + j >= rows) continue;   // wave-uniform
            tot = fmaxf(tot, m);
            const float s = sp_scale_for(m);
            if (lane == 0) row_inv[r0 + j] = m > 0.f ? 1.f / s : 0.f;
            char* row = img + (r0 + j) * rsb;
#pragma unroll
            for (int g = 0; g < SEG; ++g) {
                const int k = g * 512 + lane * 8;
                if (k < K) {
                    const f32x4 a = v[j][g][0], b = v[j][g][1];
                    const float w[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
                    u32x4 hi, lo;
                    sp_split8(w, hi, lo);
                    *reinterpret_cast<u32x4*>(row + sp_img_off(k, 0)) = hi;
                    *reinterpret_cast<u32x4*>(row + sp_img_off(k, 1)) = lo;
                }
            }
        }
    }
    if (absmax) {   // kernel-uniform; one atomic per workgroup (all waves of a short launch arrive here together)
        __shared__ float bmax[4];
        if (lane == 0) bmax[threadIdx.x >> 6] = tot;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(reinterpret_cast<unsigned int*>(absmax), __float_as_uint(fmaxf(fmaxf(bmax[0], bmax[1]), fmaxf(bmax[2], bmax[3]))));
    }
}
// sc = {1, absmax already in sc[1]}: the common factor of a row-scaled image is 1 (the row factors travel in row_inv)
__global__ void sp_unit_scale_kernel(float* __restrict__ sc) { sc[0] = 1.f; }

// gate[i] = max |X[256 i .. 256 i + 255][:]|: lets mdl_split_gemm_nt skip the output tiles whose A rows are all zero (the
// token_projector's dX in the fused A2 + A3 backward: the local loss reads the first <= 256 tokens of a bag, the rest of d_tok is 0);
// chunk_max[j] (optional) = the same over the 32 rows of chunk j: the TN product skips all-zero chunks (mdl_split_gemm_tn).
__global__ __launch_bounds__(256) void sp_tile_absmax_kernel(const float* __restrict__ X, int64_t ldx, int64_t rows, int K,
                                                             float* __restrict__ gate, float* __restrict__ chunk_max) {
    __shared__ float red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * SPM;
    const int g = K / 4;
    float mt = 0.f;
    for (int c = wave; c < SPM / SPK; c += 4) {   // one wave per 32-row chunk
        const int64_t c0 = r0 + (int64_t)c * SPK;
        if (c0 >= rows) break;
        int64_t c1 = c0 + SPK;
        if (c1 > rows) c1 = rows;
        float m = 0.f;
        for (int64_t i = lane; i < (c1 - c0) * g; i += 64) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(X + (c0 + i / g) * ldx + (i % g) * 4);
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
        }
        m = wave_max(m);
        if (chunk_max && lane == 0) chunk_max[c0 / SPK] = m;
        mt = fmaxf(mt, m);
    }
    if (lane == 0) red[wave] = mt;
    __syncthreads();
    if (threadIdx.x == 0) gate[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- NT -------------------------------------------------------------------------------------------------------------------------------
#ifdef MDL_SP_PROBE   // tools/ab variant only: s_memtime stamps of every workgroup of the last sp_nt launch (entry | loop start | loop end | exit | hw id)
__device__ unsigned long long sp_probe_buf[8192 * 5];
#endif
// C[m][n] (+)= inv * sum_k A[m][k] B[n][k] (+ bias[n]);  rows m >= M / n >= N re-read the last valid row (discarded).
template <int TERMS, int NA>
__global__ __launch_bounds__(SP_THREADS) void sp_nt_kernel(const char* __restrict__ A, int64_t a_rsb, const float* __restrict__ a_sc,
                                                    const char* __restrict__ B, int64_t b_rsb, const float* __restrict__ b_sc,
                                                    float* __restrict__ C, int64_t ldc, int64_t M, int N, int nblk, int n_tiles,
                                                    const float* __restrict__ bias, int accumulate, float* __restrict__ absmax_out,
                                                    const float* __restrict__ row_gate, const float* __restrict__ a_row_mul,
                                                    const float* __restrict__ b_col_mul, const float* __restrict__ group_bias = nullptr,
                                                    const int32_t* __restrict__ row_group = nullptr) {
    // group_bias [G][N] + row_group [M] (round 5): C[m][:] += group_bias[row_group[m]][:] -- a bias row per GROUP of rows (MADELEINE's
    // stain-encoding columns of the first Linear as a per-bag bias: [x | e_g] W^T = x Wx^T + e_g We^T, Model.py:132, :351)
    // NA: stages of the A ring (split_engine.hpp, SmemSPn): 3 = sp_nt_mainloop3 (LDS-DMA pieces spread over the chunk, 160 KiB of LDS)
    __shared__ SmemSPn<NA> sm3;
    SmemSP& sm = reinterpret_cast<SmemSP&>(sm3);   // (epilogue staging: the first 128 KiB, whatever the ring depth)
#ifdef MDL_SP_PROBE
    const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / SP_WN, wn = wave % SP_WN;
    const int ncol = (N + SPN - 1) / SPN;
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int nt = lid % ncol;
    if (row_gate && row_gate[lid / ncol] == 0.f) return;   // block-uniform: the 256 A rows of this tile are all zero (accumulate mode)
    const int64_t m0 = (int64_t)(lid / ncol) * SPM;
    const int n0 = nt * SPN;

    const char* baseA = A + m0 * a_rsb;
    const char* baseB = B + (int64_t)n0 * b_rsb;
    uint32_t voA[SP_PW], voB[SP_PW];
#pragma unroll
    for (int i = 0; i < SP_PW; ++i) {
        int row, c;
        sp_nt_slot(wave, i, lane, row, c);
        int64_t ra = row, rb = row;
        if (m0 + ra > M - 1) ra = M - 1 - m0;
        if (n0 + rb > N - 1) rb = N - 1 - n0;
        voA[i] = (uint32_t)(ra * a_rsb + c * 16);
        voB[i] = (uint32_t)(rb * b_rsb + c * 16);
    }
    SpAcc acc;
    sp_zero(acc);
    auto dma = [&](int st, int f, int piece) {
        const int i = piece % SP_PW;
        if (piece < SP_PW) glds16_s(voA[i], sp_uniform(baseA + (int64_t)f * 128), lds_addr_of(&sm3.A[st][(wave * SP_PW + i) * 1024]));
        else glds16_s(voB[i], sp_uniform(baseB + (int64_t)f * 128), lds_addr_of(&sm3.B[st][(wave * SP_PW + i) * 1024]));
    };
#ifdef MDL_SP_PROBE
    const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
#endif
    if constexpr (NA == 3) sp_nt_mainloop3<TERMS>(sm3, acc, nblk, wm, wn, lane, dma);
    else sp_nt_mainloop<TERMS>(sm3, acc, nblk, wm, wn, lane, dma);
#ifdef MDL_SP_PROBE
    const unsigned long long pt2 = __builtin_amdgcn_s_memtime();
#endif
    const float inv = 1.f / (a_sc[0] * b_sc[0]);
    float amax = 0.f;
    char* cb = reinterpret_cast<char*>(C + m0 * ldc + n0);
    const uint32_t ldc4 = (uint32_t)ldc * 4u;
    const int cols_valid = N - n0;
    auto emit = [&](int row, int col, const f32x4& v) {
        if (col >= cols_valid) return;
        f32x4 r = v * (a_row_mul ? inv * a_row_mul[m0 + row] : inv);   // row-scaled A image: its row factor (a power of two) comes back here
        if (b_col_mul) r *= *reinterpret_cast<const f32x4*>(b_col_mul + n0 + col);   // row-scaled B image (a weight: one factor per output column)
        if (bias) r += *reinterpret_cast<const f32x4*>(bias + n0 + col);
        if (group_bias) r += *reinterpret_cast<const f32x4*>(group_bias + (int64_t)row_group[m0 + row] * N + n0 + col);
        f32x4* o = reinterpret_cast<f32x4*>(cb + (int64_t)row * ldc4 + (uint32_t)col * 4u);
        if (accumulate) r += *o;
        *o = r;   // (nontemporal stores here: null, 21.60-21.62 vs 21.63-21.64 ms, profiles/r06zc_*)
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
    };
    // (Round 6: storing straight from the MFMA accumulator layout -- one dword store of a wave = two whole 128-B lines, no LDS transpose --
    // measured SLOWER: 262144 x 2048 x 512 1.62 -> 1.84 ms, profiles/r06zb_*: the epilogue is bound by store issue, and 128 dword stores
    // per wave cost more than 32 dwordx4 ones.)
    if (m0 + SPM <= M) sp_epilogue_rows<true>(acc, sm, wave, wm, wn, lane, SPM, emit);
    else sp_epilogue_rows<false>(acc, sm, wave, wm, wn, lane, (int)(M - m0), emit);
    if (absmax_out) sp_block_absmax(absmax_out, amax, reinterpret_cast<float*>(&sm.B[1][0]));   // (B stages: outside the epilogue's transpose areas)
#ifdef MDL_SP_PROBE
    if (tid == 0 && blockIdx.x < 8192) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* o = sp_probe_buf + (size_t)blockIdx.x * 5;
        o[0] = pt0; o[1] = pt1; o[2] = pt2; o[3] = __builtin_amdgcn_s_memtime(); o[4] = ((unsigned long long)xcc << 32) | hw;
    }
#endif
}
#ifdef MDL_SP_PROBE
extern "C" int mdl_debug_sp_probe_read(unsigned long long* host_out, int n_wg) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(mdl::sp_probe_buf), (size_t)n_wg * 5 * sizeof(unsigned long long));
}
#endif

// The same product on the tall tile (512 rows x 128 columns per workgroup) for N <= 128: no row gate (callers with a row gate have wide
// outputs), everything else as sp_nt_kernel.
template <int TERMS>
__global__ __launch_bounds__(SP_THREADS) void sp_nt_tall_kernel(const char* __restrict__ A, int64_t a_rsb, const float* __restrict__ a_sc,
                                                         const char* __restrict__ B, int64_t b_rsb, const float* __restrict__ b_sc,
                                                         float* __restrict__ C, int64_t ldc, int64_t M, int N, int nblk, int n_tiles,
                                                         const float* __restrict__ bias, int accumulate, float* __restrict__ absmax_out,
                                                         const float* __restrict__ a_row_mul, const float* __restrict__ b_col_mul) {
    __shared__ SmemSPT sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int64_t m0 = (int64_t)lid * SPT_M;
    const char* baseA = A + m0 * a_rsb;
    uint32_t voA[SPT_PA], voB[SPT_PB];
#pragma unroll
    for (int i = 0; i < SPT_PA; ++i) {
        int row, c;
        sp_tall_slot(wave, i, SPT_PA, lane, row, c);
        int64_t ra = row;
        if (m0 + ra > M - 1) ra = M - 1 - m0;
        voA[i] = (uint32_t)(ra * a_rsb + c * 16);
    }
#pragma unroll
    for (int j = 0; j < SPT_PB; ++j) {
        int row, c;
        sp_tall_slot(wave, j, SPT_PB, lane, row, c);
        if (row > N - 1) row = N - 1;
        voB[j] = (uint32_t)((int64_t)row * b_rsb + c * 16);
    }
    SpAccT acc;
    sp_zero(acc);
    sp_nt_tall_mainloop<TERMS>(
        sm, acc, nblk, wave, lane,
        [&](int st, int f, int i) { glds16_s(voA[i], sp_uniform(baseA + (int64_t)f * 128), lds_addr_of(&sm.A[st][(wave * SPT_PA + i) * 1024])); },
        [&](int st, int f, int j) { glds16_s(voB[j], sp_uniform(B + (int64_t)f * 128), lds_addr_of(&sm.B[st][(wave * SPT_PB + j) * 1024])); });
    const float inv = 1.f / (a_sc[0] * b_sc[0]);
    float amax = 0.f;
    char* cb = reinterpret_cast<char*>(C + m0 * ldc);
    const uint32_t ldc4 = (uint32_t)ldc * 4u;
    auto emit = [&](int row, int col, const f32x4& v) {
        if (col >= N) return;
        f32x4 r = v * (a_row_mul ? inv * a_row_mul[m0 + row] : inv);
        if (b_col_mul) r *= *reinterpret_cast<const f32x4*>(b_col_mul + col);
        if (bias) r += *reinterpret_cast<const f32x4*>(bias + col);
        f32x4* o = reinterpret_cast<f32x4*>(cb + (int64_t)row * ldc4 + (uint32_t)col * 4u);
        if (accumulate) r += *o;
        *o = r;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
    };
    if (m0 + SPT_M <= M) sp_epilogue_rows_tall<true>(acc, sm, wave, lane, SPT_M, emit);
    else sp_epilogue_rows_tall<false>(acc, sm, wave, lane, (int)(M - m0), emit);
    if (absmax_out) sp_block_absmax(absmax_out, amax, reinterpret_cast<float*>(&sm.B[1][0]));
}

// Chunk lists for the TN loop: list[sp][0 .. count[sp]) = the 32-row chunks of split sp whose chunk_max (sp_tile_absmax_kernel over the
// fp32 tensor the B image was built from) is not zero, in ascending order.  One workgroup per split; flags in LDS, serial compaction.
constexpr int SP_MAX_LIST = 2048;
__global__ __launch_bounds__(256) void sp_chunk_list_kernel(const float* __restrict__ chunk_max, int64_t T, int64_t tok_per_split,
                                                            int32_t* __restrict__ list, int32_t* __restrict__ count, int stride) {
    __shared__ uint8_t flag[SP_MAX_LIST];
    const int sp = blockIdx.x;
    const int64_t ts = (int64_t)sp * tok_per_split;   // a multiple of 32
    int64_t te = ts + tok_per_split;
    if (te > T) te = T;
    const int nc = te > ts ? (int)((te - ts + SPK - 1) / SPK) : 0;
    for (int c = threadIdx.x; c < nc; c += 256) flag[c] = !(chunk_max[ts / SPK + c] == 0.f);   // (NaN counts as non-zero)
    __syncthreads();
    if (threadIdx.x == 0) {
        int n = 0;
        for (int c = 0; c < nc; ++c)
            if (flag[c]) list[(int64_t)sp * stride + n++] = c;
        count[sp] = n;
    }
}

// ---- TN -------------------------------------------------------------------------------------------------------------------------------
// slab[sp][m][n] = inv * sum_{t in split sp} A[t][m] B[t][n].  A rows t >= T re-read row T - 1; the B image must continue with >= 32
// all-zero rows after row T - 1 (their products vanish).  Columns >= Mi / >= N fetch column 0 (discarded).
template <int TERMS, int NA = 2>   // NA = 3: sp_tn_mainloop3 on SmemSP3
__global__ __launch_bounds__(SP_THREADS) void sp_tn_kernel(const char* __restrict__ A, int64_t a_rsb, const float* __restrict__ a_sc, int Mi,
                                                    const char* __restrict__ B, int64_t b_rsb, const float* __restrict__ b_sc, int N,
                                                    float* __restrict__ slab, int64_t T, int64_t tok_per_split, int n_tiles,
                                                    const int32_t* __restrict__ chunk_list, const int32_t* __restrict__ chunk_count,
                                                    int list_stride) {
    __shared__ SmemSPn<NA> sm3;
    SmemSP& sm = reinterpret_cast<SmemSP&>(sm3);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / SP_WN, wn = wave % SP_WN;
    const int nmt = (Mi + SPM - 1) / SPM, nnt = (N + SPN - 1) / SPN;
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int mt = lid % nmt, nt = (lid / nmt) % nnt, sp = lid / (nmt * nnt);
    const int i0 = mt * SPM, n0 = nt * SPN;
    const int64_t ts = (int64_t)sp * tok_per_split;
    int64_t te = ts + tok_per_split;
    if (te > T) te = T;
    // with a chunk list only the chunks in which the B operand is not identically zero are visited (logical index -> chunk of the split)
    const int32_t* __restrict__ clist = chunk_list ? chunk_list + (int64_t)sp * list_stride : nullptr;
    const int64_t nch = clist ? chunk_count[sp] : ((te > ts) ? (te - ts + SPK - 1) / SPK : 0);

    // per piece q: kr = (4w + q) * 2 + (lane >> 5): plane = kr >> 5, token = kr & 31; global chunk src = (lane & 31) ^ ((kr & 3) << 2)
    uint32_t tokq[SP_PW], coA[SP_PW], coB[SP_PW];
#pragma unroll
    for (int q = 0; q < SP_PW; ++q) {
        const int kr = (wave * SP_PW + q) * 2 + (lane >> 5), p = kr >> 5, src = (lane & 31) ^ ((kr & 3) << 2);
        tokq[q] = kr & 31;
        const int ca = i0 + src * 8, cbn = n0 + src * 8;
        coA[q] = (uint32_t)sp_img_off(ca < Mi ? ca : 0, p);
        coB[q] = (uint32_t)sp_img_off(cbn < N ? cbn : 0, p);
    }
    const char* baseA = A + ts * a_rsb;
    const char* baseB = B + ts * b_rsb;
    SpAcc acc;
    sp_zero(acc);
    auto dma = [&](int st, int64_t fl, int piece) {
        const int q = piece % SP_PW;
        const int64_t f = clist ? (int64_t)clist[fl] : fl;
        if (piece < SP_PW) {
            uint32_t tk = tokq[q];
            const int64_t left = T - 1 - (ts + f * SPK);   // >= 0
            if (left < SPK) tk = tk < (uint32_t)left ? tk : (uint32_t)left;   // uniform branch: only the chunk at the end of A
            glds16_s(tk * (uint32_t)a_rsb + coA[q], sp_uniform(baseA + f * SPK * a_rsb), lds_addr_of(&sm3.A[st][(wave * SP_PW + q) * 1024]));
        } else {
            glds16_s(tokq[q] * (uint32_t)b_rsb + coB[q], sp_uniform(baseB + f * SPK * b_rsb), lds_addr_of(&sm3.B[st][(wave * SP_PW + q) * 1024]));
        }
    };
    if constexpr (NA == 3) sp_tn_mainloop3<TERMS>(sm3, acc, nch, wm, wn, lane, dma);
    else sp_tn_mainloop<TERMS>(sm3, acc, nch, wm, wn, lane, dma);
    const float inv = 1.f / (a_sc[0] * b_sc[0]);
    float* so = slab + (int64_t)sp * Mi * N + (int64_t)i0 * N + n0;
    const int cols_valid = N - n0;
    auto emit = [&](int row, int col, const f32x4& v) {
        if (col >= cols_valid) return;
        *reinterpret_cast<f32x4*>(so + (int64_t)row * N + col) = v * inv;
    };
    if (i0 + SPM <= Mi) sp_epilogue_rows<true>(acc, sm, wave, wm, wn, lane, SPM, emit);
    else sp_epilogue_rows<false>(acc, sm, wave, wm, wn, lane, Mi - i0, emit);
}

static inline int sp_tn_splits(int64_t T, int Mi, int N) {
    const int tiles = ((Mi + SPM - 1) / SPM) * ((N + SPN - 1) / SPN);
    return splits_for(T, tiles, 256);   // one workgroup per CU
}
static inline int64_t sp_tn_tps(int64_t T, int S) {
    const int64_t tps = (T + S - 1) / S;
    return ((tps + SPK - 1) / SPK) * SPK;
}

}  // namespace mdl

using namespace mdl;

/* Builds the split image of X [rows, K] (row stride ldx floats): img rows of rsb bytes (>= 4 K), `pad_rows` all-zero rows appended
 * (the B operand of mdl_split_gemm_tn needs 32).  scale (device float[2]) receives {scale, absmax}. */
extern "C" int mdl_split_image(const float* X, int64_t ldx, int64_t rows, int K, void* img, int64_t rsb, int64_t pad_rows,
                               float* scale, void* stream) {
    if (!X || !img || !scale || rows < 0 || K < 32 || (K % 32) || ldx < K || (ldx & 3) || rsb < (int64_t)K * 4 || (rsb & 15) || pad_rows < 0)
        return MDL_E_ARG;
    if (!host_aligned16(X) || !host_aligned16(img)) return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scale, 0, 2 * sizeof(float), s);
    if (e != hipSuccess) return (int)e;
    if (pad_rows > 0) {
        e = hipMemsetAsync((char*)img + rows * rsb, 0, (size_t)(pad_rows * rsb), s);
        if (e != hipSuccess) return (int)e;
    }
    int rc = sp_launch_absmax(X, ldx, rows, K, scale + 1, s);
    if (rc) return rc;
    rc = sp_launch_scale(scale, s);
    if (rc) return rc;
    if (rows > 0) {
        const int64_t n = rows * (K / 8);
        hipLaunchKernelGGL(sp_convert_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, X, ldx, rows, K, (char*)img, rsb,
                           (const float*)scale);
        MDL_LAUNCH_CHECK();
    }
    return MDL_OK;
}

/* Row-scaled split image of X [rows, K]: row r scaled by its own power of two s_r (max |s_r X[r][:]| in [2^13, 2^14); 1 for a zero row);
 * row_inv (device float[rows]) receives 1 / s_r, scale = {1, max |X|}.  As the A operand of mdl_split_gemm_nt pass a_row_mul = row_inv. */
extern "C" int mdl_split_image_rows(const float* X, int64_t ldx, int64_t rows, int K, void* img, int64_t rsb, int64_t pad_rows,
                                    float* row_inv, float* scale, void* stream) {
    if (!X || !img || !row_inv || rows < 0 || K < 32 || (K % 32) || ldx < K || (ldx & 3) || rsb < (int64_t)K * 4 || (rsb & 15) ||
        pad_rows < 0)
        return MDL_E_ARG;
    if (!host_aligned16(X) || !host_aligned16(img)) return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (scale) {   // scale == NULL: the caller supplies the constant {1, .} itself -- the whole image is ONE launch (weights: every forward)
        e = hipMemsetAsync(scale, 0, 2 * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
    }
    if (pad_rows > 0) {
        e = hipMemsetAsync((char*)img + rows * rsb, 0, (size_t)(pad_rows * rsb), s);
        if (e != hipSuccess) return (int)e;
    }
    if (rows > 0) {
        int64_t nb = (rows + 3) / 4;
        if (nb > 8192) nb = 8192;
        if (K <= 512) {
            int64_t n4 = (rows + 15) / 16;   // 4 waves x 4 rows per block iteration
            if (n4 > 4096) n4 = 4096;
            hipLaunchKernelGGL((sp_image_rows_reg_kernel<1, 4>), dim3((unsigned)n4), dim3(256), 0, s, X, ldx, rows, K, (char*)img, rsb, row_inv,
                               scale ? scale + 1 : nullptr);
        } else if (K <= 1024) {
            int64_t n2 = (rows + 7) / 8;
            if (n2 > 4096) n2 = 4096;
            hipLaunchKernelGGL((sp_image_rows_reg_kernel<2, 2>), dim3((unsigned)n2), dim3(256), 0, s, X, ldx, rows, K, (char*)img, rsb, row_inv,
                               scale ? scale + 1 : nullptr);
        } else {
            hipLaunchKernelGGL(sp_image_rows_kernel, dim3((unsigned)nb), dim3(256), 0, s, X, ldx, rows, K, (char*)img, rsb, row_inv,
                               scale ? scale + 1 : nullptr);
        }
        MDL_LAUNCH_CHECK();
    }
    if (scale) {
        hipLaunchKernelGGL(sp_unit_scale_kernel, dim3(1), dim3(1), 0, s, scale);
        MDL_LAUNCH_CHECK();
    }
    return MDL_OK;
}

/* C [M, N] (row stride ldc floats) (+)= sum_k A[m][k] B[n][k] (+ bias[n]) on the split images A (M rows) and B (N rows) of K columns;
 * a_scale / b_scale: device floats, the images' scales.  absmax_out (device float, may be NULL): atomically raised to max |C|
 * (the caller zeroes it).  N % 4 == 0, K % 32 == 0. */
extern "C" int mdl_split_tile_absmax(const float* X, int64_t ldx, int64_t rows, int K, float* gate, float* chunk_max, void* stream) {
    if (!X || !gate || rows < 0 || K < 4 || (K & 3) || ldx < K || (ldx & 3)) return MDL_E_ARG;
    if (!host_aligned16(X)) return MDL_E_ALIGN;
    if (rows == 0) return MDL_OK;
    hipLaunchKernelGGL(sp_tile_absmax_kernel, dim3((unsigned)((rows + SPM - 1) / SPM)), dim3(256), 0, (hipStream_t)stream, X, ldx, rows, K, gate,
                       chunk_max);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

static int sp_gemm_nt_impl(const void* A, int64_t a_rsb, const float* a_scale, const void* B, int64_t b_rsb, const float* b_scale,
                           float* C, int64_t ldc, int64_t M, int N, int K, const float* bias, int accumulate, float* absmax_out,
                           const float* row_gate, const float* a_row_mul, const float* b_col_mul, int terms, void* stream,
                           const float* group_bias, const int32_t* row_group) {
    if ((group_bias == nullptr) != (row_group == nullptr)) return MDL_E_ARG;
    if (group_bias && !host_aligned16(group_bias)) return MDL_E_ALIGN;
    if (terms != 2 && terms != 3) return MDL_E_ARG;
    if (row_gate && (!accumulate || bias)) return MDL_E_ARG;   // skipping a tile is only the identity when it would add zeros
    if (!A || !B || !C || !a_scale || !b_scale || M < 0 || N < 4 || (N & 3) || K < 32 || (K % 32) || ldc < N || (ldc & 3)) return MDL_E_ARG;
    if (a_rsb < (int64_t)K * 4 || b_rsb < (int64_t)K * 4 || (a_rsb & 15) || (b_rsb & 15)) return MDL_E_ARG;
    if (!host_aligned16(A) || !host_aligned16(B) || !host_aligned16(C) || !host_aligned16(bias)) return MDL_E_ALIGN;
    if (M == 0) return MDL_OK;
    if (N <= SPT_N && !row_gate && !group_bias && M > SPM) {   // narrow output (the token_projector): the 512 x 128 tile
        const int64_t tt = (M + SPT_M - 1) / SPT_M;
        if (tt > 0x7fffffff || a_rsb * SPT_M > 0x7fffffff || b_rsb * SPT_N > 0x7fffffff) return MDL_E_UNSUPPORTED;
        hipLaunchKernelGGL(terms == 2 ? sp_nt_tall_kernel<2> : sp_nt_tall_kernel<3>, dim3((unsigned)tt), dim3(SP_THREADS), 0, (hipStream_t)stream,
                           (const char*)A, a_rsb, a_scale, (const char*)B, b_rsb, b_scale, C, ldc, M, N, K / 32, (int)tt, bias, accumulate,
                           absmax_out, a_row_mul, b_col_mul);
        MDL_LAUNCH_CHECK();
        return MDL_OK;
    }
    const int64_t tiles = ((M + SPM - 1) / SPM) * ((N + SPN - 1) / SPN);
    if (tiles > 0x7fffffff || a_rsb * SPM > 0x7fffffff || b_rsb * SPN > 0x7fffffff) return MDL_E_UNSUPPORTED;
    const bool na3 = sp_nt_stages() == 3;   // sp_nt_mainloop3 (split_engine.hpp)
    hipLaunchKernelGGL(terms == 2 ? (na3 ? sp_nt_kernel<2, 3> : sp_nt_kernel<2, 2>) : (na3 ? sp_nt_kernel<3, 3> : sp_nt_kernel<3, 2>),
                       dim3((unsigned)tiles), dim3(SP_THREADS), 0, (hipStream_t)stream,
                       (const char*)A, a_rsb, a_scale, (const char*)B, b_rsb, b_scale, C, ldc, M, N, K / 32, (int)tiles, bias, accumulate,
                       absmax_out, row_gate, a_row_mul, b_col_mul, group_bias, row_group);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}
extern "C" int mdl_split_gemm_nt(const void* A, int64_t a_rsb, const float* a_scale, const void* B, int64_t b_rsb, const float* b_scale,
                                 float* C, int64_t ldc, int64_t M, int N, int K, const float* bias, int accumulate, float* absmax_out,
                                 const float* row_gate, const float* a_row_mul, const float* b_col_mul, int terms, void* stream) {
    return sp_gemm_nt_impl(A, a_rsb, a_scale, B, b_rsb, b_scale, C, ldc, M, N, K, bias, accumulate, absmax_out, row_gate, a_row_mul, b_col_mul,
                           terms, stream, nullptr, nullptr);
}
/* mdl_split_gemm_nt with a bias row per GROUP of rows: C[m][:] += group_bias[row_group[m]][:] (group_bias [G][N] contiguous fp32,
 * row_group int32 [M] with values in [0, G)). */
extern "C" int mdl_split_gemm_nt_group_bias(const void* A, int64_t a_rsb, const float* a_scale, const void* B, int64_t b_rsb,
                                            const float* b_scale, float* C, int64_t ldc, int64_t M, int N, int K, const float* bias,
                                            const float* a_row_mul, const float* b_col_mul, const float* group_bias,
                                            const int32_t* row_group, int terms, void* stream) {
    if (!group_bias || !row_group) return MDL_E_ARG;
    return sp_gemm_nt_impl(A, a_rsb, a_scale, B, b_rsb, b_scale, C, ldc, M, N, K, bias, 0, nullptr, nullptr, a_row_mul, b_col_mul, terms, stream,
                           group_bias, row_group);
}

extern "C" int64_t mdl_split_gemm_tn_ws_bytes(int64_t T, int Mi, int N) {
    if (T < 0 || Mi < 32 || N < 32) return MDL_E_ARG;
    const int S = sp_tn_splits(T, Mi, N);
    // slabs | chunk lists [S][tps / 32] + counts [S]
    return (((int64_t)S * Mi * N * 4 + 15) & ~(int64_t)15) + (int64_t)S * (sp_tn_tps(T, S) / SPK + 1) * 4 + 64;
}

/* out [N][Mi] (contiguous: a Linear's dW with A = image(X), B = image(dY)) = sum_t B[t][n] A[t][m] over the T rows of the two
 * token-major images (Mi / N columns).  The B image must be followed by >= 32 all-zero rows.  Mi % 32 == 0, N % 32 == 0. */
extern "C" int mdl_split_gemm_tn(const void* A, int64_t a_rsb, const float* a_scale, int Mi, const void* B, int64_t b_rsb,
                                 const float* b_scale, int N, float* out, int64_t T, const float* b_chunk_max, void* ws, int terms,
                                 void* stream) {
    if (terms != 2 && terms != 3) return MDL_E_ARG;
    if (!A || !B || !out || !ws || !a_scale || !b_scale || T < 0 || Mi < 32 || (Mi % 32) || N < 32 || (N % 32)) return MDL_E_ARG;
    if (a_rsb < (int64_t)Mi * 4 || b_rsb < (int64_t)N * 4 || (a_rsb & 15) || (b_rsb & 15) || a_rsb * 32 > 0x7fffffff || b_rsb * 32 > 0x7fffffff)
        return MDL_E_ARG;
    if (!host_aligned16(A) || !host_aligned16(B) || !host_aligned16(out) || !host_aligned16(ws)) return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int S = sp_tn_splits(T, Mi, N);
    const int64_t tps = sp_tn_tps(T, S);
    const int tiles = ((Mi + SPM - 1) / SPM) * ((N + SPN - 1) / SPN) * S;
    const int stride = (int)(tps / SPK);
    int32_t* list = nullptr;
    int32_t* count = nullptr;
    if (b_chunk_max && stride <= SP_MAX_LIST && T > 0) {   // skip the 32-row chunks in which the tensor B is the image of is all zero
        list = (int32_t*)((char*)ws + (((int64_t)S * Mi * N * 4 + 15) & ~(int64_t)15));
        count = list + (int64_t)S * stride;
        hipLaunchKernelGGL(sp_chunk_list_kernel, dim3(S), dim3(256), 0, s, b_chunk_max, T, tps, list, count, stride);
        MDL_LAUNCH_CHECK();
    }
    const bool tn3 = sp_tn_stages() == 3;
    hipLaunchKernelGGL(terms == 2 ? (tn3 ? sp_tn_kernel<2, 3> : sp_tn_kernel<2, 2>) : (tn3 ? sp_tn_kernel<3, 3> : sp_tn_kernel<3, 2>), dim3(tiles), dim3(SP_THREADS), 0, s, (const char*)A, a_rsb, a_scale, Mi, (const char*)B, b_rsb, b_scale, N,
                       (float*)ws, T, tps, tiles, (const int32_t*)list, (const int32_t*)count, stride);
    MDL_LAUNCH_CHECK();
    return lin_launch_reduce((const float*)ws, out, Mi, N, S, s);
}
