// linear_fp32.hip -- the Linear layers of the pre-attention MLP, forward and backward, as
// exact-fp32 contractions on v_mfma_f32_32x32x2_f32 with the gate kernels' tile engine (abmil_gate.hip): 128 x 256
// output tile, BK = 16, 4 waves (2 x 2) x (2 x 4) MFMA tiles, every operand by LDS-DMA, two LDS stages, one barrier per
// chunk.  Replaces nn.Linear of reference madeleine/models/Model.py:351, :355, :359 (bias-free: the bias and its
// gradient live in the fused LayerNorm kernel, preattn_act.hip) and its autograd:
//     forward : Y[t, n]  = sum_k X[t, k]  W[n, k]     = lin_nn(A = X,  B = W^T [K][N] (transposed copy, <= 4 MiB))
//     dX      : dX[t, k] = sum_n dY[t, n] W[n, k]     = lin_nn(A = dY, B = W   [N][K] as stored)
//     dW      : dW[n, k] = sum_t dY[t, n] X[t, k]     = lin_tn(A = X, B = dY), split over tokens, slabs reduced + transposed
// Shapes: N % 256 == 0 with K % 32 == 0, or N % 128 == 0 with K % 256 == 0; row strides % 4 == 0.  N % 256 == 0 runs the 128 x 256
// tile (its dX, whose output width is K, on the ragged-column variant when K % 256 != 0); N = 128 (mod 256) -- the
// token_projector Linear(2048, 128), reference Model.py:140 -- the "tall" 256 x 128 geometry of the same engine (waves
// stacked in M) and, for dW, the role-swapped lin_tn (dY as the 128-wide operand, the slab is dW itself).  T <= 256 rows --
// the slide projector Linear(2048, 512) on the pooled embeddings, Model.py:145 -- runs a plain LDS-tiled fp32 FMA kernel
// (three strided calls for Y, dX, dW): 0.03 % of the step's FLOPs, not worth a matrix-core tile.  Optional bias / dbias.
#include "tile_engine.hpp"

namespace mdl {

constexpr int LBM = 128, LBN = 256, LBK = 16;

// out[c][r] = in[r][c]   (32 x 32 LDS tiles; R, C multiples of 32)
__global__ __launch_bounds__(256) void lin_transpose_kernel(const float* __restrict__ in, int R, int C, float* __restrict__ out) {
    __shared__ float tile[32][33];
    const int rb = blockIdx.y * 32, cb = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + i * 8][tx] = in[(int64_t)(rb + ty + i * 8) * C + cb + tx];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) out[(int64_t)(cb + ty + i * 8) * R + rb + tx] = tile[tx][ty + i * 8];
}

// C[t, n] = sum_k A[t, k] B[k][n];  A [T, Kc] rows (stride lda), B [Kc][Nc] row-major, C [T, Nc] (stride ldc).
// A lands as the XOR-swizzled row image (ds_read_b128 fragments, k-pair permutation), B as one 1-KiB row per wave
// instruction (conflict-free ds_read_b32) -- exactly the gate forward's staging (tile_engine.hpp).  Consecutive workgroups
// of an XCD share the token tile (xcd_remap), so A is fetched from HBM once per XCD.
// RAGN: Nc is not a multiple of 256 (Nc % 4 == 0): the last column tile fetches column n0 for its out-of-range columns (their
// accumulators are discarded) and masks its stores -- the dX of a Linear whose input width is not a multiple of 256 (config 5's
// 768 + 32 = 800 channels, reference Model.py:132 makes that input require a gradient through embedding.weight).
template <bool RAGN>
__global__ __launch_bounds__(256) void lin_nn_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                     float* __restrict__ C, int64_t ldc, int64_t T, int Nc, int Kc,
                                                     int n_tiles, const float* __restrict__ bias) {
    __shared__ __attribute__((aligned(16))) TileSmem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: SGPR addressing downstream
    const int wm = wave >> 1, wn = wave & 1;
    const int ncol = RAGN ? (Nc + LBN - 1) / LBN : Nc / LBN;
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int nt = lid % ncol;
    const int64_t t0 = (int64_t)(lid / ncol) * LBM;
    const int n0 = nt * LBN;

    const char* baseA = reinterpret_cast<const char*>(A + t0 * lda);
    const char* baseB = reinterpret_cast<const char*>(B + n0 + (int64_t)(wave * 4) * Nc);
    uint32_t voA[2];
    rows_voff(voA, T - t0, lda, wave, lane);
    const uint32_t voB = (RAGN && n0 + lane * 4 >= Nc) ? 0u : lane * 16;
    const int64_t rowB = (int64_t)Nc * 4;
    const int colb[4] = {wn * 128, wn * 128 + 32, wn * 128 + 64, wn * 128 + 96};
    f32x16 acc[2][4];
    tile_zero(acc);
    tile_loop_nn(acc, sm, Kc / LBK, wm, colb, lane, [&](int st, int f, int piece) {
        if (piece == 0) rows_issue(baseA + (int64_t)f * (LBK * 4), voA, sm.A[st], wave);
        else krows_issue2(baseB + ((int64_t)f * LBK + (piece - 1) * 2) * rowB, rowB, voB, sm.B[st], wave, (piece - 1) * 2);
    });
    char* cb = reinterpret_cast<char*>(C + t0 * ldc + n0);
    const uint32_t ldc4 = (uint32_t)ldc * 4u;
    auto emit = [&](int row_u, int rl, int lane_col, const f32x4& v, int) {
        if (RAGN && n0 + lane_col >= Nc) return;
        f32x4 r = v;
        if (bias) r += *reinterpret_cast<const f32x4*>(bias + n0 + lane_col);
        *reinterpret_cast<f32x4*>(cb + (int64_t)row_u * ldc4 + ((uint32_t)rl * ldc4 + (uint32_t)lane_col * 4u)) = r;
    };
    if (t0 + LBM <= T) tile_epilogue_rows<true>(acc, sm, wave, wm, colb, lane, LBM, emit);
    else tile_epilogue_rows<false>(acc, sm, wave, wm, colb, lane, (int)(T - t0), emit);
}

// The same contraction with the "tall" geometry: 256 rows x 128 columns per workgroup, wave w owns rows 64 w .. 64 w + 63 and
// all 128 columns (still 2 x 4 MFMA tiles per wave, 24 KiB per stage).  For Nc = 128 (mod 256).
__global__ __launch_bounds__(256) void lin_nn_tall_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                          float* __restrict__ C, int64_t ldc, int64_t T, int Nc, int Kc,
                                                          int n_tiles, const float* __restrict__ bias) {
    __shared__ __attribute__((aligned(16))) TileSmem sm;
    constexpr int TM = 256, TN = 128;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncol = Nc / TN;
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int nt = lid % ncol;
    const int64_t t0 = (int64_t)(lid / ncol) * TM;
    const int n0 = nt * TN;
    float* Ab = reinterpret_cast<float*>(&sm);          // [2][256 rows][16 k]
    float* Bb = Ab + 2 * TM * LBK;                      // [2][16 k][128]

    const char* baseA = reinterpret_cast<const char*>(A + t0 * lda);
    uint32_t voA[4];   // wave w fills LDS slots [(4w+q)*64, +64), q = 0..3 (rows 16 (4w+q) ..)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int sl = (wave * 4 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        int64_t rr = row;
        if (rr > T - t0 - 1) rr = T - t0 - 1;
        voA[q] = (uint32_t)(rr * lda * 4 + kq * 16);
    }
    // B: two 512-B row segments per instruction (lanes 0-31 row r, lanes 32-63 row r+1); wave w fills k-rows 4w .. 4w+3
    const char* baseB = reinterpret_cast<const char*>(B + n0 + (int64_t)(wave * 4) * Nc);
    const int64_t rowB = (int64_t)Nc * 4;
    const uint32_t voB = (uint32_t)((lane >> 5) * rowB) + (lane & 31) * 16;
    const int colb[4] = {0, 32, 64, 96};
    f32x16 acc[2][4];
    tile_zero(acc);
    tile_loop_nn_g<TN>(acc, Ab, TM * LBK, Bb, Kc / LBK, wave, colb, lane, [&](int st, int f, int piece) {
        if (piece < 2) {
            const char* a = baseA + (int64_t)f * (LBK * 4);
            glds16_s(voA[piece * 2 + 0], a, lds_addr_of(Ab + st * (TM * LBK) + (wave * 4 + piece * 2 + 0) * 256));
            glds16_s(voA[piece * 2 + 1], a, lds_addr_of(Ab + st * (TM * LBK) + (wave * 4 + piece * 2 + 1) * 256));
        } else {
            const char* b = baseB + (int64_t)f * LBK * rowB;
            glds16_s(voB, b, lds_addr_of(Bb + (st * LBK + wave * 4 + 0) * TN));
            glds16_s(voB, b + 2 * rowB, lds_addr_of(Bb + (st * LBK + wave * 4 + 2) * TN));
        }
    });
    char* cb = reinterpret_cast<char*>(C + t0 * ldc + n0);
    const uint32_t ldc4 = (uint32_t)ldc * 4u;
    auto emit = [&](int row_u, int rl, int lane_col, const f32x4& v, int) {
        f32x4 r = v;
        if (bias) r += *reinterpret_cast<const f32x4*>(bias + n0 + lane_col);
        *reinterpret_cast<f32x4*>(cb + (int64_t)row_u * ldc4 + ((uint32_t)rl * ldc4 + (uint32_t)lane_col * 4u)) = r;
    };
    if (t0 + TM <= T) tile_epilogue_rows<true>(acc, sm, wave, wave, colb, lane, TM, emit);
    else tile_epilogue_rows<false>(acc, sm, wave, wave, colb, lane, (int)(T - t0), emit);
}

// slab[sp][k][n] = sum_{t in split sp} X[t, k] dY[t, n];  both operands are K(= t)-major in memory: natural LDS images.
// dY rows t >= T read from `zrow` (zeros); X rows t >= T re-read row T-1 (their products with the zero rows vanish);
// columns k >= Kx of the last k-tile read column 0 (discarded).
__global__ __launch_bounds__(256) void lin_tn_kernel(const float* __restrict__ X, int64_t ldx, int Kx,
                                                     const float* __restrict__ dY, int64_t ldy, int Ny,
                                                     float* __restrict__ slab, const float* __restrict__ zrow, int64_t T,
                                                     int64_t tok_per_split, int n_splits, int n_tiles) {
    __shared__ __attribute__((aligned(16))) TileSmem sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int nkt = (Kx + LBM - 1) / LBM, nnt = Ny / LBN;
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int kt = lid % nkt, ntile = (lid / nkt) % nnt, sp = lid / (nkt * nnt);
    const int k0 = kt * LBM, n0 = ntile * LBN;
    const int64_t ts = (int64_t)sp * tok_per_split;
    int64_t te = ts + tok_per_split;
    if (te > T) te = T;

    // A: two 512-B row segments of X per wave instruction (lanes 0-31 row r, lanes 32-63 row r+1); B: one 1-KiB row of dY
    const int ka = k0 + (lane & 31) * 4;
    const uint32_t colA = (uint32_t)((ka < Kx) ? ka : 0) * 4u;
    const uint32_t ldx4 = (uint32_t)ldx * 4u;
    const char* Xb = reinterpret_cast<const char*>(X);
    const char* Yb = reinterpret_cast<const char*>(dY + n0);
    const char* Zb = reinterpret_cast<const char*>(zrow);
    const uint32_t voB = lane * 16;
    float (*As)[TBK][TBM] = reinterpret_cast<float (*)[TBK][TBM]>(&sm.A[0][0]);
    f32x16 acc[2][4];
    tile_zero(acc);
    const int colb[4] = {wn * 128, wn * 128 + 32, wn * 128 + 64, wn * 128 + 96};
    const int64_t nch = (te > ts) ? (te - ts + LBK - 1) / LBK : 0;
    tile_loop_tn(acc, sm, nch, wm, colb, lane, [&](int st, int64_t f, int piece) {
        const int64_t tb = ts + f * LBK;
        if (piece == 0) {
            const int64_t left = T - 1 - tb;   // >= 0: last valid row relative to this chunk
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int r0 = (wave * 2 + q) * 2;
                uint32_t r = r0 + (lane >> 5);
                if (left < 2 * 8) r = r < (uint32_t)left ? r : (uint32_t)left;   // uniform branch: only the chunk at the end of X
                glds16_s(r * ldx4 + colA, Xb + tb * (int64_t)ldx4, lds_addr_of(&As[st][r0][0]));
            }
        } else {
#pragma unroll
            for (int q = (piece - 1) * 2; q < (piece - 1) * 2 + 2; ++q) {
                const int64_t t = tb + wave * 4 + q;
                glds16_s(voB, t < T ? Yb + t * ldy * 4 : Zb, lds_addr_of(&sm.B[st][wave * 4 + q][0]));
            }
        }
    });
    float* __restrict__ so = slab + (int64_t)sp * Kx * Ny;
    const int l32 = lane & 31;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kr = k0 + wm * 64 + rt * 32 + acc_row(r, lane);
            if (kr < Kx) {
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) so[(int64_t)kr * Ny + n0 + colb[ct] + l32] = acc[rt][ct][r];
            }
        }
}

// dW[n][k] = sum_s slab[s][k][n]   (32 x 32 LDS transpose; K, N multiples of 32)
__global__ __launch_bounds__(256) void lin_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dW, int K, int N,
                                                         int S) {
    __shared__ float tile[32][33];
    const int kb = blockIdx.y * 32, nb = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = kb + ty + i * 8;
        // 8 slab reads in flight per thread (a dependent one-load-per-iteration loop ran this pass at ~2 TB/s); the additions stay
        // in slab order
        const float* __restrict__ src = slab + (int64_t)k * N + nb + tx;
        const int64_t st = (int64_t)K * N;
        float v = 0.f;
        int sp = 0;
        for (; sp + 8 <= S; sp += 8) {
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
#pragma unroll
    for (int i = 0; i < 4; ++i) dW[(int64_t)(nb + ty + i * 8) * K + kb + tx] = tile[tx][ty + i * 8];
}

// out[r][c] = sum_s slab[s][r][c]   (no transpose: the role-swapped lin_tn already produced the [N][K] layout)
__global__ __launch_bounds__(256) void lin_reduce_plain_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t n,
                                                               int S) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4 v = *reinterpret_cast<const f32x4*>(slab + i);
    int s = 1;
    for (; s + 8 <= S; s += 8) {   // 8 reads in flight, additions in slab order
        f32x4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const f32x4*>(slab + (int64_t)(s + u) * n + i);
#pragma unroll
        for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; s < S; ++s) v += *reinterpret_cast<const f32x4*>(slab + (int64_t)s * n + i);
    *reinterpret_cast<f32x4*>(out + i) = v;
}

// column sums of dY [T, N] (bias gradient): part[b][n] over row blocks, then ln_reduce-style final sum.  N % 4 == 0.
__global__ __launch_bounds__(256) void lin_colsum_part_kernel(const float* __restrict__ dY, int64_t ldy, int64_t T, int N,
                                                              float* __restrict__ part, int64_t rows_per_block) {
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > T) r1 = T;
    for (int c4 = threadIdx.x; c4 < N / 4; c4 += 256) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int64_t r = r0; r < r1; ++r) s += *reinterpret_cast<const f32x4*>(dY + r * ldy + c4 * 4);
        *reinterpret_cast<f32x4*>(part + (int64_t)blockIdx.x * N + c4 * 4) = s;
    }
}
__global__ __launch_bounds__(256) void lin_colsum_final_kernel(const float* __restrict__ part, int nb, int N, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += part[(int64_t)b * N + c];
    out[c] = s;
}

// Small-M products (M <= 256 rows): C[m, n] = sum_k A[m sam + k sak] B[n sbn + k sbk] (+ bias[n]), 64 x 64 tile, 16-deep
// chunks through LDS, 4 x 4 outputs per thread, plain fp32 FMAs.  The K range is split over blockIdx.z (the projector's
// 64 x 512 x 2048 product would otherwise be 8 workgroups of 128 dependent global->LDS round trips); partials go to
// part[z][M][N] and lin_small_reduce_kernel adds them in z order (deterministic) together with the bias.
__global__ __launch_bounds__(256) void lin_small_kernel(const float* __restrict__ A, int64_t sam, int64_t sak,
                                                        const float* __restrict__ B, int64_t sbn, int64_t sbk,
                                                        float* __restrict__ part, int M, int N, int K, int k_per_split) {
    __shared__ float As[16][65], Bs[16][65];
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64, tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int kb = blockIdx.z * k_per_split;
    int ke = kb + k_per_split;
    if (ke > K) ke = K;
    float acc[4][4] = {};
    for (int k0 = kb; k0 < ke; k0 += 16) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256;
            // consecutive threads walk the unit-stride axis of each operand (coalesced either way)
            const int ra = (sak == 1) ? (e >> 4) : (e & 63), ka = (sak == 1) ? (e & 15) : (e >> 6);
            const int rb = (sbk == 1) ? (e >> 4) : (e & 63), kq = (sbk == 1) ? (e & 15) : (e >> 6);
            As[ka][ra] = (k0 + ka < ke && m0 + ra < M) ? A[(int64_t)(m0 + ra) * sam + (int64_t)(k0 + ka) * sak] : 0.f;
            Bs[kq][rb] = (k0 + kq < ke && n0 + rb < N) ? B[(int64_t)(n0 + rb) * sbn + (int64_t)(k0 + kq) * sbk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = As[kk][ty * 4 + i];
                b[i] = Bs[kk][tx * 4 + i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    float* __restrict__ out = part + (int64_t)blockIdx.z * M * N;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < N) out[(int64_t)m * N + n] = acc[i][j];
        }
    }
}
__global__ __launch_bounds__(256) void lin_small_reduce_kernel(const float* __restrict__ part, int SK, float* __restrict__ C,
                                                               int64_t ldc, int M, int N, const float* __restrict__ bias) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float v = bias ? bias[n] : 0.f;
    for (int z = 0; z < SK; ++z) v += part[(int64_t)z * M * N + i];
    C[(int64_t)m * ldc + n] = v;
}
static inline int lin_small_splits(int K) {
    int sk = K / 128;
    return sk < 1 ? 1 : (sk > 16 ? 16 : sk);
}
static inline int64_t lin_small_ws(int M, int N, int K) { return (int64_t)lin_small_splits(K) * M * N * 4; }

constexpr int LIN_SMALL_T = 256;   // at most this many rows: lin_small_kernel
constexpr int LIN_COLSUM_BLOCKS = 512;

// shared with split_gemm.hip: dW[n][k] = sum_s slab[s][k][n]
int lin_launch_reduce(const float* slab, float* dW, int K, int N, int S, hipStream_t s) {
    hipLaunchKernelGGL(lin_reduce_kernel, dim3(N / 32, K / 32), dim3(256), 0, s, slab, dW, K, N, S);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

static inline int lin_splits(int64_t T, int N, int K) { return splits_for(T, ((K + LBM - 1) / LBM) * (N / LBN)); }
static inline int64_t lin_tps(int64_t T, int S) {
    const int64_t tps = (T + S - 1) / S;
    return ((tps + LBK - 1) / LBK) * LBK;
}
static inline int64_t up16b(int64_t b) { return (b + 15) & ~(int64_t)15; }

}  // namespace mdl

using namespace mdl;

// Supported geometries (mirrored by functional.linear_supported):
//   T <= 256 rows          : N % 4 == 0, K % 4 == 0 (fp32 FMA kernel)
//   N % 256 == 0           : K % 32 == 0 (dX through the ragged-column tile when K is not a multiple of 256)
//   N % 256 == 128         : K % 256 == 0 (tall tile, role-swapped dW)
static int lin_check(int64_t T, int N, int K) {
    if (T < 0 || N < 1 || K < 1) return MDL_E_ARG;
    if (T <= LIN_SMALL_T) return ((N % 4) || (K % 4)) ? MDL_E_UNSUPPORTED : MDL_OK;
    if ((N % LBN) == 0) return (K % 32) ? MDL_E_UNSUPPORTED : MDL_OK;
    if ((N % 128) == 0) return (K % LBN) ? MDL_E_UNSUPPORTED : MDL_OK;
    return MDL_E_UNSUPPORTED;
}
static inline bool lin_wide(int N) { return (N % LBN) == 0; }
static inline int lin_splits_any(int64_t T, int N, int K) {
    return lin_wide(N) ? lin_splits(T, N, K) : splits_for(T, ((N + LBM - 1) / LBM) * (K / LBN));   // swapped roles for N = 128
}

static int lin_small_launch(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbn, int64_t sbk, float* C, int64_t ldc,
                            int M, int N, int K, const float* bias, float* part, hipStream_t s) {
    if (M <= 0 || N <= 0) return MDL_OK;
    const int SK = lin_small_splits(K);
    const int kps = ((K + SK - 1) / SK + 15) / 16 * 16;
    hipLaunchKernelGGL(lin_small_kernel, dim3((N + 63) / 64, (M + 63) / 64, SK), dim3(256), 0, s, A, sam, sak, B, sbn, sbk, part, M, N, K,
                       kps);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(lin_small_reduce_kernel, dim3((unsigned)(((int64_t)M * N + 255) / 256)), dim3(256), 0, s, (const float*)part, SK, C,
                       ldc, M, N, bias);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

extern "C" int64_t mdl_linear_fwd_ws_bytes(int64_t T, int N, int K) {
    const int rc = lin_check(T, N, K);
    if (rc) return rc;
    if (T <= LIN_SMALL_T) return lin_small_ws((int)T, N, K) + 64;   // split-K partials
    return (int64_t)N * K * 4 + 64;  // W^T [K][N]
}

extern "C" int mdl_linear_fwd(const float* X, int64_t ldx, const float* W, const float* bias, float* Y, int64_t ldy, int64_t T, int N,
                              int K, void* ws, void* stream) {
    const int rc = lin_check(T, N, K);
    if (rc) return rc;
    if (!X || !W || !Y || !ws || ldx < K || ldy < N || (ldx & 3) || (ldy & 3)) return MDL_E_ARG;
    if (!host_aligned16(X) || !host_aligned16(W) || !host_aligned16(Y) || !host_aligned16(ws) || !host_aligned16(bias))
        return MDL_E_ALIGN;
    if (T == 0) return MDL_OK;
    hipStream_t s = (hipStream_t)stream;
    if (T <= LIN_SMALL_T) return lin_small_launch(X, ldx, 1, W, K, 1, Y, ldy, (int)T, N, K, bias, (float*)ws, s);
    float* WT = (float*)ws;
    hipLaunchKernelGGL(lin_transpose_kernel, dim3(K / 32, N / 32), dim3(256), 0, s, W, N, K, WT);
    MDL_LAUNCH_CHECK();
    if (lin_wide(N)) {
        const int64_t tiles = ((T + LBM - 1) / LBM) * (N / LBN);
        if (tiles > 0x7fffffff) return MDL_E_UNSUPPORTED;
        hipLaunchKernelGGL(lin_nn_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, s, X, ldx, (const float*)WT, Y, ldy, T, N, K,
                           (int)tiles, bias);
    } else {
        const int64_t tiles = ((T + 255) / 256) * (N / 128);
        if (tiles > 0x7fffffff) return MDL_E_UNSUPPORTED;
        hipLaunchKernelGGL(lin_nn_tall_kernel, dim3((unsigned)tiles), dim3(256), 0, s, X, ldx, (const float*)WT, Y, ldy, T, N, K,
                           (int)tiles, bias);
    }
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

extern "C" int64_t mdl_linear_bwd_ws_bytes(int64_t T, int N, int K) {
    const int rc = lin_check(T, N, K);
    if (rc) return rc;
    if (T <= LIN_SMALL_T) {   // split-K partials of dX (T x K, contraction N) and dW (N x K, contraction T), used one after the other
        const int64_t a = lin_small_ws((int)T, K, N), b = lin_small_ws(N, K, (int)T);
        return (a > b ? a : b) + 64;
    }
    const int S = lin_splits_any(T, N, K);
    // slabs | zero row | column-sum partials
    return up16b((int64_t)S * K * N * 4) + up16b((int64_t)(N > K ? N : K) * 4 + 1024) + up16b((int64_t)LIN_COLSUM_BLOCKS * N * 4) + 64;
}

/* dX may be NULL (first layer without stain encoding: the bags need no gradient); dbias may be NULL. */
extern "C" int mdl_linear_bwd(const float* X, int64_t ldx, const float* W, const float* dY, int64_t ldy, float* dX, int64_t lddx,
                              float* dW, float* dbias, int64_t T, int N, int K, void* ws, void* stream) {
    const int rc = lin_check(T, N, K);
    if (rc) return rc;
    if (!X || !W || !dY || !dW || !ws || ldx < K || ldy < N || (ldx & 3) || (ldy & 3)) return MDL_E_ARG;
    if (dX && (lddx < K || (lddx & 3))) return MDL_E_ARG;
    if (!host_aligned16(X) || !host_aligned16(W) || !host_aligned16(dY) || !host_aligned16(dW) || !host_aligned16(ws) ||
        !host_aligned16(dX))
        return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (T <= LIN_SMALL_T) {
        int r = MDL_OK;
        if (dX) r = lin_small_launch(dY, ldy, 1, W, 1, K, dX, lddx, (int)T, K, N, nullptr, (float*)ws, s);   // dX[t,k] = sum_n dY[t,n] W[n,k]
        if (r) return r;
        r = lin_small_launch(dY, 1, ldy, X, 1, ldx, dW, K, N, K, (int)T, nullptr, (float*)ws, s);             // dW[n,k] = sum_t dY[t,n] X[t,k]
        if (r) return r;
        if (dbias) {
            hipLaunchKernelGGL(lin_colsum_part_kernel, dim3(1), dim3(256), 0, s, dY, ldy, T, N, dbias, (int64_t)(T > 0 ? T : 1));
            MDL_LAUNCH_CHECK();
        }
        return MDL_OK;
    }
    const bool wide = lin_wide(N);
    const int S = lin_splits_any(T, N, K);
    const int64_t tps = lin_tps(T, S);
    float* slab = (float*)ws;
    float* zrow = (float*)((char*)ws + up16b((int64_t)S * K * N * 4));
    float* cpart = (float*)((char*)zrow + up16b((int64_t)(N > K ? N : K) * 4 + 1024));
    {
        const hipError_t e = hipMemsetAsync(zrow, 0, (size_t)(N > K ? N : K) * 4 + 1024, s);
        if (e != hipSuccess) return (int)e;
    }
    if (dX && T > 0) {  // dX[t, k] = sum_n dY[t, n] W[n][k]: W as stored is the K(= n)-major B operand
        const int64_t tiles = ((T + LBM - 1) / LBM) * ((K + LBN - 1) / LBN);
        if (tiles > 0x7fffffff) return MDL_E_UNSUPPORTED;
        if (K % LBN)
            hipLaunchKernelGGL(lin_nn_kernel<true>, dim3((unsigned)tiles), dim3(256), 0, s, dY, ldy, W, dX, lddx, T, K, N, (int)tiles,
                               (const float*)nullptr);
        else
            hipLaunchKernelGGL(lin_nn_kernel<false>, dim3((unsigned)tiles), dim3(256), 0, s, dY, ldy, W, dX, lddx, T, K, N, (int)tiles,
                               (const float*)nullptr);
        MDL_LAUNCH_CHECK();
    }
    if (wide) {
        const int nkt = (K + LBM - 1) / LBM, nnt = N / LBN;
        const int tiles = nkt * nnt * S;
        hipLaunchKernelGGL(lin_tn_kernel, dim3(tiles), dim3(256), 0, s, X, ldx, K, dY, ldy, N, slab, (const float*)zrow, T, tps, S, tiles);
        MDL_LAUNCH_CHECK();
        hipLaunchKernelGGL(lin_reduce_kernel, dim3(N / 32, K / 32), dim3(256), 0, s, (const float*)slab, dW, K, N, S);
        MDL_LAUNCH_CHECK();
    } else {  // roles swapped: slab[s][n][k] = sum_t dY[t, n] X[t, k] is dW's own layout (128-row operand = dY, 256-wide = X)
        const int nkt = (N + LBM - 1) / LBM, nnt = K / LBN;
        const int tiles = nkt * nnt * S;
        hipLaunchKernelGGL(lin_tn_kernel, dim3(tiles), dim3(256), 0, s, dY, ldy, N, X, ldx, K, slab, (const float*)zrow, T, tps, S, tiles);
        MDL_LAUNCH_CHECK();
        const int64_t n = (int64_t)N * K;
        hipLaunchKernelGGL(lin_reduce_plain_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, (const float*)slab, dW, n, S);
        MDL_LAUNCH_CHECK();
    }
    if (dbias) {
        int nb = (int)((T + 255) / 256);
        if (nb > LIN_COLSUM_BLOCKS) nb = LIN_COLSUM_BLOCKS;
        const int64_t rpb = (T + nb - 1) / nb;
        hipLaunchKernelGGL(lin_colsum_part_kernel, dim3(nb), dim3(256), 0, s, dY, ldy, T, N, cpart, rpb);
        MDL_LAUNCH_CHECK();
        hipLaunchKernelGGL(lin_colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, s, (const float*)cpart, (int)((T + rpb - 1) / rpb), N, dbias);
        MDL_LAUNCH_CHECK();
    }
    return MDL_OK;
}
