// linear_fp32.hip -- the Linear layers of the pre-attention MLP, forward and backward, as
// exact-fp32 contractions on v_mfma_f32_32x32x2_f32 with the gate kernels' tile engine (abmil_gate.hip): 128 x 256
// output tile, BK = 16, 4 waves (2 x 2) x (2 x 4) MFMA tiles, every operand by LDS-DMA, two LDS stages, one barrier per
// chunk.  Replaces nn.Linear of reference madeleine/models/Model.py:351, :355, :359 (bias-free: the bias and its
// gradient live in the fused LayerNorm kernel, preattn_act.hip) and its autograd:
//     forward : Y[t, n]  = sum_k X[t, k]  W[n, k]     = lin_nn(A = X,  B = W^T [K][N] (transposed copy, <= 4 MiB))
//     dX      : dX[t, k] = sum_n dY[t, n] W[n, k]     = lin_nn(A = dY, B = W   [N][K] as stored)
//     dW      : dW[n, k] = sum_t dY[t, n] X[t, k]     = lin_tn(A = X, B = dY), split over tokens, slabs reduced + transposed
// Shapes: N % 256 == 0 (lin_nn output width / lin_tn slab width), contraction length % 16 == 0, row strides % 4 == 0.
#include "gate_common.hpp"

namespace mdl {

constexpr int LBM = 128, LBN = 256, LBK = 16;

__device__ __forceinline__ void lin_zero(f32x16 (&acc)[2][4]) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
}

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
// instruction (conflict-free ds_read_b32) -- exactly the gate forward's staging.  Consecutive workgroups of an XCD share
// the token tile (xcd_remap), so A is fetched from HBM once per XCD.
__global__ __launch_bounds__(256, 2) void lin_nn_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ B,
                                                        float* __restrict__ C, int64_t ldc, int64_t T, int Nc, int Kc,
                                                        int n_tiles) {
    __shared__ __attribute__((aligned(16))) struct {
        float A[2][LBM * LBK];
        float B[2][LBK][LBN];
    } sm;
    const int tid = threadIdx.x, lane = tid & 63, wm = (tid >> 6) >> 1, wn = (tid >> 6) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncol = Nc / LBN;
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int nt = lid % ncol;
    const int64_t t0 = (int64_t)(lid / ncol) * LBM;
    const int n0 = nt * LBN;

    const float* srcA[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sl = (wave * 2 + q) * 64 + lane, row = sl >> 2, kq = (sl & 3) ^ ((row >> 2) & 3);
        int64_t t = t0 + row;
        if (t > T - 1) t = T - 1;  // rows past T re-read row T-1 (discarded in the epilogue)
        srcA[q] = A + t * lda + kq * 4;
    }
    const float* __restrict__ srcB = B + n0 + lane * 4;
    auto issue = [&](int st, int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16(srcA[q] + k0, &sm.A[st][(wave * 2 + q) * 256]);
#pragma unroll
        for (int q = 0; q < 4; ++q) glds16(srcB + (int64_t)(k0 + wave * 4 + q) * Nc, &sm.B[st][wave * 4 + q][0]);
    };
    const int l32 = lane & 31, kh = lane >> 5;
    const int colb[4] = {wn * 128, wn * 128 + 32, wn * 128 + 64, wn * 128 + 96};
    int offA[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = wm * 64 + rt * 32 + l32;
        offA[rt] = r * LBK + ((kh ^ ((r >> 2) & 3)) << 2);
    }
    f32x16 acc[2][4];
    lin_zero(acc);
    const int nch = Kc / LBK;
    issue(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nch; ++ch) {
        const int st = ch & 1;
        if (ch + 1 < nch) issue(st ^ 1, (ch + 1) * LBK);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x4 fa[2];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) fa[rt] = *reinterpret_cast<const f32x4*>(&sm.A[st][offA[rt] ^ (g << 3)]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float fb[4];
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) fb[ct] = sm.B[st][8 * g + 4 * kh + e][colb[ct] + l32];
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int rt = m & 1, ct = m >> 1;
                    acc[rt][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[rt][e], fb[ct], acc[rt][ct], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t t = t0 + wm * 64 + rt * 32 + acc_row(r, lane);
            if (t < T) {
                float* __restrict__ o = C + t * ldc + n0 + l32;
#pragma unroll
                for (int ct = 0; ct < 4; ++ct) o[colb[ct]] = acc[rt][ct][r];
            }
        }
}

// slab[sp][k][n] = sum_{t in split sp} X[t, k] dY[t, n];  both operands are K(= t)-major in memory: natural LDS images.
// Rows t >= T read from `zrow` (zeros); columns k >= Kx of the last k-tile read column 0 (discarded).
__global__ __launch_bounds__(256, 2) void lin_tn_kernel(const float* __restrict__ X, int64_t ldx, int Kx,
                                                        const float* __restrict__ dY, int64_t ldy, int Ny,
                                                        float* __restrict__ slab, const float* __restrict__ zrow, int64_t T,
                                                        int64_t tok_per_split, int n_splits, int n_tiles) {
    __shared__ __attribute__((aligned(16))) struct {
        float A[2][LBK][LBM];
        float B[2][LBK][LBN];
    } sm;
    const int tid = threadIdx.x, lane = tid & 63, wm = (tid >> 6) >> 1, wn = (tid >> 6) & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nkt = (Kx + LBM - 1) / LBM, nnt = Ny / LBN;
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int kt = lid % nkt, ntile = (lid / nkt) % nnt, sp = lid / (nkt * nnt);
    const int k0 = kt * LBM, n0 = ntile * LBN;
    const int64_t ts = (int64_t)sp * tok_per_split;
    int64_t te = ts + tok_per_split;
    if (te > T) te = T;

    // A: two 512-B row segments of X per wave instruction; B: one 1-KiB row segment of dY per wave instruction
    const int ka = k0 + (lane & 31) * 4;
    const int64_t offXa = (ka < Kx) ? ka : 0;
    const int64_t offYb = n0 + lane * 4;
    auto issue = [&](int st, int64_t tb) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int r0 = (wave * 2 + q) * 2;
            const int64_t t = tb + r0 + (lane >> 5);
            glds16(t < T ? X + t * ldx + offXa : zrow + (lane & 31) * 4, &sm.A[st][r0][0]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t t = tb + wave * 4 + q;
            glds16(t < T ? dY + t * ldy + offYb : zrow + lane * 4, &sm.B[st][wave * 4 + q][0]);
        }
    };
    f32x16 acc[2][4];
    lin_zero(acc);
    const int colb[4] = {wn * 128, wn * 128 + 32, wn * 128 + 64, wn * 128 + 96};
    const int l32 = lane & 31, kh = lane >> 5;
    const int64_t nch = (te > ts) ? (te - ts + LBK - 1) / LBK : 0;
    if (nch > 0) issue(0, ts);
    __syncthreads();
    for (int64_t ch = 0; ch < nch; ++ch) {
        const int st = (int)(ch & 1);
        if (ch + 1 < nch) issue(st ^ 1, ts + (ch + 1) * LBK);
#pragma unroll
        for (int kk = 0; kk < LBK / 2; ++kk) {
            const int k = kk * 2 + kh;
            const float a0 = sm.A[st][k][wm * 64 + l32];
            const float a1 = sm.A[st][k][wm * 64 + 32 + l32];
            float b[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) b[ct] = sm.B[st][k][colb[ct] + l32];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b[ct], acc[0][ct], 0, 0, 0);
                acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b[ct], acc[1][ct], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    float* __restrict__ so = slab + (int64_t)sp * Kx * Ny;
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
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += slab[((int64_t)s * K + k) * N + nb + tx];
        tile[ty + i * 8][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) dW[(int64_t)(nb + ty + i * 8) * K + kb + tx] = tile[tx][ty + i * 8];
}

static inline int lin_splits(int64_t T, int N, int K) { return splits_for(T, ((K + LBM - 1) / LBM) * (N / LBN)); }
static inline int64_t lin_tps(int64_t T, int S) {
    const int64_t tps = (T + S - 1) / S;
    return ((tps + LBK - 1) / LBK) * LBK;
}
static inline int64_t up16b(int64_t b) { return (b + 15) & ~(int64_t)15; }

}  // namespace mdl

using namespace mdl;

static int lin_check(int64_t T, int N, int K) {
    if (T < 0 || N < 1 || K < 1) return MDL_E_ARG;
    if ((N % LBN) || (K % 32)) return MDL_E_UNSUPPORTED;
    return MDL_OK;
}

extern "C" int64_t mdl_linear_fwd_ws_bytes(int64_t T, int N, int K) {
    const int rc = lin_check(T, N, K);
    if (rc) return rc;
    return (int64_t)N * K * 4 + 64;  // W^T [K][N]
}

extern "C" int mdl_linear_fwd(const float* X, int64_t ldx, const float* W, float* Y, int64_t ldy, int64_t T, int N, int K, void* ws,
                              void* stream) {
    const int rc = lin_check(T, N, K);
    if (rc) return rc;
    if (!X || !W || !Y || !ws || ldx < K || ldy < N || (ldx & 3) || (ldy & 3)) return MDL_E_ARG;
    if (!host_aligned16(X) || !host_aligned16(W) || !host_aligned16(Y) || !host_aligned16(ws)) return MDL_E_ALIGN;
    if (T == 0) return MDL_OK;
    hipStream_t s = (hipStream_t)stream;
    float* WT = (float*)ws;
    hipLaunchKernelGGL(lin_transpose_kernel, dim3(K / 32, N / 32), dim3(256), 0, s, W, N, K, WT);
    MDL_LAUNCH_CHECK();
    const int64_t tiles = ((T + LBM - 1) / LBM) * (N / LBN);
    if (tiles > 0x7fffffff) return MDL_E_UNSUPPORTED;
    hipLaunchKernelGGL(lin_nn_kernel, dim3((unsigned)tiles), dim3(256), 0, s, X, ldx, (const float*)WT, Y, ldy, T, N, K, (int)tiles);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

extern "C" int64_t mdl_linear_bwd_ws_bytes(int64_t T, int N, int K) {
    const int rc = lin_check(T, N, K);
    if (rc) return rc;
    const int S = lin_splits(T, N, K);
    return up16b((int64_t)S * K * N * 4) + up16b((int64_t)(N > K ? N : K) * 4 + 1024) + 64;  // slabs | zero row
}

/* dX may be NULL (first layer: the bags need no gradient).  dX requires K % 256 == 0. */
extern "C" int mdl_linear_bwd(const float* X, int64_t ldx, const float* W, const float* dY, int64_t ldy, float* dX, int64_t lddx,
                              float* dW, int64_t T, int N, int K, void* ws, void* stream) {
    const int rc = lin_check(T, N, K);
    if (rc) return rc;
    if (!X || !W || !dY || !dW || !ws || ldx < K || ldy < N || (ldx & 3) || (ldy & 3)) return MDL_E_ARG;
    if (dX && ((K % LBN) || lddx < K || (lddx & 3))) return MDL_E_UNSUPPORTED;
    if (!host_aligned16(X) || !host_aligned16(W) || !host_aligned16(dY) || !host_aligned16(dW) || !host_aligned16(ws) ||
        !host_aligned16(dX))
        return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const int S = lin_splits(T, N, K);
    const int64_t tps = lin_tps(T, S);
    float* slab = (float*)ws;
    float* zrow = (float*)((char*)ws + up16b((int64_t)S * K * N * 4));
    {
        const hipError_t e = hipMemsetAsync(zrow, 0, (size_t)(N > K ? N : K) * 4 + 1024, s);
        if (e != hipSuccess) return (int)e;
    }
    if (dX && T > 0) {  // dX[t, k] = sum_n dY[t, n] W[n][k]: W as stored is the K(= n)-major B operand
        const int64_t tiles = ((T + LBM - 1) / LBM) * (K / LBN);
        if (tiles > 0x7fffffff) return MDL_E_UNSUPPORTED;
        hipLaunchKernelGGL(lin_nn_kernel, dim3((unsigned)tiles), dim3(256), 0, s, dY, ldy, W, dX, lddx, T, K, N, (int)tiles);
        MDL_LAUNCH_CHECK();
    }
    const int nkt = (K + LBM - 1) / LBM, nnt = N / LBN;
    const int tiles = nkt * nnt * S;
    hipLaunchKernelGGL(lin_tn_kernel, dim3(tiles), dim3(256), 0, s, X, ldx, K, dY, ldy, N, slab, (const float*)zrow, T, tps, S, tiles);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(lin_reduce_kernel, dim3(N / 32, K / 32), dim3(256), 0, s, (const float*)slab, dW, K, N, S);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}
