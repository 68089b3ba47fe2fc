// infonce.hip -- L1: symmetric InfoNCE with in-batch negatives, batched over S problems (stains).
//
// Replaces InfoNCE.info_nce, negative_keys=None branch (reference madeleine/utils/loss.py:92,111-127):
//     q_hat,p_hat = F.normalize(q), F.normalize(p)          (x / max(|x|, 1e-12), loss.py:132)
//     logits = q_hat p_hat^T / T ;  loss = CE(logits, diag)  [+ CE(logits^T, diag), weights 1/2,1/2]
// The reference calls it once per stain (trainer.py:33); here all stains of a step go through one set
// of launches.  At temperature 0.001 the logits are cosines x 1000, so the similarity contraction runs on
// the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) and every logsumexp is max-subtracted.
//
// Two paths: the staged kernels (normalise | logits | lse | loss ; coef | grad | norm-bwd: the default, and the faster one) and, for
// Kp <= 256 and D <= 512, the opt-in fused kernels at the end of this file (one launch forward, one backward; same bits).
// Workspace (floats; Kp = Kmax rounded up to 32):
//   Qn,Pn [S,Kp,D] | rq,rp [S,Kp] | Z [S,Kp,Kp] = logits | m0,l0,m1,l1 [S,Kp] | coef [S,Kp,Kp] | dQn,dPn [S,Kp,D]
// (m, l) = (max, log sum exp(z - max)) per row (0) / column (1), kept SEPARATE like torch's log_softmax: at
// T = 0.001 the logits are O(100) and lse - z_ii is O(1e-6), below fp32 resolution at 100.
#include <cstdlib>

#include "common.hpp"

namespace mdl {

struct NceWs {
    float *Qn, *Pn, *rq, *rp, *Z, *m0, *l0, *m1, *l1, *coef, *dQn, *dPn, *rl;
    int* done;
    int Kp;
};
static inline int nce_kp(int Kmax) { return ((Kmax + 31) / 32) * 32; }
static inline NceWs nce_ws(void* ws, int S, int Kmax, int D) {
    NceWs w;
    w.Kp = nce_kp(Kmax);
    const int64_t rows = (int64_t)S * w.Kp;
    float* p = (float*)ws;
    w.Qn = p; p += rows * D;
    w.Pn = p; p += rows * D;
    w.rq = p; p += rows;
    w.rp = p; p += rows;
    w.Z = p; p += rows * w.Kp;
    w.m0 = p; p += rows;
    w.l0 = p; p += rows;
    w.m1 = p; p += rows;
    w.l1 = p; p += rows;
    w.coef = p; p += rows * w.Kp;
    w.dQn = p; p += rows * D;
    w.dPn = p; p += rows * D;
    w.rl = p; p += 2 * rows;          // fused forward: per-row loss of each direction
    w.done = (int*)p;                 // [S] workgroups of the stain that have finished
    return w;
}

// Sum of squares of one row in the order BOTH paths use: lane (l32, kh) of a wave owns row l32 and the k's kh*4 .. kh*4+3 of every
// group of 8 (the streaming pattern of the similarity product); one fmaf chain per half, the two halves added.  p = row + kh * 4.
__device__ __forceinline__ float nce_row_ss(const float* __restrict__ p, int D, bool live) {
    float ss = 0.f;
    if (live)
        for (int k0 = 0; k0 < D; k0 += 8) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(p + k0);
#pragma unroll
            for (int i = 0; i < 4; ++i) ss = fmaf(v[i], v[i], ss);
        }
    return ss + __shfl_xor(ss, 32, 64);
}
__device__ __forceinline__ float nce_inv_norm(float ss, bool live) { return live ? 1.f / fmaxf(sqrtf(ss), 1e-12f) : 0.f; }

// one wave per 32 rows of Q (which = 0) or P (1): normalised rows + reciprocal clamped norms; padded rows -> 0
__global__ __launch_bounds__(64) void nce_normalize_kernel(const float* __restrict__ Q, const float* __restrict__ P,
                                                           const int32_t* __restrict__ cnt, float* __restrict__ Qn,
                                                           float* __restrict__ Pn, float* __restrict__ rq,
                                                           float* __restrict__ rp, int Kmax, int Kp, int D) {
    const int nrb = Kp / 32, rb = blockIdx.x % nrb, s = blockIdx.x / nrb, which = blockIdx.y;
    const int lane = threadIdx.x, l32 = lane & 31, kh = lane >> 5, r = rb * 32 + l32;
    const bool live = r < cnt[s] && r < Kmax;
    const float* __restrict__ src = (which ? P : Q) + ((int64_t)s * Kmax + (r < Kmax ? r : Kmax - 1)) * D + kh * 4;
    float* __restrict__ dst = (which ? Pn : Qn) + ((int64_t)s * Kp + r) * D + kh * 4;
    const float inv = nce_inv_norm(nce_row_ss(src, D, live), live);
    for (int k0 = 0; k0 < D; k0 += 8) {
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live) v = *reinterpret_cast<const f32x4*>(src + k0) * inv;
        *reinterpret_cast<f32x4*>(dst + k0) = v;
    }
    if (kh == 0) ((which ? rp : rq) + (int64_t)s * Kp)[r] = inv;
}

// one wave per 32x32 block of Z = Qn Pn^T * inv_T.  Each lane streams its own row 16 B at a time; the 8
// k's of a step are split (k0..k0+3 | k0+4..k0+7) between the two half-waves of the MFMA's K=2.
__global__ __launch_bounds__(64) void nce_logits_kernel(const float* __restrict__ Qn, const float* __restrict__ Pn,
                                                        float* __restrict__ Z, int Kp, int D, float inv_T) {
    const int lane = threadIdx.x, l32 = lane & 31, kh = lane >> 5;
    const int j0 = blockIdx.x * 32, i0 = blockIdx.y * 32, s = blockIdx.z;
    const float* __restrict__ a = Qn + ((int64_t)s * Kp + i0 + l32) * D + kh * 4;
    const float* __restrict__ b = Pn + ((int64_t)s * Kp + j0 + l32) * D + kh * 4;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < D; k0 += 8) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(a + k0);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(b + k0);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
    }
    float* __restrict__ z = Z + ((int64_t)s * Kp + i0) * Kp + j0 + l32;
#pragma unroll
    for (int r = 0; r < 16; ++r) z[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * kh) * Kp] = acc[r] * inv_T;
}

// one wave per (row, dir): dir 0 -> lse over columns of row i; dir 1 -> lse over rows of column i
__global__ __launch_bounds__(64) void nce_lse_kernel(const float* __restrict__ Z, const int32_t* __restrict__ cnt,
                                                     float* __restrict__ m0, float* __restrict__ l0,
                                                     float* __restrict__ m1, float* __restrict__ l1, int Kp) {
    const int i = blockIdx.x % Kp, s = blockIdx.x / Kp, dir = blockIdx.y, lane = threadIdx.x;
    const int k = cnt[s];
    float* om = (dir ? m1 : m0) + (int64_t)s * Kp + i;
    float* ol = (dir ? l1 : l0) + (int64_t)s * Kp + i;
    if (i >= k) {
        if (lane == 0) {
            *om = 0.f;
            *ol = 0.f;
        }
        return;
    }
    const float* __restrict__ z = Z + (int64_t)s * Kp * Kp + (dir ? (int64_t)i : (int64_t)i * Kp);
    const int64_t stride = dir ? Kp : 1;
    float mx = -INFINITY;
    for (int j = lane; j < k; j += 64) mx = fmaxf(mx, z[j * stride]);
    mx = wave_max(mx);
    float sm = 0.f;
    for (int j = lane; j < k; j += 64) sm += expf(z[j * stride] - mx);
    sm = wave_sum(sm);
    if (lane == 0) {
        *om = mx;
        *ol = logf(sm);
    }
}

__global__ __launch_bounds__(256) void nce_loss_kernel(const float* __restrict__ Z, const float* __restrict__ m0,
                                                       const float* __restrict__ l0, const float* __restrict__ m1,
                                                       const float* __restrict__ l1, const int32_t* __restrict__ cnt,
                                                       float* __restrict__ loss, float* __restrict__ row_loss, int Kmax, int Kp,
                                                       int symmetric) {
    __shared__ float red[4];
    const int s = blockIdx.x, tid = threadIdx.x, k = cnt[s];
    float v = 0.f;
    for (int i = tid; i < k; i += 256) {
        const float d = Z[((int64_t)s * Kp + i) * Kp + i];
        const int64_t o = (int64_t)s * Kp + i;
        const float r0 = l0[o] - (d - m0[o]);  // -log_softmax(z)[i] = log sum exp(z - max) - (z_ii - max)
        const float r = symmetric ? 0.5f * r0 + 0.5f * (l1[o] - (d - m1[o])) : r0;
        if (row_loss) row_loss[(int64_t)s * Kmax + i] = r;   // reduction='none' (loss.py:58: F.cross_entropy(..., reduction))
        v += r;
    }
    if (row_loss)
        for (int i = k + tid; i < Kmax; i += 256) row_loss[(int64_t)s * Kmax + i] = 0.f;
    v = wave_sum(v);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    if (tid == 0) loss[s] = (k > 0) ? (red[0] + red[1] + red[2] + red[3]) / (float)k : 0.f;
}

// coef = dL/d(cosine_ij) = 1/T (w0 g_i softmax_row + w1 g_j softmax_col - (w0 g_i + w1 g_j) delta_ij); zero outside [0,k)^2.
// g_i = d_loss / k (mean reduction) or, with d_row != NULL, the upstream gradient of row i's own loss (reduction 'none').
__global__ void nce_coef_kernel(const float* __restrict__ Z, const float* __restrict__ m0, const float* __restrict__ l0,
                                const float* __restrict__ m1, const float* __restrict__ l1,
                                const int32_t* __restrict__ cnt,
                                const float* __restrict__ d_loss, const float* __restrict__ d_row, int Kmax,
                                float* __restrict__ coef, int Kp, float inv_T, int symmetric) {
    const int s = blockIdx.y;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)Kp * Kp) return;
    const int i = (int)(e / Kp), j = (int)(e % Kp), k = cnt[s];
    float c = 0.f;
    if (i < k && j < k) {
        const float z = Z[(int64_t)s * Kp * Kp + e];
        // (softmax - delta) FIRST: for a saturated softmax (T = 0.001) p_ii -> 1 and the subtraction is exact (Sterbenz); scaling by
        // the loss weights afterwards keeps the cancellation out of the rounding of the products
        const float dlt = (i == j) ? 1.f : 0.f;
        const float a = expf((z - m0[(int64_t)s * Kp + i]) - l0[(int64_t)s * Kp + i]) - dlt;
        const float b = symmetric ? expf((z - m1[(int64_t)s * Kp + j]) - l1[(int64_t)s * Kp + j]) - dlt : 0.f;
        const float w0 = symmetric ? 0.5f : 1.f, w1 = symmetric ? 0.5f : 0.f;
        if (d_row) c = (w0 * d_row[(int64_t)s * Kmax + i] * a + w1 * d_row[(int64_t)s * Kmax + j] * b) * inv_T;
        else c = (w0 * a + w1 * b) * d_loss[s] * inv_T / (float)k;
    }
    coef[(int64_t)s * Kp * Kp + e] = c;
}

// dXn block [32 rows x 32 feature cols]: dir 0: dQn = coef Pn ; dir 1: dPn = coef^T Qn.  One wave per block.
__global__ __launch_bounds__(64) void nce_grad_kernel(const float* __restrict__ coef, const float* __restrict__ Qn,
                                                      const float* __restrict__ Pn, float* __restrict__ dQn,
                                                      float* __restrict__ dPn, int Kp, int D, int S) {
    const int lane = threadIdx.x, l32 = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.x * 32, i0 = blockIdx.y * 32, s = blockIdx.z % S, dir = blockIdx.z / S;
    const float* __restrict__ cf = coef + (int64_t)s * Kp * Kp;
    const float* __restrict__ Y = (dir ? Qn : Pn) + (int64_t)s * Kp * D + n0 + l32;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < Kp; k0 += 8) {
        float av[4], bv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int kk = k0 + kh * 4 + i;
            av[i] = dir ? cf[(int64_t)kk * Kp + i0 + l32] : cf[(int64_t)(i0 + l32) * Kp + kk];
            bv[i] = Y[(int64_t)kk * D];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
    }
    float* __restrict__ o = (dir ? dPn : dQn) + ((int64_t)s * Kp + i0) * D + n0 + l32;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(int64_t)((r & 3) + 8 * (r >> 2) + 4 * kh) * D] = acc[r];
}

// backward of F.normalize: dX = rn (dXn - Xn <Xn,dXn>)  (clamped rows, |x| < 1e-12: dX = rn dXn); pad rows -> 0
__global__ __launch_bounds__(64) void nce_norm_bwd_kernel(const float* __restrict__ Qn, const float* __restrict__ Pn,
                                                          const float* __restrict__ rq, const float* __restrict__ rp,
                                                          const float* __restrict__ dQn, const float* __restrict__ dPn,
                                                          const int32_t* __restrict__ cnt, float* __restrict__ dQ,
                                                          float* __restrict__ dP, int Kmax, int Kp, int D) {
    const int r = blockIdx.x % Kmax, s = blockIdx.x / Kmax, which = blockIdx.y, lane = threadIdx.x;
    const float* __restrict__ xn = (which ? Pn : Qn) + ((int64_t)s * Kp + r) * D;
    const float* __restrict__ g = (which ? dPn : dQn) + ((int64_t)s * Kp + r) * D;
    float* __restrict__ o = (which ? dP : dQ) + ((int64_t)s * Kmax + r) * D;
    if (r >= cnt[s]) {
        for (int k = lane; k < D; k += 64) o[k] = 0.f;
        return;
    }
    const float rn = (which ? rp : rq)[(int64_t)s * Kp + r];
    float dot = 0.f;
    for (int k = lane; k < D; k += 64) dot += xn[k] * g[k];
    dot = wave_sum(dot);
    if (rn >= 0.99e12f) dot = 0.f;  // norm was clamped at 1e-12: normalize() is then linear in x
    for (int k = lane; k < D; k += 64) o[k] = rn * (g[k] - xn[k] * dot);
}


// ================================================================================================
// Fused path (Kp <= 256, D <= 512 -- every configuration of the training loop: k <= 256 cases per global batch): ONE launch forward,
// ONE launch backward.  No normalised copies: the similarity product runs on the raw rows and is scaled by the two reciprocal norms,
// which fall out of the same streaming pass (each lane owns a row of its operand).
//   forward  workgroup = 32 rows of Z (dir 0) or of Z^T (dir 1: the column log-sum-exp of the symmetric loss is the row one of the
//            transposed problem; its strip is recomputed, 67 MFLOP at k = 256), 4 waves x 2 column blocks of 32; the strip goes through
//            LDS for the row reductions and the coalesced store of Z; the last workgroup of a stain to finish reduces the row losses
//            (fixed order: deterministic).
//   backward workgroup = 32 rows of dQ (dir 0) or dP (dir 1): coefficient strip (softmax - delta) in LDS -> MFMA against the
//            normalised rows of the other side -> normalize() backward on the strip in LDS.
// ================================================================================================
constexpr int NCE_FK = 256, NCE_FD = 512;

__device__ __forceinline__ int nce_acc_row(int r, int kh) { return (r & 3) + 8 * (r >> 2) + 4 * kh; }

__global__ __launch_bounds__(256) void nce_fused_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ P,
                                                            const int32_t* __restrict__ cnt, float* __restrict__ rq,
                                                            float* __restrict__ rp, float* __restrict__ Z, float* __restrict__ m0,
                                                            float* __restrict__ l0, float* __restrict__ m1, float* __restrict__ l1,
                                                            float* __restrict__ rl, int* __restrict__ done, float* __restrict__ loss,
                                                            float* __restrict__ row_loss, int S, int Kmax, int Kp, int D, float inv_T,
                                                            int symmetric) {
    __shared__ float strip[32][NCE_FK + 1];
    __shared__ float red[4];
    __shared__ int last;
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * 32, dir = blockIdx.y, s = blockIdx.z;
    const int k = cnt[s] < Kmax ? cnt[s] : Kmax;
    const float* __restrict__ X = (dir ? P : Q) + (int64_t)s * Kmax * D;
    const float* __restrict__ Y = (dir ? Q : P) + (int64_t)s * Kmax * D;
    // each lane streams its own row twice: sum of squares (nce_row_ss, the staged path's order), then the similarity product on the
    // rows normalised on the fly (x * 1/|x|: the products the staged path stores); the second pass hits the cache
    const int ra = i0 + l32;
    const bool live_a = ra < k;
    const float* __restrict__ a = X + (int64_t)(ra < Kmax ? ra : Kmax - 1) * D + kh * 4;
    const float rna = nce_inv_norm(nce_row_ss(a, D, live_a), live_a);   // x / max(|x|, 1e-12), loss.py:132
    if (wave == 0 && kh == 0) (dir ? rp : rq)[(int64_t)s * Kp + ra] = rna;
#pragma unroll 1
    for (int jb = wave; jb * 32 < Kp; jb += 4) {
        const int rb = jb * 32 + l32;
        const bool live_b = rb < k;
        const float* __restrict__ b = Y + (int64_t)(rb < Kmax ? rb : Kmax - 1) * D + kh * 4;
        const float rnb = nce_inv_norm(nce_row_ss(b, D, live_b), live_b);
        if (blockIdx.x == 0 && kh == 0) (dir ? rq : rp)[(int64_t)s * Kp + rb] = rnb;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int k0 = 0; k0 < D; k0 += 8) {
            f32x4 av = f32x4{0.f, 0.f, 0.f, 0.f}, bv = av;   // padding rows may hold anything
            if (live_a) av = *reinterpret_cast<const f32x4*>(a + k0) * rna;
            if (live_b) bv = *reinterpret_cast<const f32x4*>(b + k0) * rnb;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) strip[nce_acc_row(r, kh)][jb * 32 + l32] = acc[r] * inv_T;
    }
    __syncthreads();
    // row reductions (max-subtracted, (m, l) kept separate), diagonal, coalesced store of the strip
    float* __restrict__ om = (dir ? m1 : m0) + (int64_t)s * Kp;
    float* __restrict__ ol = (dir ? l1 : l0) + (int64_t)s * Kp;
#pragma unroll 1
    for (int rr = 0; rr < 8; ++rr) {
        const int row = wave * 8 + rr, i = i0 + row;
        float m = 0.f, l = 0.f, r0 = 0.f;
        if (i < k) {   // wave-uniform
            float mx = -INFINITY;
            for (int j = lane; j < k; j += 64) mx = fmaxf(mx, strip[row][j]);
            mx = wave_max(mx);
            float sm = 0.f;
            for (int j = lane; j < k; j += 64) sm += expf(strip[row][j] - mx);
            sm = wave_sum(sm);
            m = mx;
            l = logf(sm);
            r0 = l - (strip[row][i] - m);   // -log_softmax(z)[i]
        }
        if (lane == 0) {
            om[i] = m;
            ol[i] = l;
            rl[((int64_t)dir * S + s) * Kp + i] = r0;
        }
        if (dir == 0)
            for (int j = lane; j < Kp; j += 64) Z[((int64_t)s * Kp + i) * Kp + j] = strip[row][j];
    }
    // the last workgroup of this stain reduces the row losses
    __threadfence();
    __syncthreads();
    if (tid == 0) last = atomicAdd(&done[s], 1) == (int)(gridDim.x * gridDim.y) - 1;
    __syncthreads();
    if (!last) return;
    __threadfence();
    const volatile float* r0p = rl + (int64_t)s * Kp;
    const volatile float* r1p = rl + ((int64_t)S + s) * Kp;
    float v = 0.f;
    for (int i = tid; i < k; i += 256) {
        const float r = symmetric ? 0.5f * r0p[i] + 0.5f * r1p[i] : r0p[i];
        if (row_loss) row_loss[(int64_t)s * Kmax + i] = r;   // reduction='none' (loss.py:58)
        v += r;
    }
    if (row_loss)
        for (int i = k + tid; i < Kmax; i += 256) row_loss[(int64_t)s * Kmax + i] = 0.f;
    v = wave_sum(v);
    if (lane == 0) red[wave] = v;
    __syncthreads();
    if (tid == 0) {
        loss[s] = (k > 0) ? (red[0] + red[1] + red[2] + red[3]) / (float)k : 0.f;
        done[s] = 0;   // ready for the next call on this workspace
    }
}

__global__ __launch_bounds__(256) void nce_fused_bwd_kernel(const float* __restrict__ Q, const float* __restrict__ P,
                                                            const float* __restrict__ rq, const float* __restrict__ rp,
                                                            const float* __restrict__ Z, const float* __restrict__ m0,
                                                            const float* __restrict__ l0, const float* __restrict__ m1,
                                                            const float* __restrict__ l1, const int32_t* __restrict__ cnt,
                                                            const float* __restrict__ d_loss, const float* __restrict__ d_row,
                                                            float* __restrict__ dQ, float* __restrict__ dP, int Kmax, int Kp, int D,
                                                            float inv_T, int symmetric) {
    __shared__ float cs[32][NCE_FK + 1];
    __shared__ float og[32][NCE_FD + 4];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i0 = blockIdx.x * 32, dir = blockIdx.y, s = blockIdx.z;
    const int k = cnt[s] < Kmax ? cnt[s] : Kmax;
    if (i0 >= Kmax) return;   // no such rows in dQ / dP
    const float* __restrict__ X = (dir ? P : Q) + (int64_t)s * Kmax * D;
    const float* __restrict__ Y = (dir ? Q : P) + (int64_t)s * Kmax * D;
    const float* __restrict__ ry = (dir ? rq : rp) + (int64_t)s * Kp;
    const float* __restrict__ rx = (dir ? rp : rq) + (int64_t)s * Kp;
    float* __restrict__ dX = (dir ? dP : dQ) + (int64_t)s * Kmax * D;
    const int64_t zb = (int64_t)s * Kp * Kp, vb = (int64_t)s * Kp;
    // coefficient strip: cs[rr][kk] = dL/d(cosine) between strip row i0 + rr and row kk of the other side
    const float w0 = symmetric ? 0.5f : 1.f, w1 = symmetric ? 0.5f : 0.f;
    for (int e = tid; e < 32 * Kp; e += 256) {
        const int rr = dir ? (e & 31) : e / Kp, kk = dir ? (e >> 5) : e % Kp;
        const int i = dir ? kk : i0 + rr, j = dir ? i0 + rr : kk;   // (row, column) of Z
        float c = 0.f;
        if (i < k && j < k) {
            const float z = Z[zb + (int64_t)i * Kp + j];
            const float dlt = (i == j) ? 1.f : 0.f;   // (softmax - delta) first: exact for a saturated softmax (see nce_coef_kernel)
            const float a = expf((z - m0[vb + i]) - l0[vb + i]) - dlt;
            const float b = symmetric ? expf((z - m1[vb + j]) - l1[vb + j]) - dlt : 0.f;
            if (d_row) c = (w0 * d_row[(int64_t)s * Kmax + i] * a + w1 * d_row[(int64_t)s * Kmax + j] * b) * inv_T;
            else c = (w0 * a + w1 * b) * d_loss[s] * inv_T / (float)k;
        }
        cs[rr][kk] = c;
    }
    __syncthreads();
    // og[32][D] = cs[32][Kp] . Y[Kp][D]: wave w owns the 32-column blocks w, w + 4, ... (<= 4 at D = 512)
    const int nblk = D / 32;
    f32x16 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
    const int kend = (k + 7) & ~7;   // coefficients are zero from k on
    for (int k0 = 0; k0 < kend; k0 += 8) {
        float av[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) av[i] = cs[l32][k0 + kh * 4 + i];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int nb = wave + 4 * u;
            if (nb < nblk) {   // wave-uniform
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int kk = k0 + kh * 4 + i;
                    const float bv = kk < k ? Y[(int64_t)kk * D + nb * 32 + l32] * ry[kk] : 0.f;   // = the staged path's normalised row
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv, acc[u], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int nb = wave + 4 * u;
        if (nb < nblk)
#pragma unroll
            for (int r = 0; r < 16; ++r) og[nce_acc_row(r, kh)][nb * 32 + l32] = acc[u][r];
    }
    __syncthreads();
    // backward of F.normalize on the strip: dX = rn (dXn - Xn <Xn, dXn>)  (clamped rows: dX = rn dXn); padding rows -> 0.
    // 4 rows per trip: their loads are independent (one row at a time is a chain of L2 latencies)
    constexpr int NC = NCE_FD / 64;
#pragma unroll 1
    for (int g = 0; g < 2; ++g) {
        float xn[4][NC], rn[4], dot[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = wave * 8 + g * 4 + u, i = i0 + row;
            const bool live = i < k;
            rn[u] = live ? rx[i] : 0.f;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int c = lane + 64 * j;
                xn[u][j] = (live && c < D) ? X[(int64_t)i * D + c] * rn[u] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = wave * 8 + g * 4 + u;
            float d = 0.f;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int c = lane + 64 * j;
                if (c < D) d += xn[u][j] * og[row][c];
            }
            d = wave_sum(d);
            dot[u] = rn[u] >= 0.99e12f ? 0.f : d;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = wave * 8 + g * 4 + u, i = i0 + row;
            if (i < Kmax) {   // wave-uniform
                float* __restrict__ o = dX + (int64_t)i * D;
#pragma unroll
                for (int j = 0; j < NC; ++j) {
                    const int c = lane + 64 * j;
                    if (c < D) o[c] = i < k ? rn[u] * (og[row][c] - xn[u][j] * dot[u]) : 0.f;
                }
            }
        }
    }
}

}  // namespace mdl

using namespace mdl;

// The fused kernels are opt-in (MADELEINE_INFONCE_FUSED=1): same bits as the staged ones, but SLOWER at every size measured on MI355X
// (tools/exp_infonce.py, forward / backward in us: S=3 k=32 58 / 39 against 54 / 24; S=4 k=256 97 / 199 against 63 / 61) -- the
// problem is latency-bound, launches on one stream pipeline (~5 us each), and fusing trades 4-16x of the wave-level parallelism for
// them (64 workgroups at k = 256 where the staged gradient product runs 1024 waves).
static bool nce_fused(int Kp, int D) { return Kp <= NCE_FK && D <= NCE_FD && getenv("MADELEINE_INFONCE_FUSED"); }

extern "C" int64_t mdl_infonce_ws_bytes(int S, int Kmax, int D) {
    if (S < 0 || Kmax < 0 || D < 8 || (D % 32)) return MDL_E_ARG;
    const int64_t Kp = nce_kp(Kmax), rows = (int64_t)S * Kp;
    return (4 * rows * D + 8 * rows + 2 * rows * Kp + S) * 4 + 64;
}

extern "C" int mdl_infonce_fwd(const float* Q, const float* P, const int32_t* cnt, float* loss, float* row_loss, int S, int Kmax,
                               int D, float temperature, int symmetric, void* ws, void* stream) {
    if (!Q || !P || !cnt || !loss || !ws) return MDL_E_ARG;
    if (S < 0 || Kmax < 0 || D < 8 || (D % 32) || !(temperature > 0.f)) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    if (S == 0) return MDL_OK;
    hipStream_t st = (hipStream_t)stream;
    const NceWs w = nce_ws(ws, S, Kmax, D);
    const int Kp = w.Kp;
    if (Kp > 0 && nce_fused(Kp, D)) {   // fused: one launch (the workgroup counters start at zero and reset themselves)
        const hipError_t e = hipMemsetAsync(w.done, 0, (size_t)S * sizeof(int), st);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(nce_fused_fwd_kernel, dim3(Kp / 32, symmetric ? 2 : 1, S), dim3(256), 0, st, Q, P, cnt, w.rq, w.rp, w.Z,
                           w.m0, w.l0, w.m1, w.l1, w.rl, w.done, loss, row_loss, S, Kmax, Kp, D, 1.f / temperature, symmetric);
        MDL_LAUNCH_CHECK();
        return MDL_OK;
    }
    if (Kp > 0) {
        hipLaunchKernelGGL(nce_normalize_kernel, dim3(S * (Kp / 32), 2), dim3(64), 0, st, Q, P, cnt, w.Qn, w.Pn, w.rq, w.rp, Kmax,
                           Kp, D);
        MDL_LAUNCH_CHECK();
        hipLaunchKernelGGL(nce_logits_kernel, dim3(Kp / 32, Kp / 32, S), dim3(64), 0, st, w.Qn, w.Pn, w.Z, Kp, D,
                           1.f / temperature);
        MDL_LAUNCH_CHECK();
        hipLaunchKernelGGL(nce_lse_kernel, dim3(S * Kp, symmetric ? 2 : 1), dim3(64), 0, st, w.Z, cnt, w.m0, w.l0, w.m1, w.l1, Kp);
        MDL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(nce_loss_kernel, dim3(S), dim3(256), 0, st, w.Z, w.m0, w.l0, w.m1, w.l1, cnt, loss, row_loss, Kmax, Kp, symmetric);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

extern "C" int mdl_infonce_bwd(const float* Q, const float* P, const float* d_loss, const float* d_row_loss, const int32_t* cnt,
                               float* dQ, float* dP, int S, int Kmax, int D, float temperature, int symmetric, void* ws, void* stream) {
    if (!Q || !P || (!d_loss && !d_row_loss) || !cnt || !dQ || !dP || !ws) return MDL_E_ARG;
    if (S < 0 || Kmax < 0 || D < 8 || (D % 32) || !(temperature > 0.f)) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    if (S == 0 || Kmax == 0) return MDL_OK;
    hipStream_t st = (hipStream_t)stream;
    const NceWs w = nce_ws(ws, S, Kmax, D);
    const int Kp = w.Kp;
    if (nce_fused(Kp, D)) {   // the forward ran fused: no normalised copies in ws
        hipLaunchKernelGGL(nce_fused_bwd_kernel, dim3(Kp / 32, 2, S), dim3(256), 0, st, Q, P, w.rq, w.rp, w.Z, w.m0, w.l0, w.m1, w.l1,
                           cnt, d_loss, d_row_loss, dQ, dP, Kmax, Kp, D, 1.f / temperature, symmetric);
        MDL_LAUNCH_CHECK();
        return MDL_OK;
    }
    hipLaunchKernelGGL(nce_coef_kernel, dim3((Kp * Kp + 255) / 256, S), dim3(256), 0, st, w.Z, w.m0, w.l0, w.m1, w.l1, cnt,
                       d_loss, d_row_loss, Kmax, w.coef, Kp, 1.f / temperature, symmetric);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(nce_grad_kernel, dim3(D / 32, Kp / 32, S * 2), dim3(64), 0, st, w.coef, w.Qn, w.Pn, w.dQn, w.dPn, Kp, D,
                       S);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(nce_norm_bwd_kernel, dim3(S * Kmax, 2), dim3(64), 0, st, w.Qn, w.Pn, w.rq, w.rp, w.dQn, w.dPn, cnt, dQ,
                       dP, Kmax, Kp, D);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}
