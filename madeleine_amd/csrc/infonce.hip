// infonce.hip -- L1: symmetric InfoNCE with in-batch negatives, batched over S problems (stains).
//
// Replaces InfoNCE.info_nce, negative_keys=None branch (reference madeleine/utils/loss.py:92,111-127):
//     q_hat,p_hat = F.normalize(q), F.normalize(p)          (x / max(|x|, 1e-12), loss.py:132)
//     logits = q_hat p_hat^T / T ;  loss = CE(logits, diag)  [+ CE(logits^T, diag), weights 1/2,1/2]
// The reference calls it once per stain (trainer.py:33); here all stains of a step go through one set
// of launches.  At temperature 0.001 the logits are cosines x 1000, so the similarity contraction runs on
// the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) and every logsumexp is max-subtracted.
//
// Workspace (floats; Kp = Kmax rounded up to 32):
//   Qn,Pn [S,Kp,D] | rq,rp [S,Kp] | Z [S,Kp,Kp] = logits | m0,l0,m1,l1 [S,Kp] | coef [S,Kp,Kp] | dQn,dPn [S,Kp,D]
// (m, l) = (max, log sum exp(z - max)) per row (0) / column (1), kept SEPARATE like torch's log_softmax: at
// T = 0.001 the logits are O(100) and lse - z_ii is O(1e-6), below fp32 resolution at 100.
#include "common.hpp"

namespace mdl {

struct NceWs {
    float *Qn, *Pn, *rq, *rp, *Z, *m0, *l0, *m1, *l1, *coef, *dQn, *dPn;
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
    return w;
}

// one wave per (row, which): normalised row + reciprocal clamped norm; padded rows -> 0
__global__ __launch_bounds__(64) void nce_normalize_kernel(const float* __restrict__ Q, const float* __restrict__ P,
                                                           const int32_t* __restrict__ cnt, float* __restrict__ Qn,
                                                           float* __restrict__ Pn, float* __restrict__ rq,
                                                           float* __restrict__ rp, int Kmax, int Kp, int D) {
    const int r = blockIdx.x % Kp, s = blockIdx.x / Kp, which = blockIdx.y, lane = threadIdx.x;
    const float* __restrict__ src = (which ? P : Q) + ((int64_t)s * Kmax + r) * D;
    float* __restrict__ dst = (which ? Pn : Qn) + ((int64_t)s * Kp + r) * D;
    float* __restrict__ rn = (which ? rp : rq) + (int64_t)s * Kp + r;
    const bool live = r < cnt[s] && r < Kmax;
    float ss = 0.f;
    if (live)
        for (int k = lane; k < D; k += 64) ss += src[k] * src[k];
    ss = wave_sum(ss);
    const float inv = live ? 1.f / fmaxf(sqrtf(ss), 1e-12f) : 0.f;
    for (int k = lane; k < D; k += 64) dst[k] = live ? src[k] * inv : 0.f;
    if (lane == 0) *rn = inv;
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

}  // namespace mdl

using namespace mdl;

extern "C" int64_t mdl_infonce_ws_bytes(int S, int Kmax, int D) {
    if (S < 0 || Kmax < 0 || D < 8 || (D % 32)) return MDL_E_ARG;
    const int64_t Kp = nce_kp(Kmax), rows = (int64_t)S * Kp;
    return (4 * rows * D + 6 * rows + 2 * rows * Kp) * 4 + 64;
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
    if (Kp > 0) {
        hipLaunchKernelGGL(nce_normalize_kernel, dim3(S * Kp, 2), dim3(64), 0, st, Q, P, cnt, w.Qn, w.Pn, w.rq, w.rp, Kmax,
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

extern "C" int mdl_infonce_bwd(const float* d_loss, const float* d_row_loss, const int32_t* cnt, float* dQ, float* dP, int S, int Kmax,
                               int D, float temperature, int symmetric, void* ws, void* stream) {
    if ((!d_loss && !d_row_loss) || !cnt || !dQ || !dP || !ws) return MDL_E_ARG;
    if (S < 0 || Kmax < 0 || D < 8 || (D % 32) || !(temperature > 0.f)) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    if (S == 0 || Kmax == 0) return MDL_OK;
    hipStream_t st = (hipStream_t)stream;
    const NceWs w = nce_ws(ws, S, Kmax, D);
    const int Kp = w.Kp;
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
