// got.hip -- G0-G3: Graph Optimal Transport token alignment (forward + hand-written reverse sweep).
//
// Replaces, for k cases with n sub-sampled tokens each (n <= 256, d <= 128):
//   GOT                                (reference madeleine/utils/loss.py:278-302)
//   cost_matrix_batch_torch            (:162-176)   C0[b,i,j] = 1 - <v^_i, q^_j>,  x^ = x / (|x| + 1e-12)
//   global-threshold ReLU              (:288-292)   thr = min + .1 (max - min) over the WHOLE batch tensor
//   IPOT_torch_batch_uniform           (:179-193)   T <- delta * (A.T) * sigma^T, 30 it, beta .5   (Wasserstein)
//   IPOT_distance / batch_trace        (:196-207)   wd_b = sum_ij C_ij T_ij
//   cos_batch_torch                    (:210-233)   thresholded intra costs Cs, Ct
//   GW_torch_batch / GW_distance       (:236-275)   5 x [ C_g = Cst - 2 Cs g Ct^T ; g = IPOT(C_g, beta .1, 20 it) ]
// The reference back-propagates through every unrolled IPOT iteration (no detach: loss.py:204, :245-248
// detaches only the RETURNED gamma).  The backward here replays the iterations in reverse from the stored
// per-iteration plans T_t and scaling vectors (delta_t, sigma_t) -- the same tensors autograd's tape holds.
//
// Organisation: one 256-thread workgroup per case; matrices live in the caller's workspace (L2-resident,
// n*n*4 B <= 256 KiB each), vectors in LDS.  A matrix pass walks rows by wave and columns by lane
// (coalesced), so a row reduction is one 64-lane shuffle reduce and a column reduction is a per-lane
// accumulator merged across the 4 waves through LDS.  Cs and Ct are bitwise symmetric here (the same fma
// chain computes <x_i,x_j> and <x_j,x_i>), so the reference's transposes (:233, :240-247) are identities.
#include "common.hpp"

namespace mdl {

constexpr int GOT_MAXN = 256;
constexpr int GOT_MAXD = 128;
constexpr int WD_ITERS = 30;
constexpr int GW_OUTER = 5;
constexpr int GW_INNER = 20;
constexpr float WD_INV_BETA = 2.0f;    // beta 0.5 (loss.py:179 default, GOT passes only the iteration count :294)
constexpr float GW_INV_BETA = 10.0f;   // lamda = 1e-1 (loss.py:269)
constexpr float THR_BETA = 0.1f;       // loss.py:288, :226

struct GotWs {
    // per-case regions (float offsets from the case base)
    int64_t per_case;
    int64_t oVh, oQh, orV, orQ, oC0, oCs0, oCt0, oC, oWT, oWd, oWs, oCs, oCt, ors, ort, oCg, oGT, oGd, oGs, oP1, oP2, oP3,
        ogT, ogA, ogCs, ogCt, oG, ogC0;
    // global regions (float offsets from ws base)
    int64_t g_ext, g_thr, g_gthr, g_wd, g_gwd, g_cases;
};

__host__ __device__ inline int64_t up4(int64_t x) { return (x + 3) & ~(int64_t)3; }

__host__ __device__ inline GotWs got_layout(int k, int n, int d) {
    GotWs w;
    const int64_t nn = up4((int64_t)n * n), nd = up4((int64_t)n * d), nv = up4(n);
    int64_t o = 0;
    w.oVh = o; o += nd;
    w.oQh = o; o += nd;
    w.orV = o; o += nv;
    w.orQ = o; o += nv;
    w.oC0 = o; o += nn;
    w.oCs0 = o; o += nn;
    w.oCt0 = o; o += nn;
    w.oC = o; o += nn;                       // thresholded cross cost
    w.oWT = o; o += nn * WD_ITERS;           // T_1..T_30
    w.oWd = o; o += nv * WD_ITERS;           // delta_1..30
    w.oWs = o; o += nv * (WD_ITERS + 1);     // sigma_0..30
    w.oCs = o; o += nn;
    w.oCt = o; o += nn;
    w.ors = o; o += nv;
    w.ort = o; o += nv;
    w.oCg = o; o += nn * GW_OUTER;           // C_gamma of every outer iteration
    w.oGT = o; o += nn * GW_OUTER * GW_INNER;
    w.oGd = o; o += nv * GW_OUTER * GW_INNER;
    w.oGs = o; o += nv * GW_OUTER * (GW_INNER + 1);
    w.oP1 = o; o += nn;
    w.oP2 = o; o += nn;
    w.oP3 = o; o += nn;
    w.ogT = o; o += nn;
    w.ogA = o; o += nn;
    w.ogCs = o; o += nn;
    w.ogCt = o; o += nn;
    w.oG = o; o += nn;
    w.ogC0 = o; o += nn;                     // d/d(raw cross cost), written by the WD sweep, read by cost_bwd
    w.per_case = o;
    int64_t g = (int64_t)k * w.per_case;
    w.g_cases = 0;
    w.g_ext = g; g += up4((int64_t)k * 6);   // per-case (min,max) x3
    w.g_thr = g; g += 32;                    // [0..5] extrema used, [6..8] thresholds, [9..14] tie counts, [15..17] gthr
    w.g_gthr = g; g += up4((int64_t)k * 3);  // per-case threshold-gradient partials
    w.g_wd = g; g += up4(k);
    w.g_gwd = g; g += up4(k);
    w.g_cases = g;                           // total floats
    return w;
}

// ---------------------------------------------------------------------------------------------------------
// block-wide helpers (256 threads = 4 waves).  LDS scratch is passed in by the kernels.
// ---------------------------------------------------------------------------------------------------------
struct Ctx {
    int n, tid, lane, wave;
    float* colbuf;  // LDS [4][GOT_MAXN]
};

// merge per-lane column accumulators of the 4 waves: out[j] = f(sum_w cacc_w[j])
template <class F>
__device__ __forceinline__ void col_finish(const Ctx& c, const float (&cacc)[4], F&& f) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = c.lane + 64 * q;
        if (j < c.n) c.colbuf[c.wave * GOT_MAXN + j] = cacc[q];
    }
    __syncthreads();
    for (int j = c.tid; j < c.n; j += 256)
        f(j, ((c.colbuf[j] + c.colbuf[GOT_MAXN + j]) + c.colbuf[2 * GOT_MAXN + j]) + c.colbuf[3 * GOT_MAXN + j]);
    __syncthreads();
}

// C = X * Y (row-major n x n, generic pointers), optional transposes handled by the callers via symmetric inputs.
// ta: use X^T (X[k][i]); tb: use Y^T (Y[j][k]).  Each wave owns 4 rows at a time, each lane up to 4 columns.
template <bool TA, bool TB, class Epi>
__device__ void matmul(const Ctx& c, const float* __restrict__ X, const float* __restrict__ Y, Epi&& epi) {
    const int n = c.n;
    for (int i0 = c.wave * 4; i0 < n; i0 += 16) {
        float acc[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[r][q] = 0.f;
        for (int k = 0; k < n; ++k) {
            float x[4], y[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + r;
                x[r] = (i < n) ? (TA ? X[(int64_t)k * n + i] : X[(int64_t)i * n + k]) : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = c.lane + 64 * q;
                y[q] = (j < n) ? (TB ? Y[(int64_t)j * n + k] : Y[(int64_t)k * n + j]) : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[r][q] = fmaf(x[r], y[q], acc[r][q]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + r, j = c.lane + 64 * q;
                if (i < n && j < n) epi(i, j, acc[r][q]);
            }
    }
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------
// IPOT forward (loss.py:179-193) on a materialised cost matrix Cm; stores T_1..T_iters, delta_1.., sigma_0..
// ---------------------------------------------------------------------------------------------------------
__device__ void ipot_forward(const Ctx& c, const float* __restrict__ Cm, float inv_beta, int iters, float* __restrict__ Thist,
                             float* __restrict__ dhist, float* __restrict__ shist, float* sig, float* del, float* del2) {
    const int n = c.n;
    const int64_t nn = up4((int64_t)n * n);
    const int nv = (int)up4(n);
    const float fn = (float)n;
    for (int j = c.tid; j < n; j += 256) {
        sig[j] = 1.f / fn;
        shist[j] = 1.f / fn;
    }
    __syncthreads();
    // delta_1 from T_0 = 1:  r_i = sum_j A_ij sigma_j
    for (int i = c.wave; i < n; i += 4) {
        float r = 0.f;
        for (int j = c.lane; j < n; j += 64) r += expf(-Cm[(int64_t)i * n + j] * inv_beta) * sig[j];
        r = wave_sum(r);
        if (c.lane == 0) del[i] = 1.f / (fn * r);
    }
    __syncthreads();
    for (int t = 1; t <= iters; ++t) {
        const float* __restrict__ Tp = (t >= 2) ? Thist + (int64_t)(t - 2) * nn : nullptr;
        float* __restrict__ Tn = Thist + (int64_t)(t - 1) * nn;
        // column pass: a_j = sum_i Q_ij delta_i ; sigma_t = 1 / (m a_j)
        float cacc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = c.wave; i < n; i += 4) {
            const float di = del[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int j = c.lane + 64 * q;
                if (j < n) {
                    const int64_t e = (int64_t)i * n + j;
                    const float Q = expf(-Cm[e] * inv_beta) * (Tp ? Tp[e] : 1.f);
                    cacc[q] += Q * di;
                }
            }
        }
        col_finish(c, cacc, [&](int j, float a) {
            const float s = 1.f / (fn * a);
            sig[j] = s;
            shist[(int64_t)t * nv + j] = s;
        });
        for (int i = c.tid; i < n; i += 256) dhist[(int64_t)(t - 1) * nv + i] = del[i];
        // elementwise + row pass: T_t = delta_i Q_ij sigma_j ; next delta from r_i = sum_j A_ij T_t,ij sigma_j
        for (int i = c.wave; i < n; i += 4) {
            const float di = del[i];
            float r = 0.f;
            for (int j = c.lane; j < n; j += 64) {
                const int64_t e = (int64_t)i * n + j;
                const float A = expf(-Cm[e] * inv_beta);
                const float Tv = di * (A * (Tp ? Tp[e] : 1.f)) * sig[j];
                Tn[e] = Tv;
                r += A * Tv * sig[j];
            }
            r = wave_sum(r);
            if (c.lane == 0) del2[i] = 1.f / (fn * r);
        }
        __syncthreads();
        for (int i = c.tid; i < n; i += 256) del[i] = del2[i];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// IPOT reverse sweep.  In: gT (n x n, gradient wrt the returned plan T_iters; overwritten), out: gA accumulated
// over iterations is folded into gC = gA * A * (-1/beta) written to gC_out (n x n).  Vectors in LDS.
// ---------------------------------------------------------------------------------------------------------
__device__ void ipot_backward(const Ctx& c, const float* __restrict__ Cm, float inv_beta, int iters,
                              const float* __restrict__ Thist, const float* __restrict__ dhist,
                              const float* __restrict__ shist, float* __restrict__ gT, float* __restrict__ gA,
                              float* __restrict__ gC_out, float* gsig, float* ga, float* gr, float* gdel) {
    const int n = c.n;
    const int64_t nn = up4((int64_t)n * n);
    const int nv = (int)up4(n);
    const float fn = (float)n;
    for (int64_t e = c.tid; e < (int64_t)n * n; e += 256) gA[e] = 0.f;
    for (int j = c.tid; j < n; j += 256) gsig[j] = 0.f;
    __syncthreads();
    for (int t = iters; t >= 1; --t) {
        const float* __restrict__ Tp = (t >= 2) ? Thist + (int64_t)(t - 2) * nn : nullptr;
        const float* __restrict__ dl = dhist + (int64_t)(t - 1) * nv;
        const float* __restrict__ sg = shist + (int64_t)t * nv;
        const float* __restrict__ so = shist + (int64_t)(t - 1) * nv;
        // P1: gdel_i = sum_j gT_ij Q_ij sig_j ; u_j = sum_i gT_ij del_i Q_ij ; ga_j = -m sig_j^2 (gsig_j + u_j)
        {
            float cacc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i = c.wave; i < n; i += 4) {
                const float di = dl[i];
                float r = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = c.lane + 64 * q;
                    if (j < n) {
                        const int64_t e = (int64_t)i * n + j;
                        const float Q = expf(-Cm[e] * inv_beta) * (Tp ? Tp[e] : 1.f);
                        const float gq = gT[e] * Q;
                        r += gq * sg[j];
                        cacc[q] += gq * di;
                    }
                }
                r = wave_sum(r);
                if (c.lane == 0) gdel[i] = r;
            }
            col_finish(c, cacc, [&](int j, float u) { ga[j] = -fn * sg[j] * sg[j] * (gsig[j] + u); });
        }
        // P2: gdel_i += sum_j Q_ij ga_j ; gr_i = -n del_i^2 gdel_i
        for (int i = c.wave; i < n; i += 4) {
            float r = 0.f;
            for (int j = c.lane; j < n; j += 64) {
                const int64_t e = (int64_t)i * n + j;
                r += expf(-Cm[e] * inv_beta) * (Tp ? Tp[e] : 1.f) * ga[j];
            }
            r = wave_sum(r);
            if (c.lane == 0) {
                const float di = dl[i];
                gr[i] = -fn * di * di * (gdel[i] + r);
            }
        }
        __syncthreads();
        // P3: gsig_old_j = sum_i Q_ij gr_i ; gQ = gT del sig + ga del + gr sig_old ; gA += gQ T_prev ; gT <- gQ A
        {
            float cacc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i = c.wave; i < n; i += 4) {
                const float di = dl[i], gri = gr[i];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = c.lane + 64 * q;
                    if (j < n) {
                        const int64_t e = (int64_t)i * n + j;
                        const float A = expf(-Cm[e] * inv_beta);
                        const float Tv = Tp ? Tp[e] : 1.f;
                        cacc[q] += A * Tv * gri;
                        const float gQ = gT[e] * di * sg[j] + ga[j] * di + gri * so[j];
                        gA[e] += gQ * Tv;
                        gT[e] = gQ * A;
                    }
                }
            }
            col_finish(c, cacc, [&](int j, float v) { gsig[j] = v; });
        }
    }
    // dL/dC = dL/dA * dA/dC = gA * (-1/beta) A
    for (int64_t e = c.tid; e < (int64_t)n * n; e += 256) gC_out[e] = -inv_beta * gA[e] * expf(-Cm[e] * inv_beta);
    __syncthreads();
}

// =========================================================================================================
// K1: normalise tokens, raw cost matrices, per-case extrema
// =========================================================================================================
__global__ __launch_bounds__(256) void got_prep_kernel(const float* __restrict__ V, const float* __restrict__ Q, float* ws,
                                                       int k, int n, int d) {
    __shared__ float red[4][6];
    const GotWs L = got_layout(k, n, d);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* base = ws + (int64_t)b * L.per_case;
    float* Vh = base + L.oVh;
    float* Qh = base + L.oQh;
    // normalise: one wave per token row
    for (int r = wave; r < 2 * n; r += 4) {
        const bool isq = r >= n;
        const int i = isq ? r - n : r;
        const float* src = (isq ? Q : V) + ((int64_t)b * n + i) * d;
        float ss = 0.f;
        for (int e = lane; e < d; e += 64) ss += src[e] * src[e];
        ss = wave_sum(ss);
        const float nr = sqrtf(ss);
        const float s = 1.f / (nr + 1e-12f);
        float* dst = (isq ? Qh : Vh) + (int64_t)i * d;
        for (int e = lane; e < d; e += 64) dst[e] = src[e] * s;
        if (lane == 0) (base + (isq ? L.orQ : L.orV))[i] = nr;
    }
    __syncthreads();
    // raw costs 1 - <x_i, y_j>: thread per (i,j) element, 3 matrices
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t e = tid; e < (int64_t)n * n; e += 256) {
        const int i = (int)(e / n), j = (int)(e % n);
        const float* vi = Vh + (int64_t)i * d;
        const float* vj = Vh + (int64_t)j * d;
        const float* qi = Qh + (int64_t)i * d;
        const float* qj = Qh + (int64_t)j * d;
        float c0 = 0.f, cs = 0.f, ct = 0.f;
        for (int x = 0; x < d; ++x) {
            c0 = fmaf(vi[x], qj[x], c0);
            cs = fmaf(vi[x], vj[x], cs);
            ct = fmaf(qi[x], qj[x], ct);
        }
        c0 = 1.f - c0;
        cs = 1.f - cs;
        ct = 1.f - ct;
        base[L.oC0 + e] = c0;
        base[L.oCs0 + e] = cs;
        base[L.oCt0 + e] = ct;
        mn[0] = fminf(mn[0], c0); mx[0] = fmaxf(mx[0], c0);
        mn[1] = fminf(mn[1], cs); mx[1] = fmaxf(mx[1], cs);
        mn[2] = fminf(mn[2], ct); mx[2] = fmaxf(mx[2], ct);
    }
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        float a = mn[m], z = mx[m];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a = fminf(a, __shfl_xor(a, o, 64));
            z = fmaxf(z, __shfl_xor(z, o, 64));
        }
        if (lane == 0) {
            red[wave][2 * m] = a;
            red[wave][2 * m + 1] = z;
        }
    }
    __syncthreads();
    if (tid < 6) {
        float v = red[0][tid];
        for (int w = 1; w < 4; ++w) v = (tid & 1) ? fmaxf(v, red[w][tid]) : fminf(v, red[w][tid]);
        ws[L.g_ext + (int64_t)b * 6 + tid] = v;
    }
}

// K2: global extrema (local to this call) -> minmax_out; thresholds from minmax_in if given, else the local ones
__global__ __launch_bounds__(64) void got_minmax_kernel(float* ws, float* __restrict__ minmax_out,
                                                        const float* __restrict__ minmax_in, int k, int n, int d) {
    const GotWs L = got_layout(k, n, d);
    const int tid = threadIdx.x;
    if (tid < 6) {
        float v = (tid & 1) ? -INFINITY : INFINITY;
        for (int b = 0; b < k; ++b) {
            const float x = ws[L.g_ext + (int64_t)b * 6 + tid];
            v = (tid & 1) ? fmaxf(v, x) : fminf(v, x);
        }
        if (minmax_out) minmax_out[tid] = v;
        ws[L.g_thr + tid] = minmax_in ? minmax_in[tid] : v;
    }
    __syncthreads();
    if (tid < 3) {
        const float lo = ws[L.g_thr + 2 * tid], hi = ws[L.g_thr + 2 * tid + 1];
        ws[L.g_thr + 6 + tid] = lo + THR_BETA * (hi - lo);
    }
}

// =========================================================================================================
// K3: Wasserstein branch forward
// =========================================================================================================
__global__ __launch_bounds__(256) void got_wd_kernel(float* ws, int k, int n, int d) {
    __shared__ float sig[GOT_MAXN], del[GOT_MAXN], del2[GOT_MAXN], colbuf[4 * GOT_MAXN], red[4];
    const GotWs L = got_layout(k, n, d);
    const int b = blockIdx.x, tid = threadIdx.x;
    Ctx c{n, tid, tid & 63, tid >> 6, colbuf};
    float* base = ws + (int64_t)b * L.per_case;
    const float thr = ws[L.g_thr + 6];
    float* C = base + L.oC;
    for (int64_t e = tid; e < (int64_t)n * n; e += 256) C[e] = fmaxf(base[L.oC0 + e] - thr, 0.f);
    __syncthreads();
    ipot_forward(c, C, WD_INV_BETA, WD_ITERS, base + L.oWT, base + L.oWd, base + L.oWs, sig, del, del2);
    const float* T = base + L.oWT + (int64_t)(WD_ITERS - 1) * up4((int64_t)n * n);
    float s = 0.f;
    for (int64_t e = tid; e < (int64_t)n * n; e += 256) s += C[e] * T[e];
    s = wave_sum(s);
    if (c.lane == 0) red[c.wave] = s;
    __syncthreads();
    if (tid == 0) ws[L.g_wd + b] = (red[0] + red[1]) + (red[2] + red[3]);
}

// =========================================================================================================
// K4: Gromov-Wasserstein branch forward
// =========================================================================================================
__device__ void gw_cgamma(const Ctx& c, const float* __restrict__ Cs, const float* __restrict__ Ct,
                          const float* __restrict__ gamma /* nullptr => uniform 1/n^2 */, const float* rs, const float* rt,
                          float* __restrict__ P1, float* __restrict__ Cg) {
    const int n = c.n;
    if (gamma) {
        // P1 = gamma * Ct^T (= gamma * Ct, Ct symmetric);  Cg = rs_i + rt_j - 2 (Cs P1)_ij
        matmul<false, false>(c, gamma, Ct, [&](int i, int j, float v) { P1[(int64_t)i * n + j] = v; });
    } else {
        // gamma = 1/n^2 everywhere: (gamma Ct^T)_kj = (1/n^2) sum_l Ct_jl ; reuse rt? no: plain row sums of Ct
        for (int j = c.wave; j < n; j += 4) {
            float s = 0.f;
            for (int l = c.lane; l < n; l += 64) s += Ct[(int64_t)j * n + l];
            s = wave_sum(s) / ((float)n * (float)n);
            for (int kk = c.lane; kk < n; kk += 64) P1[(int64_t)kk * n + j] = s;
        }
        __syncthreads();
    }
    matmul<false, false>(c, Cs, P1, [&](int i, int j, float v) { Cg[(int64_t)i * n + j] = (rs[i] + rt[j]) - 2.f * v; });
}

__global__ __launch_bounds__(256) void got_gw_kernel(float* ws, int k, int n, int d) {
    __shared__ float sig[GOT_MAXN], del[GOT_MAXN], del2[GOT_MAXN], colbuf[4 * GOT_MAXN], rs[GOT_MAXN], rt[GOT_MAXN], red[4];
    const GotWs L = got_layout(k, n, d);
    const int b = blockIdx.x, tid = threadIdx.x;
    Ctx c{n, tid, tid & 63, tid >> 6, colbuf};
    float* base = ws + (int64_t)b * L.per_case;
    const int64_t nn = up4((int64_t)n * n);
    const int nv = (int)up4(n);
    const float thr_s = ws[L.g_thr + 7], thr_t = ws[L.g_thr + 8];
    float* Cs = base + L.oCs;
    float* Ct = base + L.oCt;
    for (int64_t e = tid; e < (int64_t)n * n; e += 256) {
        Cs[e] = fmaxf(base[L.oCs0 + e] - thr_s, 0.f);
        Ct[e] = fmaxf(base[L.oCt0 + e] - thr_t, 0.f);
    }
    __syncthreads();
    // rs_i = (1/n) sum_k Cs_ik^2 ; rt_j = (1/n) sum_l Ct_jl^2     (Cst = rs 1^T + 1 rt^T, loss.py:240-241)
    for (int i = c.wave; i < n; i += 4) {
        float a = 0.f, z = 0.f;
        for (int j = c.lane; j < n; j += 64) {
            const float x = Cs[(int64_t)i * n + j], y = Ct[(int64_t)i * n + j];
            a += x * x;
            z += y * y;
        }
        a = wave_sum(a);
        z = wave_sum(z);
        if (c.lane == 0) {
            rs[i] = a / (float)n;
            rt[i] = z / (float)n;
            base[L.ors + i] = rs[i];
            base[L.ort + i] = rt[i];
        }
    }
    __syncthreads();
    const float* gamma = nullptr;
    for (int o = 0; o < GW_OUTER; ++o) {
        float* Cg = base + L.oCg + (int64_t)o * nn;
        gw_cgamma(c, Cs, Ct, gamma, rs, rt, base + L.oP1, Cg);
        float* Th = base + L.oGT + (int64_t)o * GW_INNER * nn;
        ipot_forward(c, Cg, GW_INV_BETA, GW_INNER, Th, base + L.oGd + (int64_t)o * GW_INNER * nv,
                     base + L.oGs + (int64_t)o * (GW_INNER + 1) * nv, sig, del, del2);
        gamma = Th + (int64_t)(GW_INNER - 1) * nn;
    }
    // final C_gamma (not stored: only the distance) -> P2 ; gwd_b = sum C_gamma * gamma
    gw_cgamma(c, Cs, Ct, gamma, rs, rt, base + L.oP1, base + L.oP2);
    float s = 0.f;
    for (int64_t e = tid; e < (int64_t)n * n; e += 256) s += base[L.oP2 + e] * gamma[e];
    s = wave_sum(s);
    if (c.lane == 0) red[c.wave] = s;
    __syncthreads();
    if (tid == 0) ws[L.g_gwd + b] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(64) void got_sum_kernel(const float* ws, float* __restrict__ out, int k, int n, int d) {
    const GotWs L = got_layout(k, n, d);
    if (threadIdx.x < 2) {
        const float* src = ws + (threadIdx.x ? L.g_gwd : L.g_wd);
        float s = 0.f;
        for (int b = 0; b < k; ++b) s += src[b];
        out[threadIdx.x] = s;
    }
}

// =========================================================================================================
// backward K6: Wasserstein branch.  wd_b = sum C*T(C):  gC = g (T + dT/dC^T [C]);  masked to raw-cost gradient
// in place into gT region?  -> written to the per-case G matrix (oG) as d/dC0 (cross), plus threshold partial.
// =========================================================================================================
__global__ __launch_bounds__(256) void got_wd_bwd_kernel(float* ws, const float* __restrict__ d_out, int k, int n, int d) {
    __shared__ float gsig[GOT_MAXN], ga[GOT_MAXN], gr[GOT_MAXN], gdel[GOT_MAXN], colbuf[4 * GOT_MAXN], red[4];
    const GotWs L = got_layout(k, n, d);
    const int b = blockIdx.x, tid = threadIdx.x;
    Ctx c{n, tid, tid & 63, tid >> 6, colbuf};
    float* base = ws + (int64_t)b * L.per_case;
    const int64_t nn = up4((int64_t)n * n);
    const float g = d_out[0];
    const float* C = base + L.oC;
    const float* Tf = base + L.oWT + (int64_t)(WD_ITERS - 1) * nn;
    float* gT = base + L.ogT;
    for (int64_t e = tid; e < (int64_t)n * n; e += 256) gT[e] = g * C[e];
    __syncthreads();
    ipot_backward(c, C, WD_INV_BETA, WD_ITERS, base + L.oWT, base + L.oWd, base + L.oWs, gT, base + L.ogA, base + L.oP1, gsig,
                  ga, gr, gdel);
    // total dL/dC = g T_final + (through IPOT) ; mask by relu ; accumulate -sum as threshold gradient
    const float thr = ws[L.g_thr + 6];
    float* G = base + L.ogC0;
    float s = 0.f;
    for (int64_t e = tid; e < (int64_t)n * n; e += 256) {
        const float gc = g * Tf[e] + base[L.oP1 + e];
        const float m = (base[L.oC0 + e] - thr > 0.f) ? gc : 0.f;
        G[e] = m;
        s -= m;
    }
    s = wave_sum(s);
    if (c.lane == 0) red[c.wave] = s;
    __syncthreads();
    if (tid == 0) ws[L.g_gthr + (int64_t)b * 3 + 0] = (red[0] + red[1]) + (red[2] + red[3]);
}

// =========================================================================================================
// backward K7: Gromov-Wasserstein branch -> gCs, gCt (wrt thresholded intra costs), then masked to raw costs
// =========================================================================================================
__global__ __launch_bounds__(256) void got_gw_bwd_kernel(float* ws, const float* __restrict__ d_out, int k, int n, int d) {
    __shared__ float gsig[GOT_MAXN], ga[GOT_MAXN], gr[GOT_MAXN], gdel[GOT_MAXN], colbuf[4 * GOT_MAXN];
    __shared__ float grs[GOT_MAXN], grt[GOT_MAXN], red[4][2];
    const GotWs L = got_layout(k, n, d);
    const int b = blockIdx.x, tid = threadIdx.x;
    Ctx c{n, tid, tid & 63, tid >> 6, colbuf};
    float* base = ws + (int64_t)b * L.per_case;
    const int64_t nn = up4((int64_t)n * n);
    const int nv = (int)up4(n);
    const int64_t N2 = (int64_t)n * n;
    const float g = d_out[1];
    const float* Cs = base + L.oCs;
    const float* Ct = base + L.oCt;
    float* gCs = base + L.ogCs;
    float* gCt = base + L.ogCt;
    float* G = base + L.oG;     // gradient wrt the current C_gamma
    float* P1 = base + L.oP1;
    float* P2 = base + L.oP2;
    float* P3 = base + L.oP3;
    float* gT = base + L.ogT;
    for (int64_t e = tid; e < N2; e += 256) {
        gCs[e] = 0.f;
        gCt[e] = 0.f;
    }
    for (int i = tid; i < n; i += 256) {
        grs[i] = 0.f;
        grt[i] = 0.f;
    }
    // seed: gwd = sum C_gamma_final * gamma5 (gamma5 detached as a factor) => G = g * gamma5
    const float* gamma5 = base + L.oGT + ((int64_t)(GW_OUTER - 1) * GW_INNER + (GW_INNER - 1)) * nn;
    for (int64_t e = tid; e < N2; e += 256) G[e] = g * gamma5[e];
    __syncthreads();

    // o = GW_OUTER: the final C_gamma (uses gamma of outer GW_OUTER-1); o = GW_OUTER-1 .. 0: the loop bodies
    for (int o = GW_OUTER; o >= 0; --o) {
        // C_gamma^{(o)} = rs_i + rt_j - 2 (Cs gam Ct)_ij with gam = gamma^{(o-1)} (uniform for o == 0)
        const float* gam = (o >= 1) ? base + L.oGT + ((int64_t)(o - 1) * GW_INNER + (GW_INNER - 1)) * nn : nullptr;
        // Cst gradient: grs_i += sum_j G_ij ; grt_j += sum_i G_ij
        {
            float cacc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i = c.wave; i < n; i += 4) {
                float r = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int j = c.lane + 64 * q;
                    if (j < n) {
                        const float v = G[(int64_t)i * n + j];
                        r += v;
                        cacc[q] += v;
                    }
                }
                r = wave_sum(r);
                if (c.lane == 0) grs[i] += r;
            }
            col_finish(c, cacc, [&](int j, float v) { grt[j] += v; });
        }
        // M = Cs gam Ct:  gCs += -2 G (gam Ct)^T = -2 G Ct gam^T ; gCt += -2 G^T (Cs gam) ; ggam = -2 Cs G Ct
        if (gam) {
            matmul<false, false>(c, G, Ct, [&](int i, int j, float v) { P1[(int64_t)i * n + j] = v; });          // G Ct
            matmul<false, true>(c, P1, gam, [&](int i, int j, float v) { gCs[(int64_t)i * n + j] -= 2.f * v; });  // (G Ct) gam^T
            matmul<false, false>(c, Cs, gam, [&](int i, int j, float v) { P2[(int64_t)i * n + j] = v; });        // Cs gam
            matmul<true, false>(c, G, P2, [&](int i, int j, float v) { gCt[(int64_t)i * n + j] -= 2.f * v; });    // G^T (Cs gam)
            matmul<false, false>(c, Cs, P1, [&](int i, int j, float v) { gT[(int64_t)i * n + j] = -2.f * v; });   // Cs (G Ct)
        } else {
            // gam = 1/n^2 (constant): (G Ct gam^T)_ik = (1/n^2) sum_l (G Ct)_il ; (G^T Cs gam)_jl = (1/n^2) sum_i G_ij sum_k Cs_ik
            const float inv = 1.f / ((float)n * (float)n);
            matmul<false, false>(c, G, Ct, [&](int i, int j, float v) { P1[(int64_t)i * n + j] = v; });
            for (int i = c.wave; i < n; i += 4) {
                float r = 0.f, cs = 0.f;
                for (int j = c.lane; j < n; j += 64) {
                    r += P1[(int64_t)i * n + j];
                    cs += Cs[(int64_t)i * n + j];
                }
                r = wave_sum(r) * inv;
                cs = wave_sum(cs) * inv;
                for (int kk = c.lane; kk < n; kk += 64) gCs[(int64_t)i * n + kk] -= 2.f * r;
                if (c.lane == 0) gdel[i] = cs;  // (Cs gam)_i* = cs (constant along columns)
            }
            __syncthreads();
            // gCt_jl += -2 sum_i G_ij (Cs gam)_il = -2 sum_i G_ij cs_i   (same for every l)
            {
                float cacc[4] = {0.f, 0.f, 0.f, 0.f};
                for (int i = c.wave; i < n; i += 4) {
                    const float w = gdel[i];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int j = c.lane + 64 * q;
                        if (j < n) cacc[q] += G[(int64_t)i * n + j] * w;
                    }
                }
                col_finish(c, cacc, [&](int j, float v) { ga[j] = -2.f * v; });
                for (int64_t e = tid; e < N2; e += 256) gCt[e] += ga[(int)(e / n)];
                __syncthreads();
            }
            break;  // gamma^{(-1)} is a constant: nothing further upstream
        }
        // gT now holds d/d gamma^{(o-1)}; back through IPOT of outer o-1 -> gradient wrt C_gamma^{(o-1)} into G
        const int oo = o - 1;
        ipot_backward(c, base + L.oCg + (int64_t)oo * nn, GW_INV_BETA, GW_INNER, base + L.oGT + (int64_t)oo * GW_INNER * nn,
                      base + L.oGd + (int64_t)oo * GW_INNER * nv, base + L.oGs + (int64_t)oo * (GW_INNER + 1) * nv, gT,
                      base + L.ogA, G, gsig, ga, gr, gdel);
    }
    // Cst terms: rs_i = (1/n) sum_k Cs_ik^2  => gCs_ik += (2/n) Cs_ik grs_i ; same for Ct
    const float thr_s = ws[L.g_thr + 7], thr_t = ws[L.g_thr + 8];
    float ss = 0.f, st = 0.f;
    for (int64_t e = tid; e < N2; e += 256) {
        const int i = (int)(e / n);
        float a = gCs[e] + (2.f / (float)n) * Cs[e] * grs[i];
        float z = gCt[e] + (2.f / (float)n) * Ct[e] * grt[i];
        a = (base[L.oCs0 + e] - thr_s > 0.f) ? a : 0.f;
        z = (base[L.oCt0 + e] - thr_t > 0.f) ? z : 0.f;
        gCs[e] = a;  // now gradients wrt the RAW intra costs
        gCt[e] = z;
        ss -= a;
        st -= z;
    }
    ss = wave_sum(ss);
    st = wave_sum(st);
    if (c.lane == 0) {
        red[c.wave][0] = ss;
        red[c.wave][1] = st;
    }
    __syncthreads();
    if (tid < 2) ws[L.g_gthr + (int64_t)b * 3 + 1 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// K8: threshold gradients -> extrema gradients; tie counts for the local extrema
__global__ __launch_bounds__(256) void got_thr_bwd_kernel(float* ws, float* __restrict__ d_minmax, int k, int n, int d) {
    __shared__ float cnt_s[6];
    const GotWs L = got_layout(k, n, d);
    const int tid = threadIdx.x;
    if (tid < 3) {
        float s = 0.f;
        for (int b = 0; b < k; ++b) s += ws[L.g_gthr + (int64_t)b * 3 + tid];
        ws[L.g_thr + 15 + tid] = s;
        // thr = min + beta (max - min): d/dmin = (1-beta) gthr ; d/dmax = beta gthr
        if (d_minmax) {
            d_minmax[2 * tid] = (1.f - THR_BETA) * s;
            d_minmax[2 * tid + 1] = THR_BETA * s;
        }
    }
    if (tid < 6) cnt_s[tid] = 0.f;
    __syncthreads();
    // count LOCAL elements equal to each extremum (torch's min()/max() backward spreads the gradient evenly over ties)
    float cnt[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    float ex[6];
#pragma unroll
    for (int m = 0; m < 6; ++m) ex[m] = ws[L.g_thr + m];
    const int64_t N2 = (int64_t)n * n;
    for (int64_t e = tid; e < (int64_t)k * N2; e += 256) {
        const float* base = ws + (e / N2) * L.per_case;
        const int64_t r = e % N2;
        const float v0 = base[L.oC0 + r], v1 = base[L.oCs0 + r], v2 = base[L.oCt0 + r];
        cnt[0] += v0 == ex[0]; cnt[1] += v0 == ex[1];
        cnt[2] += v1 == ex[2]; cnt[3] += v1 == ex[3];
        cnt[4] += v2 == ex[4]; cnt[5] += v2 == ex[5];
    }
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const float v = wave_sum(cnt[m]);
        if ((tid & 63) == 0) atomicAdd(&cnt_s[m], v);
    }
    __syncthreads();
    if (tid < 6) ws[L.g_thr + 9 + tid] = cnt_s[tid];
}

// K9: raw-cost gradients (+ extrema routing) -> normalised-token gradients -> dV, dQ
__global__ __launch_bounds__(256) void got_cost_bwd_kernel(const float* __restrict__ V, const float* __restrict__ Q, float* ws,
                                                           float* __restrict__ dV, float* __restrict__ dQ,
                                                           const float* __restrict__ d_minmax_in, int k, int n, int d) {
    const GotWs L = got_layout(k, n, d);
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* base = ws + (int64_t)b * L.per_case;
    const int64_t N2 = (int64_t)n * n;
    float* G0 = base + L.ogC0;    // d/dC0
    float* Gs = base + L.ogCs;    // d/dCs0
    float* Gt = base + L.ogCt;    // d/dCt0
    {
        // gradient of the six extrema, spread evenly over the (local) elements that attain them.  In global mode
        // (thresholds supplied by the caller) an extremum owned by another rank matches no local element.
        float gex[6], ex[6], cnt[6];
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            ex[m] = ws[L.g_thr + m];
            cnt[m] = ws[L.g_thr + 9 + m];
            const float gthr = ws[L.g_thr + 15 + m / 2];
            const float gloc = (m & 1) ? THR_BETA * gthr : (1.f - THR_BETA) * gthr;
            gex[m] = d_minmax_in ? d_minmax_in[m] : gloc;
        }
        for (int64_t e = tid; e < N2; e += 256) {
            const float v0 = base[L.oC0 + e], v1 = base[L.oCs0 + e], v2 = base[L.oCt0 + e];
            float a = G0[e], s = Gs[e], t = Gt[e];
            if (v0 == ex[0] && cnt[0] > 0.f) a += gex[0] / cnt[0];
            if (v0 == ex[1] && cnt[1] > 0.f) a += gex[1] / cnt[1];
            if (v1 == ex[2] && cnt[2] > 0.f) s += gex[2] / cnt[2];
            if (v1 == ex[3] && cnt[3] > 0.f) s += gex[3] / cnt[3];
            if (v2 == ex[4] && cnt[4] > 0.f) t += gex[4] / cnt[4];
            if (v2 == ex[5] && cnt[5] > 0.f) t += gex[5] / cnt[5];
            G0[e] = a;
            Gs[e] = s;
            Gt[e] = t;
        }
        __syncthreads();
    }
    // C0_ij = 1 - <v^_i,q^_j>; Cs0_ij = 1 - <v^_i,v^_j>; Ct0_ij = 1 - <q^_i,q^_j>
    //   gv^_i = - sum_j G0_ij q^_j - sum_j (Gs_ij + Gs_ji) v^_j ;  gq^_j = - sum_i G0_ij v^_i - sum_i (Gt_ji + Gt_ij) q^_i
    // then x^ = x / (r + eps):  gx = s gx^ - (<x^, gx^> / r) x^   (r = |x|, s = 1/(r+eps); r == 0 -> s gx^)
    const float* Vh = base + L.oVh;
    const float* Qh = base + L.oQh;
    for (int row = wave; row < 2 * n; row += 4) {
        const bool isq = row >= n;
        const int i = isq ? row - n : row;
        float acc[2] = {0.f, 0.f};  // lane owns features lane, lane+64 (d <= 128)
        for (int j = 0; j < n; ++j) {
            float wc, wi;
            if (!isq) {
                wc = G0[(int64_t)i * n + j];                                   // pairs with q^_j
                wi = Gs[(int64_t)i * n + j] + Gs[(int64_t)j * n + i];          // pairs with v^_j
            } else {
                wc = G0[(int64_t)j * n + i];                                   // pairs with v^_j
                wi = Gt[(int64_t)i * n + j] + Gt[(int64_t)j * n + i];          // pairs with q^_j
            }
            const float* xc = (isq ? Vh : Qh) + (int64_t)j * d;
            const float* xi = (isq ? Qh : Vh) + (int64_t)j * d;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = lane + 64 * u;
                if (e < d) acc[u] -= wc * xc[e] + wi * xi[e];
            }
        }
        const float* xh = (isq ? Qh : Vh) + (int64_t)i * d;
        float dot = 0.f;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = lane + 64 * u;
            if (e < d) dot += xh[e] * acc[u];
        }
        dot = wave_sum(dot);
        const float r = (base + (isq ? L.orQ : L.orV))[i];
        const float s = 1.f / (r + 1e-12f);
        const float proj = (r > 0.f) ? dot / r : 0.f;
        float* out = (isq ? dQ : dV) + ((int64_t)b * n + i) * d;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = lane + 64 * u;
            if (e < d) out[e] = s * acc[u] - proj * xh[e];
        }
    }
}

}  // namespace mdl

using namespace mdl;

static int got_check(int k, int n, int d) {
    if (k < 0 || n < 0 || d < 1) return MDL_E_ARG;
    if (n > GOT_MAXN || d > GOT_MAXD) return MDL_E_UNSUPPORTED;
    return MDL_OK;
}

extern "C" int64_t mdl_got_ws_bytes(int k, int n, int d) {
    const int rc = got_check(k, n, d);
    if (rc) return rc;
    return got_layout(k, n, d).g_cases * 4 + 64;
}

extern "C" int mdl_got_fwd(const float* V, const float* Q, float* out, float* minmax_out, const float* minmax_in, int k, int n,
                           int d, void* ws, void* stream) {
    const int rc = got_check(k, n, d);
    if (rc) return rc;
    if (!V || !Q || !out || !ws) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (k == 0 || n == 0) {
        const hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(float), s);
        return e == hipSuccess ? MDL_OK : (int)e;
    }
    float* w = (float*)ws;
    hipLaunchKernelGGL(got_prep_kernel, dim3(k), dim3(256), 0, s, V, Q, w, k, n, d);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(got_minmax_kernel, dim3(1), dim3(64), 0, s, w, minmax_out, minmax_in, k, n, d);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(got_wd_kernel, dim3(k), dim3(256), 0, s, w, k, n, d);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(got_gw_kernel, dim3(k), dim3(256), 0, s, w, k, n, d);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(got_sum_kernel, dim3(1), dim3(64), 0, s, (const float*)w, out, k, n, d);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

extern "C" int mdl_got_extrema(const float* V, const float* Q, float* minmax_out, int k, int n, int d, void* ws,
                               void* stream) {
    const int rc = got_check(k, n, d);
    if (rc) return rc;
    if (!V || !Q || !minmax_out || !ws) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    if (k == 0 || n == 0) return MDL_E_ARG;  // extrema of an empty batch are undefined
    hipStream_t s = (hipStream_t)stream;
    float* w = (float*)ws;
    hipLaunchKernelGGL(got_prep_kernel, dim3(k), dim3(256), 0, s, V, Q, w, k, n, d);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(got_minmax_kernel, dim3(1), dim3(64), 0, s, w, minmax_out, (const float*)nullptr, k, n, d);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

extern "C" int mdl_got_bwd_begin(const float* d_out, float* d_minmax, int k, int n, int d, void* ws, void* stream) {
    const int rc = got_check(k, n, d);
    if (rc) return rc;
    if (!d_out || !ws) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (k == 0 || n == 0) {
        if (d_minmax) {
            const hipError_t e = hipMemsetAsync(d_minmax, 0, 6 * sizeof(float), s);
            if (e != hipSuccess) return (int)e;
        }
        return MDL_OK;
    }
    float* w = (float*)ws;
    hipLaunchKernelGGL(got_wd_bwd_kernel, dim3(k), dim3(256), 0, s, w, d_out, k, n, d);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(got_gw_bwd_kernel, dim3(k), dim3(256), 0, s, w, d_out, k, n, d);
    MDL_LAUNCH_CHECK();
    hipLaunchKernelGGL(got_thr_bwd_kernel, dim3(1), dim3(256), 0, s, w, d_minmax, k, n, d);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

extern "C" int mdl_got_bwd_finish(const float* V, const float* Q, float* dV, float* dQ, const float* d_minmax_total, int k,
                                  int n, int d, void* ws, void* stream) {
    const int rc = got_check(k, n, d);
    if (rc) return rc;
    if (!V || !Q || !dV || !dQ || !ws) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    if (k == 0 || n == 0) return MDL_OK;
    hipLaunchKernelGGL(got_cost_bwd_kernel, dim3(k), dim3(256), 0, (hipStream_t)stream, V, Q, (float*)ws, dV, dQ,
                       d_minmax_total, k, n, d);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

extern "C" int mdl_got_bwd(const float* V, const float* Q, const float* d_out, float* dV, float* dQ, int k, int n, int d,
                           void* ws, void* stream) {
    int rc = mdl_got_bwd_begin(d_out, nullptr, k, n, d, ws, stream);
    if (rc) return rc;
    return mdl_got_bwd_finish(V, Q, dV, dQ, nullptr, k, n, d, ws, stream);
}
