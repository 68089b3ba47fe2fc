// preattn_act.hip -- fused LayerNorm -> GELU(erf) -> Dropout, forward and backward.
//
// Replaces the three elementwise modules that follow every Linear of the reference's pre-attention MLP
//     nn.LayerNorm(W) -> nn.GELU() -> nn.Dropout(0.1)           (reference madeleine/models/Model.py:352-354,
//                                                                  :356-358, :360-362;  W = 512, 512, 2048)
// and their autograd.  In torch these are 3 forward + 5 backward kernels moving ~140 B per element; fused they
// move 8 B (fwd: read x, write y) and 12 B (bwd: read x, dy, write dx) per element -- the op is purely HBM-bound.
// LayerNorm semantics: biased variance, eps inside the sqrt (torch default 1e-5), affine.  GELU: exact erf form.
// Dropout: inverted scaling 1/(1-p); mask = explicit uint8 [rows,W] if given, else a counter hash of
// (seed, row*W + col) regenerated in backward (nothing stored; one hash per element pair).
//
// One wave per row: lane L owns columns {4L + 256 i .. +3}, i < W/256 (W = 512 -> 2 float4, 2048 -> 8 float4), so
// a row is W/256 coalesced 1 KiB loads, the mean/variance are two 64-lane reductions, and in backward the per-
// column sums for d(gamma), d(beta) accumulate in registers across the rows a wave walks.
#include "split_engine.hpp"

namespace mdl {

constexpr int ACT_BLOCK = 256;  // 4 waves

// Standard normal CDF through the Numerical-Recipes erfc fit  erfc(z) = t exp(-z^2 + P(t)), t = 1/(1 + z/2)  (fractional
// error < 1.2e-7 everywhere in exact arithmetic, < 2e-6 evaluated in fp32): one v_rcp, one v_exp and 10 FMAs instead of
// libm erff's ~50 instructions with branches -- these kernels were VALU-bound on erff, not HBM-bound.  The erfc form keeps
// the RELATIVE accuracy of the far negative tail (GELU(x) -> x * Phi(x), Phi tiny).
__device__ __forceinline__ float norm_cdf_f(float x) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.5f, z, 1.f));
    float p = 0.17087277f;
    p = fmaf(t, p, -0.82215223f);
    p = fmaf(t, p, 1.48851587f);
    p = fmaf(t, p, -1.13520398f);
    p = fmaf(t, p, 0.27886807f);
    p = fmaf(t, p, -0.18628806f);
    p = fmaf(t, p, 0.09678418f);
    p = fmaf(t, p, 0.37409196f);
    p = fmaf(t, p, 1.00002368f);
    const float h = 0.5f * t * __expf(fmaf(-z, z, fmaf(t, p, -1.26551223f)));   // = erfc(z) / 2
    return x < 0.f ? h : 1.f - h;
}
__device__ __forceinline__ float gelu_f(float v) { return v * norm_cdf_f(v); }
__device__ __forceinline__ float gelu_grad_f(float v) {
    const float pdf = 0.3989422804014327f * __expf(-0.5f * v * v);
    return norm_cdf_f(v) + v * pdf;
}

// bf16 mode only (the fp32 parity path keeps the erfc forms above): Phi(x) ~= sigmoid(x P(x^2)), P = p0 + p1 x^2 + p2 x^4 fitted to the
// exact x Phi(x) on [-6, 6] by minimax-weighted least squares (tools/fit_gelu_sigmoid.py): |x sigma(x P) - GELU(x)| <= 2.5e-5 and the
// derivative of the approximation is within 1.1e-4 of GELU'(x) -- two orders of magnitude below the bf16 grid of the O(1) values these
// kernels store (2^-9 relative), at 15 VALU issue slots instead of 26 (forward) and 21 instead of 33 (backward): the bf16 kernels are
// VALU-bound, not HBM-bound.  The argument is clamped to [-6, 6] (beyond it Phi is 0 / 1 to 1e-9; the quartic term would otherwise
// change sign near |x| = 15).  Coefficients below are pre-multiplied by -log2(e) for v_exp_f32.
constexpr float GS_P0 = 1.59501576f, GS_P1 = 7.40113011e-2f, GS_P2 = -7.03034956e-4f, GS_NL2E = -1.4426950408889634f;
__device__ __forceinline__ float gelu_sig_f(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -6.f, 6.f);
    const float x2 = xc * xc;
    const float q = fmaf(x2, fmaf(x2, GS_P2 * GS_NL2E, GS_P1 * GS_NL2E), GS_P0 * GS_NL2E);
    const float e = __builtin_amdgcn_exp2f(xc * q);          // exp(-xc P(xc^2))
    return x * __builtin_amdgcn_rcpf(1.f + e);
}
__device__ __forceinline__ float gelu_sig_grad_f(float x) {   // d/dx [x sigma(x P(x^2))] = s + x s (1 - s) (p0 + 3 p1 x^2 + 5 p2 x^4)
    const float xc = __builtin_amdgcn_fmed3f(x, -6.f, 6.f);
    const float x2 = xc * xc;
    const float q = fmaf(x2, fmaf(x2, GS_P2 * GS_NL2E, GS_P1 * GS_NL2E), GS_P0 * GS_NL2E);
    const float e = __builtin_amdgcn_exp2f(xc * q);
    const float sg = __builtin_amdgcn_rcpf(1.f + e);
    const float r = fmaf(x2, fmaf(x2, 5.f * GS_P2, 3.f * GS_P1), GS_P0);
    return fmaf(xc * r, (e * sg) * sg, sg);                  // s (1 - s) = e s^2
}
template <class IO>
__device__ __forceinline__ float act_gelu(float v) {
    if constexpr (sizeof(IO) == 2) return gelu_sig_f(v);
    else return gelu_f(v);
}
template <class IO>
__device__ __forceinline__ float act_gelu_grad(float v) {
    if constexpr (sizeof(IO) == 2) return gelu_sig_grad_f(v);
    else return gelu_grad_f(v);
}

// Round 5, bf16 mode: the sigmoid form above still spends two quarter-rate transcendentals per element (v_exp_f32, v_rcp_f32 = 8 of its
// 15 issue slots), and the bf16 LayerNorm kernels are VALU-bound (forward 35 slots per element against 24 at the HBM rate).  Polynomials
// instead -- nothing but FMAs, which the compiler packs two elements per instruction (v_pk_fma_f32):
//     GELU(x)  = x (1/2 + xc P(xc^2)),   GELU'(x) = 1/2 + xc R(xc^2),   xc = clamp(x, -4, 4),   P, R of degree 7 in xc^2
// minimax fits on [-4, 4] (tools/fit_gelu_poly.py): |GELU error| <= 3.8e-5, |GELU' error| <= 2.6e-4 -- an order of magnitude below the
// bf16 grid of the O(1) values these kernels store (2^-9 relative); beyond +-4 the clamped argument leaves an error of <= 4.2e-5 |x|
// (x Phi(-4) instead of ~0 on the far negative side; 3.3e-4 at |x| = 8, profiles/r05x_gelu_polynomial_fit.txt).
// 6 issue slots per element instead of 15 (forward) / 21 (backward).  -DMDL_GELU_POLY=0 keeps the sigmoid form.
// split-mode (fp32 storage, image out) kernels with the hash dropout mode fixed at compile time.  Same-box A/B at config 2
// (profiles/r05z): forward 0.447 -> 0.435 ms avg (3 of 3 pairs), backward 0.660 -> 0.678 (3 of 3 the other way) -- so forward only.
#ifndef MDL_LN_IMG_DM_FWD
#define MDL_LN_IMG_DM_FWD 1
#endif
#ifndef MDL_LN_IMG_DM
#define MDL_LN_IMG_DM 0   // the backward
#endif
#ifndef MDL_GELU_POLY
#define MDL_GELU_POLY 1
#endif
constexpr float GP_P[8] = {3.986733996e-01f, -6.588784796e-02f, 9.505400654e-03f, -1.006490985e-03f,
                           7.485512461e-05f, -3.657138657e-06f, 1.041962646e-07f, -1.301297819e-09f};
constexpr float GP_R[8] = {7.967216592e-01f, -2.620298841e-01f, 5.591485367e-02f, -7.687450676e-03f,
                           6.876469145e-04f, -3.845970028e-05f, 1.213811852e-06f, -1.641988682e-08f};
__device__ __forceinline__ f32x4 clamp4(const f32x4& x, float lim) {
    f32x4 c;
#pragma unroll
    for (int e = 0; e < 4; ++e) c[e] = __builtin_amdgcn_fmed3f(x[e], -lim, lim);
    return c;
}
__device__ __forceinline__ f32x4 poly7(const f32x4& x2, const float (&c)[8]) {
    f32x4 p = x2 * c[7] + c[6];
#pragma unroll
    for (int k = 5; k >= 0; --k) p = p * x2 + c[k];
    return p;
}
// element-wise over a float4 (the LayerNorm kernels work on groups of 4 columns)
template <class IO>
__device__ __forceinline__ f32x4 act_gelu4(const f32x4& t) {
    f32x4 o;
    if constexpr (sizeof(IO) == 2 && MDL_GELU_POLY) {
        const f32x4 xc = clamp4(t, 4.f);
        o = t * (xc * poly7(xc * xc, GP_P) + 0.5f);
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_gelu<IO>(t[e]);
    }
    return o;
}
template <class IO>
__device__ __forceinline__ f32x4 act_gelu_grad4(const f32x4& t) {
    f32x4 o;
    if constexpr (sizeof(IO) == 2 && MDL_GELU_POLY) {
        const f32x4 xc = clamp4(t, 4.f);
        o = xc * poly7(xc * xc, GP_R) + 0.5f;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = act_gelu_grad<IO>(t[e]);
    }
    return o;
}

struct ActDrop {
    float inv;
    uint32_t thr, key;
    const uint8_t* keep;
    int on;
};
// Keep decisions of the 4 consecutive elements [idx4, idx4 + 4) of a row (idx4 % 4 == 0): ONE counter hash per element
// PAIR -- the even element takes the low 16 bits, the odd one the high 16 bits -- and 32-bit index arithmetic inside a
// row (a row never straddles a 2^32 boundary because W divides 2^32).  row_key = key ^ (hi32(row * W) * golden).
// Output: the dropout MULTIPLIER of each element (1/(1-p) or 0) -- booleans crossing the mode branch were materialised as
// packed bytes and unpacked again (~3 extra VALU instructions per element in kernels that are VALU-bound).
__device__ __forceinline__ void act_keep4(const ActDrop& d, uint32_t row_key, uint32_t lo4, int64_t idx4, float (&k)[4]) {
    if (!d.on) {
        k[0] = k[1] = k[2] = k[3] = 1.f;
    } else if (d.keep) {
        const uint32_t m = *reinterpret_cast<const uint32_t*>(d.keep + idx4);
        k[0] = (m & 0xFFu) ? d.inv : 0.f;
        k[1] = (m & 0xFF00u) ? d.inv : 0.f;
        k[2] = (m & 0xFF0000u) ? d.inv : 0.f;
        k[3] = (m >> 24) ? d.inv : 0.f;
    } else {
        const uint32_t h0 = mix32(lo4 ^ row_key), h1 = mix32((lo4 + 2u) ^ row_key);
        k[0] = (h0 & 0xFFFFu) >= d.thr ? d.inv : 0.f;
        k[1] = (h0 >> 16) >= d.thr ? d.inv : 0.f;
        k[2] = (h1 & 0xFFFFu) >= d.thr ? d.inv : 0.f;
        k[3] = (h1 >> 16) >= d.thr ? d.inv : 0.f;
    }
}
// DM >= 0 fixes the mode at compile time (0 off, 1 counter hash, 2 uint8 masks): no mode branch inside the column loop, which otherwise
// cuts the row's arithmetic into a dozen basic blocks the scheduler cannot interleave; DM < 0 = the run-time form above
template <int DM>
__device__ __forceinline__ f32x4 act_keep4_t(const ActDrop& d, uint32_t row_key, uint32_t lo4, int64_t idx4) {
    float k[4];
    if constexpr (DM < 0) {
        act_keep4(d, row_key, lo4, idx4, k);
    } else if constexpr (DM == 0) {
        k[0] = k[1] = k[2] = k[3] = 1.f;
    } else if constexpr (DM == 2) {
        const uint32_t m = *reinterpret_cast<const uint32_t*>(d.keep + idx4);
        k[0] = (m & 0xFFu) ? d.inv : 0.f;
        k[1] = (m & 0xFF00u) ? d.inv : 0.f;
        k[2] = (m & 0xFF0000u) ? d.inv : 0.f;
        k[3] = (m >> 24) ? d.inv : 0.f;
    } else {
        const uint32_t h0 = mix32(lo4 ^ row_key), h1 = mix32((lo4 + 2u) ^ row_key);
        k[0] = (h0 & 0xFFFFu) >= d.thr ? d.inv : 0.f;
        k[1] = (h0 >> 16) >= d.thr ? d.inv : 0.f;
        k[2] = (h1 & 0xFFFFu) >= d.thr ? d.inv : 0.f;
        k[3] = (h1 >> 16) >= d.thr ? d.inv : 0.f;
    }
    return f32x4{k[0], k[1], k[2], k[3]};
}
static inline int act_drop_mode(const ActDrop& d) { return !d.on ? 0 : (d.keep ? 2 : 1); }
// the launchers' choice: compile-time mode for the (VALU-bound) bf16 kernels, the run-time form for fp32 storage (HBM-bound; measured
// no gain there)
template <class IO>
static inline int ln_dm(const ActDrop& d) { return sizeof(IO) == 2 ? act_drop_mode(d) : -1; }
__device__ __forceinline__ uint32_t act_row_key(const ActDrop& d, int64_t row_base) {
    return d.key ^ ((uint32_t)((uint64_t)row_base >> 32) * 0x9E3779B9U);
}

// Column ownership of a lane inside its wave's segment of NV groups of 4 columns.  fp32: group i = columns i*256 + 4L .. +3 (one 16-B
// load).  bf16 with an even NV: groups 2p, 2p+1 = the 8 CONSECUTIVE columns p*512 + 8L .. +7, so a lane still moves 16 B per memory
// instruction (with the fp32 mapping the bf16 kernels issued 8-B loads / stores -- twice the memory instructions per byte -- and ran
// SLOWER than their fp32 siblings on half the bytes).
template <class IO, int NV>
struct ColMap {
    static constexpr bool WIDE = (sizeof(IO) == 2) && (NV % 2 == 0);
    static __device__ __forceinline__ int off(int i, int lane) {
        return WIDE ? (i >> 1) * 512 + lane * 8 + (i & 1) * 4 : i * 256 + lane * 4;
    }
};
// the NV groups of one row segment (p = row base + segment base), nontemporal
template <class IO, int NV>
__device__ __forceinline__ void row_load(const IO* __restrict__ p, int lane, f32x4 (&v)[NV]) {
    if constexpr (ColMap<IO, NV>::WIDE) {
#pragma unroll
        for (int q = 0; q < NV / 2; ++q) {
            const bf16x8 w = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p + q * 512 + lane * 8));
            v[2 * q] = __builtin_convertvector(__builtin_shufflevector(w, w, 0, 1, 2, 3), f32x4);
            v[2 * q + 1] = __builtin_convertvector(__builtin_shufflevector(w, w, 4, 5, 6, 7), f32x4);
        }
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = ld4_nt(p + i * 256 + lane * 4);
    }
}
template <int NV>
__device__ __forceinline__ void row_zero(f32x4 (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// store of group i (fp32 mapping) or, at odd i, of the pair (i-1, i) as one 16-B store (wide bf16 mapping)
template <class IO, int NV>
__device__ __forceinline__ void group_store(IO* __restrict__ p, int lane, int i, const f32x4& prev, const f32x4& cur) {
    if constexpr (ColMap<IO, NV>::WIDE) {
        if (i & 1) {
            const bf16x4 a = __builtin_convertvector(prev, bf16x4), b = __builtin_convertvector(cur, bf16x4);
            *reinterpret_cast<bf16x8*>(p + (i >> 1) * 512 + lane * 8) = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
        }
    } else {
        st4(p + i * 256 + lane * 4, cur);
    }
}

// split-image store of 4 consecutive columns c .. c + 3 of an image row: sp_img_store4 (split_engine.hpp) -- the fp32 column map gives lane l
// columns i * 256 + 4 l, which is the lane order its pair exchange needs; every lane of the wave stores (rows are wave-uniform)
__device__ __forceinline__ void img_store4(char* __restrict__ row, int c, const f32x4& v, float s) { sp_img_store4(row, c, v, s); }

// Geometry: a 256-thread block = 4 waves.  WPR waves share one row (each owns a 256*NV-column segment), so a block
// works on 4/WPR rows at a time: W = 512 -> NV 2, WPR 1 (one wave per row); W = 2048 -> NV 2, WPR 4 (one block per
// row; keeps the backward at ~110 VGPRs = 4 waves/SIMD instead of 256+ = 1 wave/SIMD with a whole row per wave).
template <int WPR>
__device__ __forceinline__ void row_allreduce2(float& a, float& b, float (*red)[4][2], int slot_wave0, int wv) {
    a = wave_sum(a);
    b = wave_sum(b);
    if (WPR > 1) {
        if ((threadIdx.x & 63) == 0) {
            red[0][wv][0] = a;
            red[0][wv][1] = b;
        }
        __syncthreads();
        float x = 0.f, y = 0.f;
#pragma unroll
        for (int w = 0; w < WPR; ++w) {
            x += red[0][slot_wave0 + w][0];
            y += red[0][slot_wave0 + w][1];
        }
        a = x;
        b = y;
        __syncthreads();
    }
}

// Rows of one workgroup: [base0, base1) in steps of `step` row blocks of RPB rows.  Default: the grid-stride order;
// -DMDL_ACT_CONTIG: a contiguous range per workgroup (A/B through tools/ab: null, see the forward kernel).
struct ActRange {
    int64_t base0, base1, step;
};
// rows [lo, hi) of this workgroup's row group (the whole tensor, or -- grouped backward -- the rows of group blockIdx.y)
__device__ __forceinline__ ActRange act_range(int64_t lo, int64_t hi, int RPB) {
#ifndef MDL_ACT_CONTIG
    return ActRange{lo + (int64_t)blockIdx.x * RPB, hi, (int64_t)gridDim.x * RPB};
#else
    const int64_t nrb = (hi - lo + RPB - 1) / RPB, per = (nrb + gridDim.x - 1) / gridDim.x;
    const int64_t b0 = (int64_t)blockIdx.x * per, b1 = b0 + per < nrb ? b0 + per : nrb;
    return ActRange{lo + b0 * RPB, lo + b1 * RPB, (int64_t)RPB};
#endif
}

// IMG: 0 = y only; 1 = split image only (the output feeds contractions of the split engine only); 2 = both (fp32 kernels only)
template <int NV, int WPR, class IO, int IMG = 0, int DM = -1>
__global__ __launch_bounds__(ACT_BLOCK) void ln_gelu_drop_fwd_kernel(const IO* __restrict__ x,
                                                                     const float* __restrict__ bias,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta,
                                                                     IO* __restrict__ y, float* __restrict__ mean_o,
                                                                     float* __restrict__ rstd_o, int64_t rows, float eps,
                                                                     ActDrop drop, char* __restrict__ img = nullptr,
                                                                     const float* __restrict__ img_sc = nullptr,
                                                                     const float* __restrict__ row_mul = nullptr,
                                                                     float* __restrict__ rstd_max = nullptr,
                                                                     const int64_t* __restrict__ cu = nullptr, int bias_gstride = 0) {
    // rstd_max (split variant): raised to max_r rstd[r] * row_mul[r] -- the factor of the backward's dx-image bound, taken here where the
    // statistics are computed instead of by a pass over rstd in every backward (one atomic per wave)
    // cu (grouped forward, round 6): rows [cu[g], cu[g + 1]) of group g = blockIdx.y, whose bias row is bias + g * bias_gstride -- the
    // stain-encoding columns of the first Linear folded into a per-bag bias (Model.py:125-132, :351) for the engines whose GEMM epilogue
    // does not add it (bf16, exact fp32)
    constexpr int W = NV * 256 * WPR, RPB = 4 / WPR;
    const int64_t row_lo = cu ? cu[blockIdx.y] : 0;
    if (cu) {
        rows = cu[blockIdx.y + 1];
        if (bias) bias += (int64_t)blockIdx.y * bias_gstride;
    }
    __shared__ float red[1][4][2];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, slot = wv / WPR, seg = wv % WPR;
    typedef ColMap<IO, NV> CM;
    const int cb = seg * NV * 256;   // first column of this wave's segment
    float rmax = 0.f;
    f32x4 g[NV], b[NV], lb[NV];   // lb: bias of the preceding Linear (added here instead of in the GEMM epilogue)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g[i] = *reinterpret_cast<const f32x4*>(gamma + cb + CM::off(i, lane));
        b[i] = *reinterpret_cast<const f32x4*>(beta + cb + CM::off(i, lane));
        lb[i] = bias ? *reinterpret_cast<const f32x4*>(bias + cb + CM::off(i, lane)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // the next row's loads are issued before the current row's reductions / epilogue: a wave always has a row in flight
    // (one row at a time left the kernel latency-bound at ~50 % of the HBM rate)
    f32x4 vn[NV];
    // Row order: grid-stride (row block b, b + grid, ...).  Round 5 tried one CONTIGUOUS range of rows per workgroup (-DMDL_ACT_CONTIG),
    // because a plain 1 read : 1 write float4 stream gains 15-25 % from it (tools/micro/hbm_rate.hip: 4.6-5.0 -> 5.8-6.1 TB/s at 8-16
    // workgroups per CU); in these kernels it is a null -- forward 0.436 / 0.441 vs 0.437 / 0.427 ms, backward 0.657 / 0.659 vs 0.637 /
    // 0.653 ms in a same-box A/B (profiles/r05d_contig_rows_and_dw_stream_ab.txt): they are not at the streaming ceiling the order moves.
    const ActRange rg = act_range(row_lo, rows, RPB);
    {
        const int64_t r = rg.base0 + slot;
        if (rg.base0 < rg.base1 && r < rows) row_load<IO, NV>(x + r * W + cb, lane, vn);
        else row_zero<NV>(vn);
    }
    for (int64_t base = rg.base0; base < rg.base1; base += rg.step) {
        const int64_t r = base + slot;
        const bool live = r < rows;
        f32x4 v[NV];
        float s = 0.f, dummy = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = vn[i] + lb[i];
        {
            const int64_t rn = r + rg.step;
            if (base + rg.step < rg.base1 && rn < rows) row_load<IO, NV>(x + rn * W + cb, lane, vn);
            else row_zero<NV>(vn);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        row_allreduce2<WPR>(s, dummy, red, slot * WPR, wv);
        const float mean = s * (1.f / W);
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 c = v[i] - mean;
            q += (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
        }
        row_allreduce2<WPR>(q, dummy, red, slot * WPR, wv);
        const float rstd = rsqrtf(q * (1.f / W) + eps);
        if (live) {
            IO* __restrict__ yr = y + r * W + cb;
            const int64_t rb = r * W;
            const uint32_t rkey = act_row_key(drop, rb);
            f32x4 prev = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int ci = cb + CM::off(i, lane);
                const f32x4 kp = act_keep4_t<DM>(drop, rkey, (uint32_t)rb + (uint32_t)ci, rb + ci);
                const f32x4 t = (v[i] - mean) * rstd * g[i] + b[i];
                const f32x4 o = act_gelu4<IO>(t) * kp;
                if (IMG != 1) group_store<IO, NV>(yr, lane, i, prev, o);
                if (IMG != 0) img_store4(img + r * (int64_t)(W * 4), ci, o, img_sc[0]);
                prev = o;
            }
            if (lane == 0 && seg == 0) {
                mean_o[r] = mean;
                rstd_o[r] = rstd;
            }
            rmax = fmaxf(rmax, row_mul ? rstd * row_mul[r] : rstd);
        }
    }
    // one atomic per WORKGROUP: at a few ten thousand rows every wave of the grid arrives here at once, and 8192 atomics on one address
    // took 77 us of a 31-us launch (tools/exp_ln_fwd_atomic.py)
    if (rstd_max) {   // kernel-uniform
        __shared__ float bmax[4];
        if (lane == 0) bmax[wv] = rmax;
        __syncthreads();
        if (threadIdx.x == 0)
            atomicMax(reinterpret_cast<unsigned int*>(rstd_max), __float_as_uint(fmaxf(fmaxf(bmax[0], bmax[1]), fmaxf(bmax[2], bmax[3]))));
    }
}

// IMG: dx is written as a split image (rows of 4 W bytes at `img`, scale img_sc[0]) instead of dx
template <int NV, int WPR, class IO, bool IMG = false, int DM = -1>
__global__ __launch_bounds__(ACT_BLOCK) void ln_gelu_drop_bwd_kernel(const IO* __restrict__ x,
                                                                     const float* __restrict__ bias,
                                                                     const float* __restrict__ gamma,
                                                                     const float* __restrict__ beta,
                                                                     const float* __restrict__ mean_i,
                                                                     const float* __restrict__ rstd_i,
                                                                     const IO* __restrict__ dy, IO* __restrict__ dx,
                                                                     float* __restrict__ part, int64_t rows_all, ActDrop drop,
                                                                     char* __restrict__ img = nullptr,
                                                                     const float* __restrict__ img_sc = nullptr,
                                                                     const float* __restrict__ row_mul = nullptr,
                                                                     const int64_t* __restrict__ cu = nullptr, int bias_gstride = 0) {
    // cu (grouped backward, round 5): rows [cu[g], cu[g + 1]) of group g = blockIdx.y are this workgroup's; its partial column sums
    // (slot blockIdx.y * gridDim.x + blockIdx.x) then belong to ONE group: the dbias third of them is the gradient of that group's bias row
    // bias_gstride != 0 (round 6): x does NOT hold the group's bias row yet (bf16 / exact-fp32 engines): it is bias + g * bias_gstride
    constexpr int W = NV * 256 * WPR, RPB = 4 / WPR;
    const int64_t row_lo = cu ? cu[blockIdx.y] : 0, rows = cu ? cu[blockIdx.y + 1] : rows_all;
    if (cu && bias) bias += (int64_t)blockIdx.y * bias_gstride;
    __shared__ float red[1][4][2];
    __shared__ float csum[WPR < 4 ? 3 * W : 1];  // WPR < 4: several waves own the same columns -> merged through LDS
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, slot = wv / WPR, seg = wv % WPR;
    typedef ColMap<IO, NV> CM;
    const int cb = seg * NV * 256;
    f32x4 g[NV], b[NV], lb[NV], sg[NV], sb[NV], sx[NV];   // sx: column sums of dx = gradient of the Linear's bias
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        g[i] = *reinterpret_cast<const f32x4*>(gamma + cb + CM::off(i, lane));
        b[i] = *reinterpret_cast<const f32x4*>(beta + cb + CM::off(i, lane));
        lb[i] = bias ? *reinterpret_cast<const f32x4*>(bias + cb + CM::off(i, lane)) : f32x4{0.f, 0.f, 0.f, 0.f};
        sg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        sb[i] = sg[i];
        sx[i] = sg[i];
    }
    f32x4 xn[NV], gn[NV];   // next row, prefetched (see the forward kernel)
    const ActRange rg = act_range(row_lo, rows, RPB);   // (row order: see the forward kernel)
    {
        const int64_t r = rg.base0 + slot;
        if (rg.base0 < rg.base1 && r < rows) {
            row_load<IO, NV>(x + r * W + cb, lane, xn);
            row_load<IO, NV>(dy + r * W + cb, lane, gn);
        } else {
            row_zero<NV>(xn);
            row_zero<NV>(gn);
        }
    }
    for (int64_t base = rg.base0; base < rg.base1; base += rg.step) {
        const int64_t r = base + slot;
        const bool live = r < rows;
        const float mean = live ? mean_i[r] : 0.f, rstd = live ? rstd_i[r] : 0.f;
        f32x4 xc[NV], gc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            xc[i] = xn[i];
            gc[i] = gn[i];
        }
        {
            const int64_t rn = r + rg.step;
            if (base + rg.step < rg.base1 && rn < rows) {
                row_load<IO, NV>(x + rn * W + cb, lane, xn);
                row_load<IO, NV>(dy + rn * W + cb, lane, gn);
            } else {
                row_zero<NV>(xn);
                row_zero<NV>(gn);
            }
        }
        f32x4 xh[NV], dxh[NV];
        float s1 = 0.f, s2 = 0.f;
        const int64_t rb = (live ? r : 0) * W;
        const uint32_t rkey = act_row_key(drop, rb);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const f32x4 xv = xc[i] + lb[i], gv = gc[i];
            const int ci = cb + CM::off(i, lane);
            const f32x4 kp = act_keep4_t<DM>(drop, rkey, (uint32_t)rb + (uint32_t)ci, rb + ci);
            const f32x4 h = (xv - mean) * rstd;
            const f32x4 t = h * g[i] + b[i];
            const f32x4 dt = gv * kp * act_gelu_grad4<IO>(t);  // d/d(LN output); rows past the end carry gv = 0
            sg[i] += dt * h;
            sb[i] += dt;
            const f32x4 dh = dt * g[i];
            xh[i] = h;
            dxh[i] = dh;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s1 += dh[e];
                s2 += dh[e] * h[e];
            }
        }
        row_allreduce2<WPR>(s1, s2, red, slot * WPR, wv);
        const float m1 = s1 * (1.f / W), m2 = s2 * (1.f / W);
        if (live) {
            IO* __restrict__ o = dx + r * W + cb;
            f32x4 prev = {0.f, 0.f, 0.f, 0.f};
            // image row factor: the common scale, times row_mul[r] when the image pairs (in the dW contraction over rows) with a
            // row-scaled image of this layer's input -- row_mul[r] = that image's 1 / s_r, so that the row factors cancel in the sum
            const float isc = IMG ? (row_mul ? img_sc[0] * row_mul[r] : img_sc[0]) : 0.f;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                f32x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = rstd * (dxh[i][e] - m1 - xh[i][e] * m2);
                sx[i] += v;
                if (IMG) img_store4(img + r * (int64_t)(W * 4), cb + CM::off(i, lane), v, isc);
                else group_store<IO, NV>(o, lane, i, prev, v);
                prev = v;
            }
        }
    }
    float* __restrict__ prow = part + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 * W;  // [block][dgamma W | dbeta W | dbias W]
    if (WPR == 4) {  // every wave owns its own column segment
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            *reinterpret_cast<f32x4*>(prow + cb + CM::off(i, lane)) = sg[i];
            *reinterpret_cast<f32x4*>(prow + W + cb + CM::off(i, lane)) = sb[i];
            *reinterpret_cast<f32x4*>(prow + 2 * W + cb + CM::off(i, lane)) = sx[i];
        }
    } else {  // waves (= rows) add into one LDS row in wave order (deterministic)
#pragma unroll
        for (int w = 0; w < ACT_BLOCK / 64; ++w) {
            if (wv == w) {
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    f32x4* pg = reinterpret_cast<f32x4*>(&csum[cb + CM::off(i, lane)]);
                    f32x4* pb = reinterpret_cast<f32x4*>(&csum[W + cb + CM::off(i, lane)]);
                    f32x4* px = reinterpret_cast<f32x4*>(&csum[2 * W + cb + CM::off(i, lane)]);
                    if (w < WPR) {   // first wave on this column segment
                        *pg = sg[i];
                        *pb = sb[i];
                        *px = sx[i];
                    } else {
                        *pg += sg[i];
                        *pb += sb[i];
                        *px += sx[i];
                    }
                }
            }
            __syncthreads();
        }
        for (int c = threadIdx.x; c < 3 * W; c += ACT_BLOCK) prow[c] = csum[c];
    }
}

// dgamma|dbeta|dbias[c] = sum over blocks of part[block][c]: 32 columns x 8 block-groups per workgroup (coalesced 128-B
// row segments, 8 loads in flight per thread), the 8 partial sums merged through LDS in a fixed order.
__global__ __launch_bounds__(256) void ln_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                        float* __restrict__ dbeta, float* __restrict__ dbias, int nblocks,
                                                        int W) {
    __shared__ float red[8][32];
    const int cl = threadIdx.x & 31, grp = threadIdx.x >> 5, c = blockIdx.x * 32 + cl;
    float v = 0.f;
    if (c < 3 * W) {
        int k = grp;
        for (; k + 56 < nblocks; k += 64) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = part[(int64_t)(k + 8 * u) * 3 * W + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) v += t[u];
        }
        for (; k < nblocks; k += 8) v += part[(int64_t)k * 3 * W + c];
    }
    red[grp][cl] = v;
    __syncthreads();
    if (grp == 0 && c < 3 * W) {
        float t = red[0][cl];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += red[g][cl];
        if (c < W) dgamma[c] = t;
        else if (c < 2 * W) dbeta[c - W] = t;
        else if (dbias) dbias[c - 2 * W] = t;
    }
}

// Rigorous output bounds -> image scales (one workgroup; W <= 4096).
//   forward : |GELU(LN(x)) keep| <= (max|gamma| sqrt(W - 1) + max|beta|) / (1 - p)       (|x^_i| <= sqrt(W - 1), |GELU(t)| <= |t|)
//   backward: |dx_i| = rstd |dh_i - mean(dh) - x^_i mean(dh x^)| <= rstd_max D (2 + sqrt(W)),  D = 1.13 max|gamma| max|dy| / (1 - p)
//             (|GELU'| <= 1.13, mean|x^| <= 1)
// sc[0] = scale, sc[1] = the bound.  aux[0] = max rstd, aux[1] = max |dy| (backward only).
__global__ __launch_bounds__(256) void ln_bound_scale_kernel(const float* __restrict__ gamma, const float* __restrict__ beta, int W,
                                                             float inv_keep, const float* __restrict__ aux, float* __restrict__ sc,
                                                             float* __restrict__ zero_me = nullptr,
                                                             const float* __restrict__ aux1 = nullptr) {
    // zero_me: a float this launch clears (the forward's rstd_max accumulator); aux1: max |dy| when it does not lie at aux[1]
    if (zero_me && threadIdx.x == 0) *zero_me = 0.f;
    __shared__ float red[2][4];
    float g = 0.f, b = 0.f;
    for (int c = threadIdx.x; c < W; c += 256) {
        g = fmaxf(g, fabsf(gamma[c]));
        b = fmaxf(b, fabsf(beta[c]));
    }
    g = wave_max(g);
    b = wave_max(b);
    if ((threadIdx.x & 63) == 0) {
        red[0][threadIdx.x >> 6] = g;
        red[1][threadIdx.x >> 6] = b;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        g = fmaxf(fmaxf(red[0][0], red[0][1]), fmaxf(red[0][2], red[0][3]));
        b = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
        const float bound = aux ? aux[0] * (1.13f * g * (aux1 ? aux1[0] : aux[1]) * inv_keep) * (2.f + sqrtf((float)W))
                                : (g * sqrtf((float)(W - 1)) + b) * inv_keep;
        sc[1] = bound;
        sc[0] = sp_scale_for(bound);
    }
}

// *out is raised to max_r |a[r] b[r]| (the caller zeroes it)
__global__ __launch_bounds__(256) void absmax_prod_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n,
                                                          float* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) m = fmaxf(m, fabsf(a[i] * b[i]));
    sp_atomic_absmax(out, m);
}

int sp_launch_absmax(const float* X, int64_t ldx, int64_t rows, int K, float* out, hipStream_t s);   // split_gemm.hip
int sp_launch_absmax_flat(const float* x, int64_t n, float* out, hipStream_t s);

static inline ActDrop make_act_drop(float p, uint64_t seed, const uint8_t* keep) {
    ActDrop d;
    d.on = p > 0.f ? 1 : 0;
    d.inv = d.on ? 1.f / (1.f - p) : 1.f;
    d.thr = drop_threshold(p);
    d.key = (uint32_t)(seed * 0x9E3779B97F4A7C15ULL >> 32) ^ (uint32_t)seed;
    d.keep = keep;
    return d;
}
static inline bool act_width_ok(int W) { return W == 256 || W == 512 || W == 1024 || W == 2048 || W == 4096; }
static inline int act_rpb(int W) { return W >= 2048 ? 1 : (W == 1024 ? 2 : 4); }  // rows per block iteration (= 4 / WPR)
static inline int act_blocks(int64_t rows, int W) {
    int64_t b = (rows + act_rpb(W) - 1) / act_rpb(W);
    if (b > 2048) b = 2048;  // 256 CUs x 8 blocks; grid-stride over the rest
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace mdl

using namespace mdl;

extern "C" int64_t mdl_ln_gelu_drop_bwd_ws_bytes(int64_t rows, int W) {
    if (rows < 0) return MDL_E_ARG;
    if (!act_width_ok(W)) return MDL_E_UNSUPPORTED;   // 512 * n_heads for n_heads in {1, 2, 4, 8}, and 256
    return (int64_t)act_blocks(rows, W) * 3 * W * 4 + 128;   // + aux floats of the split variant
}

#define MDL_DISPATCH_W(W, ...)                                                       \
    switch (W) {                                                                     \
        case 256: { constexpr int NV = 1, WPR = 1; __VA_ARGS__; } break;             \
        case 512: { constexpr int NV = 2, WPR = 1; __VA_ARGS__; } break;             \
        case 1024: { constexpr int NV = 2, WPR = 2; __VA_ARGS__; } break;            \
        case 2048: { constexpr int NV = 2, WPR = 4; __VA_ARGS__; } break;            \
        case 4096: { constexpr int NV = 4, WPR = 4; __VA_ARGS__; } break;            \
        default: return MDL_E_UNSUPPORTED;                                           \
    }

__global__ __launch_bounds__(256) void ln_group_bias_kernel(const float* __restrict__ part, float* __restrict__ dgb, int nbx, int W) {
    const int g = blockIdx.x;
    for (int c = threadIdx.x; c < W; c += 256) {
        float v = 0.f;
        for (int b = 0; b < nbx; ++b) v += part[((int64_t)g * nbx + b) * 3 * W + 2 * W + c];
        dgb[(int64_t)g * W + c] = v;
    }
}
static inline int ln_group_nbx(int64_t rows, int W, int G) {   // workgroups per group: G * nbx <= 2048 partial slots
    int nbx = 2048 / (G > 0 ? G : 1);
    const int64_t per = (rows / (G > 0 ? G : 1) + act_rpb(W) - 1) / act_rpb(W);   // no more than an average group has row blocks
    if (nbx > per) nbx = (int)per;
    return nbx < 1 ? 1 : nbx;
}

template <class IO>
static int ln_fwd_launch(const IO* x, const float* bias, const float* gamma, const float* beta, IO* y, float* mean, float* rstd, int64_t rows, int W,
                         float eps, float p_drop, uint64_t seed, const uint8_t* keep, void* stream, const int64_t* cu_groups = nullptr,
                         int G = 0) {
    if (!x || !gamma || !beta || !y || !mean || !rstd || rows < 0) return MDL_E_ARG;
    if (cu_groups && (G < 1 || G > 2048 || !bias)) return MDL_E_ARG;
    if (cu_groups && rows > 0) {   // grouped: nbx workgroups per group, the group's bias row (never the one-wave-per-row 2048-wide geometry)
        if (!(p_drop >= 0.f && p_drop < 1.f) || !(eps > 0.f)) return MDL_E_ARG;
        if (!host_aligned16(x) || !host_aligned16(y) || !host_aligned16(gamma) || !host_aligned16(beta) || !host_aligned16(bias)) return MDL_E_ALIGN;
        const ActDrop dg = make_act_drop(p_drop, seed, keep);
        const dim3 grid(ln_group_nbx(rows, W, G), G);
#define MDL_LN_FWD_G(DMV)                                                                                                                    \
    hipLaunchKernelGGL((ln_gelu_drop_fwd_kernel<NV, WPR, IO, 0, DMV>), grid, dim3(ACT_BLOCK), 0, (hipStream_t)stream, x, bias, gamma, beta, y, mean, \
                       rstd, rows, eps, dg, (char*)nullptr, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, cu_groups, W)
        const int dmg = ln_dm<IO>(dg);   // (dropout mode fixed at compile time where the un-grouped launch does so)
        MDL_DISPATCH_W(W, {
            if (dmg < 0) MDL_LN_FWD_G(-1);
            else if (dmg == 0) MDL_LN_FWD_G(0);
            else if (dmg == 1) MDL_LN_FWD_G(1);
            else MDL_LN_FWD_G(2);
            MDL_LAUNCH_CHECK();
        });
#undef MDL_LN_FWD_G
        return MDL_OK;
    }
    if (!(p_drop >= 0.f && p_drop < 1.f) || !(eps > 0.f)) return MDL_E_ARG;
    if (!host_aligned16(x) || !host_aligned16(y) || !host_aligned16(gamma) || !host_aligned16(beta) || !host_aligned16(bias))
        return MDL_E_ALIGN;
    if (rows == 0) return MDL_OK;
    const ActDrop d = make_act_drop(p_drop, seed, keep);
// (geometry of the 2048-wide kernels as macros: round 5 swept forward <4,2> / <2,4>, backward <8,1> / <2,4> and grid caps 768 / 1024 on fp32
// and bf16 storage -- the shipped <8,1> / <4,2> / 2048 is the best or within noise everywhere, profiles/r05i_ln_2048_geometry_variants.txt;
// the bf16 kernels (then 4.0 / 3.9 TB/s) were bound by their VALU work, not by the row geometry: see act_gelu4 (polynomial GELU, round 5:
// 2048-wide forward 4.7 TB/s, backward 4.5; the sweep repeated after that change confirms <8,1> / <4,2>, profiles/r05x_*)
#ifndef MDL_LN_F_NV
#define MDL_LN_F_NV 8
#define MDL_LN_F_WPR 1
#endif
#ifndef MDL_LN_GRID_CAP
#define MDL_LN_GRID_CAP 2048
#endif
    if (W == 2048) {  // forward: a whole 2048-wide row per wave (153 VGPRs, no block barriers) beats 4 waves per row
        int64_t nb = (rows + 4 / MDL_LN_F_WPR - 1) / (4 / MDL_LN_F_WPR);
        if (nb > MDL_LN_GRID_CAP) nb = MDL_LN_GRID_CAP;
#define MDL_LN_FWD(NVV, WPRV, DMV)                                                                                                  \
    hipLaunchKernelGGL((ln_gelu_drop_fwd_kernel<NVV, WPRV, IO, 0, DMV>), dim3((unsigned)nb), dim3(ACT_BLOCK), 0, (hipStream_t)stream, x, \
                       bias, gamma, beta, y, mean, rstd, rows, eps, d)
#define MDL_LN_FWD_DM(NVV, WPRV)                  \
    do {                                          \
        if (dm < 0) MDL_LN_FWD(NVV, WPRV, -1);    \
        else if (dm == 0) MDL_LN_FWD(NVV, WPRV, 0); \
        else if (dm == 1) MDL_LN_FWD(NVV, WPRV, 1); \
        else MDL_LN_FWD(NVV, WPRV, 2);            \
    } while (0)
        const int dm = ln_dm<IO>(d);
        MDL_LN_FWD_DM(MDL_LN_F_NV, MDL_LN_F_WPR);
        MDL_LAUNCH_CHECK();
        return MDL_OK;
    }
    MDL_DISPATCH_W(W, {
        const int nb = act_blocks(rows, W);
        const int dm = ln_dm<IO>(d);
        MDL_LN_FWD_DM(NV, WPR);
        MDL_LAUNCH_CHECK();
    });
#undef MDL_LN_FWD_DM
#undef MDL_LN_FWD
    return MDL_OK;
}

template <class IO>
static int ln_bwd_launch(const IO* x, const float* bias, const float* gamma, const float* beta, const float* mean, const float* rstd,
                         const IO* dy, IO* dx, float* dgamma, float* dbeta, float* dbias, int64_t rows, int W, float p_drop, uint64_t seed,
                         const uint8_t* keep, void* ws, void* stream, const int64_t* cu_groups = nullptr, int G = 0,
                         float* dgroup_bias = nullptr) {
    if (!x || !gamma || !beta || !mean || !rstd || !dy || !dx || !dgamma || !dbeta || !ws || rows < 0) return MDL_E_ARG;
    if (cu_groups) {   // grouped: per-group bias rows in, per-group bias gradients out (dbias = their sum over the groups)
        if (G < 1 || G > 2048 || !bias || !dgroup_bias) return MDL_E_ARG;
        if (!(p_drop >= 0.f && p_drop < 1.f)) return MDL_E_ARG;
        if (!host_aligned16(x) || !host_aligned16(dy) || !host_aligned16(dx) || !host_aligned16(gamma) || !host_aligned16(beta) ||
            !host_aligned16(bias))
            return MDL_E_ALIGN;
        hipStream_t sg = (hipStream_t)stream;
        const ActDrop dd = make_act_drop(p_drop, seed, keep);
        const int nbx = ln_group_nbx(rows, W, G);
        const dim3 grid(nbx, G);
#define MDL_LN_BWD_G(DMV)                                                                                                                    \
    hipLaunchKernelGGL((ln_gelu_drop_bwd_kernel<NV, WPR, IO, false, DMV>), grid, dim3(ACT_BLOCK), 0, sg, x, bias, gamma, beta, mean, rstd, dy, dx, \
                       (float*)ws, rows, dd, (char*)nullptr, (const float*)nullptr, (const float*)nullptr, cu_groups, W)
        const int dmg = ln_dm<IO>(dd);
        MDL_DISPATCH_W(W, {
            if (dmg < 0) MDL_LN_BWD_G(-1);
            else if (dmg == 0) MDL_LN_BWD_G(0);
            else if (dmg == 1) MDL_LN_BWD_G(1);
            else MDL_LN_BWD_G(2);
            MDL_LAUNCH_CHECK();
        });
#undef MDL_LN_BWD_G
        hipLaunchKernelGGL(ln_reduce_kernel, dim3((3 * W + 31) / 32), dim3(256), 0, sg, (const float*)ws, dgamma, dbeta, dbias, nbx * G, W);
        MDL_LAUNCH_CHECK();
        hipLaunchKernelGGL(ln_group_bias_kernel, dim3(G), dim3(256), 0, sg, (const float*)ws, dgroup_bias, nbx, W);
        MDL_LAUNCH_CHECK();
        return MDL_OK;
    }
    if (!(p_drop >= 0.f && p_drop < 1.f)) return MDL_E_ARG;
    if (!host_aligned16(x) || !host_aligned16(dy) || !host_aligned16(dx) || !host_aligned16(gamma) || !host_aligned16(beta) ||
        !host_aligned16(bias))
        return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const ActDrop d = make_act_drop(p_drop, seed, keep);
    int nb = rows > 0 ? act_blocks(rows, W) : 0;
    if (W == 2048 && nb > 0) {
        // 2048 wide: 2 waves per row (NV = 4, ~226 VGPRs) instead of 4 -- the 4-wave version was bound by its per-row block barriers
        // rather than by VALU or HBM: bf16 0.97 -> 0.82 ms, fp32 1.35 -> 1.25 ms at config 2 (tools/exp_ln.py)
#ifndef MDL_LN_B_NV
#define MDL_LN_B_NV 4
#define MDL_LN_B_WPR 2
#endif
        int64_t b2 = (rows + 4 / MDL_LN_B_WPR - 1) / (4 / MDL_LN_B_WPR);
        if (b2 > MDL_LN_GRID_CAP) b2 = MDL_LN_GRID_CAP;
        nb = (int)b2;
#define MDL_LN_BWD(NVV, WPRV, DMV)                                                                                                  \
    hipLaunchKernelGGL((ln_gelu_drop_bwd_kernel<NVV, WPRV, IO, false, DMV>), dim3(nb), dim3(ACT_BLOCK), 0, s, x, bias, gamma, beta, mean, \
                       rstd, dy, dx, (float*)ws, rows, d)
#define MDL_LN_BWD_DM(NVV, WPRV)                  \
    do {                                          \
        if (dm < 0) MDL_LN_BWD(NVV, WPRV, -1);    \
        else if (dm == 0) MDL_LN_BWD(NVV, WPRV, 0); \
        else if (dm == 1) MDL_LN_BWD(NVV, WPRV, 1); \
        else MDL_LN_BWD(NVV, WPRV, 2);            \
    } while (0)
        const int dm = ln_dm<IO>(d);
        MDL_LN_BWD_DM(MDL_LN_B_NV, MDL_LN_B_WPR);
        MDL_LAUNCH_CHECK();
    } else
    MDL_DISPATCH_W(W, {
        if (nb > 0) {
            const int dm = ln_dm<IO>(d);
            MDL_LN_BWD_DM(NV, WPR);
            MDL_LAUNCH_CHECK();
        }
    });
#undef MDL_LN_BWD_DM
#undef MDL_LN_BWD
    hipLaunchKernelGGL(ln_reduce_kernel, dim3((3 * W + 31) / 32), dim3(256), 0, s, (const float*)ws, dgamma, dbeta, dbias, nb, W);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

extern "C" int mdl_ln_gelu_drop_fwd(const float* x, const float* bias, const float* gamma, const float* beta, float* y,
                                    float* mean, float* rstd, int64_t rows, int W, float eps, float p_drop, uint64_t seed,
                                    const uint8_t* keep, void* stream) {
    return ln_fwd_launch<float>(x, bias, gamma, beta, y, mean, rstd, rows, W, eps, p_drop, seed, keep, stream);
}

extern "C" int mdl_ln_gelu_drop_bwd(const float* x, const float* bias, const float* gamma, const float* beta, const float* mean,
                                    const float* rstd, const float* dy, float* dx, float* dgamma, float* dbeta, float* dbias,
                                    int64_t rows, int W, float p_drop, uint64_t seed, const uint8_t* keep, void* ws,
                                    void* stream) {
    return ln_bwd_launch<float>(x, bias, gamma, beta, mean, rstd, dy, dx, dgamma, dbeta, dbias, rows, W, p_drop, seed, keep, ws,
                                stream);
}

extern "C" int mdl_ln_gelu_drop_fwd_bf16(const uint16_t* x, const float* bias, const float* gamma, const float* beta,
                                         uint16_t* y, float* mean, float* rstd, int64_t rows, int W, float eps, float p_drop,
                                         uint64_t seed, const uint8_t* keep, void* stream) {
    return ln_fwd_launch<bf16_t>((const bf16_t*)x, bias, gamma, beta, (bf16_t*)y, mean, rstd, rows, W, eps, p_drop, seed, keep,
                                 stream);
}

extern "C" int mdl_ln_gelu_drop_bwd_bf16(const uint16_t* x, const float* bias, const float* gamma, const float* beta,
                                         const float* mean, const float* rstd, const uint16_t* dy, uint16_t* dx, float* dgamma,
                                         float* dbeta, float* dbias, int64_t rows, int W, float p_drop, uint64_t seed,
                                         const uint8_t* keep, void* ws, void* stream) {
    return ln_bwd_launch<bf16_t>((const bf16_t*)x, bias, gamma, beta, mean, rstd, (const bf16_t*)dy, (bf16_t*)dx, dgamma, dbeta,
                                 dbias, rows, W, p_drop, seed, keep, ws, stream);
}

/* Grouped forms for the engines whose GEMM epilogue adds no bias (exact fp32, bf16): rows [cu_groups[g], cu_groups[g + 1]) of group g take
 * the bias row group_bias[g][W] (the bias of the preceding Linear + the bag's stain-encoding row times the encoding columns of the weight,
 * Model.py:125-132, :351: [x | e_g] W^T = x Wx^T + e_g We^T); the backward returns dgroup_bias [G][W] (per-group column sums of dx, merged
 * in a fixed order).  ws of the backward: mdl_ln_gelu_drop_bwd_groups_ws_bytes(rows, W, G).  W <= 1024 (the first block is 512 wide). */
extern "C" int mdl_ln_gelu_drop_fwd_groups(const float* x, const float* group_bias, const float* gamma, const float* beta, float* y, float* mean,
                                           float* rstd, int64_t rows, int W, float eps, float p_drop, uint64_t seed, const uint8_t* keep,
                                           const int64_t* cu_groups, int G, void* stream) {
    if (!cu_groups) return MDL_E_ARG;
    if (W > 1024) return MDL_E_UNSUPPORTED;
    return ln_fwd_launch<float>(x, group_bias, gamma, beta, y, mean, rstd, rows, W, eps, p_drop, seed, keep, stream, cu_groups, G);
}
extern "C" int mdl_ln_gelu_drop_fwd_groups_bf16(const uint16_t* x, const float* group_bias, const float* gamma, const float* beta, uint16_t* y,
                                                float* mean, float* rstd, int64_t rows, int W, float eps, float p_drop, uint64_t seed,
                                                const uint8_t* keep, const int64_t* cu_groups, int G, void* stream) {
    if (!cu_groups) return MDL_E_ARG;
    if (W > 1024) return MDL_E_UNSUPPORTED;
    return ln_fwd_launch<bf16_t>((const bf16_t*)x, group_bias, gamma, beta, (bf16_t*)y, mean, rstd, rows, W, eps, p_drop, seed, keep, stream,
                                 cu_groups, G);
}
extern "C" int mdl_ln_gelu_drop_bwd_groups(const float* x, const float* group_bias, const float* gamma, const float* beta, const float* mean,
                                           const float* rstd, const float* dy, float* dx, float* dgamma, float* dbeta, float* dgroup_bias,
                                           int64_t rows, int W, float p_drop, uint64_t seed, const uint8_t* keep, const int64_t* cu_groups, int G,
                                           void* ws, void* stream) {
    if (!cu_groups) return MDL_E_ARG;
    if (W > 1024) return MDL_E_UNSUPPORTED;
    return ln_bwd_launch<float>(x, group_bias, gamma, beta, mean, rstd, dy, dx, dgamma, dbeta, (float*)nullptr, rows, W, p_drop, seed, keep, ws,
                                stream, cu_groups, G, dgroup_bias);
}
extern "C" int mdl_ln_gelu_drop_bwd_groups_bf16(const uint16_t* x, const float* group_bias, const float* gamma, const float* beta,
                                                const float* mean, const float* rstd, const uint16_t* dy, uint16_t* dx, float* dgamma,
                                                float* dbeta, float* dgroup_bias, int64_t rows, int W, float p_drop, uint64_t seed,
                                                const uint8_t* keep, const int64_t* cu_groups, int G, void* ws, void* stream) {
    if (!cu_groups) return MDL_E_ARG;
    if (W > 1024) return MDL_E_UNSUPPORTED;
    return ln_bwd_launch<bf16_t>((const bf16_t*)x, group_bias, gamma, beta, mean, rstd, (const bf16_t*)dy, (bf16_t*)dx, dgamma, dbeta,
                                 (float*)nullptr, rows, W, p_drop, seed, keep, ws, stream, cu_groups, G, dgroup_bias);
}

/* mdl_ln_gelu_drop_fwd whose output is written as a SPLIT IMAGE (rows of 4 W bytes at img; scale[2] = {scale, bound} from the
 * parameters: |y| <= (max|gamma| sqrt(W-1) + max|beta|) / (1-p)) -- and, when y != NULL, as fp32 as well. */
extern "C" int mdl_ln_gelu_drop_fwd_split(const float* x, const float* bias, const float* gamma, const float* beta, float* y, void* img,
                                          float* scale, float* mean, float* rstd, int64_t rows, int W, float eps, float p_drop,
                                          uint64_t seed, const uint8_t* keep, const float* row_mul, float* rstd_max, void* stream) {
    if (!x || !gamma || !beta || !img || !scale || !mean || !rstd || rows < 0) return MDL_E_ARG;
    if (!(p_drop >= 0.f && p_drop < 1.f) || !(eps > 0.f)) return MDL_E_ARG;
    if (!act_width_ok(W) || W > 4096) return MDL_E_UNSUPPORTED;
    if (!host_aligned16(x) || !host_aligned16(y) || !host_aligned16(img) || !host_aligned16(gamma) || !host_aligned16(beta) ||
        !host_aligned16(bias))
        return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const ActDrop d = make_act_drop(p_drop, seed, keep);
    hipLaunchKernelGGL(ln_bound_scale_kernel, dim3(1), dim3(256), 0, s, gamma, beta, W, d.inv, (const float*)nullptr, scale, rstd_max,
                       (const float*)nullptr);
    MDL_LAUNCH_CHECK();
    if (rows == 0) return MDL_OK;
#define MDL_LN_FWD_IMG(NVV, WPRV, NB)                                                                                                    \
    do {                                                                                                                               \
        if (y) hipLaunchKernelGGL((ln_gelu_drop_fwd_kernel<NVV, WPRV, float, 2>), dim3((unsigned)(NB)), dim3(ACT_BLOCK), 0, s, x, bias, \
                                  gamma, beta, y, mean, rstd, rows, eps, d, (char*)img, (const float*)scale, row_mul, rstd_max);        \
        else if (MDL_LN_IMG_DM_FWD && act_drop_mode(d) == 1)                                                                            \
            hipLaunchKernelGGL((ln_gelu_drop_fwd_kernel<NVV, WPRV, float, 1, MDL_LN_IMG_DM_FWD ? 1 : -1>), dim3((unsigned)(NB)), dim3(ACT_BLOCK), 0, \
                               s, x, bias, gamma, beta, y, mean, rstd, rows, eps, d, (char*)img, (const float*)scale, row_mul, rstd_max); \
        else hipLaunchKernelGGL((ln_gelu_drop_fwd_kernel<NVV, WPRV, float, 1>), dim3((unsigned)(NB)), dim3(ACT_BLOCK), 0, s, x, bias,   \
                                gamma, beta, y, mean, rstd, rows, eps, d, (char*)img, (const float*)scale, row_mul, rstd_max);          \
    } while (0)
    if (W == 2048) {
        int64_t nb = (rows + 3) / 4;
        if (nb > 2048) nb = 2048;
        MDL_LN_FWD_IMG(8, 1, nb);
    } else {
        MDL_DISPATCH_W(W, { MDL_LN_FWD_IMG(NV, WPR, act_blocks(rows, W)); });
    }
#undef MDL_LN_FWD_IMG
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

/* mdl_ln_gelu_drop_bwd whose dx is written as a SPLIT IMAGE followed by 32 all-zero rows (the B operand of mdl_split_gemm_tn);
 * dx_scale[2] = {scale, bound} from the rigorous bound rstd_max 1.13 max|gamma| max|dy| (2 + sqrt(W)) / (1-p).  dy_absmax: device
 * float holding max |dy| (e.g. from the epilogue of the kernel that produced dy), or NULL: computed here by one pass over dy.
 * ws: mdl_ln_gelu_drop_bwd_ws_bytes(rows, W) + 64 bytes. */
// gradient of the group bias rows: dgb[g][c] = sum over the nbx partial slots of group g of their dbias third
static int ln_bwd_split_impl(const float* x, const float* bias, const float* gamma, const float* beta, const float* mean,
                             const float* rstd, const float* dy, const float* dy_absmax, void* dx_img, float* dx_scale,
                             float* dgamma, float* dbeta, float* dbias, int64_t rows, int W, float p_drop, uint64_t seed,
                             const uint8_t* keep, const float* row_mul, const float* rstd_max, void* ws, void* stream,
                             const int64_t* cu_groups, int G, float* dgroup_bias) {
    if (!x || !gamma || !beta || !mean || !rstd || !dy || !dx_img || !dx_scale || !dgamma || !dbeta || !ws || rows < 0) return MDL_E_ARG;
    if (cu_groups && (G < 1 || G > 2048 || !dgroup_bias)) return MDL_E_ARG;
    if (!(p_drop >= 0.f && p_drop < 1.f)) return MDL_E_ARG;
    if (!act_width_ok(W) || W > 4096) return MDL_E_UNSUPPORTED;
    if (!host_aligned16(x) || !host_aligned16(dy) || !host_aligned16(dx_img) || !host_aligned16(gamma) || !host_aligned16(beta) ||
        !host_aligned16(bias) || !host_aligned16(ws))
        return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const ActDrop d = make_act_drop(p_drop, seed, keep);
    int nb = rows > 0 ? act_blocks(rows, W) : 0;
    if (W == 2048 && nb > 0) {
        int64_t b2 = (rows + 1) / 2;
        if (b2 > 2048) b2 = 2048;
        nb = (int)b2;
    }
    int slots = act_blocks(rows, W);
    dim3 grid(nb > 0 ? nb : 1);
    if (cu_groups) {   // nbx workgroups per group, one partial slot each: [G][nbx][3 W]
        const int nbx = ln_group_nbx(rows, W, G);
        grid = dim3(nbx, G);
        nb = nbx * G;
        slots = slots > nb ? slots : nb;
    }
    float* part = (float*)ws;
    float* aux = (float*)((char*)ws + (((int64_t)slots * 3 * W * 4 + 15) & ~(int64_t)15));   // [0] max rstd, [1] max |dy|
    hipError_t e = hipMemsetAsync((char*)dx_img + rows * (int64_t)W * 4, 0, (size_t)32 * W * 4, s);
    if (e != hipSuccess) return (int)e;
    int rc = MDL_OK;
    if (rstd_max && dy_absmax) {
        // both factors of the bound were published by the kernels that produced them (the forward: max rstd * row_mul; the producer of
        // dy: max |dy|): no memset, no pass over rstd, no copy -- the bound kernel reads them where they lie
        hipLaunchKernelGGL(ln_bound_scale_kernel, dim3(1), dim3(256), 0, s, gamma, beta, W, d.inv, rstd_max, dx_scale, (float*)nullptr,
                           dy_absmax);
        MDL_LAUNCH_CHECK();
    } else {
        e = hipMemsetAsync(aux, 0, 2 * sizeof(float), s);
        if (e != hipSuccess) return (int)e;
        if (rstd_max) {
            e = hipMemcpyAsync(aux, rstd_max, sizeof(float), hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return (int)e;
        } else if (row_mul && rows > 0) {   // the image holds row_mul[r] dx[r][:]: bound through max_r rstd[r] row_mul[r]
            int nbm = (int)((rows + 2047) / 2048);
            if (nbm > 2048) nbm = 2048;
            hipLaunchKernelGGL(absmax_prod_kernel, dim3(nbm), dim3(256), 0, s, rstd, row_mul, rows, aux);
            MDL_LAUNCH_CHECK();
        } else {
            rc = sp_launch_absmax_flat(rstd, rows, aux, s);
        }
        if (rc) return rc;
        if (dy_absmax) {
            e = hipMemcpyAsync(aux + 1, dy_absmax, sizeof(float), hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) return (int)e;
        } else {
            rc = sp_launch_absmax(dy, W, rows, W, aux + 1, s);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(ln_bound_scale_kernel, dim3(1), dim3(256), 0, s, gamma, beta, W, d.inv, (const float*)aux, dx_scale, (float*)nullptr,
                           (const float*)nullptr);
        MDL_LAUNCH_CHECK();
    }
    const bool hash_dm = MDL_LN_IMG_DM && act_drop_mode(d) == 1;   // (A/B: the counter-hash mode fixed at compile time)
#define MDL_LN_BWD_IMG(NVV, WPRV)                                                                                                       \
    do {                                                                                                                                \
        if (hash_dm)                                                                                                                    \
            hipLaunchKernelGGL((ln_gelu_drop_bwd_kernel<NVV, WPRV, float, true, MDL_LN_IMG_DM ? 1 : -1>), grid, dim3(ACT_BLOCK), 0, s, x, bias, \
                               gamma, beta, mean, rstd, dy, (float*)nullptr, part, rows, d, (char*)dx_img, (const float*)dx_scale, row_mul, \
                               cu_groups);                                                                                              \
        else                                                                                                                            \
            hipLaunchKernelGGL((ln_gelu_drop_bwd_kernel<NVV, WPRV, float, true>), grid, dim3(ACT_BLOCK), 0, s, x, bias, gamma, beta, mean, \
                               rstd, dy, (float*)nullptr, part, rows, d, (char*)dx_img, (const float*)dx_scale, row_mul, cu_groups);    \
    } while (0)
    if (W == 2048 && nb > 0) {
        MDL_LN_BWD_IMG(4, 2);
        MDL_LAUNCH_CHECK();
    } else
    MDL_DISPATCH_W(W, {
        if (nb > 0) {
            MDL_LN_BWD_IMG(NV, WPR);
            MDL_LAUNCH_CHECK();
        }
    });
#undef MDL_LN_BWD_IMG
    hipLaunchKernelGGL(ln_reduce_kernel, dim3((3 * W + 31) / 32), dim3(256), 0, s, (const float*)part, dgamma, dbeta, dbias, nb, W);
    MDL_LAUNCH_CHECK();
    if (cu_groups) {
        hipLaunchKernelGGL(ln_group_bias_kernel, dim3(G), dim3(256), 0, s, (const float*)part, dgroup_bias, nb / G, W);
        MDL_LAUNCH_CHECK();
    }
    return MDL_OK;
}

extern "C" int mdl_ln_gelu_drop_bwd_split(const float* x, const float* bias, const float* gamma, const float* beta, const float* mean,
                                          const float* rstd, const float* dy, const float* dy_absmax, void* dx_img, float* dx_scale,
                                          float* dgamma, float* dbeta, float* dbias, int64_t rows, int W, float p_drop, uint64_t seed,
                                          const uint8_t* keep, const float* row_mul, const float* rstd_max, void* ws, void* stream) {
    return ln_bwd_split_impl(x, bias, gamma, beta, mean, rstd, dy, dy_absmax, dx_img, dx_scale, dgamma, dbeta, dbias, rows, W, p_drop, seed, keep,
                             row_mul, rstd_max, ws, stream, nullptr, 0, nullptr);
}

/* The same for a pre-LN tensor that carries a per-GROUP bias row (x[r] = product[r] + group_bias[g(r)], rows of a group contiguous:
 * cu_groups int64 [G + 1] on the device, G <= 2048): additionally dgroup_bias [G][W] = sum over the rows of each group of dx -- the
 * gradient of the group rows (MADELEINE's stain-encoding columns folded out of the first Linear: Model.py:125-132, :351).  The launch
 * is organised by group (rows of one group per workgroup), so the per-group sums are the existing per-workgroup dbias partials, merged
 * in a fixed order.  ws: mdl_ln_gelu_drop_bwd_groups_ws_bytes(rows, W, G). */
extern "C" int64_t mdl_ln_gelu_drop_bwd_groups_ws_bytes(int64_t rows, int W, int G) {
    if (rows < 0 || G < 1 || G > 2048) return MDL_E_ARG;
    if (!act_width_ok(W)) return MDL_E_UNSUPPORTED;
    int64_t slots = act_blocks(rows, W);
    const int64_t nb = (int64_t)ln_group_nbx(rows, W, G) * G;
    if (nb > slots) slots = nb;
    return slots * 3 * W * 4 + 128;
}
extern "C" int mdl_ln_gelu_drop_bwd_split_groups(const float* x, const float* bias, const float* gamma, const float* beta, const float* mean,
                                                 const float* rstd, const float* dy, const float* dy_absmax, void* dx_img, float* dx_scale,
                                                 float* dgamma, float* dbeta, float* dbias, int64_t rows, int W, float p_drop, uint64_t seed,
                                                 const uint8_t* keep, const float* row_mul, const float* rstd_max, const int64_t* cu_groups,
                                                 int G, float* dgroup_bias, void* ws, void* stream) {
    if (!cu_groups) return MDL_E_ARG;
    return ln_bwd_split_impl(x, bias, gamma, beta, mean, rstd, dy, dy_absmax, dx_img, dx_scale, dgamma, dbeta, dbias, rows, W, p_drop, seed, keep,
                             row_mul, rstd_max, ws, stream, cu_groups, G, dgroup_bias);
}
