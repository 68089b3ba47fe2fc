// common.hpp -- shared device helpers for the gfx950 kernels of libmadeleine_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/madeleine_amd.h"

#define MDL_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t _e = hipGetLastError();                  \
        if (_e != hipSuccess) return (int)_e;               \
    } while (0)

namespace mdl {

constexpr int WAVE = 64;
constexpr int HID = MDL_HIDDEN;  // 512

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 4 consecutive elements <-> f32x4 for the two storage types of the activation tensors (fp32 parity mode, bf16 mode:
// v_cvt_pk_bf16_f32 round-to-nearest-even on store, a 16-bit shift on load).  p must be 16-B (fp32) / 8-B (bf16) aligned.
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const bf16_t* p) { return __builtin_convertvector(*reinterpret_cast<const bf16x4*>(p), f32x4); }
__device__ __forceinline__ f32x4 ld4_nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); }
__device__ __forceinline__ f32x4 ld4_nt(const bf16_t* p) {
    return __builtin_convertvector(__builtin_nontemporal_load(reinterpret_cast<const bf16x4*>(p)), f32x4);
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, f32x4 v) { *reinterpret_cast<bf16x4*>(p) = __builtin_convertvector(v, bf16x4); }

__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool host_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- wave-level reductions (wave = 64 lanes) ---------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// The same sum on the DPP network (no LDS crossbar: __shfl_xor is ds_bpermute_b32, ~100 cycles of latency per step and a dependent
// chain of six): two quad permutes + the half-row and row mirrors give every lane the sum of its row of 16, four v_readlane + three adds
// the wave total (uniform).  For latency-bound reductions inside serial sweeps (the GOT / IPOT row sums).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);   // row_half_mirror
    v += dpp_mov<0x140>(v);   // row_mirror
    const int b = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---- counter-based dropout RNG ------------------------------------------------------------------
// keep(idx) is a pure function of (seed, stream, idx): forward and backward regenerate identical
// masks without storing them.  Two rounds of a xorshift-multiply finaliser over a 64-bit counter folded with the seed.
// The multiplies are 24-bit (full-rate VALU; v_mul_lo_u32 is quarter rate and two of them were a sixth of the gate epilogue's issue
// time) in the multiply-ADD form x <- lo24(x) * K + x (one v_mad_u32_u24, K even): that equals lo24(x) * (K + 1) + (x & 0xFF000000)
// mod 2^32, which is INJECTIVE on 32-bit counters -- K + 1 is odd, so the low 24 bits of the result determine lo24(x), and the top byte
// of x then follows from the top byte of the result.  (Round 4's plain v_mul_u32_u24 dropped bits 24-31 after the first fold: idx and
// idx ^ (d << 24 | d << 8) hashed alike, i.e. keep masks repeated between token rows of any activation tensor above 2^24 elements --
// ADVICE round 4.)  Constants chosen by tools/rng_quality.py --search over the worst keep-mask correlation under every 1- and 2-bit
// input difference and the (d << 24 | d << 8) family: 0.22 (lowbias32, the two-multiply finaliser used before round 4: 0.27); the
// same tool reports rate, lag / a-b / adjacent-key correlations, bucket chi-square at n = 2^25 and the distinct-hash count over 2^26
// consecutive counters (100 %).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16;
    x = __umul24(x, 0x58E58AU) + x;
    x ^= x >> 13;
    x = __umul24(x, 0xCA6D40U) + x;
    x ^= x >> 16;
    return x;
}
// 32 random bits for element idx: one lowbias32 round over the low word keyed by (seed, high word).  The two
// 16-bit halves drive the two Bernoulli draws (tanh branch / sigmoid branch) of a gate element.  key = host-side mix of the seed.
__device__ __forceinline__ uint32_t rng_u32(uint32_t key, uint64_t idx) {
    const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    return mix32(lo ^ key ^ (hi * 0x9E3779B9U));
}
// threshold = round(p * 2^16); keep iff u16 >= threshold  => P(keep) = 1 - thr/65536 (exact for p = 0.25)
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
    double t = (double)p * 65536.0 + 0.5;
    if (t < 0) t = 0;
    if (t > 65535.0) t = 65535.0;
    return (uint32_t)t;
}

// activations of the gate epilogue: v_exp_f32 / v_rcp_f32 based, absolute error ~2e-7 (the libm forms cost ~5x the
// instructions and the epilogue is VALU-bound beside the MFMA stream)
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
// The same two activations of a pre-activation z = acc * s + bias with the scaling of the exponent folded into the FMA that forms z
// (round 5: the gate epilogue is VALU-bound beside the matrix cores, every instruction removed is wall time):
//   tanh(z)    = 1 - 2 / (1 + 2^(acc * (2 log2e s) + 2 log2e bias))       -> gate_tanh_pre(acc, 2 log2e s, 2 log2e bias)
//   sigmoid(z) = 1 / (1 + 2^(acc * (-log2e s) - log2e bias))              -> gate_sigmoid_pre(acc, -log2e s, -log2e bias)
// one fma + v_exp_f32 + add + v_rcp_f32 (+ fma): 5 and 4 instructions instead of 7 and 6.
constexpr float MDL_LOG2E = 1.4426950408889634f;
__device__ __forceinline__ float gate_tanh_pre(float acc, float s2, float b2) {
    return fmaf(-2.f, __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(acc, s2, b2))), 1.f);
}
__device__ __forceinline__ float gate_sigmoid_pre(float acc, float s1, float b1) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(fmaf(acc, s1, b1)));
}

// XCD-aware remap (8 XCDs; block b runs on XCD b % 8): returns a logical id such that logical ids
// [x*per, (x+1)*per) all run on XCD x, i.e. consecutive logical tiles share one L2.  Bijective for
// any n (cdna_hip_programming.md T1).
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n / 8, r = n % 8, xcd = bid % 8, pos = bid / 8;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + pos;
}

}  // namespace mdl
