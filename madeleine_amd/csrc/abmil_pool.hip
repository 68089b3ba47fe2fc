// abmil_pool.hip -- A3: softmax over patches + attention-weighted pooling (forward, backward).
//
// Replaces  F.softmax(A, dim=1)                       (reference madeleine/models/abmil.py:55)
//      and  (embeddings * attention).sum(dim=1)       (reference madeleine/models/Model.py:416-417)
// without materialising the [BM,N,512,H] product.  HBM-bound: the forward reads E once
// (8 KiB + 16 B per token at H=4) and writes 8 KiB per BAG; the backward reads E once and writes dE once.
//
// Layout: head-major E [T, H*512] (see include/madeleine_amd.h), scores [T,H], pooled [n_bags,H*512].
//
// Forward  = pool_partial (grid: chunks of 128 tokens x bags; online-softmax partial per chunk)
//          + pool_combine (grid: bags; merges the per-chunk (max, sum, weighted sum) triples in chunk
//            order -> deterministic, no atomics).
// Backward = pool_bwd (one wave per token row; lane-local dot with d_pooled, one 64-lane reduction
//            per head, dE and d_scores written in the same pass).
#include <hip/hip_ext.h>
#include <type_traits>

#include "split_engine.hpp"

#ifndef MDL_POOL_U
#define MDL_POOL_U 8   // token rows in flight per thread in the forward's accumulation loop
#endif
#ifndef MDL_POOL_PRE
#define MDL_POOL_PRE 1   // pool_partial_kernel: request the first rows of a chunk before its softmax statistics
#endif

namespace mdl {

// Element types of E: float, bf16_t, or img_t = a split-fp16 image row (csrc/split_engine.hpp) addressed in channel units (4 bytes per
// channel: 64 B of hi plane | 64 B of lo plane per 32 channels).  In the split GEMM mode the last pre_attn LayerNorm kernel writes E as
// an image only; the pooling kernels rebuild the fp32 values ((hi + lo) / scale: exact sum, power-of-two scale) -- E is never stored
// twice.
struct img_t {
    uint32_t u;
};
template <class TE>
struct PoolLd {
    // ld == decode(ld_raw): the two halves of a load, so that a kernel can request rows long before it consumes them
    typedef f32x4 raw_t;
    static __device__ __forceinline__ f32x4 ld(const TE* __restrict__ rowp, int col, float) { return ld4_nt(rowp + col); }
    static __device__ __forceinline__ raw_t ld_raw(const TE* __restrict__ rowp, int col) { return ld4_nt(rowp + col); }
    static __device__ __forceinline__ f32x4 decode(const raw_t& r, int, float) { return r; }
};
template <>
struct PoolLd<img_t> {
    // col = 4 x (lane index of a full wave): lanes 2k, 2k+1 own channels [8k, 8k+4), [8k+4, 8k+8) of one 32-channel block.  ONE 16-B load
    // per lane -- the even lane fetches the hi plane of the 8 channels, the odd lane their lo plane -- and the halves each lane is
    // missing come from its neighbour by DPP (two 8-B loads per lane cost the pooling kernels 12 %).  Every lane of the wave must call.
    typedef u32x4 raw_t;
    static __device__ __forceinline__ raw_t ld_raw(const img_t* __restrict__ rowp, int col) {   // col % 4 == 0
        const bool odd = (col >> 2) & 1;
        const char* p = reinterpret_cast<const char*>(rowp) + (col >> 5) * 128 + ((col & 31) & ~7) * 2 + (odd ? 64 : 0);
        return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    }
    static __device__ __forceinline__ f32x4 ld(const img_t* __restrict__ rowp, int col, float inv) { return decode(ld_raw(rowp, col), col, inv); }
    static __device__ __forceinline__ f32x4 decode(const raw_t& w, int col, float inv) {
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        const bool odd = (col >> 2) & 1;
        // even: w = hi[0..7]: keeps hi[0..3] = w.xy, sends hi[4..7] = w.zw;  odd: w = lo[0..7]: keeps lo[4..7] = w.zw, sends lo[0..3] = w.xy
        const uint32_t s0 = odd ? w.x : w.z, s1 = odd ? w.y : w.w;
        const uint32_t r0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s0, 0xB1, 0xF, 0xF, false);   // quad_perm [1,0,3,2]: lane ^ 1
        const uint32_t r1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s1, 0xB1, 0xF, 0xF, false);
        const h4 h = __builtin_bit_cast(h4, odd ? u32x2{r0, r1} : u32x2{w.x, w.y});
        const h4 l = __builtin_bit_cast(h4, odd ? u32x2{w.z, w.w} : u32x2{r0, r1});
        f32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = ((float)h[i] + (float)l[i]) * inv;
        return v;
    }
};

constexpr int POOL_CHUNK = 128;  // tokens per forward workgroup
constexpr int POOL_BWD_TOKENS = 128;  // tokens per backward workgroup (4 waves)

struct BagSpan {
    int64_t start, len;
};
// A "view" (intra-modality half-bag views, reference Model.py:419-440) is a dense bag restricted to the token index list
// idx[0..n_idx): logical token i of bag b is the physical row b*N + idx[i]; the same list for every bag.
__device__ __forceinline__ BagSpan bag_span(int b, int64_t N, const int64_t* cu, int64_t n_idx = -1) {
    BagSpan s;
    if (n_idx >= 0) {
        s.start = (int64_t)b * N;
        s.len = n_idx;
    } else if (cu) {
        s.start = cu[b];
        s.len = cu[b + 1] - s.start;
    } else {
        s.start = (int64_t)b * N;
        s.len = N;
    }
    return s;
}

// blockDim.x == H*128: thread f owns channels [4f, 4f+4) (head f/128).
// LIN: `scores` already ARE the (un-normalised) weights -- the relu / leaky_relu / sigmoid attention activations of
// abmil.py:56-61, which the reference pools without a softmax: p = s, chunk statistics (m, l) = (0, [chunk == 0]) so that
// pool_combine's merge is the plain sum (M = 0, L = 1).
#ifdef MDL_POOL_WPE   // A/B: occupancy target of the forward kernel (waves per SIMD)
#define POOL_FWD_ATTR __attribute__((amdgpu_waves_per_eu(MDL_POOL_WPE, MDL_POOL_WPE)))
#else
#define POOL_FWD_ATTR
#endif
template <int H, class TE, bool IDX = false, bool LIN = false>
__global__ __launch_bounds__(H * 128) POOL_FWD_ATTR void pool_partial_kernel(const TE* __restrict__ E, int64_t ldE,
                                                               const float* __restrict__ scores,
                                                               float* __restrict__ part_acc,
                                                               float* __restrict__ part_m,
                                                               float* __restrict__ part_l, int64_t N,
                                                               const int64_t* __restrict__ cu, int max_chunks,
                                                               const int32_t* __restrict__ idx = nullptr, int64_t n_idx = -1,
                                                               const float* __restrict__ e_scale = nullptr) {
    constexpr int NT = H * 128;
    constexpr int NW = NT / 64;
    __shared__ float p_s[POOL_CHUNK * H];  // exp(s - m_chunk), [t][c]
    __shared__ float red_s[NW * H];
    __shared__ float stat_s[2 * H];
    __shared__ int32_t tok_s[IDX ? POOL_CHUNK : 1];   // physical token of each logical token of the chunk (views)

    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const BagSpan sp = bag_span(b, N, cu, IDX ? n_idx : -1);
    const int64_t t0 = (int64_t)chunk * POOL_CHUNK;
    if (t0 >= sp.len) return;  // block-uniform
    const int nt = (int)((sp.len - t0 < POOL_CHUNK) ? (sp.len - t0) : POOL_CHUNK);

    // ---- chunk softmax statistics: thread tid holds score (t = tid / H, c = tid % H) ----------
    const int st = tid / H, sc = tid % H;
    const int64_t prow = (st < nt) ? (IDX ? (int64_t)idx[t0 + st] : t0 + st) : 0;
    if (IDX && sc == 0 && st < nt) tok_s[st] = (int32_t)prow;
    const float s = (st < nt) ? scores[(sp.start + prow) * H + sc] : (LIN ? 0.f : -INFINITY);

    // The first U token rows of the chunk are requested BEFORE the softmax statistics (round 5): that phase -- two block reductions,
    // three barriers -- is a latency bubble at the head of every workgroup, and the row stream does not depend on it.  The requests follow
    // the score load (loads return in order: the statistics then wait for vmcnt(U), not 0), they are unconditional (a branch around them
    // would make the join wait for vmcnt(0): a short tail chunk re-requests its last row for u >= nt, which meets p_s == 0 below), and
    // the phase's barriers (POOL_SYNC) wait for the LDS traffic only -- __syncthreads would drain the row loads in flight.
    constexpr int U = MDL_POOL_U;
    typedef PoolLd<TE> L;
    constexpr bool PRE = MDL_POOL_PRE && !IDX;
    const TE* __restrict__ Er = E + (sp.start + (IDX ? 0 : t0)) * ldE;
    typename L::raw_t r0[U];
    if constexpr (PRE) {
#pragma unroll
        for (int u = 0; u < U; ++u) r0[u] = L::ld_raw(Er + (int64_t)(u < nt ? u : nt - 1) * ldE, tid * 4);
    }
#if MDL_POOL_PRE
#define POOL_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#else
#define POOL_SYNC() __syncthreads()
#endif
    if (LIN) {
        p_s[st * H + sc] = s;
        if (tid < H) {
            const int64_t o = ((int64_t)b * max_chunks + chunk) * H + tid;
            part_m[o] = 0.f;
            part_l[o] = chunk == 0 ? 1.f : 0.f;
        }
        POOL_SYNC();
    }
    float mx = s;
    if (!LIN) {
#pragma unroll
    for (int o = 32; o >= H; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    const int lane = tid & 63, wave = tid >> 6;
    if (lane < H) red_s[wave * H + lane] = mx;
    POOL_SYNC();
    float m = red_s[sc];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = fmaxf(m, red_s[w * H + sc]);
    const float p = (st < nt) ? expf(s - m) : 0.f;
    p_s[st * H + sc] = p;
    float sm = p;
#pragma unroll
    for (int o = 32; o >= H; o >>= 1) sm += __shfl_xor(sm, o, 64);
    POOL_SYNC();  // red_s reads done; p_s written
    if (lane < H) red_s[wave * H + lane] = sm;
    POOL_SYNC();
    if (tid < H) {
        float l = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) l += red_s[w * H + tid];
        const int64_t o = ((int64_t)b * max_chunks + chunk) * H + tid;
        part_m[o] = m;  // tid < H => sc == tid, st == 0
        part_l[o] = l;
    }
    }   // !LIN

    // ---- weighted accumulation: thread owns one float4 column, loops over the chunk's tokens -----
    const int ca = tid / 128;
    const float inv = e_scale ? 1.f / e_scale[0] : 1.f;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    int t = 0;
    if constexpr (PRE) {   // the rows requested at the top: the arithmetic and order of the loop below (p_s is 0 for tokens >= nt)
#pragma unroll
        for (int u = 0; u < U; ++u) acc += p_s[u * H + ca] * L::decode(r0[u], tid * 4, inv);
        t = U;
    }
    for (; t + U <= nt; t += U) {
        f32x4 x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = PoolLd<TE>::ld(Er + (int64_t)(IDX ? tok_s[t + u] : t + u) * ldE, tid * 4, inv);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float w = p_s[(t + u) * H + ca];
            acc += w * x[u];
        }
    }
    for (; t < nt; ++t) {
        const f32x4 x = PoolLd<TE>::ld(Er + (int64_t)(IDX ? tok_s[t] : t) * ldE, tid * 4, inv);
        acc += p_s[t * H + ca] * x;
    }
    *reinterpret_cast<f32x4*>(part_acc + ((int64_t)b * max_chunks + chunk) * (H * HID) + (int64_t)tid * 4) = acc;
#undef POOL_SYNC
}

template <int H>
__global__ __launch_bounds__(H * 128) void pool_combine_kernel(const float* __restrict__ part_acc,
                                                               const float* __restrict__ part_m,
                                                               const float* __restrict__ part_l,
                                                               float* __restrict__ pooled, float* __restrict__ stat_m,
                                                               float* __restrict__ stat_l, int64_t N,
                                                               const int64_t* __restrict__ cu, int max_chunks,
                                                               int64_t n_idx = -1) {
    const int b = blockIdx.x, tid = threadIdx.x, ca = tid / 128;
    const BagSpan sp = bag_span(b, N, cu, n_idx);
    const int nchunks = (int)((sp.len + POOL_CHUNK - 1) / POOL_CHUNK);
    f32x4 out = {0.f, 0.f, 0.f, 0.f};
    float M = 0.f, L = 1.f;
    if (nchunks > 0) {
        const float* pm = part_m + (int64_t)b * max_chunks * H + ca;
        const float* pl = part_l + (int64_t)b * max_chunks * H + ca;
        M = pm[0];
        for (int k = 1; k < nchunks; ++k) M = fmaxf(M, pm[(int64_t)k * H]);
        L = 0.f;
        const float* pa = part_acc + (int64_t)b * max_chunks * (H * HID) + (int64_t)tid * 4;
        int k = 0;
        constexpr int U = 8;   // 8 partial rows in flight per thread (the merge was a chain of dependent 16-B loads: 17 us per launch)
        for (; k + U <= nchunks; k += U) {
            f32x4 x[U];
            float mk[U], lk[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                x[u] = *reinterpret_cast<const f32x4*>(pa + (int64_t)(k + u) * (H * HID));
                mk[u] = pm[(int64_t)(k + u) * H];
                lk[u] = pl[(int64_t)(k + u) * H];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {   // same order of the sums as the one-by-one loop below
                const float f = expf(mk[u] - M);
                L += f * lk[u];
                out += f * x[u];
            }
        }
        for (; k < nchunks; ++k) {
            const float f = expf(pm[(int64_t)k * H] - M);
            L += f * pl[(int64_t)k * H];
            out += f * *reinterpret_cast<const f32x4*>(pa + (int64_t)k * (H * HID));
        }
        const float rl = 1.f / L;
        out *= rl;
    }
    *reinterpret_cast<f32x4*>(pooled + (int64_t)b * (H * HID) + (int64_t)tid * 4) = out;
    if ((tid & 127) == 0) {
        stat_m[(int64_t)b * H + ca] = M;
        stat_l[(int64_t)b * H + ca] = L;
    }
}

// One wave per token row.  Lane L, slot i in [0,2H): channels [i*256 + 4L, +4), head i/2.
// IDX (views): d_scores == nullptr -> dE-only pass (no read of E: dE[t] += w[t,c] d_pooled[b,c,:]); both outputs accumulate.
// LIN (see pool_partial_kernel): w = the given weight, d_weight = <E[t,c,:], d_pooled[b,c,:]> (no softmax Jacobian).
#ifdef MDL_POOL_BWD_WPE   // A/B: occupancy target of the backward kernel (waves per SIMD)
#define POOL_BWD_ATTR __attribute__((amdgpu_waves_per_eu(MDL_POOL_BWD_WPE, MDL_POOL_BWD_WPE)))
#else
#define POOL_BWD_ATTR
#endif
template <int H, class TE, bool IDX = false, bool LIN = false>
__global__ __launch_bounds__(256) POOL_BWD_ATTR void pool_bwd_kernel(const TE* __restrict__ E, int64_t ldE,
                                                       const float* __restrict__ scores,
                                                       const float* __restrict__ pooled,
                                                       const float* __restrict__ stat_m,
                                                       const float* __restrict__ stat_l,
                                                       const float* __restrict__ d_pooled, TE* __restrict__ dE,
                                                       int accumulate, float* __restrict__ d_scores,
                                                       int accumulate_scores, int64_t N,
                                                       const int64_t* __restrict__ cu,
                                                       const int32_t* __restrict__ idx = nullptr, int64_t n_idx = -1,
                                                       const float* __restrict__ e_scale = nullptr) {
    const int b = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float inv = e_scale ? 1.f / e_scale[0] : 1.f;
    const BagSpan sp = bag_span(b, N, cu, IDX ? n_idx : -1);
    const int64_t t0 = (int64_t)chunk * POOL_BWD_TOKENS;
    if (t0 >= sp.len) return;
    const int nt = (int)((sp.len - t0 < POOL_BWD_TOKENS) ? (sp.len - t0) : POOL_BWD_TOKENS);

    f32x4 dp[2 * H];
    float D[H], m[H], rl[H];
#pragma unroll
    for (int c = 0; c < H; ++c) D[c] = 0.f;
    const int64_t boff = (int64_t)b * (H * HID) + lane * 4;
#pragma unroll
    for (int i = 0; i < 2 * H; ++i) {
        dp[i] = *reinterpret_cast<const f32x4*>(d_pooled + boff + i * 256);
        if (!LIN) {
            const f32x4 pl = *reinterpret_cast<const f32x4*>(pooled + boff + i * 256);
            D[i / 2] += dp[i].x * pl.x + dp[i].y * pl.y + dp[i].z * pl.z + dp[i].w * pl.w;
        }
    }
#pragma unroll
    for (int c = 0; c < H; ++c) {
        if (LIN) {
            m[c] = 0.f;
            rl[c] = 1.f;
        } else {
            D[c] = wave_sum(D[c]);  // <pooled[b,c,:], d_pooled[b,c,:]> = sum_t w_t dw_t
            m[c] = stat_m[(int64_t)b * H + c];
            rl[c] = 1.f / stat_l[(int64_t)b * H + c];
        }
    }

    for (int t = wave; t < nt; t += 4) {
        const int64_t row = sp.start + (IDX ? (int64_t)idx[t0 + t] : t0 + t);
        float w[H], dw[H];
#pragma unroll
        for (int c = 0; c < H; ++c) w[c] = LIN ? scores[row * H + c] : expf(scores[row * H + c] - m[c]) * rl[c];
        if (!IDX || d_scores) {
            const TE* __restrict__ er = E + row * ldE;
            f32x4 x[2 * H];
#pragma unroll
            for (int i = 0; i < 2 * H; ++i) x[i] = PoolLd<TE>::ld(er, lane * 4 + i * 256, inv);
#pragma unroll
            for (int c = 0; c < H; ++c) {
                const f32x4 a = x[2 * c] * dp[2 * c] + x[2 * c + 1] * dp[2 * c + 1];
                dw[c] = wave_sum(a.x + a.y + a.z + a.w);
            }
        }
        if constexpr (!std::is_same<TE, img_t>::value) {   // (an image E: scores-only pass, dE is never written here)
        if (dE) {  // dE == nullptr: scores-only pass (the dE term is folded into the gate's dX epilogue, mdl_abmil_attnpool_bwd)
            TE* __restrict__ gr = dE + row * ldE + lane * 4;
#pragma unroll
            for (int i = 0; i < 2 * H; ++i) {
                f32x4 g = w[i / 2] * dp[i];
                if (accumulate) g += ld4(gr + i * 256);
                st4(gr + i * 256, g);
            }
        }
        }
        if (IDX && !d_scores) continue;
        float ds = 0.f;
#pragma unroll
        for (int c = 0; c < H; ++c)
            if (lane == c) ds = LIN ? dw[c] : w[c] * (dw[c] - D[c]);
        if (lane < H) {
            if (accumulate_scores) ds += d_scores[row * H + lane];
            d_scores[row * H + lane] = ds;
        }
    }
}

static inline int64_t pool_max_chunks(int64_t max_len) { return (max_len + POOL_CHUNK - 1) / POOL_CHUNK; }

}  // namespace mdl

using namespace mdl;

extern "C" int64_t mdl_abmil_pool_ws_bytes(int64_t n_bags, int64_t max_len, int H) {
    if (n_bags < 0 || max_len < 0 || H < 1 || H > MDL_MAX_HEADS) return MDL_E_ARG;
    const int64_t mc = pool_max_chunks(max_len);
    // part_acc [n_bags][mc][H*512] + part_m, part_l [n_bags][mc][H]; each region 16-byte aligned
    const int64_t acc = n_bags * mc * H * HID * 4;
    const int64_t st = ((n_bags * mc * H * 4 + 15) / 16) * 16;
    return acc + 2 * st + 64;
}

#define MDL_DISPATCH_H(H, ...)                       \
    switch (H) {                                     \
        case 1: { constexpr int HH = 1; __VA_ARGS__; } break; \
        case 2: { constexpr int HH = 2; __VA_ARGS__; } break; \
        case 4: { constexpr int HH = 4; __VA_ARGS__; } break; \
        case 8: { constexpr int HH = 8; __VA_ARGS__; } break; \
        default: return MDL_E_UNSUPPORTED;           \
    }

// Dispatch timer of the A3 forward (the north_star kernel, bench.py's roofline): when a slot is armed, the NEXT pooling forward launches
// its two kernels through hipExtLaunchKernel with start / stop events -- the begin / end timestamps of the dispatches themselves (what
// rocprofv3 --kernel-trace reports), not the distance between two markers in a busy stream.  Nothing waits: the slot is read after the
// timed region.  Process-wide, not thread-safe (one training thread per process).
constexpr int POOL_TIMER_SLOTS = 64;
static hipEvent_t g_pool_ev[POOL_TIMER_SLOTS][4];
static bool g_pool_ev_made[POOL_TIMER_SLOTS];
static int g_pool_armed = -1;
static bool g_pool_used[POOL_TIMER_SLOTS];

extern "C" int mdl_pool_timer_arm(int slot) {
    if (slot >= POOL_TIMER_SLOTS) return MDL_E_ARG;
    if (slot >= 0 && !g_pool_ev_made[slot]) {
        for (int i = 0; i < 4; ++i) {
            const hipError_t e = hipEventCreate(&g_pool_ev[slot][i]);
            if (e != hipSuccess) return (int)e;
        }
        g_pool_ev_made[slot] = true;
    }
    if (slot >= 0) g_pool_used[slot] = false;
    g_pool_armed = slot;
    return MDL_OK;
}
// ms[0] = pool_partial, ms[1] = pool_combine, ms[2] = start of the first to end of the second.  Blocks until the slot's launches finished.
extern "C" int mdl_pool_timer_read(int slot, float* ms) {
    if (slot < 0 || slot >= POOL_TIMER_SLOTS || !ms || !g_pool_ev_made[slot] || !g_pool_used[slot]) return MDL_E_ARG;
    hipError_t e = hipEventSynchronize(g_pool_ev[slot][3]);
    if (e != hipSuccess) return (int)e;
    e = hipEventElapsedTime(&ms[0], g_pool_ev[slot][0], g_pool_ev[slot][1]);
    if (e != hipSuccess) return (int)e;
    e = hipEventElapsedTime(&ms[1], g_pool_ev[slot][2], g_pool_ev[slot][3]);
    if (e != hipSuccess) return (int)e;
    e = hipEventElapsedTime(&ms[2], g_pool_ev[slot][0], g_pool_ev[slot][3]);
    return e == hipSuccess ? MDL_OK : (int)e;
}

template <class TE, bool LIN = false>
static int pool_fwd_launch(const TE* E, int64_t ldE, const float* scores, float* pooled, float* stat_m, float* stat_l,
                           int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H, void* ws, void* stream,
                           const float* e_scale = nullptr) {
    if (!E || !scores || !pooled || !stat_m || !stat_l || !ws) return MDL_E_ARG;
    if (n_bags < 0 || max_len < 0 || ldE < (int64_t)H * HID || (ldE & 3)) return MDL_E_ARG;
    if (!cu_seqlens && N != max_len) return MDL_E_ARG;
    if (!host_aligned16(E) || !host_aligned16(pooled) || !host_aligned16(ws)) return MDL_E_ALIGN;
    if (n_bags == 0) return MDL_OK;
    if (n_bags > 65535) return MDL_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int mc = (int)pool_max_chunks(max_len);
    float* part_acc = (float*)ws;
    const int64_t st = ((n_bags * (int64_t)mc * H * 4 + 15) / 16) * 16;
    float* part_m = (float*)((char*)ws + n_bags * (int64_t)mc * H * HID * 4);
    float* part_l = (float*)((char*)part_m + st);
    const int slot = mc > 0 ? g_pool_armed : -1;
    g_pool_armed = -1;
    MDL_DISPATCH_H(H, {
        if (slot >= 0) {   // the same two launches with the dispatches' own begin / end events
            hipExtLaunchKernelGGL((pool_partial_kernel<HH, TE, false, LIN>), dim3(mc, (unsigned)n_bags), dim3(HH * 128), 0, s,
                                  g_pool_ev[slot][0], g_pool_ev[slot][1], 0, E, ldE, scores, part_acc, part_m, part_l, N, cu_seqlens, mc,
                                  (const int32_t*)nullptr, (int64_t)-1, e_scale);
            MDL_LAUNCH_CHECK();
            hipExtLaunchKernelGGL((pool_combine_kernel<HH>), dim3((unsigned)n_bags), dim3(HH * 128), 0, s, g_pool_ev[slot][2],
                                  g_pool_ev[slot][3], 0, (const float*)part_acc, (const float*)part_m, (const float*)part_l, pooled, stat_m,
                                  stat_l, N, cu_seqlens, mc, (int64_t)-1);
            MDL_LAUNCH_CHECK();
            g_pool_used[slot] = true;
        } else {
            if (mc > 0) {
                hipLaunchKernelGGL((pool_partial_kernel<HH, TE, false, LIN>), dim3(mc, (unsigned)n_bags), dim3(HH * 128), 0, s, E, ldE, scores,
                                   part_acc, part_m, part_l, N, cu_seqlens, mc, (const int32_t*)nullptr, (int64_t)-1, e_scale);
                MDL_LAUNCH_CHECK();
            }
            hipLaunchKernelGGL((pool_combine_kernel<HH>), dim3((unsigned)n_bags), dim3(HH * 128), 0, s, part_acc, part_m, part_l,
                               pooled, stat_m, stat_l, N, cu_seqlens, mc);
            MDL_LAUNCH_CHECK();
        }
    });
    return MDL_OK;
}

template <class TE, bool LIN = false>
static int pool_bwd_launch(const TE* E, int64_t ldE, const float* scores, const float* pooled, const float* stat_m,
                           const float* stat_l, const float* d_pooled, TE* dE, int accumulate, float* d_scores,
                           int accumulate_scores, int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H,
                           void* stream, const float* e_scale = nullptr) {
    if (!E || !scores || !d_pooled || !d_scores) return MDL_E_ARG;   // dE may be NULL
    if (!LIN && (!pooled || !stat_m || !stat_l)) return MDL_E_ARG;
    if (n_bags < 0 || max_len < 0 || ldE < (int64_t)H * HID || (ldE & 3)) return MDL_E_ARG;
    if (!cu_seqlens && N != max_len) return MDL_E_ARG;
    if (!host_aligned16(E) || !host_aligned16(dE) || !host_aligned16(pooled) || !host_aligned16(d_pooled)) return MDL_E_ALIGN;
    if (n_bags == 0 || max_len == 0) return MDL_OK;
    if (n_bags > 65535) return MDL_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int nc = (int)((max_len + POOL_BWD_TOKENS - 1) / POOL_BWD_TOKENS);
    MDL_DISPATCH_H(H, {
        hipLaunchKernelGGL((pool_bwd_kernel<HH, TE, false, LIN>), dim3(nc, (unsigned)n_bags), dim3(256), 0, s, E, ldE, scores, pooled, stat_m,
                           stat_l, d_pooled, dE, accumulate, d_scores, accumulate_scores, N, cu_seqlens, (const int32_t*)nullptr, (int64_t)-1,
                           e_scale);
        MDL_LAUNCH_CHECK();
    });
    return MDL_OK;
}

template <class TE>
static int pool_view_fwd_launch(const TE* E, int64_t ldE, const float* scores, float* pooled, float* stat_m, float* stat_l,
                                int64_t n_bags, int64_t N, const int32_t* token_idx, int64_t n_idx, int H, void* ws, void* stream) {
    if (!E || !scores || !pooled || !stat_m || !stat_l || !ws || !token_idx) return MDL_E_ARG;
    if (n_bags < 0 || N < 0 || n_idx < 0 || n_idx > N || ldE < (int64_t)H * HID || (ldE & 3)) return MDL_E_ARG;
    if (!host_aligned16(E) || !host_aligned16(pooled) || !host_aligned16(ws)) return MDL_E_ALIGN;
    if (n_bags == 0) return MDL_OK;
    if (n_bags > 65535) return MDL_E_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int mc = (int)pool_max_chunks(n_idx);
    float* part_acc = (float*)ws;
    const int64_t st = ((n_bags * (int64_t)mc * H * 4 + 15) / 16) * 16;
    float* part_m = (float*)((char*)ws + n_bags * (int64_t)mc * H * HID * 4);
    float* part_l = (float*)((char*)part_m + st);
    MDL_DISPATCH_H(H, {
        if (mc > 0) {
            hipLaunchKernelGGL((pool_partial_kernel<HH, TE, true>), dim3(mc, (unsigned)n_bags), dim3(HH * 128), 0, s, E, ldE, scores,
                               part_acc, part_m, part_l, N, (const int64_t*)nullptr, mc, token_idx, n_idx);
            MDL_LAUNCH_CHECK();
        }
        hipLaunchKernelGGL((pool_combine_kernel<HH>), dim3((unsigned)n_bags), dim3(HH * 128), 0, s, part_acc, part_m, part_l,
                           pooled, stat_m, stat_l, N, (const int64_t*)nullptr, mc, n_idx);
        MDL_LAUNCH_CHECK();
    });
    return MDL_OK;
}

template <class TE>
static int pool_view_bwd_launch(const TE* E, int64_t ldE, const float* scores, const float* pooled, const float* stat_m,
                                const float* stat_l, const float* d_pooled, TE* dE, float* d_scores, int64_t n_bags, int64_t N,
                                const int32_t* token_idx, int64_t n_idx, int H, void* stream) {
    if (!E || !scores || !pooled || !stat_m || !stat_l || !d_pooled || !token_idx || (!dE && !d_scores)) return MDL_E_ARG;
    if (n_bags < 0 || N < 0 || n_idx < 0 || n_idx > N || ldE < (int64_t)H * HID || (ldE & 3)) return MDL_E_ARG;
    if (!host_aligned16(E) || !host_aligned16(dE) || !host_aligned16(pooled) || !host_aligned16(d_pooled)) return MDL_E_ALIGN;
    if (n_bags == 0 || n_idx == 0) return MDL_OK;
    if (n_bags > 65535) return MDL_E_UNSUPPORTED;
    const int nc = (int)((n_idx + POOL_BWD_TOKENS - 1) / POOL_BWD_TOKENS);
    MDL_DISPATCH_H(H, {
        hipLaunchKernelGGL((pool_bwd_kernel<HH, TE, true>), dim3(nc, (unsigned)n_bags), dim3(256), 0, (hipStream_t)stream, E, ldE,
                           scores, pooled, stat_m, stat_l, d_pooled, dE, 1, d_scores, 1, N, (const int64_t*)nullptr, token_idx, n_idx);
        MDL_LAUNCH_CHECK();
    });
    return MDL_OK;
}

extern "C" int mdl_abmil_pool_view_fwd(const float* E, int64_t ldE, const float* scores, float* pooled, float* stat_m, float* stat_l,
                                       int64_t n_bags, int64_t N, const int32_t* token_idx, int64_t n_idx, int H, void* ws,
                                       void* stream) {
    return pool_view_fwd_launch<float>(E, ldE, scores, pooled, stat_m, stat_l, n_bags, N, token_idx, n_idx, H, ws, stream);
}
extern "C" int mdl_abmil_pool_view_bwd(const float* E, int64_t ldE, const float* scores, const float* pooled, const float* stat_m,
                                       const float* stat_l, const float* d_pooled, float* dE, float* d_scores, int64_t n_bags,
                                       int64_t N, const int32_t* token_idx, int64_t n_idx, int H, void* stream) {
    return pool_view_bwd_launch<float>(E, ldE, scores, pooled, stat_m, stat_l, d_pooled, dE, d_scores, n_bags, N, token_idx, n_idx, H,
                                       stream);
}
extern "C" int mdl_abmil_pool_view_fwd_bf16(const uint16_t* E, int64_t ldE, const float* scores, float* pooled, float* stat_m,
                                            float* stat_l, int64_t n_bags, int64_t N, const int32_t* token_idx, int64_t n_idx, int H,
                                            void* ws, void* stream) {
    return pool_view_fwd_launch<bf16_t>((const bf16_t*)E, ldE, scores, pooled, stat_m, stat_l, n_bags, N, token_idx, n_idx, H, ws,
                                        stream);
}
extern "C" int mdl_abmil_pool_view_bwd_bf16(const uint16_t* E, int64_t ldE, const float* scores, const float* pooled,
                                            const float* stat_m, const float* stat_l, const float* d_pooled, uint16_t* dE,
                                            float* d_scores, int64_t n_bags, int64_t N, const int32_t* token_idx, int64_t n_idx, int H,
                                            void* stream) {
    return pool_view_bwd_launch<bf16_t>((const bf16_t*)E, ldE, scores, pooled, stat_m, stat_l, d_pooled, (bf16_t*)dE, d_scores, n_bags,
                                        N, token_idx, n_idx, H, stream);
}

extern "C" int mdl_abmil_pool_fwd(const float* E, int64_t ldE, const float* scores, float* pooled, float* stat_m,
                                  float* stat_l, int64_t n_bags, int64_t N, const int64_t* cu_seqlens,
                                  int64_t max_len, int H, void* ws, void* stream) {
    return pool_fwd_launch<float>(E, ldE, scores, pooled, stat_m, stat_l, n_bags, N, cu_seqlens, max_len, H, ws, stream);
}

extern "C" int mdl_abmil_pool_bwd(const float* E, int64_t ldE, const float* scores, const float* pooled,
                                  const float* stat_m, const float* stat_l, const float* d_pooled, float* dE,
                                  int accumulate, float* d_scores, int accumulate_scores, int64_t n_bags, int64_t N,
                                  const int64_t* cu_seqlens, int64_t max_len, int H, void* stream) {
    return pool_bwd_launch<float>(E, ldE, scores, pooled, stat_m, stat_l, d_pooled, dE, accumulate, d_scores, accumulate_scores,
                                  n_bags, N, cu_seqlens, max_len, H, stream);
}

extern "C" int mdl_abmil_pool_fwd_bf16(const uint16_t* E, int64_t ldE, const float* scores, float* pooled, float* stat_m,
                                       float* stat_l, int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len,
                                       int H, void* ws, void* stream) {
    return pool_fwd_launch<bf16_t>((const bf16_t*)E, ldE, scores, pooled, stat_m, stat_l, n_bags, N, cu_seqlens, max_len, H, ws,
                                   stream);
}

extern "C" int mdl_abmil_pool_bwd_bf16(const uint16_t* E, int64_t ldE, const float* scores, const float* pooled,
                                       const float* stat_m, const float* stat_l, const float* d_pooled, uint16_t* dE,
                                       int accumulate, float* d_scores, int accumulate_scores, int64_t n_bags, int64_t N,
                                       const int64_t* cu_seqlens, int64_t max_len, int H, void* stream) {
    return pool_bwd_launch<bf16_t>((const bf16_t*)E, ldE, scores, pooled, stat_m, stat_l, d_pooled, (bf16_t*)dE, accumulate,
                                   d_scores, accumulate_scores, n_bags, N, cu_seqlens, max_len, H, stream);
}

// ---- E as a split image (the split GEMM mode: E exists as the image its LayerNorm kernel wrote, nothing else) ----
extern "C" int mdl_abmil_pool_fwd_img(const void* E_img, int64_t e_rsb, const float* e_scale, const float* scores, float* pooled,
                                      float* stat_m, float* stat_l, int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len,
                                      int H, void* ws, void* stream) {
    if (!e_scale || (e_rsb & 15)) return MDL_E_ARG;
    return pool_fwd_launch<img_t>((const img_t*)E_img, e_rsb / 4, scores, pooled, stat_m, stat_l, n_bags, N, cu_seqlens, max_len, H, ws,
                                  stream, e_scale);
}
// the score gradients of the pooling (d_scores (+)= ...); the dE term belongs to the gate dX epilogue (mdl_abmil_attnpool_bwd_split)
extern "C" int mdl_abmil_pool_dscores_img(const void* E_img, int64_t e_rsb, const float* e_scale, const float* scores, const float* pooled,
                                          const float* stat_m, const float* stat_l, const float* d_pooled, float* d_scores,
                                          int accumulate_scores, int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len,
                                          int H, void* stream) {
    if (!e_scale || (e_rsb & 15)) return MDL_E_ARG;
    return pool_bwd_launch<img_t>((const img_t*)E_img, e_rsb / 4, scores, pooled, stat_m, stat_l, d_pooled, (img_t*)nullptr, 0, d_scores,
                                  accumulate_scores, n_bags, N, cu_seqlens, max_len, H, stream, e_scale);
}

// ---- weighted (non-softmax) pooling: pooled[b,c,:] = sum_t weights[t,c] E[t,c,:] (abmil.py:56-61 activations + Model.py:416-417) ----
extern "C" int mdl_abmil_wpool_fwd(const float* E, int64_t ldE, const float* weights, float* pooled, float* scratch_m, float* scratch_l,
                                   int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H, void* ws, void* stream) {
    return pool_fwd_launch<float, true>(E, ldE, weights, pooled, scratch_m, scratch_l, n_bags, N, cu_seqlens, max_len, H, ws, stream);
}
extern "C" int mdl_abmil_wpool_bwd(const float* E, int64_t ldE, const float* weights, const float* d_pooled, float* dE, int accumulate,
                                   float* d_weights, int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H,
                                   void* stream) {
    return pool_bwd_launch<float, true>(E, ldE, weights, nullptr, nullptr, nullptr, d_pooled, dE, accumulate, d_weights, 0, n_bags, N,
                                        cu_seqlens, max_len, H, stream);
}
extern "C" int mdl_abmil_wpool_fwd_bf16(const uint16_t* E, int64_t ldE, const float* weights, float* pooled, float* scratch_m,
                                        float* scratch_l, int64_t n_bags, int64_t N, const int64_t* cu_seqlens, int64_t max_len, int H,
                                        void* ws, void* stream) {
    return pool_fwd_launch<bf16_t, true>((const bf16_t*)E, ldE, weights, pooled, scratch_m, scratch_l, n_bags, N, cu_seqlens, max_len, H,
                                         ws, stream);
}
extern "C" int mdl_abmil_wpool_bwd_bf16(const uint16_t* E, int64_t ldE, const float* weights, const float* d_pooled, uint16_t* dE,
                                        int accumulate, float* d_weights, int64_t n_bags, int64_t N, const int64_t* cu_seqlens,
                                        int64_t max_len, int H, void* stream) {
    return pool_bwd_launch<bf16_t, true>((const bf16_t*)E, ldE, weights, nullptr, nullptr, nullptr, d_pooled, (bf16_t*)dE, accumulate,
                                         d_weights, 0, n_bags, N, cu_seqlens, max_len, H, stream);
}
