// gate_common.hpp -- pieces shared by the fp32 (abmil_gate.hip) and bf16 (abmil_gate_bf16.hip) gate kernels
#pragma once
#include <stdlib.h>
#include "common.hpp"

namespace mdl {

constexpr int GATE_JT = HID / 128;  // 4 j-tiles of 128 gate columns per head

// Head <-> XCD affinity.  Workgroup b runs on XCD b % 8 (observed dispatch; used for speed only, never for
// correctness).  XCD x works on head x % H and on the (x / H)-th interleaved share of that head's token tiles, so
// the 2 MiB of a head's Wa|Wb stay resident in ONE XCD's 4 MiB L2 instead of all 8 MiB cycling through every L2
// (profiles/r01a: TCC hit rate 76 % with the token-major mapping).  `li` = this workgroup's index within its XCD.
struct XcdHead {
    int c, share, nshare, li;
};
__device__ __forceinline__ XcdHead xcd_head(int bid, int H) {
    XcdHead m;
    const int x = bid & 7;
    m.nshare = 8 / H;  // H in {1,2,4,8}
    m.c = x % H;
    m.share = x / H;
    m.li = bid >> 3;
    return m;
}
static inline int64_t xcd_head_grid(int64_t units, int per_unit, int H) {  // units = tiles of one head, split over 8/H XCDs
    const int nshare = 8 / H;
    return 8 * ((units + nshare - 1) / nshare) * per_unit;
}

// Persistent gate forward (one workgroup runs the GATE_JT column tiles of a token tile): a tile costs `gain` of a tile launched as its own
// workgroup (measured: 0.96 split, 0.93 bf16), but the last wave of workgroups is GATE_JT tiles long.  256 CUs, one workgroup each.
static inline bool gate_persist_pays(int64_t grid_tiles, double gain, int per = 4) {
    const int64_t rounds_tiles = (grid_tiles + 255) / 256, rounds_persist = (grid_tiles / per + 255) / 256;
    return (double)rounds_persist * per * gain < (double)rounds_tiles;
}

struct DropCfg {
    float p, inv;
    uint32_t thr;   // 16-bit threshold
    uint32_t key;   // rng_key(seed)
    const uint8_t* ka;
    const uint8_t* kb;
    int on;
    int bytes;      // byte-field mode of the counter hash (see gate_keep2_bytes): thr is a multiple of 256
};

// rng_u32(key, idx) = mix32(lo32(idx) ^ key ^ (hi32(idx) * golden)): the part that depends on the high word only, hoisted once per
// group of elements that share it (a 64-bit add, a shift and a quarter-rate v_mul_lo_u32 per element otherwise)
__device__ __forceinline__ uint32_t drop_row_key(const DropCfg& d, int64_t idx0) {
    return d.key ^ ((uint32_t)((uint64_t)idx0 >> 32) * 0x9E3779B9U);
}
// Counter-hash keep decisions of one gate element (tanh branch, sigmoid branch).  Two field widths:
//   16-bit fields (any p):  h = mix32(lo32(idx) ^ row_key);  ka = h[15:0] >= thr,  kb = h[31:16] >= thr
//   byte fields (round 5; DropCfg.bytes: thr % 256 == 0, e.g. the gate's p = 0.25 -> P(keep) exact): ONE hash serves the element PAIR
//   (2q, 2q + 1):  h = mix32((lo32(idx) >> 1) ^ row_key);  element 2q: ka = h[7:0] >= thr >> 8, kb = h[15:8] >= ...;  element 2q + 1:
//   bytes 2 and 3 -- half of the hash instructions of the VALU-bound forward epilogue.  Forward, dz pass and mask export all come here.
// gate_keep2_hash<BYTES>(d, idx_even, e, row_key, ...): element idx_even + e with idx_even even and e a compile-time offset, so that the
// two elements of a pair share the hash after common-subexpression elimination.
template <bool BYTES>
__device__ __forceinline__ void gate_keep2_hash(const DropCfg& d, int64_t idx_even, int e, uint32_t row_key, bool& ka, bool& kb) {
    if (BYTES) {
        const uint32_t h = mix32((((uint32_t)idx_even >> 1) + (uint32_t)(e >> 1)) ^ row_key);
        const uint32_t f = (e & 1) ? (h >> 16) : h, t8 = d.thr >> 8;
        ka = (f & 0xFFu) >= t8;
        kb = ((f >> 8) & 0xFFu) >= t8;
    } else {
        const uint32_t h = mix32(((uint32_t)idx_even + (uint32_t)e) ^ row_key);
        ka = (h & 0xFFFFu) >= d.thr;
        kb = (h >> 16) >= d.thr;
    }
}
// DM: dropout mode of a forward-kernel instantiation: 0 = off, 1 = counter hash with 16-bit fields, 2 = explicit uint8 masks, 3 = counter
// hash with byte fields
template <int DM>
__device__ __forceinline__ void gate_keep2_fwd(const DropCfg& d, int64_t idx_even, int e, uint32_t row_key, bool& ka, bool& kb) {
    if (DM == 0) {
        ka = kb = true;
    } else if (DM == 2) {
        ka = d.ka[idx_even + e] != 0;
        kb = d.kb[idx_even + e] != 0;
    } else {
        gate_keep2_hash<DM == 3>(d, idx_even, e, row_key, ka, kb);
    }
}
static inline int gate_drop_mode(const DropCfg& d) { return !d.on ? 0 : (d.ka ? 2 : (d.bytes ? 3 : 1)); }
// run-time form (dz passes, mask export): any idx; row_key = drop_row_key of an index with the same high word
__device__ __forceinline__ void drop_keep2(const DropCfg& d, int64_t idx, uint32_t row_key, bool& ka, bool& kb) {
    if (!d.on) {
        ka = kb = true;
    } else if (d.ka) {
        ka = d.ka[idx] != 0;
        kb = d.kb[idx] != 0;
    } else if (d.bytes) {
        const uint32_t h = mix32(((uint32_t)idx >> 1) ^ row_key);
        const uint32_t f = ((uint32_t)idx & 1u) ? (h >> 16) : h, t8 = d.thr >> 8;
        ka = (f & 0xFFu) >= t8;
        kb = ((f >> 8) & 0xFFu) >= t8;
    } else {
        const uint32_t h = mix32((uint32_t)idx ^ row_key);
        ka = (h & 0xFFFFu) >= d.thr;
        kb = (h >> 16) >= d.thr;
    }
}

// MFMA 32x32 C/D layout: register r of lane l holds row (r&3) + 8*(r>>2) + 4*(l>>5), column l&31.
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// LDS-DMA, saddr form.  M0 = LDS destination of the wave (wave-uniform); restored because hipcc owns M0.
__device__ __forceinline__ void glds16_s(uint32_t voff, const void* sbase, uint32_t lds_addr) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p;
}

__device__ __forceinline__ void gate_dz(const DropCfg& d, float ds, float wcv, float a, float b, int64_t idx, uint32_t row_key,
                                        float& dza, float& dzb, float& pab) {
    bool keep_a, keep_b;
    drop_keep2(d, idx, row_key, keep_a, keep_b);
    const float ka = keep_a ? d.inv : 0.f;
    const float kb = keep_b ? d.inv : 0.f;
    const float ad = a * ka, bd = b * kb;
    const float g = ds * wcv;
    dza = g * bd * ka * (1.f - a * a);
    dzb = g * ad * kb * (b * (1.f - b));
    pab = ds * ad * bd;
}

// the same with the dropout mode fixed at compile time (DM as gate_keep2_fwd): element idx_even + e, idx_even even, e a compile-time
// offset -- no mode branches per element, and in byte-field mode the two elements of a pair share one hash
template <int DM>
__device__ __forceinline__ void gate_dz_t(const DropCfg& d, float ds, float wcv, float a, float b, int64_t idx_even, int e, uint32_t row_key,
                                          float& dza, float& dzb, float& pab) {
    bool keep_a, keep_b;
    gate_keep2_fwd<DM>(d, idx_even, e, row_key, keep_a, keep_b);
    const float ka = keep_a ? d.inv : 0.f;
    const float kb = keep_b ? d.inv : 0.f;
    const float ad = a * ka, bd = b * kb;
    const float g = ds * wcv;
    dza = g * bd * ka * (1.f - a * a);
    dzb = g * ad * kb * (b * (1.f - b));
    pab = ds * ad * bd;
}

static inline DropCfg make_drop(float p, uint64_t seed, const uint8_t* ka, const uint8_t* kb) {
    DropCfg d;
    d.on = (p > 0.f) ? 1 : 0;
    d.p = p;
    d.inv = d.on ? 1.f / (1.f - p) : 1.f;
    d.thr = drop_threshold(p);
    d.bytes = (d.on && d.thr > 0 && (d.thr & 0xFFu) == 0 && !getenv("MADELEINE_DROP_16BIT")) ? 1 : 0;
    d.key = (uint32_t)(seed * 0x9E3779B97F4A7C15ULL >> 32) ^ (uint32_t)seed;
    d.ka = ka;
    d.kb = kb;
    return d;
}

// Token splits of the dW-type contractions.  Base: ~4k tokens per split.  When there is enough work the count is rounded up
// so that the total number of output tiles (tiles_per_split x splits) is a whole number of "rounds" of the 768 workgroup
// slots of the chip (256 CUs x 3 resident workgroups of this tile engine): 4096 equal tiles on 768 slots is 5.33 rounds,
// i.e. a last round that is 2/3 idle; 3072 or 768 tiles are exact.  Splits stay >= 1024 tokens and <= 192.
// `slots`: resident workgroups of the kernel on the chip: 768 for the 156-VGPR fp32 TN kernels, 512 for the bf16 TN kernels (200-216
// VGPRs: two per CU) -- with 768 assumed the bf16 Linear dW launched 768 tiles on 512 slots (1.5 rounds).
// Round 6: MADELEINE_SPLIT_TOKENS (default MDL_SPLIT_TOKENS) = tokens per split to aim for once the chip is filled: every split writes
// one fp32 slab of the whole gradient that a reduction kernel re-reads (config 2, gate dW: 64 splits = 537 MB written + read, 127 us of
// reduction alone), so beyond one full round of workgroups more splits only add traffic.  The count falls from the 4096-token base
// towards T / quantum but never below the `slots / tiles_per_split` that give every CU a workgroup.  Same-box A/B at config 2
// (profiles/r06e_split_tokens_ab.txt): 4096 -> 32768 tokens: gate dX + dW 5.91 -> 5.72 ms (fp32 values), 2.30 -> 2.08 ms (bf16); step
// 22.20 -> 21.93-22.05 ms, bf16 10.41-10.49 -> 10.19-10.23 ms.  MADELEINE_SPLIT_TOKENS=4096 restores the round-5 counts.
#ifndef MDL_SPLIT_TOKENS
#define MDL_SPLIT_TOKENS 32768
#endif
static inline int splits_for(int64_t T, int tiles_per_split, int64_t slots = 768) {
    static const int64_t quantum = getenv("MADELEINE_SPLIT_TOKENS") ? atoll(getenv("MADELEINE_SPLIT_TOKENS")) : MDL_SPLIT_TOKENS;
    int64_t s = (T + 4095) / 4096;
    if (s < 1) s = 1;
    if (s > 64) s = 64;
    if (quantum > 4096) {
        const int64_t fill = (slots + tiles_per_split - 1) / tiles_per_split, sq = (T + quantum - 1) / quantum;
        const int64_t lo = s < fill ? s : fill;
        s = sq > lo ? sq : lo;
        if (s > 64) s = 64;
    }
    if (s * tiles_per_split >= slots / 2) {
        const int64_t rounds = (s * tiles_per_split + slots - 1) / slots;
        const int64_t want = (rounds * slots + tiles_per_split - 1) / tiles_per_split;
        if (want <= 192 && T / want >= 1024) s = want;
    }
    return (int)s;
}
static inline int gate_splits(int64_t T, int H) { return splits_for(T, 16 * H); }  // 4 k-tiles x 4 column tiles per head

constexpr int DZ_ROWS = 256;  // token rows per workgroup
// launches KERNEL<..., DM> for the run-time dropout mode dm (gate_drop_mode)
#define MDL_DISPATCH_DM(dm, LAUNCH) \
    do {                            \
        if ((dm) == 0) LAUNCH(0);   \
        else if ((dm) == 1) LAUNCH(1); \
        else if ((dm) == 3) LAUNCH(3); \
        else LAUNCH(2);             \
    } while (0)

// dz pass of the fp32-MFMA / bf16 modes (gate_dz_kernel below).  slabV [nblk][H][4][512]: dba | dbb | dwc | (dbc at [0]).
template <class T>
__device__ __forceinline__ void ldv(const T* p, float (&v)[16 / sizeof(T)]);
template <>
__device__ __forceinline__ void ldv<float>(const float* p, float (&v)[4]) {
    const f32x4 x = ld4_nt(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = x[i];
}
template <>
__device__ __forceinline__ void ldv<bf16_t>(const bf16_t* p, float (&v)[8]) {
    const bf16x8 x = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p));
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)x[i];
}
__device__ __forceinline__ void stv(float* p, const float (&v)[4]) { *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]}; }
__device__ __forceinline__ void stv(bf16_t* p, const float (&v)[8]) {
    bf16x8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
    *reinterpret_cast<bf16x8*>(p) = o;
}
__device__ __forceinline__ void stv(bf16_t* p, const float (&v)[4]) { st4(p, f32x4{v[0], v[1], v[2], v[3]}); }
__device__ __forceinline__ void stv(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

template <class T> struct DzRaw;
template <> struct DzRaw<float> { typedef f32x4 type; };
template <> struct DzRaw<bf16_t> { typedef bf16x8 type; };
#ifndef MDL_DZ_UNROLL
#define MDL_DZ_UNROLL 2   // rows in flight per thread (measured: 2 -> 0.80 ms, 4 -> 0.81 ms for the bf16 pass of config 2)
#endif
// Round 5: workgroup = DZ_ROWS token rows x ALL heads (64 H threads: thread = (head c, lane q)), as sp_gate_dz_kernel: every row is one
// contiguous H x 512-element read of each activation and one contiguous H x 1024-element write of dz (a workgroup per (rows, head)
// touched 1 KiB of every 4 KiB of a bf16 row: 2.5 TB/s).  Lane q owns VEC = 16 B / sizeof(TI) columns at q VEC + h (64 VEC), h < NH:
// every memory instruction of the wave moves 1 KiB contiguous.  Rows in order, MDL_DZ_UNROLL rows of raw loads in flight.
// DM: dropout mode fixed at compile time (gate_keep2_fwd; the run-time form cost ~50 instructions per element).
template <class TI, class TO, int DM>
__global__ __launch_bounds__(64 * MDL_MAX_HEADS) void gate_dz_kernel(const float* __restrict__ wc, const TI* __restrict__ act_a,
                                                                    const TI* __restrict__ act_b, const float* __restrict__ d_scores,
                                                                    TO* __restrict__ dz, float* __restrict__ slabV, int64_t T, int H,
                                                                    DropCfg drop) {
    constexpr int VEC = 16 / sizeof(TI), NH = HID / (64 * VEC);
    typedef typename DzRaw<TI>::type raw_t;
    const int tid = threadIdx.x, q = tid & 63, c = tid >> 6;
    const int64_t bx = blockIdx.x;
    const int64_t r0 = bx * DZ_ROWS;
    int64_t r1 = r0 + DZ_ROWS;
    if (r1 > T) r1 = T;
    float vw[NH][VEC], sa[NH][VEC], sb[NH][VEC], sw[NH][VEC];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            vw[h][i] = wc[c * HID + h * 64 * VEC + q * VEC + i];
            sa[h][i] = sb[h][i] = sw[h][i] = 0.f;
        }
    float sds = 0.f;
    constexpr int UNR = MDL_DZ_UNROLL;   // UNR rows x NH x (a, b) x 16 B in flight per thread
    for (int64_t rb = r0; rb < r1; rb += UNR) {
        raw_t ra[UNR][NH], rv[UNR][NH];
        float ds[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t r = rb + u;
            const bool ok = r < r1;
            const int64_t o = ((ok ? r : rb) * H + c) * HID + q * VEC;
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                ra[u][h] = __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(act_a + o + h * 64 * VEC));
                rv[u][h] = __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(act_b + o + h * 64 * VEC));
            }
            ds[u] = ok ? d_scores[r * H + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int64_t r = rb + u;
            if (r < r1) {   // block-uniform
                const int64_t o = (r * H + c) * HID + q * VEC;
                const uint32_t rkey = drop_row_key(drop, o);   // the head's 512 elements share the high word
                TO* __restrict__ out = dz + (r * H + c) * 1024 + q * VEC;
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    float za[VEC], zb[VEC];
#pragma unroll
                    for (int i = 0; i < VEC; ++i) {
                        float w;
                        gate_dz_t<DM>(drop, ds[u], vw[h][i], (float)ra[u][h][i], (float)rv[u][h][i], o + h * 64 * VEC, i, rkey, za[i], zb[i], w);
                        sw[h][i] += w;
                        sa[h][i] += za[i];
                        sb[h][i] += zb[i];
                    }
                    stv(out + h * 64 * VEC, za);
                    stv(out + HID + h * 64 * VEC, zb);
                }
                sds += ds[u];
            }
        }
    }
    float* __restrict__ o = slabV + (bx * H + c) * 4 * HID + q * VEC;
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            o[h * 64 * VEC + i] = sa[h][i];
            o[HID + h * 64 * VEC + i] = sb[h][i];
            o[2 * HID + h * 64 * VEC + i] = sw[h][i];
        }
    if (q == 0) o[3 * HID] = sds;
}


// Optional pooling term of the fused A2+A3 backward: dE[t, c, :] = (dz . W)[t, c, :] + w[t, c] * d_pooled[bag(t), c, :]
// with w = softmax weight of token t in its bag = exp(score - m) / l.  scores == nullptr: no pooling term.
struct PoolTerm {
    const float* scores;    // [T, H] raw scores
    const float* stat_m;    // [n_bags, H]
    const float* stat_l;    // [n_bags, H]
    const float* d_pooled;  // [n_bags, H*512]
    const int* row_bag;     // [T] bag index of every token row, or nullptr for dense bags of N tokens
    int64_t N;
};
__device__ __forceinline__ float pool_term_weight(const PoolTerm& pt, int64_t t, int c, int H, int& bag) {
    bag = pt.row_bag ? pt.row_bag[t] : (int)(t / pt.N);
    return expf(pt.scores[t * H + c] - pt.stat_m[(int64_t)bag * H + c]) * (1.f / pt.stat_l[(int64_t)bag * H + c]);
}

// Epilogue through LDS for 2-byte outputs (bf16 mode): the 32x32 MFMA layout gives a lane one COLUMN of 16 rows, i.e. 2-byte
// stores 64 B apart; a wave-private [32][64] fp32 transpose tile turns the wave's 64 x 128 sub-tile into row-contiguous groups
// of 8 columns per lane: emit(row_u, rl, lane_col, lo, hi, cp) gets columns lane_col .. +7 (tile coordinates through colb) of
// tile row row_u + rl (row_u wave-uniform, rl = lane >> 3) as two float4 -> one 16-B store of 8 bf16; 8 lanes cover 128
// contiguous bytes of a row.  `tile` = 2048 floats private to the wave, in staging memory that is free (main loop done).
template <bool FULL, int NCT, class Emit>
__device__ __forceinline__ void epilogue_rows8(const f32x16 (&acc)[2][NCT], float* tile, int wm, const int (&colb)[NCT], int lane,
                                               int rows_valid, Emit&& emit) {
    const int l32 = lane & 31, rl = lane >> 3, g = lane & 7;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int cp = 0; cp < NCT / 2; ++cp) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) tile[acc_row(r, lane) * 64 + c2 * 32 + l32] = acc[rt][cp * 2 + c2][r];
            const int lane_col = ((g >> 2) ? colb[cp * 2 + 1] : colb[cp * 2]) + (g & 3) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(&tile[(i * 8 + rl) * 64 + g * 8]);
                const f32x4 hi = *reinterpret_cast<const f32x4*>(&tile[(i * 8 + rl) * 64 + g * 8 + 4]);
                const int row_u = wm * 64 + rt * 32 + i * 8;
                if (FULL || row_u + rl < rows_valid) emit(row_u, rl, lane_col, lo, hi, cp);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
}
__device__ __forceinline__ void st8_bf16(bf16_t* p, const f32x4& lo, const f32x4& hi) {   // p 16-B aligned
    typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
    bf16x8v o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        o[i] = (bf16_t)lo[i];
        o[4 + i] = (bf16_t)hi[i];
    }
    *reinterpret_cast<bf16x8v*>(p) = o;
}

// launchers of the reduction / finalize kernels defined in abmil_gate.hip (shared with the bf16 path)
int gate_launch_finalize(const float* part, const float* bc, float* scores, int64_t n, int H, hipStream_t s);
int gate_launch_reduce_w(const float* slabW, float* dWa, float* dWb, int H, int S, hipStream_t s);
int gate_launch_reduce_v(const float* slabV, float* dba, float* dbb, float* dwc, float* dbc, int H, int S, hipStream_t s);

}  // namespace mdl
