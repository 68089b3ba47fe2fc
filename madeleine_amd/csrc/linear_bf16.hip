// linear_bf16.hip -- N1 in the bf16 mode: the Linears of the pre-attention MLP / token projector (reference
// madeleine/models/Model.py:351, :355, :359, :140) on v_mfma_f32_32x32x16_bf16, replacing the library GEMMs the autocast run used
// in round 1 (hipBLASLt Cijk_*: ~0.6 ms per 262144 x 512 x 512 product).  Activations X, Y, dY, dX are bf16 (the bf16 mode keeps
// them in bf16 end to end, model.py); the parameters W, bias and their gradients are fp32 (autocast keeps master weights fp32).
//   forward : Y[t, n]  = sum_k X[t, k] W[n, k] (+ bias[n])       NT engine, B = bf16 copy of W
//   dX      : dX[t, k] = sum_n dY[t, n] W[n, k]                   NT engine, B = bf16 copy of W^T
//   dW      : dW[n, k] = sum_t dY[t, n] X[t, k]                   TN engine (ds_read_b64_tr_b16), fp32 slabs over token splits
//   dbias   : column sums of dY (fp32)
// Geometry: N % 128 == 0 (forward: 128 x 256 tile when N % 256 == 0, else 128 x 128), K % 32 == 0 (ragged last 256-column tile of dX /
// dW: clamped operand fetches, masked stores);
// leading dimensions multiples of 8 elements, 16-B aligned bases.  T is free (row tails are clamped / zero-filled).
#include <type_traits>

#include "tile_engine_bf16.hpp"

namespace mdl {

// ---- weight images -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linb_w_cast_kernel(const float* __restrict__ W, bf16_t* __restrict__ Wb, int64_t n) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i < n) st4(Wb + i, ld4(W + i));
}
// WT[k][n] = W[n][k]; N, K multiples of 32
__global__ __launch_bounds__(256) void linb_w_transpose_kernel(const float* __restrict__ W, bf16_t* __restrict__ WT, int N, int K) {
    __shared__ float tile[32][33];
    const int nb = blockIdx.y * 32, kb = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) tile[ty + i * 8][tx] = W[(int64_t)(nb + ty + i * 8) * K + kb + tx];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) WT[(int64_t)(kb + ty + i * 8) * N + nb + tx] = (bf16_t)tile[tx][ty + i * 8];
}

// ---- NT product: C[t, n0 + n] = sum_k A[t][k] B[n0 + n][k] (+ bias) -----------------------------------
// Logical tile id = row tile * n_ct + column tile, XCD-remapped so the column tiles of one row tile (which re-read the same A rows)
// share an L2.
template <int NCT>   // 32-column tiles per wave: 4 -> 128 x 256 tile, 2 -> 128 x 128 (the token projector's 128 outputs)
__global__ __launch_bounds__(256, 2) void linb_nt_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
                                                         const float* __restrict__ bias, bf16_t* __restrict__ C, int64_t ldc,
                                                         int64_t T, int Kc, int ncols, int n_ct, int n_tiles) {
    constexpr int NS = (NCT == 4) ? 2 : 3;   // measured (tools/exp_linear_bf16.py): 128 x 256: 2 stages at 3 workgroups per CU (168 VGPRs, 52 KiB) beat 3 stages at 2; 128 x 128 (K = 2048, HBM-side): the 3-stage ring wins
    __shared__ SmemNTR<NS> sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int ct = lid % n_ct;
    const int64_t t0 = (int64_t)(lid / n_ct) * BBM;
    constexpr int BN = 64 * NCT;
    const int n0 = ct * BN;

    // LDS-DMA in the saddr form: wave-uniform 64-bit chunk base + 32-bit lane offset
    const char* baseA = reinterpret_cast<const char*>(A + t0 * lda);
    const char* baseB = reinterpret_cast<const char*>(B + (int64_t)n0 * Kc);
    uint32_t voA[2], voB[NCT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        int row, kq;
        nt_slot(wave * 2 + q, lane, row, kq);
        int64_t r = row;
        if (t0 + r > T - 1) r = T - 1 - t0;   // row tail: re-read the last row, those outputs are not stored
        voA[q] = (uint32_t)(r * lda * 2 + kq * 16);
    }
#pragma unroll
    for (int q = 0; q < NCT; ++q) {
        int row, kq;
        nt_slot(wave * NCT + q, lane, row, kq);
        int nr = n0 + row;
        if (nr > ncols - 1) nr = ncols - 1;   // ragged last column tile: duplicate rows, their outputs are not stored
        voB[q] = (uint32_t)((int64_t)(nr - n0) * Kc * 2 + kq * 16);
    }
    const uint32_t ldsA = lds_addr_of(&sm.A[0][0]) + wave * 2 * 1024, ldsB = lds_addr_of(&sm.B[0][0]) + wave * NCT * 1024;
    auto issue = [&](int st, int64_t ch) {
        const char* a = baseA + ch * (BBK * 2);
        const char* b = baseB + ch * (BBK * 2);
#pragma unroll
        for (int q = 0; q < 2; ++q) glds16_s(voA[q], a, ldsA + st * (BBM * BBK * 2) + q * 1024);
#pragma unroll
        for (int q = 0; q < NCT; ++q) glds16_s(voB[q], b, ldsB + st * (BBN * BBK * 2) + q * 1024);
    };
    int colb[NCT];
#pragma unroll
    for (int i = 0; i < NCT; ++i) colb[i] = wn * (32 * NCT) + i * 32;
    int offA[2], offB[NCT];
    nt_offsets(wm, colb, lane, offA, offB);
    f32x16 acc[2][NCT];
    zero_acc8(acc);
    nt_mainloop_ring<NS, NCT>(sm, acc, Kc / BBK, issue, offA, offB);

    float* tile = reinterpret_cast<float*>(&sm) + wave * (32 * 64);
    bf16_t* ob = C + t0 * ldc + n0;
    auto emit = [&](int row_u, int rl, int lane_col, const f32x4& lo, const f32x4& hi, int) {
        if (n0 + lane_col >= ncols) return;   // ncols % 8 == 0: an 8-column group is entirely in or out
        bf16_t* o = ob + (int64_t)row_u * ldc + ((uint32_t)rl * (uint32_t)ldc + (uint32_t)lane_col);
        f32x4 a = lo, b = hi;
        if (bias) {
            a += *reinterpret_cast<const f32x4*>(bias + n0 + lane_col);
            b += *reinterpret_cast<const f32x4*>(bias + n0 + lane_col + 4);
        }
        st8_bf16(o, a, b);
    };
    if (t0 + BBM <= T) epilogue_rows8<true>(acc, tile, wm, colb, lane, BBM, emit);
    else epilogue_rows8<false>(acc, tile, wm, colb, lane, (int)(T - t0), emit);
}

// ---- NT product on the 256 x 256 x 64 tile (8 waves) ----------------------------------------------------------------------------
// Round-2 lab result (tools/micro/gemm_lab_bf16.hip): the only loop variant that moves the ~1 PF ceiling of the
// bf16 engine is the one with twice the MFMA work per staged byte AND whole 128-B cache lines per row and chunk (BK = 64): 1.2 PF
// for the main loop against 1.0.  Workgroup = 8 waves (2 x 4), each 128 x 64 = 4 x 2 MFMA tiles; LDS stage = 256 rows x 128 B per
// operand, 16-B chunk c of row r stored at chunk c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 on 128-B rows); two stages =
// 128 KiB, one workgroup per CU; the in-wave pipeline of the fp32 engine (fragments of k-step s+1 requested behind the first MFMA
// of step s, one barrier per chunk before its last step, the next-but-one chunk's 8 LDS-DMA pieces between that step's MFMAs).
// Used for T >= 4096 rows, output width % 256 == 0, contraction >= 1024 and % 64 == 0 (the dX of the 512 -> 2048 Linear); the 128 x 256 x 32 kernel covers the rest.
// Round 5: PER > 1 = persistent workgroup over PER consecutive column tiles of one row tile (n_ct % PER == 0): the next tile's first
// chunk (same A rows, next 256 weight rows) is requested before the epilogue of the current one, which then stages through stage 1
// only (see gate_fwd256_bf16_kernel).  With the prologue hidden the tile also serves contractions of 512 (linb_use_q).
template <int PER, int NA = 2>   // NA = 3 (PER = 1 only): nt256_mainloop3 on SmemQ3
__global__ __launch_bounds__(512) void linb_nt256_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ B,
                                                         const float* __restrict__ bias, bf16_t* __restrict__ C, int64_t ldc,
                                                         int64_t T, int Kc, int n_ct, int n_tiles) {
    static_assert(NA == 2 || PER == 1, "the three-stage ring serves one tile per workgroup");
    __shared__ typename std::conditional<NA == 3, SmemQ3, SmemQ>::type sm3;
    SmemQ& sm = reinterpret_cast<SmemQ&>(sm3);   // (epilogue staging / persistent prefetch: the two-stage view)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;   // rows wm*128 + rt*32 (rt < 4), columns wn*64 + ct*32 (ct < 2)
    const int lid = xcd_remap(blockIdx.x, n_tiles / PER) * PER;   // first tile of this workgroup
    const int64_t t0 = (int64_t)(lid / n_ct) * QM;

    // DMA: one instruction = 8 rows x 128 B; wave w issues row blocks 4w .. 4w+3 of each operand
    const char* baseA = reinterpret_cast<const char*>(A + t0 * lda);
    uint32_t voA[4], voB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int row, c;
        nt256_slot(wave, i, lane, row, c);
        int64_t ra = row;
        if (t0 + ra > T - 1) ra = T - 1 - t0;   // row tail: re-read the last row (not stored)
        voA[i] = (uint32_t)(ra * lda * 2 + c * 16);
        voB[i] = (uint32_t)((int64_t)row * Kc * 2 + c * 16);
    }
#pragma unroll 1
    for (int k = 0; k < PER; ++k) {
        const int n0 = (lid % n_ct + k) * QN;
        const char* baseB = reinterpret_cast<const char*>(B + (int64_t)n0 * Kc);
        auto dma = [&](int st, int64_t f, int piece) {   // piece 0..7: 0-3 = A row blocks, 4-7 = B row blocks
            const int i = piece & 3;
            if (piece < 4) glds16_s(voA[i], uniform_ptr(baseA + f * (QK * 2)), lds_addr_of(&sm3.A[st][(wave * 4 + i) * 1024]));
            else glds16_s(voB[i], uniform_ptr(baseB + f * (QK * 2)), lds_addr_of(&sm3.B[st][(wave * 4 + i) * 1024]));
        };
        f32x16 acc[4][2];
        if constexpr (NA == 3) nt256_mainloop3(sm3, acc, Kc / QK, wm, wn, lane, dma);
        else nt256_mainloop(sm3, acc, Kc / QK, wm, wn, lane, dma, k > 0);
        if (k + 1 < PER) {   // the next column tile's first chunk travels during this epilogue
            const char* baseBn = baseB + (int64_t)QN * Kc * 2;
#pragma unroll
            for (int piece = 0; piece < 8; ++piece) {
                const int i = piece & 3;
                if (piece < 4) glds16_s(voA[i], uniform_ptr(baseA), lds_addr_of(&sm.A[0][(wave * 4 + i) * 1024]));
                else glds16_s(voB[i], uniform_ptr(baseBn), lds_addr_of(&sm.B[0][(wave * 4 + i) * 1024]));
            }
        }
        bf16_t* ob = C + t0 * ldc + n0;
        auto emit = [&](int row_u, int rl, int lane_col, const f32x4& lo, const f32x4& hi, int) {
            bf16_t* o = ob + (int64_t)row_u * ldc + ((uint32_t)rl * (uint32_t)ldc + (uint32_t)lane_col);
            f32x4 a = lo, b = hi;
            if (bias) {
                a += *reinterpret_cast<const f32x4*>(bias + n0 + lane_col);
                b += *reinterpret_cast<const f32x4*>(bias + n0 + lane_col + 4);
            }
            st8_bf16(o, a, b);
        };
        nt256_epilogue(acc, sm, wave, wm, wn, lane, (int)((T - t0 < QM) ? (T - t0) : QM), emit, PER > 1);
    }
}
// Launches the 256-tile NT kernel.  Measured at config 2 (profiles/r05v): persistence pays on the SHORT contraction with many column tiles
// (K = 512, N = 2048: 0.706 -> 0.655 ms, and beats the 128 x 256 x 32 kernel there), and costs on the long one (K = 2048, 2 column tiles:
// 0.61 -> 0.67 ms: the 1-MiB A tile is re-read in sequence instead of side by side) -- so PER = 4 below 1024, 1 from there on.
// MADELEINE_BF16_LIN_PERSIST=0 forces PER = 1 (A/B switch).
static inline void linb_launch_nt256(hipStream_t s, const bf16_t* A, int64_t lda, const bf16_t* B, const float* bias, bf16_t* C, int64_t ldc,
                                     int64_t T, int Kc, int n_ct, int64_t tiles) {
    static const bool persist = !(getenv("MADELEINE_BF16_LIN_PERSIST") && atoi(getenv("MADELEINE_BF16_LIN_PERSIST")) == 0);
    const int per = (persist && Kc < 1024 && n_ct % 4 == 0) ? 4 : 1;
    const dim3 grid((unsigned)(tiles / per));
    if (per == 4) hipLaunchKernelGGL(linb_nt256_kernel<4>, grid, dim3(512), 0, s, A, lda, B, bias, C, ldc, T, Kc, n_ct, (int)tiles);
    else if (bf16_lin_stages() == 3 || bf16_lin_stages() == 31) hipLaunchKernelGGL((linb_nt256_kernel<1, 3>), grid, dim3(512), 0, s, A, lda, B, bias, C, ldc, T, Kc, n_ct, (int)tiles);
    else hipLaunchKernelGGL((linb_nt256_kernel<1, 2>), grid, dim3(512), 0, s, A, lda, B, bias, C, ldc, T, Kc, n_ct, (int)tiles);
}

// ---- TN product: slab[sp][n0 + m][k0 + n] = sum_{t in split sp} dY[t][n0 + m] X[t][k0 + n] --------------
// Token rows past T: X re-reads row T-1, dY reads a zero row (per-lane address form of the LDS-DMA, last chunk only).
__global__ __launch_bounds__(256, 2) void linb_tn_kernel(const bf16_t* __restrict__ dY, int64_t lddy, const bf16_t* __restrict__ X,
                                                         int64_t ldx, const bf16_t* __restrict__ zrow, float* __restrict__ slab,
                                                         int64_t T, int N, int K, int64_t tok_per_split, int n_splits, int n_tiles) {
    __shared__ SmemTN sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int n_kt = (K + 255) / 256, n_nt = N / 128;
    const int kt = lid % n_kt, ntile = (lid / n_kt) % n_nt, sp = lid / (n_kt * n_nt);
    if (sp >= n_splits) return;
    const int n0 = ntile * 128, k0 = kt * 256;
    const int64_t ts = (int64_t)sp * tok_per_split;
    int64_t te = ts + tok_per_split;
    if (te > T) te = T;
    const int64_t nch = (te > ts) ? (te - ts + TNK - 1) / TNK : 0;

    const char* baseA = reinterpret_cast<const char*>(dY + ts * lddy + n0);
    const char* baseB = reinterpret_cast<const char*>(X + ts * ldx + k0);
    const uint32_t ldA2 = (uint32_t)lddy * 2u, ldB2 = (uint32_t)ldx * 2u;
    const uint32_t cA = lane & 15, cB = lane & 31;
    uint32_t kA[2], kB[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) kA[q] = (wave * 2 + q) * 4 + (lane >> 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) kB[q] = (wave * 4 + q) * 2 + (lane >> 5);
    auto uptr = [](const char* p) {   // the chunk base is wave-uniform: keep it in SGPRs for the saddr form
        const uint64_t v = (uint64_t)p;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
        return (const char*)(((uint64_t)hi << 32) | lo);
    };
    auto dma = [&](int st, int64_t f, int piece) {
        const int64_t left = T - 1 - (ts + f * TNK);   // index of the last valid token row within this chunk
        if (piece < 2) {
            const uint32_t k = kA[piece];
            const uint32_t cs = (cA ^ ((k & 3) << 2)) << 4;
            if (left < TNK - 1) {   // uniform: only the chunk that crosses T
                const char* p = ((int64_t)k <= left) ? baseA + (f * TNK + k) * (int64_t)ldA2 + cs : reinterpret_cast<const char*>(zrow);
                glds16(p, &sm.A[st][(wave * 2 + piece) * 512]);
            } else {
                glds16_s(k * ldA2 + cs, uptr(baseA + f * TNK * (int64_t)ldA2), lds_addr_of(&sm.A[st][(wave * 2 + piece) * 512]));
            }
        } else {
            const int q = piece - 2;
            uint32_t k = kB[q];
            uint32_t cg = cB ^ ((k & 3) << 2);        // 16-B chunk of the X row this lane fetches (source-side swizzle)
            if (k0 + (int)cg * 8 >= K) cg = 0;        // ragged last column tile: re-read chunk 0, those columns are not stored
            const uint32_t cs = cg << 4;
            if (left < TNK - 1) k = ((int64_t)k <= left) ? k : (uint32_t)(left > 0 ? left : 0);
            glds16_s(k * ldB2 + cs, uptr(baseB + f * TNK * (int64_t)ldB2), lds_addr_of(&sm.B[st][(wave * 4 + q) * 512]));
        }
    };
    f32x16 acc[2][4];
    zero_acc8(acc);
    tn_mainloop(sm, acc, nch, wm, wn, lane, dma);

    float* __restrict__ so = slab + (int64_t)sp * N * K;
    const int l32 = lane & 31;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nr = n0 + wm * 64 + rt * 32 + acc_row(r, lane);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int kc = k0 + wn * 128 + ct * 32 + l32;
                if (kc < K) so[(int64_t)nr * K + kc] = acc[rt][ct][r];
            }
        }
}

__global__ __launch_bounds__(256) void linb_reduce_kernel(const float* __restrict__ slab, float* __restrict__ out, int64_t n, int S) {
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

// column sums of dY (bf16) -> part[b][n] (fp32) over row blocks; each thread owns 4 columns, 8 row lanes per column group
__global__ __launch_bounds__(256) void linb_colsum_part_kernel(const bf16_t* __restrict__ dY, int64_t ldy, int64_t T, int N,
                                                               float* __restrict__ part, int64_t rows_per_block) {
    __shared__ f32x4 red[256];
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block;
    if (r1 > T) r1 = T;
    const int cg = threadIdx.x & 31, rg = threadIdx.x >> 5;   // 32 column groups x 8 row lanes per pass of 128 columns
    for (int cb = 0; cb < N; cb += 128) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        const int c = cb + cg * 4;
        if (c < N)
            for (int64_t r = r0 + rg; r < r1; r += 8) s += ld4(dY + r * ldy + c);
        red[threadIdx.x] = s;
        __syncthreads();
        if (rg == 0 && c < N) {
#pragma unroll
            for (int i = 1; i < 8; ++i) s += red[i * 32 + cg];
            *reinterpret_cast<f32x4*>(part + (int64_t)blockIdx.x * N + c) = s;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void linb_colsum_final_kernel(const float* __restrict__ part, int nb, int N, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float s = 0.f;
    for (int b = 0; b < nb; ++b) s += part[(int64_t)b * N + c];
    out[c] = s;
}

static inline int64_t up16l(int64_t b) { return (b + 15) & ~(int64_t)15; }
constexpr int LINB_COLSUM_BLOCKS = 1024;

// ---- TN product on the 256 x 256 tile (round 5): tn256_mainloop, 64-token chunks; N % 256 == 0 (rows of the slab = dY columns), ragged K
// tail as above.  Token rows past T: dY reads the zero row (per-lane address form, last chunk only), X re-reads row T - 1.
template <int NA>   // NA = 3: tn256_mainloop3 on SmemQ3
__global__ __launch_bounds__(512) void linb_tn256_kernel(const bf16_t* __restrict__ dY, int64_t lddy, const bf16_t* __restrict__ X,
                                                         int64_t ldx, const bf16_t* __restrict__ zrow, float* __restrict__ slab,
                                                         int64_t T, int N, int K, int64_t tok_per_split, int n_splits, int n_tiles) {
    __shared__ typename std::conditional<NA == 3, SmemQ3, SmemQ>::type sm;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int lid = xcd_remap(blockIdx.x, n_tiles);
    const int n_kt = (K + 255) / 256, n_nt = N / 256;
    const int kt = lid % n_kt, ntile = (lid / n_kt) % n_nt, sp = lid / (n_kt * n_nt);
    if (sp >= n_splits) return;
    const int n0 = ntile * 256, k0 = kt * 256;
    const int64_t ts = (int64_t)sp * tok_per_split;
    int64_t te = ts + tok_per_split;
    if (te > T) te = T;
    const int64_t nch = (te > ts) ? (te - ts + TQK - 1) / TQK : 0;

    const char* baseA = reinterpret_cast<const char*>(dY + ts * lddy + n0);
    const char* baseB = reinterpret_cast<const char*>(X + ts * ldx + k0);
    const uint32_t ldA2 = (uint32_t)lddy * 2u, ldB2 = (uint32_t)ldx * 2u;
    uint32_t rowq[4], csA[4], csB[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        int row, src;
        tn256_slot(wave, q, lane, row, src);
        rowq[q] = (uint32_t)row;
        csA[q] = (uint32_t)src << 4;
        csB[q] = (k0 + src * 8 >= K) ? 0u : ((uint32_t)src << 4);   // ragged last column tile: re-read chunk 0, those columns are not stored
    }
    auto dma = [&](int st, int64_t f, int piece) {
        const int q = piece & 3;
        const int64_t left = T - 1 - (ts + f * TQK);   // index of the last valid token row within this chunk
        if (piece < 4) {
            if (left < TQK - 1) {   // uniform: only the chunk that crosses T
                const char* p = ((int64_t)rowq[q] <= left) ? baseA + (f * TQK + rowq[q]) * (int64_t)ldA2 + csA[q] : reinterpret_cast<const char*>(zrow);
                glds16(p, &sm.A[st][(wave * 4 + q) * 1024]);
            } else {
                glds16_s(rowq[q] * ldA2 + csA[q], uniform_ptr(baseA + f * TQK * (int64_t)ldA2), lds_addr_of(&sm.A[st][(wave * 4 + q) * 1024]));
            }
        } else {
            uint32_t k = rowq[q];
            if (left < TQK - 1) k = ((int64_t)k <= left) ? k : (uint32_t)(left > 0 ? left : 0);
            glds16_s(k * ldB2 + csB[q], uniform_ptr(baseB + f * TQK * (int64_t)ldB2), lds_addr_of(&sm.B[st][(wave * 4 + q) * 1024]));
        }
    };
    f32x16 acc[4][2];
    if constexpr (NA == 3) tn256_mainloop3(sm, acc, nch, wm, wn, lane, dma);
    else tn256_mainloop(sm, acc, nch, wm, wn, lane, dma);

    float* __restrict__ so = slab + (int64_t)sp * N * K;
    const int l32 = lane & 31;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int nr = n0 + wm * 128 + rt * 32 + acc_row(r, lane);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int kc = k0 + wn * 64 + ct * 32 + l32;
                if (kc < K) so[(int64_t)nr * K + kc] = acc[rt][ct][r];
            }
        }
}

struct LinbWs {
    int S;
    int64_t tps;
    int64_t oW, oWT, oslab, ozrow, ocpart, total;
};
// the 256-tile TN kernel: output rows (N) in whole 256-row tiles and enough tokens to amortise one workgroup per CU
static inline bool linb_tn_use_q(int64_t T, int N) {
    static const bool off = getenv("MADELEINE_BF16_TN256") && atoi(getenv("MADELEINE_BF16_TN256")) == 0;   // A/B switch
    return !off && T >= 16384 && N % 256 == 0;
}
static inline LinbWs linb_ws(int64_t T, int N, int K) {
    LinbWs w;
    const bool q = linb_tn_use_q(T, N);
    const int tiles = (N / (q ? 256 : 128)) * ((K + 255) / 256);
    const int chunk = q ? TQK : TNK;
    w.S = splits_for(T, tiles > 0 ? tiles : 1, q ? 256 : 512);
    int64_t tps = (T + w.S - 1) / w.S;
    w.tps = ((tps + chunk - 1) / chunk) * chunk;
    if (w.tps < chunk) w.tps = chunk;
    w.S = (int)((T + w.tps - 1) / w.tps);
    if (w.S < 1) w.S = 1;
    int64_t o = 0;
    w.oW = o; o += up16l((int64_t)N * K * 2);
    w.oWT = o; o += up16l((int64_t)N * K * 2);
    w.oslab = o; o += up16l((int64_t)w.S * N * K * 4);
    w.ozrow = o; o += 64;
    w.ocpart = o; o += up16l((int64_t)LINB_COLSUM_BLOCKS * N * 4);
    w.total = o + 64;
    return w;
}
// 256 x 256 x 64 tile: contractions >= 1024 (measured: K = 2048 0.70 -> 0.61 ms), and -- persistent over 4 column tiles, see
// linb_launch_nt256 -- contractions of 512..1023 when the output has a multiple of 4 column tiles.  MADELEINE_BF16_LIN256_MINK=1024
// restores the round-4 rule (A/B switch).
static inline bool linb_use_q(int64_t T, int64_t Kc, int64_t n_out) {
    static const int64_t mink = getenv("MADELEINE_BF16_LIN256_MINK") ? atoll(getenv("MADELEINE_BF16_LIN256_MINK")) : 512;
    if (T < 4096 || (Kc % QK) != 0 || (n_out % QN) != 0) return false;
    return Kc >= 1024 || (Kc >= mink && (n_out / QN) % 4 == 0);
}
static inline bool linb_geom_fwd(int64_t N, int64_t K) { return N > 0 && K > 0 && N % 128 == 0 && K % BBK == 0 && N <= (1 << 20) && K <= (1 << 20); }
static inline bool linb_geom_bwd(int64_t N, int64_t K) { return linb_geom_fwd(N, K); }

}  // namespace mdl

using namespace mdl;

extern "C" int mdl_linear_bf16_supported(int64_t N, int64_t K, int backward) {
    return (backward ? linb_geom_bwd(N, K) : linb_geom_fwd(N, K)) ? 1 : 0;
}

extern "C" int64_t mdl_linear_fwd_bf16_ws_bytes(int64_t T, int64_t N, int64_t K) {
    if (T < 0 || N < 1 || K < 1) return MDL_E_ARG;
    return up16l(N * K * 2) + 64;
}

extern "C" int mdl_linear_fwd_bf16(const uint16_t* X, int64_t ldx, const float* W, const float* bias, uint16_t* Y, int64_t ldy,
                                   int64_t T, int64_t N, int64_t K, void* ws, void* stream) {
    if (!X || !W || !Y || !ws || T < 0 || N < 1 || K < 1 || ldx < K || ldy < N) return MDL_E_ARG;
    if (!linb_geom_fwd(N, K) || (ldx & 7) || (ldy & 7)) return MDL_E_UNSUPPORTED;
    if (!host_aligned16(X) || !host_aligned16(W) || !host_aligned16(Y) || !host_aligned16(ws) || (bias && !host_aligned16(bias)))
        return MDL_E_ALIGN;
    if (T == 0) return MDL_OK;
    hipStream_t s = (hipStream_t)stream;
    bf16_t* Wb = (bf16_t*)ws;
    hipLaunchKernelGGL(linb_w_cast_kernel, dim3((unsigned)((N * K / 4 + 255) / 256)), dim3(256), 0, s, W, Wb, N * K);
    MDL_LAUNCH_CHECK();
    const bool wide = (N % BBN) == 0;
    if (wide && linb_use_q(T, K, N)) {
        const int n_ct = (int)(N / QN);
        const int64_t tiles = ((T + QM - 1) / QM) * n_ct;
        if (tiles > 0x7fffffff) return MDL_E_UNSUPPORTED;
        linb_launch_nt256(s, (const bf16_t*)X, ldx, (const bf16_t*)Wb, bias, (bf16_t*)Y, ldy, T, (int)K, n_ct, tiles);
        MDL_LAUNCH_CHECK();
        return MDL_OK;
    }
    const int n_ct = (int)(N / (wide ? BBN : 128));
    const int64_t tiles = ((T + BBM - 1) / BBM) * n_ct;
    if (tiles > 0x7fffffff) return MDL_E_UNSUPPORTED;
    if (wide)
        hipLaunchKernelGGL(linb_nt_kernel<4>, dim3((unsigned)tiles), dim3(256), 0, s, (const bf16_t*)X, ldx, (const bf16_t*)Wb, bias,
                           (bf16_t*)Y, ldy, T, (int)K, (int)N, n_ct, (int)tiles);
    else
        hipLaunchKernelGGL(linb_nt_kernel<2>, dim3((unsigned)tiles), dim3(256), 0, s, (const bf16_t*)X, ldx, (const bf16_t*)Wb, bias,
                           (bf16_t*)Y, ldy, T, (int)K, (int)N, n_ct, (int)tiles);
    MDL_LAUNCH_CHECK();
    return MDL_OK;
}

extern "C" int64_t mdl_linear_bwd_bf16_ws_bytes(int64_t T, int64_t N, int64_t K) {
    if (T < 0 || N < 1 || K < 1 || N > (1 << 20) || K > (1 << 20)) return MDL_E_ARG;
    return linb_ws(T, (int)N, (int)K).total;
}

extern "C" int mdl_linear_bwd_bf16(const uint16_t* X, int64_t ldx, const float* W, const uint16_t* dY, int64_t lddy, uint16_t* dX,
                                   int64_t lddx, float* dW, float* dbias, int64_t T, int64_t N, int64_t K, void* ws, void* stream) {
    if (!X || !W || !dY || !dW || !ws || T < 0 || N < 1 || K < 1 || ldx < K || lddy < N || (dX && lddx < K)) return MDL_E_ARG;
    if (!linb_geom_bwd(N, K) || (ldx & 7) || (lddy & 7) || (dX && (lddx & 7))) return MDL_E_UNSUPPORTED;
    if (!host_aligned16(X) || !host_aligned16(W) || !host_aligned16(dY) || !host_aligned16(dW) || !host_aligned16(ws) ||
        (dX && !host_aligned16(dX)))
        return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const LinbWs L = linb_ws(T, (int)N, (int)K);
    char* base = (char*)ws;
    bf16_t* WT = (bf16_t*)(base + L.oWT);
    float* slab = (float*)(base + L.oslab);
    bf16_t* zrow = (bf16_t*)(base + L.ozrow);
    float* cpart = (float*)(base + L.ocpart);
    if (T == 0) {
        hipError_t e = hipMemsetAsync(dW, 0, (size_t)N * K * 4, s);
        if (e != hipSuccess) return (int)e;
        if (dbias) {
            e = hipMemsetAsync(dbias, 0, (size_t)N * 4, s);
            if (e != hipSuccess) return (int)e;
        }
        return MDL_OK;
    }
    {
        const hipError_t e = hipMemsetAsync(zrow, 0, 64, s);
        if (e != hipSuccess) return (int)e;
    }
    if (dX) {   // dX = dY W: NT with B = W^T rows [K][N]
        hipLaunchKernelGGL(linb_w_transpose_kernel, dim3((unsigned)(K / 32), (unsigned)(N / 32)), dim3(256), 0, s, W, WT, (int)N, (int)K);
        MDL_LAUNCH_CHECK();
        if ((K % QN) == 0 && linb_use_q(T, N, K)) {
            const int n_ct = (int)(K / QN);
            const int64_t tiles = ((T + QM - 1) / QM) * n_ct;
            if (tiles > 0x7fffffff) return MDL_E_UNSUPPORTED;
            linb_launch_nt256(s, (const bf16_t*)dY, lddy, (const bf16_t*)WT, nullptr, (bf16_t*)dX, lddx, T, (int)N, n_ct, tiles);
            MDL_LAUNCH_CHECK();
        } else {
        const int n_ct = (int)((K + BBN - 1) / BBN);
        const int64_t tiles = ((T + BBM - 1) / BBM) * n_ct;
        if (tiles > 0x7fffffff) return MDL_E_UNSUPPORTED;
        hipLaunchKernelGGL(linb_nt_kernel<4>, dim3((unsigned)tiles), dim3(256), 0, s, (const bf16_t*)dY, lddy, (const bf16_t*)WT,
                           (const float*)nullptr, (bf16_t*)dX, lddx, T, (int)N, (int)K, n_ct, (int)tiles);
        MDL_LAUNCH_CHECK();
        }
    }
    {
        const bool q = linb_tn_use_q(T, (int)N);
        const int64_t tiles = (int64_t)L.S * (N / (q ? 256 : 128)) * ((K + 255) / 256);
        if (tiles > 0x7fffffff) return MDL_E_UNSUPPORTED;
        if (q && (bf16_lin_stages() == 3 || bf16_lin_stages() == 32))
            hipLaunchKernelGGL(linb_tn256_kernel<3>, dim3((unsigned)tiles), dim3(512), 0, s, (const bf16_t*)dY, lddy, (const bf16_t*)X, ldx,
                               (const bf16_t*)zrow, slab, T, (int)N, (int)K, L.tps, L.S, (int)tiles);
        else if (q)
            hipLaunchKernelGGL(linb_tn256_kernel<2>, dim3((unsigned)tiles), dim3(512), 0, s, (const bf16_t*)dY, lddy, (const bf16_t*)X, ldx,
                               (const bf16_t*)zrow, slab, T, (int)N, (int)K, L.tps, L.S, (int)tiles);
        else
        hipLaunchKernelGGL(linb_tn_kernel, dim3((unsigned)tiles), dim3(256), 0, s, (const bf16_t*)dY, lddy, (const bf16_t*)X, ldx,
                           (const bf16_t*)zrow, slab, T, (int)N, (int)K, L.tps, L.S, (int)tiles);
        MDL_LAUNCH_CHECK();
        const int64_t n = N * K;
        hipLaunchKernelGGL(linb_reduce_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, s, (const float*)slab, dW, n, L.S);
        MDL_LAUNCH_CHECK();
    }
    if (dbias) {
        int nb = (int)((T + 511) / 512);
        if (nb > LINB_COLSUM_BLOCKS) nb = LINB_COLSUM_BLOCKS;
        const int64_t rpb = (T + nb - 1) / nb;
        nb = (int)((T + rpb - 1) / rpb);
        hipLaunchKernelGGL(linb_colsum_part_kernel, dim3(nb), dim3(256), 0, s, (const bf16_t*)dY, lddy, T, (int)N, cpart, rpb);
        MDL_LAUNCH_CHECK();
        hipLaunchKernelGGL(linb_colsum_final_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, (const float*)cpart, nb, (int)N, dbias);
        MDL_LAUNCH_CHECK();
    }
    return MDL_OK;
}
