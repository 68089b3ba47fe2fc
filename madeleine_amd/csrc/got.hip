// got.hip -- C ABI of the Graph Optimal Transport kernels (G0-G3, reference madeleine/utils/loss.py:162-302).
// The kernels live in got_impl.inc, parametrised by the workgroup size and compiled in got_t1024.hip (1024 threads =
// 16 waves x 128 VGPRs; a 512-thread build with 256 VGPRs per wave measured slower at every n once the chain for
// n > 128 was split into per-phase launches, tools/got_ab.py); this file validates arguments and launches.
#include "common.hpp"
#include "got_batch.hpp"

namespace mdl {
#define MDL_GOT_DECL(NS)                                         \
    namespace NS {                                               \
    int launch_prep(const GotBatch&, hipStream_t);               \
    int launch_main(const GotBatch&, hipStream_t);               \
    int launch_bwd_begin(const GotBatch&, hipStream_t);          \
    int launch_bwd_finish(const GotBatch&, hipStream_t);         \
    int64_t ws_floats(int, int, int);                            \
    }
MDL_GOT_DECL(got1024)
#undef MDL_GOT_DECL
constexpr int GOT_MAXN = 512;   // 256 < n <= 512: the workspace-resident class of got_impl.inc
constexpr int GOT_MAXD = 128;
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
    return got1024::ws_floats(k, n, d) * 4 + 64;
}

static GotBatch one(float* ws, int k, int n, int d) {
    GotBatch B{};
    B.np = 1;
    B.d = d;
    B.p[0].ws = ws;
    B.p[0].k = k;
    B.p[0].n = n;
    return B;
}

extern "C" int mdl_got_fwd(const float* V, const float* Q, float* out, float* minmax_out, const float* minmax_in, int k, int n,
                           int d, void* ws, void* stream) {
    int rc = got_check(k, n, d);
    if (rc) return rc;
    if (!V || !Q || !out || !ws) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (k == 0 || n == 0) {
        const hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(float), s);
        return e == hipSuccess ? MDL_OK : (int)e;
    }
    GotBatch B = one((float*)ws, k, n, d);
    B.p[0].V = V;
    B.p[0].Q = Q;
    B.p[0].out = out;
    B.p[0].mm_out = minmax_out;
    B.p[0].mm_in = minmax_in;
    rc = got1024::launch_prep(B, s);
    if (rc) return rc;
    return got1024::launch_main(B, s);
}

extern "C" int mdl_got_extrema(const float* V, const float* Q, float* minmax_out, int k, int n, int d, void* ws,
                               void* stream) {
    const int rc = got_check(k, n, d);
    if (rc) return rc;
    if (!V || !Q || !minmax_out || !ws) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    if (k == 0 || n == 0) return MDL_E_ARG;  // extrema of an empty batch are undefined
    GotBatch B = one((float*)ws, k, n, d);
    B.p[0].V = V;
    B.p[0].Q = Q;
    B.p[0].mm_out = minmax_out;
    return got1024::launch_prep(B, (hipStream_t)stream);
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
    GotBatch B = one((float*)ws, k, n, d);
    B.p[0].d_out = d_out;
    B.p[0].d_mm = d_minmax;
    return got1024::launch_bwd_begin(B, s);
}

extern "C" int mdl_got_bwd_finish(const float* V, const float* Q, float* dV, float* dQ, const float* d_minmax_total, int k,
                                  int n, int d, void* ws, void* stream) {
    const int rc = got_check(k, n, d);
    if (rc) return rc;
    if (!V || !Q || !dV || !dQ || !ws) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    if (k == 0 || n == 0) return MDL_OK;
    GotBatch B = one((float*)ws, k, n, d);
    B.p[0].V = V;
    B.p[0].Q = Q;
    B.p[0].dV = dV;
    B.p[0].dQ = dQ;
    B.p[0].d_mm_total = d_minmax_total;
    return got1024::launch_bwd_finish(B, (hipStream_t)stream);
}

extern "C" int mdl_got_bwd(const float* V, const float* Q, const float* d_out, float* dV, float* dQ, int k, int n, int d,
                           void* ws, void* stream) {
    int rc = mdl_got_bwd_begin(d_out, nullptr, k, n, d, ws, stream);
    if (rc) return rc;
    return mdl_got_bwd_finish(V, Q, dV, dQ, nullptr, k, n, d, ws, stream);
}

// ---- several problems in one launch sequence (got_batch.hpp): 1 <= np <= MDL_GOT_MAX_BATCH non-empty problems, every n <= 256 ----
static int batch_begin(GotBatch& B, int np, const int* k, const int* n, int d, void* const* ws) {
    if (np < 1 || np > GOT_MAXP || !k || !n || !ws) return MDL_E_ARG;
    B = GotBatch{};
    B.np = np;
    B.d = d;
    for (int p = 0; p < np; ++p) {
        const int rc = got_check(k[p], n[p], d);
        if (rc) return rc;
        if (k[p] < 1 || n[p] < 1) return MDL_E_ARG;           // the caller filters empty problems
        if (n[p] > 256) return MDL_E_UNSUPPORTED;               // the workspace-resident class is not batched
        if (!ws[p]) return MDL_E_ARG;
        if (!host_aligned16(ws[p])) return MDL_E_ALIGN;
        B.p[p].ws = (float*)ws[p];
        B.p[p].k = k[p];
        B.p[p].n = n[p];
    }
    return MDL_OK;
}

extern "C" int mdl_got_extrema_multi(int np, const float* const* V, const float* const* Q, float* const* minmax_out, const int* k,
                                     const int* n, int d, void* const* ws, void* stream) {
    GotBatch B;
    const int rc = batch_begin(B, np, k, n, d, ws);
    if (rc) return rc;
    if (!V || !Q || !minmax_out) return MDL_E_ARG;
    for (int p = 0; p < np; ++p) {
        if (!V[p] || !Q[p] || !minmax_out[p]) return MDL_E_ARG;
        B.p[p].V = V[p];
        B.p[p].Q = Q[p];
        B.p[p].mm_out = minmax_out[p];
    }
    return got1024::launch_prep(B, (hipStream_t)stream);
}

extern "C" int mdl_got_fwd_multi(int np, const float* const* V, const float* const* Q, float* const* out, const float* const* minmax_in,
                                 const int* k, const int* n, int d, void* const* ws, void* stream) {
    GotBatch B;
    int rc = batch_begin(B, np, k, n, d, ws);
    if (rc) return rc;
    if (!V || !Q || !out) return MDL_E_ARG;
    for (int p = 0; p < np; ++p) {
        if (!V[p] || !Q[p] || !out[p]) return MDL_E_ARG;
        B.p[p].V = V[p];
        B.p[p].Q = Q[p];
        B.p[p].out = out[p];
        B.p[p].mm_in = minmax_in ? minmax_in[p] : nullptr;
    }
    rc = got1024::launch_prep(B, (hipStream_t)stream);
    if (rc) return rc;
    return got1024::launch_main(B, (hipStream_t)stream);
}

extern "C" int mdl_got_bwd_begin_multi(int np, const float* const* d_out, float* const* d_minmax, const int* k, const int* n, int d,
                                       void* const* ws, void* stream) {
    GotBatch B;
    const int rc = batch_begin(B, np, k, n, d, ws);
    if (rc) return rc;
    if (!d_out) return MDL_E_ARG;
    for (int p = 0; p < np; ++p) {
        if (!d_out[p]) return MDL_E_ARG;
        B.p[p].d_out = d_out[p];
        B.p[p].d_mm = d_minmax ? d_minmax[p] : nullptr;
    }
    return got1024::launch_bwd_begin(B, (hipStream_t)stream);
}

extern "C" int mdl_got_bwd_finish_multi(int np, const float* const* V, const float* const* Q, float* const* dV, float* const* dQ,
                                        const float* const* d_minmax_total, const int* k, const int* n, int d, void* const* ws,
                                        void* stream) {
    GotBatch B;
    const int rc = batch_begin(B, np, k, n, d, ws);
    if (rc) return rc;
    if (!V || !Q || !dV || !dQ) return MDL_E_ARG;
    for (int p = 0; p < np; ++p) {
        if (!V[p] || !Q[p] || !dV[p] || !dQ[p]) return MDL_E_ARG;
        B.p[p].V = V[p];
        B.p[p].Q = Q[p];
        B.p[p].dV = dV[p];
        B.p[p].dQ = dQ[p];
        B.p[p].d_mm_total = d_minmax_total ? d_minmax_total[p] : nullptr;
    }
    return got1024::launch_bwd_finish(B, (hipStream_t)stream);
}
