// got.hip -- C ABI of the Graph Optimal Transport kernels (G0-G3, reference madeleine/utils/loss.py:162-302).
// The kernels live in got_impl.inc, parametrised by the workgroup size and compiled in got_t1024.hip (1024 threads =
// 16 waves x 128 VGPRs; a 512-thread build with 256 VGPRs per wave measured slower at every n once the chain for
// n > 128 was split into per-phase launches, tools/got_ab.py); this file validates arguments and launches.
#include "common.hpp"

namespace mdl {
#define MDL_GOT_DECL(NS)                                                                                                      \
    namespace NS {                                                                                                            \
    int launch_prep(const float*, const float*, float*, float*, const float*, int, int, int, hipStream_t);                    \
    int launch_main(float*, float*, int, int, int, hipStream_t);                                                              \
    int launch_bwd_begin(float*, const float*, float*, int, int, int, hipStream_t);                                           \
    int launch_bwd_finish(const float*, const float*, float*, float*, float*, const float*, int, int, int, hipStream_t);      \
    int64_t ws_floats(int, int, int);                                                                                         \
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
    float* w = (float*)ws;
    rc = got1024::launch_prep(V, Q, w, minmax_out, minmax_in, k, n, d, s);
    if (rc) return rc;
    return got1024::launch_main(w, out, k, n, d, s);
}

extern "C" int mdl_got_extrema(const float* V, const float* Q, float* minmax_out, int k, int n, int d, void* ws,
                               void* stream) {
    const int rc = got_check(k, n, d);
    if (rc) return rc;
    if (!V || !Q || !minmax_out || !ws) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    if (k == 0 || n == 0) return MDL_E_ARG;  // extrema of an empty batch are undefined
    return got1024::launch_prep(V, Q, (float*)ws, minmax_out, nullptr, k, n, d, (hipStream_t)stream);
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
    return got1024::launch_bwd_begin((float*)ws, d_out, d_minmax, k, n, d, s);
}

extern "C" int mdl_got_bwd_finish(const float* V, const float* Q, float* dV, float* dQ, const float* d_minmax_total, int k,
                                  int n, int d, void* ws, void* stream) {
    const int rc = got_check(k, n, d);
    if (rc) return rc;
    if (!V || !Q || !dV || !dQ || !ws) return MDL_E_ARG;
    if (!host_aligned16(ws)) return MDL_E_ALIGN;
    if (k == 0 || n == 0) return MDL_OK;
    return got1024::launch_bwd_finish(V, Q, (float*)ws, dV, dQ, d_minmax_total, k, n, d, (hipStream_t)stream);
}

extern "C" int mdl_got_bwd(const float* V, const float* Q, const float* d_out, float* dV, float* dQ, int k, int n, int d,
                           void* ws, void* stream) {
    int rc = mdl_got_bwd_begin(d_out, nullptr, k, n, d, ws, stream);
    if (rc) return rc;
    return mdl_got_bwd_finish(V, Q, dV, dQ, nullptr, k, n, d, ws, stream);
}
