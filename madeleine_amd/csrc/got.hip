// got.hip -- G0-G3 (placeholder until the IPOT/GW kernels land; returns MDL_E_UNSUPPORTED)
#include "common.hpp"
extern "C" int64_t mdl_got_ws_bytes(int k, int n, int d) { (void)k; (void)n; (void)d; return MDL_E_UNSUPPORTED; }
extern "C" int mdl_got_fwd(const float*, const float*, float*, float*, const float*, int, int, int, void*, void*) { return MDL_E_UNSUPPORTED; }
extern "C" int mdl_got_bwd(const float*, const float*, const float*, float*, float*, float*, const float*, int, int, int, void*, void*) { return MDL_E_UNSUPPORTED; }
