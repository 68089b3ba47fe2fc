// version.hip -- library identification
#include "common.hpp"
extern "C" const char* mdl_version(void) { return "madeleine_amd 0.2 gfx950"; }
extern "C" int mdl_abi_version(void) { return MDL_ABI_VERSION; }

// CU-masked streams (include/madeleine_amd.h): runtime plumbing, no kernel
extern "C" int mdl_stream_create_cu_mask(uint32_t n_words, const uint32_t* mask, void** stream_out) {
    if (!mask || !stream_out || n_words < 1 || n_words > 32) return MDL_E_ARG;
    hipStream_t s = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&s, n_words, mask);
    if (e != hipSuccess) return (int)e;
    *stream_out = (void*)s;
    return MDL_OK;
}
extern "C" int mdl_stream_destroy(void* stream) {
    if (!stream) return MDL_E_ARG;
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    return e == hipSuccess ? MDL_OK : (int)e;
}
