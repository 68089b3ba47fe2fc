// version.hip -- library identification
#include "common.hpp"
extern "C" const char* mdl_version(void) { return "madeleine_amd 0.2 gfx950"; }
extern "C" int mdl_abi_version(void) { return MDL_ABI_VERSION; }
