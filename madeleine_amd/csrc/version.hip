// version.hip -- library identification
#include "common.hpp"
extern "C" const char* mdl_version(void) { return "madeleine_amd 0.1 gfx950"; }
