// Identity entry points of libsedhip.so (include/sednet_hip.h).
#include "common.h"
extern "C" int sed_abi_version(void) { return 8; }
extern "C" const char* sed_build_arch(void) { return "gfx950"; }
