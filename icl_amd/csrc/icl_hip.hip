// icl_hip.hip — gfx950 translation unit of libicl_hip.so (C ABI in include/icl_hip.h).
// Build: hipcc --offload-arch=gfx950 -O3 -fPIC, linked with icl_hip_noslp.hip (see icl_amd/build.py).
#define ICL_TWO_UNITS 1
#include "device_env_hip.h"
#include "../../include/icl_hip.h"
#include "icl_abi.inc"
