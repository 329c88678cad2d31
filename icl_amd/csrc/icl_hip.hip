// icl_hip.hip — gfx950 translation unit of libicl_hip.so (C ABI in include/icl_hip.h).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC (see icl_amd/build.py).
#include "device_env_hip.h"
#include "../../include/icl_hip.h"
#include "icl_abi.inc"
