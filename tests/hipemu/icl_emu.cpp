// icl_emu.cpp — TEST ONLY: the same C ABI (include/icl_hip.h) with every kernel run on CPU fibers.
// Host pointers instead of device pointers; used by tests/ to check kernels without a GPU.
#include "hipemu.h"
#include "../../include/icl_hip.h"
#include "../../icl_amd/csrc/icl_abi.inc"
