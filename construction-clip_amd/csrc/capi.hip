// ABI version + small utilities of libcclip_hip.so.
#include "cclip_common.h"
#include "../../include/cclip_hip.h"

extern "C" int cclip_abi_version(void) { return CCLIP_ABI_VERSION; }
