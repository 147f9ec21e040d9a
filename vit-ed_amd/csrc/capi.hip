// ABI bookkeeping entry points.
#include "common.h"

extern "C" int vited_abi_version(void) { return VITED_ABI_VERSION; }

extern "C" const char* vited_strerror(int code) {
    switch (code) {
        case VITED_OK: return "ok";
        case VITED_ERR_BAD_ARG: return "bad argument (null pointer, non-positive size, or inconsistent strides)";
        case VITED_ERR_UNSUPPORTED: return "unsupported shape/dtype combination";
        case VITED_ERR_LAUNCH: return "HIP kernel launch failed";
        case VITED_ERR_WORKSPACE: return "workspace too small (see vited_*_workspace_bytes)";
        default: return "unknown vited error code";
    }
}
