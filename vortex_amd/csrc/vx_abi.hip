// Library-level entry points of libvortex_hip.so (see include/vortex_hip.h).
#include "vx_common.h"

extern "C" int vx_version(void) { return 100; }   // 0.1.0

extern "C" const char *vx_error_string(int code) {
    switch (code) {
        case VX_OK: return "ok";
        case VX_ERR_INVALID_ARG: return "invalid argument (null pointer, non-positive dimension or bad enum)";
        case VX_ERR_UNSUPPORTED: return "a documented limit of libvortex_hip is exceeded";
        case VX_ERR_WORKSPACE: return "workspace smaller than the matching *_workspace_bytes() query";
        case VX_ERR_LAUNCH: return "HIP reported an error for a kernel launch";
        case VX_ERR_ALIGNMENT: return "a device pointer is not aligned as include/vortex_hip.h requires";
        default: return "unknown libvortex_hip error code";
    }
}
