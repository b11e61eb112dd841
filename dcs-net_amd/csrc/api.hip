// api.hip — version / error-string entry points of include/dcsnet_hip.h.
#include "dcs_common.h"

extern "C" int dcs_abi_version(void) { return 2; }

extern "C" const char* dcs_error_string(int code) {
    switch (code) {
        case DCS_OK: return "ok";
        case DCS_ERR_BADARG: return "bad argument (null pointer, non-positive dimension or unsupported geometry)";
        case DCS_ERR_LAUNCH: return "HIP kernel launch failed";
        case DCS_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}
