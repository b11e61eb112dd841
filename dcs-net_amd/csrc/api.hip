// api.hip — version / error-string entry points of include/dcsnet_hip.h.
#include "dcs_common.h"

extern "C" int dcs_abi_version(void) { return 6; }

extern "C" const char* dcs_error_string(int code) {
    switch (code) {
        case DCS_OK: return "ok";
        case DCS_ERR_BADARG: return "bad argument (null pointer, non-positive dimension or unsupported geometry)";
        case DCS_ERR_LAUNCH: return "HIP kernel launch failed";
        case DCS_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown error";
    }
}

hipError_t dcs_ensure_dynamic_lds(const void* fn, size_t bytes) {
    struct Entry { const void* fn; size_t bytes; };
    static Entry table[64];
    static int n = 0;
    if (bytes <= 64 * 1024) return hipSuccess;
    for (int i = 0; i < n; ++i)
        if (table[i].fn == fn) {
            if (table[i].bytes >= bytes) return hipSuccess;
            const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
            if (e == hipSuccess) table[i].bytes = bytes;
            return e;
        }
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess && n < 64) { table[n].fn = fn; table[n].bytes = bytes; ++n; }
    return e;
}

namespace { int g_conv_precision = 0; }

int dcs_conv_precision() { return g_conv_precision; }

extern "C" int dcs_set_conv_precision(int mode) {
    if (mode != 0 && mode != 1) return DCS_ERR_BADARG;
    g_conv_precision = mode;
    return DCS_OK;
}

extern "C" int dcs_get_conv_precision(void) { return g_conv_precision; }
