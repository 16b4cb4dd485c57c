// Status strings / version of the vqk C-ABI (include/vqk.h).
#include <stdint.h>
#include "common.h"

extern "C" {

const char* vqk_status_str(int status) {
    switch (status) {
        case VQK_OK: return "ok";
        case VQK_ERR_SHAPE: return "unsupported or inconsistent shape";
        case VQK_ERR_DTYPE: return "unsupported dtype";
        case VQK_ERR_ALIGN: return "pointer is not 16-byte aligned";
        case VQK_ERR_LAUNCH: return "kernel launch failed";
        case VQK_ERR_ARG: return "bad argument";
        case VQK_ERR_WORKSPACE: return "workspace too small";
        default: return "unknown status";
    }
}

int vqk_version(void) { return 1; }
const char* vqk_arch(void) { return "gfx950"; }

}  // extern "C"

// deterministic mode: see include/vqk.h
namespace vqkd {
DetState& det_state() {
    static thread_local DetState st = {0, nullptr, 0};
    return st;
}
DetState& scratch_state() {
    static thread_local DetState st = {0, nullptr, 0};
    return st;
}
}  // namespace vqkd

extern "C" int vqk_set_scratch(void* ws, int64_t ws_bytes) {
    if (ws && ws_bytes < 0) return VQK_ERR_ARG;
    if (ws && (reinterpret_cast<uintptr_t>(ws) & 15)) return VQK_ERR_ALIGN;       // the split slices are written with 16-byte stores
    vqkd::DetState& d = vqkd::scratch_state();
    d.on = ws ? 1 : 0;
    d.ws = reinterpret_cast<float*>(ws);
    d.bytes = ws ? ws_bytes : 0;
    return VQK_OK;
}

extern "C" int vqk_set_deterministic(int on, void* ws, int64_t ws_bytes) {
    if (on && ws && ws_bytes < 0) return VQK_ERR_ARG;
    if (on && ws && (reinterpret_cast<uintptr_t>(ws) & 15)) return VQK_ERR_ALIGN;
    vqkd::DetState& d = vqkd::det_state();
    d.on = on ? 1 : 0;
    d.ws = (on && ws) ? reinterpret_cast<float*>(ws) : nullptr;
    d.bytes = (on && ws) ? ws_bytes : 0;
    return VQK_OK;
}
