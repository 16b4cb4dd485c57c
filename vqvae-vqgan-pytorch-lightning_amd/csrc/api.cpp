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
DetState& tile_queue_state() {
    static thread_local DetState st = {0, nullptr, 0};
    return st;
}
}  // namespace vqkd

extern "C" int vqk_set_tile_queue(void* ws, int64_t ws_bytes) {
    if (ws && ws_bytes < 64) return VQK_ERR_WORKSPACE;           // eight per-XCD counters + the census word
    if (ws && (reinterpret_cast<uintptr_t>(ws) & 15)) return VQK_ERR_ALIGN;
    vqkd::DetState& d = vqkd::tile_queue_state();
    d.on = ws ? 1 : 0;
    d.ws = reinterpret_cast<float*>(ws);
    d.bytes = ws ? ws_bytes : 0;
    return VQK_OK;
}

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

// ---------------------------------------------------------------- tuning slots (include/vqk.h: vqk_set_tuning)
#include <string.h>
#include <stdlib.h>
namespace vqkd {
static TuneSlot g_tune[] = {
    {"MX", 0, 0},
    {"MX_1X1", 0, 0},
    {"TW16", 0, 0},
    {"STREAM_BLOCKS", 0, 0},
    {"MX_MIN_TILES", 0, 0},
    {"FPROP_SPLITK", 0, 0},
    {"SK_BLOCKS", 0, 0},
    {"SK_MINSTEPS", 0, 0},
    {"SK_MAXMB", 0, 0},
    {"UPS_PHASE", 0, 0},
    {"WGRAD_BLOCKS", 0, 0},
    {"WGMX", 0, 0},
    {"WGRAD_GEN_BLOCKS", 0, 0},
    {"WGRAD_NO_PW16", 0, 0},
    {"WGRAD_NO_P16K", 0, 0},
    {"MX_HALF", 0, 0},
    {"MX_HALF_HW", 0, 0},
    {"UPFIRDN_TILE", 0, 0},
    {"GN_BLOCKS_REDUCE", 0, 0},
    {"GN_BLOCKS_APPLY", 0, 0},
    {"GN_NT_MB", 0, 0},
    {"GN_NO_SMALL", 0, 0},
    {"WGMX_COEF_E4", 0, 0},
    {"GN_CLUSTER_MAX_HW", 0, 0},
    {"COMM_CUS", 0, 0},
    {"MX_QUARTER", 0, 0},
    {"MX_S2", 0, 0},
    {"MX_S2_DGRAD_MIN", 0, 0},
    {"VQ_LDS", 0, 0},
    {"UPS_MERGE", 0, 0},
    {"TILE_QUEUE", 0, 0},
    {"X3_WL", 0, 0},
    {"X3_WGRAD_COEF_E4", 0, 0},
    {"X3_WGRAD_FOLD", 0, 0}
};
static constexpr int kTune = (int)(sizeof(g_tune) / sizeof(g_tune[0]));
TuneSlot* tune_slot(const char* name) {
    for (int i = 0; i < kTune; ++i)
        if (strcmp(g_tune[i].name, name) == 0) return &g_tune[i];
    abort();                                                     // a call site names a slot that is not in the table above
}
}  // namespace vqkd

extern "C" int vqk_set_tuning(const char* name, int value) {
    if (!name) return VQK_ERR_ARG;
    for (int i = 0; i < vqkd::kTune; ++i)
        if (strcmp(vqkd::g_tune[i].name, name) == 0) {
            __atomic_store_n(&vqkd::g_tune[i].value, value, __ATOMIC_RELAXED);
            __atomic_store_n(&vqkd::g_tune[i].is_set, 1, __ATOMIC_RELEASE);
            return VQK_OK;
        }
    return VQK_ERR_ARG;
}

extern "C" int vqk_reset_tuning(void) {
    for (int i = 0; i < vqkd::kTune; ++i) __atomic_store_n(&vqkd::g_tune[i].is_set, 0, __ATOMIC_RELAXED);
    return VQK_OK;
}

extern "C" int vqk_tuning_count(void) { return vqkd::kTune; }

extern "C" const char* vqk_tuning_name(int i) { return (i >= 0 && i < vqkd::kTune) ? vqkd::g_tune[i].name : nullptr; }
