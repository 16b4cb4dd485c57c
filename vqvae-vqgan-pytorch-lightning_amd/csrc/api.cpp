// Status strings / version of the vqk C-ABI (include/vqk.h).
#include <stdint.h>
#include <new>
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

// Workspace contexts (include/vqk.h: vqk_ctx_*): the three caller-owned workspaces a launch may need -- split-K scratch, tile-queue
// words, deterministic-mode slices -- travel together in a context object.  A host thread has ONE current context (its own default
// context until vqk_ctx_make_current names another); the launchers read it through det_state() / scratch_state() /
// tile_queue_state().  The rounds-1-5 setters below are thin wrappers over the thread's current context.
struct vqk_ctx {
    vqkd::DetState det, scratch, tq;
};
namespace vqkd {
static vqk_ctx& default_ctx() {
    static thread_local vqk_ctx c = {{0, nullptr, 0}, {0, nullptr, 0}, {0, nullptr, 0}};
    return c;
}
static vqk_ctx*& current_ctx() {
    static thread_local vqk_ctx* cur = nullptr;
    return cur;
}
static vqk_ctx& ctx() {
    vqk_ctx* c = current_ctx();
    return c ? *c : default_ctx();
}
DetState& det_state() { return ctx().det; }
DetState& scratch_state() { return ctx().scratch; }
DetState& tile_queue_state() { return ctx().tq; }
}  // namespace vqkd

extern "C" int vqk_ctx_create(vqk_ctx** out) {
    if (!out) return VQK_ERR_ARG;
    *out = new (std::nothrow) vqk_ctx{{0, nullptr, 0}, {0, nullptr, 0}, {0, nullptr, 0}};
    return *out ? VQK_OK : VQK_ERR_WORKSPACE;
}

extern "C" int vqk_ctx_destroy(vqk_ctx* c) {
    if (c && vqkd::current_ctx() == c) vqkd::current_ctx() = nullptr;      // (other threads must not hold it current any more: caller's contract)
    delete c;
    return VQK_OK;
}

extern "C" int vqk_ctx_make_current(vqk_ctx* c) {
    vqkd::current_ctx() = c;                                     // NULL: back to the calling thread's default context
    return VQK_OK;
}

extern "C" int vqk_ctx_set_tile_queue(vqk_ctx* c, void* ws, int64_t ws_bytes) {
    if (ws && ws_bytes < 64) return VQK_ERR_WORKSPACE;           // eight per-XCD counters + the census word
    if (ws && (reinterpret_cast<uintptr_t>(ws) & 15)) return VQK_ERR_ALIGN;
    vqkd::DetState& d = c ? c->tq : vqkd::ctx().tq;
    d.on = ws ? 1 : 0;
    d.ws = reinterpret_cast<float*>(ws);
    d.bytes = ws ? ws_bytes : 0;
    return VQK_OK;
}

extern "C" int vqk_ctx_set_scratch(vqk_ctx* c, void* ws, int64_t ws_bytes) {
    if (ws && ws_bytes < 0) return VQK_ERR_ARG;
    if (ws && (reinterpret_cast<uintptr_t>(ws) & 15)) return VQK_ERR_ALIGN;       // the split slices are written with 16-byte stores
    vqkd::DetState& d = c ? c->scratch : vqkd::ctx().scratch;
    d.on = ws ? 1 : 0;
    d.ws = reinterpret_cast<float*>(ws);
    d.bytes = ws ? ws_bytes : 0;
    return VQK_OK;
}

extern "C" int vqk_ctx_set_deterministic(vqk_ctx* c, int on, void* ws, int64_t ws_bytes) {
    if (on && ws && ws_bytes < 0) return VQK_ERR_ARG;
    if (on && ws && (reinterpret_cast<uintptr_t>(ws) & 15)) return VQK_ERR_ALIGN;
    vqkd::DetState& d = c ? c->det : vqkd::ctx().det;
    d.on = on ? 1 : 0;
    d.ws = (on && ws) ? reinterpret_cast<float*>(ws) : nullptr;
    d.bytes = (on && ws) ? ws_bytes : 0;
    return VQK_OK;
}

// the setters of rounds 1-5: the calling thread's CURRENT context
extern "C" int vqk_set_tile_queue(void* ws, int64_t ws_bytes) { return vqk_ctx_set_tile_queue(nullptr, ws, ws_bytes); }
extern "C" int vqk_set_scratch(void* ws, int64_t ws_bytes) { return vqk_ctx_set_scratch(nullptr, ws, ws_bytes); }
extern "C" int vqk_set_deterministic(int on, void* ws, int64_t ws_bytes) { return vqk_ctx_set_deterministic(nullptr, on, ws, ws_bytes); }

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
    {"X3_WGRAD_FOLD", 0, 0},
    {"X3_WGRAD_PHASE", 0, 0},
    {"X3_WGRAD_PHASE_CAP_PCT", 0, 0},
    {"X3_WGRAD_PHASE_COEF_E4", 0, 0}
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
