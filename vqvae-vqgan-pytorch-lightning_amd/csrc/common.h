// Shared device helpers for the vqk kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vqk.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef unsigned short bf16_raw;

#define VQK_LDS __attribute__((address_space(3)))
#define VQK_GLB __attribute__((address_space(1)))

__device__ __forceinline__ float bf16_to_f32(bf16_raw v) { return __uint_as_float(((unsigned)v) << 16); }
// round-to-nearest-even, NaN kept quiet
__device__ __forceinline__ bf16_raw f32_to_bf16(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_raw)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_raw)(u >> 16);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kPer16B = 4;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_raw> {
    static constexpr int kPer16B = 8;
    __device__ static __forceinline__ float ld(const bf16_raw* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_raw* p, float v) { *p = f32_to_bf16(v); }
};

// 16-byte vector of T <-> float array
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void load(const float* p, float (&o)[4]) {
        f32x4 v = *reinterpret_cast<const f32x4*>(p);
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
    __device__ static __forceinline__ void load_nt(const float* p, float (&o)[4]) {
        f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
    }
    __device__ static __forceinline__ void store(float* p, const float (&o)[4]) {
        f32x4 v = {o[0], o[1], o[2], o[3]};
        *reinterpret_cast<f32x4*>(p) = v;
    }
    __device__ static __forceinline__ void store_nt(float* p, const float (&o)[4]) {
        f32x4 v = {o[0], o[1], o[2], o[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
    }
};
// two fp32 -> packed bf16 pair in ONE instruction (v_cvt_pk_bf16_f32: hardware round-to-nearest-even, NaN stays NaN).  The
// software form above costs ~12 VALU instructions per element with an exec-masked NaN branch: it made the bf16 stores of the
// GroupNorm passes the largest single item of their instruction count (round 4).
__device__ __forceinline__ unsigned vqk_pack_bf16x2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
typedef __attribute__((ext_vector_type(4))) unsigned int vqk_u32x4;
__device__ __forceinline__ vqk_u32x4 vqk_pack_bf16x8(const float (&o)[8]) {
    const vqk_u32x4 v = {vqk_pack_bf16x2(o[0], o[1]), vqk_pack_bf16x2(o[2], o[3]), vqk_pack_bf16x2(o[4], o[5]), vqk_pack_bf16x2(o[6], o[7])};
    return v;
}

template <> struct Vec16<bf16_raw> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void unpack(const vqk_u32x4& v, float (&o)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(v[i] << 16); o[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u); }
    }
    __device__ static __forceinline__ void load(const bf16_raw* p, float (&o)[8]) {
        unpack(*reinterpret_cast<const vqk_u32x4*>(p), o);
    }
    __device__ static __forceinline__ void load_nt(const bf16_raw* p, float (&o)[8]) {
        unpack(__builtin_nontemporal_load(reinterpret_cast<const vqk_u32x4*>(p)), o);
    }
    __device__ static __forceinline__ void store(bf16_raw* p, const float (&o)[8]) {
        *reinterpret_cast<vqk_u32x4*>(p) = vqk_pack_bf16x8(o);
    }
    // streaming store (nt): the line is not kept in L2 / Infinity Cache on its way to HBM
    __device__ static __forceinline__ void store_nt(bf16_raw* p, const float (&o)[8]) {
        __builtin_nontemporal_store(vqk_pack_bf16x8(o), reinterpret_cast<vqk_u32x4*>(p));
    }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// deterministic mode (vqk_set_deterministic, api.cpp): per host thread, like the block caps -- set, launch, reset in one place
namespace vqkd {
struct DetState { int on; float* ws; int64_t bytes; };
DetState& det_state();
DetState& scratch_state();      // vqk_set_scratch: zero-initialised fp32 scratch of the current stream (split-K partial sums)
DetState& tile_queue_state();   // vqk_set_tile_queue: the current stream's tile-queue words (conv_geom.h: ConvGeom::tq)
}
// launch heuristics that tools/ sweep (include/vqk.h: vqk_set_tuning): process-wide slots, relaxed atomics; a call site
// resolves its slot once and then reads one int per launch.  The library itself never reads the environment.
namespace vqkd {
struct TuneSlot {
    const char* name; int value; int is_set;
    int get(int dflt) const { return __atomic_load_n(&is_set, __ATOMIC_RELAXED) ? __atomic_load_n(&value, __ATOMIC_RELAXED) : dflt; }
};
TuneSlot* tune_slot(const char* name);      // nullptr-safe: an unknown name aborts at first use (a typo in the source)
}
#define VQK_TUNE(name, dflt) ([]() -> const vqkd::TuneSlot* { static const vqkd::TuneSlot* s = vqkd::tune_slot(name); return s; }()->get(dflt))
#define VQK_CHECK_LAUNCH() do { if (hipGetLastError() != hipSuccess) return VQK_ERR_LAUNCH; } while (0)
#define VQK_REQUIRE(cond, code) do { if (!(cond)) return (code); } while (0)
static inline bool vqk_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline hipStream_t vqk_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int vqk_grid_1d(int64_t work_items, int per_block, int cap = 256 * 8) {
    int64_t g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}
