// Weight gradients of the implicit-GEMM convolutions for gfx950 (split out of conv.hip in round 6): the general one-tap-per-block
// kernels (fp32 exact / bf16, any stride, 1x1, thin channel counts), the double-buffered 1x1 form, the single-role all-taps 3x3
// kernels (8x8 / 8x16 patches) and the host side that picks among them and the role-split kernel (conv_wgmx.hip), the split-product
// kernel (conv_x3.hip) and the edge-conv kernels (conv_thin_f32.hip, conv_edge.hip).
//   dW[co][tap][ci] = sum_pix dy[pix][co] * x[pix (+) tap][ci]: the contraction runs over pixels, so both operands are pixel-major in
//   LDS and the bf16 fragments come from ds_read_b64_tr_b16 (hardware transpose read); fp32 fragments are plain ds_read_b32.
//   Split-K over pixel ranges, partials combined with fp32 atomics (ordered workspace sums in deterministic mode).
// Replaces autograd's weight gradient of the F.conv2d calls of vqvae/modules/autoencoder.py:57-60, :102-105, :114, :132, :153, :170 and
// of the discriminator's conv2d_resample.py:107-122.
#include "conv_geom.h"
#include <math.h>

namespace {

using vqkd::ConvGeom;
using vqkd::xcd_remap;

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const VQK_GLB void*)src, (VQK_LDS void*)lds_wave_base, 16, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// wgrad.  grid = (co tiles * ci tiles, taps, splits).  Tile 128 co x 128 ci, K-step KP pixels.
// ------------------------------------------------------------------------------------------------
template <typename T> struct WgradFrag;

// bf16: LDS rows are pixels, 256 B (128 channels) each, 16-B chunk index XORed with (row&3)<<2.
template <> struct WgradFrag<bf16_raw> {
    static constexpr int KP = 64;           // pixels per K-step
    static constexpr int ROWB = 256;        // bytes per LDS row
    // one 32x32x16 step: 16 pixels starting at row k0; operand columns [cbase, cbase+32)
    __device__ static __forceinline__ bf16x8_t frag(const char* tile, int k0, int cbase, int lane) {
        const int i = lane & 15, grp = (lane >> 4) & 1, kgrp = lane >> 5;
        const int col = cbase + 16 * grp + 4 * (i & 3);
        const int row = k0 + 8 * kgrp + (i >> 2);
        const int lc = col >> 3;
        const char* p0 = tile + row * ROWB + ((lc ^ ((row & 3) << 2)) << 4) + ((col & 7) << 1);
        const char* p1 = p0 + 4 * ROWB;     // rows +4: same (row&3) -> same swizzle
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VQK_LDS s16x4*)p0);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VQK_LDS s16x4*)p1);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8_t, v);
    }
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            float* __restrict__ dw, const char* __restrict__ zeros,
                                                            ConvGeom g, int pix_per_split, int64_t split_stride);

template <>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel<bf16_raw>(const bf16_raw* __restrict__ x,
                                                                      const bf16_raw* __restrict__ dy,
                                                                      float* __restrict__ dw,
                                                                      const char* __restrict__ zeros, ConvGeom g,
                                                                      int pix_per_split, int64_t split_stride) {
    typedef WgradFrag<bf16_raw> F;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_a = smem;                       // dy tile  [64 pix][128 co]
    char* lds_b = smem + F::KP * F::ROWB;     // x tile   [64 pix][128 ci]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_ci = (g.cin + 127) >> 7;
    const int tco = blockIdx.x / tiles_ci, tci = blockIdx.x - tco * tiles_ci;
    const int co0 = tco * 128, ci0 = tci * 128;
    const int tap = blockIdx.y, kh = tap / g.ks, kw = tap - kh * g.ks;
    const int p_begin = blockIdx.z * pix_per_split;
    const int p_end = min(g.m, p_begin + pix_per_split);

    // load slots: one wave instruction = 4 rows x 16 chunks; 16 instructions per tile, 4 per wave
    const int pc = lane & 15, rsub = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int hw = g.h * g.w;
    for (int p0 = p_begin; p0 < p_end; p0 += F::KP) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 4 * (4 * wave + t) + rsub;          // (row & 3) == rsub
            const int lc = pc ^ (rsub << 2);
            const int p = p0 + row;
            const bool pv = p < p_end;
            // A: dy[p][co0 + lc*8 ..]
            const bool oka = pv && (co0 + lc * 8) < g.cout;
            const void* sa = oka ? (const void*)(dy + (int64_t)p * g.cout + co0 + lc * 8) : (const void*)zeros;
            glds16(sa, lds_a + (4 * wave + t) * 1024);
            // B: x[p (+) tap][ci0 + lc*8 ..]
            bool okb = pv && (ci0 + lc * 8) < g.cin;
            const void* sb = zeros;
            if (okb) {
                const int img = p / hw, rem = p - img * hw;
                const int oh = rem / g.w, ow = rem - oh * g.w;
                const int ih = oh * g.stride + kh - g.pad, iw = ow * g.stride + kw - g.pad;
                if (ih >= 0 && ih < g.vh && iw >= 0 && iw < g.vw && !(g.zs && ((ih | iw) & 1)))
                    sb = x + (((int64_t)img * g.h_in + (ih >> g.ups)) * g.w_in + (iw >> g.ups)) * g.cin + ci0 + lc * 8;
            }
            glds16(sb, lds_b + (4 * wave + t) * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int k0 = 0; k0 < F::KP; k0 += 16) {
            bf16x8_t a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = F::frag(lds_a, k0, wm * 64 + i * 32, lane);
                b[i] = F::frag(lds_b, k0, wn * 64 + i * 32, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const int taps = g.ks * g.ks;
    dw += (int64_t)blockIdx.z * split_stride;                    // deterministic mode: every split has its own copy of dW in the workspace
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = ci0 + wn * 64 + j * 32 + (lane & 31);
        if (ci >= g.cin) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < g.cout) atomicAdd(dw + ((int64_t)co * taps + tap) * g.cin + ci, acc[i][j][r] * g.acc_scale);
            }
    }
}

// double-buffered form of conv_wgrad_kernel<bf16_raw> for the 1x1 convs (one tap: few tiles, long pixel ranges per block)
__global__ __launch_bounds__(256, 2) void conv_wgrad_db_kernel(const bf16_raw* __restrict__ x,
                                                                      const bf16_raw* __restrict__ dy,
                                                                      float* __restrict__ dw,
                                                                      const char* __restrict__ zeros, ConvGeom g,
                                                                      int pix_per_split, int64_t split_stride) {
    typedef WgradFrag<bf16_raw> F;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // TWO stages of (dy tile [64 pix][128 co], x tile [64 pix][128 ci]): the K-step p0 + KP is in flight (LDS-DMA) while step p0
    // runs on the matrix pipe -- the loop used to load, wait, compute, which left the 1x1 / strided weight gradients
    // latency-bound (167 us for 128->256 @128^2, bs = 32: 2.4 TB/s)
    constexpr int TILE = F::KP * F::ROWB, STAGE = 2 * TILE;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_ci = (g.cin + 127) >> 7;
    const int tco = blockIdx.x / tiles_ci, tci = blockIdx.x - tco * tiles_ci;
    const int co0 = tco * 128, ci0 = tci * 128;
    const int tap = blockIdx.y, kh = tap / g.ks, kw = tap - kh * g.ks;
    const int p_begin = blockIdx.z * pix_per_split;
    const int p_end = min(g.m, p_begin + pix_per_split);

    // load slots: one wave instruction = 4 rows x 16 chunks; 16 instructions per tile, 4 per wave
    const int pc = lane & 15, rsub = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int hw = g.h * g.w;
    auto issue = [&](int p0, char* st) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 4 * (4 * wave + t) + rsub;          // (row & 3) == rsub
            const int lc = pc ^ (rsub << 2);
            const int p = p0 + row;
            const bool pv = p < p_end;
            // A: dy[p][co0 + lc*8 ..]
            const bool oka = pv && (co0 + lc * 8) < g.cout;
            const void* sa = oka ? (const void*)(dy + (int64_t)p * g.cout + co0 + lc * 8) : (const void*)zeros;
            glds16(sa, st + (4 * wave + t) * 1024);
            // B: x[p (+) tap][ci0 + lc*8 ..]
            bool okb = pv && (ci0 + lc * 8) < g.cin;
            const void* sb = zeros;
            if (okb) {
                const int img = p / hw, rem = p - img * hw;
                const int oh = rem / g.w, ow = rem - oh * g.w;
                const int ih = oh * g.stride + kh - g.pad, iw = ow * g.stride + kw - g.pad;
                if (ih >= 0 && ih < g.vh && iw >= 0 && iw < g.vw && !(g.zs && ((ih | iw) & 1)))
                    sb = x + (((int64_t)img * g.h_in + (ih >> g.ups)) * g.w_in + (iw >> g.ups)) * g.cin + ci0 + lc * 8;
            }
            glds16(sb, st + TILE + (4 * wave + t) * 1024);
        }
    };
    if (p_begin < p_end) issue(p_begin, smem);
    int buf = 0;
    for (int p0 = p_begin; p0 < p_end; p0 += F::KP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                        // step p0 has landed; the other stage's readers are done
        const char* lds_a = smem + buf * STAGE;
        const char* lds_b = lds_a + TILE;
        if (p0 + F::KP < p_end) issue(p0 + F::KP, smem + (buf ^ 1) * STAGE);
#pragma unroll
        for (int k0 = 0; k0 < F::KP; k0 += 16) {
            bf16x8_t a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = F::frag(lds_a, k0, wm * 64 + i * 32, lane);
                b[i] = F::frag(lds_b, k0, wn * 64 + i * 32, lane);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        buf ^= 1;
    }
    const int taps = g.ks * g.ks;
    dw += (int64_t)blockIdx.z * split_stride;                    // deterministic mode: every split has its own copy of dW in the workspace
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = ci0 + wn * 64 + j * 32 + (lane & 31);
        if (ci >= g.cin) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < g.cout) atomicAdd(dw + ((int64_t)co * taps + tap) * g.cin + ci, acc[i][j][r] * g.acc_scale);
            }
    }
}

// fp32: LDS rows are pixels, 512 B (128 fp32 channels) each, no swizzle; fragments are ds_read_b32.
template <>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel<float>(const float* __restrict__ x,
                                                                   const float* __restrict__ dy,
                                                                   float* __restrict__ dw,
                                                                   const char* __restrict__ zeros, ConvGeom g,
                                                                   int pix_per_split, int64_t split_stride) {
    constexpr int KP = 32, ROWB = 512;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* lds_a = smem;
    char* lds_b = smem + KP * ROWB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_ci = (g.cin + 127) >> 7;
    const int tco = blockIdx.x / tiles_ci, tci = blockIdx.x - tco * tiles_ci;
    const int co0 = tco * 128, ci0 = tci * 128;
    const int tap = blockIdx.y, kh = tap / g.ks, kw = tap - kh * g.ks;
    const int p_begin = blockIdx.z * pix_per_split;
    const int p_end = min(g.m, p_begin + pix_per_split);

    const int pc = lane & 31, rsub = lane >> 5;     // one wave instruction = 2 rows x 32 chunks
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int hw = g.h * g.w;
    for (int p0 = p_begin; p0 < p_end; p0 += KP) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = 2 * (4 * wave + t) + rsub;
            const int p = p0 + row;
            const bool pv = p < p_end;
            const bool oka = pv && (co0 + pc * 4) < g.cout;
            const void* sa = oka ? (const void*)(dy + (int64_t)p * g.cout + co0 + pc * 4) : (const void*)zeros;
            glds16(sa, lds_a + (4 * wave + t) * 1024);
            bool okb = pv && (ci0 + pc * 4) < g.cin;
            const void* sb = zeros;
            if (okb) {
                const int img = p / hw, rem = p - img * hw;
                const int oh = rem / g.w, ow = rem - oh * g.w;
                const int ih = oh * g.stride + kh - g.pad, iw = ow * g.stride + kw - g.pad;
                if (ih >= 0 && ih < g.vh && iw >= 0 && iw < g.vw && !(g.zs && ((ih | iw) & 1)))
                    sb = x + (((int64_t)img * g.h_in + (ih >> g.ups)) * g.w_in + (iw >> g.ups)) * g.cin + ci0 + pc * 4;
            }
            glds16(sb, lds_b + (4 * wave + t) * 1024);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int kslot = lane >> 5, c = lane & 31;
#pragma unroll 4
        for (int k0 = 0; k0 < KP; k0 += 2) {
            float a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const float*>(lds_a + (k0 + kslot) * ROWB + (wm * 64 + i * 32 + c) * 4);
                b[i] = *reinterpret_cast<const float*>(lds_b + (k0 + kslot) * ROWB + (wn * 64 + i * 32 + c) * 4);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    const int taps = g.ks * g.ks;
    dw += (int64_t)blockIdx.z * split_stride;                    // deterministic mode: every split has its own copy of dW in the workspace
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = ci0 + wn * 64 + j * 32 + (lane & 31);
        if (ci >= g.cin) continue;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (co < g.cout) atomicAdd(dw + ((int64_t)co * taps + tap) * g.cin + ci, acc[i][j][r] * g.acc_scale);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// wgrad, 3x3, bf16, "all taps per block": the block owns a 64 co x 64 ci x 9 taps slice of dW and walks a
// range of 8x8 pixel patches.  Per patch it stages dy [64 px][64 co] and the x HALO [10x10 px][64 ci] once;
// the nine taps read the halo at shifted rows (ds_read_b64_tr_b16 fragments), so each input byte enters LDS
// once instead of nine times.  Wave (i, j) owns the 32 co x 32 ci tile of all 9 taps (144 accumulators).
//
// LDS layout: every tile is split into two 32-channel HALF tiles of [rows][64 B].  One global_load_lds piece
// (1 KiB) is 16 rows of one half tile, and one transpose read of a wave (4 rows x 64 B per 32 lanes) is 256
// contiguous bytes => conflict-free with no swizzle, and every fragment address is `lane base + immediate`
// once the (pixel group, tap) loops are unrolled (the XOR-swizzled version spent ~5 VALU per MFMA on addresses).
// Stages are double-buffered (2 x 22 KiB).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8_t tr_frag2(const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VQK_LDS s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VQK_LDS s16x4*)(p + 256));      // rows +4
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

// PW16: 8x16 pixel patches (dy 128 px, x halo 10x18) instead of 8x8 -- 72 MFMAs per wave between two block barriers
// instead of 36, 1.41 instead of 1.56 halo pixels per output pixel; 2 x 40 KiB stages per block = exactly two blocks per CU.
template <bool PW16>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_halo_kernel(const bf16_raw* __restrict__ x,
                                                                    const bf16_raw* __restrict__ dy,
                                                                    float* __restrict__ dw,
                                                                    const char* __restrict__ zeros, ConvGeom g,
                                                                    int patches_per_split) {
    constexpr int PWD = PW16 ? 16 : 8, PIX = 8 * PWD, HWD = PWD + 2, HROWS = 10 * HWD;   // patch width, pixels, halo
    constexpr int X_ROWS = (HROWS + 15) / 16 * 16;                            // 112 / 192
    constexpr int DY_HALF = PIX * 64, X_HALF = X_ROWS * 64;                   // bytes per half tile
    constexpr int STAGE = 2 * DY_HALF + 2 * X_HALF;                           // 22528 / 40960
    constexpr int DYP = PIX / 16, XP = X_ROWS / 16;                           // 1 KiB pieces per half tile
    constexpr int PIECES = 2 * DYP + 2 * XP, NSLOT = (PIECES + 3) / 4, DYSLOTS = 2 * DYP / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_ci = (g.cin + 63) >> 6;
    // the (co, ci) tiles of ONE pixel range sit on one XCD (consecutive virtual ids): every dy / x byte is needed by
    // tiles_ci / tiles_co blocks, and only the first of them should have to go to HBM for it
    const int vb = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int bx = vb % (int)gridDim.x, by = vb / (int)gridDim.x;
    const int tco = bx / tiles_ci, tci = bx - tco * tiles_ci;
    const int co0 = tco * 64, ci0 = tci * 64;
    const int pw = g.w / PWD, ph = g.h >> 3;
    const int total_patches = g.n * ph * pw;
    const int p_begin = by * patches_per_split;
    const int p_end = min(total_patches, p_begin + patches_per_split);
    if (p_begin >= p_end) return;

    const int wi = wave >> 1, wj = wave & 1;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // ---- per-lane load slots (fixed over the patch loop): piece q = wave + 4*s
    //   q < 2*DYP : dy, half = q / DYP, rows 16*(q % DYP) .. +15 ; else x halo, half = (q-2*DYP)/XP, rows 16*((q-2*DYP)%XP) ..
    int s_dy[NSLOT], s_dx[NSLOT], s_choff[NSLOT];
    unsigned s_dst[NSLOT];
    bool s_ok[NSLOT];
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
        const int q = wave + 4 * sl;
        const bool isdy = q < 2 * DYP;
        const int half = isdy ? q / DYP : (q - 2 * DYP) / XP;
        const int prow = isdy ? q % DYP : (q - 2 * DYP) % XP;
        const int row = prow * 16 + (lane >> 2);
        s_choff[sl] = half * 32 + (lane & 3) * 8;
        if (isdy) {
            s_dy[sl] = row / PWD; s_dx[sl] = row % PWD;
            s_ok[sl] = (co0 + s_choff[sl]) < g.cout;
            s_dst[sl] = (unsigned)(half * DY_HALF + prow * 1024);
        } else {
            const int hy = row / HWD, hx = row - hy * HWD;
            s_dy[sl] = hy - 1; s_dx[sl] = hx - 1;
            s_ok[sl] = q < PIECES && row < HROWS && (ci0 + s_choff[sl]) < g.cin;
            s_dst[sl] = (unsigned)(2 * DY_HALF + half * X_HALF + prow * 1024);
        }
    }
    // Per-slot source pointers for patch (0, 0) of image 0: an interior patch (its halo inside the image, no upsample) is
    // then `base + one scalar offset` per piece.  The piece issue rate -- not HBM, LDS reads or load latency -- bounds this
    // kernel (timing-only builds at 128->128 @256^2: no loads 739 -> 458 us; no x-fragment LDS reads, a third LDS stage
    // with counted vmcnt, or VGPR-staged loads instead of LDS-DMA: no gain / slower), so the per-piece address arithmetic
    // and bounds tests are worth removing: -4...5 % on the large maps.
    const char* s_base[NSLOT];
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
        if (sl < DYSLOTS) s_base[sl] = reinterpret_cast<const char*>(dy + ((int64_t)s_dy[sl] * g.w + s_dx[sl]) * g.cout + co0 + s_choff[sl]);
        else s_base[sl] = reinterpret_cast<const char*>(x + ((int64_t)s_dy[sl] * g.w_in + s_dx[sl]) * g.cin + ci0 + s_choff[sl]);
    }
    auto issue = [&](int patch, char* st) {
        const int img = patch / (ph * pw), rem = patch - img * (ph * pw);
        const int pyi = rem / pw, pxi = rem - pyi * pw;
        const int py0 = pyi * 8, px0 = pxi * PWD;
        const bool interior = !g.ups && py0 >= 1 && py0 + 8 < g.h && px0 >= 1 && px0 + PWD < g.w;
        if (interior) {
            const int64_t pix = ((int64_t)img * g.h + py0) * g.w + px0;
            const int64_t off_dy = pix * g.cout * 2, off_x = pix * g.cin * 2;       // bytes (bf16)
#pragma unroll
            for (int sl = 0; sl < NSLOT; ++sl) {
                if (wave + 4 * sl >= PIECES) continue;
                const void* src = s_ok[sl] ? (const void*)(s_base[sl] + (sl < DYSLOTS ? off_dy : off_x)) : (const void*)zeros;
                glds16(src, st + s_dst[sl]);
            }
            return;
        }
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            if (wave + 4 * sl >= PIECES) continue;
            const int iy = py0 + s_dy[sl], ix = px0 + s_dx[sl];
            const void* src = zeros;
            if (sl < DYSLOTS) {                                // (the first 2*DYP pieces = DYSLOTS slots are the dy pieces)
                if (s_ok[sl]) src = dy + (((int64_t)img * g.h + iy) * g.w + ix) * g.cout + co0 + s_choff[sl];
            } else if (s_ok[sl] && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w) {
                src = x + (((int64_t)img * g.h_in + (iy >> g.ups)) * g.w_in + (ix >> g.ups)) * g.cin + ci0 + s_choff[sl];
            }
            glds16(src, st + s_dst[sl]);
        }
    };

    // ---- fragment lane bases: half tile of this wave + (i>>2)*64 + 32*grp + 8*(i&3) + k-group rows
    const int li = lane & 15, grp = (lane >> 4) & 1, kgrp = lane >> 5;
    const unsigned frag_lane = (unsigned)((li >> 2) * 64 + 32 * grp + 8 * (li & 3));
    const unsigned a_lane = (unsigned)(wi * DY_HALF) + frag_lane + (unsigned)(kgrp * 8 * 64);
    // k-group 1 = pixels 8..15 of the MFMA's 16: the next patch row (8x8 patches: +HWD halo rows) or the same row's
    // second half (8x16 patches: +8 halo rows)
    const unsigned b_lane = (unsigned)(2 * DY_HALF + wj * X_HALF) + frag_lane + (unsigned)(kgrp * (PW16 ? 8 : HWD) * 64);

    issue(p_begin, smem);
    for (int pch = p_begin; pch < p_end; ++pch) {
        const unsigned cur = (unsigned)(((pch - p_begin) & 1) * STAGE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // stage `cur` landed; everyone left the other stage
        if (pch + 1 < p_end) issue(pch + 1, smem + (STAGE - cur));
        const char* pa = smem + cur + a_lane;
        const char* pb = smem + cur + b_lane;
        if constexpr (PW16) {
            // 8x16 patches: the x fragment of (patch row gk, tap row ty, tap column tx) depends on gk + ty only, so a
            // rolling window of three halo rows (9 fragments in registers) serves all nine taps and every patch row
            // reads THREE new fragments instead of nine (2.5x fewer LDS reads; -0.7 % time: LDS reads do not bound it)
            bf16x8_t bwin[3][3];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) bwin[r][tx] = tr_frag2(pb + (r * HWD + tx) * 64);
#pragma unroll
            for (int gk = 0; gk < 8; ++gk) {
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) bwin[(gk + 2) % 3][tx] = tr_frag2(pb + ((gk + 2) * HWD + tx) * 64);
                const bf16x8_t a = tr_frag2(pa + gk * 16 * 64);
#pragma unroll
                for (int t = 0; t < 9; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bwin[(gk + t / 3) % 3][t % 3], acc[t], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int gk = 0; gk < PIX / 16; ++gk) {            // 16 pixels = patch rows 2gk, 2gk+1
                const bf16x8_t a = tr_frag2(pa + gk * 16 * 64);
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const bf16x8_t b = tr_frag2(pb + ((2 * gk + t / 3) * HWD + (t % 3)) * 64);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
                }
            }
        }
    }
    const int ci = ci0 + wj * 32 + (lane & 31);
    if (ci < g.cin) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
                if (co < g.cout) atomicAdd(dw + ((int64_t)co * 9 + t) * g.cin + ci, acc[t][r] * g.acc_scale);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// The 8x16-patch wgrad for whole 64-channel tiles (Cin % 64 == 0, Cout % 64 == 0: every 3x3 conv of the model) with
// lean piece addressing: the per-slot source state is ONE 32-bit lane offset per piece against a wave-uniform patch
// base (`global_load_lds v_off, s[base]`), so an interior patch issues its ten pieces with no address arithmetic, no
// bounds tests and no zero-page select; boundary patches recompute their coordinates from the lane id.  -3 % over the
// step's launches against conv3x3_wgrad_halo_kernel<true> (-6 % at 128->128 @256^2).  Measured and NOT kept: the ten
// pieces issued two per patch row BETWEEN the MFMA groups instead of ahead of them (+7...9 % on the 128^2 / 256^2
// maps: an LDS-DMA piece issued among ds_reads costs more than one issued in a batch); three specialised copies of
// the stage (last / interior / boundary patch) made hipcc duplicate the 144 accumulators across the merge and spill.
// Same LDS image, fragment reads and MFMA order as conv3x3_wgrad_halo_kernel<true> => bit-identical partial sums.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_p16_kernel(const bf16_raw* __restrict__ x,
                                                                   const bf16_raw* __restrict__ dy,
                                                                   float* __restrict__ dw,
                                                                   const char* __restrict__ zeros, ConvGeom g,
                                                                   int patches_per_split) {
    constexpr int PWD = 16, PIX = 128, HWD = 18, HROWS = 180, X_ROWS = 192;
    constexpr int DY_HALF = PIX * 64, X_HALF = X_ROWS * 64, STAGE = 2 * DY_HALF + 2 * X_HALF;     // 40960
    constexpr int NDY = 4, NX = 6;                               // pieces per wave and stage: 16 dy + 24 x over 4 waves
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_ci = g.cin >> 6;
    const int vb = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int bx = vb % (int)gridDim.x, by = vb / (int)gridDim.x;
    const int tco = bx / tiles_ci, tci = bx - tco * tiles_ci;
    const int co0 = tco * 64, ci0 = tci * 64;
    const int pw = g.w >> 4, ph = g.h >> 3;
    const int total_patches = g.n * ph * pw;
    const int p_begin = by * patches_per_split;
    const int p_end = min(total_patches, p_begin + patches_per_split);
    if (p_begin >= p_end) return;

    const int wi = wave >> 1, wj = wave & 1;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // piece q = wave + 4*sl.  q < 16: dy, half = q >> 3, patch row = q & 7, lane>>2 = patch column, (lane&3)*8 channels;
    // q >= 16: x halo, r = q - 16, half = r / 12, halo rows 16*(r % 12) + (lane >> 2) (rows 180..191 are padding that no
    // fragment reads: those lanes fetch row 179 again).
    const int lrow = lane >> 2, lch = (lane & 3) * 8;
    unsigned dyoff[NDY], xoff[NX];                               // byte offsets against the patch bases (interior patches)
#pragma unroll
    for (int sl = 0; sl < NDY; ++sl) {
        const int q = wave + 4 * sl, half = q >> 3, prow = q & 7;
        dyoff[sl] = (unsigned)((((prow * g.w + lrow) * g.cout) + co0 + half * 32 + lch) * 2);
    }
#pragma unroll
    for (int sl = 0; sl < NX; ++sl) {
        const int r = wave + 4 * sl, half = r / 12, row = min((r % 12) * 16 + lrow, HROWS - 1);
        const int hy = row / HWD, hx = row - hy * HWD;
        xoff[sl] = (unsigned)((((hy * g.w_in + hx) * g.cin) + ci0 + half * 32 + lch) * 2);
    }
    struct PatchPos { int img, py0, px0; bool interior; const char* bdy; const char* bx; };
    auto decode = [&](int patch) -> PatchPos {
        PatchPos pp;
        pp.img = patch / (ph * pw);
        const int rem = patch - pp.img * (ph * pw);
        const int pyi = rem / pw, pxi = rem - pyi * pw;
        pp.py0 = pyi * 8; pp.px0 = pxi * PWD;
        pp.interior = !g.ups && pp.py0 >= 1 && pp.py0 + 8 < g.h && pp.px0 >= 1 && pp.px0 + PWD < g.w;
        const int64_t pix = ((int64_t)pp.img * g.h + pp.py0) * g.w + pp.px0;
        pp.bdy = reinterpret_cast<const char*>(dy + pix * g.cout);
        pp.bx = reinterpret_cast<const char*>(x + (pix - g.w - 1) * g.cin);         // halo origin (py0 - 1, px0 - 1)
        return pp;
    };
    // one piece of the next stage; `sl` is a compile-time constant at every call site
    auto piece = [&](bool interior, int sl, const PatchPos& pp, char* st) {
        if (sl < NDY) {
            const int q = wave + 4 * sl;
            char* dst = st + (q >> 3) * DY_HALF + (q & 7) * 1024;
            if (interior || true) {                              // dy pixels of a patch are always inside the image
                glds16(pp.bdy + dyoff[sl], dst);
            }
        } else {
            const int r = wave + 4 * (sl - NDY), half = r / 12, pr = r % 12;
            char* dst = st + 2 * DY_HALF + half * X_HALF + pr * 1024;
            if (interior) {
                glds16(pp.bx + xoff[sl - NDY], dst);
            } else {
                const int row = min(pr * 16 + lrow, HROWS - 1);
                const int hy = row / HWD, hx = row - hy * HWD;
                const int iy = pp.py0 + hy - 1, ix = pp.px0 + hx - 1;
                const void* src = zeros;
                if (iy >= 0 && iy < g.h && ix >= 0 && ix < g.w)
                    src = x + (((int64_t)pp.img * g.h_in + (iy >> g.ups)) * g.w_in + (ix >> g.ups)) * g.cin + ci0 + half * 32 + lch;
                glds16(src, dst);
            }
        }
    };

    const int li = lane & 15, grp = (lane >> 4) & 1, kgrp = lane >> 5;
    const unsigned frag_lane = (unsigned)((li >> 2) * 64 + 32 * grp + 8 * (li & 3));
    const unsigned a_lane = (unsigned)(wi * DY_HALF) + frag_lane + (unsigned)(kgrp * 8 * 64);
    const unsigned b_lane = (unsigned)(2 * DY_HALF + wj * X_HALF) + frag_lane + (unsigned)(kgrp * 8 * 64);

    {
        const PatchPos p0 = decode(p_begin);
#pragma unroll
        for (int sl = 0; sl < NDY + NX; ++sl) piece(p0.interior, sl, p0, smem);
    }
    for (int pch = p_begin; pch < p_end; ++pch) {
        const unsigned cur = (unsigned)(((pch - p_begin) & 1) * STAGE);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // stage `cur` landed; everyone left the other stage
        const bool has_next = pch + 1 < p_end;
        const PatchPos pp = decode(has_next ? pch + 1 : pch);
        char* nst = smem + (STAGE - cur);
        const char* pa = smem + cur + a_lane;
        const char* pb = smem + cur + b_lane;
        if (has_next) {
            if (pp.interior) {
#pragma unroll
                for (int sl = 0; sl < NDY + NX; ++sl) piece(true, sl, pp, nst);
            } else {
#pragma unroll
                for (int sl = 0; sl < NDY + NX; ++sl) piece(false, sl, pp, nst);
            }
        }
        bf16x8_t bwin[3][3];
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) bwin[r][tx] = tr_frag2(pb + (r * HWD + tx) * 64);
#pragma unroll
        for (int gk = 0; gk < 8; ++gk) {
#pragma unroll
            for (int tx = 0; tx < 3; ++tx) bwin[(gk + 2) % 3][tx] = tr_frag2(pb + ((gk + 2) * HWD + tx) * 64);
            const bf16x8_t a = tr_frag2(pa + gk * 16 * 64);
#pragma unroll
            for (int t = 0; t < 9; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bwin[(gk + t / 3) % 3][t % 3], acc[t], 0, 0, 0);
        }
    }
    const int ci = ci0 + wj * 32 + (lane & 31);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
            atomicAdd(dw + ((int64_t)co * 9 + t) * g.cin + ci, acc[t][r] * g.acc_scale);
        }
}

// deterministic split-K of the general weight-gradient kernels: dw[i] += sum over the splits' private copies, in split order
__global__ __launch_bounds__(256) void wgrad_split_reduce_kernel(const float* __restrict__ part, int64_t elems, int splits,
                                                                 float* __restrict__ dw) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += part[(int64_t)k * elems + i];
    dw[i] += s;
}

// per-thread launch state shared with conv.hip (test hook, deterministic mode, grid cap beside a concurrent kernel)
#define g_force_variant (vqkd::conv_force_variant())
#define g_wgrad_blocks (vqkd::conv_wgrad_blocks())
#define g_det (vqkd::det_state().on)
#define g_det_ws (vqkd::det_state().ws)
#define g_det_ws_bytes (vqkd::det_state().bytes)
inline int make_geom(ConvGeom& g, int dtype, int n, int h_in, int w_in, int cin, int cout, int ksize, int ups) {
    return vqkd::conv_make_geom(g, dtype, n, h_in, w_in, cin, cout, ksize, ups);
}

}  // namespace

extern "C" {

static int wgrad_general(int dtype, const void* x, const void* dy, float* dw, int n, int h_in, int w_in, int cin, int cout,
                         int ksize, int stride, int pad, int mode, int h_out, int w_out, const void* zeros, void* stream,
                         int dy_pool = 0, float dy_scale = 1.0f, int fold = 0) {
    VQK_REQUIRE(x && dy && dw && zeros, VQK_ERR_ARG);
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(dy) && vqk_aligned16(zeros), VQK_ERR_ALIGN);
    VQK_REQUIRE(mode >= 0 && mode <= 1 && (stride == 1 || stride == 2) && pad >= 0, VQK_ERR_ARG);
    ConvGeom g;
    const int rc = make_geom(g, dtype, n, h_in, w_in, cin, cout, ksize, mode);
    if (rc) return rc;
    const int epc = dtype == VQK_F32 ? 4 : 8;
    VQK_REQUIRE(cout % epc == 0, VQK_ERR_SHAPE);
    g.acc_scale = dy_scale;                                      // dW += dy_scale * (x^T dy): every kernel below scales its partial sums
    const bool plain = stride == 1 && pad == (ksize >> 1) && h_out == g.h && w_out == g.w;
    if (!plain) {
        g.stride = stride; g.pad = pad;
        VQK_REQUIRE(h_out > 0 && w_out > 0, VQK_ERR_SHAPE);
        g.h = h_out; g.w = w_out;
        const int64_t m = (int64_t)n * h_out * w_out;
        VQK_REQUIRE(m < 0x7fffff00, VQK_ERR_SHAPE);
        g.m = (int)m;
    }
    if (fold) {
        // (hi | lo) pair operands of the split-product mode: only the matrix/auxiliary-wave kernel has the folded tile classes
        VQK_REQUIRE(plain && dtype == VQK_BF16 && ksize == 3 && (g.h % 8) == 0 && (g.w % 16) == 0 && (cin % 128) == 0 && (cout % 128) == 0
                    && !dy_pool && !g_det && g_force_variant != 0 && VQK_TUNE("WGMX", 1) && VQK_TUNE("WGRAD_BLOCKS", 0) == 0
                    && VQK_TUNE("WGRAD_NO_PW16", 0) == 0, VQK_ERR_SHAPE);
    }
    if (plain && dtype == VQK_BF16 && ksize == 3 && (g.h % 8) == 0 && (g.w % 8) == 0 && g_force_variant != 0) {
        const int tiles = fold ? 3 * (cout / 128) * (cin / 128) : ((cout + 63) / 64) * ((cin + 63) / 64);
        const bool no_pw16 = VQK_TUNE("WGRAD_NO_PW16", 0) != 0;
        const bool pw16 = (g.w % 16) == 0 && !no_pw16;
        const int total_patches = g.n * (g.h / 8) * (g.w / (pw16 ? 16 : 8));
        // split-K over pixel patches.  Cost model fitted on MI355X (tools/convbench.py sweeps): MFMA time falls with the
        // number of resident blocks (up to 2 per CU) while every split adds one fp32 atomic pass over dW (~1.1 TB/s):
        // t(s) = F / (R * min(1, tiles*s/512)) + s * |dW| / B  =>  s* = sqrt(0.16 * pixels / tiles) below the block cap.
        const int target = VQK_TUNE("WGRAD_BLOCKS", 0);
        const int cap = g_wgrad_blocks > 0 ? g_wgrad_blocks : 512;
        int splits;
        if (target > 0) splits = (target + tiles - 1) / tiles;
        else {
            splits = (int)(sqrt(0.16 * (double)g.m / tiles) + 0.5);
            if (splits > (cap + tiles - 1) / tiles) splits = (cap + tiles - 1) / tiles;
        }
        const int minp = pw16 ? 2 : 4;                                              // >= 256 pixels per block
        if (splits > (total_patches + minp - 1) / minp) splits = (total_patches + minp - 1) / minp;
        if (splits < 1 || g_det) splits = 1;                     // deterministic: one block per dW tile, no cross-block sums
        const int pps = (total_patches + splits - 1) / splits;
        splits = (total_patches + pps - 1) / pps;
        const dim3 grid((unsigned)tiles, (unsigned)splits);
        const bool no_p16k = VQK_TUNE("WGRAD_NO_P16K", 0) != 0;
        const int wgmx = VQK_TUNE("WGMX", 1);
        if (pw16 && (cin % 64) == 0 && (cout % 64) == 0 && wgmx && target == 0) {
            // matrix/auxiliary-wave form: ONE 512-thread block per CU, so half as many resident blocks as the cost model
            // above assumes: s* = sqrt(0.08 * pixels / tiles) under half the block cap
            const double coef = VQK_TUNE("WGMX_COEF_E4", 800) * 1e-4;          // (knob in units of 1e-4)
            const int comm = VQK_TUNE("COMM_CUS", 0);               // CUs left to a running collective (conv_mx.hip)
            const int nph = dy_pool >= 6 ? 4 : 1;                   // all four phases of an upsample conv in one launch: 4 x the blocks
            const int capm = (cap / 2 - comm) / nph > tiles ? (cap / 2 - comm) / nph : tiles;
            int sm = (int)(sqrt(coef * (double)g.m / tiles) + 0.5);
            if (sm > (capm + tiles - 1) / tiles) sm = (capm + tiles - 1) / tiles;
            if (sm > (total_patches + 3) / 4) sm = (total_patches + 3) / 4;          // >= 4 patches per block
            float* part = nullptr;
            if (g_det) {                                         // partial tiles in the workspace, summed in split order
                const int64_t per_split = (int64_t)tiles * 64 * 9 * 64 * 4;
                const int fit = (int)(g_det_ws_bytes / per_split);
                if (sm > fit) sm = fit;
                part = sm >= 2 ? g_det_ws : nullptr;             // a single split owns every element: its atomics are plain adds
            }
            if (sm < 1) sm = 1;
            const int ppm = (total_patches + sm - 1) / sm;
            sm = (total_patches + ppm - 1) / ppm;
            if (sm < 2) part = nullptr;
            ConvGeom gm = g;
            gm.dy_pool = dy_pool; gm.acc_scale = dy_scale; gm.fold = fold;
            return vqkd::launch_conv3x3_wgrad_mx(x, dy, dw, zeros, gm, tiles, sm * nph, ppm, vqk_stream(stream), part);
        }
        if (dy_pool || fold) return VQK_ERR_SHAPE;              // half-resolution dy / pair operands exist on the matrix/auxiliary-wave kernel only
        if (pw16 && (cin % 64) == 0 && (cout % 64) == 0 && !no_p16k) {
            static const hipError_t attr = hipFuncSetAttribute((const void*)conv3x3_wgrad_p16_kernel,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 40960);
            (void)attr;
            hipLaunchKernelGGL(conv3x3_wgrad_p16_kernel, grid, dim3(256), 2 * 40960, vqk_stream(stream),
                               (const bf16_raw*)x, (const bf16_raw*)dy, dw, (const char*)zeros, g, pps);
        } else if (pw16) {
            static const hipError_t attr = hipFuncSetAttribute((const void*)conv3x3_wgrad_halo_kernel<true>,
                                                               hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 40960);
            (void)attr;
            hipLaunchKernelGGL(conv3x3_wgrad_halo_kernel<true>, grid, dim3(256), 2 * 40960, vqk_stream(stream),
                               (const bf16_raw*)x, (const bf16_raw*)dy, dw, (const char*)zeros, g, pps);
        } else {
            hipLaunchKernelGGL(conv3x3_wgrad_halo_kernel<false>, grid, dim3(256), 2 * 22528, vqk_stream(stream),
                               (const bf16_raw*)x, (const bf16_raw*)dy, dw, (const char*)zeros, g, pps);
        }
        VQK_CHECK_LAUNCH();
        return VQK_OK;
    }
    if (plain && dtype == VQK_F32 && ksize == 3 && mode == 0 && !g_det && !dy_pool && g_force_variant != 0 && (g.w % 4) == 0) {
        // the edge convs' weight gradients in the fp32 modes (conv_thin_f32.hip)
        if (cin == 4 && (cout == 64 || cout == 128 || cout == 256))
            return vqkd::launch_conv3x3_wgrad_thin_f32(0, (const float*)dy, (const float*)x, dw, n, g.h, g.w, cout, dy_scale, vqk_stream(stream));
        if (cout == 4 && (cin == 64 || cin == 128 || cin == 256))
            return vqkd::launch_conv3x3_wgrad_thin_f32(1, (const float*)x, (const float*)dy, dw, n, g.h, g.w, cin, dy_scale, vqk_stream(stream));
    }
    const int kp = dtype == VQK_F32 ? 32 : 64;
    const int tiles = ((cout + 127) / 128) * ((cin + 127) / 128) * ksize * ksize;
    // split the pixel range so that ~2048 blocks are in flight, each with >= 4 K-steps (fp32); the double-buffered bf16
    // kernel keeps two 64-KiB blocks per CU busy with ~512 blocks and a quarter of the atomic passes over dW (each block
    // ends with one: at 2048 blocks the 1x1 shortcut's 128 x 256 gradient cost 134 MB of atomics per launch)
    // (measured, tools/wgrad_gen_bench.py: 1x1 128->256 @128^2 167 -> 67 us; the strided 3x3 gathers are faster on the
    // single-stage kernel with ~2048 blocks -- 421 vs 302 us at 128->256 @257^2 stride 2 -- and keep it)
    const int wg_target = VQK_TUNE("WGRAD_GEN_BLOCKS", 512);
    const bool db = dtype == VQK_BF16 && ksize == 1;
    const int target_blocks = db ? wg_target : 2048;
    int splits = (target_blocks + tiles - 1) / tiles;
    const int max_splits = (g.m + 4 * kp - 1) / (4 * kp);
    if (splits > max_splits) splits = max_splits;
    const int64_t dw_elems = (int64_t)cout * ksize * ksize * cin;
    if (g_det) {                                                 // deterministic: private copies of dW per split + ordered reduce
        const int64_t fit = g_det_ws ? g_det_ws_bytes / (dw_elems * 4) : 0;
        if (splits > fit) splits = (int)fit;
    }
    if (splits < 1) splits = 1;
    int pps = (g.m + splits - 1) / splits;
    pps = ((pps + kp - 1) / kp) * kp;
    splits = (g.m + pps - 1) / pps;
    const dim3 grid((unsigned)(((cout + 127) / 128) * ((cin + 127) / 128)), (unsigned)(ksize * ksize), (unsigned)splits);
    float* dst = dw;
    int64_t sstride = 0;
    const bool det_split = g_det && splits > 1;
    if (det_split) {
        dst = g_det_ws; sstride = dw_elems;
        if (hipMemsetAsync(g_det_ws, 0, (size_t)splits * dw_elems * 4, vqk_stream(stream)) != hipSuccess) return VQK_ERR_LAUNCH;
    }
    if (dtype == VQK_F32)
        hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), 32768, vqk_stream(stream), (const float*)x, (const float*)dy, dst, (const char*)zeros, g, pps, sstride);
    else if (db) {
        static const hipError_t attr = hipFuncSetAttribute((const void*)conv_wgrad_db_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        if (attr != hipSuccess) return VQK_ERR_LAUNCH;
        hipLaunchKernelGGL(conv_wgrad_db_kernel, grid, dim3(256), 65536, vqk_stream(stream), (const bf16_raw*)x, (const bf16_raw*)dy, dst, (const char*)zeros, g, pps, sstride);
    } else {
        hipLaunchKernelGGL(conv_wgrad_kernel<bf16_raw>, grid, dim3(256), 32768, vqk_stream(stream), (const bf16_raw*)x, (const bf16_raw*)dy, dst, (const char*)zeros, g, pps, sstride);
    }
    if (det_split)
        hipLaunchKernelGGL(wgrad_split_reduce_kernel, dim3((unsigned)((dw_elems + 255) / 256)), dim3(256), 0, vqk_stream(stream),
                           (const float*)g_det_ws, dw_elems, splits, dw);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_conv2d_wgrad(int dtype, const void* x, const void* dy, float* dw, int n, int h_in, int w_in, int cin, int cout,
                     int ksize, int ups, const void* zeros, void* stream) {
    VQK_REQUIRE(ups == 0 || ups == 1, VQK_ERR_ARG);
    return wgrad_general(dtype, x, dy, dw, n, h_in, w_in, cin, cout, ksize, 1, ksize >> 1, ups, h_in << ups, w_in << ups,
                         zeros, stream);
}

int vqk_conv2d_wgrad_x3_f32(const float* x, const float* dy, float* dw, int n, int h_in, int w_in, int cin, int cout, int ups,
                            float scale, void* stream) {
    // split-product weight gradient straight from the fp32 tensors (csrc/conv_x3.hip: conv3x3_wgrad_x3_kernel)
    VQK_REQUIRE(x && dy && dw, VQK_ERR_ARG);
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(dy), VQK_ERR_ALIGN);
    VQK_REQUIRE(ups == 0 || ups == 1 || ups == 2, VQK_ERR_ARG);
    VQK_REQUIRE(!g_det && g_force_variant != 0, VQK_ERR_SHAPE);      // atomics only: deterministic mode keeps the exact-fp32 kernel
    ConvGeom g;
    if (ups == 2 || (ups == 1 && VQK_TUNE("X3_WGRAD_PHASE", 1) && (h_in % 8) == 0 && (w_in % 8) == 0)) {
        // the 2x2-resampling convs in phase form (4/9 of the MFMAs): patches over the LOW-resolution grid.  ups = 1: x [n][h_in][w_in] is
        // that grid, dy [n][2 h_in][2 w_in]; ups = 2: dy = the pooled gradient [n][h_in / 2][w_in / 2] of a conv over x [n][h_in][w_in]
        VQK_REQUIRE(ups == 1 || ((h_in % 16) == 0 && (w_in % 16) == 0), VQK_ERR_SHAPE);
        const int hl = ups == 2 ? h_in / 2 : h_in, wl = ups == 2 ? w_in / 2 : w_in;
        const int rcp = make_geom(g, VQK_F32, n, hl, wl, cin, cout, 3, 0);
        if (rcp) return rcp;
        g.ntap = 4; g.phase_mode = ups == 2 ? 2 : 1;
        g.acc_scale = scale;
        return vqkd::launch_conv3x3_wgrad_x3(x, dy, dw, g, g_wgrad_blocks, vqk_stream(stream));
    }
    const int rc = make_geom(g, VQK_F32, n, h_in, w_in, cin, cout, 3, ups);
    if (rc) return rc;
    g.acc_scale = scale;
    return vqkd::launch_conv3x3_wgrad_x3(x, dy, dw, g, g_wgrad_blocks, vqk_stream(stream));
}

int vqk_conv2d_wgrad_x3(const void* x_pair, const void* dy_pair, float* dw, int n, int h_in, int w_in, int cin, int cout, int ups,
                        float scale, const void* zeros, void* stream) {
    // x_pair [n, h_in, w_in, 2 cin], dy_pair [n, h, w, 2 cout] bf16 (vqk_split_pair_f32); dw fp32 [cout][3][3][cin] +=
    VQK_REQUIRE(ups == 0 || ups == 1, VQK_ERR_ARG);
    VQK_REQUIRE((cin % 64) == 0 && (cout % 64) == 0, VQK_ERR_SHAPE);
    return wgrad_general(VQK_BF16, x_pair, dy_pair, dw, n, h_in, w_in, 2 * cin, 2 * cout, 3, 1, 1, ups, h_in << ups, w_in << ups,
                         zeros, stream, 0, scale, 1);
}

int vqk_conv2d_wgrad_pooled_dy(int dtype, const void* x, const void* dy_pooled, float* dw, int n, int h, int w, int cin,
                               int cout, float scale, const void* zeros, void* stream) {
    VQK_REQUIRE(dtype == VQK_BF16 && (h % 8) == 0 && (w % 16) == 0 && (cin % 64) == 0 && (cout % 64) == 0, VQK_ERR_SHAPE);
    const int wgmx = VQK_TUNE("WGMX", 1);
    VQK_REQUIRE(wgmx && g_force_variant != 0 && VQK_TUNE("WGRAD_BLOCKS", 0) == 0 && VQK_TUNE("WGRAD_NO_PW16", 0) == 0, VQK_ERR_SHAPE);
    return wgrad_general(dtype, x, dy_pooled, dw, n, h, w, cin, cout, 3, 1, 1, 0, h, w, zeros, stream, 1, scale);
}

int vqk_conv2d_wgrad_ups_phase(int dtype, const void* x, const void* dy, float* dw, int n, int h, int w, int cin, int cout,
                                float scale, const void* zeros, void* stream) {
    VQK_REQUIRE(dtype == VQK_BF16 && (h % 8) == 0 && (w % 16) == 0 && (cin % 64) == 0 && (cout % 64) == 0, VQK_ERR_SHAPE);
    const int wgmx = VQK_TUNE("WGMX", 1);
    VQK_REQUIRE(wgmx && g_force_variant != 0 && VQK_TUNE("WGRAD_BLOCKS", 0) == 0 && VQK_TUNE("WGRAD_NO_PW16", 0) == 0 && !g_det,
                VQK_ERR_SHAPE);
    if (VQK_TUNE("UPS_MERGE", 1))                                // the four output phases as ONE launch (phase = a dimension of the grid)
        return wgrad_general(dtype, x, dy, dw, n, h, w, cin, cout, 3, 1, 1, 0, h, w, zeros, stream, 6, scale);
    for (int ph = 0; ph < 4; ++ph) {                             // one launch per output phase (a, b) = (ph >> 1, ph & 1)
        const int rc = wgrad_general(dtype, x, dy, dw, n, h, w, cin, cout, 3, 1, 1, 0, h, w, zeros, stream, 2 + ph, scale);
        if (rc) return rc;
    }
    return VQK_OK;
}

int vqk_conv2d_wgrad_pooled_dy_phase(int dtype, const void* x, const void* dy_pooled, float* dw, int n, int h, int w, int cin,
                                     int cout, float scale, const void* zeros, void* stream) {
    // x [n, 2h, 2w, cin] (the conv's input), dy_pooled [n, h, w, cout]; dw[Cout][3][3][Cin] += scale * wgrad(x, unpool(dy_pooled)).
    // The phase-form kernel with the operands' roles swapped (conv_wgmx.hip, dy_pool = 7): its "x" operand is dy_pooled (cout
    // channels), its phase-gathered "dy" operand is x (cin channels); h, w: the POOLED grid.
    VQK_REQUIRE(dtype == VQK_BF16 && (h % 8) == 0 && (w % 16) == 0 && (cin % 64) == 0 && (cout % 64) == 0, VQK_ERR_SHAPE);
    const int wgmx = VQK_TUNE("WGMX", 1);
    VQK_REQUIRE(wgmx && g_force_variant != 0 && VQK_TUNE("WGRAD_BLOCKS", 0) == 0 && VQK_TUNE("WGRAD_NO_PW16", 0) == 0 && !g_det &&
                VQK_TUNE("UPS_MERGE", 1) != 0, VQK_ERR_SHAPE);
    return wgrad_general(dtype, dy_pooled, x, dw, n, h, w, cout, cin, 3, 1, 1, 0, h, w, zeros, stream, 7, scale);
}

int vqk_conv2d_wgrad_general(int dtype, const void* x, const void* dy, float* dw, int n, int h_in, int w_in, int cin,
                             int cout, int ksize, int stride, int pad, int mode, int h_out, int w_out, const void* zeros,
                             void* stream) {
    return wgrad_general(dtype, x, dy, dw, n, h_in, w_in, cin, cout, ksize, stride, pad, mode, h_out, w_out, zeros, stream);
}

int vqk_conv2d_wgrad_general_scaled(int dtype, const void* x, const void* dy, float* dw, int n, int h_in, int w_in, int cin,
                                    int cout, int ksize, int stride, int pad, int mode, int h_out, int w_out, float scale,
                                    const void* zeros, void* stream) {
    return wgrad_general(dtype, x, dy, dw, n, h_in, w_in, cin, cout, ksize, stride, pad, mode, h_out, w_out, zeros, stream, 0, scale);
}

}  // extern "C"
