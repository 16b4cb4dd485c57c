// ------------------------------------------------------------------------------------------------
// Edge convolutions of the autoencoder: the 3x3 convs with ONE 16-byte chunk of channels on one side -- the encoder's
// first conv (3 -> C, vqvae/modules/autoencoder.py:114) and the decoder's last conv (C -> 3, :170), whose 3 channels are
// padded to 8 bf16.  Their weight gradients are GEMMs with N = 9 taps x 8 channels = 72 columns:
//
//   conv3x3_wgrad_thin_kernel<MODE>:  out[wc][tap][t] = sum_pix WIDE[pix][wc] * THIN[pix + s * off(tap)][t]
//     MODE 0 (thin = x, wide = dy: the first conv):  dW[co = wc][tap][ci = t],  s = +1
//     MODE 1 (thin = dy, wide = x: the last conv):   dW[co = t][tap][ci = wc],  s = -1  (dW[co][tap][ci] = sum_p dy[p][co] x[p + off][ci],
//                                                    re-indexed by the x pixel q = p + off)
//   HBM-bound: the wide tensor is read ONCE (256 B per pixel), the thin one stays in L2.  The all-taps 64x64-tile kernel
//   these layers ran on before padded the 8 thin channels to a 64-wide tile (8x the multiply-adds: 250 us per launch at
//   256^2, bs = 32, against ~95 us of memory time).
//
//   block = 256 threads, one per CU, persistent over 128-pixel patches, THREE LDS stages, loads two patches ahead:
//     wide tile  [128 px][128 ch]  (32 KiB, 16-byte chunk index XOR (row & 3) << 2: conflict-free transposing reads)
//     im2col tile [128 px][9 taps][8 ch] (18 KiB): one 16-byte LDS-DMA per (pixel, tap), borders from a zero page
//   both by global_load_lds; per patch and wave 8 k-steps x 3 MFMAs (32 wide channels x 96 columns, 72 used) on
//   transposing fragment reads (ds_read_b64_tr_b16).  Split-K over the blocks through a caller WORKSPACE (one
//   128 x 72 partial per block, plain stores) + an ordered reduce pass: no atomics, run-to-run deterministic.
// ------------------------------------------------------------------------------------------------
#include "conv_geom.h"

namespace {

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const VQK_GLB void*)src, (VQK_LDS void*)lds_wave_base, 16, 0, 0);
}

typedef __attribute__((ext_vector_type(8))) short s16x8;

// 32 columns x 16 rows (k) of a row-major [rows][cols] bf16 LDS tile as an MFMA operand: lane (col = lane & 31, k group
// = lane >> 5) receives rows k0 + 8*kgrp .. +7 of its column.  `pitch`: bytes per row; `swz`: the wide tile's chunk swizzle.
template <int PITCH, bool SWZ>
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int k0, int cbase, int lane) {
    const int i = lane & 15, grp = (lane >> 4) & 1, kgrp = lane >> 5;
    const int col = cbase + 16 * grp + 4 * (i & 3);
    const int row = k0 + 8 * kgrp + (i >> 2);
    const char* p0;
    if (SWZ) p0 = tile + row * PITCH + (((col >> 3) ^ ((row & 3) << 2)) << 4) + ((col & 7) << 1);
    else p0 = tile + row * PITCH + col * 2;
    const char* p1 = p0 + 4 * PITCH;                             // rows +4: same (row & 3), same swizzle
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VQK_LDS s16x4*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VQK_LDS s16x4*)p1);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

#ifndef VQK_EDGE_ABL
#define VQK_EDGE_ABL 0       // timing-only ablation bits (tools/ab_build.sh): 1 im2col pieces from the zero page, 2 no wide pieces, 4 no MFMA loop
#endif
constexpr int EDGE_CW = 128;                                     // wide channels
constexpr int EDGE_PIX = 128;                                    // pixels per patch
constexpr int EDGE_WIDE_B = EDGE_PIX * EDGE_CW * 2;              // 32768
constexpr int EDGE_IM_PITCH = 144;                               // 9 taps x 16 B
constexpr int EDGE_IM_B = EDGE_PIX * EDGE_IM_PITCH;              // 18432
constexpr int EDGE_STAGE = EDGE_WIDE_B + EDGE_IM_B;              // 51200
constexpr int EDGE_LDS = 3 * EDGE_STAGE + 4096;                  // + 1 KiB of scratch per wave (dummy pieces)
constexpr int EDGE_OUT = EDGE_CW * 72;                           // 9216 partial sums per block

template <int MODE>
__global__ __launch_bounds__(256, 1) void conv3x3_wgrad_thin_kernel(const bf16_raw* __restrict__ wide,
                                                                    const bf16_raw* __restrict__ thin,
                                                                    float* __restrict__ ws, const char* __restrict__ zeros,
                                                                    int n, int h, int w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = (int)(((int64_t)n * h * w) / EDGE_PIX);
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int cnt = (total - b + G - 1) / G;                     // patches b, b + G, ...
    const int hw = h * w;
    constexpr int SGN = MODE == 0 ? 1 : -1;

    // slot = (pixel of the patch, tap) is a property of the lane: decoded ONCE (the per-patch address arithmetic used to
    // redo three integer divisions per slot -- 2400 of the 3400 cycles a patch took with all loads stubbed out).  A patch is
    // 128 consecutive pixels that never straddle an image (h*w % 128 == 0) and lie in ONE row (w % 128 == 0) or cover whole
    // rows (128 % w == 0): pixel = (y0 + ry, x0 + rx) with (ry, rx) fixed per slot.
    int s_ry[5], s_rx[5], s_ty[5], s_tx[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const int q = wave + 4 * t;
        const int slot = q * 64 + lane;
        const int pl = slot / 9, tap = slot - pl * 9;
        s_ry[t] = w >= EDGE_PIX ? 0 : pl / w;
        s_rx[t] = w >= EDGE_PIX ? pl : pl - (pl / w) * w;
        s_ty[t] = SGN * (tap / 3 - 1);
        s_tx[t] = SGN * (tap - (tap / 3) * 3 - 1);
    }
    auto issue = [&](int patch, int stage) {
        char* st = smem + stage * EDGE_STAGE;
        const int64_t p0 = (int64_t)patch * EDGE_PIX;
        const int img = (int)(p0 / hw), rem = (int)(p0 - (int64_t)img * hw);          // wave-uniform
        const int y0 = rem / w, x0 = rem - y0 * w;
#pragma unroll
        for (int t = 0; t < 8; ++t) {                            // wide: piece q covers rows 4q .. 4q+3
            const int q = wave * 8 + t;
            const int row = 4 * q + (lane >> 4), pc = lane & 15;
            const int lc = pc ^ ((row & 3) << 2);
            if (!((VQK_EDGE_ABL & 2) && n > 0)) glds16(wide + (p0 + row) * EDGE_CW + lc * 8, st + q * 1024);
            else glds16(zeros, st + q * 1024);
        }
        const bf16_raw* timg = thin + (int64_t)img * hw * 8;
#pragma unroll
        for (int t = 0; t < 5; ++t) {                            // im2col: slot = pixel * 9 + tap, 64 slots per piece
            const int q = wave + 4 * t;
            if (q < 18) {
                const int yy = y0 + s_ry[t] + s_ty[t], xx = x0 + s_rx[t] + s_tx[t];
                const void* src = zeros;
                if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w && !((VQK_EDGE_ABL & 1) && n > 0))
                    src = timg + ((int64_t)yy * w + xx) * 8;
                glds16(src, st + EDGE_WIDE_B + q * 1024);
            } else {
                glds16(zeros, smem + 3 * EDGE_STAGE + wave * 1024);      // keeps every wave at 13 operations per patch
            }
        }
    };

    f32x16 acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;

    if (cnt > 0) issue(b, 0);
    if (cnt > 1) issue(b + G, 1);
    for (int i = 0; i < cnt; ++i) {
        // in-order completion: everything but the newest patch's 13 operations of this wave has landed
        if (i + 1 < cnt) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                         // patch i is in LDS; patch i-1's stage is free again
        if (i + 2 < cnt) issue(b + (i + 2) * G, (i + 2) % 3);
        const char* st = smem + (i % 3) * EDGE_STAGE;
#pragma unroll
        for (int k0 = 0; k0 < (((VQK_EDGE_ABL & 4) && n > 0) ? 0 : EDGE_PIX); k0 += 16) {
            const bf16x8_t a = tr_frag<EDGE_CW * 2, true>(st, k0, 32 * wave, lane);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const bf16x8_t bj = tr_frag<EDGE_IM_PITCH, false>(st + EDGE_WIDE_B, k0, 32 * j, lane);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bj, acc[j], 0, 0, 0);
            }
        }
    }
    // partial of this block: acc[j][r] = out[wc = 32*wave + (r&3) + 8*(r>>2) + 4*(lane>>5)][col = 32*j + (lane&31)]
    float* part = ws + (int64_t)b * EDGE_OUT;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int col = 32 * j + (lane & 31);
        if (col >= 72) continue;
        const int tap = col >> 3, t = col & 7;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int wc = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int idx = MODE == 0 ? (wc * 9 + tap) * 8 + t : (t * 9 + tap) * EDGE_CW + wc;
            part[idx] = acc[j][r];
        }
    }
}

// dw[i] += sum over the blocks' partials in a FIXED order (deterministic): block = 16 outputs x 16 partial groups, every
// thread sums its group's <= 16 partials (all loads independent: one round trip, not 64 dependent ones), the groups are
// combined through LDS in group order
// thin_true < 8 (the TRUE channel count of the thin side -- 3 for the image / reconstruction): dw is the parameter's own,
// unpadded gradient ([128][9][thin_true] when thin_in, [thin_true][9][128] otherwise); the padded channels' sums are dropped
__global__ __launch_bounds__(256) void wgrad_thin_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int blocks,
                                                                int thin_true, int thin_in) {
    __shared__ float part[16][17];
    const int o = threadIdx.x & 15, kg = threadIdx.x >> 4;
    const int i = (int)blockIdx.x * 16 + o;
    float v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = kg * 16 + j;
        v[j] = k < blocks ? ws[(int64_t)k * EDGE_OUT + i] : 0.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += v[j];
    part[kg][o] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        float t = 0.0f;
#pragma unroll
        for (int g2 = 0; g2 < 16; ++g2) t += part[g2][threadIdx.x];
        const int i2 = (int)blockIdx.x * 16 + (int)threadIdx.x;
        if (thin_true >= 8) dw[i2] += t;
        else if (thin_in) { const int ci = i2 & 7; if (ci < thin_true) dw[(i2 >> 3) * thin_true + ci] += t; }     // i2 = (co * 9 + tap) * 8 + ci
        else if (i2 < thin_true * 9 * EDGE_CW) dw[i2] += t;                                                      // i2 = (co * 9 + tap) * 128 + ci
    }
}


// ------------------------------------------------------------------------------------------------
// conv3x3_thin_out_kernel: the decoder's last conv (autoencoder.py:170: 128 -> 3 channels, padded to 8, + bias + tanh).
// A GEMM with N = 8 output columns: on the 32-wide tiles of the stream kernel three quarters of the MFMA work multiplied
// padding (262 us at 256^2, bs = 32, against ~95 us of memory time for the 128-channel input).  Here the 16x16x32 MFMA carries
// the weights as the A operand (rows = output channels, 8 of 16 used) and SIXTEEN pixels as the B operand:
//   block = 256 threads, persistent over 8x32-pixel tiles x two 64-channel halves ("units"); the 10x34 halo of a unit is
//   staged in LDS by global_load_lds (144-byte pixel pitch = 8 data slots + 1 pad slot fetched from the zero page:
//   conflict-free ds_read_b128 of 16 pixels x 4 k-slices), double-buffered; all nine taps read it at shifted addresses;
//   the 36 weight fragments of both halves stay in registers; wave w owns rows 2w, 2w+1 of the tile (4 groups of 16 pixels).
// ------------------------------------------------------------------------------------------------
constexpr int TO_TH = 8, TO_TW = 32, TO_HW2 = TO_TW + 2, TO_HPIX = (TO_TH + 2) * TO_HW2;      // 340 halo pixels
constexpr int TO_PITCH = 144, TO_PIECES = 48, TO_STAGE = TO_PIECES * 1024, TO_LDS = 2 * TO_STAGE;
typedef __attribute__((ext_vector_type(4))) unsigned int eu32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int eu32x2;

__global__ __launch_bounds__(256, 1) void conv3x3_thin_out_kernel(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ wgt,
                                                                  const float* __restrict__ bias, bf16_raw* __restrict__ y,
                                                                  const char* __restrict__ zeros, int n, int h, int w, int act) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, kq = lane >> 4;
    const int tiles_x = w / TO_TW, tiles_y = h / TO_TH;
    const int total = n * tiles_y * tiles_x;
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    const int cnt = (total - b + G - 1) / G;
    if (cnt <= 0) return;

    // weights: A fragment (half, tap, ks): row co = lane & 15 (rows 8..15 are zero), columns half*64 + ks*32 + kq*8 .. + 7
    bf16x8_t wf[2][9][2];
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                eu32x4 v = {0u, 0u, 0u, 0u};
                if (px < 8) v = *reinterpret_cast<const eu32x4*>(wgt + ((int64_t)px * 9 + tap) * 128 + hf * 64 + ks * 32 + kq * 8);
                wf[hf][tap][ks] = __builtin_bit_cast(bf16x8_t, v);
            }
    float bv[4] = {0.f, 0.f, 0.f, 0.f};
    if (bias && kq < 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) bv[i] = bias[4 * kq + i];
    }

    // this lane's twelve halo slots per unit: slot -> (halo pixel, 16-byte chunk; chunk 8 = pad)
    int rel[12], flg[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
        const int slot = (wave * 12 + t) * 64 + lane;
        const int hp = slot / 9, pc = slot - hp * 9;
        const int hy = hp / TO_HW2, hx = hp - hy * TO_HW2;
        rel[t] = (((hy - 1) * w + (hx - 1)) * 128 + pc * 8) * 2;
        flg[t] = (hy == 0 ? 1 : 0) | (hy == TO_TH + 1 ? 2 : 0) | (hx == 0 ? 4 : 0) | (hx == TO_TW + 1 ? 8 : 0) |
                 ((pc == 8 || hp >= TO_HPIX) ? 16 : 0);
    }
    struct Tile { int img, y0, x0; };
    auto tile_of = [&](int i) -> Tile {
        int t = b + i * G;
        Tile tl;
        const int txi = t % tiles_x; t /= tiles_x;
        const int tyi = t % tiles_y;
        tl.img = t / tiles_y; tl.y0 = tyi * TO_TH; tl.x0 = txi * TO_TW;
        return tl;
    };
    auto issue = [&](const Tile& tl, int hf, int stage) {
        const char* base = reinterpret_cast<const char*>(x) + ((((int64_t)tl.img * h + tl.y0) * w + tl.x0) * 128 + hf * 64) * 2;
        const int tb = (tl.y0 == 0 ? 1 : 0) | (tl.y0 + TO_TH == h ? 2 : 0) | (tl.x0 == 0 ? 4 : 0) | (tl.x0 + TO_TW == w ? 8 : 0) | 16;
        char* st = smem + stage * TO_STAGE + wave * 12 * 1024;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            const void* src = (flg[t] & tb) ? (const void*)zeros : (const void*)(base + rel[t]);
            glds16(src, st + t * 1024);
        }
    };

    f32x4 acc[4];
    Tile cur = tile_of(0);
    issue(cur, 0, 0);
    int stage = 0;
    for (int i = 0; i < cnt; ++i) {
        const Tile nxt = tile_of(i + 1 < cnt ? i + 1 : i);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                     // this unit has landed; the other stage's readers are done
            if (hf == 0) issue(cur, 1, stage ^ 1);
            else if (i + 1 < cnt) issue(nxt, 0, stage ^ 1);
            const char* st = smem + stage * TO_STAGE;
            if (hf == 0) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) acc[gq] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ty = tap / 3, tx = tap - ty * 3;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {             // group gq: tile row 2*wave + (gq >> 1), columns 16*(gq & 1) ..
                        const int hp = (2 * wave + (gq >> 1) + ty) * TO_HW2 + 16 * (gq & 1) + px + tx;
                        const bf16x8_t bf = *reinterpret_cast<const bf16x8_t*>(st + hp * TO_PITCH + (ks * 4 + kq) * 16);
                        acc[gq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[hf][tap][ks], bf, acc[gq], 0, 0, 0);
                    }
            }
            stage ^= 1;
        }
        // epilogue: lane (pixel px of its groups, output channels 4 kq .. + 3); only kq < 2 holds real rows
        if (kq < 2) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[gq][e] + bv[e];
                    if (act == 1) v[e] = tanhf(v[e]);
                }
                const int yy = cur.y0 + 2 * wave + (gq >> 1), xx = cur.x0 + 16 * (gq & 1) + px;
                const eu32x2 o = {vqkd::pack_bf16x2(v[0], v[1]), vqkd::pack_bf16x2(v[2], v[3])};
                *reinterpret_cast<eu32x2*>(y + (((int64_t)cur.img * h + yy) * w + xx) * 8 + 4 * kq) = o;
            }
        }
        cur = nxt;
    }
}

}  // namespace

extern "C" {

int64_t vqk_conv2d_wgrad_edge_ws_bytes(void) { return (int64_t)256 * EDGE_OUT * 4; }

int vqk_conv2d_wgrad_edge(int dtype, const void* x, const void* dy, float* dw, void* ws, int64_t ws_bytes, int n, int h,
                          int w, int cin, int cout, const void* zeros, void* stream) {
    return vqk_conv2d_wgrad_edge_true(dtype, x, dy, dw, ws, ws_bytes, n, h, w, cin, cout, 8, zeros, stream);
}

int vqk_conv2d_wgrad_edge_true(int dtype, const void* x, const void* dy, float* dw, void* ws, int64_t ws_bytes, int n, int h,
                               int w, int cin, int cout, int thin_true, const void* zeros, void* stream) {
    VQK_REQUIRE(x && dy && dw && ws && zeros, VQK_ERR_ARG);
    VQK_REQUIRE(thin_true >= 1 && thin_true <= 8, VQK_ERR_ARG);
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(dy) && vqk_aligned16(ws) && vqk_aligned16(zeros), VQK_ERR_ALIGN);
    VQK_REQUIRE(dtype == VQK_BF16, VQK_ERR_DTYPE);
    VQK_REQUIRE(n > 0 && h > 0 && w > 0, VQK_ERR_SHAPE);
    const bool first = cin == 8 && cout == EDGE_CW, last = cout == 8 && cin == EDGE_CW;
    const int64_t m = (int64_t)n * h * w;
    // a 128-pixel patch lies inside one image, and inside one row or on whole rows (the kernel decodes a slot's pixel once)
    VQK_REQUIRE((first || last) && ((int64_t)h * w) % EDGE_PIX == 0 && (w % EDGE_PIX == 0 || EDGE_PIX % w == 0) && m < 0x7fffffffLL,
                VQK_ERR_SHAPE);
    const int total = (int)(m / EDGE_PIX);
    int blocks = total < 256 ? total : 256;
    VQK_REQUIRE(ws_bytes >= (int64_t)blocks * EDGE_OUT * 4, VQK_ERR_ARG);
    hipStream_t st = vqk_stream(stream);
    if (first) {
        static const hipError_t attr = hipFuncSetAttribute((const void*)conv3x3_wgrad_thin_kernel<0>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, EDGE_LDS);
        if (attr != hipSuccess) return VQK_ERR_LAUNCH;
        hipLaunchKernelGGL(conv3x3_wgrad_thin_kernel<0>, dim3((unsigned)blocks), dim3(256), EDGE_LDS, st, (const bf16_raw*)dy,
                           (const bf16_raw*)x, (float*)ws, (const char*)zeros, n, h, w);
    } else {
        static const hipError_t attr = hipFuncSetAttribute((const void*)conv3x3_wgrad_thin_kernel<1>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, EDGE_LDS);
        if (attr != hipSuccess) return VQK_ERR_LAUNCH;
        hipLaunchKernelGGL(conv3x3_wgrad_thin_kernel<1>, dim3((unsigned)blocks), dim3(256), EDGE_LDS, st, (const bf16_raw*)x,
                           (const bf16_raw*)dy, (float*)ws, (const char*)zeros, n, h, w);
    }
    VQK_CHECK_LAUNCH();
    static_assert(EDGE_OUT % 16 == 0, "reduce blocks own 16 outputs");
    hipLaunchKernelGGL(wgrad_thin_reduce_kernel, dim3(EDGE_OUT / 16), dim3(256), 0, st, (const float*)ws, dw, blocks, thin_true, first ? 1 : 0);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_conv2d_thin_out(int dtype, const void* x, const void* w, const float* bias, void* y, int n, int h, int wd, int cin,
                        int cout, int act, const void* zeros, void* stream) {
    VQK_REQUIRE(x && w && y && zeros, VQK_ERR_ARG);
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(w) && vqk_aligned16(y) && vqk_aligned16(zeros), VQK_ERR_ALIGN);
    VQK_REQUIRE(dtype == VQK_BF16, VQK_ERR_DTYPE);
    VQK_REQUIRE(act == 0 || act == 1, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && cin == 128 && cout == 8 && h > 0 && wd > 0 && (h % TO_TH) == 0 && (wd % TO_TW) == 0 &&
                (int64_t)n * h * wd * 128 * 2 < 0x7fffffff0LL, VQK_ERR_SHAPE);
    const int total = n * (h / TO_TH) * (wd / TO_TW);
    const int blocks = total < 256 ? total : 256;
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv3x3_thin_out_kernel,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, TO_LDS);
    if (attr != hipSuccess) return VQK_ERR_LAUNCH;
    hipLaunchKernelGGL(conv3x3_thin_out_kernel, dim3((unsigned)blocks), dim3(256), TO_LDS, vqk_stream(stream), (const bf16_raw*)x,
                       (const bf16_raw*)w, bias, (bf16_raw*)y, (const char*)zeros, n, h, wd, act);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

}  // extern "C"
