// ------------------------------------------------------------------------------------------------
// conv3x3_wgrad_mx_kernel: the 8x16-patch weight-gradient kernel (conv.hip: conv3x3_wgrad_p16_kernel) split into MATRIX
// waves and AUXILIARY waves, bf16, whole 64-channel tiles.  dW[co][tap][ci] += sum_pix dy[pix][co] * x[pix (+) tap][ci]
// (the weight gradient of the F.conv2d calls of vqvae/modules/autoencoder.py:57-60, :102-105, :132, :153).
//
// Why: in the single-role kernel every wave issues its share of the stage's forty 1-KiB LDS-DMA pieces in front of its 72
// MFMAs; a timing-only build without the pieces ran 739 -> 458 us at 128->128 @256^2: the piece issue, not the
// matrix pipe, bounds it.  Here
//   block = 512 threads, one block per CU, tile 64 co x 64 ci x 9 taps over a range of 8x16-pixel patches:
//     waves 0-3  "M": transposing fragment reads (ds_read_b64_tr_b16) + 72 MFMAs per patch, order pinned: the three x
//                fragments of halo row gk+3 and the dy fragment of patch row gk+1 are requested between the MFMAs of patch
//                row gk (rolling three-row window as before); 144 accumulators per lane, fp32 atomics into dW at the end.
//     waves 4-7  "X": the forty pieces of patch p+2 (dy patch + x halo, same LDS image as the single-role kernel) while
//                patch p is computed: THREE 40-KiB stages, one s_barrier per patch; the X waves wait with COUNTED vmcnt, so a patch's
//                pieces have two whole intervals to land.
// Same fragment reads and MFMA order per accumulator as conv3x3_wgrad_p16_kernel => bit-identical partial sums per block.
//
// NT = 4 (round 5): the weight gradient of a nearest-x2 UPSAMPLE conv in PHASE form.  Output pixel (2i+a, 2j+b) of the 3x3 conv
// over the upsampled image reads low-resolution pixel (i + ((a+ky-1)>>1), j + ((b+kx-1)>>1)): per output phase (a, b) only a 2x2
// window of low-resolution shifts occurs, so  dW[ky][kx] = sum over the four phases of G_ab[r(a,ky)][s(b,kx)],
// G_ab[r][s] = sum_ij dy[2i+a][2j+b] x[i+r][j+s] -- sixteen low-resolution tap GEMMs over a quarter of the pixels each instead of
// nine full-resolution ones: 4/9 of the multiply-adds, like the phase form of the forward / data gradient (conv_mx.hip).  One
// launch per phase (g.dy_pool = 2 + 2a + b): x is the LOW-resolution input with the ordinary halo, the dy patch is gathered at
// stride 2 from the full-resolution gradient by the LDS-DMA source addresses, four accumulators per wave, and the final atomic
// pass adds G[ty][tx] to every tap (ky, kx) it stands for (1, 2 or 4 of them).
// ------------------------------------------------------------------------------------------------
#include "conv_geom.h"

namespace {
using vqkd::ConvGeom;
using vqkd::xcd_remap;

#ifndef VQK_WGMX_ABL
#define VQK_WGMX_ABL 0       // timing-only ablation bits: 1 no pieces, 2 no atomic pass over dW
#endif
#ifndef VQK_WGMX_NST
#define VQK_WGMX_NST 3       // LDS stages (3: 120 KiB, 4: all 160 KiB)
#endif
#ifndef VQK_WGMX_PIN
#define VQK_WGMX_PIN 1
#endif
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

__device__ __forceinline__ void glds16(const void* src, char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const VQK_GLB void*)src, (VQK_LDS void*)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ bf16x8_t tr_frag2(const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VQK_LDS s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VQK_LDS s16x4*)(p + 256));      // rows +4
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

template <int NT>
__global__ __launch_bounds__(512, 2) void conv3x3_wgrad_mx_kernel(const bf16_raw* __restrict__ x,
                                                                  const bf16_raw* __restrict__ dy, float* __restrict__ dw,
                                                                  const char* __restrict__ zeros, ConvGeom g,
                                                                  int patches_per_split, float* __restrict__ part) {
    constexpr int PWD = 16, PIX = 128, HWD = 18, HROWS = 180, X_ROWS = 192;
    constexpr int DY_HALF = PIX * 64, X_HALF = X_ROWS * 64, STAGE = 2 * DY_HALF + 2 * X_HALF;     // 40960
    constexpr int NDY = 4, NX = 6;                               // pieces per X wave and stage: 16 dy + 24 x over 4 waves
    constexpr int NST = VQK_WGMX_NST, LEAD = NST - 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // g.fold (split-product mode, conv_x3.hip): x / dy are (hi | lo) pair tensors with pixel pitches g.cin / g.cout = twice the TRUE
    // channel counts; tile class 0 = dy_hi^T x_hi, 1 = dy_hi^T x_lo, 2 = dy_lo^T x_hi, all three added onto the same dW tile
    const int cin_t = g.fold ? g.cin >> 1 : g.cin, cout_t = g.fold ? g.cout >> 1 : g.cout;
    const int tiles_ci = cin_t >> 6;
    const int vb = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int bx0 = vb % (int)gridDim.x, by = vb / (int)gridDim.x;
    const int tiles_pair = tiles_ci * (cout_t >> 6);
    const int cls = g.fold ? bx0 / tiles_pair : 0;
    const int bx = g.fold ? bx0 - cls * tiles_pair : bx0;
    const int tco = bx / tiles_ci, tci = bx - tco * tiles_ci;
    const int co0 = tco * 64, ci0 = tci * 64;                    // dW tile (true channels)
    const int co0m = co0 + (cls == 2 ? cout_t : 0), ci0m = ci0 + (cls == 1 ? cin_t : 0);      // operand channels in memory
    const int pw = g.w >> 4, ph = g.h >> 3;
    const int total_patches = g.n * ph * pw;
    // NT = 4: g.dy_pool = 2 + phase (one phase per launch) or 6 (ALL FOUR phases in one launch: phase = by & 3, split = by >> 2 --
    // a block still owns one phase, i.e. four accumulators, but the launch fills the chip with a quarter of the splits per phase:
    // per-block prologue and atomic pass are amortised over 4x the patches of the one-phase-per-launch form)
    // g.dy_pool = 7: as 6 with the operands' ROLES SWAPPED -- the weight gradient of a conv followed by a 2x2 average pool from the
    // POOLED gradient: dW[ky][kx] = sum_YX dyp[Y>>1][X>>1] x[Y+ky-1][X+kx-1] re-indexed over the phases (a', b') of the FULL-resolution
    // x: G'[r'][s'] = sum_ij x[2i+a'][2j+b'] dyp[i+r'][j+s'] over the same 2x2 windows.  The phase-gathered operand ("dy" of this
    // kernel) is x, the windowed low-resolution operand ("x" of this kernel) is the pooled gradient; the tile comes out transposed
    // (rows = the conv's input channels) and stands for the MIRRORED taps (2 - ky, 2 - kx): the final pass scatters accordingly.
    const bool merged = NT == 4 && g.dy_pool >= 6, swapped = NT == 4 && g.dy_pool == 7;
    const int phase = NT == 4 ? (merged ? (by & 3) : g.dy_pool - 2) : 0;
    const int split = merged ? (by >> 2) : by;
    const int p_begin = split * patches_per_split;
    const int p_end = min(total_patches, p_begin + patches_per_split);
    if (p_begin >= p_end) return;
    auto patch_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    if (wave < 4) {
        // ================================================================= M waves
        const int wi = wave >> 1, wj = wave & 1;
        f32x16 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
        const int ph_a = phase >> 1, ph_b = phase & 1;           // output phase (a, b)
        const int li = lane & 15, grp = (lane >> 4) & 1, kgrp = lane >> 5;
        const unsigned frag_lane = (unsigned)((li >> 2) * 64 + 32 * grp + 8 * (li & 3));
        const unsigned a_lane = (unsigned)(wi * DY_HALF) + frag_lane + (unsigned)(kgrp * 8 * 64);
        const unsigned b_lane = (unsigned)(2 * DY_HALF + wj * X_HALF) + frag_lane + (unsigned)(kgrp * 8 * 64);
        patch_barrier();                                         // stage 0 landed
        int si = 0;
        for (int pch = p_begin; pch < p_end; ++pch) {
            const char* pa = smem + si * STAGE + a_lane;
            const char* pb = smem + si * STAGE + b_lane;
            if constexpr (NT == 4) {
                // 2x2 window of the phase: halo rows gk + a + {0, 1}, halo columns b + {0, 1}; slot gk % 2 holds halo row gk + a
                bf16x8_t bw2[2][2], a2[2];
                const char* pw = pb + (ph_a * HWD + ph_b) * 64;
                a2[0] = tr_frag2(pa);
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int tx = 0; tx < 2; ++tx) bw2[r][tx] = tr_frag2(pw + (r * HWD + tx) * 64);
#pragma unroll
                for (int gk = 0; gk < 8; ++gk) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[gk & 1], bw2[gk & 1][0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[gk & 1], bw2[gk & 1][1], acc[1], 0, 0, 0);
                    if (gk < 7) {
                        a2[(gk + 1) & 1] = tr_frag2(pa + (gk + 1) * 16 * 64);
#pragma unroll
                        for (int tx = 0; tx < 2; ++tx) bw2[gk & 1][tx] = tr_frag2(pw + ((gk + 2) * HWD + tx) * 64);
                    }
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[gk & 1], bw2[(gk + 1) & 1][0], acc[2], 0, 0, 0);
                    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[gk & 1], bw2[(gk + 1) & 1][1], acc[3], 0, 0, 0);
                }
            } else {
            bf16x8_t bwin[3][3], a[2];
            a[0] = tr_frag2(pa);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int tx = 0; tx < 3; ++tx) bwin[r][tx] = tr_frag2(pb + (r * HWD + tx) * 64);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gk = 0; gk < 8; ++gk) {
                // taps of tap row 0 first: they are the last readers of window row gk % 3, which the fragments of halo
                // row gk + 3 then overwrite while the other six MFMAs run
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gk & 1], bwin[gk % 3][t], acc[t], 0, 0, 0);
                if (gk < 7) {
                    a[(gk + 1) & 1] = tr_frag2(pa + (gk + 1) * 16 * 64);
#pragma unroll
                    for (int tx = 0; tx < 3; ++tx) bwin[gk % 3][tx] = tr_frag2(pb + ((gk + 3) * HWD + tx) * 64);
                }
#pragma unroll
                for (int t = 3; t < 9; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gk & 1], bwin[(gk + t / 3) % 3][t % 3], acc[t], 0, 0, 0);
                if (VQK_WGMX_PIN) {
                    if (gk < 7) {
                        // MFMA, next dy fragment (2 reads) between the first three MFMAs, then one x-fragment read per MFMA
                        SGB(0x008, 1); SGB(0x100, 1); SGB(0x008, 1); SGB(0x100, 1); SGB(0x008, 1);
#pragma unroll
                        for (int k = 0; k < 6; ++k) { SGB(0x100, 1); SGB(0x008, 1); }
                    } else {
                        SGB(0x008, 9);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            patch_barrier();
            si = si == NST - 1 ? 0 : si + 1;
        }
        const int ci = ci0 + wj * 32 + (lane & 31);
        if ((VQK_WGMX_ABL & 2) && g.n > 0 && acc[0][0] != 12345.678f) return;      // timing-only: no atomic pass
        if constexpr (NT == 4) {
            // G[ty][tx] of phase (a, b) stands for the taps ky in {ty ? (a ? 2 : 1) : 0 .. ty ? 2 : (a ? 1 : 0)}, kx likewise
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ty = t >> 1, tx = t & 1;
                const int ky0 = ty ? (ph_a ? 2 : 1) : 0, ky1 = ty ? 2 : (ph_a ? 1 : 0);
                const int kx0 = tx ? (ph_b ? 2 : 1) : 0, kx1 = tx ? 2 : (ph_b ? 1 : 0);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
                    const float v = acc[t][r] * g.acc_scale;
                    for (int ky = ky0; ky <= ky1; ++ky)
                        for (int kx = kx0; kx <= kx1; ++kx) {
                            if (swapped) atomicAdd(dw + ((int64_t)ci * 9 + (2 - ky) * 3 + (2 - kx)) * g.cout + co, v);
                            else atomicAdd(dw + ((int64_t)co * 9 + ky * 3 + kx) * g.cin + ci, v);
                        }
                }
            }
            return;
        } else {
        if (part) {
            // deterministic mode: this block's partial tile goes to the workspace slot (split by, tile bx) with plain stores;
            // wgrad_mx_reduce_kernel adds the splits in index order
            float* mine = part + ((int64_t)by * gridDim.x + bx) * (64 * 9 * 64);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int col = wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
                    mine[(col * 9 + t) * 64 + wj * 32 + (lane & 31)] = acc[t][r] * g.acc_scale;
                }
            return;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
                atomicAdd(dw + ((int64_t)co * 9 + t) * cin_t + ci, acc[t][r] * g.acc_scale);
            }
        return;
        }
    }

    // ===================================================================== X waves: the stage's forty LDS-DMA pieces
    // piece q = xw + 4*sl.  q < 16: dy, half = q >> 3, patch row = q & 7, lane>>2 = patch column, (lane&3)*8 channels;
    // q >= 16: x halo, r = q - 16, half = r / 12, halo rows 16*(r % 12) + (lane >> 2) (rows 180..191 are padding that no
    // fragment reads: those lanes fetch row 179 again).
    const int xw = wave - 4;
    const int lrow = lane >> 2, lch = (lane & 3) * 8;
    unsigned dyoff[NDY], xoff[NX];                               // byte offsets against the patch bases (interior patches)
#pragma unroll
    for (int sl = 0; sl < NDY; ++sl) {
        const int q = xw + 4 * sl, half = q >> 3, prow = q & 7;
        // dy_pool 1: dy at HALF resolution (every pooled pixel stands for its 2x2 block); >= 2 (NT = 4): dy at TWICE the resolution
        // of the patch grid, one phase of it gathered at stride 2
        dyoff[sl] = g.dy_pool == 1 ? (unsigned)(((((prow >> 1) * (g.w >> 1) + (lrow >> 1)) * g.cout) + co0m + half * 32 + lch) * 2)
                  : g.dy_pool >= 2 ? (unsigned)(((((2 * prow) * (2 * g.w) + 2 * lrow) * g.cout) + co0m + half * 32 + lch) * 2)
                                   : (unsigned)((((prow * g.w + lrow) * g.cout) + co0m + half * 32 + lch) * 2);
    }
#pragma unroll
    for (int sl = 0; sl < NX; ++sl) {
        const int r = xw + 4 * sl, half = r / 12, row = min((r % 12) * 16 + lrow, HROWS - 1);
        const int hy = row / HWD, hx = row - hy * HWD;
        xoff[sl] = (unsigned)((((hy * g.w_in + hx) * g.cin) + ci0m + half * 32 + lch) * 2);
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
        const int64_t dpix = g.dy_pool == 1 ? ((int64_t)pp.img * (g.h >> 1) + (pp.py0 >> 1)) * (g.w >> 1) + (pp.px0 >> 1)
                           : g.dy_pool >= 2 ? ((int64_t)pp.img * (2 * g.h) + 2 * pp.py0 + (phase >> 1)) * (2 * g.w) + 2 * pp.px0 + (phase & 1)
                                            : pix;
        pp.bdy = reinterpret_cast<const char*>(dy + dpix * g.cout);
        pp.bx = reinterpret_cast<const char*>(x + (pix - g.w - 1) * g.cin);         // halo origin (py0 - 1, px0 - 1)
        return pp;
    };
    auto piece = [&](bool interior, int sl, const PatchPos& pp, char* st) {
        if (sl < NDY) {
            const int q = xw + 4 * sl;
            glds16(pp.bdy + dyoff[sl], st + (q >> 3) * DY_HALF + (q & 7) * 1024);
        } else {
            const int r = xw + 4 * (sl - NDY), half = r / 12, pr = r % 12;
            char* dst = st + 2 * DY_HALF + half * X_HALF + pr * 1024;
            if (interior) {
                glds16(pp.bx + xoff[sl - NDY], dst);
            } else {
                const int row = min(pr * 16 + lrow, HROWS - 1);
                const int hy = row / HWD, hx = row - hy * HWD;
                const int iy = pp.py0 + hy - 1, ix = pp.px0 + hx - 1;
                const void* src = zeros;
                if (iy >= 0 && iy < g.h && ix >= 0 && ix < g.w)
                    src = x + (((int64_t)pp.img * g.h_in + (iy >> g.ups)) * g.w_in + (ix >> g.ups)) * g.cin + ci0m + half * 32 + lch;
                glds16(src, dst);
            }
        }
    };
    auto issue = [&](int patch, int stage) {
        if ((VQK_WGMX_ABL & 1) && g.n > 0) {                     // timing-only: ten harmless loads keep the wait counts valid
#pragma unroll
            for (int sl = 0; sl < NDY + NX; ++sl) glds16(zeros, smem + NST * STAGE - 1024 * 4 + xw * 1024);
            return;
        }
        const PatchPos pp = decode(patch);
        char* st = smem + stage * STAGE;
        if (pp.interior) {
#pragma unroll
            for (int sl = 0; sl < NDY + NX; ++sl) piece(true, sl, pp, st);
        } else {
#pragma unroll
            for (int sl = 0; sl < NDY + NX; ++sl) piece(false, sl, pp, st);
        }
    };
    // In-order completion makes the waits countable: every issue() is exactly NDY + NX = 10 vector-memory operations of this
    // wave, so "at most 10 k outstanding" = everything but the k newest batches has landed.  A patch's pieces are requested
    // LEAD = NST - 1 intervals before the barrier that publishes them (a vmcnt(0) at the top of each interval gave them ONE).
    auto wait_batches = [&](int k) {                             // k: batches that may still be in flight
        if (k >= 2) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (k == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    const int last = p_end - 1;
#pragma unroll
    for (int k = 0; k < LEAD; ++k)
        if (p_begin + k <= last) issue(p_begin + k, k);
    wait_batches(min(last, p_begin + LEAD - 1) - p_begin);
    patch_barrier();                                             // publishes patch p_begin
    int sn = LEAD % NST;                                         // stage of patch pch + LEAD
    for (int pch = p_begin; pch <= last; ++pch) {
        if (pch + LEAD <= last) issue(pch + LEAD, sn);
        wait_batches(pch + 1 <= last ? min(last, pch + LEAD) - (pch + 1) : 0);       // patch pch + 1 has landed
        patch_barrier();                                         // publishes patch pch + 1; patch pch's stage is free again
        sn = sn == NST - 1 ? 0 : sn + 1;
    }
}

// dw[(co0 + col) * 9 + t][ci0 + cil] += sum over the splits, in split order, of part[split][tile][(col * 9 + t) * 64 + cil]
__global__ __launch_bounds__(256) void wgrad_mx_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int tiles,
                                                              int splits, int cin) {
    const int tile = blockIdx.y, tiles_ci = cin >> 6;
    const int e = (int)(blockIdx.x * 256 + threadIdx.x);          // element of the 64 x 9 x 64 tile
    if (e >= 64 * 9 * 64) return;
    const int cil = e & 63, ct = e >> 6, col = ct / 9, t = ct - col * 9;
    const int tco = tile / tiles_ci, tci = tile - tco * tiles_ci;
    float s = 0.0f;
    for (int k = 0; k < splits; ++k) s += part[((int64_t)k * tiles + tile) * (64 * 9 * 64) + e];
    dw[((int64_t)(tco * 64 + col) * 9 + t) * cin + tci * 64 + cil] += s;
}

}  // namespace

namespace vqkd {

int launch_conv3x3_wgrad_mx(const void* x, const void* dy, float* dw, const void* zeros, const ConvGeom& g, int tiles,
                            int splits, int pps, hipStream_t st, float* part) {
    if (g.fold && (g.dy_pool || part)) return VQK_ERR_ARG;        // split-product operands: plain 3x3 form, atomics only
    if (g.dy_pool >= 2) {                                         // one output phase of an upsample conv: the 2x2-window form
        if (g.dy_pool > 7 || part || (g.dy_pool >= 6 && (splits & 3))) return VQK_ERR_ARG;
        static const hipError_t attr4 = hipFuncSetAttribute((const void*)conv3x3_wgrad_mx_kernel<4>,
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, VQK_WGMX_NST * 40960);
        if (attr4 != hipSuccess) return VQK_ERR_LAUNCH;
        hipLaunchKernelGGL(conv3x3_wgrad_mx_kernel<4>, dim3((unsigned)tiles, (unsigned)splits), dim3(512), VQK_WGMX_NST * 40960, st,
                           (const bf16_raw*)x, (const bf16_raw*)dy, dw, (const char*)zeros, g, pps, part);
        return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
    }
    static const hipError_t attr = hipFuncSetAttribute((const void*)conv3x3_wgrad_mx_kernel<9>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, VQK_WGMX_NST * 40960);
    (void)attr;
    hipLaunchKernelGGL(conv3x3_wgrad_mx_kernel<9>, dim3((unsigned)tiles, (unsigned)splits), dim3(512), VQK_WGMX_NST * 40960, st,
                       (const bf16_raw*)x, (const bf16_raw*)dy, dw, (const char*)zeros, g, pps, part);
    if (hipGetLastError() != hipSuccess) return VQK_ERR_LAUNCH;
    if (part) {
        hipLaunchKernelGGL(wgrad_mx_reduce_kernel, dim3((64 * 9 * 64) / 256, (unsigned)tiles), dim3(256), 0, st, (const float*)part, dw,
                           tiles, splits, g.cin);
        if (hipGetLastError() != hipSuccess) return VQK_ERR_LAUNCH;
    }
    return VQK_OK;
}

}  // namespace vqkd
