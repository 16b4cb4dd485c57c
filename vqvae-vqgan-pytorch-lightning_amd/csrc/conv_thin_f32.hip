// ------------------------------------------------------------------------------------------------
// The two EDGE convs of the autoencoder in the fp32 compute modes (exact fp32 arithmetic, no matrix pipe):
//   vqvae/modules/autoencoder.py:132  conv_in  3 -> C   (the padded 3-channel image: 4 fp32 channels)
//   vqvae/modules/autoencoder.py:170  conv_out C -> 3   (the reconstruction, + bias + tanh)
// One side of these GEMMs is 4 channels wide: on 32x32 MFMA tiles 7/8 of the matrix pipe multiplies padding (measured before this
// file, bs 32 @256x256: conv_out forward 4.6 ms on the 128-cout-tile kernel, each weight gradient 5.7 ms on the general kernel at
// 6.8 TF).  They are memory-bound problems -- the wide tensor (1.07 GB at bs 32) has to be read once -- with 36 multiply-adds per
// wide element: plain v_fma_f32 with the THIN operand in SCALAR registers (a wave works on ONE pixel at a time, its 64 lanes are 64
// wide channels; the 3x3x4 window of the thin tensor is wave-uniform: s_load_dwordx4 + a sliding window) or, for conv_out's forward,
// with the WEIGHTS in scalar registers (a lane is a pixel, the halo comes from LDS).
//
//   conv3x3_thin_out_f32_kernel      y[p][0..3]   = act(sum_{tap,c} x[p + tap][c] w[co][tap][c] + b)          (C -> 4)
//   conv3x3_wgrad_thin_f32_kernel<0> dW[co][tap][0..3] += sum_p dy[p][co] x[p + tap][0..3]                    (thin x, wide dy)
//   conv3x3_wgrad_thin_f32_kernel<1> dW[0..3][tap][ci] += sum_p dy[p][0..3] x[p + tap][ci]                    (wide x, thin dy)
//   conv3x3_thin_in_f32_kernel       y[p][co]     = sum_{tap,c<4} x[p + tap][c] w[co][tap][c] (+ b)           (4 -> C: conv_in's
//                                    forward, and conv_out's data gradient with the transposed / flipped operand)
// ------------------------------------------------------------------------------------------------
#include "conv_geom.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ float thin_act(float v, int act) {
    if (act == 1) return tanhf(v);
    if (act == 2) return fmaxf(v, 0.0f);
    if (act == 3) return v > 0.0f ? v : 0.2f * v;
    return v;
}

// ---------------------------------------------------------------------------------------------- C -> 4, forward
// block = 8 x 32 output pixels, one per thread; 16-channel chunks of the 10 x 34 halo through LDS (80-byte rows: the 16 lanes of a
// ds_read_b128 group hit 16 distinct 16-byte bank slots), next chunk's halo in flight while this one is multiplied; weights are
// wave-uniform: scalar loads, one SGPR operand per v_fma_f32.
__global__ __launch_bounds__(256) void conv3x3_thin_out_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, const float* __restrict__ res,
                                                                   float* __restrict__ y, int n, int h, int wd, int cin, int act,
                                                                   float acc_scale, float out_gain) {
    constexpr int TH = 8, TW = 32, HW2 = TW + 2, HROWS = (TH + 2) * HW2, RS = 80;
    constexpr int PIECES = HROWS * 4, NSLOT = (PIECES + 255) / 256;      // 16-byte pieces of one chunk, per thread
    __shared__ __attribute__((aligned(16))) char smem[HROWS * RS];
    const int tid = threadIdx.x;
    const int tiles_x = wd / TW, tiles_y = h / TH;
    int t = blockIdx.x;
    const int txi = t % tiles_x; t /= tiles_x;
    const int tyi = t % tiles_y;
    const int img = t / tiles_y;
    const int py0 = tyi * TH, px0 = txi * TW;
    const int ty = tid >> 5, tx = tid & 31;
    const float* ximg = x + (int64_t)img * h * wd * cin;

    const float* src[NSLOT];
    unsigned dst[NSLOT];
    bool ok[NSLOT], live[NSLOT];
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
        const int e = tid + 256 * sl, row = e >> 2, slot = e & 3;
        live[sl] = e < PIECES;
        const int hy = row / HW2, hx = row - hy * HW2;
        const int iy = py0 + hy - 1, ix = px0 + hx - 1;
        ok[sl] = live[sl] && iy >= 0 && iy < h && ix >= 0 && ix < wd;
        src[sl] = ximg + ((int64_t)(ok[sl] ? iy : 0) * wd + (ok[sl] ? ix : 0)) * cin + slot * 4;
        dst[sl] = (unsigned)(row * RS + slot * 16);
    }
    f32x4 stage[NSLOT];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src[sl] + c * 16);       // clamped address: always valid
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
            stage[sl] = ok[sl] ? v : z;
        }
    };
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    const int nch = cin >> 4;
    load_chunk(0);
    for (int c = 0; c < nch; ++c) {
        __syncthreads();                                         // the previous chunk is consumed
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl)
            if (live[sl]) *reinterpret_cast<f32x4*>(smem + dst[sl]) = stage[sl];
        __syncthreads();
        if (c + 1 < nch) load_chunk(c + 1);                      // in flight during the multiply-adds below
        const float* wc = w + c * 16;
#pragma unroll 1                                                 // (unrolled, hipcc hoists all 576 scalar weight loads of a chunk and spills them)
        for (int tap = 0; tap < 9; ++tap) {
            const char* p = smem + ((ty + tap / 3) * HW2 + tx + tap % 3) * RS;
            f32x4 xv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) xv[q] = *reinterpret_cast<const f32x4*>(p + q * 16);
#pragma unroll
            for (int co = 0; co < 4; ++co) {
                const float* wr = wc + (int64_t)(co * 9 + tap) * cin;          // wave-uniform: scalar loads
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[co] = __builtin_fmaf(xv[q][e], wr[q * 4 + e], acc[co]);
            }
        }
    }
    const int64_t o = (((int64_t)img * h + py0 + ty) * wd + px0 + tx) * 4;
    f32x4 v;
#pragma unroll
    for (int co = 0; co < 4; ++co) v[co] = thin_act(acc[co] * acc_scale + (bias ? bias[co] : 0.0f), act) * out_gain;
    if (res) v += *reinterpret_cast<const f32x4*>(res + o);
    *reinterpret_cast<f32x4*>(y + o) = v;
}

// wave-uniform 16-byte load (the address is uniform by construction: readfirstlane'd wave id, block id, loop counters)
__device__ __forceinline__ u32x4 uniform_load16(const float* p) {
    return *reinterpret_cast<const u32x4*>(p);
}

// ---------------------------------------------------------------------------------------------- weight gradients
// G[wc][kh][kw][tc] = sum over pixels q of wide[q][wc] * thin[q + S (kh - 1, kw - 1)][tc], S = +1 (MODE 0: wide = dy, thin = x)
// or -1 (MODE 1: wide = x, thin = dy: dW[tc][kh][kw][wc] = sum_p dy[p][tc] x[p + tap][wc] re-indexed over q = p + tap).
// A wave owns 64 wide channels (one per lane) and walks image rows pixel by pixel; the thin tensor's 3x3 window (9 x 4 floats) lives
// in scalar registers and slides along the row (3 scalar 16-byte loads per pixel); 36 v_fma_f32 per pixel and lane.
template <int MODE>
__global__ __launch_bounds__(256) void conv3x3_wgrad_thin_f32_kernel(const float* __restrict__ wide, const float* __restrict__ thin,
                                                                     float* __restrict__ dw, int n, int h, int w, int cw,
                                                                     float scale, int rows_per_block) {
    constexpr int S = MODE == 0 ? 1 : -1;
    __shared__ float red[4 * 36 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncg = cw >> 6, wpg = 4 / ncg;                      // channel groups of 64, waves per group
    const int cg = wave % ncg, wslot = wave / ncg;
    const int total_rows = n * h;
    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(total_rows, row0 + rows_per_block);
    float acc[3][3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][b][c] = 0.0f;

    for (int r = row0 + wslot; r < row1; r += wpg) {
        const int img = r / h, yy = r - img * h;
        const float* wrow = wide + (int64_t)r * w * cw + cg * 64 + lane;
        const float* trow[3];
        unsigned rmask[3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int ty = yy + rr - 1;
            const bool okr = ty >= 0 && ty < h;
            trow[rr] = thin + ((int64_t)img * h + (okr ? ty : yy)) * w * 4;
            rmask[rr] = okr ? 0xffffffffu : 0u;
        }
        // window columns j = 0, 1, 2 <-> image columns x - 1, x, x + 1
        u32x4 tw[3][3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            tw[rr][0] = z;
            tw[rr][1] = uniform_load16(trow[rr]) & rmask[rr];
            tw[rr][2] = uniform_load16(trow[rr] + 4) & rmask[rr];
        }
        for (int x0 = 0; x0 < w; x0 += 4) {
            float wv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) wv[k] = wrow[(int64_t)(x0 + k) * cw];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int xx = x0 + k;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const u32x4 tv = tw[1 + S * (kh - 1)][1 + S * (kw - 1)];
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[kh][kw][c] = __builtin_fmaf(wv[k], __uint_as_float(tv[c]), acc[kh][kw][c]);
                    }
                // slide: column xx + 2 enters (zero beyond the image; the load address stays inside the row)
                const bool okc = xx + 2 < w;
                const unsigned cm = okc ? 0xffffffffu : 0u;
                const int cx = okc ? xx + 2 : w - 1;
#pragma unroll
                for (int rr = 0; rr < 3; ++rr) {
                    tw[rr][0] = tw[rr][1];
                    tw[rr][1] = tw[rr][2];
                    tw[rr][2] = uniform_load16(trow[rr] + cx * 4) & (rmask[rr] & cm);
                }
            }
        }
    }
    // fold the waves of a channel group, then one atomic per element and block
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[(wave * 36 + (kh * 3 + kw) * 4 + c) * 64 + lane] = acc[kh][kw][c];
    __syncthreads();
    if (wslot == 0) {
        const int wc = cg * 64 + lane;
#pragma unroll
        for (int k = 0; k < 36; ++k) {
            float s = 0.0f;
            for (int ws = 0; ws < wpg; ++ws) s += red[((ws * ncg + cg) * 36 + k) * 64 + lane];
            const int tap = k >> 2, tc = k & 3;
            float* dst = MODE == 0 ? dw + ((int64_t)wc * 9 + tap) * 4 + tc : dw + ((int64_t)tc * 9 + tap) * cw + wc;
            atomicAdd(dst, s * scale);
        }
    }
}

// ---------------------------------------------------------------------------------------------- 4 -> C, forward / data gradient
// y[p][co] = sum_{tap, c < 4} x[p + tap][c] * w[co][tap][c] (+ bias[co]): a lane is an output channel with its 36 weights in
// registers, a wave walks image rows pixel by pixel with the thin 3x3x4 window in scalar registers and stores 256 contiguous bytes
// per pixel.  conv_in's forward (autoencoder.py:132) and, with wt = the [C][3][3][4] transposed / flipped operand, conv_out's data
// gradient (:170).
__global__ __launch_bounds__(256) void conv3x3_thin_in_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ y, int n,
                                                                  int h, int wd, int cout, int rows_per_block) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncg = cout >> 6, wpg = 4 / ncg;
    const int cg = wave % ncg, wslot = wave / ncg;
    const int co = cg * 64 + lane;
    float wr[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + ((int64_t)co * 9 + tap) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) wr[tap][c] = v[c];
    }
    const float b = bias ? bias[co] : 0.0f;
    const int total_rows = n * h;
    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(total_rows, row0 + rows_per_block);
    for (int r = row0 + wslot; r < row1; r += wpg) {
        const int img = r / h, yy = r - img * h;
        float* yrow = y + (int64_t)r * wd * cout + co;
        const float* trow[3];
        unsigned rmask[3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const int ty = yy + rr - 1;
            const bool okr = ty >= 0 && ty < h;
            trow[rr] = x + ((int64_t)img * h + (okr ? ty : yy)) * wd * 4;
            rmask[rr] = okr ? 0xffffffffu : 0u;
        }
        u32x4 tw[3][3];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr) {
            const u32x4 z = {0u, 0u, 0u, 0u};
            tw[rr][0] = z;
            tw[rr][1] = uniform_load16(trow[rr]) & rmask[rr];
            tw[rr][2] = uniform_load16(trow[rr] + 4) & rmask[rr];
        }
        for (int xx = 0; xx < wd; ++xx) {
            float a0 = b, a1 = 0.0f;                             // two chains: 36 dependent fmas would serialise on the 4-cycle latency
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const u32x4 tv = tw[kh][kw];
                    a0 = __builtin_fmaf(wr[kh * 3 + kw][0], __uint_as_float(tv[0]), a0);
                    a1 = __builtin_fmaf(wr[kh * 3 + kw][1], __uint_as_float(tv[1]), a1);
                    a0 = __builtin_fmaf(wr[kh * 3 + kw][2], __uint_as_float(tv[2]), a0);
                    a1 = __builtin_fmaf(wr[kh * 3 + kw][3], __uint_as_float(tv[3]), a1);
                }
            yrow[(int64_t)xx * cout] = a0 + a1;
            const bool okc = xx + 2 < wd;
            const unsigned cm = okc ? 0xffffffffu : 0u;
            const int cx = okc ? xx + 2 : wd - 1;
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) {
                tw[rr][0] = tw[rr][1];
                tw[rr][1] = tw[rr][2];
                tw[rr][2] = uniform_load16(trow[rr] + cx * 4) & (rmask[rr] & cm);
            }
        }
    }
}

}  // namespace

namespace vqkd {

// x fp32 [n][h][w][cin], w fp32 [4][3][3][cin], y fp32 [n][h][w][4]
int launch_conv3x3_thin_out_f32(const float* x, const float* w, const float* bias, const float* res, float* y, int n, int h, int wd,
                                int cin, int act, float acc_scale, float out_gain, hipStream_t st) {
    if ((h & 7) || (wd & 31) || (cin & 15)) return VQK_ERR_SHAPE;
    const int64_t tiles = (int64_t)n * (h / 8) * (wd / 32);
    if (tiles > 0x7fffffff) return VQK_ERR_SHAPE;
    hipLaunchKernelGGL(conv3x3_thin_out_f32_kernel, dim3((unsigned)tiles), dim3(256), 0, st, x, w, bias, res, y, n, h, wd, cin, act,
                       acc_scale, out_gain);
    return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
}

// mode 0: thin = x [n][h][w][4], wide = dy [n][h][w][cw], dw [cw][3][3][4]; mode 1: wide = x [..][cw], thin = dy [..][4], dw [4][3][3][cw]
int launch_conv3x3_wgrad_thin_f32(int mode, const float* wide, const float* thin, float* dw, int n, int h, int w, int cw, float scale,
                                  hipStream_t st) {
    if ((cw != 64 && cw != 128 && cw != 256) || (w & 3) || w < 4) return VQK_ERR_SHAPE;
    const int rows = n * h;
    int rpb = (rows + 1023) / 1024;                              // ~4 blocks per CU
    const int wpg = 4 / (cw >> 6);
    rpb = ((rpb + wpg - 1) / wpg) * wpg;
    if (rpb < wpg) rpb = wpg;
    const unsigned blocks = (unsigned)((rows + rpb - 1) / rpb);
    if (mode == 0)
        hipLaunchKernelGGL(conv3x3_wgrad_thin_f32_kernel<0>, dim3(blocks), dim3(256), 0, st, wide, thin, dw, n, h, w, cw, scale, rpb);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_thin_f32_kernel<1>, dim3(blocks), dim3(256), 0, st, wide, thin, dw, n, h, w, cw, scale, rpb);
    return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
}

// x fp32 [n][h][w][4], w fp32 [cout][3][3][4], y fp32 [n][h][w][cout]
int launch_conv3x3_thin_in_f32(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int cout, hipStream_t st) {
    if ((cout != 64 && cout != 128 && cout != 256) || wd < 2) return VQK_ERR_SHAPE;
    const int rows = n * h;
    int rpb = (rows + 2047) / 2048;
    const int wpg = 4 / (cout >> 6);
    rpb = ((rpb + wpg - 1) / wpg) * wpg;
    if (rpb < wpg) rpb = wpg;
    const unsigned blocks = (unsigned)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(conv3x3_thin_in_f32_kernel, dim3(blocks), dim3(256), 0, st, x, w, bias, y, n, h, wd, cout, rpb);
    return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
}

}  // namespace vqkd
