// ------------------------------------------------------------------------------------------------
// The two EDGE convs of the autoencoder in the fp32 compute modes (exact fp32 arithmetic, no matrix pipe):
//   vqvae/modules/autoencoder.py:132  conv_in  3 -> C   (the padded 3-channel image: 4 fp32 channels)
//   vqvae/modules/autoencoder.py:170  conv_out C -> 3   (the reconstruction, + bias + tanh)
// One side of these GEMMs is 4 channels wide: on 32x32 MFMA tiles 7/8 of the matrix pipe multiplies padding (measured before this
// file, bs 32 @256x256: conv_out forward 4.6 ms on the 128-cout-tile kernel, each weight gradient 5.7 ms on the general kernel at
// 6.8 TF).  They are memory-bound problems -- the wide tensor (1.07 GB at bs 32) has to be read or written once -- with 36
// multiply-adds per wide element: plain v_fma_f32.  A wave works on ONE pixel at a time, its 64 lanes are 64 wide channels, and the
// 3x3x4 window of the thin tensor is the same for every lane (staged rows in LDS, broadcast reads); conv_out's forward has the
// WEIGHTS wave-uniform instead (scalar registers; a lane is a pixel, the halo comes from LDS).  0.36-0.51 ms per launch = 2-3 TB/s.
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

// ---------------------------------------------------------------------------------------------- the thin operand through LDS
// Both kernels below walk image rows pixel by pixel with a wave = 64 wide channels; what they need of the THIN tensor is the 3x3x4
// window around the pixel -- the same for every lane.  Round 6, first form: scalar loads (s_load_dwordx4) + a sliding window in
// SGPRs: latency-bound (every pixel waited for a scalar-cache round trip: 0.5-0.8 ms per launch against ~0.25 at memory speed).  Now
// the block stages the R + 2 thin rows it needs ONCE in LDS -- zero rows / columns beyond the image, so the inner loop has no bounds
// test -- and a wave reads its window with same-address (broadcast) ds_read_b128, four pixels (twelve reads) ahead of the FMAs.
constexpr int THIN_R = 8;                                         // output rows per block (h % 8 == 0; else 4 / 2 / 1)

// rows y0 - 1 .. y0 + R of image `img` into smem [R + 2][w + 2] x 16 B (column 0 = image column -1)
__device__ __forceinline__ void stage_thin_rows(const float* __restrict__ thin, char* smem, int img, int y0, int R, int h, int w, int tid) {
    const int wp = w + 2, total = (R + 2) * wp;
    for (int e = tid; e < total; e += 256) {
        const int rr = e / wp, cc = e - rr * wp;
        const int yy = y0 + rr - 1, xx = cc - 1;
        const bool ok = yy >= 0 && yy < h && xx >= 0 && xx < w;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (ok) v = *reinterpret_cast<const f32x4*>(thin + (((int64_t)img * h + yy) * w + xx) * 4);
        *reinterpret_cast<f32x4*>(smem + (int64_t)e * 16) = v;
    }
}

// ---------------------------------------------------------------------------------------------- weight gradients
// G[wc][kh][kw][tc] = sum over pixels q of wide[q][wc] * thin[q + S (kh - 1, kw - 1)][tc], S = +1 (MODE 0: wide = dy, thin = x)
// or -1 (MODE 1: wide = x, thin = dy: dW[tc][kh][kw][wc] = sum_p dy[p][tc] x[p + tap][wc] re-indexed over q = p + tap).
// A wave owns 64 wide channels (one per lane) and walks the block's rows pixel by pixel: 36 v_fma_f32 per pixel and lane.
template <int MODE>
__global__ __launch_bounds__(256) void conv3x3_wgrad_thin_f32_kernel(const float* __restrict__ wide, const float* __restrict__ thin,
                                                                     float* __restrict__ dw, int n, int h, int w, int cw,
                                                                     float scale, int R) {
    constexpr int S = MODE == 0 ? 1 : -1;
    extern __shared__ __attribute__((aligned(16))) char smem[];  // thin rows, then reused for the block reduction
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncg = cw >> 6, wpg = 4 / ncg;                      // channel groups of 64, waves per group
    const int cg = wave % ncg, wslot = wave / ncg;
    const int bpi = h / R;                                       // blocks per image
    const int img = blockIdx.x / bpi, y0 = (blockIdx.x - img * bpi) * R;
    const int wp = w + 2;
    stage_thin_rows(thin, smem, img, y0, R, h, w, tid);
    __syncthreads();
    float acc[3][3][4];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][b][c] = 0.0f;

    for (int r = wslot; r < R; r += wpg) {
        const float* wrow = wide + (((int64_t)img * h + y0 + r) * w) * cw + cg * 64 + lane;
        const char* t0 = smem + (int64_t)r * wp * 16;            // staged row r = image row y0 + r - 1: the window's top row
        f32x4 tw[3][6];                                          // window columns x - 1 .. x + 4 of the three rows
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int j = 0; j < 2; ++j) tw[rr][j] = *reinterpret_cast<const f32x4*>(t0 + ((int64_t)rr * wp + j) * 16);
        // the wide operand runs EIGHT pixels ahead in registers (a wave's load is 256 contiguous bytes per pixel, nothing else hides
        // its HBM latency: three blocks of four waves per CU)
        float wn[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) wn[k] = wrow[(int64_t)min(k, w - 1) * cw];
        for (int x0 = 0; x0 < w; x0 += 4) {
            float wv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { wv[k] = wn[k]; wn[k] = wn[k + 4]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) wn[4 + k] = wrow[(int64_t)min(x0 + 8 + k, w - 1) * cw];     // (clamped: the loads stay unconditional)
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int j = 2; j < 6; ++j) tw[rr][j] = *reinterpret_cast<const f32x4*>(t0 + ((int64_t)rr * wp + x0 + j) * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const f32x4 tv = tw[1 + S * (kh - 1)][k + 1 + S * (kw - 1)];
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[kh][kw][c] = __builtin_fmaf(wv[k], tv[c], acc[kh][kw][c]);
                    }
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) { tw[rr][0] = tw[rr][4]; tw[rr][1] = tw[rr][5]; }
        }
    }
    // fold the waves of a channel group, then one atomic per element and block
    __syncthreads();                                             // everyone is done with the staged rows
    float* red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int c = 0; c < 4; ++c) red[(wave * 36 + (kh * 3 + kw) * 4 + c) * 64 + lane] = acc[kh][kw][c];
    __syncthreads();
    if constexpr (MODE == 0) {
        // dW[wc][tap][tc] is contiguous in (tap, tc): the block's threads walk the DESTINATION index so that a wave's atomics are
        // 256 consecutive bytes (lanes = channels made them 144-byte-strided: 4.7 M scattered atomics per launch)
        for (int d = tid; d < cw * 36; d += 256) {
            const int wc = d / 36, k = d - wc * 36, g2 = wc >> 6, l2 = wc & 63;
            float s = 0.0f;
            for (int ws = 0; ws < wpg; ++ws) s += red[((ws * ncg + g2) * 36 + k) * 64 + l2];
            atomicAdd(dw + d, s * scale);
        }
    } else if (wslot == 0) {
        const int wc = cg * 64 + lane;
#pragma unroll
        for (int k = 0; k < 36; ++k) {
            float s = 0.0f;
            for (int ws = 0; ws < wpg; ++ws) s += red[((ws * ncg + cg) * 36 + k) * 64 + lane];
            const int tap = k >> 2, tc = k & 3;
            atomicAdd(dw + ((int64_t)tc * 9 + tap) * cw + wc, s * scale);
        }
    }
}

// ---------------------------------------------------------------------------------------------- 4 -> C, forward / data gradient
// y[p][co] = sum_{tap, c < 4} x[p + tap][c] * w[co][tap][c] (+ bias[co]): a lane is an output channel with its 36 weights in
// registers, a wave walks the block's rows pixel by pixel and stores 256 contiguous bytes per pixel.  conv_in's forward
// (autoencoder.py:132) and, with wt = the [C][3][3][4] transposed / flipped operand, conv_out's data gradient (:170).
__global__ __launch_bounds__(256) void conv3x3_thin_in_f32_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                  const float* __restrict__ bias, float* __restrict__ y, int n,
                                                                  int h, int wd, int cout, int R) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncg = cout >> 6, wpg = 4 / ncg;
    const int cg = wave % ncg, wslot = wave / ncg;
    const int co = cg * 64 + lane;
    const int bpi = h / R;
    const int img = blockIdx.x / bpi, y0 = (blockIdx.x - img * bpi) * R;
    const int wp = wd + 2;
    stage_thin_rows(x, smem, img, y0, R, h, wd, tid);
    float wr[9][4];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(w + ((int64_t)co * 9 + tap) * 4);
#pragma unroll
        for (int c = 0; c < 4; ++c) wr[tap][c] = v[c];
    }
    const float b = bias ? bias[co] : 0.0f;
    __syncthreads();
    for (int r = wslot; r < R; r += wpg) {
        float* yrow = y + (((int64_t)img * h + y0 + r) * wd) * cout + co;
        const char* t0 = smem + (int64_t)r * wp * 16;
        f32x4 tw[3][6];
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int j = 0; j < 2; ++j) tw[rr][j] = *reinterpret_cast<const f32x4*>(t0 + ((int64_t)rr * wp + j) * 16);
        for (int x0 = 0; x0 < wd; x0 += 4) {
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int j = 2; j < 6; ++j) tw[rr][j] = *reinterpret_cast<const f32x4*>(t0 + ((int64_t)rr * wp + x0 + j) * 16);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float a0 = b, a1 = 0.0f;                         // two chains: 36 dependent fmas would serialise on their latency
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw) {
                        const f32x4 tv = tw[kh][k + kw];
                        a0 = __builtin_fmaf(wr[kh * 3 + kw][0], tv[0], a0);
                        a1 = __builtin_fmaf(wr[kh * 3 + kw][1], tv[1], a1);
                        a0 = __builtin_fmaf(wr[kh * 3 + kw][2], tv[2], a0);
                        a1 = __builtin_fmaf(wr[kh * 3 + kw][3], tv[3], a1);
                    }
                yrow[(int64_t)(x0 + k) * cout] = a0 + a1;
            }
#pragma unroll
            for (int rr = 0; rr < 3; ++rr) { tw[rr][0] = tw[rr][4]; tw[rr][1] = tw[rr][5]; }
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

static inline int thin_rows_per_block(int h) { return (h % 8) == 0 ? 8 : (h % 4) == 0 ? 4 : (h % 2) == 0 ? 2 : 1; }

// mode 0: thin = x [n][h][w][4], wide = dy [n][h][w][cw], dw [cw][3][3][4]; mode 1: wide = x [..][cw], thin = dy [..][4], dw [4][3][3][cw]
int launch_conv3x3_wgrad_thin_f32(int mode, const float* wide, const float* thin, float* dw, int n, int h, int w, int cw, float scale,
                                  hipStream_t st) {
    if ((cw != 64 && cw != 128 && cw != 256) || (w & 3) || w < 4) return VQK_ERR_SHAPE;
    const int R = thin_rows_per_block(h);
    size_t lds = (size_t)(R + 2) * (w + 2) * 16;
    if (lds < 4 * 36 * 64 * 4) lds = 4 * 36 * 64 * 4;            // the block reduction reuses the buffer
    if (lds > 64 * 1024) return VQK_ERR_SHAPE;
    const unsigned blocks = (unsigned)(n * (h / R));
    if (mode == 0)
        hipLaunchKernelGGL(conv3x3_wgrad_thin_f32_kernel<0>, dim3(blocks), dim3(256), lds, st, wide, thin, dw, n, h, w, cw, scale, R);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_thin_f32_kernel<1>, dim3(blocks), dim3(256), lds, st, wide, thin, dw, n, h, w, cw, scale, R);
    return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
}

// x fp32 [n][h][w][4], w fp32 [cout][3][3][4], y fp32 [n][h][w][cout]
int launch_conv3x3_thin_in_f32(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int cout, hipStream_t st) {
    if ((cout != 64 && cout != 128 && cout != 256) || (wd & 3) || wd < 4) return VQK_ERR_SHAPE;
    const int R = thin_rows_per_block(h);
    const size_t lds = (size_t)(R + 2) * (wd + 2) * 16;
    if (lds > 64 * 1024) return VQK_ERR_SHAPE;
    const unsigned blocks = (unsigned)(n * (h / R));
    hipLaunchKernelGGL(conv3x3_thin_in_f32_kernel, dim3(blocks), dim3(256), lds, st, x, w, bias, y, n, h, wd, cout, R);
    return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
}

}  // namespace vqkd
