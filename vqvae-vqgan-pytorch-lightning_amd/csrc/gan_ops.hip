// Small kernels of the VQ-GAN losses (LPIPS / StyleGAN2 discriminator path), NHWC:
//   activation backward (bias_act grad=1 semantics for relu / lrelu, bias_act.cu:60-75 of the reference)
//   upfirdn2d in NHWC (upfirdn2d.cu:29-92 semantics: zero-stuff, pad, FIR, decimate)
//   max-pool 2x2 fwd/bwd (torchvision vgg16.features pools), per-channel affine (LPIPS z-score, networks.py:51-52)
//   LPIPS tap: unit-normalise over channels, squared difference, 1x1 "lin", spatial mean (lpips.py:31-38, utils.py:6-8)
//   minibatch-stddev layer fwd/bwd (discriminator.py:271-293), L1 loss, GAN losses (loss.py:11-51)
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

template <typename T>
__global__ __launch_bounds__(256) void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                      int64_t n, int act, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float yv = Elem<T>::ld(y + i);
        float g = Elem<T>::ld(dy + i) * scale;
        if (act == 1) g *= 1.0f - yv * yv;
        else if (act == 2) g = yv > 0.0f ? g : 0.0f;
        else if (act == 3) g = yv > 0.0f ? g : 0.2f * g;
        Elem<T>::st(dx + i, g);
    }
}

// the same on whole 16-byte vectors (n a multiple of the vector length, 16-byte aligned pointers): the scalar form above
// moved 2 bytes per lane and instruction; measured 33.1 -> 31.7 us per call over the VQ-GAN step (the pass is bandwidth-bound)
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_vec_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                          int64_t nvec, int act, float scale) {
    constexpr int V = Vec16<T>::N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float yv[V], g[V];
        Vec16<T>::load(y + i * V, yv);
        Vec16<T>::load(dy + i * V, g);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float t = g[k] * scale;
            if (act == 1) t *= 1.0f - yv[k] * yv[k];
            else if (act == 2) t = yv[k] > 0.0f ? t : 0.0f;
            else if (act == 3) t = yv[k] > 0.0f ? t : 0.2f * t;
            g[k] = t;
        }
        Vec16<T>::store(dx + i * V, g);
    }
}

// act_bwd on [rows][c] plus the column sums of its RESULT (the bias gradient, rounded as stored): a thread owns one 16-byte
// channel slot and strides over rows (the colsum kernel's mapping), so the bias gradient costs no second pass over dx
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_colsum_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                             int64_t rows, int c, int64_t rows_per_block, int act, float scale,
                                                             float* __restrict__ colsum, float cs_scale) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sh = reinterpret_cast<float*>(smem);            // [c]
    for (int i = threadIdx.x; i < c; i += 256) sh[i] = 0.f;
    __syncthreads();
    const int vpp = c / V;                                  // <= 256 (checked by the launcher)
    const int slot = threadIdx.x % vpp, rlane = threadIdx.x / vpp, rstep = 256 / vpp;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    if (rlane < rstep) {
        float a[V];
#pragma unroll
        for (int k = 0; k < V; ++k) a[k] = 0.f;
        for (int64_t r = r0 + rlane; r < r1; r += rstep) {
            float yv[V], g[V];
            Vec16<T>::load(y + r * c + slot * V, yv);
            Vec16<T>::load(dy + r * c + slot * V, g);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                float t = g[k] * scale;
                if (act == 1) t *= 1.0f - yv[k] * yv[k];
                else if (act == 2) t = yv[k] > 0.0f ? t : 0.0f;
                else if (act == 3) t = yv[k] > 0.0f ? t : 0.2f * t;
                g[k] = t;
            }
            Vec16<T>::store(dx + r * c + slot * V, g);
#pragma unroll
            for (int k = 0; k < V; ++k) {                   // the sum of the values AS STORED (what a separate colsum pass reads)
                if constexpr (sizeof(T) == 2) a[k] += bf16_to_f32(f32_to_bf16(g[k]));
                else a[k] += g[k];
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) atomicAdd(sh + slot * V + k, a[k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < c; i += 256) atomicAdd(colsum + i, sh[i] * cs_scale);
}

// y[n,oy,ox,:] = gain * sum_{fy,fx} F[fy][fx] * U[oy*down + fy - pad0][...]; U = zero-stuffed x
template <typename T>
__global__ __launch_bounds__(256) void upfirdn_nhwc_kernel(const T* __restrict__ x, const float* __restrict__ f,
                                                           T* __restrict__ y, int n, int h, int w, int c, int fh, int fw,
                                                           int upx, int upy, int downx, int downy, int px0, int py0,
                                                           int flip, float gain, int oh, int ow) {
    constexpr int V = Vec16<T>::N;
    const int vpp = c / V;
    const int64_t total = (int64_t)n * oh * ow * vpp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpp);
        int64_t p = i / vpp;
        const int ox = (int)(p % ow); p /= ow;
        const int oy = (int)(p % oh);
        const int img = (int)(p / oh);
        float acc[V];
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] = 0.0f;
        for (int ky = 0; ky < fh; ++ky) {
            const int uy = oy * downy + ky - py0;
            if (uy < 0 || uy % upy) continue;
            const int iy = uy / upy;
            if (iy >= h) continue;
            for (int kx = 0; kx < fw; ++kx) {
                const int ux = ox * downx + kx - px0;
                if (ux < 0 || ux % upx) continue;
                const int ix = ux / upx;
                if (ix >= w) continue;
                const float fv = flip ? f[ky * fw + kx] : f[(fh - 1 - ky) * fw + (fw - 1 - kx)];
                float xv[V];
                Vec16<T>::load(x + (((int64_t)img * h + iy) * w + ix) * c + v * V, xv);
#pragma unroll
                for (int k = 0; k < V; ++k) acc[k] = __fmaf_rn(xv[k], fv, acc[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < V; ++k) acc[k] *= gain;
        Vec16<T>::store(y + i * V, acc);
    }
}

template <typename T, bool BWD>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out,
                                                      int n, int h, int w, int c) {
    constexpr int V = Vec16<T>::N;
    const int oh = h >> 1, ow = w >> 1, vpp = c / V;
    const int64_t total = (int64_t)n * oh * ow * vpp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpp);
        int64_t p = i / vpp;
        const int ox = (int)(p % ow); p /= ow;
        const int oy = (int)(p % oh);
        const int img = (int)(p / oh);
        const int64_t base = (((int64_t)img * h + 2 * oy) * w + 2 * ox) * c + v * V;
        float a[4][V];
        Vec16<T>::load(x + base, a[0]); Vec16<T>::load(x + base + c, a[1]);
        Vec16<T>::load(x + base + (int64_t)w * c, a[2]); Vec16<T>::load(x + base + (int64_t)w * c + c, a[3]);
        if (!BWD) {
            float o[V];
#pragma unroll
            for (int k = 0; k < V; ++k) o[k] = fmaxf(fmaxf(a[0][k], a[1][k]), fmaxf(a[2][k], a[3][k]));
            Vec16<T>::store(out + i * V, o);
        } else {
            float g[V], o[4][V];
            Vec16<T>::load(dy + i * V, g);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                int arg = 0; float m = a[0][k];                 // first maximum in window scan order wins
#pragma unroll
                for (int q = 1; q < 4; ++q) if (a[q][k] > m) { m = a[q][k]; arg = q; }
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q][k] = q == arg ? g[k] : 0.0f;
            }
            Vec16<T>::store(out + base, o[0]); Vec16<T>::store(out + base + c, o[1]);
            Vec16<T>::store(out + base + (int64_t)w * c, o[2]); Vec16<T>::store(out + base + (int64_t)w * c + c, o[3]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void channel_affine_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, T* __restrict__ y,
                                                             int64_t npix, int c) {
    const int64_t total = npix * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % c);
        Elem<T>::st(y + i, __fmaf_rn(Elem<T>::ld(x + i), scale[ch], shift ? shift[ch] : 0.0f));
    }
}

// one wavefront per pixel.  out[img] += (1/hw) sum_c w_c (fx_c/(|fx|+eps) - fy_c/(|fy|+eps))^2
// segmented butterfly sum over groups of `width` consecutive lanes (width a power of two <= 64)
__device__ __forceinline__ float seg_sum(float v, int width) {
    for (int off = width >> 1; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// Vectorised form: a lane owns one 16-byte channel slot of a pixel (c / V lanes per pixel, 64 / (c / V) pixels per wave
// pass), both feature vectors are read ONCE into registers, the channel reductions are segmented shuffles, and the
// per-image spatial mean is accumulated per wave and flushed with one atomic per (wave, image) -- the scalar version
// issued one global atomic per PIXEL onto n addresses (3.5 ms per tap at 16 x 256^2).
template <typename T>
__global__ __launch_bounds__(256) void lpips_tap_fwd_vec_kernel(const T* __restrict__ fx, const T* __restrict__ fy,
                                                                const float* __restrict__ lin, int64_t npix, int64_t hw,
                                                                int c, int64_t pix_per_block, float* __restrict__ out) {
    constexpr int V = Vec16<T>::N;
    const int lpp = c / V;                                   // lanes per pixel
    const int ppw = 64 / lpp;                                // pixels per wave pass
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / lpp, slot = lane - sub * lpp;
    float lw[V];
#pragma unroll
    for (int i = 0; i < V; ++i) lw[i] = lin[slot * V + i];
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
    int64_t cur_img = -1;
    float img_acc = 0.f;
    const float inv_hw = 1.0f / (float)hw;
    for (int64_t base = p0 + (int64_t)wave * ppw; base < p1; base += 4 * ppw) {
        const int64_t pix = base + sub;
        const bool ok = pix < p1;
        float a[V], b[V];
        if (ok) { Vec16<T>::load(fx + pix * c + slot * V, a); Vec16<T>::load(fy + pix * c + slot * V, b); }
        else {
#pragma unroll
            for (int i = 0; i < V; ++i) { a[i] = 0.f; b[i] = 0.f; }
        }
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) { sx = __fmaf_rn(a[i], a[i], sx); sy = __fmaf_rn(b[i], b[i], sy); }
        sx = seg_sum(sx, lpp); sy = seg_sum(sy, lpp);
        const float rx = 1.0f / (sqrtf(sx) + 1e-10f), ry = 1.0f / (sqrtf(sy) + 1e-10f);
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) { const float d = a[i] * rx - b[i] * ry; acc = __fmaf_rn(lw[i] * d, d, acc); }
        acc = seg_sum(acc, lpp);
        if (ok && slot == 0) {
            const int64_t img = pix / hw;
            if (img != cur_img) {
                if (cur_img >= 0) atomicAdd(out + cur_img, img_acc * inv_hw);
                cur_img = img; img_acc = 0.f;
            }
            img_acc += acc;
        }
    }
    // flush: normally the whole wave ended on one image -> one atomic per wave
    const int64_t mine = slot == 0 ? cur_img : -2;
    const int64_t first = __shfl(cur_img, 0, 64);
    const bool uniform = __all(mine == -2 || mine == first || mine == -1);
    // ... and normally the block's four waves too -> one atomic per block (the wave sums meet in LDS, fixed order)
    __shared__ long long s_img[4];
    __shared__ float s_val[4];
    float tot = 0.f;
    if (uniform) tot = wave_sum((slot == 0 && cur_img >= 0) ? img_acc : 0.f);
    else if (slot == 0 && cur_img >= 0) atomicAdd(out + cur_img, img_acc * inv_hw);
    if (lane == 0) { s_img[wave] = uniform ? (long long)first : -1; s_val[wave] = tot; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const bool same = s_img[0] == s_img[1] && s_img[0] == s_img[2] && s_img[0] == s_img[3];
        if (same) {
            if (s_img[0] >= 0) atomicAdd(out + s_img[0], ((s_val[0] + s_val[1]) + (s_val[2] + s_val[3])) * inv_hw);
        } else {
            for (int w = 0; w < 4; ++w)
                if (s_img[w] >= 0) atomicAdd(out + s_img[w], s_val[w] * inv_hw);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void lpips_tap_fwd_kernel(const T* __restrict__ fx, const T* __restrict__ fy,
                                                            const float* __restrict__ lin, int64_t npix, int64_t hw, int c,
                                                            float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    for (int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += (int64_t)gridDim.x * 4) {
        const T* px = fx + pix * c;
        const T* py = fy + pix * c;
        float sx = 0.f, sy = 0.f;
        for (int k = lane; k < c; k += 64) {
            const float a = Elem<T>::ld(px + k), b = Elem<T>::ld(py + k);
            sx = __fmaf_rn(a, a, sx); sy = __fmaf_rn(b, b, sy);
        }
        sx = wave_sum(sx); sy = wave_sum(sy);
        const float rx = 1.0f / (sqrtf(sx) + 1e-10f), ry = 1.0f / (sqrtf(sy) + 1e-10f);
        float acc = 0.f;
        for (int k = lane; k < c; k += 64) {
            const float d = Elem<T>::ld(px + k) * rx - Elem<T>::ld(py + k) * ry;
            acc = __fmaf_rn(lin[k] * d, d, acc);
        }
        acc = wave_sum(acc);
        if (lane == 0) atomicAdd(out + pix / hw, acc / (float)hw);
    }
}

// d fy:  t_c = -2 g w_c d_c / hw ;  dfy_k = t_k/(ny+eps) - fy_k <t, fy> / (ny (ny+eps)^2)
template <typename T>
__global__ __launch_bounds__(256) void lpips_tap_bwd_kernel(const T* __restrict__ fx, const T* __restrict__ fy,
                                                            const float* __restrict__ lin, const float* __restrict__ gout,
                                                            float gscale, int64_t npix, int64_t hw, int c,
                                                            T* __restrict__ dfy) {
    const int lane = threadIdx.x & 63;
    for (int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); pix < npix; pix += (int64_t)gridDim.x * 4) {
        const T* px = fx + pix * c;
        const T* py = fy + pix * c;
        const float g = gscale * (gout ? gout[pix / hw] : 1.0f) / (float)hw;     // upstream gradient of THIS pixel's image
        float sx = 0.f, sy = 0.f;
        for (int k = lane; k < c; k += 64) {
            const float a = Elem<T>::ld(px + k), b = Elem<T>::ld(py + k);
            sx = __fmaf_rn(a, a, sx); sy = __fmaf_rn(b, b, sy);
        }
        sx = wave_sum(sx); sy = wave_sum(sy);
        const float nx = sqrtf(sx), ny = sqrtf(sy);
        const float rx = 1.0f / (nx + 1e-10f), ry = 1.0f / (ny + 1e-10f);
        float tdot = 0.f;
        for (int k = lane; k < c; k += 64) {
            const float b = Elem<T>::ld(py + k);
            const float d = Elem<T>::ld(px + k) * rx - b * ry;
            tdot = __fmaf_rn(-2.0f * g * lin[k] * d, b, tdot);
        }
        tdot = wave_sum(tdot);
        const float corr = ny > 0.f ? tdot * ry * ry / ny : 0.0f;
        for (int k = lane; k < c; k += 64) {
            const float b = Elem<T>::ld(py + k);
            const float d = Elem<T>::ld(px + k) * rx - b * ry;
            Elem<T>::st(dfy + pix * c + k, -2.0f * g * lin[k] * d * ry - b * corr);
        }
    }
}

// Vectorised backward (same lane <-> 16-byte channel slot mapping as lpips_tap_fwd_vec_kernel): both feature vectors are
// read once, the three channel reductions are segmented shuffles, one 16-byte store per lane.  The scalar form reads
// fx / fy three times with 2-byte loads (185 us average per tap at n16; this one moves the same bytes at HBM rate).
template <typename T>
__global__ __launch_bounds__(256) void lpips_tap_bwd_vec_kernel(const T* __restrict__ fx, const T* __restrict__ fy,
                                                                const float* __restrict__ lin, const float* __restrict__ gout,
                                                                float gscale, int64_t npix, int64_t hw, int c,
                                                                int64_t pix_per_block, T* __restrict__ dfy) {
    constexpr int V = Vec16<T>::N;
    const int lpp = c / V;
    const int ppw = 64 / lpp;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane / lpp, slot = lane - sub * lpp;
    float lw[V];
#pragma unroll
    for (int i = 0; i < V; ++i) lw[i] = lin[slot * V + i];
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block, p1 = min(npix, p0 + pix_per_block);
    for (int64_t base = p0 + (int64_t)wave * ppw; base < p1; base += 4 * ppw) {
        const int64_t pix = base + sub;
        const bool ok = pix < p1;
        float a[V], b[V];
        if (ok) { Vec16<T>::load(fx + pix * c + slot * V, a); Vec16<T>::load(fy + pix * c + slot * V, b); }
        else {
#pragma unroll
            for (int i = 0; i < V; ++i) { a[i] = 0.f; b[i] = 0.f; }
        }
        const float g = ok ? gscale * (gout ? gout[pix / hw] : 1.0f) / (float)hw : 0.0f;
        float sx = 0.f, sy = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) { sx = __fmaf_rn(a[i], a[i], sx); sy = __fmaf_rn(b[i], b[i], sy); }
        sx = seg_sum(sx, lpp); sy = seg_sum(sy, lpp);
        const float nx = sqrtf(sx), ny = sqrtf(sy);
        const float rx = 1.0f / (nx + 1e-10f), ry = 1.0f / (ny + 1e-10f);
        float t[V], tdot = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float d = a[i] * rx - b[i] * ry;
            t[i] = -2.0f * g * lw[i] * d;
            tdot = __fmaf_rn(t[i], b[i], tdot);
        }
        tdot = seg_sum(tdot, lpp);
        const float corr = ny > 0.f ? tdot * ry * ry / ny : 0.0f;
        float o[V];
#pragma unroll
        for (int i = 0; i < V; ++i) o[i] = t[i] * ry - b[i] * corr;
        if (ok) Vec16<T>::store(dfy + pix * c + slot * V, o);
    }
}

// minibatch-stddev: sample b = g*(n/G) + m.  stat[m] = mean_{c,h,w} sqrt(var_g + 1e-8).  one block per m.
template <typename T>
__global__ __launch_bounds__(1024) void mbstd_stat_kernel(const T* __restrict__ x, int n, int64_t chw, int gsz,
                                                         float* __restrict__ stat) {
    __shared__ float part[16];
    const int m = blockIdx.x, cols = n / gsz;
    float acc = 0.f;
    for (int64_t e = threadIdx.x; e < chw; e += blockDim.x) {
        float mean = 0.f, v[8];
        for (int g = 0; g < gsz; ++g) { v[g] = Elem<T>::ld(x + ((int64_t)(g * cols + m)) * chw + e); mean += v[g]; }
        mean /= (float)gsz;
        float var = 0.f;
        for (int g = 0; g < gsz; ++g) var += (v[g] - mean) * (v[g] - mean);
        acc += sqrtf(var / (float)gsz + 1e-8f);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];         // fixed order
        stat[m] = t / (float)chw;
    }
}

// y[b][pix][0..c) = x ; y[b][pix][c] = stat[b % cols] ; y[b][pix][c+1..cp) = 0
template <typename T>
__global__ __launch_bounds__(256) void mbstd_concat_kernel(const T* __restrict__ x, const float* __restrict__ stat,
                                                           T* __restrict__ y, int n, int64_t hw, int c, int cp, int cols) {
    const int64_t total = (int64_t)n * hw * cp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ch = (int)(i % cp);
        const int64_t pix = i / cp;
        const int b = (int)(pix / hw);
        float v = 0.f;
        if (ch < c) v = Elem<T>::ld(x + pix * c + ch);
        else if (ch == c) v = stat[b % cols];
        Elem<T>::st(y + i, v);
    }
}

// dx[b][e] = dy[b][e (channels < c)] + dstat[m] * (x - mean_g) / (G * chw * std_e),  dstat[m] = sum dy[.., channel c]
template <typename T>
__global__ __launch_bounds__(1024) void mbstd_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                        int n, int64_t hw, int c, int cp, int gsz) {
    __shared__ float part[16];
    __shared__ float dstat;
    const int m = blockIdx.x, cols = n / gsz;
    const int64_t chw = hw * c;
    float acc = 0.f;
    for (int64_t q = threadIdx.x; q < (int64_t)gsz * hw; q += blockDim.x) {
        const int g = (int)(q / hw);
        const int64_t pix = (int64_t)(g * cols + m) * hw + (q - g * hw);
        acc += Elem<T>::ld(dy + pix * cp + c);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += part[w];
        dstat = t;
    }
    __syncthreads();
    const float ds = dstat / ((float)gsz * (float)chw);
    for (int64_t e = threadIdx.x; e < chw; e += blockDim.x) {
        const int64_t pixe = (int64_t)((unsigned)e / (unsigned)c); const int ch = (int)(e - pixe * c);   // chw < 2^31 (launcher)
        float mean = 0.f, v[8];
        for (int g = 0; g < gsz; ++g) { v[g] = Elem<T>::ld(x + ((int64_t)(g * cols + m)) * chw + e); mean += v[g]; }
        mean /= (float)gsz;
        float var = 0.f;
        for (int g = 0; g < gsz; ++g) var += (v[g] - mean) * (v[g] - mean);
        const float sd = sqrtf(var / (float)gsz + 1e-8f);
        for (int g = 0; g < gsz; ++g) {
            const int64_t b = (int64_t)(g * cols + m);
            const float up = Elem<T>::ld(dy + (b * hw + pixe) * cp + ch);
            Elem<T>::st(dx + b * chw + e, up + ds * (v[g] - mean) / sd);
        }
    }
}

// second-order piece of the minibatch-stddev layer (R1 regularisation differentiates the backward pass):
// first backward   dx_ge = up_ge + c_m r_ge,  r = (x - mean_g)/sd,  c_m = ds_m/(G E),  ds_m = sum of dy's extra channel
// given v = cotangent of dx:  d(dy)[.., ch<c] = v ;  d(dy)[.., c] = A_m/(G E), A_m = sum_{g,e} v r  (every pixel of the
// group) ;  d(x)_ge = c_m [ (v_ge - mean_g v)/sd - r_ge (sum_g v r)/(G sd) ].   one block per m.
template <typename T>
__global__ __launch_bounds__(256) void mbstd_bwd_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            const T* __restrict__ v, T* __restrict__ ddy,
                                                            T* __restrict__ dxx, int n, int64_t hw, int c, int cp, int gsz) {
    __shared__ float part[2][4];
    __shared__ float tot[2];
    const int m = blockIdx.x, cols = n / gsz;
    const int64_t chw = hw * c;
    float ds = 0.f, am = 0.f;
    for (int64_t q = threadIdx.x; q < (int64_t)gsz * hw; q += 256) {
        const int g = (int)(q / hw);
        ds += Elem<T>::ld(dy + ((int64_t)(g * cols + m) * hw + (q - g * hw)) * cp + c);
    }
    for (int64_t e = threadIdx.x; e < chw; e += 256) {
        float mean = 0.f, xv[8], vv[8];
        for (int g = 0; g < gsz; ++g) {
            xv[g] = Elem<T>::ld(x + ((int64_t)(g * cols + m)) * chw + e);
            vv[g] = Elem<T>::ld(v + ((int64_t)(g * cols + m)) * chw + e);
            mean += xv[g];
        }
        mean /= (float)gsz;
        float var = 0.f;
        for (int g = 0; g < gsz; ++g) var += (xv[g] - mean) * (xv[g] - mean);
        const float sd = sqrtf(var / (float)gsz + 1e-8f);
        for (int g = 0; g < gsz; ++g) am += vv[g] * (xv[g] - mean) / sd;
    }
    ds = wave_sum(ds); am = wave_sum(am);
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = ds; part[1][threadIdx.x >> 6] = am; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tot[0] = (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]);
        tot[1] = (part[1][0] + part[1][1]) + (part[1][2] + part[1][3]);
    }
    __syncthreads();
    const float ge = (float)gsz * (float)chw;
    const float cm = tot[0] / ge, aext = tot[1] / ge;
    for (int64_t e = threadIdx.x; e < chw; e += 256) {
        const int64_t pixe = e / c; const int ch = (int)(e - pixe * c);
        float mean = 0.f, vmean = 0.f, xv[8], vv[8];
        for (int g = 0; g < gsz; ++g) {
            xv[g] = Elem<T>::ld(x + ((int64_t)(g * cols + m)) * chw + e);
            vv[g] = Elem<T>::ld(v + ((int64_t)(g * cols + m)) * chw + e);
            mean += xv[g]; vmean += vv[g];
        }
        mean /= (float)gsz; vmean /= (float)gsz;
        float var = 0.f;
        for (int g = 0; g < gsz; ++g) var += (xv[g] - mean) * (xv[g] - mean);
        const float sd = sqrtf(var / (float)gsz + 1e-8f);
        float svr = 0.f;
        for (int g = 0; g < gsz; ++g) svr += vv[g] * (xv[g] - mean) / sd;
        for (int g = 0; g < gsz; ++g) {
            const int64_t b = (int64_t)(g * cols + m);
            const float r = (xv[g] - mean) / sd;
            Elem<T>::st(dxx + b * chw + e, cm * ((vv[g] - vmean) / sd - r * svr / ((float)gsz * sd)));
            Elem<T>::st(ddy + (b * hw + pixe) * cp + ch, vv[g]);
        }
    }
    for (int64_t q = threadIdx.x; q < (int64_t)gsz * hw * (cp - c); q += 256) {
        const int extra = (int)(q % (cp - c));
        const int64_t gp = q / (cp - c);
        const int g = (int)(gp / hw);
        const int64_t pix = (int64_t)(g * cols + m) * hw + (gp - g * hw);
        Elem<T>::st(ddy + pix * cp + c + extra, extra == 0 ? aext : 0.0f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void l1_sum_kernel(const T* __restrict__ r, const float* __restrict__ t, int64_t n,
                                                     float* __restrict__ out) {
    __shared__ float part[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        acc += fabsf(t[i] - Elem<T>::ld(r + i));
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

// d = s * (a1 * sign(r - t) + a2 * 2 (r - t))      (L1 + L2 reconstruction terms, loss.py:118-121)
template <typename T>
__global__ __launch_bounds__(256) void l1l2_bwd_kernel(const T* __restrict__ r, const float* __restrict__ t, int64_t n,
                                                       float a1, float a2, const float* __restrict__ gs,
                                                       T* __restrict__ d, int accumulate) {
    const float s = gs ? *gs : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float df = Elem<T>::ld(r + i) - t[i];
        float g = s * (a1 * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)) + a2 * 2.0f * df);
        if (accumulate) g += Elem<T>::ld(d + i);
        Elem<T>::st(d + i, g);
    }
}

__device__ __forceinline__ float softplus_f(float v) { return fmaxf(v, 0.f) + log1pf(__expf(-fabsf(v))); }

// loss.py:11-51.  mode 0 hinge, 1 non-saturating.  which 0: generator(fake) ; 1: discriminator(real, fake)
__global__ void gan_loss_kernel(const float* __restrict__ real, const float* __restrict__ fake, int n, int mode, int which,
                                float* __restrict__ loss, float* __restrict__ dreal, float* __restrict__ dfake,
                                const float* __restrict__ gs) {
    __shared__ float part[4];
    const float s = (gs ? *gs : 1.0f) / (float)n;
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float f = fake[i];
        if (which == 0) {
            if (mode == 0) { acc += -f; if (dfake) dfake[i] = -s; }
            else { acc += softplus_f(-f); if (dfake) dfake[i] = -s / (1.0f + __expf(f)); }
        } else {
            const float r = real[i];
            if (mode == 0) {
                acc += fmaxf(1.0f - r, 0.f) + fmaxf(1.0f + f, 0.f);
                if (dreal) dreal[i] = (1.0f - r > 0.f) ? -s : 0.f;
                if (dfake) dfake[i] = (1.0f + f > 0.f) ? s : 0.f;
            } else {
                acc += softplus_f(-r) + softplus_f(f);
                if (dreal) dreal[i] = -s / (1.0f + __expf(r));
                if (dfake) dfake[i] = s / (1.0f + __expf(-f));
            }
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && loss) loss[0] = ((part[0] + part[1]) + (part[2] + part[3])) / (float)n;
}

}  // namespace

#define LAUNCH_T(dtype, K, grid, st, ...)                                                                    \
    do {                                                                                                     \
        if ((dtype) == VQK_F32) hipLaunchKernelGGL((K<float>), grid, dim3(256), 0, st, __VA_ARGS__);         \
        else if ((dtype) == VQK_BF16) hipLaunchKernelGGL((K<bf16_raw>), grid, dim3(256), 0, st, __VA_ARGS__); \
        else return VQK_ERR_DTYPE;                                                                           \
    } while (0)

template <typename T> struct Raw16;                           // a 16-byte channel vector kept raw (bf16 stays packed in LDS)
template <> struct Raw16<float> {
    typedef f32x4 type;
    __device__ static __forceinline__ void unpack(const f32x4& v, float (&o)[4]) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
};
template <> struct Raw16<bf16_raw> {
    typedef u16x8 type;
    __device__ static __forceinline__ void unpack(const u16x8& v, float (&o)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = bf16_to_f32(v[i]);
    }
};

// Fast path of the discriminator's blurs (upfirdn2d.py:214-268 with up = 1, a 4x4 FIR, down in {1, 2}): a thread produces
// FOUR consecutive output pixels of one 16-byte channel slot, so the (3*down + 4) input columns of each filter row are
// loaded once and reused from registers (7-10 loads per output instead of 16, no per-tap integer division).
template <typename T, int DOWN>
__global__ __launch_bounds__(256) void upfirdn_fir4_kernel(const T* __restrict__ x, const float* __restrict__ f,
                                                           T* __restrict__ y, int n, int h, int w, int c, int px0, int py0,
                                                           int flip, float gain, int oh, int ow) {
    constexpr int V = Vec16<T>::N, OUTX = 4, NC = (OUTX - 1) * DOWN + 4;
    __shared__ float fs[16];
    if (threadIdx.x < 16) {
        const int ky = threadIdx.x >> 2, kx = threadIdx.x & 3;
        fs[threadIdx.x] = (flip ? f[ky * 4 + kx] : f[(3 - ky) * 4 + (3 - kx)]) * gain;
    }
    __syncthreads();
    const int vpp = c / V;
    const int owq = (ow + OUTX - 1) / OUTX;
    const int64_t total = (int64_t)n * oh * owq * vpp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        // 32-bit index arithmetic (the launcher guarantees total < 2^31): 64-bit div/mod by run-time values costs more
        // than the 512 FMAs of the iteration
        const unsigned iu = (unsigned)i;
        const unsigned pu = iu / (unsigned)vpp;
        const int v = (int)(iu - pu * (unsigned)vpp);
        const unsigned pv = pu / (unsigned)owq;
        const int oxq = (int)(pu - pv * (unsigned)owq);
        const unsigned pw = pv / (unsigned)oh;
        const int oy = (int)(pv - pw * (unsigned)oh);
        const int img = (int)pw;
        const int ox0 = oxq * OUTX;
        const int ix0 = ox0 * DOWN - px0, iy0 = oy * DOWN - py0;
        float acc[OUTX][V];
#pragma unroll
        for (int o = 0; o < OUTX; ++o)
#pragma unroll
            for (int k = 0; k < V; ++k) acc[o][k] = 0.0f;
        // BRANCH-FREE loads, one filter row per loop trip: a pixel outside the image is read at the clamped coordinate and
        // its filter tap zeroed.  (With `if (inside) load` and the rows unrolled, every one of the 40 loads sat in its own
        // exec-masked block behind its own s_waitcnt -- forty dependent memory round trips per thread, 195 us for 335 MB at
        // 128 ch @256^2; unrolled AND branch-free the compiler hoists all 40 loads: 260 registers, one wave per SIMD.)
#pragma unroll 1
        for (int ky = 0; ky < 4; ++ky) {
            const int iy = iy0 + ky;
            const bool rowin = iy >= 0 && iy < h;
            const T* xrow = x + (((int64_t)img * h + min(max(iy, 0), h - 1)) * w) * c + v * V;
            typename Raw16<T>::type col[NC];                     // raw 16-byte pieces: bf16 stays packed until it is used
            float cz[NC];
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                const int ix = ix0 + q;
                cz[q] = (rowin && ix >= 0 && ix < w) ? 1.0f : 0.0f;
                col[q] = *reinterpret_cast<const typename Raw16<T>::type*>(xrow + (int64_t)min(max(ix, 0), w - 1) * c);
            }
#pragma unroll
            for (int q = 0; q < NC; ++q) {
                float fq[V];
                Raw16<T>::unpack(col[q], fq);
#pragma unroll
                for (int o = 0; o < OUTX; ++o) {
                    const int kx = q - o * DOWN;
                    if (kx >= 0 && kx < 4) {
                        const float fv = fs[ky * 4 + kx] * cz[q];
#pragma unroll
                        for (int k = 0; k < V; ++k) acc[o][k] = __fmaf_rn(fq[k], fv, acc[o][k]);
                    }
                }
            }
        }
        T* yrow = y + (((int64_t)img * oh + oy) * ow + ox0) * c + v * V;
#pragma unroll
        for (int o = 0; o < OUTX; ++o)
            if (ox0 + o < ow) Vec16<T>::store(yrow + (int64_t)o * c, acc[o]);
    }
}


// LDS-tiled form of the discriminator's three resampling filters (upfirdn2d.py:214-268 with a 4x4 FIR): blur (up 1, down 1),
// blur + decimate (down 2) and its adjoint, zero-stuff + blur (up 2).  The register form above re-read every input pixel 4-7x
// from L1/L2 and reached 1.5 TB/s of algorithmic bytes (354 us for 128 ch @256^2, bs 16); here a block stages the input
// footprint of an 8 x 16 output tile x 8 channel slots (16 B each) in LDS once, with zero fill outside the image, and every
// thread produces four consecutive output pixels of one slot from LDS.
// MASK: the result is multiplied by the slope of a relu / leaky-relu taken from the sign of `mask` (an activated tensor of the
// output's shape) -- the backward of "activation, then blur" (discriminator.py:104-120 followed by conv2d_resample.py:119) as ONE
// pass: t = act'(y0) * blur^T(dB) without storing blur^T(dB).
template <typename T, int UP, int DOWN, bool MASK = false>
__global__ __launch_bounds__(256) void upfirdn_tile_kernel(const T* __restrict__ x, const float* __restrict__ f, T* __restrict__ y,
                                                           int n, int h, int w, int c, int px0, int py0, int flip, float gain,
                                                           int oh, int ow, int tiles_x, int tiles_y,
                                                           const T* __restrict__ mask = nullptr, float slope = 1.0f) {
    constexpr int V = Vec16<T>::N, TOH = 8, TOW = 16;
    constexpr int IH = ((TOH - 1) * DOWN + 3) / UP + 2, IW = ((TOW - 1) * DOWN + 3) / UP + 2;     // input rows / columns per tile
    typedef typename Raw16<T>::type raw_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    raw_t* tile = reinterpret_cast<raw_t*>(smem);              // [IH][IW][8 slots]
    __shared__ float fs[16];
    if (threadIdx.x < 16) {
        const int ky = threadIdx.x >> 2, kx = threadIdx.x & 3;
        fs[threadIdx.x] = (flip ? f[ky * 4 + kx] : f[(3 - ky) * 4 + (3 - kx)]) * gain;
    }
    int b = (int)blockIdx.x;
    const int txi = b % tiles_x; b /= tiles_x;
    const int tyi = b % tiles_y; b /= tiles_y;
    const int slices = c / (8 * V);
    const int cs = b % slices, img = b / slices;
    const int oy0 = tyi * TOH, ox0 = txi * TOW;
    // first input row / column of the footprint: floor((o0 * DOWN - p0) / UP)
    const int uy0 = oy0 * DOWN - py0, ux0 = ox0 * DOWN - px0;
    const int iy0 = UP == 1 ? uy0 : (uy0 >> 1), ix0 = UP == 1 ? ux0 : (ux0 >> 1);
    const T* ximg = x + (int64_t)img * h * w * c + cs * 8 * V;
    // fill: every thread's loads are issued back to back (clamped address + select instead of `if (inside) load`, which put
    // each load in its own exec-masked block behind its own s_waitcnt: one memory round trip per loop trip)
    constexpr int TOT = IH * IW * 8, NIT = (TOT + 255) / 256;
    raw_t fill[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = min(it * 256 + (int)threadIdx.x, TOT - 1);
        const int slot = i & 7, pix = i >> 3;
        const int ry = pix / IW, rx = pix - ry * IW;
        const int iy = iy0 + ry, ix = ix0 + rx;
        const bool inside = (unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w;
        const int iyc = min(max(iy, 0), h - 1), ixc = min(max(ix, 0), w - 1);
        const raw_t v = *reinterpret_cast<const raw_t*>(ximg + ((int64_t)iyc * w + ixc) * c + slot * V);
        fill[it] = inside ? v : raw_t{};
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int i = it * 256 + (int)threadIdx.x;
        if (i < TOT) tile[i] = fill[it];
    }
    __syncthreads();
    const int slot = threadIdx.x & 7, oxq = (threadIdx.x >> 3) & 3, oyl = threadIdx.x >> 5;
    const int oy = oy0 + oyl;
    if (oy >= oh) return;
    float acc[4][V];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
        for (int k = 0; k < V; ++k) acc[o][k] = 0.0f;
    const int uy = oy * DOWN - py0;
    // the thread's four outputs read NCOL distinct input columns per filter row: each is fetched from LDS and expanded to
    // fp32 ONCE (the kernel is VALU-bound: 16 taps x 8 channels per output vector).  UP = 2: only the even columns of the
    // zero-stuffed image carry data; the parity of the thread's first column is block-uniform and selects one of two
    // fully unrolled bodies (static register indexing).
    constexpr int NCOL = UP == 1 ? 3 * DOWN + 4 : 4;
    const int ux_first = (ox0 + 4 * oxq) * DOWN - px0;           // U column of output 0, tap 0
    auto body = [&](auto par_c) {
        constexpr int PAR = decltype(par_c)::value;               // UP = 2: ux_first & 1
        // UP = 2: U column ux_first + o + kx = 2a + PAR + o + kx carries input column a + ((o + kx + PAR) >> 1), a = ux_first >> 1
        // (arithmetic shift: floor, also left of the image); q below is the second term
        const int rx_first = (UP == 1 ? ux_first : (ux_first >> 1)) - ix0;
#pragma unroll 1
        for (int ky = 0; ky < 4; ++ky) {                        // (not unrolled: one row of columns in registers at a time)
            const int u = uy + ky;
            if (UP == 2 && (u & 1)) continue;                   // a stuffed zero row
            const int ry = (UP == 1 ? u : (u >> 1)) - iy0;
            const raw_t* trow = tile + ry * IW * 8 + slot;
            float col[NCOL][V];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) Raw16<T>::unpack(trow[(rx_first + q) * 8], col[q]);
#pragma unroll
            for (int o = 0; o < 4; ++o)
#pragma unroll
                for (int kx = 0; kx < 4; ++kx) {
                    if (UP == 2 && (((o + kx) & 1) != PAR)) continue;
                    constexpr int dummy = 0; (void)dummy;
                    const int q = UP == 1 ? o * DOWN + kx : ((o + kx + PAR) >> 1);
                    const float fv = fs[ky * 4 + kx];
#pragma unroll
                    for (int k = 0; k < V; ++k) acc[o][k] = __fmaf_rn(col[q][k], fv, acc[o][k]);
                }
        }
    };
    if (UP == 2 && (ux_first & 1)) body(std::integral_constant<int, 1>{});
    else body(std::integral_constant<int, 0>{});
    const int64_t yoff = (((int64_t)img * oh + oy) * ow + ox0 + 4 * oxq) * c + cs * 8 * V + slot * V;
    T* yrow = y + yoff;
    if constexpr (MASK) {
        float mv[4][V];
#pragma unroll
        for (int o = 0; o < 4; ++o) Vec16<T>::load(mask + yoff + (int64_t)min(o, ow - 1 - (ox0 + 4 * oxq)) * c, mv[o]);   // (clamped: no load beyond the row)
#pragma unroll
        for (int o = 0; o < 4; ++o)
#pragma unroll
            for (int k = 0; k < V; ++k) acc[o][k] = mv[o][k] > 0.0f ? acc[o][k] : slope * acc[o][k];
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
        if (ox0 + 4 * oxq + o < ow) Vec16<T>::store(yrow + (int64_t)o * c, acc[o]);
}

extern "C" {

int vqk_act_backward(int dtype, const void* dy, const void* y, void* dx, int64_t n, int act, float scale, void* stream) {
    VQK_REQUIRE(dy && y && dx, VQK_ERR_ARG);
    VQK_REQUIRE(act >= 0 && act <= 3, VQK_ERR_ARG);
    if (n <= 0) return VQK_OK;
    const int v = dtype == VQK_F32 ? 4 : 8;
    if ((dtype == VQK_F32 || dtype == VQK_BF16) && n % v == 0 && vqk_aligned16(dy) && vqk_aligned16(y) && vqk_aligned16(dx)) {
        const dim3 vgrid(vqk_grid_1d(n / v, 256 * 2));
        if (dtype == VQK_F32) hipLaunchKernelGGL(act_bwd_vec_kernel<float>, vgrid, dim3(256), 0, vqk_stream(stream), (const float*)dy, (const float*)y, (float*)dx, n / v, act, scale);
        else hipLaunchKernelGGL(act_bwd_vec_kernel<bf16_raw>, vgrid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)dy, (const bf16_raw*)y, (bf16_raw*)dx, n / v, act, scale);
        VQK_CHECK_LAUNCH();
        return VQK_OK;
    }
    const dim3 grid(vqk_grid_1d(n, 256 * 4));
    if (dtype == VQK_F32) hipLaunchKernelGGL(act_bwd_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), (const float*)dy, (const float*)y, (float*)dx, n, act, scale);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(act_bwd_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)dy, (const bf16_raw*)y, (bf16_raw*)dx, n, act, scale);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_act_backward_colsum_scaled(int dtype, const void* dy, const void* y, void* dx, int64_t rows, int c, int act, float scale,
                                   float colsum_scale, float* colsum, void* stream);
int vqk_act_backward_colsum(int dtype, const void* dy, const void* y, void* dx, int64_t rows, int c, int act, float scale,
                            float* colsum, void* stream) {
    return vqk_act_backward_colsum_scaled(dtype, dy, y, dx, rows, c, act, scale, 1.0f, colsum, stream);
}

int vqk_act_backward_colsum_scaled(int dtype, const void* dy, const void* y, void* dx, int64_t rows, int c, int act, float scale,
                                   float colsum_scale, float* colsum, void* stream) {
    VQK_REQUIRE(dy && y && dx && colsum, VQK_ERR_ARG);
    VQK_REQUIRE(act >= 0 && act <= 3 && rows >= 0 && c > 0, VQK_ERR_ARG);
    VQK_REQUIRE(dtype == VQK_F32 || dtype == VQK_BF16, VQK_ERR_DTYPE);
    const int v = dtype == VQK_F32 ? 4 : 8;
    VQK_REQUIRE(c % v == 0 && c / v <= 256 && c <= 8192, VQK_ERR_SHAPE);
    VQK_REQUIRE(vqk_aligned16(dy) && vqk_aligned16(y) && vqk_aligned16(dx), VQK_ERR_ALIGN);
    if (rows == 0) return VQK_OK;
    int64_t blocks = (rows + 63) / 64; if (blocks > 2048) blocks = 2048;
    const int64_t rpb = (rows + blocks - 1) / blocks;
    blocks = (rows + rpb - 1) / rpb;
    const size_t lds = (size_t)c * 4;
    hipStream_t st = vqk_stream(stream);
    if (dtype == VQK_F32) hipLaunchKernelGGL(act_bwd_colsum_kernel<float>, dim3((unsigned)blocks), dim3(256), lds, st, (const float*)dy, (const float*)y, (float*)dx, rows, c, rpb, act, scale, colsum, colsum_scale);
    else hipLaunchKernelGGL(act_bwd_colsum_kernel<bf16_raw>, dim3((unsigned)blocks), dim3(256), lds, st, (const bf16_raw*)dy, (const bf16_raw*)y, (bf16_raw*)dx, rows, c, rpb, act, scale, colsum, colsum_scale);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_upfirdn2d_nhwc(int dtype, const void* x, const float* f, void* y, int n, int h, int w, int c, int fh, int fw, int upx,
                       int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                       int out_h, int out_w, void* stream) {
    VQK_REQUIRE(x && f && y, VQK_ERR_ARG);
    const int v = dtype == VQK_F32 ? 4 : 8;
    VQK_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % v == 0 && fh >= 1 && fw >= 1, VQK_ERR_SHAPE);
    VQK_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, VQK_ERR_ARG);
    VQK_REQUIRE(out_w == (w * upx + padx0 + padx1 - fw + downx) / downx, VQK_ERR_SHAPE);
    VQK_REQUIRE(out_h == (h * upy + pady0 + pady1 - fh + downy) / downy, VQK_ERR_SHAPE);
    VQK_REQUIRE(out_w >= 1 && out_h >= 1, VQK_ERR_SHAPE);
    // LDS-tiled form: 4x4 FIR, (up, down) in {(1,1), (1,2), (2,1)}, whole groups of 8 channel slots
    const int tile_on = VQK_TUNE("UPFIRDN_TILE", 1);
    if (tile_on && fh == 4 && fw == 4 && upx == upy && downx == downy && c % (8 * v) == 0 &&
        ((upx == 1 && downx == 1) || (upx == 2 && downx == 1))) {     // (down 2: the register form below is faster, 189 vs 249 us)
        const int tiles_x = (out_w + 15) / 16, tiles_y = (out_h + 7) / 8;
        const int64_t blocks = (int64_t)n * tiles_y * tiles_x * (c / (8 * v));
        if (blocks < 0x7fffffff) {
            hipStream_t st = vqk_stream(stream);
            const dim3 g((unsigned)blocks);
#define VQK_UFT(T, U, D) do { constexpr int ih = (7 * D + 3) / U + 2, iw = (15 * D + 3) / U + 2; \
                hipLaunchKernelGGL((upfirdn_tile_kernel<T, U, D>), g, dim3(256), (size_t)ih * iw * 8 * 16, st, (const T*)x, f, (T*)y, n, h, w, c, \
                                   padx0, pady0, flip, gain, out_h, out_w, tiles_x, tiles_y); } while (0)
            if (dtype == VQK_F32) { if (upx == 2) VQK_UFT(float, 2, 1); else if (downx == 2) VQK_UFT(float, 1, 2); else VQK_UFT(float, 1, 1); }
            else { if (upx == 2) VQK_UFT(bf16_raw, 2, 1); else if (downx == 2) VQK_UFT(bf16_raw, 1, 2); else VQK_UFT(bf16_raw, 1, 1); }
#undef VQK_UFT
            VQK_CHECK_LAUNCH();
            return VQK_OK;
        }
    }
    const int64_t tot4 = (int64_t)n * out_h * ((out_w + 3) / 4) * (c / v);
    if (upx == 1 && upy == 1 && fh == 4 && fw == 4 && downx == downy && (downx == 1 || downx == 2) && (dtype == VQK_F32 || dtype == VQK_BF16) &&
        tot4 < 0x7fffffff) {
        const dim3 g4(vqk_grid_1d(tot4, 256, 256 * 64));
        hipStream_t st = vqk_stream(stream);
        if (dtype == VQK_F32) {
            if (downx == 1) hipLaunchKernelGGL((upfirdn_fir4_kernel<float, 1>), g4, dim3(256), 0, st, (const float*)x, f, (float*)y, n, h, w, c, padx0, pady0, flip, gain, out_h, out_w);
            else hipLaunchKernelGGL((upfirdn_fir4_kernel<float, 2>), g4, dim3(256), 0, st, (const float*)x, f, (float*)y, n, h, w, c, padx0, pady0, flip, gain, out_h, out_w);
        } else {
            if (downx == 1) hipLaunchKernelGGL((upfirdn_fir4_kernel<bf16_raw, 1>), g4, dim3(256), 0, st, (const bf16_raw*)x, f, (bf16_raw*)y, n, h, w, c, padx0, pady0, flip, gain, out_h, out_w);
            else hipLaunchKernelGGL((upfirdn_fir4_kernel<bf16_raw, 2>), g4, dim3(256), 0, st, (const bf16_raw*)x, f, (bf16_raw*)y, n, h, w, c, padx0, pady0, flip, gain, out_h, out_w);
        }
        VQK_CHECK_LAUNCH();
        return VQK_OK;
    }
    const int64_t total = (int64_t)n * out_h * out_w * (c / v);
    const dim3 grid(vqk_grid_1d(total, 256, 256 * 16));
    if (dtype == VQK_F32) hipLaunchKernelGGL(upfirdn_nhwc_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), (const float*)x, f, (float*)y, n, h, w, c, fh, fw, upx, upy, downx, downy, padx0, pady0, flip, gain, out_h, out_w);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(upfirdn_nhwc_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)x, f, (bf16_raw*)y, n, h, w, c, fh, fw, upx, upy, downx, downy, padx0, pady0, flip, gain, out_h, out_w);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_upfirdn2d_act_backward(int dtype, const void* x, const float* f, const void* y_act, void* out, int n, int h, int w, int c,
                                int padx0, int padx1, int pady0, int pady1, int flip, float gain, int act, int out_h, int out_w,
                                void* stream) {
    VQK_REQUIRE(x && f && y_act && out, VQK_ERR_ARG);
    VQK_REQUIRE(dtype == VQK_F32 || dtype == VQK_BF16, VQK_ERR_DTYPE);
    VQK_REQUIRE(act == 2 || act == 3, VQK_ERR_ARG);
    const int v = dtype == VQK_F32 ? 4 : 8;
    VQK_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % (8 * v) == 0, VQK_ERR_SHAPE);
    VQK_REQUIRE(out_w == w + padx0 + padx1 - 3 && out_h == h + pady0 + pady1 - 3 && out_w >= 1 && out_h >= 1, VQK_ERR_SHAPE);
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(y_act) && vqk_aligned16(out), VQK_ERR_ALIGN);
    const int tiles_x = (out_w + 15) / 16, tiles_y = (out_h + 7) / 8;
    const int64_t blocks = (int64_t)n * tiles_y * tiles_x * (c / (8 * v));
    VQK_REQUIRE(blocks < 0x7fffffff, VQK_ERR_SHAPE);
    const float slope = act == 3 ? 0.2f : 0.0f;
    hipStream_t st = vqk_stream(stream);
    constexpr int ih = 7 + 3 + 2, iw = 15 + 3 + 2;
    if (dtype == VQK_F32)
        hipLaunchKernelGGL((upfirdn_tile_kernel<float, 1, 1, true>), dim3((unsigned)blocks), dim3(256), (size_t)ih * iw * 8 * 16, st, (const float*)x, f,
                           (float*)out, n, h, w, c, padx0, pady0, flip, gain, out_h, out_w, tiles_x, tiles_y, (const float*)y_act, slope);
    else
        hipLaunchKernelGGL((upfirdn_tile_kernel<bf16_raw, 1, 1, true>), dim3((unsigned)blocks), dim3(256), (size_t)ih * iw * 8 * 16, st, (const bf16_raw*)x, f,
                           (bf16_raw*)out, n, h, w, c, padx0, pady0, flip, gain, out_h, out_w, tiles_x, tiles_y, (const bf16_raw*)y_act, slope);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_maxpool2x2(int dtype, const void* x, const void* dy, void* out, int n, int h, int w, int c, int backward, void* stream) {
    VQK_REQUIRE(x && out && (!backward || dy), VQK_ERR_ARG);
    const int v = dtype == VQK_F32 ? 4 : 8;
    VQK_REQUIRE(n > 0 && h > 0 && w > 0 && !(h & 1) && !(w & 1) && c > 0 && c % v == 0, VQK_ERR_SHAPE);
    const int64_t total = (int64_t)n * (h / 2) * (w / 2) * (c / v);
    const dim3 grid(vqk_grid_1d(total, 256, 256 * 16));
    hipStream_t st = vqk_stream(stream);
    if (dtype == VQK_F32) {
        if (backward) hipLaunchKernelGGL((maxpool_kernel<float, true>), grid, dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)out, n, h, w, c);
        else hipLaunchKernelGGL((maxpool_kernel<float, false>), grid, dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)out, n, h, w, c);
    } else if (dtype == VQK_BF16) {
        if (backward) hipLaunchKernelGGL((maxpool_kernel<bf16_raw, true>), grid, dim3(256), 0, st, (const bf16_raw*)x, (const bf16_raw*)dy, (bf16_raw*)out, n, h, w, c);
        else hipLaunchKernelGGL((maxpool_kernel<bf16_raw, false>), grid, dim3(256), 0, st, (const bf16_raw*)x, (const bf16_raw*)dy, (bf16_raw*)out, n, h, w, c);
    } else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_channel_affine(int dtype, const void* x, const float* scale, const float* shift, void* y, int64_t npix, int c, void* stream) {
    VQK_REQUIRE(x && scale && y, VQK_ERR_ARG);
    VQK_REQUIRE(npix >= 0 && c > 0, VQK_ERR_SHAPE);
    if (npix == 0) return VQK_OK;
    const dim3 grid(vqk_grid_1d(npix * c, 256 * 4));
    if (dtype == VQK_F32) hipLaunchKernelGGL(channel_affine_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), (const float*)x, scale, shift, (float*)y, npix, c);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(channel_affine_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)x, scale, shift, (bf16_raw*)y, npix, c);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_lpips_tap(int dtype, const void* fx, const void* fy, const float* lin, int n, int64_t hw, int c, float* out,
                  const float* gout, float gscale, void* dfy, void* stream) {
    VQK_REQUIRE(fx && fy && lin && (out || dfy), VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && hw > 0 && c > 0, VQK_ERR_SHAPE);
    const int64_t npix = (int64_t)n * hw;
    const dim3 grid(vqk_grid_1d(npix, 4));
    hipStream_t st = vqk_stream(stream);
    // vectorised forms: a lane owns a 16-byte channel slot (c / V lanes per pixel); the grid is sized so that a wave makes
    // about eight passes (a grid of npix / 1024 blocks left the 512-channel taps on 4-16 blocks: 513 us for 8 MB)
    const int v = dtype == VQK_F32 ? 4 : 8;
    const int lpp = (c % v) == 0 ? c / v : 0;
    const bool vec = lpp >= 1 && lpp <= 64 && (lpp & (lpp - 1)) == 0 && vqk_aligned16(fx) && vqk_aligned16(fy) &&
                     (!dfy || vqk_aligned16(dfy));
    const int64_t per_pass = vec ? 4 * (64 / lpp) : 0;                       // pixels one block handles per loop trip
    int64_t ppb = per_pass * 8;
    if (vec && (npix + ppb - 1) / ppb > 8192) ppb = ((npix + 8191) / 8192 + per_pass - 1) / per_pass * per_pass;
    const dim3 vgrid(vec ? (unsigned)((npix + ppb - 1) / ppb) : 1u);
    if (dfy) {
        if (vec && dtype == VQK_F32) hipLaunchKernelGGL(lpips_tap_bwd_vec_kernel<float>, vgrid, dim3(256), 0, st, (const float*)fx, (const float*)fy, lin, gout, gscale, npix, hw, c, ppb, (float*)dfy);
        else if (vec && dtype == VQK_BF16) hipLaunchKernelGGL(lpips_tap_bwd_vec_kernel<bf16_raw>, vgrid, dim3(256), 0, st, (const bf16_raw*)fx, (const bf16_raw*)fy, lin, gout, gscale, npix, hw, c, ppb, (bf16_raw*)dfy);
        else if (dtype == VQK_F32) hipLaunchKernelGGL(lpips_tap_bwd_kernel<float>, grid, dim3(256), 0, st, (const float*)fx, (const float*)fy, lin, gout, gscale, npix, hw, c, (float*)dfy);
        else if (dtype == VQK_BF16) hipLaunchKernelGGL(lpips_tap_bwd_kernel<bf16_raw>, grid, dim3(256), 0, st, (const bf16_raw*)fx, (const bf16_raw*)fy, lin, gout, gscale, npix, hw, c, (bf16_raw*)dfy);
        else return VQK_ERR_DTYPE;
    } else {
        if (vec && dtype == VQK_F32) hipLaunchKernelGGL(lpips_tap_fwd_vec_kernel<float>, vgrid, dim3(256), 0, st, (const float*)fx, (const float*)fy, lin, npix, hw, c, ppb, out);
        else if (vec && dtype == VQK_BF16) hipLaunchKernelGGL(lpips_tap_fwd_vec_kernel<bf16_raw>, vgrid, dim3(256), 0, st, (const bf16_raw*)fx, (const bf16_raw*)fy, lin, npix, hw, c, ppb, out);
        else if (dtype == VQK_F32) hipLaunchKernelGGL(lpips_tap_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)fx, (const float*)fy, lin, npix, hw, c, out);
        else if (dtype == VQK_BF16) hipLaunchKernelGGL(lpips_tap_fwd_kernel<bf16_raw>, grid, dim3(256), 0, st, (const bf16_raw*)fx, (const bf16_raw*)fy, lin, npix, hw, c, out);
        else return VQK_ERR_DTYPE;
    }
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_mbstd(int dtype, const void* x, const void* dy, void* out, float* stat, int n, int64_t hw, int c, int cpad, int group,
              int backward, void* stream) {
    VQK_REQUIRE(x && out && (backward ? dy != nullptr : stat != nullptr), VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && hw > 0 && c > 0 && cpad > c && group >= 1 && group <= 8 && n % group == 0, VQK_ERR_SHAPE);
    VQK_REQUIRE(hw * c < 0x7fffffff, VQK_ERR_SHAPE);
    const int cols = n / group;
    const int mthreads = hw * c >= 4096 ? 1024 : 256;             // one block per group column
    hipStream_t st = vqk_stream(stream);
    if (!backward) {
        if (dtype == VQK_F32) {
            hipLaunchKernelGGL(mbstd_stat_kernel<float>, dim3(cols), dim3(mthreads), 0, st, (const float*)x, n, hw * c, group, stat);
            hipLaunchKernelGGL(mbstd_concat_kernel<float>, dim3(vqk_grid_1d((int64_t)n * hw * cpad, 256)), dim3(256), 0, st, (const float*)x, stat, (float*)out, n, hw, c, cpad, cols);
        } else if (dtype == VQK_BF16) {
            hipLaunchKernelGGL(mbstd_stat_kernel<bf16_raw>, dim3(cols), dim3(mthreads), 0, st, (const bf16_raw*)x, n, hw * c, group, stat);
            hipLaunchKernelGGL(mbstd_concat_kernel<bf16_raw>, dim3(vqk_grid_1d((int64_t)n * hw * cpad, 256)), dim3(256), 0, st, (const bf16_raw*)x, stat, (bf16_raw*)out, n, hw, c, cpad, cols);
        } else return VQK_ERR_DTYPE;
    } else {
        if (dtype == VQK_F32) hipLaunchKernelGGL(mbstd_bwd_kernel<float>, dim3(cols), dim3(mthreads), 0, st, (const float*)x, (const float*)dy, (float*)out, n, hw, c, cpad, group);
        else if (dtype == VQK_BF16) hipLaunchKernelGGL(mbstd_bwd_kernel<bf16_raw>, dim3(cols), dim3(mthreads), 0, st, (const bf16_raw*)x, (const bf16_raw*)dy, (bf16_raw*)out, n, hw, c, cpad, group);
        else return VQK_ERR_DTYPE;
    }
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_mbstd_double_backward(int dtype, const void* x, const void* dy, const void* v, void* ddy, void* dxx, int n,
                               int64_t hw, int c, int cpad, int group, void* stream) {
    VQK_REQUIRE(x && dy && v && ddy && dxx, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && hw > 0 && c > 0 && cpad > c && group >= 1 && group <= 8 && n % group == 0, VQK_ERR_SHAPE);
    const int cols = n / group;
    hipStream_t st = vqk_stream(stream);
    if (dtype == VQK_F32) hipLaunchKernelGGL(mbstd_bwd_bwd_kernel<float>, dim3(cols), dim3(256), 0, st, (const float*)x, (const float*)dy, (const float*)v, (float*)ddy, (float*)dxx, n, hw, c, cpad, group);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(mbstd_bwd_bwd_kernel<bf16_raw>, dim3(cols), dim3(256), 0, st, (const bf16_raw*)x, (const bf16_raw*)dy, (const bf16_raw*)v, (bf16_raw*)ddy, (bf16_raw*)dxx, n, hw, c, cpad, group);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_l1_sum(int dtype, const void* recon, const float* target, int64_t n, float* out, void* stream) {
    VQK_REQUIRE(recon && target && out, VQK_ERR_ARG);
    if (n <= 0) return VQK_OK;
    const dim3 grid(vqk_grid_1d(n, 256 * 8));
    if (dtype == VQK_F32) hipLaunchKernelGGL(l1_sum_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), (const float*)recon, target, n, out);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(l1_sum_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)recon, target, n, out);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_l1l2_backward(int dtype, const void* recon, const float* target, int64_t n, float a1, float a2, const float* gscale_dev,
                      void* d, int accumulate, void* stream) {
    VQK_REQUIRE(recon && target && d, VQK_ERR_ARG);
    if (n <= 0) return VQK_OK;
    const dim3 grid(vqk_grid_1d(n, 256 * 4));
    if (dtype == VQK_F32) hipLaunchKernelGGL(l1l2_bwd_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), (const float*)recon, target, n, a1, a2, gscale_dev, (float*)d, accumulate);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(l1l2_bwd_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)recon, target, n, a1, a2, gscale_dev, (bf16_raw*)d, accumulate);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_gan_loss(const float* logits_real, const float* logits_fake, int n, int mode, int which, float* loss, float* dreal,
                 float* dfake, const float* gscale_dev, void* stream) {
    VQK_REQUIRE(logits_fake && (which == 0 || logits_real), VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && (mode == 0 || mode == 1) && (which == 0 || which == 1), VQK_ERR_ARG);
    hipLaunchKernelGGL(gan_loss_kernel, dim3(1), dim3(256), 0, vqk_stream(stream), logits_real, logits_fake, n, mode, which, loss,
                       dreal, dfake, gscale_dev);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

}  // extern "C"
