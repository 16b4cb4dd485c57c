// Memory-bound pointwise / pooling kernels of the train step (NHWC, 16 bytes per lane):
//   avg-pool 2x2 and its backward       vqvae/modules/autoencoder.py:89-91
//   backward of the nearest x2 upsample  vqvae/modules/autoencoder.py:104-106
//   clamp + normalise + NCHW->NHWC       vqvae/modules/abstract_modules/base_autoencoder.py:31-50
//   MSE loss and its backward through the decoder's tanh   vqvae/model.py:272, autoencoder.py:179
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void pool2x2_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w,
                                                      int c, float scale) {
    constexpr int V = Vec16<T>::N;
    const int oh = h >> 1, ow = w >> 1, vpp = c / V;
    const int64_t total = (int64_t)n * oh * ow * vpp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpp);
        int64_t p = i / vpp;
        const int ox = (int)(p % ow); p /= ow;
        const int oy = (int)(p % oh);
        const int img = (int)(p / oh);
        const T* s = x + (((int64_t)img * h + 2 * oy) * w + 2 * ox) * c + v * V;
        float a[V], b[V], d[V], e[V], o[V];
        Vec16<T>::load(s, a); Vec16<T>::load(s + c, b);
        Vec16<T>::load(s + (int64_t)w * c, d); Vec16<T>::load(s + (int64_t)w * c + c, e);
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = ((a[k] + b[k]) + (d[k] + e[k])) * scale;
        Vec16<T>::store(y + i * V, o);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void unpool2x2_kernel(const T* __restrict__ x, T* __restrict__ y, int n, int h, int w,
                                                        int c, float scale) {
    constexpr int V = Vec16<T>::N;
    const int oh = h * 2, ow = w * 2, vpp = c / V;
    const int64_t total = (int64_t)n * oh * ow * vpp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % vpp);
        int64_t p = i / vpp;
        const int ox = (int)(p % ow); p /= ow;
        const int oy = (int)(p % oh);
        const int img = (int)(p / oh);
        float a[V];
        Vec16<T>::load(x + (((int64_t)img * h + (oy >> 1)) * w + (ox >> 1)) * c + v * V, a);
#pragma unroll
        for (int k = 0; k < V; ++k) a[k] *= scale;
        Vec16<T>::store(y + i * V, a);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void preprocess_kernel(const float* __restrict__ img, T* __restrict__ xp,
                                                         float* __restrict__ target, int n, int h, int w, int cpad) {
    const int64_t hw = (int64_t)h * w, total = (int64_t)n * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / hw, p = i - b * hw;
        for (int ch = 0; ch < cpad; ++ch) {
            float v = 0.0f;
            if (ch < 3) {
                v = img[(b * 3 + ch) * hw + p];
                v = fminf(fmaxf(v, 0.0f), 1.0f);
                v = (v - 0.5f) / 0.5f;
            }
            Elem<T>::st(xp + i * cpad + ch, v);
            if (target) target[i * cpad + ch] = v;
        }
    }
}

// preprocess with the training augmentation of base_autoencoder.py:20-22,44-48 fused in: per-sample random resized crop
// (box[b] = {x0, y0, w, h} in source pixels, resampled to the full H x W with bilinear interpolation, corner-aligned)
// and horizontal flip (flip[b] != 0), then clamp / normalise / NHWC pad exactly as preprocess_kernel.
template <typename T>
__global__ __launch_bounds__(256) void augment_preprocess_kernel(const float* __restrict__ img, const float* __restrict__ box,
                                                                 const int32_t* __restrict__ flip, T* __restrict__ xp,
                                                                 float* __restrict__ target, int n, int h, int w, int cpad) {
    const int64_t hw = (int64_t)h * w, total = (int64_t)n * hw;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / hw, p = i - b * hw;
        const int oy = (int)(p / w);
        int ox = (int)(p - (int64_t)oy * w);
        if (flip[b]) ox = w - 1 - ox;
        const float x0 = box[b * 4 + 0], y0 = box[b * 4 + 1], bw = box[b * 4 + 2], bh = box[b * 4 + 3];
        float sx = x0 + (w > 1 ? (float)ox * (bw - 1.0f) / (float)(w - 1) : 0.0f);
        float sy = y0 + (h > 1 ? (float)oy * (bh - 1.0f) / (float)(h - 1) : 0.0f);
        sx = fminf(fmaxf(sx, 0.0f), (float)(w - 1));
        sy = fminf(fmaxf(sy, 0.0f), (float)(h - 1));
        const int ix = (int)sx, iy = (int)sy;                    // <= w-1 / h-1 after the clamp; weight 0 on the far tap there
        const float fx = sx - (float)ix, fy = sy - (float)iy;
        const int ix1 = min(ix + 1, w - 1), iy1 = min(iy + 1, h - 1);
        for (int ch = 0; ch < cpad; ++ch) {
            float v = 0.0f;
            if (ch < 3) {
                const float* pl = img + (b * 3 + ch) * hw;
                const float c00 = fminf(fmaxf(pl[(int64_t)iy * w + ix], 0.f), 1.f), c01 = fminf(fmaxf(pl[(int64_t)iy * w + ix1], 0.f), 1.f);
                const float c10 = fminf(fmaxf(pl[(int64_t)iy1 * w + ix], 0.f), 1.f), c11 = fminf(fmaxf(pl[(int64_t)iy1 * w + ix1], 0.f), 1.f);
                const float top = c00 + fx * (c01 - c00), bot = c10 + fx * (c11 - c10);
                v = top + fy * (bot - top);
                v = (v - 0.5f) / 0.5f;
            }
            Elem<T>::st(xp + i * cpad + ch, v);
            if (target) target[i * cpad + ch] = v;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void sse_kernel(const T* __restrict__ r, const float* __restrict__ t, int64_t n,
                                                  float* __restrict__ loss) {
    __shared__ float part[4];
    float acc = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float d = Elem<T>::ld(r + i) - t[i];
        acc = __fmaf_rn(d, d, acc);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (part[0] + part[1]) + (part[2] + part[3]));
}

template <typename T>
__global__ __launch_bounds__(256) void mse_tanh_bwd_kernel(const T* __restrict__ r, const float* __restrict__ t,
                                                           int64_t n, float gscale, const float* __restrict__ gs,
                                                           int through_tanh, T* __restrict__ d) {
    if (gs) gscale *= *gs;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float y = Elem<T>::ld(r + i);
        float g = gscale * 2.0f * (y - t[i]);
        if (through_tanh) g *= 1.0f - y * y;
        Elem<T>::st(d + i, g);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                       int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float yv = Elem<T>::ld(y + i);
        Elem<T>::st(dx + i, Elem<T>::ld(dy + i) * (1.0f - yv * yv));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(const T* __restrict__ x, const T* __restrict__ y2, T* __restrict__ y,
                                                    float a, float b, int64_t nvec) {
    constexpr int V = Vec16<T>::N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float u[V], w[V];
        Vec16<T>::load(x + i * V, u);
        if (y2) {
            Vec16<T>::load(y2 + i * V, w);
#pragma unroll
            for (int k = 0; k < V; ++k) u[k] = a * u[k] + b * w[k];
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) u[k] = a * u[k];
        }
        Vec16<T>::store(y + i * V, u);
    }
}


// ------------------------------------------------------------------------------------------------
// Evaluation metrics of the test loop (vqvae/model.py:491-553: torchmetrics MSE / PSNR / SSIM on [0,1] NCHW fp32 images).
// pair_stats: out[0] += sum (p - t)^2, out[1] = min t, out[2] = max t, out[3] = min p, out[4] = max p (out[1..4] must be
// initialised to +inf / -inf / +inf / -inf).  ssim_sum: the torchmetrics SSIM map (Gaussian window, valid region) summed
// per image; data_range = max(p range, t range) of THIS batch is read from a pair_stats record on the device.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_min_f(float* addr, float v) {
    unsigned* a = reinterpret_cast<unsigned*>(addr);
    unsigned old = *a;
    while (__uint_as_float(old) > v) {
        const unsigned prev = atomicCAS(a, old, __float_as_uint(v));
        if (prev == old) break;
        old = prev;
    }
}
__device__ __forceinline__ void atomic_max_f(float* addr, float v) {
    unsigned* a = reinterpret_cast<unsigned*>(addr);
    unsigned old = *a;
    while (__uint_as_float(old) < v) {
        const unsigned prev = atomicCAS(a, old, __float_as_uint(v));
        if (prev == old) break;
        old = prev;
    }
}

__global__ __launch_bounds__(256) void pair_stats_kernel(const float* __restrict__ p, const float* __restrict__ t, int64_t n,
                                                         float* __restrict__ out) {
    __shared__ float part[5][4];
    float sse = 0.f, tmin = INFINITY, tmax = -INFINITY, pmin = INFINITY, pmax = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float a = p[i], b = t[i], d = a - b;
        sse = __fmaf_rn(d, d, sse);
        tmin = fminf(tmin, b); tmax = fmaxf(tmax, b); pmin = fminf(pmin, a); pmax = fmaxf(pmax, a);
    }
    sse = wave_sum(sse);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        tmin = fminf(tmin, __shfl_xor(tmin, off, 64)); tmax = fmaxf(tmax, __shfl_xor(tmax, off, 64));
        pmin = fminf(pmin, __shfl_xor(pmin, off, 64)); pmax = fmaxf(pmax, __shfl_xor(pmax, off, 64));
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[0][wv] = sse; part[1][wv] = tmin; part[2][wv] = tmax; part[3][wv] = pmin; part[4][wv] = pmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(out, (part[0][0] + part[0][1]) + (part[0][2] + part[0][3]));
        atomic_min_f(out + 1, fminf(fminf(part[1][0], part[1][1]), fminf(part[1][2], part[1][3])));
        atomic_max_f(out + 2, fmaxf(fmaxf(part[2][0], part[2][1]), fmaxf(part[2][2], part[2][3])));
        atomic_min_f(out + 3, fminf(fminf(part[3][0], part[3][1]), fminf(part[3][2], part[3][3])));
        atomic_max_f(out + 4, fmaxf(fmaxf(part[4][0], part[4][1]), fmaxf(part[4][2], part[4][3])));
    }
}

// block = 16 x 16 outputs of one (image, channel) plane; the (16 + ks - 1)^2 input patches of p and t sit in LDS
template <int KS>
__global__ __launch_bounds__(256) void ssim_sum_kernel(const float* __restrict__ p, const float* __restrict__ t, int c, int h,
                                                       int w, const float* __restrict__ win, const float* __restrict__ stats,
                                                       float k1, float k2, float* __restrict__ out) {
    constexpr int T = 16, R = T + KS - 1;
    __shared__ float sp[R][R + 1], stt[R][R + 1], sw[KS * KS], part[4];
    const int oh = h - KS + 1, ow = w - KS + 1;
    const int plane = blockIdx.z, img = plane / c;
    const int ox0 = blockIdx.x * T, oy0 = blockIdx.y * T;
    const float* pp = p + (int64_t)plane * h * w;
    const float* tp = t + (int64_t)plane * h * w;
    for (int i = threadIdx.x; i < R * R; i += 256) {
        const int y = i / R, x = i - y * R, gy = min(oy0 + y, h - 1), gx = min(ox0 + x, w - 1);
        sp[y][x] = pp[(int64_t)gy * w + gx];
        stt[y][x] = tp[(int64_t)gy * w + gx];
    }
    for (int i = threadIdx.x; i < KS * KS; i += 256) sw[i] = win[i];
    __syncthreads();
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float mp = 0.f, mt = 0.f, epp = 0.f, ett = 0.f, ept = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < KS; ++ky)
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            const float wv = sw[ky * KS + kx], a = sp[ty + ky][tx + kx], b = stt[ty + ky][tx + kx];
            mp = __fmaf_rn(wv, a, mp); mt = __fmaf_rn(wv, b, mt);
            epp = __fmaf_rn(wv, a * a, epp); ett = __fmaf_rn(wv, b * b, ett); ept = __fmaf_rn(wv, a * b, ept);
        }
    const float range = fmaxf(stats[4] - stats[3], stats[2] - stats[1]);
    const float c1 = (k1 * range) * (k1 * range), c2 = (k2 * range) * (k2 * range);
    float v = 0.f;
    if (ox0 + tx < ow && oy0 + ty < oh) {
        const float mpp = mp * mp, mtt = mt * mt, mpt = mp * mt;
        const float spp = epp - mpp, stt_ = ett - mtt, spt = ept - mpt;
        v = ((2.f * mpt + c1) * (2.f * spt + c2)) / ((mpp + mtt + c1) * (spp + stt_ + c2));
    }
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + img, (part[0] + part[1]) + (part[2] + part[3]));
}

}  // namespace

#define DISPATCH_T(dtype, KERNEL, grid, lds, st, ...)                                                      \
    do {                                                                                                   \
        if ((dtype) == VQK_F32) hipLaunchKernelGGL(KERNEL<float>, grid, dim3(256), lds, st, __VA_ARGS__);   \
        else if ((dtype) == VQK_BF16) hipLaunchKernelGGL(KERNEL<bf16_raw>, grid, dim3(256), lds, st, __VA_ARGS__); \
        else return VQK_ERR_DTYPE;                                                                         \
    } while (0)

extern "C" {

int vqk_pool2x2(int dtype, const void* x, void* y, int n, int h, int w, int c, float scale, void* stream) {
    VQK_REQUIRE(x && y, VQK_ERR_ARG);
    const int v = dtype == VQK_F32 ? 4 : 8;
    VQK_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && !(h & 1) && !(w & 1) && c % v == 0, VQK_ERR_SHAPE);
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(y), VQK_ERR_ALIGN);
    const int64_t total = (int64_t)n * (h / 2) * (w / 2) * (c / v);
    const dim3 grid(vqk_grid_1d(total, 256, 256 * 16));
    if (dtype == VQK_F32) hipLaunchKernelGGL(pool2x2_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), (const float*)x, (float*)y, n, h, w, c, scale);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(pool2x2_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)x, (bf16_raw*)y, n, h, w, c, scale);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_unpool2x2(int dtype, const void* x, void* y, int n, int h, int w, int c, float scale, void* stream) {
    VQK_REQUIRE(x && y, VQK_ERR_ARG);
    const int v = dtype == VQK_F32 ? 4 : 8;
    VQK_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && c % v == 0, VQK_ERR_SHAPE);
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(y), VQK_ERR_ALIGN);
    const int64_t total = (int64_t)n * (h * 2) * (w * 2) * (c / v);
    const dim3 grid(vqk_grid_1d(total, 256, 256 * 16));
    if (dtype == VQK_F32) hipLaunchKernelGGL(unpool2x2_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), (const float*)x, (float*)y, n, h, w, c, scale);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(unpool2x2_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)x, (bf16_raw*)y, n, h, w, c, scale);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_preprocess(const float* images, void* x_pad, int dtype, float* target, int n, int h, int w, int cpad, void* stream) {
    VQK_REQUIRE(images && x_pad, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && h > 0 && w > 0 && cpad >= 3, VQK_ERR_SHAPE);
    const dim3 grid(vqk_grid_1d((int64_t)n * h * w, 256, 256 * 16));
    if (dtype == VQK_F32) hipLaunchKernelGGL(preprocess_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), images, (float*)x_pad, target, n, h, w, cpad);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(preprocess_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), images, (bf16_raw*)x_pad, target, n, h, w, cpad);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_augment_preprocess(const float* images, const float* box, const int32_t* flip, void* x_pad, int dtype, float* target,
                           int n, int h, int w, int cpad, void* stream) {
    VQK_REQUIRE(images && box && flip && x_pad, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && h > 0 && w > 0 && cpad >= 3, VQK_ERR_SHAPE);
    const dim3 grid(vqk_grid_1d((int64_t)n * h * w, 256, 256 * 16));
    if (dtype == VQK_F32) hipLaunchKernelGGL(augment_preprocess_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), images, box, flip, (float*)x_pad, target, n, h, w, cpad);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(augment_preprocess_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), images, box, flip, (bf16_raw*)x_pad, target, n, h, w, cpad);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_sse(int dtype, const void* recon, const float* target, int64_t n, float* loss, void* stream) {
    VQK_REQUIRE(recon && target && loss, VQK_ERR_ARG);
    if (n <= 0) return VQK_OK;
    const dim3 grid(vqk_grid_1d(n, 256 * 8));
    if (dtype == VQK_F32) hipLaunchKernelGGL(sse_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), (const float*)recon, target, n, loss);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(sse_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)recon, target, n, loss);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_tanh_backward(int dtype, const void* dy, const void* y, void* dx, int64_t n, void* stream) {
    VQK_REQUIRE(dy && y && dx, VQK_ERR_ARG);
    if (n <= 0) return VQK_OK;
    const dim3 grid(vqk_grid_1d(n, 256 * 4));
    if (dtype == VQK_F32) hipLaunchKernelGGL(tanh_bwd_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), (const float*)dy, (const float*)y, (float*)dx, n);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(tanh_bwd_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)dy, (const bf16_raw*)y, (bf16_raw*)dx, n);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_mse_tanh_backward(int dtype, const void* recon, const float* target, int64_t n, float gscale, const float* gscale_dev,
                          int through_tanh, void* d_pre, void* stream) {
    VQK_REQUIRE(recon && target && d_pre, VQK_ERR_ARG);
    if (n <= 0) return VQK_OK;
    const dim3 grid(vqk_grid_1d(n, 256 * 4));
    if (dtype == VQK_F32) hipLaunchKernelGGL(mse_tanh_bwd_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), (const float*)recon, target, n, gscale, gscale_dev, through_tanh, (float*)d_pre);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(mse_tanh_bwd_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)recon, target, n, gscale, gscale_dev, through_tanh, (bf16_raw*)d_pre);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_axpby(int dtype, const void* x, const void* y2, void* y, float a, float b, int64_t n, void* stream) {
    VQK_REQUIRE(x && y, VQK_ERR_ARG);
    const int v = dtype == VQK_F32 ? 4 : 8;
    VQK_REQUIRE(n >= 0 && n % v == 0, VQK_ERR_SHAPE);
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(y) && vqk_aligned16(y2), VQK_ERR_ALIGN);
    if (n == 0) return VQK_OK;
    const dim3 grid(vqk_grid_1d(n / v, 256, 256 * 16));
    if (dtype == VQK_F32) hipLaunchKernelGGL(axpby_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), (const float*)x, (const float*)y2, (float*)y, a, b, n / v);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(axpby_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), (const bf16_raw*)x, (const bf16_raw*)y2, (bf16_raw*)y, a, b, n / v);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_pair_stats(const float* pred, const float* target, int64_t n, float* out5, void* stream) {
    VQK_REQUIRE(pred && target && out5, VQK_ERR_ARG);
    if (n <= 0) return VQK_OK;
    hipLaunchKernelGGL(pair_stats_kernel, dim3(vqk_grid_1d(n, 256 * 8)), dim3(256), 0, vqk_stream(stream), pred, target, n, out5);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_ssim_sum(const float* pred, const float* target, int n, int c, int h, int w, const float* window, int ksize,
                 const float* stats5, float k1, float k2, float* out, void* stream) {
    VQK_REQUIRE(pred && target && window && stats5 && out, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && c > 0 && h >= ksize && w >= ksize, VQK_ERR_SHAPE);
    VQK_REQUIRE((int64_t)n * c <= 65535, VQK_ERR_SHAPE);
    const dim3 grid((unsigned)((w - ksize + 1 + 15) / 16), (unsigned)((h - ksize + 1 + 15) / 16), (unsigned)(n * c));
    hipStream_t st = vqk_stream(stream);
    if (ksize == 11) hipLaunchKernelGGL(ssim_sum_kernel<11>, grid, dim3(256), 0, st, pred, target, c, h, w, window, stats5, k1, k2, out);
    else if (ksize == 7) hipLaunchKernelGGL(ssim_sum_kernel<7>, grid, dim3(256), 0, st, pred, target, c, h, w, window, stats5, k1, k2, out);
    else return VQK_ERR_ARG;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

}  // extern "C"


// ------------------------------------------------------------------------------------------------
// DIAGNOSTIC: stand-in for a collective's kernel (include/vqk.h: vqk_probe_stream_add, tools/comm_probe.py)
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void probe_stream_add_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t vecs,
                                                               int passes, int sleep) {
    const int64_t per = (vecs + gridDim.x - 1) / gridDim.x;
    const int64_t v0 = (int64_t)blockIdx.x * per, v1 = min(vecs, v0 + per);
    for (int p = 0; p < passes; ++p)
        for (int64_t v = v0 + threadIdx.x; v < v1; v += 256) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + 4 * v);
            f32x4 b = *reinterpret_cast<const f32x4*>(dst + 4 * v);
            b += a;
            *reinterpret_cast<f32x4*>(dst + 4 * v) = b;
            for (int s = 0; s < sleep; ++s) __builtin_amdgcn_s_sleep(64);
        }
}
}  // namespace

extern "C" int vqk_probe_stream_add(const float* src, float* dst, int64_t bytes, int blocks, int passes, int sleep, void* stream) {
    VQK_REQUIRE(src && dst, VQK_ERR_ARG);
    VQK_REQUIRE(bytes > 0 && (bytes % 16) == 0 && blocks > 0 && passes > 0 && sleep >= 0, VQK_ERR_ARG);
    VQK_REQUIRE(vqk_aligned16(src) && vqk_aligned16(dst), VQK_ERR_ALIGN);
    // 32 KiB of (unused) dynamic LDS per block, like a collective's kernel: a CU that hosts one of these blocks has no room
    // left for a 150-KiB block of the persistent conv kernels -- the CU is HELD, not shared
    hipLaunchKernelGGL(probe_stream_add_kernel, dim3((unsigned)blocks), dim3(256), 32768, vqk_stream(stream), src, dst, bytes / 16, passes,
                       sleep);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}
