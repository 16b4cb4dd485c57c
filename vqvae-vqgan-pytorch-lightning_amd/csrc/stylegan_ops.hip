// The reference's two native plugins, rebuilt for gfx950 behind the C-ABI:
//   bias_act  (vqvae/modules/loss/stylegan2_discriminator/utils/ops/bias_act.cpp:32-90, bias_act.cu:24-147)
//   upfirdn2d (.../upfirdn2d.cpp:16-94, upfirdn2d.cu:29-200)
// Only what the discriminator path reaches is implemented: act in {linear, lrelu}, fp32,
// grad in {0, 1}; any 2-D FIR / up / down / padding for upfirdn2d.  Contiguous NCHW fp32.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ x, const float* __restrict__ b,
                                                       const float* __restrict__ yref, const float* __restrict__ dy,
                                                       float* __restrict__ y, int64_t numel, int64_t inner, int channels,
                                                       int grad, int act, float alpha, float gain, float clamp) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < numel; i += (int64_t)gridDim.x * 256) {
        float v = x[i];
        const float ref = yref ? yref[i] : 0.0f;
        const float up = dy ? dy[i] : 1.0f;
        if (grad == 0 && b) v += b[(i / inner) % channels];
        float r = v;
        if (act == 3) {
            // forward: sign of the biased input; backward: sign of the saved output / gain
            const float s = grad == 0 ? v : (gain != 0.0f ? ref / gain : 0.0f);
            r = s > 0.0f ? v : v * alpha;
        }
        r *= gain * up;
        if (clamp >= 0.0f) {
            if (grad == 0) r = (r > -clamp && r < clamp) ? r : (r >= 0.0f ? clamp : -clamp);
            else r = (ref > -clamp && ref < clamp) ? r : 0.0f;
        }
        y[i] = r;
    }
}

// y[n,c,oy,ox] = gain * sum_{fy,fx} F[fy][fx] * U[oy*downy + fy][ox*downx + fx]
// U = zero-stuffed (up) then padded/cropped input; F = f flipped unless `flip` (true convolution).
__global__ __launch_bounds__(256) void upfirdn2d_kernel(const float* __restrict__ x, const float* __restrict__ f,
                                                        float* __restrict__ y, int nc, int in_h, int in_w, int fh, int fw,
                                                        int upx, int upy, int downx, int downy, int padx0, int pady0,
                                                        int flip, float gain, int out_h, int out_w) {
    const int64_t total = (int64_t)nc * out_h * out_w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int ox = (int)(i % out_w);
        const int oy = (int)((i / out_w) % out_h);
        const int64_t plane = i / ((int64_t)out_w * out_h);
        const float* xp = x + plane * in_h * in_w;
        float acc = 0.0f;
        for (int fy = 0; fy < fh; ++fy) {
            const int uy = oy * downy + fy - pady0;          // coordinate in the zero-stuffed image
            if (uy < 0 || uy % upy) continue;
            const int iy = uy / upy;
            if (iy >= in_h) continue;
            for (int fx = 0; fx < fw; ++fx) {
                const int ux = ox * downx + fx - padx0;
                if (ux < 0 || ux % upx) continue;
                const int ix = ux / upx;
                if (ix >= in_w) continue;
                const float fv = flip ? f[fy * fw + fx] : f[(fh - 1 - fy) * fw + (fw - 1 - fx)];
                acc = __fmaf_rn(xp[iy * in_w + ix], fv, acc);
            }
        }
        y[i] = acc * gain;
    }
}

}  // namespace

extern "C" {

int vqk_bias_act(const float* x, const float* b, const float* xref, const float* yref, const float* dy, float* y,
                 int64_t numel, int64_t inner, int channels, int grad, int act, float alpha, float gain, float clamp,
                 void* stream) {
    (void)xref;                                   // only 'swish' (ref='x') reads xref; not on the path
    VQK_REQUIRE(x && y, VQK_ERR_ARG);
    VQK_REQUIRE(act == 1 || act == 3, VQK_ERR_ARG);
    VQK_REQUIRE(grad == 0 || grad == 1, VQK_ERR_ARG);
    VQK_REQUIRE(!(grad == 1 && act == 3 && !yref), VQK_ERR_ARG);
    VQK_REQUIRE(numel >= 0 && inner > 0 && (b == nullptr || channels > 0), VQK_ERR_SHAPE);
    if (numel == 0) return VQK_OK;
    hipLaunchKernelGGL(bias_act_kernel, dim3(vqk_grid_1d(numel, 256 * 4)), dim3(256), 0, vqk_stream(stream), x, b, yref, dy,
                       y, numel, inner, b ? channels : 1, grad, act, alpha, gain, clamp);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_upfirdn2d(const float* x, const float* f, float* y, int n, int c, int in_h, int in_w, int fh, int fw, int upx,
                  int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain,
                  int out_h, int out_w, void* stream) {
    VQK_REQUIRE(x && f && y, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && c > 0 && in_h > 0 && in_w > 0 && fh >= 1 && fw >= 1, VQK_ERR_SHAPE);
    VQK_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, VQK_ERR_ARG);
    // upfirdn2d.cpp:32-33
    VQK_REQUIRE(out_w == (in_w * upx + padx0 + padx1 - fw + downx) / downx, VQK_ERR_SHAPE);
    VQK_REQUIRE(out_h == (in_h * upy + pady0 + pady1 - fh + downy) / downy, VQK_ERR_SHAPE);
    VQK_REQUIRE(out_w >= 1 && out_h >= 1, VQK_ERR_SHAPE);
    const int64_t total = (int64_t)n * c * out_h * out_w;
    hipLaunchKernelGGL(upfirdn2d_kernel, dim3(vqk_grid_1d(total, 256, 256 * 16)), dim3(256), 0, vqk_stream(stream), x, f, y,
                       n * c, in_h, in_w, fh, fw, upx, upy, downx, downy, padx0, pady0, flip, gain, out_h, out_w);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

}  // extern "C"
