// Fused AdamW over the flat fp32 parameter arena (vqvae/model.py:428 -> torch.optim.AdamW semantics):
//   p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps)
// One launch for every tensor: the decay / no-decay split (model.py:419-425) is a per-segment weight
// decay looked up by binary search; an optional bf16 shadow copy is refreshed in the same pass.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                    const int64_t* __restrict__ seg_end,
                                                    const float* __restrict__ seg_wd, int nseg, float lr, float b1,
                                                    float b2, float eps, float step_size, float inv_sqrt_bc2,
                                                    float gscale, bf16_raw* __restrict__ shadow) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int lo = 0, hi = nseg - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_end[mid] > i) hi = mid; else lo = mid + 1; }
        const float wd = seg_wd[lo];
        const float gr = g[i] * gscale;
        float pv = p[i] * (1.0f - lr * wd);
        float mv = gr;
        if (m) { mv = b1 * m[i] + (1.0f - b1) * gr; m[i] = mv; }
        const float vv = b2 * v[i] + (1.0f - b2) * gr * gr;
        v[i] = vv;
        const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
        pv -= step_size * (mv / denom);
        p[i] = pv;
        if (shadow) shadow[i] = f32_to_bf16(pv);
    }
}

}  // namespace

extern "C" int vqk_adamw(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end,
                         const float* seg_wd, int nseg, float lr, float beta1, float beta2, float eps, int step,
                         float grad_scale, void* shadow, void* stream) {
    VQK_REQUIRE(p && g && v && seg_end && seg_wd, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && nseg > 0 && step >= 1, VQK_ERR_ARG);
    VQK_REQUIRE(m || beta1 == 0.0f, VQK_ERR_ARG);      // m == NULL only when beta1 == 0 (then m == g)
    if (n == 0) return VQK_OK;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adamw_kernel, dim3(vqk_grid_1d(n, 256, 256 * 16)), dim3(256), 0, vqk_stream(stream), p, g, m, v, n,
                       seg_end, seg_wd, nseg, lr, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)),
                       grad_scale, reinterpret_cast<bf16_raw*>(shadow));
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}
