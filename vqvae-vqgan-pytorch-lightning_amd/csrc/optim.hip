// Fused AdamW over the flat fp32 parameter arena (vqvae/model.py:428 -> torch.optim.AdamW semantics):
//   p *= 1 - lr*wd ; m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps)
// One launch for every tensor: the decay / no-decay split (model.py:419-425) is a per-segment weight
// decay looked up by binary search; an optional bf16 shadow copy is refreshed in the same pass.
#include "common.h"

namespace {

// Block b owns the contiguous elements [b*CHUNK, (b+1)*CHUNK): 16-byte accesses, and the segment (weight decay)
// lookup -- a binary search over ~250 segment ends -- is redone only when a thread leaves its cached segment.
constexpr int ADAMW_CHUNK = 256 * 4 * 8;

__device__ __forceinline__ void adamw_one(float& pv, float gr, float* mp, float& vv, float wd, float lr, float b1, float b2,
                                          float eps, float step_size, float inv_sqrt_bc2, float gscale) {
    gr *= gscale;
    pv *= (1.0f - lr * wd);
    float mv = gr;
    if (mp) { mv = b1 * (*mp) + (1.0f - b1) * gr; *mp = mv; }
    vv = b2 * vv + (1.0f - b2) * gr * gr;
    const float denom = sqrtf(vv) * inv_sqrt_bc2 + eps;
    pv -= step_size * (mv / denom);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, int64_t n,
                                                    const int64_t* __restrict__ seg_end,
                                                    const float* __restrict__ seg_wd, int nseg, float lr, float b1,
                                                    float b2, float eps, float step_size, float inv_sqrt_bc2,
                                                    float gscale, bf16_raw* __restrict__ shadow, int vec_ok) {
    const int64_t base = (int64_t)blockIdx.x * ADAMW_CHUNK;
    int64_t s_lo = 0, s_hi = -1;                         // cached segment [s_lo, s_hi)
    float wd = 0.0f;
    auto lookup = [&](int64_t i) {
        if (i >= s_lo && i < s_hi) return;
        int lo = 0, hi = nseg - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_end[mid] > i) hi = mid; else lo = mid + 1; }
        wd = seg_wd[lo];
        s_lo = lo ? seg_end[lo - 1] : 0;
        s_hi = seg_end[lo];
    };
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        const int64_t i = base + it * 1024 + threadIdx.x * 4;
        if (i >= n) break;
        lookup(i);
        if (vec_ok && i + 4 <= n && i + 4 <= s_hi) {
            f32x4 pv = *reinterpret_cast<const f32x4*>(p + i);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
            f32x4 vv = *reinterpret_cast<const f32x4*>(v + i);
            f32x4 mv;
            if (m) mv = *reinterpret_cast<const f32x4*>(m + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float pe = pv[e], ve = vv[e], me = m ? mv[e] : 0.0f;
                adamw_one(pe, gv[e], m ? &me : nullptr, ve, wd, lr, b1, b2, eps, step_size, inv_sqrt_bc2, gscale);
                pv[e] = pe; vv[e] = ve; if (m) mv[e] = me;
            }
            *reinterpret_cast<f32x4*>(p + i) = pv;
            *reinterpret_cast<f32x4*>(v + i) = vv;
            if (m) *reinterpret_cast<f32x4*>(m + i) = mv;
            if (shadow) {
                u16x4 sv = {f32_to_bf16(pv[0]), f32_to_bf16(pv[1]), f32_to_bf16(pv[2]), f32_to_bf16(pv[3])};
                *reinterpret_cast<u16x4*>(shadow + i) = sv;
            }
        } else {
            for (int e = 0; e < 4 && i + e < n; ++e) {
                lookup(i + e);
                float pe = p[i + e], ve = v[i + e];
                adamw_one(pe, g[i + e], m ? m + i + e : nullptr, ve, wd, lr, b1, b2, eps, step_size, inv_sqrt_bc2, gscale);
                p[i + e] = pe; v[i + e] = ve;
                if (shadow) shadow[i + e] = f32_to_bf16(pe);
            }
        }
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
    const int vec_ok = vqk_aligned16(p) && vqk_aligned16(g) && vqk_aligned16(v) && (!m || vqk_aligned16(m)) &&
                       (!shadow || (reinterpret_cast<uintptr_t>(shadow) & 7u) == 0);
    const int64_t blocks = (n + ADAMW_CHUNK - 1) / ADAMW_CHUNK;
    VQK_REQUIRE(blocks < 0x7fffffff, VQK_ERR_SHAPE);
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, vqk_stream(stream), p, g, m, v, n,
                       seg_end, seg_wd, nseg, lr, beta1, beta2, eps, (float)((double)lr / bc1), (float)(1.0 / sqrt(bc2)),
                       grad_scale, reinterpret_cast<bf16_raw*>(shadow), vec_ok);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}
