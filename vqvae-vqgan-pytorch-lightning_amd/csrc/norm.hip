// GroupNorm (unbiased variance) + SiLU, NHWC.  Restates vqvae/modules/autoencoder.py:25-39 and the SiLU
// that follows every norm (:65,:68,:139-140,:176-177); backward per SURVEY Appendix B:
//   dxhat = dy_pre * w_c ;  dx = (dxhat - mean(dxhat) - xhat * sum(dxhat*xhat)/(M-1)) * rstd
// (mean over M, the xhat term over M-1 because the variance is the unbiased one).
// Thread mapping: a thread owns one 16-byte channel vector slot (fixed channels) and strides over pixels,
// so per-channel affine terms / partial sums live in registers.
#include "common.h"
#ifndef GN_CL_SPIN_CAP
#define GN_CL_SPIN_CAP (1u << 20)     // polls before a cluster block gives up waiting (x ~0.25 us each: ~0.2 s)
#endif
__device__ int g_gn_cluster_timeouts = 0;     // cluster blocks that gave up (gn_cluster_bwd_kernel); read by vqk_gn_cluster_timeouts
#ifndef GN_CL_SLEEP
#define GN_CL_SLEEP 8     // s_sleep argument (x 64 cycles) between two polls of a cluster's ticket
#endif
#ifndef GN_UNROLL
#define GN_UNROLL 2
#endif
// The two-kernel passes stream their output with nontemporal stores (the lines are not parked in L2 / Infinity Cache on
// their way out): 256->256 @128^2 apply 113 -> 87 us (6.2 TB/s), forward pair 325 -> 294 us at 128 ch @256^2, backward
// pairs -2...11 %; -0.1 ms/step.  (The single-kernel small-map form keeps ordinary stores: its outputs are cache-sized.)
#define GN_STORE_FWD Vec16<T>::store_nt
#define GN_STORE_BWD Vec16<T>::store_nt
#include <stdlib.h>

namespace {

template <typename T> struct Raw16;
template <> struct Raw16<float> {
    typedef f32x4 type;
    __device__ static __forceinline__ void unpack(const f32x4& v, float (&o)[4]) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
};
template <> struct Raw16<bf16_raw> {
    typedef vqk_u32x4 type;
    __device__ static __forceinline__ void unpack(const vqk_u32x4& v, float (&o)[8]) { Vec16<bf16_raw>::unpack(v, o); }
};
// raw 16-byte load of one channel vector (NT: nontemporal)
template <typename T, bool NT>
__device__ __forceinline__ typename Raw16<T>::type gn_ld(const T* p) {
    typedef typename Raw16<T>::type raw_t;
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const raw_t*>(p));
    return *reinterpret_cast<const raw_t*>(p);
}
// The streaming passes walk a thread's pixels p, p + pstep, ... as a software pipeline: GN_DEPTH vectors per tensor are in
// flight AHEAD of the one being computed; the loads are unconditional (the address is clamped to the block's last pixel), so
// the compiler waits with counted vmcnt instead of draining the queue at every conditional (round 4: the skip-addend load
// of the backward was issued and waited for inside the loop -- one exposed HBM round trip per pixel vector).
#ifndef GN_DEPTH
#define GN_DEPTH 5        // swept 2 ... 5 (tools/ab_gn_depth.sh): 128 ch @256^2 backward 466 / 470 / 451 / 454 us, step 28.57 / 28.56 / 28.55 / 28.40 ms
#endif

// Block-level reduction shared by the two reducing kernels: every thread parks its 2V partial sums in LDS (lane-linear,
// 16-byte stores), then thread `col` adds up its column over the rows.  The previous form issued 2V LDS atomics per
// thread onto c addresses (up to 64 threads per address): serialised, several microseconds per block, i.e. most of a
// block's time on the 64^2 / 128^2 maps.
template <int V>
__device__ __forceinline__ void block_colsum(const float (&p0)[V], const float (&p1)[V], float* part, double* colsum, int c) {
    const int vpp = c / V, rows = 256 / vpp;
    float* mine = part + (size_t)threadIdx.x * 2 * V;            // thread id = prow * vpp + slot
#pragma unroll
    for (int i = 0; i < V; i += 4) {
        *reinterpret_cast<f32x4*>(mine + i) = f32x4{p0[i], p0[i + 1], p0[i + 2], p0[i + 3]};
        *reinterpret_cast<f32x4*>(mine + V + i) = f32x4{p1[i], p1[i + 1], p1[i + 2], p1[i + 3]};
    }
    __syncthreads();
    for (int col = threadIdx.x; col < 2 * c; col += 256) {       // col = slot * 2V + j,  j < V: first sum, else second
        const int slot = col / (2 * V), j = col - slot * 2 * V;
        double a = 0.0;
        for (int r = 0; r < rows; ++r) a += (double)part[((size_t)r * vpp + slot) * 2 * V + j];
        colsum[(j >= V ? c : 0) + slot * V + (j % V)] = a;
    }
    __syncthreads();
}

// `dpart` (deterministic mode): the block's group sums are STORED at dpart[((n*groups + g) * gridDim.x + block) * 2 + j] and the
// consumer adds the blocks in index order; otherwise they are accumulated with fp64 atomics (order-dependent in the last bits)
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ x, int64_t hw, int c, int groups,
                                                       int pix_per_block, double* __restrict__ acc,
                                                       double* __restrict__ dpart = nullptr) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sh = reinterpret_cast<double*>(smem);          // [2][c] column sums
    float* part = reinterpret_cast<float*>(sh + 2 * c);    // [256][2V] per-thread partials
    const int vpp = c / V, slot = threadIdx.x % vpp, prow = threadIdx.x / vpp, pstep = 256 / vpp;
    const int n = blockIdx.y;
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = min(hw, p0 + pix_per_block);
    float s[V], ss[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { s[i] = 0.f; ss[i] = 0.f; }
    const T* base = x + (int64_t)n * hw * c + slot * V;
#pragma unroll GN_UNROLL
    for (int64_t p = p0 + prow; p < p1; p += pstep) {
        float v[V];
        Vec16<T>::load(base + p * c, v);
#pragma unroll
        for (int i = 0; i < V; ++i) { s[i] += v[i]; ss[i] = __fmaf_rn(v[i], v[i], ss[i]); }
    }
    block_colsum<V>(s, ss, part, sh, c);
    const int cpg = c / groups;
    for (int g = threadIdx.x; g < groups; g += 256) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < cpg; ++i) { a += sh[g * cpg + i]; b += sh[c + g * cpg + i]; }
        if (dpart) {
            double* q = dpart + (((int64_t)n * groups + g) * gridDim.x + blockIdx.x) * 2;
            q[0] = a; q[1] = b;
        } else {
            atomicAdd(&acc[((int64_t)n * groups + g) * 2 + 0], a);
            atomicAdd(&acc[((int64_t)n * groups + g) * 2 + 1], b);
        }
    }
}

__global__ void gn_finalize_kernel(const double* __restrict__ acc, float* __restrict__ stats, int total, double m,
                                   float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const double s = acc[2 * i], ss = acc[2 * i + 1];
    const double mean = s / m;
    double var = (ss - s * mean) / (m - 1.0);          // unbiased (torch.var default)
    if (var < 0.0) var = 0.0;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// sigmoid via v_exp_f32 + v_rcp_f32 (1 ulp): an IEEE division costs ~10 VALU ops and made these passes VALU-bound
__device__ __forceinline__ float sigmoid_f(float y) { return __builtin_amdgcn_rcpf(1.0f + __expf(-y)); }
__device__ __forceinline__ float silu_f(float y) { return y * sigmoid_f(y); }

template <typename T, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                       const float* __restrict__ w, const float* __restrict__ b,
                                                       T* __restrict__ y, int64_t hw, int c, int groups, int silu,
                                                       int pix_per_block) {
    constexpr int V = Vec16<T>::N;
    const int vpp = c / V, slot = threadIdx.x % vpp, prow = threadIdx.x / vpp, pstep = 256 / vpp;
    const int n = blockIdx.y, cpg = c / groups;
    float scale[V], shift[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int ch = slot * V + i, g = ch / cpg;
        const float mean = stats[((int64_t)n * groups + g) * 2], rstd = stats[((int64_t)n * groups + g) * 2 + 1];
        scale[i] = rstd * w[ch];
        shift[i] = __fmaf_rn(-mean, scale[i], b[ch]);
    }
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = min(hw, p0 + pix_per_block);
    const int64_t off = (int64_t)n * hw * c + slot * V;
#pragma unroll GN_UNROLL
    for (int64_t p = p0 + prow; p < p1; p += pstep) {
        float v[V];
        Vec16<T>::load(x + off + p * c, v);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float t = __fmaf_rn(v[i], scale[i], shift[i]);
            v[i] = SILU ? silu_f(t) : t;
        }
        GN_STORE_FWD(y + off + p * c, v);
    }
}

// Workspace protocol shared by the forward and backward pairs: ws = N*G*2 doubles of partial sums followed by N
// block counters (one per sample, each in its own 8-byte slot), all ZERO on entry; the LAST block of a sample to
// finish the consumer kernel (atomic census at the very end, so nobody waits on it) zeroes that sample's slots again:
// one persistent workspace serves every GroupNorm call of a stream without a memset launch in between.
__device__ __forceinline__ void ws_release(double* __restrict__ ws, int n_samples, int groups) {
    __shared__ unsigned last;
    unsigned* counter = reinterpret_cast<unsigned*>(ws + (int64_t)n_samples * groups * 2 + blockIdx.y);
    if (threadIdx.x == 0) last = atomicAdd(counter, 1u) == gridDim.x - 1;
    __syncthreads();
    if (last) {
        double* mine = ws + (int64_t)blockIdx.y * groups * 2;
        for (int i = threadIdx.x; i < groups * 2; i += blockDim.x) mine[i] = 0.0;
        if (threadIdx.x == 0) *counter = 0u;
    }
}

// gn_finalize + gn_apply in one pass: every thread derives (mean, rstd) of its channels' groups from the double
// sums; the first pixel block of each sample also stores them as fp32 stats for the backward.
template <typename T, bool SILU>
__global__ __launch_bounds__(256) void gn_apply_fin_kernel(const T* __restrict__ x, double* __restrict__ ws,
                                                           float* __restrict__ stats, const float* __restrict__ w,
                                                           const float* __restrict__ b, T* __restrict__ y, int64_t hw,
                                                           int c, int groups, int silu, int pix_per_block, float eps,
                                                           const double* __restrict__ part = nullptr, int nblk = 0) {
    constexpr int V = Vec16<T>::N;
    const int vpp = c / V, slot = threadIdx.x % vpp, prow = threadIdx.x / vpp, pstep = 256 / vpp;
    const int n = blockIdx.y, cpg = c / groups;
    const double m = (double)hw * cpg;
    // finalize once per block: thread g turns the double sums of group g into (mean, rstd)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* sh = reinterpret_cast<float*>(smem);            // [groups][2]
    for (int g = threadIdx.x; g < groups; g += 256) {
        double s_, ss;
        if (part) {                                         // deterministic mode: the statistics blocks' partials, in block order
            s_ = 0.0; ss = 0.0;
            const double* q = part + ((int64_t)n * groups + g) * nblk * 2;
            for (int k = 0; k < nblk; ++k) { s_ += q[2 * k]; ss += q[2 * k + 1]; }
        } else {
            s_ = ws[((int64_t)n * groups + g) * 2]; ss = ws[((int64_t)n * groups + g) * 2 + 1];
        }
        const double mean = s_ / m;
        double var = (ss - s_ * mean) / (m - 1.0);          // unbiased (torch.var default)
        if (var < 0.0) var = 0.0;
        const float mean_f = (float)mean, rstd_f = (float)(1.0 / sqrt(var + (double)eps));
        sh[2 * g] = mean_f; sh[2 * g + 1] = rstd_f;
        if (blockIdx.x == 0) {
            stats[((int64_t)n * groups + g) * 2] = mean_f;
            stats[((int64_t)n * groups + g) * 2 + 1] = rstd_f;
        }
    }
    __syncthreads();
    float scale[V], shift[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int ch = slot * V + i, g = ch / cpg;
        scale[i] = sh[2 * g + 1] * w[ch];
        shift[i] = __fmaf_rn(-sh[2 * g], scale[i], b[ch]);
    }
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = min(hw, p0 + pix_per_block);
    const int64_t off = (int64_t)n * hw * c + slot * V;
    {
        typedef typename Raw16<T>::type raw_t;
        constexpr int D = GN_DEPTH + 1;
        const int64_t last = p1 - 1, step = pstep;
        raw_t buf[D];
        int64_t p = p0 + prow;
#pragma unroll
        for (int u = 0; u < D; ++u) buf[u] = gn_ld<T, false>(x + off + min(p + u * step, last) * c);
        for (; p < p1; p += D * step) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int64_t q = p + u * step;
                float v[V];
                Raw16<T>::unpack(buf[u], v);
                buf[u] = gn_ld<T, false>(x + off + min(q + D * step, last) * c);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    float t = __fmaf_rn(v[i], scale[i], shift[i]);
                    v[i] = SILU ? silu_f(t) : t;
                }
                if (q < p1) GN_STORE_FWD(y + off + q * c, v);
            }
        }
    }
    if (!part) ws_release(ws, gridDim.y, groups);
}

// pass 1 of the backward: per-channel sums of dy_pre and dy_pre*xhat (-> dw, db) and the per-group
// sums of dxhat and dxhat*xhat (-> red[n][g][2], double).
template <typename T, bool SILU>
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const T* __restrict__ dy, float* __restrict__ dw,
                                                            float* __restrict__ db, double* __restrict__ red,
                                                            int64_t hw, int c, int groups, int silu,
                                                            int pix_per_block, double* __restrict__ gpart = nullptr,
                                                            float* __restrict__ cpart = nullptr) {
    constexpr int V = Vec16<T>::N;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* sh = reinterpret_cast<double*>(smem);          // [2][c] column sums
    float* part = reinterpret_cast<float*>(sh + 2 * c);    // [256][2V] per-thread partials
    const int vpp = c / V, slot = threadIdx.x % vpp, prow = threadIdx.x / vpp, pstep = 256 / vpp;
    const int n = blockIdx.y, cpg = c / groups;
    float mean[V], rstd[V], wv[V], bv[V], a[V], bb[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int ch = slot * V + i, g = ch / cpg;
        mean[i] = stats[((int64_t)n * groups + g) * 2]; rstd[i] = stats[((int64_t)n * groups + g) * 2 + 1];
        wv[i] = w[ch]; bv[i] = b[ch]; a[i] = 0.f; bb[i] = 0.f;
    }
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = min(hw, p0 + pix_per_block);
    const int64_t off = (int64_t)n * hw * c + slot * V;
    {
        typedef typename Raw16<T>::type raw_t;
        constexpr int D = GN_DEPTH;
        const int64_t last = p1 - 1, step = pstep;
        raw_t bx[D], bg[D];
        int64_t p = p0 + prow;
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int64_t q = min(p + u * step, last) * c;
            bx[u] = gn_ld<T, false>(x + off + q); bg[u] = gn_ld<T, false>(dy + off + q);
        }
        for (; p < p1; p += D * step) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int64_t q = p + u * step;
                float xv[V], gv[V];
                Raw16<T>::unpack(bx[u], xv);
                Raw16<T>::unpack(bg[u], gv);
                const int64_t qn = min(q + D * step, last) * c;
                bx[u] = gn_ld<T, false>(x + off + qn); bg[u] = gn_ld<T, false>(dy + off + qn);
                const float live = q < p1 ? 1.0f : 0.0f;          // (beyond the block: the clamped re-read counts for nothing)
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float xh = (xv[i] - mean[i]) * rstd[i];
                    float g = gv[i] * live;
                    if constexpr (SILU) {
                        const float yv = __fmaf_rn(xh, wv[i], bv[i]);
                        const float sg = sigmoid_f(yv);
                        g *= sg * (1.0f + yv * (1.0f - sg));
                    }
                    a[i] += g;
                    bb[i] = __fmaf_rn(g, xh, bb[i]);
                }
            }
        }
    }
    block_colsum<V>(a, bb, part, sh, c);
    // deterministic mode: per-block channel sums to cpart[(n * gridDim.x + block)][2c] (added up in (sample, block) order by
    // gn_bwd_finish_kernel) and group sums to gpart[((n*groups + g) * gridDim.x + block) * 2] (added up by the apply pass)
    for (int ch = threadIdx.x; ch < c; ch += 256) {
        if (cpart) {
            float* q = cpart + ((int64_t)n * gridDim.x + blockIdx.x) * 2 * c;
            q[ch] = (float)sh[ch]; q[c + ch] = (float)sh[c + ch];
        } else {
            atomicAdd(db + ch, (float)sh[ch]);
            atomicAdd(dw + ch, (float)sh[c + ch]);
        }
    }
    for (int g = threadIdx.x; g < groups; g += 256) {
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < cpg; ++i) {
            const int ch = g * cpg + i;
            s1 += sh[ch] * (double)w[ch];
            s2 += sh[c + ch] * (double)w[ch];
        }
        if (gpart) {
            double* q = gpart + (((int64_t)n * groups + g) * gridDim.x + blockIdx.x) * 2;
            q[0] = s1; q[1] = s2;
        } else {
            atomicAdd(&red[((int64_t)n * groups + g) * 2 + 0], s1);
            atomicAdd(&red[((int64_t)n * groups + g) * 2 + 1], s2);
        }
    }
}

// deterministic mode: db[ch] += sum over (sample, block) rows of cpart[row][ch], dw[ch] likewise, in a FIXED order: a block owns
// 8 columns; 32 row lanes add rows lane, lane + 32, ... each, then thread `col` adds the 32 lane sums in lane order
__global__ __launch_bounds__(256) void gn_bwd_finish_kernel(const float* __restrict__ cpart, int rows, int c, float* __restrict__ dw,
                                                            float* __restrict__ db) {
    __shared__ float part[32][8];
    const int col = (int)blockIdx.x * 8 + (threadIdx.x & 7), rl = threadIdx.x >> 3;
    float s = 0.f;
    if (col < 2 * c)
        for (int r = rl; r < rows; r += 32) s += cpart[(int64_t)r * 2 * c + col];
    part[rl][threadIdx.x & 7] = s;
    __syncthreads();
    if (threadIdx.x < 8 && col < 2 * c) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += part[k][threadIdx.x];
        if (col < c) db[col] += t; else dw[col - c] += t;
    }
}

// NT: x and dy are read for the last time here; for tensors beyond the Infinity Cache (>= 192 MB) nontemporal loads keep
// them from evicting what the next kernel reads (backward pair 530 -> 485 us at 128 ch @256^2, 272 -> 248 us at 256 ch
// @128^2); for cache-sized tensors they are slower (127 -> 137 us at 128 ch @128^2), so the host picks.
// ADD: 0 no addend, 1 addend at the same resolution (`add`, or dx itself when add == NULL), 2 addend at half resolution
template <typename T, bool NT, bool SILU, int ADD>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           const T* __restrict__ dy, T* __restrict__ dx,
                                                           const T* __restrict__ add,
                                                           double* __restrict__ red, int64_t hw, int c,
                                                           int groups, int silu, int accumulate, int pix_per_block,
                                                           int add_w, float add_scale,
                                                           const double* __restrict__ gpart = nullptr, int nblk = 0,
                                                           float* __restrict__ dx_colsum = nullptr) {
    // dx_colsum (optional, fp32 [C], accumulated into): the per-channel sums of the dx this pass writes -- the BIAS gradient of
    // the conv that produced x (the Upsample convs: autoencoder.py:102-105), which was a separate 185-us column-sum pass over
    // the 537-MB gradient at 128 ch @256^2.  A thread owns fixed channels, so the sums live in 8 registers.
    constexpr int V = Vec16<T>::N;
    const int vpp = c / V, slot = threadIdx.x % vpp, prow = threadIdx.x / vpp, pstep = 256 / vpp;
    const int n = blockIdx.y, cpg = c / groups;
    const double m = (double)hw * cpg;
    float mean[V], rstd[V], wv[V], bv[V], k1[V], k2[V];
    float csum[V];
#pragma unroll
    for (int i = 0; i < V; ++i) csum[i] = 0.0f;
    __shared__ float gk[256][2];                            // deterministic mode: (k1, k2) per group, summed once per block
    if (gpart) {                                            // the reduce blocks' partials, in block order
        for (int g = threadIdx.x; g < groups; g += 256) {
            double r1 = 0.0, r2 = 0.0;
            const double* q = gpart + ((int64_t)n * groups + g) * nblk * 2;
            for (int k = 0; k < nblk; ++k) { r1 += q[2 * k]; r2 += q[2 * k + 1]; }
            gk[g][0] = (float)(r1 / m); gk[g][1] = (float)(r2 / (m - 1.0));
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int ch = slot * V + i, g = ch / cpg;
        mean[i] = stats[((int64_t)n * groups + g) * 2]; rstd[i] = stats[((int64_t)n * groups + g) * 2 + 1];
        wv[i] = w[ch]; bv[i] = b[ch];
        if (gpart) {
            k1[i] = gk[g][0]; k2[i] = gk[g][1];
        } else {
            k1[i] = (float)(red[((int64_t)n * groups + g) * 2] / m);
            k2[i] = (float)(red[((int64_t)n * groups + g) * 2 + 1] / (m - 1.0));
        }
    }
    const int64_t p0 = (int64_t)blockIdx.x * pix_per_block;
    const int64_t p1 = min(hw, p0 + pix_per_block);
    const int64_t off = (int64_t)n * hw * c + slot * V;
    {
        typedef typename Raw16<T>::type raw_t;
        constexpr int D = GN_DEPTH;
        const int64_t last = p1 - 1, step = pstep;
        const T* ap = ADD == 1 ? (add ? add : dx) : add;
        const int64_t aoff = (int64_t)n * (hw >> 2) * c + slot * V;     // ADD == 2: the half-resolution addend of this sample
        auto add_ptr = [&](int64_t q) -> const T* {
            if (ADD == 2) {
                const int qi = (int)q, row = qi / add_w, col = qi - row * add_w;
                return ap + aoff + (int64_t)((row >> 1) * (add_w >> 1) + (col >> 1)) * c;
            }
            return ap + off + q * c;
        };
        raw_t bx[D], bg[D], ba[ADD ? D : 1];
        int64_t p = p0 + prow;
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int64_t q = min(p + u * step, last);
            bx[u] = gn_ld<T, NT>(x + off + q * c); bg[u] = gn_ld<T, NT>(dy + off + q * c);
            if (ADD) ba[u] = gn_ld<T, false>(add_ptr(q));
        }
        (void)accumulate;
        for (; p < p1; p += D * step) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int64_t q = p + u * step;
                float xv[V], gv[V], ov[V];
                Raw16<T>::unpack(bx[u], xv);
                Raw16<T>::unpack(bg[u], gv);
                if (ADD) {
                    Raw16<T>::unpack(ba[ADD ? u : 0], ov);
                    if (ADD == 2) {
#pragma unroll
                        for (int i = 0; i < V; ++i) ov[i] *= add_scale;
                    }
                }
                const int64_t qn = min(q + D * step, last);
                bx[u] = gn_ld<T, NT>(x + off + qn * c); bg[u] = gn_ld<T, NT>(dy + off + qn * c);
                if (ADD) ba[ADD ? u : 0] = gn_ld<T, false>(add_ptr(qn));
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float xh = (xv[i] - mean[i]) * rstd[i];
                    float g = gv[i];
                    if constexpr (SILU) {
                        const float yv = __fmaf_rn(xh, wv[i], bv[i]);
                        const float sg = sigmoid_f(yv);
                        g *= sg * (1.0f + yv * (1.0f - sg));
                    }
                    const float r = (g * wv[i] - k1[i] - xh * k2[i]) * rstd[i];
                    ov[i] = ADD ? ov[i] + r : r;
                }
                if (q < p1) {
                    GN_STORE_BWD(dx + off + q * c, ov);
                    if (dx_colsum) {                             // kernel-uniform
#pragma unroll
                        for (int i = 0; i < V; ++i) csum[i] += ov[i];
                    }
                }
            }
        }
    }
    if (dx_colsum) {
        // the block's rows of a slot are added through LDS ([prow][channel], reusing nothing else), one atomic per channel
        __shared__ float cs_part[256 * 8];
#pragma unroll
        for (int i = 0; i < V; ++i) cs_part[prow * c + slot * V + i] = csum[i];      // vpp * V = c columns, 256 / vpp rows: 256 * V floats
        __syncthreads();
        for (int ch = threadIdx.x; ch < c; ch += 256) {
            float a = 0.0f;
            for (int r = 0; r < pstep; ++r) a += cs_part[r * c + ch];
            atomicAdd(dx_colsum + ch, a);
        }
    }
    if (!gpart) ws_release(red, gridDim.y, groups);
}

// ------------------------------------------------------------------------------------------------
// Small maps (H*W <= 1024: the 16^2 / 32^2 levels, 21 of the model's 42 GroupNorms): ONE kernel per direction.  A block
// owns (sample, 32-channel slice = whole groups) over ALL pixels, keeps its <= 16 pixels per thread in registers (raw 16-byte
// vectors), reduces inside the block (no global atomics for the statistics, no workspace, no second read) and applies from
// the registers.  The two-kernel form spent most of its 20-60 us per call on launches, global atomics and the round trip.
// ------------------------------------------------------------------------------------------------
template <typename T, int PPT, bool SILU>
__global__ __launch_bounds__(256) void gn_small_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ b, T* __restrict__ y,
                                                           float* __restrict__ stats, int c, int groups, int silu, float eps) {
    constexpr int V = Vec16<T>::N, SLOTS = 32 / V, ROWS = 256 / SLOTS, HW = PPT * ROWS;
    typedef typename Raw16<T>::type raw_t;
    __shared__ double sh[2][32];                                 // per-channel sums of the slice
    __shared__ double shw[4][2][32];                             // ... per wave: added in wave order (no atomics: deterministic)
    __shared__ float gstat[32][2];                               // (mean, rstd) per channel of the slice
    const int slot = threadIdx.x % SLOTS, prow = threadIdx.x / SLOTS;
    const int n = blockIdx.y, ch0 = blockIdx.x * 32, cpg = c / groups;
    const int64_t off = (int64_t)n * HW * c + ch0 + slot * V;
    raw_t r[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) r[k] = *reinterpret_cast<const raw_t*>(x + off + (int64_t)(prow + k * ROWS) * c);
    float s[V], ss[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { s[i] = 0.f; ss[i] = 0.f; }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        float v[V];
        Raw16<T>::unpack(r[k], v);
#pragma unroll
        for (int i = 0; i < V; ++i) { s[i] += v[i]; ss[i] = __fmaf_rn(v[i], v[i], ss[i]); }
    }
    // keep the RAW vectors (not their fp32 expansion, twice the registers) alive across the reduction
#pragma unroll
    for (int k = 0; k < PPT; ++k) asm volatile("" : "+v"(r[k]));
    // the ROWS threads of a slot sit SLOTS lanes apart: fold the lanes of a wave first, then one LDS atomic per wave
#pragma unroll
    for (int i = 0; i < V; ++i) {
#pragma unroll
        for (int o = 32; o >= SLOTS; o >>= 1) { s[i] += __shfl_xor(s[i], o, 64); ss[i] += __shfl_xor(ss[i], o, 64); }
    }
    if ((threadIdx.x & 63) < SLOTS) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            shw[threadIdx.x >> 6][0][slot * V + i] = (double)s[i];
            shw[threadIdx.x >> 6][1][slot * V + i] = (double)ss[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int j = threadIdx.x >> 5, cl = threadIdx.x & 31;
        sh[j][cl] = ((shw[0][j][cl] + shw[1][j][cl]) + shw[2][j][cl]) + shw[3][j][cl];
    }
    __syncthreads();
    if (threadIdx.x < 32) {                                      // thread = channel of the slice: its group's moments
        const int g0 = (threadIdx.x / cpg) * cpg;                // cpg divides 32
        double a = 0.0, q = 0.0;
        for (int i = 0; i < cpg; ++i) { a += sh[0][g0 + i]; q += sh[1][g0 + i]; }
        const double m = (double)HW * cpg, mean = a / m;
        double var = (q - a * mean) / (m - 1.0);                 // unbiased (torch.var default)
        if (var < 0.0) var = 0.0;
        const float mean_f = (float)mean, rstd_f = (float)(1.0 / sqrt(var + (double)eps));
        gstat[threadIdx.x][0] = mean_f; gstat[threadIdx.x][1] = rstd_f;
        if (threadIdx.x == g0) {
            const int g = (ch0 + g0) / cpg;
            stats[((int64_t)n * groups + g) * 2] = mean_f;
            stats[((int64_t)n * groups + g) * 2 + 1] = rstd_f;
        }
    }
    __syncthreads();
    float scale[V], shift[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int cl = slot * V + i;
        scale[i] = gstat[cl][1] * w[ch0 + cl];
        shift[i] = __fmaf_rn(-gstat[cl][0], scale[i], b[ch0 + cl]);
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        float v[V];
        Raw16<T>::unpack(r[k], v);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const float t = __fmaf_rn(v[i], scale[i], shift[i]);
            v[i] = SILU ? silu_f(t) : t;
        }
        Vec16<T>::store(y + off + (int64_t)(prow + k * ROWS) * c, v);
    }
}

// KEEP = false (16 pixels per thread, the 32^2 maps): x / dy do not fit the registers twice, the second pass re-reads the
// block's own 128 KiB from L2 instead.
template <typename T, int PPT, bool KEEP, bool SILU>
__global__ __launch_bounds__(256) void gn_small_bwd_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                           const float* __restrict__ w, const float* __restrict__ b,
                                                           const T* __restrict__ dy, T* __restrict__ dx,
                                                           const T* __restrict__ add, float* __restrict__ dw,
                                                           float* __restrict__ db, int c, int groups, int silu,
                                                           int accumulate, float* __restrict__ cpart = nullptr) {
    constexpr int V = Vec16<T>::N, SLOTS = 32 / V, ROWS = 256 / SLOTS, HW = PPT * ROWS;
    typedef typename Raw16<T>::type raw_t;
    __shared__ float sh[2][32];                                  // per-channel sums of g and g * xhat
    __shared__ float shw[4][2][32];                              // ... per wave: added in wave order (no atomics: deterministic)
    __shared__ float kk[32][2];                                  // (k1, k2) of the channel's group
    const int slot = threadIdx.x % SLOTS, prow = threadIdx.x / SLOTS;
    const int n = blockIdx.y, ch0 = blockIdx.x * 32, cpg = c / groups;
    const int64_t off = (int64_t)n * HW * c + ch0 + slot * V;
    raw_t rx[KEEP ? PPT : 1], rg[KEEP ? PPT : 1];
    if (KEEP) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            rx[k] = *reinterpret_cast<const raw_t*>(x + off + (int64_t)(prow + k * ROWS) * c);
            rg[k] = *reinterpret_cast<const raw_t*>(dy + off + (int64_t)(prow + k * ROWS) * c);
        }
    }
    float mean[V], rstd[V], wv[V], bv[V], a[V], bb[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int ch = ch0 + slot * V + i, g = ch / cpg;
        mean[i] = stats[((int64_t)n * groups + g) * 2]; rstd[i] = stats[((int64_t)n * groups + g) * 2 + 1];
        wv[i] = w[ch]; bv[i] = b[ch]; a[i] = 0.f; bb[i] = 0.f;
    }
    auto pre = [&](float xv, float gv, int i, float& xh) -> float {      // dy before the SiLU, xhat
        xh = (xv - mean[i]) * rstd[i];
        if constexpr (SILU) {
            const float yv = __fmaf_rn(xh, wv[i], bv[i]);
            const float sg = sigmoid_f(yv);
            gv *= sg * (1.0f + yv * (1.0f - sg));
        }
        return gv;
    };
#pragma unroll 4
    for (int k = 0; k < PPT; ++k) {
        float xv[V], gv[V];
        if (KEEP) {
            Raw16<T>::unpack(rx[k], xv);
            Raw16<T>::unpack(rg[k], gv);
        } else {
            Vec16<T>::load(x + off + (int64_t)(prow + k * ROWS) * c, xv);
            Vec16<T>::load(dy + off + (int64_t)(prow + k * ROWS) * c, gv);
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float xh;
            const float g = pre(xv[i], gv[i], i, xh);
            a[i] += g;
            bb[i] = __fmaf_rn(g, xh, bb[i]);
        }
    }
    if (KEEP) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) { asm volatile("" : "+v"(rx[k])); asm volatile("" : "+v"(rg[k])); }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
#pragma unroll
        for (int o = 32; o >= SLOTS; o >>= 1) { a[i] += __shfl_xor(a[i], o, 64); bb[i] += __shfl_xor(bb[i], o, 64); }
    }
    if ((threadIdx.x & 63) < SLOTS) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            shw[threadIdx.x >> 6][0][slot * V + i] = a[i];
            shw[threadIdx.x >> 6][1][slot * V + i] = bb[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int j = threadIdx.x >> 5, cl = threadIdx.x & 31;
        sh[j][cl] = ((shw[0][j][cl] + shw[1][j][cl]) + shw[2][j][cl]) + shw[3][j][cl];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
        const int cl = threadIdx.x, ch = ch0 + cl;
        if (cpart) {                                             // deterministic mode: per-sample sums, added up in sample order afterwards
            cpart[(int64_t)n * 2 * c + ch] = sh[0][cl];
            cpart[(int64_t)n * 2 * c + c + ch] = sh[1][cl];
        } else {
            atomicAdd(db + ch, sh[0][cl]);
            atomicAdd(dw + ch, sh[1][cl]);
        }
        const int g0 = (cl / cpg) * cpg;
        double s1 = 0.0, s2 = 0.0;
        for (int i = 0; i < cpg; ++i) {
            s1 += (double)sh[0][g0 + i] * (double)w[ch0 + g0 + i];
            s2 += (double)sh[1][g0 + i] * (double)w[ch0 + g0 + i];
        }
        const double m = (double)HW * cpg;
        kk[cl][0] = (float)(s1 / m); kk[cl][1] = (float)(s2 / (m - 1.0));
    }
    __syncthreads();
    float k1[V], k2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { k1[i] = kk[slot * V + i][0]; k2[i] = kk[slot * V + i][1]; }
#pragma unroll 4
    for (int k = 0; k < PPT; ++k) {
        float xv[V], gv[V], ov[V];
        const int64_t o = off + (int64_t)(prow + k * ROWS) * c;
        if (KEEP) {
            Raw16<T>::unpack(rx[k], xv);
            Raw16<T>::unpack(rg[k], gv);
        } else {
            Vec16<T>::load(x + o, xv);
            Vec16<T>::load(dy + o, gv);
        }
        if (accumulate) Vec16<T>::load((add ? add : dx) + o, ov);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float xh;
            const float g = pre(xv[i], gv[i], i, xh);
            const float r = (g * wv[i] - k1[i] - xh * k2[i]) * rstd[i];
            ov[i] = accumulate ? ov[i] + r : r;
        }
        Vec16<T>::store(dx + o, ov);
    }
}

// ------------------------------------------------------------------------------------------------
// Mid-size maps (round 4): the single-kernel backward for maps that do not fit ONE block's registers.  CL blocks form a
// cluster per (sample, 32-channel slice); each keeps its PPT x 64 pixels of x and dy in registers, adds its group sums to the
// stream's fp64 workspace (the slots of the two-kernel form), arrives at the cluster's ticket and spins until all CL blocks
// have arrived, then applies from the registers: x and dy are read ONCE (3 tensor passes instead of 5), one launch instead
// of two.  The blocks of a cluster are consecutive block ids (dispatched in order: the oldest incomplete cluster is always
// fully resident before any younger one holds the slots it needs), the last block to LEAVE zeroes the cluster's sums and
// tickets again (workspace protocol of the other passes: zero on entry, zero on exit).  Not for deterministic mode (fp64
// atomics in arrival order); `add` at half resolution (add_w) as in gn_bwd_apply_kernel.
// ------------------------------------------------------------------------------------------------
// SL: channels per slice -- 64 (a pixel's slice is one whole 128-byte line in bf16) or 32
template <typename T, int PPT, int SL, bool SILU>
__global__ __launch_bounds__(256) void gn_cluster_bwd_kernel(const T* __restrict__ x, const float* __restrict__ stats,
                                                             const float* __restrict__ w, const float* __restrict__ b,
                                                             const T* __restrict__ dy, T* __restrict__ dx,
                                                             const T* __restrict__ add, float* __restrict__ dw,
                                                             float* __restrict__ db, double* __restrict__ red,
                                                             unsigned* __restrict__ tickets, int64_t hw, int c, int groups,
                                                             int silu, int accumulate, int cl, int add_w, float add_scale) {
    constexpr int V = Vec16<T>::N, SLOTS = SL / V, ROWS = 256 / SLOTS;
    typedef typename Raw16<T>::type raw_t;
    __shared__ float sh[2][SL];
    __shared__ float shw[4][2][SL];
    __shared__ float kk[SL][2];
    const int slot = threadIdx.x % SLOTS, prow = threadIdx.x / SLOTS;
    const int slices = c / SL;
    const int slice = (int)blockIdx.x / cl, rank = (int)blockIdx.x - slice * cl;
    const int n = blockIdx.y, ch0 = slice * SL, cpg = c / groups;
    const int64_t pix0 = (int64_t)rank * (PPT * ROWS) + prow;
    const int64_t off = (int64_t)n * hw * c + ch0 + slot * V;
    raw_t rx[PPT], rg[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        rx[k] = *reinterpret_cast<const raw_t*>(x + off + (pix0 + k * ROWS) * c);      // (ordinary loads: a 128-byte line holds
        rg[k] = *reinterpret_cast<const raw_t*>(dy + off + (pix0 + k * ROWS) * c);     // two slices -- the sibling cluster wants it too)
    }
    float mean[V], rstd[V], wv[V], bv[V], a[V], bb[V];
#pragma unroll
    for (int i = 0; i < V; ++i) {
        const int ch = ch0 + slot * V + i, g = ch / cpg;
        mean[i] = stats[((int64_t)n * groups + g) * 2]; rstd[i] = stats[((int64_t)n * groups + g) * 2 + 1];
        wv[i] = w[ch]; bv[i] = b[ch]; a[i] = 0.f; bb[i] = 0.f;
    }
    auto pre = [&](float xv, float gv, int i, float& xh) -> float {      // dy before the SiLU, xhat
        xh = (xv - mean[i]) * rstd[i];
        if constexpr (SILU) {
            const float yv = __fmaf_rn(xh, wv[i], bv[i]);
            const float sg = sigmoid_f(yv);
            gv *= sg * (1.0f + yv * (1.0f - sg));
        }
        return gv;
    };
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        float xv[V], gv[V];
        Raw16<T>::unpack(rx[k], xv);
        Raw16<T>::unpack(rg[k], gv);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float xh;
            const float g = pre(xv[i], gv[i], i, xh);
            a[i] += g;
            bb[i] = __fmaf_rn(g, xh, bb[i]);
        }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) { asm volatile("" : "+v"(rx[k])); asm volatile("" : "+v"(rg[k])); }
#pragma unroll
    for (int i = 0; i < V; ++i) {
#pragma unroll
        for (int o = 32; o >= SLOTS; o >>= 1) { a[i] += __shfl_xor(a[i], o, 64); bb[i] += __shfl_xor(bb[i], o, 64); }
    }
    if ((threadIdx.x & 63) < SLOTS) {
#pragma unroll
        for (int i = 0; i < V; ++i) {
            shw[threadIdx.x >> 6][0][slot * V + i] = a[i];
            shw[threadIdx.x >> 6][1][slot * V + i] = bb[i];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * SL) {
        const int j = threadIdx.x / SL, cl_ = threadIdx.x % SL;
        sh[j][cl_] = ((shw[0][j][cl_] + shw[1][j][cl_]) + shw[2][j][cl_]) + shw[3][j][cl_];
    }
    __syncthreads();
    // Cross-block exchange WITHOUT acquire / release fences: on this multi-die part an agent-scope release writes the die's
    // whole L2 back and an acquire invalidates it (measured: the fenced form ran 3-6x slower than the two-kernel passes).
    // Every access to the sums and tickets is an agent-scope atomic (performed at the memory side, past the per-die L2s);
    // the sums are RETURNING atomics, so the wave has their results -- they have been performed -- before its arrival is
    // counted, and a block that sees all CL arrivals reads the complete sums.
    unsigned* arrive = tickets + ((int64_t)n * slices + slice) * 2;
    if (threadIdx.x < SL) {
        const int ci = threadIdx.x, ch = ch0 + ci;
        atomicAdd(db + ch, sh[0][ci]);
        atomicAdd(dw + ch, sh[1][ci]);
        const int g0 = (ci / cpg) * cpg;
        double seen = 0.0;
        if (ci == g0) {                                          // one thread per group of the slice: its sums into the workspace
            double s1 = 0.0, s2 = 0.0;
            for (int i = 0; i < cpg; ++i) {
                s1 += (double)sh[0][g0 + i] * (double)w[ch0 + g0 + i];
                s2 += (double)sh[1][g0 + i] * (double)w[ch0 + g0 + i];
            }
            const int g = (ch0 + g0) / cpg;
            seen = __hip_atomic_fetch_add(&red[((int64_t)n * groups + g) * 2 + 0], s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            seen += __hip_atomic_fetch_add(&red[((int64_t)n * groups + g) * 2 + 1], s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("" :: "v"(seen));                           // the returned values are waited for before the barrier below
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // BOUNDED spin.  Forward progress rests on (a) every XCD dispatching its share of a grid in block-index order and (b) a
        // cluster being <= 8 consecutive block ids: the lowest undispatched block's XCD then only holds blocks whose whole cluster
        // is already dispatched, so they finish and free its slot (include/vqk.h: vqk_gn_backward_ws).  Should that ever not hold
        // (a future dispatcher, a debugger pausing one die), the block gives up after ~0.2 s, counts the event in
        // g_gn_cluster_timeouts (vqk_gn_cluster_timeouts) and falls through with whatever sums have arrived: a WRONG result that the
        // host can detect, instead of a hung GPU.
        unsigned spins = 0;
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)cl) {
            __builtin_amdgcn_s_sleep(GN_CL_SLEEP);
            if (++spins >= GN_CL_SPIN_CAP) { atomicAdd(&g_gn_cluster_timeouts, 1); break; }
        }
    }
    __syncthreads();
    if (threadIdx.x < SL) {
        const int ci = threadIdx.x, g = (ch0 + ci) / cpg;
        const double m = (double)hw * cpg;
        const double s1 = __hip_atomic_load(&red[((int64_t)n * groups + g) * 2 + 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const double s2 = __hip_atomic_load(&red[((int64_t)n * groups + g) * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        kk[ci][0] = (float)(s1 / m); kk[ci][1] = (float)(s2 / (m - 1.0));
    }
    __syncthreads();                                             // (every thread's loads above have returned: kk is written)
    if (threadIdx.x == 0) {
        // the last block to leave (every block has READ the sums by then) restores the workspace: sums and tickets zero
        const unsigned left = __hip_atomic_fetch_add(arrive + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == (unsigned)cl - 1) {
            for (int g = ch0 / cpg; g < (ch0 + SL) / cpg; ++g) {
                __hip_atomic_store(&red[((int64_t)n * groups + g) * 2 + 0], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&red[((int64_t)n * groups + g) * 2 + 1], 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(arrive + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    float k1[V], k2[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { k1[i] = kk[slot * V + i][0]; k2[i] = kk[slot * V + i][1]; }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        float xv[V], gv[V], ov[V];
        const int64_t p = pix0 + k * ROWS;
        const int64_t o = off + p * c;
        Raw16<T>::unpack(rx[k], xv);
        Raw16<T>::unpack(rg[k], gv);
        if (accumulate) {
            if (add_w) {
                const int64_t row = p / add_w, col = p - row * add_w;
                Vec16<T>::load(add + ((int64_t)n * (hw >> 2) + (row >> 1) * (add_w >> 1) + (col >> 1)) * c + ch0 + slot * V, ov);
#pragma unroll
                for (int i = 0; i < V; ++i) ov[i] *= add_scale;
            } else {
                Vec16<T>::load((add ? add : dx) + o, ov);
            }
        }
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float xh;
            const float g = pre(xv[i], gv[i], i, xh);
            const float r = (g * wv[i] - k1[i] - xh * k2[i]) * rstd[i];
            ov[i] = accumulate ? ov[i] + r : r;
        }
        Vec16<T>::store(dx + o, ov);
    }
}

// deterministic mode, GroupNorm sums left by a conv's drain as ONE SLOT PER TILE (conv_mx.hip: gn_part_nblk): block (sample, group)
// adds its nblk slot pairs in a FIXED order -- lane l: slots l, l + 64, ...; then the xor tree 32 .. 1 -- into sums[(n * G + g) * 2 + j]
__global__ __launch_bounds__(64) void gn_parts_reduce_kernel(const double* __restrict__ parts, int nblk, double* __restrict__ sums) {
    const double* q = parts + (int64_t)blockIdx.x * nblk * 2;
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nblk; k += 64) { a += q[2 * k]; b += q[2 * k + 1]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
    if (threadIdx.x == 0) { sums[(int64_t)blockIdx.x * 2] = a; sums[(int64_t)blockIdx.x * 2 + 1] = b; }
}

// pixels per thread of the small-map form for this problem, or 0 when it does not apply
inline int gn_small_ppt(int dtype, int64_t hw, int c, int groups, int max_ppt) {
    const bool off = VQK_TUNE("GN_NO_SMALL", 0) != 0;
    if (off || c % 32 || (32 % (c / groups))) return 0;
    const int rows = dtype == VQK_F32 ? 32 : 64;
    if (hw % rows) return 0;
    const int64_t ppt = hw / rows;
    if (ppt != 1 && ppt != 2 && ppt != 4 && ppt != 8 && ppt != 16) return 0;
    return ppt <= max_ppt ? (int)ppt : 0;
}

int check_gn(int dtype, int c, int groups) {
    if (dtype != VQK_F32 && dtype != VQK_BF16) return VQK_ERR_DTYPE;
    const int v = dtype == VQK_F32 ? 4 : 8;
    if (c <= 0 || groups <= 0 || c % groups || c % v) return VQK_ERR_SHAPE;
    const int vpp = c / v;
    if (vpp > 256 || 256 % vpp) return VQK_ERR_SHAPE;
    return VQK_OK;
}

// pixels per block.  The streaming (apply) passes want many blocks (~2048); the reducing passes (statistics, backward
// sums) pay per-block LDS + global atomics for every channel / group, so they want fewer, fatter blocks (~768 total):
// measured -8...27 % per reducing pass on the 64^2 / 128^2 maps, -2...5 % on 256^2 (sweep 256...4096 blocks).
inline int pick_ppb(int n, int64_t hw, bool reducing = false) {
    // round 4, re-swept with the pipelined loops (tools/gn_grid_sweep.py): fewer, fatter blocks -- a block now opens with
    // GN_DEPTH loads in flight, so short blocks are mostly prologue.  Reduce 512 (256 on <= 64x64 maps: 128 ch @64^2 backward
    // 73 -> 53 us, 256 ch @64^2 88 -> 75), apply 1024 (64^2 forward apply 29.8 -> 25.8 / 23.3 -> 17.7 us; large maps flat)
    const int tot_r = VQK_TUNE("GN_BLOCKS_REDUCE", hw <= 4096 ? 256 : 512);
    const int tot_a = VQK_TUNE("GN_BLOCKS_APPLY", 1024);
    const int total = reducing ? tot_r : tot_a;
    int64_t blocks_per_sample = (total + n - 1) / n;
    int64_t ppb = (hw + blocks_per_sample - 1) / blocks_per_sample;
    if (ppb < 64) ppb = 64;
    return (int)ppb;
}

}  // namespace

static int gn_backward_impl(int dtype, const void* x, const float* stats, const float* w, const float* b, const void* dy, void* dx,
                            float* dw, float* db, double* red, int n, int64_t hw, int c, int groups, int silu, int accumulate,
                            const void* add, int add_w, float add_scale, void* stream, int64_t ws_doubles = 0,
                            float* dx_colsum = nullptr) {
    VQK_REQUIRE(x && stats && w && b && dy && dx && dw && db && red, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && hw > 0, VQK_ERR_SHAPE);
    const int rc = check_gn(dtype, c, groups);
    if (rc) return rc;
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(dy) && vqk_aligned16(dx), VQK_ERR_ALIGN);
    hipStream_t st = vqk_stream(stream);
    // 16 pixels per thread (the 32^2 maps, re-read form) measured 106 us per call inside the step against 93 for the
    // two-kernel form next to the weight-gradient kernels: the single-kernel backward is used up to 8 pixels per thread
    if (const int ppt = (add_w || dx_colsum) ? 0 : gn_small_ppt(dtype, hw, c, groups, 8)) {
        const dim3 sgrid((unsigned)(c / 32), (unsigned)n);
        const int acc = (accumulate || add) ? 1 : 0;
        vqkd::DetState& det = vqkd::det_state();
        float* cpart = nullptr;
        if (det.on) {
            VQK_REQUIRE(det.ws && (int64_t)n * 2 * c * 4 <= det.bytes, VQK_ERR_WORKSPACE);
            cpart = det.ws;
        }
#define VQK_GN_SMALL_BWD_S(T, P, S) hipLaunchKernelGGL((gn_small_bwd_kernel<T, P, (P < 16), S>), sgrid, dim3(256), 0, st, (const T*)x, stats, w, b, (const T*)dy, (T*)dx, (const T*)add, dw, db, c, groups, silu, acc, cpart)
#define VQK_GN_SMALL_BWD(T, P) do { if (silu) VQK_GN_SMALL_BWD_S(T, P, true); else VQK_GN_SMALL_BWD_S(T, P, false); } while (0)
#define VQK_GN_SMALL_BWD_T(T) do { switch (ppt) { case 1: VQK_GN_SMALL_BWD(T, 1); break; case 2: VQK_GN_SMALL_BWD(T, 2); break; \
        case 4: VQK_GN_SMALL_BWD(T, 4); break; case 8: VQK_GN_SMALL_BWD(T, 8); break; default: VQK_GN_SMALL_BWD(T, 16); } } while (0)
        if (dtype == VQK_F32) VQK_GN_SMALL_BWD_T(float); else VQK_GN_SMALL_BWD_T(bf16_raw);
#undef VQK_GN_SMALL_BWD_T
#undef VQK_GN_SMALL_BWD
#undef VQK_GN_SMALL_BWD_S
        if (cpart) hipLaunchKernelGGL(gn_bwd_finish_kernel, dim3((unsigned)((2 * c + 7) / 8)), dim3(256), 0, st, (const float*)cpart, n, c, dw, db);
        VQK_CHECK_LAUNCH();
        return VQK_OK;
    }
    {
        // cluster form: 64-channel slices (32 when c is no multiple of 64), hw = cl * 8 * rows, workspace with the ticket
        // region (ws_doubles says so)
        const int v = dtype == VQK_F32 ? 4 : 8;
        const int sl = (c % 64 == 0 && 64 % (c / groups) == 0) ? 64 : 32;
        const int rows = 256 / (sl / v);
        const int64_t max_hw = VQK_TUNE("GN_CLUSTER_MAX_HW", 1024);
        vqkd::DetState& det0 = vqkd::det_state();
        if (!det0.on && !dx_colsum && ws_doubles >= (int64_t)n * groups * 2 + n + (int64_t)n * (c / 32) && hw <= max_hw && c % sl == 0 &&
            sl % (c / groups) == 0 && hw % (8 * rows) == 0 && (!add_w || (add_w % 2 == 0 && hw % add_w == 0))) {
            const int cl = (int)(hw / (8 * rows));
            unsigned* tickets = reinterpret_cast<unsigned*>(red + (int64_t)n * groups * 2 + n);
            const dim3 cgrid((unsigned)(c / sl * cl), (unsigned)n);
            const int acc = (accumulate || add) ? 1 : 0;
#define VQK_GN_CL_U(T, S, U) hipLaunchKernelGGL((gn_cluster_bwd_kernel<T, 8, S, U>), cgrid, dim3(256), 0, st, (const T*)x, stats, w, b, (const T*)dy, \
                                           (T*)dx, (const T*)add, dw, db, red, tickets, hw, c, groups, silu, acc, cl, add_w, add_scale)
#define VQK_GN_CL(T, S) do { if (silu) VQK_GN_CL_U(T, S, true); else VQK_GN_CL_U(T, S, false); } while (0)
            if (dtype == VQK_F32) { if (sl == 64) VQK_GN_CL(float, 64); else VQK_GN_CL(float, 32); }
            else { if (sl == 64) VQK_GN_CL(bf16_raw, 64); else VQK_GN_CL(bf16_raw, 32); }
#undef VQK_GN_CL
#undef VQK_GN_CL_U
            VQK_CHECK_LAUNCH();
            return VQK_OK;
        }
    }
    const int ppb = pick_ppb(n, hw), rppb = pick_ppb(n, hw, true);
    const dim3 grid((unsigned)((hw + ppb - 1) / ppb), (unsigned)n), rgrid((unsigned)((hw + rppb - 1) / rppb), (unsigned)n);
    const size_t lds = (size_t)2 * c * sizeof(double) + 256 * 2 * (dtype == VQK_F32 ? 4 : 8) * sizeof(float);
    const int64_t nt_mb = VQK_TUNE("GN_NT_MB", 192);
    const bool nt = (int64_t)n * hw * c * (dtype == VQK_F32 ? 4 : 2) >= (nt_mb << 20);
    // deterministic mode: group partials [n][groups][nblk][2] doubles, then channel partials [n][nblk][2c] floats, in the workspace
    vqkd::DetState& det = vqkd::det_state();
    double* gpart = nullptr;
    float* cpart = nullptr;
    const int nblk = (int)rgrid.x;
    if (det.on) {
        const int64_t gbytes = (int64_t)n * groups * nblk * 2 * 8, cbytes = (int64_t)n * nblk * 2 * c * 4;
        VQK_REQUIRE(det.ws && gbytes + cbytes <= det.bytes, VQK_ERR_WORKSPACE);
        gpart = reinterpret_cast<double*>(det.ws);
        cpart = reinterpret_cast<float*>(reinterpret_cast<char*>(det.ws) + gbytes);
    }
    const int acc = (accumulate || add) ? 1 : 0;
#define VQK_GN_BWD_A(T, S, N, A) hipLaunchKernelGGL((gn_bwd_apply_kernel<T, N, S, A>), grid, dim3(256), 0, st, (const T*)x, stats, w, b, (const T*)dy, (T*)dx, (const T*)add, red, hw, c, groups, silu, acc, ppb, add_w, add_scale, (const double*)gpart, nblk, dx_colsum)
#define VQK_GN_BWD_N(T, S, N) do { if (!acc) VQK_GN_BWD_A(T, S, N, 0); else if (add_w) VQK_GN_BWD_A(T, S, N, 2); else VQK_GN_BWD_A(T, S, N, 1); } while (0)
#define VQK_GN_BWD_S(T, S) do { \
        hipLaunchKernelGGL((gn_bwd_reduce_kernel<T, S>), rgrid, dim3(256), lds, st, (const T*)x, stats, w, b, (const T*)dy, dw, db, red, hw, c, groups, silu, rppb, gpart, cpart); \
        if (nt) VQK_GN_BWD_N(T, S, true); else VQK_GN_BWD_N(T, S, false); \
    } while (0)
#define VQK_GN_BWD(T) do { if (silu) VQK_GN_BWD_S(T, true); else VQK_GN_BWD_S(T, false); } while (0)
    if (dtype == VQK_F32) VQK_GN_BWD(float); else VQK_GN_BWD(bf16_raw);
#undef VQK_GN_BWD
#undef VQK_GN_BWD_S
#undef VQK_GN_BWD_N
#undef VQK_GN_BWD_A
    if (cpart) hipLaunchKernelGGL(gn_bwd_finish_kernel, dim3((unsigned)((2 * c + 7) / 8)), dim3(256), 0, st, (const float*)cpart, n * nblk, c, dw, db);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}


extern "C" {

int vqk_gn_stats(int dtype, const void* x, int n, int64_t hw, int c, int groups, float eps, double* acc, float* stats,
                 void* stream) {
    VQK_REQUIRE(x && acc && stats, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && hw > 0, VQK_ERR_SHAPE);
    const int rc = check_gn(dtype, c, groups);
    if (rc) return rc;
    VQK_REQUIRE(vqk_aligned16(x), VQK_ERR_ALIGN);
    const int ppb = pick_ppb(n, hw, true);
    const dim3 grid((unsigned)((hw + ppb - 1) / ppb), (unsigned)n);
    const size_t lds = (size_t)2 * c * sizeof(double) + 256 * 2 * (dtype == VQK_F32 ? 4 : 8) * sizeof(float);
    hipStream_t st = vqk_stream(stream);
    if (dtype == VQK_F32) hipLaunchKernelGGL(gn_stats_kernel<float>, grid, dim3(256), lds, st, (const float*)x, hw, c, groups, ppb, acc);
    else hipLaunchKernelGGL(gn_stats_kernel<bf16_raw>, grid, dim3(256), lds, st, (const bf16_raw*)x, hw, c, groups, ppb, acc);
    const int total = n * groups;
    hipLaunchKernelGGL(gn_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, st, acc, stats, total,
                       (double)hw * (c / groups), eps);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_gn_apply(int dtype, const void* x, const float* stats, const float* w, const float* b, void* y, int n, int64_t hw,
                 int c, int groups, int silu, void* stream) {
    VQK_REQUIRE(x && stats && w && b && y, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && hw > 0, VQK_ERR_SHAPE);
    const int rc = check_gn(dtype, c, groups);
    if (rc) return rc;
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(y), VQK_ERR_ALIGN);
    const int ppb = pick_ppb(n, hw);
    const dim3 grid((unsigned)((hw + ppb - 1) / ppb), (unsigned)n);
    hipStream_t st = vqk_stream(stream);
    if (dtype == VQK_F32) { if (silu) hipLaunchKernelGGL((gn_apply_kernel<float, true>), grid, dim3(256), 0, st, (const float*)x, stats, w, b, (float*)y, hw, c, groups, silu, ppb); else hipLaunchKernelGGL((gn_apply_kernel<float, false>), grid, dim3(256), 0, st, (const float*)x, stats, w, b, (float*)y, hw, c, groups, silu, ppb); }
    else { if (silu) hipLaunchKernelGGL((gn_apply_kernel<bf16_raw, true>), grid, dim3(256), 0, st, (const bf16_raw*)x, stats, w, b, (bf16_raw*)y, hw, c, groups, silu, ppb); else hipLaunchKernelGGL((gn_apply_kernel<bf16_raw, false>), grid, dim3(256), 0, st, (const bf16_raw*)x, stats, w, b, (bf16_raw*)y, hw, c, groups, silu, ppb); }
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_gn_forward(int dtype, const void* x, const float* w, const float* b, void* y, float* stats, double* ws, int n,
                   int64_t hw, int c, int groups, float eps, int silu, void* stream) {
    VQK_REQUIRE(x && w && b && y && stats && ws, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && hw > 0, VQK_ERR_SHAPE);
    const int rc = check_gn(dtype, c, groups);
    if (rc) return rc;
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(y), VQK_ERR_ALIGN);
    hipStream_t st = vqk_stream(stream);
    if (const int ppt = gn_small_ppt(dtype, hw, c, groups, 16)) {
        const dim3 sgrid((unsigned)(c / 32), (unsigned)n);
#define VQK_GN_SMALL_FWD_S(T, P, S) hipLaunchKernelGGL((gn_small_fwd_kernel<T, P, S>), sgrid, dim3(256), 0, st, (const T*)x, w, b, (T*)y, stats, c, groups, silu, eps)
#define VQK_GN_SMALL_FWD(T, P) do { if (silu) VQK_GN_SMALL_FWD_S(T, P, true); else VQK_GN_SMALL_FWD_S(T, P, false); } while (0)
#define VQK_GN_SMALL_FWD_T(T) do { switch (ppt) { case 1: VQK_GN_SMALL_FWD(T, 1); break; case 2: VQK_GN_SMALL_FWD(T, 2); break; \
        case 4: VQK_GN_SMALL_FWD(T, 4); break; case 8: VQK_GN_SMALL_FWD(T, 8); break; default: VQK_GN_SMALL_FWD(T, 16); } } while (0)
        if (dtype == VQK_F32) VQK_GN_SMALL_FWD_T(float); else VQK_GN_SMALL_FWD_T(bf16_raw);
#undef VQK_GN_SMALL_FWD_T
#undef VQK_GN_SMALL_FWD
#undef VQK_GN_SMALL_FWD_S
        VQK_CHECK_LAUNCH();
        return VQK_OK;
    }
    const int ppb = pick_ppb(n, hw), rppb = pick_ppb(n, hw, true);
    const dim3 grid((unsigned)((hw + ppb - 1) / ppb), (unsigned)n), rgrid((unsigned)((hw + rppb - 1) / rppb), (unsigned)n);
    const size_t lds = (size_t)2 * c * sizeof(double) + 256 * 2 * (dtype == VQK_F32 ? 4 : 8) * sizeof(float);
    vqkd::DetState& det = vqkd::det_state();
    double* part = nullptr;
    const int nblk = (int)rgrid.x;
    if (det.on) {                                            // deterministic mode: block partials + ordered sums, no atomics
        VQK_REQUIRE(det.ws && (int64_t)n * groups * nblk * 2 * 8 <= det.bytes, VQK_ERR_WORKSPACE);
        part = reinterpret_cast<double*>(det.ws);
    }
    if (dtype == VQK_F32) {
        hipLaunchKernelGGL(gn_stats_kernel<float>, rgrid, dim3(256), lds, st, (const float*)x, hw, c, groups, rppb, ws, part);
        { if (silu) hipLaunchKernelGGL((gn_apply_fin_kernel<float, true>), grid, dim3(256), (size_t)groups * 8, st, (const float*)x, ws, stats, w, b, (float*)y, hw, c, groups, silu, ppb, eps, (const double*)part, nblk); else hipLaunchKernelGGL((gn_apply_fin_kernel<float, false>), grid, dim3(256), (size_t)groups * 8, st, (const float*)x, ws, stats, w, b, (float*)y, hw, c, groups, silu, ppb, eps, (const double*)part, nblk); }
    } else {
        hipLaunchKernelGGL(gn_stats_kernel<bf16_raw>, rgrid, dim3(256), lds, st, (const bf16_raw*)x, hw, c, groups, rppb, ws, part);
        { if (silu) hipLaunchKernelGGL((gn_apply_fin_kernel<bf16_raw, true>), grid, dim3(256), (size_t)groups * 8, st, (const bf16_raw*)x, ws, stats, w, b, (bf16_raw*)y, hw, c, groups, silu, ppb, eps, (const double*)part, nblk); else hipLaunchKernelGGL((gn_apply_fin_kernel<bf16_raw, false>), grid, dim3(256), (size_t)groups * 8, st, (const bf16_raw*)x, ws, stats, w, b, (bf16_raw*)y, hw, c, groups, silu, ppb, eps, (const double*)part, nblk); }
    }
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_gn_forward_presummed_parts(int dtype, const void* x, const float* w, const float* b, void* y, float* stats,
                                   const double* parts, int nblk, double* sums, int n, int64_t hw, int c, int groups, float eps,
                                   int silu, void* stream) {
    VQK_REQUIRE(x && w && b && y && stats && parts && sums && nblk > 0, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && hw > 0, VQK_ERR_SHAPE);
    const int rc = check_gn(dtype, c, groups);
    if (rc) return rc;
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(y), VQK_ERR_ALIGN);
    hipStream_t st = vqk_stream(stream);
    hipLaunchKernelGGL(gn_parts_reduce_kernel, dim3((unsigned)(n * groups)), dim3(64), 0, st, parts, nblk, sums);
    const int ppb = pick_ppb(n, hw);
    const dim3 grid((unsigned)((hw + ppb - 1) / ppb), (unsigned)n);
    if (dtype == VQK_F32)
        { if (silu) hipLaunchKernelGGL((gn_apply_fin_kernel<float, true>), grid, dim3(256), (size_t)groups * 8, st, (const float*)x, sums, stats, w, b, (float*)y, hw, c, groups, silu, ppb, eps, (const double*)sums, 1); else hipLaunchKernelGGL((gn_apply_fin_kernel<float, false>), grid, dim3(256), (size_t)groups * 8, st, (const float*)x, sums, stats, w, b, (float*)y, hw, c, groups, silu, ppb, eps, (const double*)sums, 1); }
    else
        { if (silu) hipLaunchKernelGGL((gn_apply_fin_kernel<bf16_raw, true>), grid, dim3(256), (size_t)groups * 8, st, (const bf16_raw*)x, sums, stats, w, b, (bf16_raw*)y, hw, c, groups, silu, ppb, eps, (const double*)sums, 1); else hipLaunchKernelGGL((gn_apply_fin_kernel<bf16_raw, false>), grid, dim3(256), (size_t)groups * 8, st, (const bf16_raw*)x, sums, stats, w, b, (bf16_raw*)y, hw, c, groups, silu, ppb, eps, (const double*)sums, 1); }
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_gn_forward_presummed(int dtype, const void* x, const float* w, const float* b, void* y, float* stats, double* ws,
                             int n, int64_t hw, int c, int groups, float eps, int silu, void* stream) {
    VQK_REQUIRE(x && w && b && y && stats && ws, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && hw > 0, VQK_ERR_SHAPE);
    const int rc = check_gn(dtype, c, groups);
    if (rc) return rc;
    VQK_REQUIRE(vqk_aligned16(x) && vqk_aligned16(y), VQK_ERR_ALIGN);
    hipStream_t st = vqk_stream(stream);
    const int ppb = pick_ppb(n, hw);
    const dim3 grid((unsigned)((hw + ppb - 1) / ppb), (unsigned)n);
    if (dtype == VQK_F32)
        { if (silu) hipLaunchKernelGGL((gn_apply_fin_kernel<float, true>), grid, dim3(256), (size_t)groups * 8, st, (const float*)x, ws, stats, w, b, (float*)y, hw, c, groups, silu, ppb, eps); else hipLaunchKernelGGL((gn_apply_fin_kernel<float, false>), grid, dim3(256), (size_t)groups * 8, st, (const float*)x, ws, stats, w, b, (float*)y, hw, c, groups, silu, ppb, eps); }
    else
        { if (silu) hipLaunchKernelGGL((gn_apply_fin_kernel<bf16_raw, true>), grid, dim3(256), (size_t)groups * 8, st, (const bf16_raw*)x, ws, stats, w, b, (bf16_raw*)y, hw, c, groups, silu, ppb, eps); else hipLaunchKernelGGL((gn_apply_fin_kernel<bf16_raw, false>), grid, dim3(256), (size_t)groups * 8, st, (const bf16_raw*)x, ws, stats, w, b, (bf16_raw*)y, hw, c, groups, silu, ppb, eps); }
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

}  // extern "C"

extern "C" {

int vqk_gn_backward(int dtype, const void* x, const float* stats, const float* w, const float* b, const void* dy, void* dx,
                    float* dw, float* db, double* red, int n, int64_t hw, int c, int groups, int silu, int accumulate,
                    const void* add, void* stream) {
    return gn_backward_impl(dtype, x, stats, w, b, dy, dx, dw, db, red, n, hw, c, groups, silu, accumulate, add, 0, 1.0f, stream);
}

int vqk_gn_backward_ws(int dtype, const void* x, const float* stats, const float* w, const float* b, const void* dy,
                       void* dx, float* dw, float* db, double* red, int64_t ws_doubles, int n, int h, int wd, int c, int groups,
                       int silu, int accumulate, const void* add, const void* add_pooled, float add_scale, void* stream) {
    VQK_REQUIRE(h > 0 && wd > 0 && ws_doubles >= (int64_t)n * groups * 2 + n, VQK_ERR_ARG);
    VQK_REQUIRE(!(add && add_pooled), VQK_ERR_ARG);
    if (add_pooled) {
        VQK_REQUIRE((h % 2) == 0 && (wd % 2) == 0, VQK_ERR_ARG);
        VQK_REQUIRE((int64_t)h * wd > 1024, VQK_ERR_SHAPE);
        return gn_backward_impl(dtype, x, stats, w, b, dy, dx, dw, db, red, n, (int64_t)h * wd, c, groups, silu, 1, add_pooled, wd,
                                add_scale, stream, ws_doubles);
    }
    return gn_backward_impl(dtype, x, stats, w, b, dy, dx, dw, db, red, n, (int64_t)h * wd, c, groups, silu, accumulate, add, 0, 1.0f,
                            stream, ws_doubles);
}

int vqk_gn_backward_colsum(int dtype, const void* x, const float* stats, const float* w, const float* b, const void* dy,
                           void* dx, float* dw, float* db, double* red, int64_t ws_doubles, int n, int h, int wd, int c, int groups,
                           int silu, int accumulate, const void* add, float* dx_colsum, void* stream) {
    VQK_REQUIRE(h > 0 && wd > 0 && ws_doubles >= (int64_t)n * groups * 2 + n && dx_colsum, VQK_ERR_ARG);
    VQK_REQUIRE(vqkd::det_state().on == 0, VQK_ERR_ARG);          // (atomics in arrival order: the ordered mode keeps vqk_colsum)
    return gn_backward_impl(dtype, x, stats, w, b, dy, dx, dw, db, red, n, (int64_t)h * wd, c, groups, silu, accumulate, add, 0, 1.0f,
                            stream, ws_doubles, dx_colsum);
}

int vqk_gn_backward_pooled_add(int dtype, const void* x, const float* stats, const float* w, const float* b, const void* dy,
                               void* dx, float* dw, float* db, double* red, int n, int h, int wd, int c, int groups, int silu,
                               const void* add_pooled, float add_scale, void* stream) {
    VQK_REQUIRE(add_pooled && h > 0 && wd > 0 && (h % 2) == 0 && (wd % 2) == 0, VQK_ERR_ARG);
    VQK_REQUIRE((int64_t)h * wd > 1024, VQK_ERR_SHAPE);       // the two-kernel path (the single-kernel small-map form has no pooled add)
    return gn_backward_impl(dtype, x, stats, w, b, dy, dx, dw, db, red, n, (int64_t)h * wd, c, groups, silu, 1, add_pooled, wd,
                            add_scale, stream);
}

/* diagnostic: number of cluster blocks of the single-kernel GroupNorm backward that gave up their bounded wait since the library
 * was loaded (0 in every healthy run).  Synchronises the device. */
int vqk_gn_cluster_timeouts(int* count) {
    VQK_REQUIRE(count, VQK_ERR_ARG);
    int v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_gn_cluster_timeouts), sizeof(int), 0, hipMemcpyDeviceToHost) != hipSuccess) return VQK_ERR_LAUNCH;
    *count = v;
    return VQK_OK;
}

}  // extern "C"
