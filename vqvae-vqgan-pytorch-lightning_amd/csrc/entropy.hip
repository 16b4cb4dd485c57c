// Entropy (MaskGIT-style) quantizer loss, vqvae/modules/vector_quantizers.py:296-356 of the reference, on the
// materialised fp32 distance matrix d[N][K] (vqk_vq_distances_f32):
//   a = -d/T, p = softmax_k(a), h_i = -sum_k p log p, pbar_k = mean_i p_ik,
//   L_ent = ratio * (mean_i h_i + sum_k pbar_k log(pbar_k + 1e-5))
// and its backward:  u_k = log(pbar_k+1e-5) + pbar_k/(pbar_k+1e-5), ubar_i = sum_k p_ik u_k,
//   dL/dd_ik = -(1/T) * ratio/N * p_ik * ( -(log p_ik + h_i) + (u_k - ubar_i) )
// dz / dE then follow from two fp32 GEMMs that reuse the 1x1 conv kernels (ops.py).
// One wavefront per row for the row passes (K up to tens of thousands), fp32 throughout.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// lse_i = logsumexp_k(a_ik), h_i = lse_i - sum_k p_ik a_ik ; hsum += sum_i h_i
__global__ __launch_bounds__(256) void entropy_rows_kernel(const float* __restrict__ dmat, int64_t n, int k, float inv_t,
                                                           float* __restrict__ lse, float* __restrict__ hrow,
                                                           float* __restrict__ hsum) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const float* dr = dmat + row * k;
    float m = -INFINITY;
    for (int c = lane; c < k; c += 64) m = fmaxf(m, -dr[c] * inv_t);
    m = wave_max(m);
    float s = 0.f, sa = 0.f;
    for (int c = lane; c < k; c += 64) {
        const float a = -dr[c] * inv_t;
        const float ex = __expf(a - m);
        s += ex;
        sa = __fmaf_rn(ex, a, sa);
    }
    s = wave_sum(s);
    sa = wave_sum(sa);
    if (lane == 0) {
        const float l = m + __logf(s);
        const float h = l - sa / s;
        lse[row] = l;
        hrow[row] = h;
        atomicAdd(hsum, h);
    }
}

// psum[k] += sum over a slab of rows of p_ik  (thread = column, loop over the slab's rows)
__global__ __launch_bounds__(256) void entropy_colsum_kernel(const float* __restrict__ dmat, const float* __restrict__ lse,
                                                             int64_t n, int k, float inv_t, int rows_per_block,
                                                             float* __restrict__ psum) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= k) return;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(n, r0 + rows_per_block);
    float acc = 0.f;
    for (int64_t r = r0; r < r1; ++r) acc += __expf(-dmat[r * k + col] * inv_t - lse[r]);
    atomicAdd(psum + col, acc);
}

// pbar = psum/N ; u_k ; loss_terms[0] = sum_k pbar log(pbar + eps)
__global__ __launch_bounds__(256) void entropy_finalize_kernel(const float* __restrict__ psum, int k, float inv_n,
                                                               float* __restrict__ u, float* __restrict__ avg_term) {
    __shared__ float part[4];
    float acc = 0.f;
    for (int c = blockIdx.x * 256 + threadIdx.x; c < k; c += gridDim.x * 256) {
        const float pb = psum[c] * inv_n;
        const float lg = __logf(pb + 1e-5f);
        acc = __fmaf_rn(pb, lg, acc);
        u[c] = lg + pb / (pb + 1e-5f);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(avg_term, (part[0] + part[1]) + (part[2] + part[3]));
}

// in place: d_ik <- dd_ik = coef * p_ik * ( -(log p_ik + h_i) + (u_k - ubar_i) ),  coef = -gs*ratio/(N*T)
__global__ __launch_bounds__(256) void entropy_dd_kernel(float* __restrict__ dmat, const float* __restrict__ lse,
                                                         const float* __restrict__ hrow, const float* __restrict__ u,
                                                         int64_t n, int k, float inv_t, float coef,
                                                         const float* __restrict__ gs) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    if (gs) coef *= *gs;
    float* dr = dmat + row * k;
    const float l = lse[row], h = hrow[row];
    float ub = 0.f;
    for (int c = lane; c < k; c += 64) ub = __fmaf_rn(__expf(-dr[c] * inv_t - l), u[c], ub);
    ub = wave_sum(ub);
    for (int c = lane; c < k; c += 64) {
        const float lp = -dr[c] * inv_t - l;
        const float p = __expf(lp);
        dr[c] = coef * p * (-(lp + h) + (u[c] - ub));
    }
}

// the same cotangent written as TWO bf16 matrices hi + lo (hi = bf16(dd), lo = bf16(dd - hi): 16 significant bits, the bytes of the
// fp32 matrix): the operands of the split-product GEMMs  dd E ~ hi E_hi + hi E_lo + lo E_hi  on the bf16 MFMA kernels (ops.py)
__global__ __launch_bounds__(256) void entropy_dd_split_kernel(const float* __restrict__ dmat, const float* __restrict__ lse,
                                                               const float* __restrict__ hrow, const float* __restrict__ u,
                                                               int64_t n, int k, float inv_t, float coef, const float* __restrict__ gs,
                                                               bf16_raw* __restrict__ hi, bf16_raw* __restrict__ lo) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    if (gs) coef *= *gs;
    const float* dr = dmat + row * k;
    const float l = lse[row], h = hrow[row];
    float ub = 0.f;
    for (int c = lane; c < k; c += 64) ub = __fmaf_rn(__expf(-dr[c] * inv_t - l), u[c], ub);      // (same order as entropy_dd_kernel)
    ub = wave_sum(ub);
    unsigned* hr = reinterpret_cast<unsigned*>(hi + row * k);
    unsigned* lr = reinterpret_cast<unsigned*>(lo + row * k);
    for (int c = 2 * lane; c < k; c += 128) {                    // a lane owns two neighbouring codes: 4-byte stores (k % 4 == 0)
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float lp = -dr[c + e] * inv_t - l;
            const float pr = __expf(lp);
            v[e] = coef * pr * (-(lp + h) + (u[c + e] - ub));
        }
        const bf16_raw h0 = f32_to_bf16(v[0]), h1 = f32_to_bf16(v[1]);
        const bf16_raw l0 = f32_to_bf16(v[0] - bf16_to_f32(h0)), l1 = f32_to_bf16(v[1] - bf16_to_f32(h1));
        hr[c >> 1] = (unsigned)h0 | ((unsigned)h1 << 16);
        lr[c >> 1] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
}

// ---- ent_loss_type == 'argmax' (vector_quantizers.py:311-315): target = one_hot(argmax a) with the gradient of p.
//   L = mean_i ( lse_i - a_i[c_i] ) + sum_k m_k log(m_k + 1e-5),  m_k = hist_k / N
//   dL/da_ij = (1/N) [ p_ij ( -(log p_ij + h_i) + 1 + (u_j - ubar_i) ) - [j == c_i] ],  u from m as above
// ssum += sum_i ( lse_i + d[i][c_i] / T )
__global__ __launch_bounds__(256) void entropy_argmax_rows_kernel(const float* __restrict__ dmat, const int64_t* __restrict__ idx,
                                                                  const float* __restrict__ lse, int64_t n, int k, float inv_t,
                                                                  float* __restrict__ ssum) {
    __shared__ float part[4];
    float acc = 0.f;
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < n; r += (int64_t)gridDim.x * 256)
        acc += lse[r] + dmat[r * k + idx[r]] * inv_t;
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(ssum, (part[0] + part[1]) + (part[2] + part[3]));
}

__global__ __launch_bounds__(256) void hist_to_float_kernel(const int32_t* __restrict__ hist, int k, float* __restrict__ out) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c < k) out[c] = (float)hist[c];
}

__global__ __launch_bounds__(256) void entropy_dd_argmax_kernel(float* __restrict__ dmat, const int64_t* __restrict__ idx,
                                                                const float* __restrict__ lse, const float* __restrict__ hrow,
                                                                const float* __restrict__ u, int64_t n, int k, float inv_t,
                                                                float coef, const float* __restrict__ gs) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    if (gs) coef *= *gs;
    float* dr = dmat + row * k;
    const float l = lse[row], h = hrow[row];
    const int code = (int)idx[row];
    float ub = 0.f;
    for (int c = lane; c < k; c += 64) ub = __fmaf_rn(__expf(-dr[c] * inv_t - l), u[c], ub);
    ub = wave_sum(ub);
    for (int c = lane; c < k; c += 64) {
        const float lp = -dr[c] * inv_t - l;
        const float p = __expf(lp);
        dr[c] = coef * (p * (-(lp + h) + 1.0f + (u[c] - ub)) - (c == code ? 1.0f : 0.0f));
    }
}

// out[r][c] += a * scale[r] * m[r][c]
__global__ __launch_bounds__(256) void row_scale_add_kernel(float* __restrict__ out, const float* __restrict__ m,
                                                            const float* __restrict__ scale, int64_t rows, int c,
                                                            float a) {
    const int64_t total = rows * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256)
        out[i] = __fmaf_rn(a * scale[i / c], m[i], out[i]);
}

}  // namespace

extern "C" {

int vqk_entropy_forward_f32(const float* dmat, int64_t n, int k, float temperature, float* lse, float* hrow, float* hsum,
                            float* psum, float* u, float* avg_term, void* stream) {
    VQK_REQUIRE(dmat && lse && hrow && hsum && psum && u && avg_term, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && k > 0 && temperature > 0.f, VQK_ERR_SHAPE);
    hipStream_t st = vqk_stream(stream);
    const float inv_t = 1.0f / temperature;
    hipLaunchKernelGGL(entropy_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, dmat, n, k, inv_t, lse, hrow, hsum);
    int rpb = (int)((n + 255) / 256); if (rpb < 16) rpb = 16;
    const dim3 grid((unsigned)((k + 255) / 256), (unsigned)((n + rpb - 1) / rpb));
    hipLaunchKernelGGL(entropy_colsum_kernel, grid, dim3(256), 0, st, dmat, lse, n, k, inv_t, rpb, psum);
    hipLaunchKernelGGL(entropy_finalize_kernel, dim3(vqk_grid_1d(k, 256, 64)), dim3(256), 0, st, psum, k, 1.0f / (float)n, u, avg_term);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

// vqk_entropy_forward_f32 when lse / hrow / hsum were produced by vqk_vq_distances_stats_f32: the column pass and the finalize only
int vqk_entropy_forward_presummed_f32(const float* dmat, int64_t n, int k, float temperature, const float* lse, float* psum,
                                      float* u, float* avg_term, void* stream) {
    VQK_REQUIRE(dmat && lse && psum && u && avg_term, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && k > 0 && temperature > 0.f, VQK_ERR_SHAPE);
    hipStream_t st = vqk_stream(stream);
    const float inv_t = 1.0f / temperature;
    int rpb = (int)((n + 255) / 256); if (rpb < 16) rpb = 16;
    const dim3 grid((unsigned)((k + 255) / 256), (unsigned)((n + rpb - 1) / rpb));
    hipLaunchKernelGGL(entropy_colsum_kernel, grid, dim3(256), 0, st, dmat, lse, n, k, inv_t, rpb, psum);
    hipLaunchKernelGGL(entropy_finalize_kernel, dim3(vqk_grid_1d(k, 256, 64)), dim3(256), 0, st, psum, k, 1.0f / (float)n, u, avg_term);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_entropy_backward_f32(float* dmat, const float* lse, const float* hrow, const float* u, int64_t n, int k,
                             float temperature, float ratio, const float* gscale_dev, void* stream) {
    VQK_REQUIRE(dmat && lse && hrow && u, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && k > 0 && temperature > 0.f, VQK_ERR_SHAPE);
    const float coef = -ratio / ((float)n * temperature);
    hipLaunchKernelGGL(entropy_dd_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, vqk_stream(stream), dmat, lse, hrow, u,
                       n, k, 1.0f / temperature, coef, gscale_dev);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_entropy_backward_split_f32(const float* dmat, const float* lse, const float* hrow, const float* u, int64_t n, int k,
                                   float temperature, float ratio, const float* gscale_dev, void* hi, void* lo, void* stream) {
    VQK_REQUIRE(dmat && lse && hrow && u && hi && lo, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && k > 0 && (k & 3) == 0 && temperature > 0.f, VQK_ERR_SHAPE);
    const float coef = -ratio / ((float)n * temperature);
    hipLaunchKernelGGL(entropy_dd_split_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, vqk_stream(stream), dmat, lse, hrow, u,
                       n, k, 1.0f / temperature, coef, gscale_dev, (bf16_raw*)hi, (bf16_raw*)lo);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_entropy_argmax_forward_f32(const float* dmat, const int64_t* idx, const int32_t* hist, int64_t n, int k,
                                   float temperature, float* lse, float* hrow, float* hsum, float* ssum, float* mbuf,
                                   float* u, float* avg_term, void* stream) {
    VQK_REQUIRE(dmat && idx && hist && lse && hrow && hsum && ssum && mbuf && u && avg_term, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && k > 0 && temperature > 0.f, VQK_ERR_SHAPE);
    hipStream_t st = vqk_stream(stream);
    const float inv_t = 1.0f / temperature;
    hipLaunchKernelGGL(entropy_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, dmat, n, k, inv_t, lse, hrow, hsum);
    hipLaunchKernelGGL(entropy_argmax_rows_kernel, dim3(vqk_grid_1d(n, 256, 256)), dim3(256), 0, st, dmat, idx, lse, n, k, inv_t, ssum);
    hipLaunchKernelGGL(hist_to_float_kernel, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, st, hist, k, mbuf);
    hipLaunchKernelGGL(entropy_finalize_kernel, dim3(vqk_grid_1d(k, 256, 64)), dim3(256), 0, st, mbuf, k, 1.0f / (float)n, u, avg_term);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_entropy_argmax_backward_f32(float* dmat, const int64_t* idx, const float* lse, const float* hrow, const float* u,
                                    int64_t n, int k, float temperature, float ratio, const float* gscale_dev,
                                    void* stream) {
    VQK_REQUIRE(dmat && idx && lse && hrow && u, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && k > 0 && temperature > 0.f, VQK_ERR_SHAPE);
    const float coef = -ratio / ((float)n * temperature);
    hipLaunchKernelGGL(entropy_dd_argmax_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, vqk_stream(stream), dmat, idx, lse,
                       hrow, u, n, k, 1.0f / temperature, coef, gscale_dev);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_row_scale_add_f32(float* out, const float* m, const float* scale, int64_t rows, int c, float a, void* stream) {
    VQK_REQUIRE(out && m && scale, VQK_ERR_ARG);
    VQK_REQUIRE(rows >= 0 && c > 0, VQK_ERR_SHAPE);
    if (rows == 0) return VQK_OK;
    hipLaunchKernelGGL(row_scale_add_kernel, dim3(vqk_grid_1d(rows * c, 256)), dim3(256), 0, vqk_stream(stream), out, m, scale, rows, c, a);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// Gumbel-softmax quantizer rows, vqvae/modules/vector_quantizers.py:223-245 of the reference:
//   y = softmax((logits + g)/tau), g = -log(noise), noise ~ Exp(1) (drawn by the caller);
//   qy = softmax(logits); kl_i = sum_n qy log(qy*K + 1e-10); idx = argmax y (first maximum).
// backward: dlogits = y (dy - sum y dy)/tau + (klc/M) qy (r - sum qy r),
//   r = log(qy K + 1e-10) + qy K / (qy K + 1e-10).
// One wavefront per row; logits fp32; y / dy in the GEMM dtype T.
// ------------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256) void gumbel_rows_fwd_kernel(const float* __restrict__ logits,
                                                              const float* __restrict__ noise, int64_t n, int k,
                                                              float inv_tau, int hard, T* __restrict__ y,
                                                              int64_t* __restrict__ idx, float* __restrict__ klsum,
                                                              int32_t* __restrict__ hist, const float* __restrict__ sched) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    if (sched) inv_tau = 1.0f / sched[0];                      // scheduled temperature read on the device (graph replay)
    const float* lr = logits + row * k;
    const float* nr = noise + row * k;
    float m1 = -INFINITY, m2 = -INFINITY;
    int am = 0x7fffffff;
    for (int c = lane; c < k; c += 64) {
        const float t = (lr[c] - __logf(nr[c])) * inv_tau;
        if (t > m1) { m1 = t; am = c; }
        m2 = fmaxf(m2, lr[c]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float o = __shfl_xor(m1, off, 64);
        const int oi = __shfl_xor(am, off, 64);
        if (o > m1 || (o == m1 && oi < am)) { m1 = o; am = oi; }
    }
    m2 = wave_max(m2);
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < k; c += 64) {
        s1 += __expf((lr[c] - __logf(nr[c])) * inv_tau - m1);
        s2 += __expf(lr[c] - m2);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const float r1 = 1.0f / s1, r2 = 1.0f / s2;
    float kl = 0.f;
    for (int c = lane; c < k; c += 64) {
        const float yv = __expf((lr[c] - __logf(nr[c])) * inv_tau - m1) * r1;
        const float qy = __expf(lr[c] - m2) * r2;
        kl = __fmaf_rn(qy, __logf(qy * (float)k + 1e-10f), kl);
        Elem<T>::st(y + row * k + c, hard ? (c == am ? 1.0f : 0.0f) : yv);
    }
    kl = wave_sum(kl);
    if (lane == 0) { idx[row] = am; atomicAdd(klsum, kl); if (hist) atomicAdd(hist + am, 1); }
}

template <typename T>
__global__ __launch_bounds__(256) void gumbel_rows_bwd_kernel(const float* __restrict__ logits,
                                                              const float* __restrict__ noise, const T* __restrict__ dy,
                                                              int64_t n, int k, float inv_tau, float klc_over_m,
                                                              const float* __restrict__ gs, float* __restrict__ dlogits,
                                                              const float* __restrict__ sched) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    if (sched) { inv_tau = 1.0f / sched[0]; klc_over_m = sched[1] / (float)n; }
    if (gs) klc_over_m *= *gs;
    const float* lr = logits + row * k;
    const float* nr = noise + row * k;
    const T* dr = dy + row * k;
    float m1 = -INFINITY, m2 = -INFINITY;
    for (int c = lane; c < k; c += 64) {
        m1 = fmaxf(m1, (lr[c] - __logf(nr[c])) * inv_tau);
        m2 = fmaxf(m2, lr[c]);
    }
    m1 = wave_max(m1); m2 = wave_max(m2);
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < k; c += 64) {
        s1 += __expf((lr[c] - __logf(nr[c])) * inv_tau - m1);
        s2 += __expf(lr[c] - m2);
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const float r1 = 1.0f / s1, r2 = 1.0f / s2;
    float ydy = 0.f, qr = 0.f;
    for (int c = lane; c < k; c += 64) {
        const float yv = __expf((lr[c] - __logf(nr[c])) * inv_tau - m1) * r1;
        const float qy = __expf(lr[c] - m2) * r2;
        const float qk = qy * (float)k;
        ydy = __fmaf_rn(yv, Elem<T>::ld(dr + c), ydy);
        qr = __fmaf_rn(qy, __logf(qk + 1e-10f) + qk / (qk + 1e-10f), qr);
    }
    ydy = wave_sum(ydy); qr = wave_sum(qr);
    for (int c = lane; c < k; c += 64) {
        const float yv = __expf((lr[c] - __logf(nr[c])) * inv_tau - m1) * r1;
        const float qy = __expf(lr[c] - m2) * r2;
        const float qk = qy * (float)k;
        const float r = __logf(qk + 1e-10f) + qk / (qk + 1e-10f);
        dlogits[row * k + c] = yv * (Elem<T>::ld(dr + c) - ydy) * inv_tau + klc_over_m * qy * (r - qr);
    }
}

}  // namespace

extern "C" {

int vqk_gumbel_forward(int dtype, const float* logits, const float* noise, int64_t n, int k, float tau, int hard, void* y,
                       int64_t* idx, float* klsum, int32_t* hist, const float* sched_dev, void* stream) {
    VQK_REQUIRE(logits && noise && y && idx && klsum, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && k > 0 && tau > 0.f, VQK_ERR_SHAPE);
    const dim3 grid((unsigned)((n + 3) / 4));
    if (dtype == VQK_F32) hipLaunchKernelGGL(gumbel_rows_fwd_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), logits, noise, n, k, 1.0f / tau, hard, (float*)y, idx, klsum, hist, sched_dev);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(gumbel_rows_fwd_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), logits, noise, n, k, 1.0f / tau, hard, (bf16_raw*)y, idx, klsum, hist, sched_dev);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_gumbel_backward(int dtype, const float* logits, const float* noise, const void* dy, int64_t n, int k, float tau,
                        float kl_cost, const float* gscale_dev, float* dlogits, const float* sched_dev, void* stream) {
    VQK_REQUIRE(logits && noise && dy && dlogits, VQK_ERR_ARG);
    VQK_REQUIRE(n > 0 && k > 0 && tau > 0.f, VQK_ERR_SHAPE);
    const dim3 grid((unsigned)((n + 3) / 4));
    const float c = kl_cost / (float)n;
    if (dtype == VQK_F32) hipLaunchKernelGGL(gumbel_rows_bwd_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), logits, noise, (const float*)dy, n, k, 1.0f / tau, c, gscale_dev, dlogits, sched_dev);
    else if (dtype == VQK_BF16) hipLaunchKernelGGL(gumbel_rows_bwd_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), logits, noise, (const bf16_raw*)dy, n, k, 1.0f / tau, c, gscale_dev, dlogits, sched_dev);
    else return VQK_ERR_DTYPE;
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

}  // extern "C"
