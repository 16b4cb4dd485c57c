// Vector-quantizer kernels: nearest-codeword assignment on exact-fp32 MFMA, gather/loss, backward,
// EMA statistics and update.  Restates vqvae/modules/vector_quantizers.py:33-56, :142-172, :337-350
// of the reference (see include/vqk.h); the arithmetic order of the assignment is the canonical one
// documented in oracle/vq_oracle.c, so indices are bit-exact against the oracle by construction.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// |row|^2 : one wavefront per row.  lane l: fma chain over l, l+64, ...; then xor butterfly 32..1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, int64_t rows, int d,
                                                         float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = x + row * d;
    float acc = 0.0f;
    for (int k = lane; k < d; k += 64) acc = __fmaf_rn(p[k], p[k], acc);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc = __fadd_rn(acc, __shfl_xor(acc, off, 64));
    if (lane == 0) out[row] = acc;
}

// ------------------------------------------------------------------------------------------------
// Assignment.  Block = 4 waves = 32 z rows x all K codes; the z rows sit in LDS (row stride D+4
// floats: conflict-free ds_read_b128), every wave walks a contiguous quarter of the 32-code tiles.
// MFMA roles: A = codes (rows i), B = z rows (cols j)  ->  each lane owns one z row (j = lane&31)
// and 16 codes of the tile, so the running (min, argmin) is lane-local; k is consumed in the order
// 8m+{0,4,1,5,2,6,3,7} because lane (., half) holds the float4 at k = 8m + 4*half.
// ------------------------------------------------------------------------------------------------
template <int ASSOC, bool WRITE_D>
__global__ __launch_bounds__(256) void vq_assign_kernel(const float* __restrict__ z, const float* __restrict__ e,
                                                        const float* __restrict__ z2, const float* __restrict__ e2,
                                                        int64_t n, int k, int d, int64_t* __restrict__ idx,
                                                        float* __restrict__ dmat) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* zt = reinterpret_cast<float*>(smem);                 // [32][d + 4]
    const int ld = d + 4;
    float* red_d = zt + 32 * ld;                                // [4][32]
    int* red_i = reinterpret_cast<int*>(red_d + 128);           // [4][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 32;

    // stage the 32 z rows (rows past n are clamped: computed, never stored)
    const int vec_per_row = d >> 2;
    for (int v = tid; v < 32 * vec_per_row; v += 256) {
        const int r = v / vec_per_row, c = v - r * vec_per_row;
        int64_t src = n0 + r; if (src >= n) src = n - 1;
        *reinterpret_cast<f32x4*>(zt + r * ld + 4 * c) = *reinterpret_cast<const f32x4*>(z + src * d + 4 * c);
    }
    __syncthreads();

    const int j = lane & 31, half = lane >> 5;
    int64_t zrow = n0 + j; if (zrow >= n) zrow = n - 1;
    const float zz = z2[zrow];
    const float* zb = zt + j * ld + 4 * half;

    const int tiles = (k + 31) >> 5;
    const int per_wave = (tiles + 3) >> 2;
    const int t_begin = wave * per_wave;
    const int t_end = min(tiles, t_begin + per_wave);

    float best = INFINITY;
    int best_i = 0x7fffffff;
    for (int t = t_begin; t < t_end; ++t) {
        int code_row = t * 32 + j; if (code_row >= k) code_row = k - 1;       // A row i == lane&31
        const float* ea = e + (int64_t)code_row * d + 4 * half;
        f32x16 acc = {0};
#pragma unroll 4
        for (int m = 0; m < d; m += 8) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ea + m);
            const f32x4 b = *reinterpret_cast<const f32x4*>(zb + m);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int code = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (code < k) {
                const float ab2 = 2.0f * acc[r];
                float dist;
                if (ASSOC == 0) dist = __fsub_rn(__fadd_rn(zz, e2[code]), ab2);
                else            dist = __fadd_rn(__fsub_rn(zz, ab2), e2[code]);
                if (dist < best) { best = dist; best_i = code; }
                if (WRITE_D && n0 + j < n) dmat[(n0 + j) * (int64_t)k + code] = dist;
            }
        }
    }
    // the two half-waves hold disjoint code subsets of the same z row
    {
        const float od = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(best_i, 32, 64);
        if (od < best || (od == best && oi < best_i)) { best = od; best_i = oi; }
    }
    if (half == 0) { red_d[wave * 32 + j] = best; red_i[wave * 32 + j] = best_i; }
    __syncthreads();
    if (tid < 32 && n0 + tid < n) {
        float bd = red_d[tid]; int bi = red_i[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float od = red_d[w * 32 + tid]; const int oi = red_i[w * 32 + tid];
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        idx[n0 + tid] = (bi == 0x7fffffff) ? 0 : (int64_t)bi;
    }
}

// ------------------------------------------------------------------------------------------------
// Register-resident form for D == DD (256 in every reference config): one wave per SIMD with up to 512 registers, so
// the lane's slice of its z row (DD/2 floats) stays in registers for the whole kernel and the code fragments are
// double-buffered half tiles (the loads of the next 64 MFMAs are in flight during the current 64).  Same MFMA
// sequence, same k order, same comparisons as vq_assign_kernel => identical indices.
// ------------------------------------------------------------------------------------------------
// STATS (Entropy quantizer, vector_quantizers.py:296-310): the row statistics of the softmax over a = -d / T ride along as an
// ONLINE log-sum-exp per lane (running max m, s = sum exp(a - m), sa = sum exp(a - m) a; one rescale per 32-code tile), merged
// over the row's eight partial states at the end: lse_i = m + log s, h_i = lse_i - sa / s, hsum += h_i -- the separate pass
// over the [N][K] matrix (entropy_rows_kernel: 0.36 ms at N = 16,384, K = 8,192) disappears under the fp32 MFMAs.
#ifndef VQK_ENT_CHAINS
#define VQK_ENT_CHAINS 1      // accumulation chains of the Entropy quantizer's distance matrix (1: the canonical single chain).
                              // 4 was MEASURED (round 5, tools/entropy_tol_probe.py): the codebook gradient at T = 0.01 moved from
                              // 7.1e-3 to 1.4e-2 off the reference's fixture -- the chain length is not what separates the two fp32
                              // results (the cancellation between -2 dd^T Z and 2 E colsum(dd) is: both sides carry it); 1 stays
#endif
template <int ASSOC, bool WRITE_D, int DD, bool STATS = false, int CH = 1>
__global__ __launch_bounds__(256) void vq_assign_reg_kernel(const float* __restrict__ z, const float* __restrict__ e,
                                                            const float* __restrict__ z2, const float* __restrict__ e2,
                                                            int64_t n, int k, int64_t* __restrict__ idx,
                                                            float* __restrict__ dmat, float inv_t = 0.0f,
                                                            float* __restrict__ lse = nullptr, float* __restrict__ hrow = nullptr,
                                                            float* __restrict__ hsum = nullptr) {
    constexpr int NF = DD / 8, HF = NF / 2;                      // float4 fragments per row slice / per half tile
    __shared__ float red_d[128];
    __shared__ int red_i[128];
    __shared__ float red_s[STATS ? 3 * 128 : 1];
    float m_run = -INFINITY, s_run = 0.0f, sa_run = 0.0f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 32;
    const int j = lane & 31, half = lane >> 5;
    int64_t zrow = n0 + j; if (zrow >= n) zrow = n - 1;
    const float zz = z2[zrow];
    f32x4 zr[NF];
    {
        const float* zp = z + zrow * DD + 4 * half;
#pragma unroll
        for (int i = 0; i < NF; ++i) zr[i] = *reinterpret_cast<const f32x4*>(zp + 8 * i);
    }
    const int tiles = (k + 31) >> 5;
    const int per_wave = (tiles + 3) >> 2;
    const int t_begin = wave * per_wave;
    const int t_end = min(tiles, t_begin + per_wave);
    auto code_ptr = [&](int t) -> const float* {
        int code_row = t * 32 + j; if (code_row >= k) code_row = k - 1;
        return e + (int64_t)code_row * DD + 4 * half;
    };
    f32x4 ab[2][HF];
    if (t_begin < t_end) {
        const float* ea = code_ptr(t_begin);
#pragma unroll
        for (int i = 0; i < HF; ++i) ab[0][i] = *reinterpret_cast<const f32x4*>(ea + 8 * i);
    }
    float best = INFINITY;
    int best_i = 0x7fffffff;
    for (int t = t_begin; t < t_end; ++t) {
        const float* ea = code_ptr(t);
        const float* en = code_ptr(t + 1 < t_end ? t + 1 : t);
#pragma unroll
        for (int i = 0; i < HF; ++i) ab[1][i] = *reinterpret_cast<const f32x4*>(ea + 8 * (HF + i));
        float e2v[16];                                           // |e|^2 of this tile's codes, loaded under the MFMAs
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int code = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            e2v[r] = e2[code < k ? code : k - 1];
        }
        // CH accumulation chains over the 256 products of a distance.  CH = 1: ONE sequential chain -- the canonical order of
        // oracle/vq_oracle.c (the plain assignment: indices bit-exact by construction).  CH = 4 (the Entropy quantizer's distance
        // MATRIX): four chains of 32 MFMAs summed as (c0 + c1) + (c2 + c3) -- a chain's rounding error grows with its length, and at
        // T = 0.01 the softmax multiplies the error of a distance by 100.  Measured: no closer to the reference's fixture (see
        // VQK_ENT_CHAINS), so the shipped build keeps CH = 1 everywhere
        f32x16 acc = {0};
        f32x16 accx[CH > 1 ? CH - 1 : 1];
        if constexpr (CH > 1) {
#pragma unroll
            for (int q = 0; q < CH - 1; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) accx[q][r] = 0.0f;
        }
        auto chain = [&](int i) -> f32x16& {                     // fragment i of NF -> its chain
            if constexpr (CH > 1) { const int q = i / (NF / CH); return q == 0 ? acc : accx[q - 1]; }
            else return acc;
        };
#pragma unroll
        for (int i = 0; i < HF; ++i) {
            f32x16& a_ = chain(i);
            a_ = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[0][i][0], zr[i][0], a_, 0, 0, 0);
            a_ = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[0][i][1], zr[i][1], a_, 0, 0, 0);
            a_ = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[0][i][2], zr[i][2], a_, 0, 0, 0);
            a_ = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[0][i][3], zr[i][3], a_, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < HF; ++i) ab[0][i] = *reinterpret_cast<const f32x4*>(en + 8 * i);
#pragma unroll
        for (int i = 0; i < HF; ++i) {
            f32x16& a_ = chain(HF + i);
            a_ = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[1][i][0], zr[HF + i][0], a_, 0, 0, 0);
            a_ = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[1][i][1], zr[HF + i][1], a_, 0, 0, 0);
            a_ = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[1][i][2], zr[HF + i][2], a_, 0, 0, 0);
            a_ = __builtin_amdgcn_mfma_f32_32x32x2f32(ab[1][i][3], zr[HF + i][3], a_, 0, 0, 0);
        }
        if constexpr (CH == 4) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = __fadd_rn(__fadd_rn(acc[r], accx[0][r]), __fadd_rn(accx[1][r], accx[2][r]));
        } else if constexpr (CH == 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = __fadd_rn(acc[r], accx[0][r]);
        }
        __builtin_amdgcn_sched_barrier(0);
        float av[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int code = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            av[r] = -INFINITY;
            if (code < k) {
                const float ab2 = 2.0f * acc[r];
                float dist;
                if (ASSOC == 0) dist = __fsub_rn(__fadd_rn(zz, e2v[r]), ab2);
                else            dist = __fadd_rn(__fsub_rn(zz, ab2), e2v[r]);
                if (dist < best) { best = dist; best_i = code; }
                if (WRITE_D && n0 + j < n) dmat[(n0 + j) * (int64_t)k + code] = dist;
                if (STATS) av[r] = -dist * inv_t;
            }
        }
        if (STATS) {
            float tmax = av[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, av[r]);
            const float mn = fmaxf(m_run, tmax);
            if (mn > -INFINITY) {                                // (a lane whose codes are all beyond K has nothing to add)
                const float sc = __expf(m_run - mn);             // exp(-inf) = 0 on the first tile
                float ts = 0.0f, tsa = 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float ex = __expf(av[r] - mn);         // 0 for the padded entries
                    ts += ex;
                    tsa = __fmaf_rn(ex, av[r] > -INFINITY ? av[r] : 0.0f, tsa);
                }
                s_run = __fmaf_rn(s_run, sc, ts);
                sa_run = __fmaf_rn(sa_run, sc, tsa);
                m_run = mn;
            }
        }
    }
    if (STATS) {
        // merge the row's partial states: the two half-waves, then the four waves
        auto merge = [](float& m, float& s_, float& sa, float om, float os, float osa) {
            const float mn = fmaxf(m, om);
            if (mn > -INFINITY) {
                const float c0 = __expf(m - mn), c1 = __expf(om - mn);
                s_ = s_ * c0 + os * c1;
                sa = sa * c0 + osa * c1;
                m = mn;
            }
        };
        merge(m_run, s_run, sa_run, __shfl_xor(m_run, 32, 64), __shfl_xor(s_run, 32, 64), __shfl_xor(sa_run, 32, 64));
        if (half == 0) { red_s[wave * 32 + j] = m_run; red_s[128 + wave * 32 + j] = s_run; red_s[256 + wave * 32 + j] = sa_run; }
    }
    {
        const float od = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(best_i, 32, 64);
        if (od < best || (od == best && oi < best_i)) { best = od; best_i = oi; }
    }
    if (half == 0) { red_d[wave * 32 + j] = best; red_i[wave * 32 + j] = best_i; }
    __syncthreads();
    if (tid < 32 && n0 + tid < n) {
        float bd = red_d[tid]; int bi = red_i[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float od = red_d[w * 32 + tid]; const int oi = red_i[w * 32 + tid];
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        idx[n0 + tid] = (bi == 0x7fffffff) ? 0 : (int64_t)bi;
        if (STATS) {
            float m = red_s[tid], s_ = red_s[128 + tid], sa = red_s[256 + tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float om = red_s[w * 32 + tid], os = red_s[128 + w * 32 + tid], osa = red_s[256 + w * 32 + tid];
                const float mn = fmaxf(m, om);
                if (mn > -INFINITY) {
                    const float c0 = __expf(m - mn), c1 = __expf(om - mn);
                    s_ = s_ * c0 + os * c1; sa = sa * c0 + osa * c1; m = mn;
                }
            }
            const float l = m + __logf(s_);
            const float h = l - sa / s_;
            lse[n0 + tid] = l;
            hrow[n0 + tid] = h;
            atomicAdd(hsum, h);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Code tiles through LDS (round 5; the Entropy quantizer's distance matrix, K = 8192): block = 128 z rows (four waves, 32 rows each
// in registers as above) x ONE part of the 32-code tiles (blockIdx.y of KS parts).  The four waves walk the SAME tiles: a tile is
// fetched ONCE per block by LDS-DMA (one 1-KiB instruction per code row, 8 rows per wave; row pitch 1040 bytes: the lanes' 16-byte
// fragment reads hit disjoint banks), double-buffered (65 KiB: two blocks per CU = two waves per SIMD), and every wave reads its
// MFMA operands from it -- a quarter of the codebook traffic of the register-streaming kernel (where every wave pulls the whole
// codebook through the vector-memory path, 4.3 GB per launch at N = 16,384, K = 8,192, with ONE wave per SIMD to hide it: 62 TF),
// the operand fetch on the LDS latency.  Same MFMA sequence per (row, code) and the same comparisons => identical distances and
// indices.  A block leaves its rows' partial (min, argmin) and online-softmax state per part in the stream's scratch
// (vqk_set_scratch); vq_lds_merge_kernel folds the KS parts (lexicographic (distance, index): order-free; log-sum-exp merge).
// ------------------------------------------------------------------------------------------------
template <int ASSOC, bool WRITE_D, bool STATS>
__global__ __launch_bounds__(256, 2) void vq_assign_lds_kernel(const float* __restrict__ z, const float* __restrict__ e,
                                                               const float* __restrict__ z2, const float* __restrict__ e2,
                                                               int64_t n, int k, float* __restrict__ dmat, float inv_t,
                                                               float* __restrict__ part) {
    constexpr int DD = 256, NF = DD / 8, PITCH = 1040, TILE = 32 * PITCH;      // bytes
    extern __shared__ __attribute__((aligned(16))) char vsm[];                  // [2][32 rows][PITCH]
    float m_run = -INFINITY, s_run = 0.0f, sa_run = 0.0f;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n0 = (int64_t)blockIdx.x * 128 + wave * 32;
    const int j = lane & 31, half = lane >> 5;
    const float zz = z2[n0 + j];
    f32x4 zr[NF];
    {
        const float* zp = z + (n0 + j) * DD + 4 * half;
#pragma unroll
        for (int i = 0; i < NF; ++i) zr[i] = *reinterpret_cast<const f32x4*>(zp + 8 * i);
    }
    const int tiles_part = (k >> 5) / (int)gridDim.y;            // 32-code tiles per part (divides: the launcher's choice)
    const int t_begin = (int)blockIdx.y * tiles_part, t_end = t_begin + tiles_part;
    auto fetch = [&](int t, int b) {                             // this wave's 8 rows of tile t -> buffer b
        const char* src = reinterpret_cast<const char*>(e + ((int64_t)t * 32 + wave * 8) * DD) + lane * 16;
        char* dst = vsm + b * TILE + wave * 8 * PITCH;
#pragma unroll
        for (int r = 0; r < 8; ++r)
            __builtin_amdgcn_global_load_lds((const VQK_GLB void*)(src + r * 1024), (VQK_LDS void*)(dst + r * PITCH), 16, 0, 0);
    };
    fetch(t_begin, 0);
    float best = INFINITY;
    int best_i = 0x7fffffff;
    for (int t = t_begin; t < t_end; ++t) {
        const int b = (t - t_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this wave's 8 rows of tile t have landed (and its older stores)
        __syncthreads();                                         // ... the other waves' too; every wave is done with tile t-1
        float e2v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) e2v[r] = e2[t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half];
        if (t + 1 < t_end) fetch(t + 1, b ^ 1);                  // into tile t-1's buffer, in flight under this tile's 128 MFMAs
        const char* ap = vsm + b * TILE + j * PITCH + half * 16;
        f32x16 acc = {0};
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(ap + i * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], zr[i][0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], zr[i][1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], zr[i][2], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], zr[i][3], acc, 0, 0, 0);
        }
        float av[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int code = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float ab2 = 2.0f * acc[r];
            float dist;
            if (ASSOC == 0) dist = __fsub_rn(__fadd_rn(zz, e2v[r]), ab2);
            else            dist = __fadd_rn(__fsub_rn(zz, ab2), e2v[r]);
            if (dist < best) { best = dist; best_i = code; }
            if (WRITE_D) dmat[(n0 + j) * (int64_t)k + code] = dist;
            av[r] = -dist * inv_t;
        }
        if (STATS) {
            float tmax = av[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) tmax = fmaxf(tmax, av[r]);
            const float mn = fmaxf(m_run, tmax);
            const float sc = __expf(m_run - mn);                 // exp(-inf) = 0 on the first tile
            float ts = 0.0f, tsa = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ex = __expf(av[r] - mn);
                ts += ex;
                tsa = __fmaf_rn(ex, av[r], tsa);
            }
            s_run = __fmaf_rn(s_run, sc, ts);
            sa_run = __fmaf_rn(sa_run, sc, tsa);
            m_run = mn;
        }
    }
    if (STATS) {                                                 // the two half-waves of a row
        const float om = __shfl_xor(m_run, 32, 64), os = __shfl_xor(s_run, 32, 64), osa = __shfl_xor(sa_run, 32, 64);
        const float mn = fmaxf(m_run, om);
        const float c0 = __expf(m_run - mn), c1 = __expf(om - mn);
        s_run = s_run * c0 + os * c1;
        sa_run = sa_run * c0 + osa * c1;
        m_run = mn;
    }
    {
        const float od = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(best_i, 32, 64);
        if (od < best || (od == best && oi < best_i)) { best = od; best_i = oi; }
    }
    if (half == 0) {                                             // part[q][blockIdx.y][row], q = d, i, m, s, sa
        const int64_t slot = (int64_t)blockIdx.y * n + n0 + j, plane = (int64_t)gridDim.y * n;
        part[slot] = best;
        reinterpret_cast<int*>(part)[plane + slot] = best_i;
        if (STATS) { part[2 * plane + slot] = m_run; part[3 * plane + slot] = s_run; part[4 * plane + slot] = sa_run; }
    }
}

// fold the KS partial states of every row: idx (lowest distance, then lowest index), lse / h of the row's softmax, hsum += h
template <bool STATS>
__global__ __launch_bounds__(256) void vq_lds_merge_kernel(const float* __restrict__ part, int64_t n, int ks, int64_t* __restrict__ idx,
                                                           float* __restrict__ lse, float* __restrict__ hrow, float* __restrict__ hsum) {
    __shared__ float hs[4];
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t plane = (int64_t)ks * n;
    float h = 0.0f;
    if (row < n) {
        float bd = part[row]; int bi = reinterpret_cast<const int*>(part)[plane + row];
        for (int p = 1; p < ks; ++p) {
            const float od = part[(int64_t)p * n + row]; const int oi = reinterpret_cast<const int*>(part)[plane + (int64_t)p * n + row];
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        idx[row] = (bi == 0x7fffffff) ? 0 : (int64_t)bi;
        if (STATS) {
            float m = part[2 * plane + row], s_ = part[3 * plane + row], sa = part[4 * plane + row];
            for (int p = 1; p < ks; ++p) {
                const float om = part[2 * plane + (int64_t)p * n + row], os = part[3 * plane + (int64_t)p * n + row],
                            osa = part[4 * plane + (int64_t)p * n + row];
                const float mn = fmaxf(m, om);
                const float c0 = __expf(m - mn), c1 = __expf(om - mn);
                s_ = s_ * c0 + os * c1; sa = sa * c0 + osa * c1; m = mn;
            }
            const float l = m + __logf(s_);
            h = l - sa / s_;
            lse[row] = l;
            hrow[row] = h;
        }
    }
    if (STATS) {
        h = wave_sum(h);
        if ((threadIdx.x & 63) == 0) hs[threadIdx.x >> 6] = h;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(hsum, (hs[0] + hs[1]) + (hs[2] + hs[3]));
    }
}

// the LDS form serves D = 256 with whole 128-row blocks and whole tiles, given the stream's scratch for the partial states;
// everything else stays on the register-streaming kernel
static int vq_lds_parts(int64_t n, int k, int d) {
    if (!(d == 256 && n > 0 && (n % 128) == 0 && (k % 32) == 0 && VQK_TUNE("VQ_LDS", 1) != 0)) return 0;
    const int tiles = k >> 5;
    int ks = 1;
    while ((n / 128) * ks < 512 && ks < 16 && tiles % (2 * ks) == 0 && tiles / (2 * ks) >= 8) ks *= 2;
    const vqkd::DetState& sc = vqkd::scratch_state();
    if (!sc.ws || sc.bytes < (int64_t)5 * ks * n * 4) return 0;
    return ks;
}
template <int ASSOC, bool STATS>
static int launch_vq_assign_lds(const float* z, const float* e, const float* z2, const float* e2, int64_t n, int k, int ks, int64_t* idx,
                                float* dmat, float inv_t, float* lse, float* hrow, float* hsum, hipStream_t st) {
    constexpr int lds = 2 * 32 * 1040;
    static const hipError_t attr = hipFuncSetAttribute((const void*)vq_assign_lds_kernel<ASSOC, true, STATS>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return VQK_ERR_LAUNCH;
    float* part = vqkd::scratch_state().ws;
    hipLaunchKernelGGL((vq_assign_lds_kernel<ASSOC, true, STATS>), dim3((unsigned)(n / 128), (unsigned)ks), dim3(256), lds, st, z, e, z2, e2,
                       n, k, dmat, inv_t, part);
    hipLaunchKernelGGL((vq_lds_merge_kernel<STATS>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)part, n, ks, idx, lse,
                       hrow, hsum);
    return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
}

// ------------------------------------------------------------------------------------------------
// q = e[idx]; sum (q-z)^2; histogram.  One wavefront per row, float4 per lane.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vq_gather_kernel(const float* __restrict__ z, const float* __restrict__ e,
                                                        const int64_t* __restrict__ idx, int64_t n, int k, int d,
                                                        float* __restrict__ q, bf16_raw* __restrict__ q_lo,
                                                        float* __restrict__ sse, int32_t* __restrict__ hist) {
    // code histogram: block-local counts in LDS first (a collapsed codebook puts every row on a few codes -- one
    // global atomic per row then serialises on those addresses), flushed as one global atomic per non-empty bin
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int32_t* lh = reinterpret_cast<int32_t*>(smem);
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (hist) {
        for (int i = threadIdx.x; i < k; i += 256) lh[i] = 0;
        __syncthreads();
    }
    float local = 0.0f;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n; row += (int64_t)gridDim.x * 4) {
        const int64_t code = idx[row];
        const float* er = e + code * d;
        const float* zr = z + row * d;
        for (int c = lane * 4; c < d; c += 256) {
            const f32x4 ev = *reinterpret_cast<const f32x4*>(er + c);
            const f32x4 zv = *reinterpret_cast<const f32x4*>(zr + c);
            if (q) *reinterpret_cast<f32x4*>(q + row * d + c) = ev;
            if (q_lo) {
                u16x4 o = {f32_to_bf16(ev[0]), f32_to_bf16(ev[1]), f32_to_bf16(ev[2]), f32_to_bf16(ev[3])};
                *reinterpret_cast<u16x4*>(q_lo + row * d + c) = o;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) { const float t = ev[i] - zv[i]; local = __fmaf_rn(t, t, local); }
        }
        if (hist && lane == 0) atomicAdd(lh + code, 1);
    }
    local = wave_sum(local);
    if (lane == 0) part[wave] = local;
    __syncthreads();
    if (threadIdx.x == 0 && sse) atomicAdd(sse, part[0] + part[1] + part[2] + part[3]);
    if (hist)
        for (int i = threadIdx.x; i < k; i += 256) { const int32_t v = lh[i]; if (v) atomicAdd(hist + i, v); }
}

template <typename TDQ>
__global__ __launch_bounds__(256) void vq_backward_kernel(const float* __restrict__ z, const float* __restrict__ e,
                                                          const int64_t* __restrict__ idx, const TDQ* __restrict__ dq,
                                                          int64_t n, int d, float cz, const float* __restrict__ gs,
                                                          float* __restrict__ dz) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (gs) cz *= *gs;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n; row += (int64_t)gridDim.x * 4) {
        const int64_t code = idx[row];
        for (int c = lane; c < d; c += 64) {
            const float zv = z[row * d + c], qv = e[code * d + c];
            const float g = dq ? Elem<TDQ>::ld(dq + row * d + c) : 0.0f;
            dz[row * d + c] = __fmaf_rn(cz, zv - qv, g);
        }
    }
}

// dE[k] += ce * sum_{rows with idx == k} (e[k] - z[row]).  Block (k, s) scans rows [s*span, (s+1)*span) of idx (L2
// resident), lists its matches in LDS and sums them channel-parallel: no atomic per (row, channel) -- a collapsed
// codebook (all rows on a few codes, the state of a fresh model) made the scatter-add version 30x slower than the
// stream -- only one atomic per (code, split, channel) at the end.
__global__ __launch_bounds__(256) void vq_code_grad_kernel(const float* __restrict__ z, const float* __restrict__ e,
                                                           const int64_t* __restrict__ idx, int64_t n, int d,
                                                           int64_t span, float ce, const float* __restrict__ gs,
                                                           float* __restrict__ de, int ordered, float* __restrict__ part = nullptr) {
    constexpr int CAP = 2048;
    __shared__ int rows[CAP];
    __shared__ int nrows;
    const int k = blockIdx.x;
    const int64_t r0 = (int64_t)blockIdx.y * span, r1 = min(n, r0 + span);
    if (gs) ce *= *gs;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};                 // channels threadIdx.x + 256*j
    bool any = false;
    for (int64_t base = r0; base < r1; base += CAP) {
        if (threadIdx.x == 0) nrows = 0;
        __syncthreads();
        const int64_t lim = min(r1, base + CAP);
        if (ordered) {
            // deterministic mode: the matching rows are listed in ROW ORDER (ballot + prefix counts, wave by wave), so every
            // channel adds them in the same order in every run
            __shared__ int wcount[4];
            for (int64_t rb = base; rb < lim; rb += 256) {
                const int64_t r = rb + threadIdx.x;
                const bool hit = r < lim && idx[r] == k;
                const unsigned long long m = __ballot(hit);
                const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
                if (lane == 0) wcount[wv] = __popcll(m);
                __syncthreads();
                int pos = nrows + __popcll(m & ((1ull << lane) - 1ull));
                for (int q = 0; q < wv; ++q) pos += wcount[q];
                if (hit) rows[pos] = (int)(r - r0);
                __syncthreads();
                if (threadIdx.x == 0) nrows += wcount[0] + wcount[1] + wcount[2] + wcount[3];
                __syncthreads();
            }
        } else {
            for (int64_t r = base + threadIdx.x; r < lim; r += 256)
                if (idx[r] == k) rows[atomicAdd(&nrows, 1)] = (int)(r - r0);
        }
        __syncthreads();
        const int cnt = nrows;
        if (cnt) {
            any = true;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c = threadIdx.x + 256 * j;
                if (c < d) {
                    const float ev = e[(int64_t)k * d + c];
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;     // independent chains: the row loads overlap
                    int i = 0;
                    for (; i + 4 <= cnt; i += 4) {
                        const float z0 = z[(r0 + rows[i]) * d + c], z1 = z[(r0 + rows[i + 1]) * d + c];
                        const float z2 = z[(r0 + rows[i + 2]) * d + c], z3 = z[(r0 + rows[i + 3]) * d + c];
                        a0 += ev - z0; a1 += ev - z1; a2 += ev - z2; a3 += ev - z3;
                    }
                    for (; i < cnt; ++i) a0 += ev - z[(r0 + rows[i]) * d + c];
                    acc[j] += (a0 + a1) + (a2 + a3);
                }
            }
        }
        __syncthreads();
    }
    if (part) {
        // deterministic mode with row splits (round 4: one block per code walked all N rows of a collapsed codebook -- 0.58 ms):
        // every (split, code) block STORES its partial row (zeros when it has no rows); vq_code_grad_reduce_kernel adds the
        // splits in index order
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c < d) part[((int64_t)blockIdx.y * gridDim.x + k) * d + c] = ce * acc[j];
        }
        return;
    }
    if (any)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = threadIdx.x + 256 * j;
            if (c < d) atomicAdd(de + (int64_t)k * d + c, ce * acc[j]);
        }
}

__global__ __launch_bounds__(256) void vq_code_grad_reduce_kernel(const float* __restrict__ part, int64_t elems, int splits,
                                                                  float* __restrict__ de) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    float s = 0.f;
    for (int q = 0; q < splits; ++q) s += part[(int64_t)q * elems + i];
    de[i] += s;
}

__global__ __launch_bounds__(256) void ema_stats_kernel(const float* __restrict__ z, const int64_t* __restrict__ idx,
                                                        int64_t n, int d, float* __restrict__ counts,
                                                        float* __restrict__ dw) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t row = (int64_t)blockIdx.x * 4 + wave; row < n; row += (int64_t)gridDim.x * 4) {
        const int64_t code = idx[row];
        if (lane == 0) atomicAdd(counts + code, 1.0f);
        for (int c = lane; c < d; c += 64) atomicAdd(dw + code * d + c, z[row * d + c]);
    }
}

// vector_quantizers.py:161,164
__global__ void ema_count_kernel(float* __restrict__ ema_count, const float* __restrict__ counts, int k, float decay,
                                 float eps, float batch) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const float c = __fadd_rn(__fmul_rn(ema_count[i], decay), __fmul_rn(1.0f - decay, counts[i]));
    ema_count[i] = __fmul_rn(__fdiv_rn(__fadd_rn(c, eps), __fadd_rn(batch, __fmul_rn((float)k, eps))), batch);
}

// vector_quantizers.py:167,169 (ema_count already updated)
__global__ void ema_weight_kernel(const float* __restrict__ ema_count, float* __restrict__ ema_weight,
                                  float* __restrict__ codebook, const float* __restrict__ dw, int64_t total, int d,
                                  float decay) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const float w = __fadd_rn(__fmul_rn(ema_weight[i], decay), __fmul_rn(1.0f - decay, dw[i]));
        ema_weight[i] = w;
        codebook[i] = __fdiv_rn(w, ema_count[i / d]);
    }
}

extern "C" {

int vqk_row_sqnorm_f32(const float* x, int64_t rows, int d, float* out, void* stream) {
    VQK_REQUIRE(x && out, VQK_ERR_ARG);
    VQK_REQUIRE(rows >= 0 && d > 0, VQK_ERR_SHAPE);
    if (rows == 0) return VQK_OK;
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, vqk_stream(stream), x, rows, d, out);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_vq_assign_f32(const float* z, const float* e, const float* z2, const float* e2, int64_t n, int k, int d,
                      int assoc, int64_t* idx, void* stream) {
    VQK_REQUIRE(z && e && z2 && e2 && idx, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && k > 0 && d > 0 && (d % 8) == 0, VQK_ERR_SHAPE);
    VQK_REQUIRE(assoc == 0 || assoc == 1, VQK_ERR_ARG);
    VQK_REQUIRE(vqk_aligned16(z) && vqk_aligned16(e), VQK_ERR_ALIGN);
    if (n == 0) return VQK_OK;
    const size_t lds = (size_t)32 * (d + 4) * 4 + 128 * 4 + 128 * 4;
    VQK_REQUIRE(lds <= 160 * 1024, VQK_ERR_SHAPE);
    const dim3 grid((unsigned)((n + 31) / 32));
    if (d == 256) {
        if (assoc == 0) hipLaunchKernelGGL((vq_assign_reg_kernel<0, false, 256>), grid, dim3(256), 0, vqk_stream(stream), z, e, z2, e2, n, k, idx, (float*)nullptr);
        else hipLaunchKernelGGL((vq_assign_reg_kernel<1, false, 256>), grid, dim3(256), 0, vqk_stream(stream), z, e, z2, e2, n, k, idx, (float*)nullptr);
        VQK_CHECK_LAUNCH();
        return VQK_OK;
    }
    if (assoc == 0) {
        if (lds > 64 * 1024) hipFuncSetAttribute((const void*)vq_assign_kernel<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((vq_assign_kernel<0, false>), grid, dim3(256), lds, vqk_stream(stream), z, e, z2, e2, n, k, d, idx, (float*)nullptr);
    } else {
        if (lds > 64 * 1024) hipFuncSetAttribute((const void*)vq_assign_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL((vq_assign_kernel<1, false>), grid, dim3(256), lds, vqk_stream(stream), z, e, z2, e2, n, k, d, idx, (float*)nullptr);
    }
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_vq_distances_f32(const float* z, const float* e, const float* z2, const float* e2, int64_t n, int k, int d,
                         int assoc, int64_t* idx, float* dmat, void* stream) {
    VQK_REQUIRE(z && e && z2 && e2 && idx && dmat, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && k > 0 && d > 0 && (d % 8) == 0, VQK_ERR_SHAPE);
    VQK_REQUIRE(assoc == 0 || assoc == 1, VQK_ERR_ARG);
    VQK_REQUIRE(vqk_aligned16(z) && vqk_aligned16(e), VQK_ERR_ALIGN);
    if (n == 0) return VQK_OK;
    const size_t lds = (size_t)32 * (d + 4) * 4 + 128 * 4 + 128 * 4;
    VQK_REQUIRE(lds <= 64 * 1024, VQK_ERR_SHAPE);
    const dim3 grid((unsigned)((n + 31) / 32));
    if (const int ks = vq_lds_parts(n, k, d))
        return assoc == 0 ? launch_vq_assign_lds<0, false>(z, e, z2, e2, n, k, ks, idx, dmat, 0.0f, nullptr, nullptr, nullptr, vqk_stream(stream))
                          : launch_vq_assign_lds<1, false>(z, e, z2, e2, n, k, ks, idx, dmat, 0.0f, nullptr, nullptr, nullptr, vqk_stream(stream));
    if (d == 256) {
        if (assoc == 0) hipLaunchKernelGGL((vq_assign_reg_kernel<0, true, 256, false, VQK_ENT_CHAINS>), grid, dim3(256), 0, vqk_stream(stream), z, e, z2, e2, n, k, idx, dmat);
        else hipLaunchKernelGGL((vq_assign_reg_kernel<1, true, 256, false, VQK_ENT_CHAINS>), grid, dim3(256), 0, vqk_stream(stream), z, e, z2, e2, n, k, idx, dmat);
        VQK_CHECK_LAUNCH();
        return VQK_OK;
    }
    if (assoc == 0) hipLaunchKernelGGL((vq_assign_kernel<0, true>), grid, dim3(256), lds, vqk_stream(stream), z, e, z2, e2, n, k, d, idx, dmat);
    else hipLaunchKernelGGL((vq_assign_kernel<1, true>), grid, dim3(256), lds, vqk_stream(stream), z, e, z2, e2, n, k, d, idx, dmat);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_vq_distances_stats_f32(const float* z, const float* e, const float* z2, const float* e2, int64_t n, int k, int d,
                               int assoc, int64_t* idx, float* dmat, float temperature, float* lse, float* hrow, float* hsum,
                               void* stream) {
    VQK_REQUIRE(z && e && z2 && e2 && idx && dmat && lse && hrow && hsum, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && k > 0 && d == 256 && temperature > 0.0f, VQK_ERR_SHAPE);
    VQK_REQUIRE(assoc == 0 || assoc == 1, VQK_ERR_ARG);
    VQK_REQUIRE(vqk_aligned16(z) && vqk_aligned16(e), VQK_ERR_ALIGN);
    if (n == 0) return VQK_OK;
    const dim3 grid((unsigned)((n + 31) / 32));
    const float inv_t = 1.0f / temperature;
    if (const int ks = vq_lds_parts(n, k, d))
        return assoc == 0 ? launch_vq_assign_lds<0, true>(z, e, z2, e2, n, k, ks, idx, dmat, inv_t, lse, hrow, hsum, vqk_stream(stream))
                          : launch_vq_assign_lds<1, true>(z, e, z2, e2, n, k, ks, idx, dmat, inv_t, lse, hrow, hsum, vqk_stream(stream));
    if (assoc == 0) hipLaunchKernelGGL((vq_assign_reg_kernel<0, true, 256, true, VQK_ENT_CHAINS>), grid, dim3(256), 0, vqk_stream(stream), z, e, z2, e2, n, k, idx, dmat, inv_t, lse, hrow, hsum);
    else hipLaunchKernelGGL((vq_assign_reg_kernel<1, true, 256, true, VQK_ENT_CHAINS>), grid, dim3(256), 0, vqk_stream(stream), z, e, z2, e2, n, k, idx, dmat, inv_t, lse, hrow, hsum);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_vq_gather_f32(const float* z, const float* e, const int64_t* idx, int64_t n, int k, int d, float* q, void* q_lo,
                      float* sse, int32_t* hist, void* stream) {
    VQK_REQUIRE(z && e && idx, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && k > 0 && d > 0 && (d % 4) == 0, VQK_ERR_SHAPE);
    if (n == 0) return VQK_OK;
    VQK_REQUIRE(!hist || k <= 16384, VQK_ERR_SHAPE);
    const int blocks = vqk_grid_1d(n, 4 * 16, 512);          // >= 16 rows per wave: few histogram flushes
    hipLaunchKernelGGL(vq_gather_kernel, dim3((unsigned)blocks), dim3(256), hist ? (size_t)k * 4 : 0, vqk_stream(stream), z, e,
                       idx, n, k, d, q, reinterpret_cast<bf16_raw*>(q_lo), sse, hist);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_vq_backward_f32(const float* z, const float* e, const int64_t* idx, const void* dq, int dq_dtype, int64_t n,
                        int k, int d, float cz, float ce, const float* gscale_dev, float* dz, float* de, void* stream) {
    VQK_REQUIRE(z && e && idx && dz, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && k > 0 && d > 0, VQK_ERR_SHAPE);
    if (n == 0) return VQK_OK;
    VQK_REQUIRE(!de || d <= 1024, VQK_ERR_SHAPE);
    const dim3 grid(vqk_grid_1d(n, 4));
    if (dq_dtype == VQK_F32)
        hipLaunchKernelGGL(vq_backward_kernel<float>, grid, dim3(256), 0, vqk_stream(stream), z, e, idx, (const float*)dq, n, d, cz, gscale_dev, dz);
    else if (dq_dtype == VQK_BF16)
        hipLaunchKernelGGL(vq_backward_kernel<bf16_raw>, grid, dim3(256), 0, vqk_stream(stream), z, e, idx, (const bf16_raw*)dq, n, d, cz, gscale_dev, dz);
    else return VQK_ERR_DTYPE;
    if (de) {
        int splits = (int)((n + 255) / 256);              // <= 256 rows per block even when every row hits one code
        if (splits > 64) splits = 64;
        vqkd::DetState& dst = vqkd::det_state();
        const int det = dst.on;
        float* part = nullptr;
        if (det) {
            // deterministic mode: rows of a (code, split) block added in row order, the splits' partial rows through the
            // workspace in split order; without a workspace that holds them: one block per code
            if (splits > 32) splits = 32;
            while (splits > 1 && (int64_t)splits * k * d * 4 > dst.bytes) splits >>= 1;
            part = (splits > 1 && dst.ws) ? dst.ws : nullptr;
            if (!part) splits = 1;
        }
        const int64_t span = (n + splits - 1) / splits;
        hipLaunchKernelGGL(vq_code_grad_kernel, dim3((unsigned)k, (unsigned)splits), dim3(256), 0, vqk_stream(stream), z, e,
                           idx, n, d, span, ce, gscale_dev, de, det, part);
        if (part)
            hipLaunchKernelGGL(vq_code_grad_reduce_kernel, dim3((unsigned)(((int64_t)k * d + 255) / 256)), dim3(256), 0,
                               vqk_stream(stream), (const float*)part, (int64_t)k * d, splits, de);
    }
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_ema_stats_f32(const float* z, const int64_t* idx, int64_t n, int k, int d, float* counts, float* dw, void* stream) {
    VQK_REQUIRE(z && idx && counts && dw, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && k > 0 && d > 0, VQK_ERR_SHAPE);
    if (n == 0) return VQK_OK;
    hipLaunchKernelGGL(ema_stats_kernel, dim3(vqk_grid_1d(n, 4)), dim3(256), 0, vqk_stream(stream), z, idx, n, d, counts, dw);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_ema_update_f32(float* ema_count, float* ema_weight, float* codebook, const float* counts, const float* dw, int k,
                       int d, float decay, float eps, float batch, void* stream) {
    VQK_REQUIRE(ema_count && ema_weight && codebook && counts && dw, VQK_ERR_ARG);
    VQK_REQUIRE(k > 0 && d > 0, VQK_ERR_SHAPE);
    hipLaunchKernelGGL(ema_count_kernel, dim3((k + 255) / 256), dim3(256), 0, vqk_stream(stream), ema_count, counts, k, decay, eps, batch);
    const int64_t total = (int64_t)k * d;
    hipLaunchKernelGGL(ema_weight_kernel, dim3(vqk_grid_1d(total, 256)), dim3(256), 0, vqk_stream(stream), ema_count,
                       ema_weight, codebook, dw, total, d, decay);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

}  // extern "C"
