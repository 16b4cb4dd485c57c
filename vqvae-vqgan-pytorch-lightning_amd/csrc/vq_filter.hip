// ------------------------------------------------------------------------------------------------
// Nearest-codeword assignment (vqvae/modules/vector_quantizers.py:37-44, :337-343) as a bf16 CANDIDATE FILTER followed by
// an EXACT fp32 re-rank of the candidates -- indices stay bit-exact against oracle/vq_oracle.c (and so against the reference's
// own torch.argmin on the fixtures), but the N*K*D products run on the bf16 matrix pipe (2.5 PF) instead of the exact-fp32
// one (157 TF), which bounded vq_assign_reg_kernel at ~27 us for (8192, 1024, 256).
//
// Why the filter cannot lose the winner.  Notation: a_k = z.e_k in exact arithmetic; Z2, E2_k the fp32 squared norms both
// paths share; D_k the distance the exact path computes, fl(fl(Z2 + E2_k) - fl(2 A_k)) (Standard/EMA) or
// fl(fl(Z2 - fl(2 A_k)) + E2_k) (Entropy), A_k the canonical fp32 fma chain; S_k = fl(E2_k - fl(2 At_k)) the filter score,
// At_k the bf16 MFMA dot product of the RNE-rounded operands with fp32 accumulation.
//   |At_k - a_k| <= [(2 u + u^2) + 2^-14 (1 + u)^2] |z| |e_k|,  u = 2^-8 (bf16 unit roundoff; products of two bf16 are
//   exact in fp32; the 256-term accumulation is bounded at 4x the sequential round-to-nearest bound, which also covers a
//   truncating or tree-shaped accumulator)            =>  |S_k - (E2_k - 2 a_k)| <= delta_k := 0.0160 |z| |e_k|
//   |D_k - (Z2 + E2_k - 2 a_k)| <= eta := 2^-13 |z| max|e| + 2^-21 (Z2 + max E2)   (fma chain + the two outer roundings)
// If k* is the exact path's argmin then D_k* <= D_j for all j, hence S_k* - delta_k* <= min_j (S_j + delta_j) + 2 eta:
// every code passing   lo_k := S_k - delta_k  <=  U + H,   U := min_j (S_j + delta_j),  H := 2^-12 |z| max|e| + 2^-20 (Z2 + max E2)
// is a candidate, and k* (with every code tied with it) always is.  The candidates are re-ranked with the canonical fp32 chain
// (k order 8j + {0,4,1,5,2,6,3,7}, the order v_mfma_f32_32x32x2_f32 consumes) and the exact distance formula; the minimum
// of (distance, index) in lexicographic order = torch.argmin's first minimum.
//
// Kernel: block = 4 waves = 32 z rows (bf16 B fragments in registers for the whole kernel) x all K codes, every wave a
// contiguous quarter of the 32-code tiles, walked from a block-dependent start (the whole grid reads the same codebook);
// A fragments = the bf16 codebook in fragment-major order (prepared once per call): sixteen coalesced 1-KiB loads per
// tile, requested a whole tile ahead.  PASS 1 keeps lo_k of the first 8 tiles of every wave in registers (8 x 4 x 32 codes
// = all of K = 1024) and reduces U per row; PASS 2 compares them (or recomputes the tiles beyond) and appends the
// candidates to an LDS list; the block then evaluates its candidates exactly, one thread per candidate (z rows staged in
// LDS, the code row in two batches of 32 independent loads), and takes the row minima with 64-bit LDS atomics on
// (orderable distance bits, index).  A block whose list overflows (a collapsed codebook: thousands of near-ties) falls
// back to the exact-fp32 MFMA loop for its 32 rows -- no host decision, graph-capturable.
//
// Round 4: the quantizer FORWARD is this one kernel (vqk_vq_forward_f32).  What depends on the codebook only -- the bf16
// fragment-major copy, |e|^2, the margin factors, max |e|^2 -- is built by vqk_vq_prepare_f32 when the codebook CHANGES (after
// the optimizer step / the EMA update), not in every step.  The block stages its 32 z rows in LDS first: |z|^2 comes from
// there in the canonical order of row_sqnorm_kernel (bit-identical), the bf16 B fragments are LDS reads instead of
// 1-KiB-strided global loads, the re-rank and the epilogue reuse the tile.  Epilogue (vector_quantizers.py:44-56 fused):
// q = e[idx] gathered as fp32 and / or bf16, sum (q - z)^2 (one atomic per block), code histogram (duplicates inside the
// block counted first: one atomic per distinct code and block).
// ------------------------------------------------------------------------------------------------
#include "common.h"

namespace {

#ifndef VQK_VQF_ABL
#define VQK_VQF_ABL 0        // timing-only ablation bits (tools/ab_build.sh): 1 no re-rank, 2 no pass 2, 4 no pass 1, 8 no z loads, 16 no epilogue
#endif
#ifndef VQK_VQB_ABL
#define VQK_VQB_ABL 0        // timing-only ablation bits of the fused backward: 1 plain stores instead of atomics, 2 no LDS adds
#endif
constexpr int FD = 256;                                          // embedding_dim of every reference config
constexpr int FCAP = 2048;                                       // candidate list capacity per block (64 per row on average)
constexpr float F_DELTA = 0.0160f;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));   // v_cvt_pk_bf16_f32: round to nearest even
}

// codebook -> bf16 (RNE), FRAGMENT-MAJOR: [tile of 32 codes][k-step s 0..15][lane 0..63][8 bf16], lane = 32 * half + (code % 32)
// holds columns 16 s + 8 half .. + 7 of its code -- every MFMA A operand of the filter is ONE coalesced 1-KiB load (the
// row-major form made each load touch 32 different lines).  + delta factors eps_e[k] = F_DELTA * sqrt(E2_k).
// One wave per code row: lane l converts columns 4 l .. 4 l + 3.
__global__ __launch_bounds__(256) void vq_filter_prep_kernel(const float* __restrict__ e, const float* __restrict__ e2_in, int k,
                                                             bf16_raw* __restrict__ eb, float* __restrict__ eps_e,
                                                             float* __restrict__ e2_out) {
    const int lane = threadIdx.x & 63;
    const int row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= k) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(e + (int64_t)row * FD + lane * 4);
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const u32x2 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
    const int col = lane * 4, s = col >> 4, half = (col >> 3) & 1, w4 = col & 7;      // 4 columns inside one 8-column fragment slot
    const int64_t dst = ((((int64_t)(row >> 5) * 16 + s) * 64 + half * 32 + (row & 31)) * 8) + w4;
    *reinterpret_cast<u32x2*>(eb + dst) = o;
    float sq;
    if (e2_in) {
        sq = e2_in[row];
    } else {
        // |e|^2 in the canonical order of vq.hip::row_sqnorm_kernel (lane l: fma chain over l, l + 64, ...; xor butterfly 32..1)
        const float* p = e + (int64_t)row * FD;
        sq = 0.0f;
#pragma unroll
        for (int kk = 0; kk < FD; kk += 64) sq = __fmaf_rn(p[kk + lane], p[kk + lane], sq);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) sq = __fadd_rn(sq, __shfl_xor(sq, off, 64));
    }
    if (lane == 0) {
        eps_e[row] = F_DELTA * sqrtf(sq);
        if (e2_out) e2_out[row] = sq;
    }
}

// max |e|^2 over the codebook (one block; behind the prep kernel on the same stream)
__global__ __launch_bounds__(256) void vq_filter_max_kernel(const float* __restrict__ e2, int k, float* __restrict__ out) {
    __shared__ float red[4];
    float m2 = 0.0f;
    for (int i = threadIdx.x; i < k; i += 256) m2 = fmaxf(m2, e2[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m2;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__device__ __forceinline__ unsigned orderable(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <int ASSOC>
__device__ __forceinline__ float exact_dist(float zz, float e2c, float ab) {
    const float ab2 = 2.0f * ab;
    if (ASSOC == 0) return __fsub_rn(__fadd_rn(zz, e2c), ab2);
    return __fadd_rn(__fsub_rn(zz, ab2), e2c);
}

// the exact-fp32 MFMA loop of vq.hip::vq_assign_kernel for this block's 32 rows (overflow fallback); zt: the block's z tile
// [32][FD + 4] in LDS (already staged), zz: |z|^2 of the lane's row; the winners go to fin[32] (LDS)
template <int ASSOC>
__device__ void exact_block(const float* __restrict__ e, const float* __restrict__ e2, int k, const float* zt, float zz,
                            float* red_d, int* red_i, int* fin) {
    constexpr int ld = FD + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, half = lane >> 5;
    const float* zb = zt + j * ld + 4 * half;
    const int tiles = (k + 31) >> 5, per_wave = (tiles + 3) >> 2;
    const int t_begin = wave * per_wave, t_end = min(tiles, t_begin + per_wave);
    float best = INFINITY;
    int best_i = 0x7fffffff;
    for (int t = t_begin; t < t_end; ++t) {
        int code_row = t * 32 + j; if (code_row >= k) code_row = k - 1;
        const float* ea = e + (int64_t)code_row * FD + 4 * half;
        f32x16 acc = {0};
#pragma unroll 4
        for (int m = 0; m < FD; m += 8) {
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
                const float dist = exact_dist<ASSOC>(zz, e2[code], acc[r]);
                if (dist < best) { best = dist; best_i = code; }
            }
        }
    }
    {
        const float od = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(best_i, 32, 64);
        if (od < best || (od == best && oi < best_i)) { best = od; best_i = oi; }
    }
    if (half == 0) { red_d[wave * 32 + j] = best; red_i[wave * 32 + j] = best_i; }
    __syncthreads();
    if (tid < 32) {
        float bd = red_d[tid]; int bi = red_i[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float od = red_d[w * 32 + tid]; const int oi = red_i[w * 32 + tid];
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        fin[tid] = (bi == 0x7fffffff) ? 0 : bi;
    }
}

// k % 32 == 0.  CT: tiles per wave whose lo values stay in REGISTERS between the passes (8 x 4 waves x 32 codes = all of
// K = 1024); tiles beyond them are recomputed in pass 2.  Dynamic LDS: 34 KiB (the block's z tile).
// z2_in / e2max_in: optional precomputed |z|^2 per row / max |e|^2 (NULL: computed here); q32 / q_lo / sse / hist: the
// optional fused outputs of the quantizer forward.
template <int ASSOC, int CT>
__global__ __launch_bounds__(256, 1) void vq_assign_filter_kernel(const float* __restrict__ z, const float* __restrict__ e,
                                                                  const bf16_raw* __restrict__ eb,
                                                                  const float* __restrict__ z2_in, const float* __restrict__ e2,
                                                                  const float* __restrict__ eps_e,
                                                                  const float* __restrict__ e2max_in, int64_t n, int k,
                                                                  int64_t* __restrict__ idx, float* __restrict__ q32,
                                                                  bf16_raw* __restrict__ q_lo, float* __restrict__ sse,
                                                                  int32_t* __restrict__ hist) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int ZLD = FD + 4;
    float* zt = reinterpret_cast<float*>(smem);                  // [32][FD + 4]: the block's z rows (rows past n: row n - 1 again)
    __shared__ unsigned cand[FCAP];
    __shared__ unsigned long long key[32];
    __shared__ float red_u[4][32];
    __shared__ float red_m[4];
    __shared__ int ncand, overflow;
    __shared__ float fb_d[128];
    __shared__ int fb_i[128];
    __shared__ float z2s[32];
    __shared__ int fin[32];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t n0 = (int64_t)blockIdx.x * 32;
    const int j = lane & 31, half = lane >> 5;
    int64_t zrow = n0 + j; if (zrow >= n) zrow = n - 1;
    if (tid < 32) key[tid] = ~0ull;
    if (tid == 0) { ncand = 0; overflow = 0; }

    const int tiles = k >> 5;
    const int per_wave = (tiles + 3) >> 2;
    const int t_begin = wave * per_wave;
    const int t_end = min(tiles, t_begin + per_wave);
    const int cnt = max(t_end - t_begin, 0);
    // every block of the grid walks the same codebook: start each block at a different tile of its waves' ranges so that the
    // 256 CUs do not all ask the L2 for the same lines at the same moment (tile order inside a wave is free)
    const int rot = cnt > 0 ? (int)(blockIdx.x % (unsigned)cnt) : 0;
    auto tile_of = [&](int tt) -> int { int q = tt + rot; if (q >= cnt) q -= cnt; return t_begin + q; };
    // A fragments of one tile: 16 coalesced 1-KiB loads (fragment-major bf16 codebook)
    auto load_tile = [&](int t, u32x4 (&dst)[16]) {
        const bf16_raw* p = eb + (int64_t)t * (16 * 64 * 8) + lane * 8;
#pragma unroll
        for (int i = 0; i < 16; ++i) dst[i] = *reinterpret_cast<const u32x4*>(p + i * (64 * 8));
    };
    // the z tile first (coalesced 16-byte loads, HBM latency), the first two codebook tiles (L2) behind it
    {
        f32x4 st[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int v = it * 256 + tid, r = v >> 6, c = v & 63;
            int64_t src = n0 + r; if (src >= n) src = n - 1;
            st[it] = ((VQK_VQF_ABL & 8) && n > 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(z + src * FD + 4 * c);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int v = it * 256 + tid, r = v >> 6, c = v & 63;
            *reinterpret_cast<f32x4*>(zt + r * ZLD + 4 * c) = st[it];
        }
    }
    u32x4 fa[3][16];                                             // fragment ring: two tiles in flight behind the one being multiplied
    if (cnt > 0) load_tile(tile_of(0), fa[0]);
    if (cnt > 1) load_tile(tile_of(1), fa[1]);
    // max E2 over the codebook: prepared with the codebook, or K floats read by every block
    float e2max;
    if (e2max_in) {
        e2max = e2max_in[0];
    } else {
        float m2 = 0.0f;
        for (int i = tid; i < k; i += 256) m2 = fmaxf(m2, e2[i]);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
        if (lane == 0) red_m[wave] = m2;
    }
    __syncthreads();
    if (!e2max_in) e2max = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
    // |z|^2: wave w owns rows 8 w .. 8 w + 7, canonical order of row_sqnorm_kernel (lane l: fma chain over l, l + 64, l + 128,
    // l + 192, then the xor butterfly 32 .. 1) => the same bits
    if (z2_in) {
        if (tid < 32) { int64_t r = n0 + tid; if (r >= n) r = n - 1; z2s[tid] = z2_in[r]; }
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float* p = zt + (wave * 8 + i) * ZLD;
            float acc = 0.0f;
#pragma unroll
            for (int kk = 0; kk < FD; kk += 64) acc = __fmaf_rn(p[kk + lane], p[kk + lane], acc);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) acc = __fadd_rn(acc, __shfl_xor(acc, off, 64));
            if (lane == 0) z2s[wave * 8 + i] = acc;
        }
    }
    // this lane's slice of its z row as bf16 B fragments: k-step s covers columns 16 s + 8 half .. + 7
    bf16x8_t zf[16];
    {
        const float* zp = zt + j * ZLD + 8 * half;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(zp + 16 * s);
            const f32x4 b = *reinterpret_cast<const f32x4*>(zp + 16 * s + 4);
            const u32x4 o = {pack2_bf16(a[0], a[1]), pack2_bf16(a[2], a[3]), pack2_bf16(b[0], b[1]), pack2_bf16(b[2], b[3])};
            zf[s] = __builtin_bit_cast(bf16x8_t, o);
        }
    }
    __syncthreads();
    const float zz = z2s[j];
    const float zn = sqrtf(zz);
    const float hmargin = 2.44140625e-4f * zn * sqrtf(e2max) + 9.5367431640625e-7f * (zz + e2max);    // 2^-12, 2^-20

    // lo / hi of one tile from its fragments: acc = bf16 MFMA dot products of (32 codes) x (32 z rows); the lane owns z row j
    // and the 16 codes t*32 + (r&3) + 8*(r>>2) + 4*half
    // (vector-memory results return IN ORDER: the tile's own |e|^2 / margin loads are issued BEFORE the fragment prefetch of a
    // later tile -- `prefetch` -- so that waiting for them does not wait for the prefetch as well)
    auto tile_scores = [&](int t, const u32x4 (&frag)[16], float (&lo)[16], float (&hi)[16], auto&& prefetch) {
        f32x4 e2q[4], epq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            e2q[q] = *reinterpret_cast<const f32x4*>(e2 + t * 32 + 8 * q + 4 * half);
            epq[q] = *reinterpret_cast<const f32x4*>(eps_e + t * 32 + 8 * q + 4 * half);
        }
        __builtin_amdgcn_sched_barrier(0);
        prefetch();
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc = {0};
#pragma unroll
        for (int i = 0; i < 16; ++i)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, frag[i]), zf[i], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float sc = __fsub_rn(e2q[r >> 2][r & 3], 2.0f * acc[r]);
            const float dl = zn * epq[r >> 2][r & 3];
            lo[r] = sc - dl; hi[r] = sc + dl;
        }
    };

    // ---------------------------------------------------------------- pass 1: U = min_k hi_k per row
    // (the NEXT tile's sixteen loads are issued before this tile's MFMAs: a whole tile of matrix work covers the L2 latency)
    float u = INFINITY;
    float lo_reg[CT][16];
    const int cnt1 = (VQK_VQF_ABL & 4) ? (int)(n == 0) : cnt;
#pragma unroll
    for (int tt = 0; tt < CT; ++tt) {
        if (tt < cnt1) {
            float hi[16];
            tile_scores(tile_of(tt), fa[tt % 3], lo_reg[tt], hi, [&]() { if (tt + 2 < cnt1) load_tile(tile_of(tt + 2), fa[(tt + 2) % 3]); });
#pragma unroll
            for (int r = 0; r < 16; ++r) u = fminf(u, hi[r]);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) lo_reg[tt][r] = INFINITY;
        }
    }
    // tiles beyond the register cache (K > 1024): same ring, three tiles per trip (ring slot = tile % 3; CT % 3 == CT_R)
    constexpr int CT_R = CT % 3;
    for (int tt = CT; tt < cnt1; tt += 3) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (tt + q < cnt1) {
                float lo[16], hi[16];
                tile_scores(tile_of(tt + q), fa[(CT_R + q) % 3], lo, hi,
                            [&]() { if (tt + q + 2 < cnt1) load_tile(tile_of(tt + q + 2), fa[(CT_R + q + 2) % 3]); });
#pragma unroll
                for (int r = 0; r < 16; ++r) u = fminf(u, hi[r]);
            }
        }
    }
    u = fminf(u, __shfl_xor(u, 32, 64));
    if (half == 0) red_u[wave][j] = u;
    __syncthreads();
    const float thr = fminf(fminf(red_u[0][j], red_u[1][j]), fminf(red_u[2][j], red_u[3][j])) + hmargin;

    // ---------------------------------------------------------------- pass 2: candidates
    // a lane's candidates of the cached tiles are counted first and appended with ONE LDS atomic (a returning atomic per
    // candidate cost a round trip each: 3 us of the kernel)
    auto tile_mask = [&](const float (&lo)[16]) -> unsigned {
        unsigned mask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) mask |= (lo[r] <= thr ? 1u : 0u) << r;
        return mask;
    };
    auto append = [&](int t, unsigned mask, int& pos) {
        while (mask) {
            const int r = __builtin_ctz(mask);
            mask &= mask - 1;
            if (pos < FCAP) cand[pos] = ((unsigned)j << 26) | (unsigned)(t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half);
            else overflow = 1;
            ++pos;
        }
    };
    const int cnt2 = (VQK_VQF_ABL & 2) ? (int)(n == 0) : cnt;
    {
        unsigned masks[CT];
        int total = 0;
#pragma unroll
        for (int tt = 0; tt < CT; ++tt) {
            masks[tt] = tt < cnt2 ? tile_mask(lo_reg[tt]) : 0u;
            total += __builtin_popcount(masks[tt]);
        }
        if (total) {
            int pos = atomicAdd(&ncand, total);
#pragma unroll
            for (int tt = 0; tt < CT; ++tt)
                if (masks[tt]) append(tile_of(tt), masks[tt], pos);
        }
    }
    if (CT < cnt2) {
        load_tile(tile_of(CT), fa[0]);
        if (CT + 1 < cnt2) load_tile(tile_of(CT + 1), fa[1]);
    }
    for (int tt = CT; tt < cnt2; tt += 3) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (tt + q < cnt2) {
                float lo[16], hi[16];
                tile_scores(tile_of(tt + q), fa[q], lo, hi, [&]() { if (tt + q + 2 < cnt2) load_tile(tile_of(tt + q + 2), fa[(q + 2) % 3]); });
                const unsigned mask = tile_mask(lo);
                if (mask) {
                    int pos = atomicAdd(&ncand, __builtin_popcount(mask));
                    append(tile_of(tt + q), mask, pos);
                }
            }
        }
    }
    __syncthreads();
    if (overflow) {                                              // block-uniform
        exact_block<ASSOC>(e, e2, k, zt, zz, fb_d, fb_i, fin);
    } else {
        // ------------------------------------------------------------ exact re-rank, one thread per candidate
        // The 256-term fma chain is sequential by definition; what can be hidden is its operand traffic: the z rows come from
        // the LDS tile, a candidate's code row arrives in two batches of 32 independent 16-byte loads (two L2 round trips
        // instead of sixteen).
        const int nc = (VQK_VQF_ABL & 1) ? (int)(n == 0) : ncand;
        for (int c = tid; c < nc; c += 256) {
            const unsigned pk = cand[c];
            const int row = (int)(pk >> 26), code = (int)(pk & 0x03ffffffu);
            const float* zr = zt + row * ZLD;
            const float* er = e + (int64_t)code * FD;
            float acc = 0.0f;
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                f32x4 ev[32];
#pragma unroll
                for (int i = 0; i < 32; ++i) ev[i] = *reinterpret_cast<const f32x4*>(er + hb * 128 + 4 * i);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x4 za = *reinterpret_cast<const f32x4*>(zr + hb * 128 + 8 * i);
                    const f32x4 zb = *reinterpret_cast<const f32x4*>(zr + hb * 128 + 8 * i + 4);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        acc = __fmaf_rn(ev[2 * i][t], za[t], acc);
                        acc = __fmaf_rn(ev[2 * i + 1][t], zb[t], acc);
                    }
                }
            }
            const float dist = exact_dist<ASSOC>(z2s[row], e2[code], acc);
            // a NaN distance is never a candidate's winner: a row whose distances are ALL NaN keeps index 0 (as torch.argmin
            // does for an all-NaN row); a row with some NaN codes returns the finite argmin here, the first NaN in torch --
            // the exact kernel (vq.hip) behaves the same way, NaN latents / codes are outside the contract
            if (dist == dist)
                atomicMin(&key[row], ((unsigned long long)orderable(dist) << 32) | (unsigned)code);
        }
        __syncthreads();
        if (tid < 32) {
            const unsigned long long kk = key[tid];
            fin[tid] = kk == ~0ull ? 0 : (int)(kk & 0xffffffffull);
        }
    }
    __syncthreads();
    if (tid < 32 && n0 + tid < n) idx[n0 + tid] = (int64_t)fin[tid];
    if (!(q32 || q_lo || sse || hist) || ((VQK_VQF_ABL & 16) && n > 0)) return;      // kernel-uniform

    // ---------------------------------------------------------------- fused epilogue (vector_quantizers.py:44-56)
    // thread (row = tid / 8, sub = tid % 8): columns 4 sub + 32 jj .. + 3, jj = 0..7 -- all 32 rows of the block in flight at
    // once (eight threads cover 128 consecutive bytes of a row per jj); q = e[idx] as fp32 and / or bf16, sum (q - z)^2
    float local = 0.0f;
    {
        const int row = tid >> 3, sub = tid & 7;
        if (n0 + row < n) {
            const int code = fin[row];
            const float* er = e + (int64_t)code * FD + sub * 4;
            const float* zr = zt + row * ZLD + sub * 4;
            f32x4 ev[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) ev[jj] = *reinterpret_cast<const f32x4*>(er + 32 * jj);
            const int64_t o = (n0 + row) * FD + sub * 4;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const f32x4 zv = *reinterpret_cast<const f32x4*>(zr + 32 * jj);
                if (q32) *reinterpret_cast<f32x4*>(q32 + o + 32 * jj) = ev[jj];
                if (q_lo) {
                    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
                    const u32x2 ob = {pack2_bf16(ev[jj][0], ev[jj][1]), pack2_bf16(ev[jj][2], ev[jj][3])};
                    *reinterpret_cast<u32x2*>(q_lo + o + 32 * jj) = ob;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) { const float dlt = ev[jj][t] - zv[t]; local = __fmaf_rn(dlt, dlt, local); }
            }
        }
    }
    if (sse) {
        local = wave_sum(local);
        if (lane == 0) red_m[wave] = local;
    }
    // histogram: duplicates inside the block are counted first (a collapsed codebook puts every row on a few codes)
    if (hist && tid < 32 && n0 + tid < n) {
        const int code = fin[tid];
        int count = 0;
        bool leader = true;
        for (int u = 0; u < 32; ++u) {
            const bool same = (n0 + u < n) && fin[u] == code;
            count += same ? 1 : 0;
            if (same && u < tid) leader = false;
        }
        if (leader) atomicAdd(hist + code, count);
    }
    if (sse) {
        __syncthreads();
        if (tid == 0) atomicAdd(sse, (red_m[0] + red_m[1]) + (red_m[2] + red_m[3]));
    }
}

// ------------------------------------------------------------------------------------------------
// Quantizer BACKWARD in one kernel (vector_quantizers.py:52-56 differentiated; replaces vq_backward_kernel +
// vq_code_grad_kernel = 8 + 51 us): block = 32 rows x 256 channels.
//   dz[row] = dq[row] + s cz (z[row] - e[idx[row]])
//   dE[k]  += s ce sum_{rows of the block with idx == k} (e[k] - z[row])
// The rows of a block that share a code are summed through an LDS tile first (chains of equal-code rows), so the
// block sends ONE coalesced fp32 atomic row per DISTINCT code to dE: a collapsed codebook (every row on a few codes: the
// state of a fresh model) costs a few atomics per block instead of one per (row, channel), a spread-out assignment at most
// N x 256 uncontended ones.  Deterministic mode keeps the ordered two-kernel form (vq.hip).
// ------------------------------------------------------------------------------------------------
template <typename TDQ, bool HAS_DQ, bool HAS_DE>
__global__ __launch_bounds__(256) void vq_backward_fused_kernel(const float* __restrict__ z, const float* __restrict__ e,
                                                                const int64_t* __restrict__ idx, const TDQ* __restrict__ dq,
                                                                int64_t n, float cz, float ce, const float* __restrict__ gs,
                                                                float* __restrict__ dz, float* __restrict__ de) {
    constexpr int ALD = FD + 32;                                 // row pitch of the difference tile: the eight rows of a wave's
    __shared__ __attribute__((aligned(16))) float diff[HAS_DE ? 32 * ALD : 4];    // 16-byte stores land in disjoint bank halves
    __shared__ int code_s[32], next_s[32], first_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 32;
    if (gs) { const float s = *gs; cz *= s; ce *= s; }
    if (tid < 32) code_s[tid] = n0 + tid < n ? (int)idx[n0 + tid] : -1;
    {
        // thread (row = tid / 8, sub = tid % 8): columns 4 sub + 32 jj .. + 3 -- every load of the block in one batch
        const int row = tid >> 3, sub = tid & 7;
        if (n0 + row < n) {
            const int code = (int)idx[n0 + row];
            const int64_t o = (n0 + row) * FD + sub * 4;
            const float* er = e + (int64_t)code * FD + sub * 4;
            f32x4 zv[8], ev[8], g[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                zv[jj] = *reinterpret_cast<const f32x4*>(z + o + 32 * jj);
                ev[jj] = *reinterpret_cast<const f32x4*>(er + 32 * jj);
                if constexpr (!HAS_DQ) g[jj] = f32x4{0.f, 0.f, 0.f, 0.f};
                else if constexpr (sizeof(TDQ) == 4) g[jj] = *reinterpret_cast<const f32x4*>(dq + o + 32 * jj);
                else {
                    const u16x4 r = *reinterpret_cast<const u16x4*>(dq + o + 32 * jj);
                    g[jj] = f32x4{bf16_to_f32(r[0]), bf16_to_f32(r[1]), bf16_to_f32(r[2]), bf16_to_f32(r[3])};
                }
            }
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                f32x4 out, df;
#pragma unroll
                for (int t = 0; t < 4; ++t) { out[t] = __fmaf_rn(cz, zv[jj][t] - ev[jj][t], g[jj][t]); df[t] = ev[jj][t] - zv[jj][t]; }
                *reinterpret_cast<f32x4*>(dz + o + 32 * jj) = out;
                if constexpr (HAS_DE) *reinterpret_cast<f32x4*>(diff + row * ALD + sub * 4 + 32 * jj) = df;
            }
        }
    }
    if constexpr (!HAS_DE) return;
    __syncthreads();
    // rows that share a code form a chain in row order: first_s[r] == r marks the chain's head, next_s the next member
    // (an LDS float atomic per element was tried first: ds_add_f32 under the 8-way bank conflicts of this layout cost 9 us)
    if (tid < 32) {
        const int code = code_s[tid];
        int first = tid, next = -1;
        if (code >= 0) {
            for (int u = 0; u < tid; ++u)
                if (code_s[u] == code) { first = u; break; }
            for (int u = tid + 1; u < 32; ++u)
                if (code_s[u] == code) { next = u; break; }
        }
        first_s[tid] = first; next_s[tid] = next;
    }
    __syncthreads();
    // wave w: heads w, w + 4, ...; lane: columns lane + 64 t.  ONE coalesced fp32 atomic row per distinct code of the block,
    // its members added in row order (the sum inside a block is deterministic; the order of the blocks' atomics is not)
    for (int r = wave; r < 32; r += 4) {
        if (code_s[r] < 0 || first_s[r] != r) continue;          // wave-uniform
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int m = r; m >= 0; m = next_s[m]) {
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] += diff[m * ALD + t * 64 + lane];
        }
        float* drow = de + (int64_t)code_s[r] * FD;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if ((VQK_VQB_ABL & 1) && n > 0) drow[t * 64 + lane] = ce * a[t];
            else atomicAdd(drow + t * 64 + lane, ce * a[t]);
        }
    }
}

// EMA statistics (vector_quantizers.py:159-163: counts[k] = #rows on code k, dw[k] = sum of their z) with the same per-block
// pre-aggregation as the fused backward: the 32 rows of a block that share a code are chained and summed through an LDS tile,
// ONE coalesced fp32 atomic row (+ one count) per distinct code and block -- ema_stats_kernel issued one atomic per (row,
// channel): 120 us at (8192, 1024, 256) on the collapsed codebook of a fresh model (every row contends for a few code rows).
__global__ __launch_bounds__(256) void ema_stats_block_kernel(const float* __restrict__ z, const int64_t* __restrict__ idx, int64_t n,
                                                              float* __restrict__ counts, float* __restrict__ dw) {
    constexpr int ALD = FD + 32;
    __shared__ __attribute__((aligned(16))) float tile[32 * ALD];
    __shared__ int code_s[32], next_s[32], first_s[32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * 32;
    if (tid < 32) code_s[tid] = n0 + tid < n ? (int)idx[n0 + tid] : -1;
    {
        const int row = tid >> 3, sub = tid & 7;
        if (n0 + row < n) {
            const int64_t o = (n0 + row) * FD + sub * 4;
            f32x4 zv[8];
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) zv[jj] = *reinterpret_cast<const f32x4*>(z + o + 32 * jj);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) *reinterpret_cast<f32x4*>(tile + row * ALD + sub * 4 + 32 * jj) = zv[jj];
        }
    }
    __syncthreads();
    if (tid < 32) {
        const int code = code_s[tid];
        int first = tid, next = -1;
        if (code >= 0) {
            for (int u = 0; u < tid; ++u)
                if (code_s[u] == code) { first = u; break; }
            for (int u = tid + 1; u < 32; ++u)
                if (code_s[u] == code) { next = u; break; }
        }
        first_s[tid] = first; next_s[tid] = next;
    }
    __syncthreads();
    for (int r = wave; r < 32; r += 4) {
        if (code_s[r] < 0 || first_s[r] != r) continue;          // wave-uniform: the chain's head
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        int cnt = 0;
        for (int m = r; m >= 0; m = next_s[m]) {
            ++cnt;
#pragma unroll
            for (int t = 0; t < 4; ++t) a[t] += tile[m * ALD + t * 64 + lane];
        }
        float* drow = dw + (int64_t)code_s[r] * FD;
#pragma unroll
        for (int t = 0; t < 4; ++t) atomicAdd(drow + t * 64 + lane, a[t]);
        if (lane == 0) atomicAdd(counts + code_s[r], (float)cnt);
    }
}

}  // namespace

extern "C" {

// workspace: bf16 fragment-major codebook | eps_e[K] | e2[K] | max e2 (256-byte aligned sections)
static inline int64_t vqf_off_eps(int k, int d) { return ((int64_t)k * d * 2 + 255) & ~(int64_t)255; }
static inline int64_t vqf_off_e2(int k, int d) { return vqf_off_eps(k, d) + (((int64_t)k * 4 + 255) & ~(int64_t)255); }
static inline int64_t vqf_off_max(int k, int d) { return vqf_off_e2(k, d) + (((int64_t)k * 4 + 255) & ~(int64_t)255); }

int64_t vqk_vq_filter_ws_bytes(int k, int d) { return vqf_off_max(k, d) + 256; }

static int vqf_prepare(const float* e, const float* e2_in, int k, int d, void* ws, hipStream_t st) {
    char* w = reinterpret_cast<char*>(ws);
    float* e2w = reinterpret_cast<float*>(w + vqf_off_e2(k, d));
    hipLaunchKernelGGL(vq_filter_prep_kernel, dim3((unsigned)((k + 3) / 4)), dim3(256), 0, st, e, e2_in, k,
                       reinterpret_cast<bf16_raw*>(w), reinterpret_cast<float*>(w + vqf_off_eps(k, d)), e2w);
    hipLaunchKernelGGL(vq_filter_max_kernel, dim3(1), dim3(256), 0, st, (const float*)e2w, k,
                       reinterpret_cast<float*>(w + vqf_off_max(k, d)));
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

static int vqf_launch(const float* z, const float* e, const void* ws, const float* z2, const float* e2, int64_t n, int k, int d,
                      int assoc, int64_t* idx, float* q32, void* q_lo, float* sse, int32_t* hist, hipStream_t st) {
    const char* w = reinterpret_cast<const char*>(ws);
    const bf16_raw* eb = reinterpret_cast<const bf16_raw*>(w);
    const float* eps_e = reinterpret_cast<const float*>(w + vqf_off_eps(k, d));
    const float* e2w = e2 ? e2 : reinterpret_cast<const float*>(w + vqf_off_e2(k, d));
    const float* e2max = reinterpret_cast<const float*>(w + vqf_off_max(k, d));
    const int per_wave = ((k >> 5) + 3) >> 2;
    const int ct = per_wave >= 8 ? 8 : per_wave >= 4 ? 4 : per_wave >= 2 ? 2 : 1;
    constexpr int lds = 34 * 1024;                               // the block's z tile
    const dim3 grid((unsigned)((n + 31) / 32));
#define VQF_LAUNCH(A, C) hipLaunchKernelGGL((vq_assign_filter_kernel<A, C>), grid, dim3(256), (size_t)lds, st, z, e, eb, z2, e2w, \
                                            eps_e, e2max, n, k, idx, q32, reinterpret_cast<bf16_raw*>(q_lo), sse, hist)
    if (assoc == 0) {
        if (ct == 8) VQF_LAUNCH(0, 8); else if (ct == 4) VQF_LAUNCH(0, 4); else if (ct == 2) VQF_LAUNCH(0, 2); else VQF_LAUNCH(0, 1);
    } else {
        if (ct == 8) VQF_LAUNCH(1, 8); else if (ct == 4) VQF_LAUNCH(1, 4); else if (ct == 2) VQF_LAUNCH(1, 2); else VQF_LAUNCH(1, 1);
    }
#undef VQF_LAUNCH
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_vq_assign_filtered_f32(const float* z, const float* e, const float* z2, const float* e2, int64_t n, int k, int d,
                               int assoc, int64_t* idx, void* ws, int64_t ws_bytes, void* stream) {
    VQK_REQUIRE(z && e && z2 && e2 && idx && ws, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && k > 0 && d == FD && (k % 32) == 0 && k < (1 << 26), VQK_ERR_SHAPE);
    VQK_REQUIRE(assoc == 0 || assoc == 1, VQK_ERR_ARG);
    VQK_REQUIRE(vqk_aligned16(z) && vqk_aligned16(e) && vqk_aligned16(ws) && vqk_aligned16(e2), VQK_ERR_ALIGN);
    VQK_REQUIRE(ws_bytes >= vqk_vq_filter_ws_bytes(k, d), VQK_ERR_ARG);
    if (n == 0) return VQK_OK;
    hipStream_t st = vqk_stream(stream);
    const int rc = vqf_prepare(e, e2, k, d, ws, st);
    if (rc != VQK_OK) return rc;
    return vqf_launch(z, e, ws, z2, e2, n, k, d, assoc, idx, nullptr, nullptr, nullptr, nullptr, st);
}

int vqk_vq_prepare_f32(const float* e, int k, int d, void* ws, int64_t ws_bytes, void* stream) {
    VQK_REQUIRE(e && ws, VQK_ERR_ARG);
    VQK_REQUIRE(k > 0 && d == FD && (k % 32) == 0 && k < (1 << 26), VQK_ERR_SHAPE);
    VQK_REQUIRE(vqk_aligned16(e) && vqk_aligned16(ws), VQK_ERR_ALIGN);
    VQK_REQUIRE(ws_bytes >= vqk_vq_filter_ws_bytes(k, d), VQK_ERR_WORKSPACE);
    return vqf_prepare(e, nullptr, k, d, ws, vqk_stream(stream));
}

int vqk_vq_forward_f32(const float* z, const float* e, const void* ws, int64_t ws_bytes, int64_t n, int k, int d, int assoc,
                       int64_t* idx, float* q, void* q_lo, float* sse, int32_t* hist, void* stream) {
    VQK_REQUIRE(z && e && ws && idx, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && k > 0 && d == FD && (k % 32) == 0 && k < (1 << 26), VQK_ERR_SHAPE);
    VQK_REQUIRE(assoc == 0 || assoc == 1, VQK_ERR_ARG);
    VQK_REQUIRE(vqk_aligned16(z) && vqk_aligned16(e) && vqk_aligned16(ws) && (!q || vqk_aligned16(q)) && (!q_lo || vqk_aligned16(q_lo)),
                VQK_ERR_ALIGN);
    VQK_REQUIRE(ws_bytes >= vqk_vq_filter_ws_bytes(k, d), VQK_ERR_WORKSPACE);
    if (n == 0) return VQK_OK;
    return vqf_launch(z, e, ws, nullptr, nullptr, n, k, d, assoc, idx, q, q_lo, sse, hist, vqk_stream(stream));
}

int vqk_ema_stats_fused_f32(const float* z, const int64_t* idx, int64_t n, int k, int d, float* counts, float* dw, void* stream) {
    VQK_REQUIRE(z && idx && counts && dw, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && k > 0 && d == FD, VQK_ERR_SHAPE);
    VQK_REQUIRE(vqk_aligned16(z), VQK_ERR_ALIGN);
    if (n == 0) return VQK_OK;
    hipLaunchKernelGGL(ema_stats_block_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, vqk_stream(stream), z, idx, n, counts, dw);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

int vqk_vq_backward_fused_f32(const float* z, const float* e, const int64_t* idx, const void* dq, int dq_dtype, int64_t n, int k,
                              int d, float cz, float ce, const float* gscale_dev, float* dz, float* de, void* stream) {
    VQK_REQUIRE(z && e && idx && dz, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && k > 0 && d == FD, VQK_ERR_SHAPE);
    VQK_REQUIRE(dq_dtype == VQK_F32 || dq_dtype == VQK_BF16, VQK_ERR_DTYPE);
    VQK_REQUIRE(vqk_aligned16(z) && vqk_aligned16(e) && vqk_aligned16(dz) && (!dq || vqk_aligned16(dq)), VQK_ERR_ALIGN);
    if (n == 0) return VQK_OK;
    const dim3 grid((unsigned)((n + 31) / 32));
#define VQB(T, Q, E) hipLaunchKernelGGL((vq_backward_fused_kernel<T, Q, E>), grid, dim3(256), 0, vqk_stream(stream), z, e, idx, \
                                       (const T*)dq, n, cz, ce, gscale_dev, dz, de)
    if (dq_dtype == VQK_F32) {
        if (dq) { if (de) VQB(float, true, true); else VQB(float, true, false); }
        else { if (de) VQB(float, false, true); else VQB(float, false, false); }
    } else {
        if (dq) { if (de) VQB(bf16_raw, true, true); else VQB(bf16_raw, true, false); }
        else { if (de) VQB(float, false, true); else VQB(float, false, false); }
    }
#undef VQB
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

}  // extern "C"
