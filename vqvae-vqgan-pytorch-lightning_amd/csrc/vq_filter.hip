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
// ------------------------------------------------------------------------------------------------
#include "common.h"

namespace {

#ifndef VQK_VQF_ABL
#define VQK_VQF_ABL 0        // timing-only ablation bits (tools/ab_build.sh): 1 no re-rank, 2 no pass 2, 4 no pass 1, 8 no z staging
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
__global__ __launch_bounds__(256) void vq_filter_prep_kernel(const float* __restrict__ e, const float* __restrict__ e2, int k,
                                                             bf16_raw* __restrict__ eb, float* __restrict__ eps_e) {
    const int lane = threadIdx.x & 63;
    const int row = (int)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= k) return;
    const f32x4 v = *reinterpret_cast<const f32x4*>(e + (int64_t)row * FD + lane * 4);
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    const u32x2 o = {pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3])};
    const int col = lane * 4, s = col >> 4, half = (col >> 3) & 1, w4 = col & 7;      // 4 columns inside one 8-column fragment slot
    const int64_t dst = ((((int64_t)(row >> 5) * 16 + s) * 64 + half * 32 + (row & 31)) * 8) + w4;
    *reinterpret_cast<u32x2*>(eb + dst) = o;
    if (lane == 0) eps_e[row] = F_DELTA * sqrtf(e2[row]);
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

// the exact-fp32 MFMA loop of vq.hip::vq_assign_kernel for this block's 32 rows (overflow fallback); zt: [32][FD + 4] in LDS
template <int ASSOC>
__device__ void exact_block(const float* __restrict__ z, const float* __restrict__ e, const float* __restrict__ z2,
                            const float* __restrict__ e2, int64_t n, int k, int64_t n0, int64_t* __restrict__ idx,
                            float* zt, float* red_d, int* red_i) {
    constexpr int ld = FD + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int v = tid; v < 32 * (FD / 4); v += 256) {
        const int r = v / (FD / 4), c = v - r * (FD / 4);
        int64_t src = n0 + r; if (src >= n) src = n - 1;
        *reinterpret_cast<f32x4*>(zt + r * ld + 4 * c) = *reinterpret_cast<const f32x4*>(z + src * FD + 4 * c);
    }
    __syncthreads();
    const int j = lane & 31, half = lane >> 5;
    int64_t zrow = n0 + j; if (zrow >= n) zrow = n - 1;
    const float zz = z2[zrow];
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

// k % 32 == 0.  CT: tiles per wave whose lo values stay in REGISTERS between the passes (8 x 4 waves x 32 codes = all of
// K = 1024); tiles beyond them are recomputed in pass 2.  Dynamic LDS: 34 KiB (the re-rank's / the fallback's z tile).
template <int ASSOC, int CT>
__global__ __launch_bounds__(256, 1) void vq_assign_filter_kernel(const float* __restrict__ z, const float* __restrict__ e,
                                                                  const bf16_raw* __restrict__ eb,
                                                                  const float* __restrict__ z2, const float* __restrict__ e2,
                                                                  const float* __restrict__ eps_e, int64_t n, int k,
                                                                  int64_t* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ unsigned cand[FCAP];
    __shared__ unsigned long long key[32];
    __shared__ float red_u[4][32];
    __shared__ float red_m[4];
    __shared__ int ncand, overflow;
    __shared__ float fb_d[128];
    __shared__ int fb_i[128];
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
    u32x4 fa[3][16];                                             // fragment ring: two tiles in flight behind the one being multiplied
    if (cnt > 0) load_tile(tile_of(0), fa[0]);
    if (cnt > 1) load_tile(tile_of(1), fa[1]);

    // this lane's slice of its z row as bf16 B fragments: k-step s covers columns 16 s + 8 half .. + 7
    bf16x8_t zf[16];
    {
        const float* zp = z + zrow * FD + 8 * half;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(zp + 16 * s);
            const f32x4 b = *reinterpret_cast<const f32x4*>(zp + 16 * s + 4);
            const u32x4 o = {pack2_bf16(a[0], a[1]), pack2_bf16(a[2], a[3]), pack2_bf16(b[0], b[1]), pack2_bf16(b[2], b[3])};
            zf[s] = __builtin_bit_cast(bf16x8_t, o);
        }
    }
    const float zz = z2[zrow];
    const float zn = sqrtf(zz);
    // max E2 over the codebook (K floats: every block reads them once; no global state between launches)
    float m2 = 0.0f;
    for (int i = tid; i < k; i += 256) m2 = fmaxf(m2, e2[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
    if (lane == 0) red_m[wave] = m2;
    __syncthreads();
    const float e2max = fmaxf(fmaxf(red_m[0], red_m[1]), fmaxf(red_m[2], red_m[3]));
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
        exact_block<ASSOC>(z, e, z2, e2, n, k, n0, idx, reinterpret_cast<float*>(smem), fb_d, fb_i);
        return;
    }

    // ---------------------------------------------------------------- exact re-rank, one thread per candidate
    // The 256-term fma chain is sequential by definition; what can be hidden is its operand traffic: the block's 32 z rows
    // are staged in LDS (dynamic LDS), and a candidate's code row arrives in two batches of 32 independent
    // 16-byte loads (two L2 round trips instead of sixteen).
    constexpr int ZLD = FD + 4;
    float* zt = reinterpret_cast<float*>(smem);
    for (int v = tid; v < ((VQK_VQF_ABL & 8) ? (int)(n == 0) : 32 * (FD / 4)); v += 256) {
        const int r = v / (FD / 4), c = v - r * (FD / 4);
        int64_t src = n0 + r; if (src >= n) src = n - 1;
        *reinterpret_cast<f32x4*>(zt + r * ZLD + 4 * c) = *reinterpret_cast<const f32x4*>(z + src * FD + 4 * c);
    }
    __syncthreads();
    const int nc = (VQK_VQF_ABL & 1) ? (int)(n == 0) : ncand;
    for (int c = tid; c < nc; c += 256) {
        const unsigned pk = cand[c];
        const int row = (int)(pk >> 26), code = (int)(pk & 0x03ffffffu);
        int64_t zr_i = n0 + row; if (zr_i >= n) zr_i = n - 1;
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
        const float dist = exact_dist<ASSOC>(z2[zr_i], e2[code], acc);
        if (dist == dist)                                        // NaN never wins (torch.argmin / the oracle keep index 0 then)
            atomicMin(&key[row], ((unsigned long long)orderable(dist) << 32) | (unsigned)code);
    }
    __syncthreads();
    if (tid < 32 && n0 + tid < n) {
        const unsigned long long kk = key[tid];
        idx[n0 + tid] = kk == ~0ull ? 0 : (int64_t)(kk & 0xffffffffull);
    }
}

}  // namespace

extern "C" {

int64_t vqk_vq_filter_ws_bytes(int k, int d) { return (int64_t)k * d * 2 + (int64_t)k * 4 + 256; }

int vqk_vq_assign_filtered_f32(const float* z, const float* e, const float* z2, const float* e2, int64_t n, int k, int d,
                               int assoc, int64_t* idx, void* ws, int64_t ws_bytes, void* stream) {
    VQK_REQUIRE(z && e && z2 && e2 && idx && ws, VQK_ERR_ARG);
    VQK_REQUIRE(n >= 0 && k > 0 && d == FD && (k % 32) == 0 && k < (1 << 26), VQK_ERR_SHAPE);
    VQK_REQUIRE(assoc == 0 || assoc == 1, VQK_ERR_ARG);
    VQK_REQUIRE(vqk_aligned16(z) && vqk_aligned16(e) && vqk_aligned16(ws) && vqk_aligned16(e2), VQK_ERR_ALIGN);
    VQK_REQUIRE(ws_bytes >= vqk_vq_filter_ws_bytes(k, d), VQK_ERR_ARG);
    if (n == 0) return VQK_OK;
    hipStream_t st = vqk_stream(stream);
    bf16_raw* eb = reinterpret_cast<bf16_raw*>(ws);
    float* eps_e = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + (((int64_t)k * d * 2 + 255) & ~(int64_t)255));
    hipLaunchKernelGGL(vq_filter_prep_kernel, dim3((unsigned)((k + 3) / 4)), dim3(256), 0, st, e, e2, k, eb, eps_e);
    VQK_CHECK_LAUNCH();
    const int per_wave = ((k >> 5) + 3) >> 2;
    const int ct = per_wave >= 8 ? 8 : per_wave >= 4 ? 4 : per_wave >= 2 ? 2 : 1;
    constexpr int lds = 34 * 1024;                               // the z tile of the re-rank / of the overflow fallback
    const dim3 grid((unsigned)((n + 31) / 32));
#define VQF_LAUNCH(A, C) hipLaunchKernelGGL((vq_assign_filter_kernel<A, C>), grid, dim3(256), (size_t)lds, st, z, e, \
                                            (const bf16_raw*)eb, z2, e2, (const float*)eps_e, n, k, idx)
    if (assoc == 0) {
        if (ct == 8) VQF_LAUNCH(0, 8); else if (ct == 4) VQF_LAUNCH(0, 4); else if (ct == 2) VQF_LAUNCH(0, 2); else VQF_LAUNCH(0, 1);
    } else {
        if (ct == 8) VQF_LAUNCH(1, 8); else if (ct == 4) VQF_LAUNCH(1, 4); else if (ct == 2) VQF_LAUNCH(1, 2); else VQF_LAUNCH(1, 1);
    }
#undef VQF_LAUNCH
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}

}  // extern "C"
