// ------------------------------------------------------------------------------------------------
// conv3x3_x3_kernel: the PARITY-GRADE 3x3 convolution on the bf16 matrix pipe -- fp32 activations in, fp32 out, every product
// x*w evaluated as THREE bf16 products with fp32 accumulation ("split products"):
//
//      x = x_hi + x_lo,  w = w_hi + w_lo      (hi = bf16_rne(v), lo = bf16_rne(v - hi); |v - hi - lo| <= 2^-18 |v|)
//      x*w ~= x_hi*w_hi + x_lo*w_hi + x_hi*w_lo      (dropped: x_lo*w_lo <= 2^-18 |x*w|)
//
// i.e. ~2^-17 relative per product against 2^-8 for the plain bf16 mode and 2^-24 for v_mfma_f32_32x32x2_f32 -- at 3/16 of the
// fp32-MFMA cost per multiply-add.  Replaces the F.conv2d calls of vqvae/modules/autoencoder.py:57-60 (ResBlock), :102-105
// (Upsample), :132 / :153 (conv_in / conv_out of the latent) in the fp32 compute mode; the exact-fp32 kernel
// (conv.hip: conv3x3_halo_breg_kernel) stays as the reference mode (ops.set_conv_products('fp32')).
//
// Structure: the persistent software pipeline of conv3x3_stream_kernel (conv.hip), 128-pixel tiles (8x16), two 256-thread blocks
// per CU.  A unit = (tile, 32-channel chunk).  While unit u runs out of LDS buffer u & 1, the fp32 halo of unit u + 1 is in flight
// HBM -> registers; after the unit's MFMAs each lane SPLITS its four floats (2 x v_cvt_pk_bf16_f32, 4 subtractions, 2 more
// conversions) and writes the hi and the lo bf16 PLANE of buffer (u + 1) & 1 (two ds_write_b64); one barrier per unit.  LDS rows are
// 80 B per plane (64 B payload): the conflict-free `lane base + immediate` fragment addressing of the stream kernel, twice.
// Weights: fragment-major hi/lo pairs (vqk_conv_pack_weights layout 5: [cot32][chunk32][tap][ks][hi|lo] x 1 KiB), streamed
// L2 -> registers one tap ahead.  Per (tap, k-substep) and wave: NI x NJ x 3 MFMAs on NI x 2 pixel fragments + NJ x 2 weight
// fragments.  fp32 epilogue straight from the accumulators (v_permlane32_swap pairs the half-wave runs: 32 B per lane).
//
// The same kernel with NTAP = 1 serves the 1x1 convs (no halo, memory-bound); its epilogue can leave the GroupNorm sums of the output
// in the consumer's workspace (ConvGeom::gn_ws).  conv3x3_wgrad_x3_kernel (below) is the weight gradient: both fp32 operands split
// in registers, three products per staged fragment pair.
//
// split_pair_kernel: fp32 [rows][C] -> bf16 [rows][2C] = (hi | lo): the operand form of the FIRST weight-gradient design of this
// mode, kept as an A/B option (conv_wgmx.hip, ConvGeom::fold; ops.X3_WGRAD_FOLD): dW = dy_hi^T x_hi + dy_hi^T x_lo + dy_lo^T x_hi as
// three tile classes of ONE launch of the bf16 matrix/auxiliary-wave weight-gradient kernel, folded onto the same dW tile by its final
// atomic pass -- 36.2 ms per step (two split passes per conv included) against 21.6 for the kernel below.
// ------------------------------------------------------------------------------------------------
#include "conv_geom.h"
#include <type_traits>
#include <math.h>

namespace {

using vqkd::ConvGeom;
using vqkd::xcd_remap;
using vqkd::pack_bf16x2;

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

__device__ __forceinline__ float x3_act(float v, int act) {
    if (act == 1) return tanhf(v);
    if (act == 2) return fmaxf(v, 0.0f);
    if (act == 3) return v > 0.0f ? v : 0.2f * v;
    return v;
}

// WL 0: the four waves as 2 (pixels) x 2 (couts), wave tile 64 px x 64 couts; WL 1: 1 x 4, wave tile 128 px x 32 couts (half the
// L2 -> register weight stream per MFMA, twice the LDS fragment reads)
// NTAP 9: 3x3, 'same' padding; NTAP 1: the 1x1 convs (ResBlock shortcuts, autoencoder.py:52-55; the latent's conv_out, :143) -- no halo, two
// phases per unit, memory-bound (fp32 in + out at ~5 TB/s).
// NTAP 4: the 2x2-RESAMPLING convs in PHASE form (the algebra of conv_mx.hip's NTAP = 4, ConvGeom::phase_mode / phase_rev): a nearest-x2
// upsample followed by a 3x3 conv (autoencoder.py:102-105), and a 3x3 conv followed by a 2x2 average pool (:89-91), collapse per
// output / input phase (a, b) to a 2x2 window of low-resolution shifts with pre-summed weights (layout 6): 4/9 of the multiply-adds.
//   phase_mode 1 (phase = a TILE dimension): y[2i+a][2j+b] = window of x at rows i+a-1.., columns j+b-1.. -- the Upsample conv's forward;
//     with phase_rev and the conv's data-gradient operand: the data gradient of conv + AvgPool from the POOLED gradient.
//   phase_mode 2 (phase = a UNIT dimension: the tile accumulates its four phases x chunks): dx[i][j] = sum over the phases of the mirrored
//     windows of the phase (a, b) of a full-resolution tensor, gathered at stride 2 -- the Upsample conv's data gradient; with
//     phase_rev and the forward operand: the FORWARD of conv + AvgPool = a 4x4 stride-2 conv (+ pooled skip, + GroupNorm sums).
template <int TWLOG, int WL, int NTAP = 9>
__global__ __launch_bounds__(256, 2) void conv3x3_x3_kernel(const float* __restrict__ x, const bf16_raw* __restrict__ wp,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            float* __restrict__ y, const char* __restrict__ zeros, ConvGeom g,
                                                            int act) {
    constexpr int HM = NTAP == 1 ? 0 : 1;                        // halo margin
    constexpr int TWD = NTAP == 4 ? 2 : 3;                       // taps per window row
    constexpr int PIX = 128, TW = 1 << TWLOG, TH = PIX / TW, HW2 = TW + 2 * HM, HROWS = (TH + 2 * HM) * HW2;
    constexpr int RS = 80;                                       // padded LDS row stride per plane (64 B payload)
    constexpr int HALO_INSTR = (HROWS + 7) / 8;                  // register pieces: 8 rows x 128 B (32 fp32 channels) per wave load
    constexpr int PLANE = HALO_INSTR * 8 * RS, BUF = 2 * PLANE;  // hi plane, lo plane (whole pieces: the last piece's padding rows are written too)
    constexpr int NSLOT = (HALO_INSTR + 3) / 4;
    constexpr int NI = WL ? 4 : 2, NJ = WL ? 1 : 2;
    constexpr int UNITW = NTAP * 4 * 1024;                       // weight bytes of one (32-cout tile, chunk): taps x 2 ks x (hi, lo)
    typedef bf16x8_t frag_t;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = g.w >> TWLOG, tiles_y = g.h / TH;
    const int pmode = NTAP == 4 ? g.phase_mode : 0;
    const int total_tiles = g.n * tiles_y * tiles_x * g.tiles_n * (pmode == 1 ? 4 : 1);
    const int nch = g.cin >> 5;                                  // 32-channel chunks
    const int nun = pmode == 2 ? 4 * nch : nch;                  // units per tile
    const int vbid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int my_tiles = (total_tiles - vbid + (int)gridDim.x - 1) / (int)gridDim.x;
    const int units = my_tiles * nun;
    if (units <= 0) return;

    const int wm = WL ? 0 : wave >> 1, wn = WL ? wave : wave & 1;
    const int p = lane & 31, kg = lane >> 5;
    // 16-wide patches: an MFMA tile's 32 pixels are two patch rows.  ds_read_b128 serves a wave in the lane groups {0-3, 12-15, 20-27} and
    // {4-11, 16-19, 28-31} (+32): with lanes 0-15 on one row and 16-31 on the next, each group straddles both rows and two of its
    // sixteen 80-byte-pitch slots collide (measured: SQ_LDS_BANK_CONFLICT = 50 % of SQ_LDS_IDX_ACTIVE).  Which lane holds which pixel
    // is free -- the epilogue uses the same map -- so the first hardware group takes the 16 consecutive pixels of row 0, the second
    // those of row 1: conflict-free.
    const bool grp_b = (p >= 4 && p < 12) || (p >= 16 && p < 20) || p >= 28;
    const int rank16 = p < 4 ? p : p < 12 ? p - 4 : p < 16 ? p - 8 : p < 20 ? p - 8 : p < 28 ? p - 12 : p - 16;
    auto pix_of = [&](int i, int& ty, int& tx) {                 // patch pixel of this lane in the wave's MFMA tile i
        if (TWLOG == 5) { ty = wm * NI + i; tx = p; }
        else { ty = wm * 2 * NI + 2 * i + (grp_b ? 1 : 0); tx = rank16; }
    };
    unsigned abase[NI];                                          // LDS byte offset of tile i's pixel, tap (0,0), ks 0, hi plane
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        int ty, tx;
        pix_of(i, ty, tx);
        abase[i] = (unsigned)((ty * HW2 + tx) * RS + kg * 16);
    }
    int slot_hy[NSLOT], slot_hx[NSLOT];
    unsigned slot_dst[NSLOT];
    bool slot_ok[NSLOT];
#pragma unroll
    for (int sl = 0; sl < NSLOT; ++sl) {
        const int q = wave + 4 * sl;
        const int hr = q * 8 + (lane >> 3);
        slot_ok[sl] = q < HALO_INSTR && hr < HROWS;
        slot_hy[sl] = hr / HW2;
        slot_hx[sl] = hr - slot_hy[sl] * HW2;
        slot_dst[sl] = (unsigned)(hr * RS + (lane & 7) * 8);     // 4 bf16 = 8 B per lane and plane
    }
    const int lchan = (lane & 7) * 4;

    struct TilePos { int img, py0, px0, nt, ph; };
    auto tile_pos = [&](int j) -> TilePos {
        int t = vbid + j * (int)gridDim.x;
        TilePos tp;
        tp.nt = t % g.tiles_n; t /= g.tiles_n;
        tp.ph = 0;
        if (pmode == 1) { tp.ph = t & 3; t >>= 2; }              // the four phases of a patch side by side: they share its halo in L2
        const int txi = t % tiles_x; t /= tiles_x;
        const int tyi = t % tiles_y;
        tp.img = t / tiles_y; tp.py0 = tyi * TH; tp.px0 = txi * TW;
        return tp;
    };
    u32x4 hreg[NSLOT];
    auto load_halo = [&](const TilePos& tp, int cu) {
        // phase_mode 2: unit cu = phase * nch + chunk reads the phase (a, b) of the full-resolution source at stride 2
        const int uph = pmode == 2 ? cu / nch : 0, c = pmode == 2 ? cu - uph * nch : cu;
        const int ss = pmode == 2 ? 2 : 1, sa = pmode == 2 ? (uph >> 1) : 0, sb = pmode == 2 ? (uph & 1) : 0;
        const float* ximg = x + (int64_t)tp.img * g.h_in * g.w_in * g.cin + c * 32 + lchan;
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            const int iy = tp.py0 + slot_hy[sl] - HM, ix = tp.px0 + slot_hx[sl] - HM;
            const bool ok = slot_ok[sl] && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w;
            const float* src = ximg + ((int64_t)((iy * ss + sa) >> g.ups) * g.w_in + ((ix * ss + sb) >> g.ups)) * g.cin;
            const void* sp = ok ? (const void*)src : (const void*)zeros;      // select, not branch
            hreg[sl] = *reinterpret_cast<const u32x4*>(sp);
        }
    };
    // split the four floats of every slot into (hi, lo) bf16 and write both planes
    auto store_halo = [&](char* buf) {
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            if (wave + 4 * sl >= HALO_INSTR) continue;
            const float f0 = __uint_as_float(hreg[sl][0]), f1 = __uint_as_float(hreg[sl][1]);
            const float f2 = __uint_as_float(hreg[sl][2]), f3 = __uint_as_float(hreg[sl][3]);
            const unsigned h01 = pack_bf16x2(f0, f1), h23 = pack_bf16x2(f2, f3);
            const float l0 = f0 - __uint_as_float(h01 << 16), l1 = f1 - __uint_as_float(h01 & 0xffff0000u);
            const float l2 = f2 - __uint_as_float(h23 << 16), l3 = f3 - __uint_as_float(h23 & 0xffff0000u);
            const u32x2 hi = {h01, h23}, lo = {pack_bf16x2(l0, l1), pack_bf16x2(l2, l3)};
            *reinterpret_cast<u32x2*>(buf + slot_dst[sl]) = hi;
            *reinterpret_cast<u32x2*>(buf + PLANE + slot_dst[sl]) = lo;
        }
    };
    const unsigned lane16 = (unsigned)lane * 16;
    const char* wroot = reinterpret_cast<const char*>(wp);
    // weights of unit cu of a tile: (phase block,) 32-cout tile, chunk.  Phase: the tile's (mode 1) or the unit's (mode 2)
    const int64_t phase_bytes = (int64_t)(g.tiles_n * 4) * nch * UNITW;
    auto unit_ph = [&](const TilePos& tp, int cu) -> int { return pmode == 2 ? cu / nch : tp.ph; };
    auto unit_w = [&](const TilePos& tp, int cu, int j) -> const char* {
        const int cot = tp.nt * 4 + (WL ? wn : wn * 2 + j);
        const int ph = unit_ph(tp, cu), cc = pmode == 2 ? cu - ph * nch : cu;
        const int wph = (NTAP == 4 && g.phase_rev) ? 3 - ph : ph;
        return wroot + wph * phase_bytes + ((int64_t)cot * nch + cc) * UNITW;
    };
    // window offset of a phase inside the halo: mode 1 (a, b), mode 2 (1 - a, 1 - b)
    auto phase_off = [&](int ph) -> unsigned {
        const int a = ph >> 1, b = ph & 1;
        return pmode == 1 ? (unsigned)((a * HW2 + b) * RS) : pmode == 2 ? (unsigned)(((1 - a) * HW2 + (1 - b)) * RS) : 0u;
    };

    f32x16 acc[NI][NJ];

    TilePos cur = tile_pos(0);
    load_halo(cur, 0);
    frag_t bw[NJ][2][2];                                         // [j][ks][hi / lo] of the current tap
    const char* wcur[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) wcur[j] = unit_w(cur, 0, j);
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) bw[j][ks][hl] = *reinterpret_cast<const frag_t*>(wcur[j] + (ks * 2 + hl) * 1024 + lane16);
    store_halo(smem);
    __syncthreads();

    int tj = 0, c = 0;
    for (int u = 0; u < units; ++u) {
        const unsigned boff = (unsigned)((u & 1) * BUF);
        int ntj = tj, nc = c + 1;
        if (nc == nun) { nc = 0; ntj = tj + 1; }
        const bool has_next = u + 1 < units;
        if (!has_next) { ntj = tj; nc = c; }                     // clamp: loads stay unconditional
        const TilePos nxt = (ntj == tj) ? cur : tile_pos(ntj);
        load_halo(nxt, nc);                                      // in flight during this unit's MFMAs
        const char* wnxt[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) wnxt[j] = unit_w(nxt, nc, j);
        const char* lbase[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) lbase[i] = smem + boff + abase[i] + (NTAP == 4 ? phase_off(unit_ph(cur, c)) : 0u);

        // pixel fragments (hi and lo plane) run one (tap, k-substep) phase ahead of the MFMAs that consume them
        frag_t a[2][NI][2];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            a[0][i][0] = *reinterpret_cast<const frag_t*>(lbase[i]);
            a[0][i][1] = *reinterpret_cast<const frag_t*>(lbase[i] + PLANE);
        }
#pragma unroll
        for (int tap = 0; tap < NTAP; ++tap) {
            const int toff = ((tap / TWD) * HW2 + (tap % TWD)) * RS;    // compile-time after unrolling (NTAP 1: 0)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                if (ks == 0) {
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        a[1][i][0] = *reinterpret_cast<const frag_t*>(lbase[i] + toff + 32);
                        a[1][i][1] = *reinterpret_cast<const frag_t*>(lbase[i] + PLANE + toff + 32);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * NI, 0);
                } else if (tap < NTAP - 1) {
                    const int toff1 = (((tap + 1) / TWD) * HW2 + ((tap + 1) % TWD)) * RS;
#pragma unroll
                    for (int i = 0; i < NI; ++i) {
                        a[0][i][0] = *reinterpret_cast<const frag_t*>(lbase[i] + toff1);
                        a[0][i][1] = *reinterpret_cast<const frag_t*>(lbase[i] + PLANE + toff1);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * NI, 0);
                }
                // three products; the dependent MFMAs on one accumulator are NI * NJ instructions apart
                if (tap == 0 && ks == 0 && c == 0) {             // first MFMA of a tile starts from C = 0
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[j][ks][0], a[ks][i][0], zero, 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[j][ks][0], a[ks][i][0], acc[i][j], 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[j][ks][0], a[ks][i][1], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[j][ks][1], a[ks][i][0], acc[i][j], 0, 0, 0);
                // rolling prefetch of the same slot for the next tap (next unit after tap 8)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl) {
                        const char* src = (tap == NTAP - 1) ? wnxt[j] + (ks * 2 + hl) * 1024 : wcur[j] + (((tap + 1) * 2 + ks) * 2 + hl) * 1024;
                        bw[j][ks][hl] = *reinterpret_cast<const frag_t*>(src + lane16);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (c == nun - 1) {                                      // tile finished: fp32 epilogue from the accumulators
            const int n0 = cur.nt * 128;
            const int cwave = WL ? wn * 32 : wn * 64;            // first cout of this wave inside the 128-cout tile
            const bool plain = act == 0 && g.out_gain == 1.0f && (g.cout & 127) == 0;
            const bool scaled = g.acc_scale != 1.0f;
            // phase_mode 1: the tile's pixels are the phase (a, b) of a 2h x 2w output
            const int ds = pmode == 1 ? 2 : 1, da = pmode == 1 ? (cur.ph >> 1) : 0, db = pmode == 1 ? (cur.ph & 1) : 0;
            const int oh = g.h * ds, ow = g.w * ds;
            // GroupNorm statistics of the stored output (g.gn_ws: vqk_conv2d_fprop_x3_gnstats; the consumer's GroupNorm then skips its
            // statistics pass over this tensor): per lane the sums of its 4-channel halves over the tile's pixels, folded over the 32
            // pixel lanes at the end of the tile, one fp64 atomic per (group, sum | sum of squares) and wave
            const bool want_stats = plain && g.gn_ws != nullptr;
            float gs[NJ][2][2], gq[NJ][2][2];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int qp = 0; qp < 2; ++qp)
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) { gs[j][qp][hf] = 0.f; gq[j][qp][hf] = 0.f; }
            if (plain) {
                // lanes l and l + 32 hold the two halves of every 8-cout run: one v_permlane32_swap per value pairs them up so
                // that each lane owns 8 consecutive couts = 32 contiguous bytes
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    int ty, tx;
                    pix_of(i, ty, tx);
                    const int64_t pix = ((int64_t)cur.img * oh + (cur.py0 + ty) * ds + da) * ow + (cur.px0 + tx) * ds + db;
                    const int64_t o0 = pix * g.cout + n0 + cwave + 8 * kg;
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int qp = 0; qp < 2; ++qp) {
                            const int cw = j * 32 + 16 * qp;
                            float v[8];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const unsigned lo = __float_as_uint(acc[i][j][8 * qp + e]);
                                const unsigned hi = __float_as_uint(acc[i][j][8 * qp + 4 + e]);
                                const auto r = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
                                v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
                            }
                            if (scaled) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] *= g.acc_scale;
                            }
                            if (bias) {
                                const float* bp = bias + n0 + cwave + 8 * kg + cw;
                                const f32x4 b0 = *reinterpret_cast<const f32x4*>(bp), b1 = *reinterpret_cast<const f32x4*>(bp + 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v[e] += b0[e]; v[4 + e] += b1[e]; }
                            }
                            if (res) {
                                const f32x4 r0 = *reinterpret_cast<const f32x4*>(res + o0 + cw);
                                const f32x4 r1 = *reinterpret_cast<const f32x4*>(res + o0 + cw + 4);
#pragma unroll
                                for (int e = 0; e < 4; ++e) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                            }
                            const f32x4 s0 = {v[0], v[1], v[2], v[3]}, s1 = {v[4], v[5], v[6], v[7]};
                            *reinterpret_cast<f32x4*>(y + o0 + cw) = s0;
                            *reinterpret_cast<f32x4*>(y + o0 + cw + 4) = s1;
                            if (want_stats) {
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    gs[j][qp][0] += v[e]; gq[j][qp][0] = __builtin_fmaf(v[e], v[e], gq[j][qp][0]);
                                    gs[j][qp][1] += v[4 + e]; gq[j][qp][1] = __builtin_fmaf(v[4 + e], v[4 + e], gq[j][qp][1]);
                                }
                            }
                        }
                }
                if (want_stats) {
                    const int cpg = g.gn_cpg, ngroups = g.cout / cpg;
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int qp = 0; qp < 2; ++qp) {
                            float a[2] = {gs[j][qp][0], gs[j][qp][1]}, q[2] = {gq[j][qp][0], gq[j][qp][1]};
#pragma unroll
                            for (int off = 1; off < 32; off <<= 1) {             // the 32 pixel lanes of this k-half
#pragma unroll
                                for (int hf = 0; hf < 2; ++hf) { a[hf] += __shfl_xor(a[hf], off, 64); q[hf] += __shfl_xor(q[hf], off, 64); }
                            }
                            if (cpg >= 8) { a[0] += a[1]; q[0] += q[1]; }         // the lane's eight couts are one group (or half of one)
                            if (cpg >= 16) { a[0] += __shfl_xor(a[0], 32, 64); q[0] += __shfl_xor(q[0], 32, 64); }     // + the other k-half's eight
                            const int cb = n0 + cwave + j * 32 + 16 * qp + 8 * kg;          // first of this lane's eight couts
                            if (p == 0 && (cpg < 16 || kg == 0)) {
                                double* w0 = g.gn_ws + ((int64_t)cur.img * ngroups + cb / cpg) * 2;
                                atomicAdd(w0, (double)a[0]); atomicAdd(w0 + 1, (double)q[0]);
                                if (cpg == 4) { atomicAdd(w0 + 2, (double)a[1]); atomicAdd(w0 + 3, (double)q[1]); }
                            }
                        }
                }
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    int ty, tx;
                    pix_of(i, ty, tx);
                    const int64_t pix = ((int64_t)cur.img * oh + (cur.py0 + ty) * ds + da) * ow + (cur.px0 + tx) * ds + db;
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const int co = n0 + cwave + j * 32 + 8 * rq + 4 * kg;
                            if (co < g.cout) {
                                const int64_t o = pix * g.cout + co;
                                f32x4 v;
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    v[e] = x3_act(acc[i][j][4 * rq + e] * g.acc_scale + (bias ? bias[co + e] : 0.0f), act) * g.out_gain;
                                if (res) {
                                    const f32x4 r = *reinterpret_cast<const f32x4*>(res + o);
                                    v += r;
                                }
                                *reinterpret_cast<f32x4*>(y + o) = v;
                            }
                        }
                }
            }
        }
        if (has_next) store_halo(smem + ((u + 1) & 1) * BUF);
        __syncthreads();
        cur = nxt; tj = ntj; c = nc;
#pragma unroll
        for (int j = 0; j < NJ; ++j) wcur[j] = wnxt[j];
    }
}

// ------------------------------------------------------------------------------------------------
// conv3x3_wgrad_x3_kernel: the weight gradient of the split-product mode WITHOUT pair tensors -- fp32 x and fp32 dy are loaded into
// registers (16 B = 4 channels of one pixel per lane and slot), split into (hi, lo) bf16 there and written to the hi / lo planes of
// ONE 44.5-KiB LDS stage; the matrix loop forms all three products from the staged data (3 MFMAs per fragment pair instead of three
// passes over three pair tensors: 3x the arithmetic intensity of the bf16 kernel per staged byte, no split passes, no folded tile
// classes).  Geometry of conv3x3_wgrad_halo_kernel<false> (conv.hip): block = 64 co x 64 ci x 9 taps over a range of 8x8-pixel
// patches, wave (i, j) owns the 32 x 32 tile of all taps (144 accumulators), fragments by ds_read_b64_tr_b16 from [rows][64 B] half
// tiles.  Pipeline: the loads of patch p + 1 are in flight during the MFMAs of patch p; store phase and matrix phase are separated
// by two barriers, and the second resident block of the CU (2 x 44.5 KiB) computes while this one converts.
// What bounds it (timing-only ablations VQK_X3ABL, profiles/round6_x3_wgrad_ablation.txt; 256 -> 256 @128^2, 32 images, 1509 us): one
// product instead of three 812 us, half the x-fragment LDS reads -3 %, no split arithmetic -7 %, a sixteenth of the final atomics
// +-0 (small maps -15 us); a role-split form (4 matrix + 4 staging waves, two stages, one block per CU) ran the large maps at the
// same rate and cost the step 2.7 ms of overlap with the GroupNorm backward -- not kept.  1855 GFLOP of bf16 MFMAs in 1453 us =
// 1277 TF: the rate at which the bf16 role-split kernels sit too (the power wall of DESIGN.md section 3).
// dW[co][tap][ci] += scale * sum_pix dy[pix][co] * x[pix (+) tap][ci]   (autoencoder.py:57-60, :102-105, :132, :153)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8_t x3_tr_frag2(const char* p) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VQK_LDS s16x4*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((VQK_LDS s16x4*)(p + 256));      // rows +4
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

#ifndef VQK_X3ABL
#define VQK_X3ABL 0          // timing-only ablations of conv3x3_wgrad_x3_kernel (tools/x3_wgrad_abl.sh): 1 one product, 2 half the x-fragment reads, 4 no split arithmetic in the stage, 8 no final atomics
#endif
// NTAP = 4: the weight gradient of the 2x2-RESAMPLING convs in phase form -- 4/9 of the MFMAs for the same staged bytes.  Patches walk the
// LOW-resolution grid (g.h x g.w), blockIdx.y = split * 4 + phase (a, b); per phase a 2x2 window of the low-resolution halo:
//   g.phase_mode 1 (nearest-x2 upsample + conv, autoencoder.py:102-105): dy [n][2h][2w] is gathered at stride 2, offset (a, b); window
//     offset (a, b); S_ab[r][s] = sum_pix dy_ab[pix] x[pix + (a + r - 1, b + s - 1)] is the gradient of every 3x3 tap that lands on that
//     low-resolution pixel: rows {0} | {1, 2} for a = 0, {0, 1} | {2} for a = 1 (the phase sums of layout 2 / 6, transposed);
//   g.phase_mode 2 (conv + 2x2 average pool, :89-91, dy = the POOLED gradient [n][h][w], x [n][2h][2w]): the 4x4 stride-2 window
//     G[u][v] = sum_pix dy[pix] x[2 pix + (u - 1, v - 1)], u = 2 r + 1 - a: x gathered at stride 2, offset (a, b), window offset
//     (1 - a, 1 - b); dW[ky][kx] = scale * sum_{a', b'} G[ky + a'][kx + b'] -- the same row sets with a := 1 - a.
template <int NTAP>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_x3_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  float* __restrict__ dw, ConvGeom g, int patches_per_split) {
    constexpr int PWD = 8, PIX = 64, HWD = 10, HROWS = 100, X_ROWS = 112;
    // bytes per 32-channel half tile (bf16), + 64: the two halves of a row are written by lanes 0-7 / 8-15 of ONE ds_write_b64 group --
    // at a multiple of 128 B apart they would share their banks (measured 20 % conflict cycles)
    constexpr int DY_HALF = PIX * 64 + 64, X_HALF = X_ROWS * 64 + 64;
    constexpr int PLANE = 2 * DY_HALF + 2 * X_HALF;              // 22784: one of (hi, lo)
    constexpr int DY_UNITS = PIX * 16, UNITS = DY_UNITS + HROWS * 16, NSLOT = (UNITS + 255) / 256;      // 16-byte fp32 units: 2624 -> 11 slots
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_ci = g.cin >> 6;
    const int vb = xcd_remap((int)(blockIdx.x + gridDim.x * blockIdx.y), (int)(gridDim.x * gridDim.y));
    const int bx = vb % (int)gridDim.x, by = vb / (int)gridDim.x;
    const int tco = bx / tiles_ci, tci = bx - tco * tiles_ci;
    const int co0 = tco * 64, ci0 = tci * 64;
    const int pw = g.w >> 3, ph = g.h >> 3;
    const int total_patches = g.n * ph * pw;
    const int phase = NTAP == 4 ? (by & 3) : 0, split = NTAP == 4 ? (by >> 2) : by;
    const int p_begin = split * patches_per_split;
    const int p_end = min(total_patches, p_begin + patches_per_split);
    if (p_begin >= p_end) return;
    // phase forms: pixel stride / offset of the gathered operand, window offset inside the halo
    const int pa_ = phase >> 1, pb_ = phase & 1;
    const int dsy = (NTAP == 4 && g.phase_mode == 1) ? 2 : 1, xsx = (NTAP == 4 && g.phase_mode == 2) ? 2 : 1;
    const int dya = dsy == 2 ? pa_ : 0, dyb = dsy == 2 ? pb_ : 0, xa = xsx == 2 ? pa_ : 0, xb = xsx == 2 ? pb_ : 0;
    const int oa = NTAP == 4 ? (g.phase_mode == 1 ? pa_ : 1 - pa_) : 0, ob = NTAP == 4 ? (g.phase_mode == 1 ? pb_ : 1 - pb_) : 0;

    const int wi = wave >> 1, wj = wave & 1;
    f32x16 acc[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // per-lane slots: unit u = tid + 256 * sl.  Slots 0..3 are dy (1024 units = 64 pixels x 16 four-channel groups): pixel row
    // (tid >> 4) + 16 sl; slots 4..10 the x halo: halo row (tid >> 4) + 16 (sl - 4) < 100.  Channels 4 * (tid & 15) in every slot.
    // Everything is affine in sl: two per-thread constants instead of per-slot register arrays (the 144 accumulators leave no room).
    static_assert(NSLOT == 11 && DY_UNITS == 1024, "slot split");
    const int c4 = tid & 15, trow = tid >> 4;
    const int ch = c4 * 4;
    const unsigned dst0 = (unsigned)((c4 >> 3) * DY_HALF + trow * 64 + (c4 & 7) * 8);                  // + sl * 1024
    const unsigned dst1 = (unsigned)(2 * DY_HALF + (c4 >> 3) * X_HALF + trow * 64 + (c4 & 7) * 8);     // + (sl - 4) * 1024
    u32x4 stage[NSLOT];
    unsigned okm = 0;                                            // bit sl: slot sl holds a pixel inside the image (applied at store time: a select on
                                                                 // the loaded value here made the wave wait for the loads before its MFMAs)
    auto load_patch = [&](int patch) {
        const int img = patch / (ph * pw), rem = patch - img * (ph * pw);
        const int pyi = rem / pw, pxi = rem - pyi * pw;
        const int py0 = pyi * 8, px0 = pxi * PWD;
        const int dyw = g.w * dsy;
        const float* dyp = dy + (((int64_t)img * g.h * dsy + (py0 + (trow >> 3)) * dsy + dya) * dyw + (px0 + (trow & 7)) * dsy + dyb) * g.cout + co0 + ch;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl)                           // patch rows (trow >> 3) + 2 sl
            stage[sl] = *reinterpret_cast<const u32x4*>(dyp + (int64_t)(2 * sl * dsy) * dyw * g.cout);
        const int xh = NTAP == 4 ? g.h * xsx : g.h_in, xw = NTAP == 4 ? g.w * xsx : g.w_in;
        const float* ximg = x + (int64_t)img * xh * xw * g.cin + ci0 + ch;
#pragma unroll
        for (int sl = 4; sl < NSLOT; ++sl) {
            const int row = trow + 16 * (sl - 4);
            const int hy = row / HWD, hx = row - hy * HWD;
            const int iy = py0 + hy - 1, ix = px0 + hx - 1;
            const bool ok = row < HROWS && iy >= 0 && iy < g.h && ix >= 0 && ix < g.w;
            const int cy = !ok ? 0 : NTAP == 4 ? iy * xsx + xa : iy >> g.ups;          // clamped address: the load stays unconditional
            const int cx = !ok ? 0 : NTAP == 4 ? ix * xsx + xb : ix >> g.ups;
            stage[sl] = *reinterpret_cast<const u32x4*>(ximg + ((int64_t)cy * xw + cx) * g.cin);
            okm = (okm & ~(1u << sl)) | (ok ? 1u << sl : 0u);
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int sl = 0; sl < NSLOT; ++sl) {
            if (sl == NSLOT - 1 && trow >= HROWS - 16 * (NSLOT - 5)) continue;       // halo rows 100..111 do not exist
            const bool ok = sl < 4 || ((okm >> sl) & 1u);
            const float f0 = ok ? __uint_as_float(stage[sl][0]) : 0.f, f1 = ok ? __uint_as_float(stage[sl][1]) : 0.f;
            const float f2 = ok ? __uint_as_float(stage[sl][2]) : 0.f, f3 = ok ? __uint_as_float(stage[sl][3]) : 0.f;
#if VQK_X3ABL & 4          // TIMING ONLY: raw words instead of the (hi, lo) split
            const u32x2 hi = {stage[sl][0], stage[sl][1]}, lo = {stage[sl][2], stage[sl][3]};
#else
            const unsigned h01 = pack_bf16x2(f0, f1), h23 = pack_bf16x2(f2, f3);
            const float l0 = f0 - __uint_as_float(h01 << 16), l1 = f1 - __uint_as_float(h01 & 0xffff0000u);
            const float l2 = f2 - __uint_as_float(h23 << 16), l3 = f3 - __uint_as_float(h23 & 0xffff0000u);
            const u32x2 hi = {h01, h23}, lo = {pack_bf16x2(l0, l1), pack_bf16x2(l2, l3)};
#endif
            const unsigned d = sl < 4 ? dst0 + sl * 1024 : dst1 + (sl - 4) * 1024;
            *reinterpret_cast<u32x2*>(smem + d) = hi;
            *reinterpret_cast<u32x2*>(smem + PLANE + d) = lo;
        }
    };

    const int li = lane & 15, grp = (lane >> 4) & 1, kgrp = lane >> 5;
    const unsigned frag_lane = (unsigned)((li >> 2) * 64 + 32 * grp + 8 * (li & 3));
    const unsigned a_lane = (unsigned)(wi * DY_HALF) + frag_lane + (unsigned)(kgrp * 8 * 64);
    // k-group 1 = pixels 8..15 of the MFMA's 16 = the next patch row: + HWD halo rows
    const unsigned b_lane = (unsigned)(2 * DY_HALF + wj * X_HALF) + frag_lane + (unsigned)(kgrp * HWD * 64);
    const char* pa = smem + a_lane;
    const char* pb = smem + b_lane;

    load_patch(p_begin);
    for (int pch = p_begin; pch < p_end; ++pch) {
        store_patch();                                           // (waits for the loads of this patch)
        __syncthreads();
        if (pch + 1 < p_end) load_patch(pch + 1);                // in flight during the MFMAs below
#pragma unroll
        for (int gk = 0; gk < PIX / 16; ++gk) {                  // 16 pixels = patch rows 2 gk, 2 gk + 1
            const bf16x8_t ah = x3_tr_frag2(pa + gk * 16 * 64), al = x3_tr_frag2(pa + PLANE + gk * 16 * 64);
#pragma unroll
            for (int t = 0; t < NTAP; ++t) {
                const int off = NTAP == 4 ? ((2 * gk + oa + t / 2) * HWD + ob + (t % 2)) * 64 : ((2 * gk + t / 3) * HWD + (t % 3)) * 64;
#if VQK_X3ABL & 2          // TIMING ONLY: half the x-fragment reads
                const bf16x8_t bh = x3_tr_frag2(pb + off), bl = bh;
#else
                const bf16x8_t bh = x3_tr_frag2(pb + off), bl = x3_tr_frag2(pb + PLANE + off);
#endif
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
#if !(VQK_X3ABL & 1)       // TIMING ONLY (1): one product instead of three
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
#endif
            }
        }
        __syncthreads();                                         // everyone left the stage before it is rewritten
    }
    const int ci = ci0 + wj * 32 + (lane & 31);
    if constexpr (NTAP == 4) {
        // window tap (r, s) of this phase -> every 3x3 tap it stands for (1, 2 or 4 of them: nine tile additions per block, as in the tap form)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int r_ = t >> 1, s_ = t & 1;
            const int ky0 = r_ == 0 ? 0 : (oa ? 2 : 1), ky1 = r_ == 0 ? (oa ? 1 : 0) : 2;
            const int kx0 = s_ == 0 ? 0 : (ob ? 2 : 1), kx1 = s_ == 0 ? (ob ? 1 : 0) : 2;
            for (int ky = ky0; ky <= ky1; ++ky)
                for (int kx = kx0; kx <= kx1; ++kx) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
                        atomicAdd(dw + ((int64_t)co * 9 + ky * 3 + kx) * g.cin + ci, acc[t][r] * g.acc_scale);
                    }
                }
        }
    } else {
#pragma unroll
        for (int t = 0; t < NTAP; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = co0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * kgrp;
#if VQK_X3ABL & 8          // TIMING ONLY: one atomic per tap instead of sixteen
                if (r == 0 || acc[t][r] == 123.456f)
#endif
                atomicAdd(dw + ((int64_t)co * 9 + t) * g.cin + ci, acc[t][r] * g.acc_scale);
            }
    }
}

// fp32 [rows][c] -> bf16 [rows][2c]: (hi | lo) per row; a thread owns 8 consecutive channels (two 16-byte loads, two 16-byte stores)
__global__ __launch_bounds__(256) void split_pair_kernel(const float* __restrict__ src, bf16_raw* __restrict__ dst, int64_t rows, int c) {
    const int c8 = c >> 3;
    const int64_t total = rows * c8;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c8;
        const int k = (int)(i - r * c8) * 8;
        const f32x4 a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + r * c + k));
        const f32x4 b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + r * c + k + 4));
        const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
        u32x4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned h = pack_bf16x2(f[2 * q], f[2 * q + 1]);
            hi[q] = h;
            lo[q] = pack_bf16x2(f[2 * q] - __uint_as_float(h << 16), f[2 * q + 1] - __uint_as_float(h & 0xffff0000u));
        }
        *reinterpret_cast<u32x4*>(dst + r * 2 * c + k) = hi;
        *reinterpret_cast<u32x4*>(dst + r * 2 * c + c + k) = lo;
    }
}

}  // namespace

namespace vqkd {

// x fp32 [N, h_in, w_in, Cin], w: layout 5, y fp32 [N, h, w, Cout]; Cin % 32 == 0, h % 8 == 0, w % 16 == 0
int launch_conv3x3_x3(const void* x, const void* w, const float* bias, const void* res, void* y, const void* zeros,
                      const ConvGeom& g, int act, int blocks_cap, hipStream_t st) {
    if ((g.ks != 3 && g.ks != 1) || (g.cin & 31) || (g.h & 7) || (g.w & 15) || (g.cout & 3) || (g.ks == 1 && g.ups)) return VQK_ERR_SHAPE;
    const int total = g.n * (g.h / 8) * (g.w / 16) * g.tiles_n * ((g.ntap == 4 && g.phase_mode == 1) ? 4 : 1);
    const int cap = blocks_cap > 0 ? blocks_cap : 512;
    const dim3 grid((unsigned)(total < cap ? total : cap));
    constexpr int lds = 2 * 2 * 184 * 80;                        // two buffers x (hi, lo) planes of the 10x18 halo (23 pieces of 8 rows)
    if (g.ntap == 4) {                                           // the 2x2-resampling convs in phase form (layout 6)
        if (g.ks != 3 || g.ups || (g.phase_mode != 1 && g.phase_mode != 2)) return VQK_ERR_SHAPE;
        hipLaunchKernelGGL((conv3x3_x3_kernel<4, 1, 4>), grid, dim3(256), lds, st, (const float*)x, (const bf16_raw*)w, bias,
                           (const float*)res, (float*)y, (const char*)zeros, g, act);
        return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
    }
    if (g.ks == 1) {                                             // 1x1: the tile's own 128 pixels (16 pieces), one tap
        hipLaunchKernelGGL((conv3x3_x3_kernel<4, 1, 1>), grid, dim3(256), 2 * 2 * 128 * 80, st, (const float*)x, (const bf16_raw*)w, bias,
                           (const float*)res, (float*)y, (const char*)zeros, g, act);
        return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
    }
    const int wl = VQK_TUNE("X3_WL", 1);
    if (wl)
        hipLaunchKernelGGL((conv3x3_x3_kernel<4, 1>), grid, dim3(256), lds, st, (const float*)x, (const bf16_raw*)w, bias,
                           (const float*)res, (float*)y, (const char*)zeros, g, act);
    else
        hipLaunchKernelGGL((conv3x3_x3_kernel<4, 0>), grid, dim3(256), lds, st, (const float*)x, (const bf16_raw*)w, bias,
                           (const float*)res, (float*)y, (const char*)zeros, g, act);
    return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
}

}  // namespace vqkd

namespace vqkd {
// x fp32 [n][h_in][w_in][cin], dy fp32 [n][h][w][cout], dw fp32 [cout][3][3][cin] +=; cin % 64 == 0, cout % 64 == 0, h % 8 == 0, w % 8 == 0.
// g.ntap == 4 (g.phase_mode 1 | 2): the phase forms -- g.h x g.w is the LOW-resolution grid, see the kernel
int launch_conv3x3_wgrad_x3(const void* x, const void* dy, float* dw, const ConvGeom& g, int blocks_cap, hipStream_t st) {
    if ((g.cin & 63) || (g.cout & 63) || (g.h & 7) || (g.w & 7) || g.ks != 3) return VQK_ERR_SHAPE;
    const int phases = g.ntap == 4 ? 4 : 1;
    if (phases == 4 && (g.ups || (g.phase_mode != 1 && g.phase_mode != 2))) return VQK_ERR_SHAPE;
    const int tiles = (g.cout >> 6) * (g.cin >> 6);
    const int total_patches = g.n * (g.h >> 3) * (g.w >> 3);
    // split-K over pixel patches: the cost model of the bf16 kernels (conv.hip::wgrad_general) with a third of their atomic passes
    // per multiply-add -- s* = sqrt(c * pixels / tiles) under the block cap (two blocks per CU)
    int cap = blocks_cap > 0 ? blocks_cap : 512;
    if (phases == 4) cap = cap * VQK_TUNE("X3_WGRAD_PHASE_CAP_PCT", 150) / 100;      // 166 VGPRs: a third block fits a CU (step 70.39 -> 69.89 ms)
    const double coef = (phases == 4 ? VQK_TUNE("X3_WGRAD_PHASE_COEF_E4", 3200) : VQK_TUNE("X3_WGRAD_COEF_E4", 3200)) * 1e-4;
    // (phase forms: the four phases are four times the blocks -- a quarter of the splits each, the same number of atomic passes)
    int splits = (int)(sqrt(coef * (double)g.m * phases / tiles) / phases + 0.5);
    if (splits > (cap + tiles * phases - 1) / (tiles * phases)) splits = (cap + tiles * phases - 1) / (tiles * phases);
    if (splits > (total_patches + 3) / 4) splits = (total_patches + 3) / 4;          // >= 256 pixels per block
    if (splits < 1) splits = 1;
    const int pps = (total_patches + splits - 1) / splits;
    splits = (total_patches + pps - 1) / pps;
    constexpr int lds = 2 * 22784;
    if (phases == 4)
        hipLaunchKernelGGL(conv3x3_wgrad_x3_kernel<4>, dim3((unsigned)tiles, (unsigned)(splits * 4)), dim3(256), lds, st, (const float*)x,
                           (const float*)dy, dw, g, pps);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_x3_kernel<9>, dim3((unsigned)tiles, (unsigned)splits), dim3(256), lds, st, (const float*)x,
                           (const float*)dy, dw, g, pps);
    return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
}
}  // namespace vqkd

extern "C" int vqk_split_pair_f32(const float* src, void* dst, int64_t rows, int c, void* stream) {
    VQK_REQUIRE(src && dst, VQK_ERR_ARG);
    VQK_REQUIRE(rows >= 0 && c > 0 && (c & 7) == 0, VQK_ERR_SHAPE);
    VQK_REQUIRE(vqk_aligned16(src) && vqk_aligned16(dst), VQK_ERR_ALIGN);
    if (rows == 0) return VQK_OK;
    hipLaunchKernelGGL(split_pair_kernel, dim3((unsigned)vqk_grid_1d(rows * (c >> 3), 256, 256 * 16)), dim3(256), 0, vqk_stream(stream),
                       src, (bf16_raw*)dst, rows, c);
    VQK_CHECK_LAUNCH();
    return VQK_OK;
}
