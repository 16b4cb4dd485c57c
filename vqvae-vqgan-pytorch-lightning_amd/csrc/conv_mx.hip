// ------------------------------------------------------------------------------------------------
// conv3x3_mx_kernel: the 3x3 stream convolution (fprop / dgrad, bf16, NHWC) split into MATRIX waves and AUXILIARY waves.
// Replaces the F.conv2d calls of vqvae/modules/autoencoder.py:57-60, :102-105, :132, :153 on the large maps.
//
// Why (measured on the single-role stream kernel, tools/ab_conv.py + tools/ab_mx.sh, round 2): its MFMA loop alone runs at
// 1.25-1.45 PF, the halo loads (HBM latency in the same in-order vector-memory queue as the weight stream) and the epilogue
// (unpack / pack / residual loads / stores on the SIMD that issues the MFMAs) cost 13 % + 11 % on top, and hipcc schedules
// the loop's ds_reads just-in-time (each phase opened with an exposed LDS round trip).  So:
//
//   block = 512 threads, one block per CU (persistent over (patch, cout-tile) tiles x 32-channel chunks = "units"):
//     waves 0-3  "M": ds_read_b128 pixel fragments (80-byte padded halo rows, conflict-free), weight fragments straight
//                from L2 by buffer loads with scalar offsets (ring three taps deep), 144 MFMAs per unit in 18 phases whose
//                instruction order is pinned (sched_group_barrier): the NEXT phase's four fragment reads are issued
//                between the first MFMAs of a phase, the two weight loads behind its last MFMA.  At the end of a tile the
//                128 accumulators are rounded to bf16 and parked in an LDS staging tile.
//     waves 4-7  "X": everything that touches HBM.  The (TH+2)x(TW+2) halo of unit u+2 is fetched by LDS-DMA
//                (buffer_load_dwordx4 ... lds) into one of THREE halo buffers while unit u is computed -- no staging
//                registers, no ds_write; the padded row layout is produced by the per-lane SOURCE address (every fifth
//                16-byte slot is a pad: its lane reads out of range, which fetches nothing).  Image borders are
//                out-of-range buffer offsets as well (zeros, no zero page, no branch).  The finished tile is drained from
//                the staging tile: bias / residual add, optional 2x2 pooling, 16-byte stores of whole 256-byte pixel rows.
//   One s_barrier per unit joins the two groups.  Each SIMD hosts one M and one X wave.
//
// Tiles: static share (tile j of block b = b + j * grid in XCD-contiguous block order) or, with ConvGeom::tq, a DYNAMIC queue:
// the blocks of an XCD draw the tiles of that XCD's static sequence from one atomic counter (same tiles, same order, same L2
// neighbourhoods).  The first X wave fetches ahead -- the atomic of tile j+1 is issued four units before the tile's first
// unit and published through an 8-entry LDS ring, so no wave ever waits for it -- and a block that starts late (its CU was
// held by a collective's kernel or another stream's) finds less work instead of a fixed share.  Every block makes exactly ONE
// fetch beyond the end of its XCD's sequence; the last block to do so (census word) zeroes the counters again.
//
// LDS: 3 halo buffers (27 KiB) + the staging tile (256 pixels x 272 bytes) + 1 KiB of statistics partials + the 64-byte
// tile ring = 150 KiB.
// Numerics: the convolution sum is rounded to bf16 ONCE more than in the stream kernel when a bias / residual / pooling
// follows (the epilogue arithmetic runs on the parked bf16 values, in fp32).
// ------------------------------------------------------------------------------------------------
#include "conv_geom.h"
#include <stdlib.h>

namespace {
using vqkd::ConvGeom;
using vqkd::pack_bf16x2;
using vqkd::xcd_remap;

#ifndef VQK_MXABL
#define VQK_MXABL 0          // timing-only ablation bits (tools/ab_build.sh): never set in the shipped build (32: half the weight stream)
#endif
#ifndef VQK_MX_RD
#define VQK_MX_RD 6          // weight ring depth in phases ((tap, k-substep) pairs; divides 18)
#endif
#ifndef VQK_MX_NT
#define VQK_MX_NT 2          // output stores: 0 default policy, 2 nontemporal
#endif
#ifndef VQK_MX_HALO_AUX
#define VQK_MX_HALO_AUX 0    // cache policy of the halo LDS-DMA: 0 default, 2 nontemporal
#endif
#ifndef VQK_MX_PRIO
#define VQK_MX_PRIO 0        // s_setprio of the matrix waves
#endif
#ifndef VQK_MX_PACE
#define VQK_MX_PACE 0        // 1: the drain of a parked tile is spread over the NEXT tile's units 0 .. nun-2 instead of bursting in unit 0 -- measured SLOWER (round 6, profiles/round6_mx_phase_attribution.txt: 598 vs 587 us at 128 -> 128 @256^2, the 1x1 form 104 vs 73 us, the step +0.57 ms): what the matrix waves lose is proportional to the auxiliary waves' work, not to its burstiness
#endif
#ifndef VQK_MX_WL
#define VQK_MX_WL 0          // matrix-wave layout: 0 = 2 (pixels) x 2 (couts), a wave owns PIX/2 pixels x 64 couts; 1 = 1 x 4, PIX pixels x 32 couts:
#endif                       // half the weight-fragment loads per MFMA (the CU's vector-memory path, profiles/round6_mx_phase_attribution.txt), twice the LDS fragment reads
#ifndef VQK_MX_JOUTER
#define VQK_MX_JOUTER 0      // MFMA order inside a phase: 0 pixel-fragment-major, 1 weight-fragment-major
#endif
#ifndef VQK_MX_PIN
#define VQK_MX_PIN 1         // pinned instruction order inside a phase
#endif

#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

// tap -> pixel offset inside a halo buffer.  Plain form: window position (tap / TWD, tap % TWD) of the (TH+2) x (TW+2) halo.
// Stride-2 form (S2): the halo buffer holds the FOUR parity sub-images of the (2 TH + 1) x (2 TW + 1) input patch, each
// (TH+1) x (TW+1) pixels; tap (ky, kx) reads sub-image (ky & 1, kx & 1) at unit stride, shifted by (ky >> 1, kx >> 1).
__host__ __device__ constexpr int mx_tap_pix(bool s2, int tap, int twd, int hw2, int shp) {
    return s2 ? (((tap / 3) & 1) * 2 + ((tap % 3) & 1)) * shp + ((tap / 3) >> 1) * hw2 + ((tap % 3) >> 1)
              : (tap / twd) * hw2 + (tap % twd);
}

// TAPW: taps per window row when NTAP = 2 (2: a 1x2 window, 1: a 2x1 window -- the mixed-parity phases of a stride-2 conv's
// data gradient).  S2: the stride-2 3x3 conv without padding of the StyleGAN2 discriminator (input (2h+1) x (2w+1)).
template <int TWLOG, bool POOL, int NTAP, int PIXELS = 256, int TAPW = 0, bool S2 = false>
__global__ __launch_bounds__(512, 2) void conv3x3_mx_kernel(const bf16_raw* __restrict__ x, const bf16_raw* __restrict__ wp,
                                                            const float* __restrict__ bias,
                                                            const bf16_raw* __restrict__ res, bf16_raw* __restrict__ y,
                                                            ConvGeom g) {
    constexpr int HM = NTAP == 1 ? 0 : 1;                        // halo margin: a 1x1 conv (NTAP = 1) reads the tile's own pixels only
    constexpr int PIX = PIXELS, TW = 1 << TWLOG, TH = PIX / TW;
    constexpr int HW2 = S2 ? TW + 1 : TW + 2 * HM;               // pixels per halo row
    constexpr int SHP = (TH + 1) * (TW + 1);                     // S2: pixels of one parity sub-image
    constexpr int HROWS = S2 ? 4 * SHP : (TH + 2 * HM) * HW2;
    constexpr int RS = 80;                                       // bytes per halo pixel: 64 data + 16 pad
    constexpr int PIECES = (HROWS * 5 + 63) / 64;                // 1-KiB LDS-DMA pieces per halo
    constexpr int NBUF = S2 ? 2 : 3;                             // halo buffers (S2: a halo is 2.5x the bytes -- two fit)
    constexpr int BUF = PIECES * 1024, STG = NBUF * BUF, SPITCH = 272, SCR = STG + PIX * SPITCH;   // SCR: 1 KiB of statistics partials
    constexpr int TQR = SCR + 1024;                              // 8-entry ring of tile ids (dynamic tile queue)
    constexpr int TWD = TAPW ? TAPW : NTAP == 9 ? 3 : NTAP == 4 ? 2 : 1;      // taps per window row
    constexpr int NI = VQK_MX_WL ? PIX / 32 : PIX / 64, NJ = VQK_MX_WL ? 1 : 2;                         // a matrix wave: NI x 32 pixels x 64 couts (PIX = 128: the half tile of the
                                                                 // 16x16 maps, twice as many blocks for a chip that their 256-pixel tiles leave half empty)
    constexpr int NQ = PIX / 16;                                 // 16-byte staging pieces per auxiliary thread and tile
    static_assert(PIX == 256 || ((PIX == 128 || PIX == 64) && !POOL && NTAP == 9), "256-pixel tiles, or plain 128- / 64-pixel part tiles");
    constexpr int NPH = NTAP * 2;                                // phases ((tap, k-substep) pairs) and weight fragments per unit
    static_assert(NTAP == 9 || ((NTAP == 4 || NTAP == 2 || NTAP == 1) && !POOL), "3x3 taps, the 2x2 / 1x2 / 2x1 taps of a phase, or a 1x1 conv");
    static_assert(!S2 || (NTAP == 9 && !POOL && PIX == 128), "stride 2: 3x3 taps, 128-pixel tiles");
    static_assert(TAPW == 0 || (NTAP == 2 && (TAPW == 1 || TAPW == 2)), "TAPW names the orientation of a 2-tap window");
    constexpr int XS = (PIECES + 3) / 4;                         // pieces per X wave
    constexpr int OOB = (int)0x80000000;
    typedef bf16x8_t frag_t;
    typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = g.w >> TWLOG, tiles_y = g.h / TH;
    // The four phases of an upsample conv in ONE launch (NTAP = 4, g.phase_mode): 1 (forward): the phase is a dimension of the
    // TILE index, next to the cout tile -- the blocks that share an input halo run side by side; the window offset, the destination
    // parity and the weight block follow the tile's phase.  2 (data gradient): the phase is a dimension of the UNIT index -- a
    // tile accumulates its four phases x chunks in registers (fp32) and is stored once (the four-launch form added the phases
    // through the bf16 tensor in memory: three more passes over it).
    const int pmode = NTAP == 4 ? g.phase_mode : 0;
    const int total_tiles = g.n * tiles_y * tiles_x * g.tiles_n * (pmode == 1 ? 4 : 1);
    const int nch = g.cpt >> 2;                                  // 32-channel chunks per tile (and phase)
    const int nun = pmode == 2 ? 4 * nch : nch;                  // units per tile
    // this block's XCD (block b runs on XCD b % 8) owns the virtual block ids [c0, c0 + cnt): xcd_remap, spelled out
    const int grid = (int)gridDim.x, q8 = grid >> 3, r8 = grid & 7, xcd = (int)blockIdx.x & 7;
    const int cnt = q8 + (xcd < r8 ? 1 : 0);
    const int c0 = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
    const int vbid = c0 + ((int)blockIdx.x >> 3);
    if (vbid >= total_tiles) return;
    // k-th tile of the XCD's sequence (static share: block i of the XCD takes k = i, i + cnt, i + 2 cnt, ...)
    auto tile_of = [&](int k) -> int { const int j = k / cnt; return c0 + (k - j * cnt) + j * grid; };
    const int tqm = g.tq ? g.tq_mode : 0;                        // 0 static, 1 first tile static, 2 every tile from the queue
    volatile VQK_LDS int* const ring = (volatile VQK_LDS int*)(smem + TQR);     // (explicit LDS pointer: a generic volatile one became flat_load + vmcnt(0))
    // id of this block's j-th tile.  (readfirstlane: the ring holds ONE value per slot -- as a scalar it keeps every address
    // derived from it wave-uniform; read as a vector value hipcc wrapped each weight load in a waterfall loop, +15 % kernel time)
    auto next_id = [&](int j) -> int { return tqm ? __builtin_amdgcn_readfirstlane(ring[j & 7]) : vbid + j * grid; };
    const bool pro_sync = tqm == 2 || (tqm && nun < 4);          // the ring's first entries are fetched before the first unit

    struct TilePos { int img, py0, px0, nt, ph; };
    auto tile_pos = [&](int t) -> TilePos {
        TilePos tp;
        tp.nt = t % g.tiles_n; t /= g.tiles_n;
        tp.ph = 0;
        if (pmode == 1) { tp.ph = t & 3; t >>= 2; }
        const int txi = t % tiles_x; t /= tiles_x;
        const int tyi = t % tiles_y;
        tp.img = t / tiles_y; tp.py0 = tyi * TH; tp.px0 = txi * TW;
        return tp;
    };
    auto unit_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    // VQK_MXABL & 64 (instrumented build, tools/mx_phase_probe.py): g.gn_ws is NOT a statistics workspace but a debug buffer -- every
    // matrix wave leaves {cycles in its unit loop, cycles parked in the unit barrier, cycles parking tiles, units} there (s_memtime)
    double* const gnws = (VQK_MXABL & 64) ? nullptr : g.gn_ws;
    unsigned long long* const dbg = (VQK_MXABL & 64) ? reinterpret_cast<unsigned long long*>(g.gn_ws) : nullptr;

    if (wave < 4) {
        // ================================================================= M waves: MFMA only
        if (VQK_MX_PRIO) __builtin_amdgcn_s_setprio(VQK_MX_PRIO);
        const int wm = VQK_MX_WL ? 0 : wave >> 1, wn = VQK_MX_WL ? wave : wave & 1;
        const int p = lane & 31, kg = lane >> 5;
        unsigned abase[NI], sbase[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int ty, tx;
            if (TWLOG == 5) { ty = wm * NI + i; tx = p; }
            else { ty = wm * 2 * NI + 2 * i + (p >> 4); tx = p & 15; }
            abase[i] = (unsigned)(((ty + (pmode ? 0 : g.tap_oy)) * HW2 + tx + (pmode ? 0 : g.tap_ox)) * RS + kg * 16);
            sbase[i] = (unsigned)(STG + (ty * TW + tx) * SPITCH + (wn * (32 * NJ) + 4 * kg) * 2);
        }
        const int lane16 = lane * 16;
        const int phase_bytes = (g.cout >> 5) * nch * (NPH * 1024);
        const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<bf16_raw*>(wp), 0, phase_bytes * (pmode ? 4 : 1), 0x00020000);
        // tile key: cout tile | phase << 16 (pmode 1); c: unit of the tile = chunk (+ nch * phase in pmode 2)
        auto unit_ph = [&](int key, int c) -> int { return pmode == 2 ? c / nch : (key >> 16); };
        auto unit_w = [&](int key, int c, int j) -> int {
            const int ph = unit_ph(key, c), cc = pmode == 2 ? c - ph * nch : c;
            const int wph = (NTAP == 4 && g.phase_rev) ? 3 - ph : ph;      // (conv_geom.h: phase_rev -- the pooled data gradient)
            return wph * phase_bytes + (((key & 0xffff) * 4 + wn * NJ + j) * nch + cc) * (NPH * 1024);
        };
        // window offset of a phase inside the halo: forward (a, b), data gradient (1 - a, 1 - b) (vqk_conv2d_ups_phase)
        auto phase_off = [&](int ph) -> int {
            const int a = ph >> 1, b = ph & 1;
            return pmode == 1 ? (a * HW2 + b) * RS : pmode == 2 ? ((1 - a) * HW2 + (1 - b)) * RS : 0;
        };
        auto tile_nt = [&](int t) -> int {
            return (t % g.tiles_n) | (pmode == 1 ? ((t / g.tiles_n) & 3) << 16 : 0);
        };
        auto wload = [&](int soff) -> frag_t {
            return __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(wsrd, lane16, soff, 0));
        };

        constexpr int RD = NTAP == 9 ? VQK_MX_RD : NTAP == 4 ? 4 : 2;
        static_assert(NPH % RD == 0, "weight ring depth must divide the phases of a unit");
        f32x16 acc[NI][NJ];
        frag_t bw[RD][NJ];
        int wcur[NJ];
        if (pro_sync) unit_barrier();                            // (joins the X waves' prologue: ring[0 ...] is published)
        const int t_first = tqm == 2 ? __builtin_amdgcn_readfirstlane(ring[0]) : vbid;
        if (t_first >= total_tiles) return;                      // queue mode 2, a late block: nothing left
        int cur_nt = tile_nt(t_first);
#pragma unroll
        for (int j = 0; j < NJ; ++j) wcur[j] = unit_w(cur_nt, 0, j);
#pragma unroll
        for (int d = 0; d < RD; ++d)
#pragma unroll
            for (int j = 0; j < NJ; ++j) bw[d][j] = wload(wcur[j] + d * 1024);
        unit_barrier();                                          // the halo of unit 0 has landed

        int tj = 0, c = 0, bi = 0, t_next = 0;
        unsigned long long t_bar = 0, t_park = 0, n_units = 0, t_c0 = 0, t_cl = 0, t_top = 0;     // (t_c0 / t_cl: MFMA sections of a tile's first / last unit)
        const unsigned long long t_begin = (VQK_MXABL & 64) ? __builtin_amdgcn_s_memtime() : 0;
        for (;;) {
            if (VQK_MXABL & 64) t_top = __builtin_amdgcn_s_memtime();
            // the id of the next tile is read one unit before it is needed (published two units earlier by the first X wave)
            if (c == nun - 2) t_next = next_id(tj + 1);
            const bool last_c = c == nun - 1, more = last_c && t_next < total_tiles;
            int nc = c + 1, nxt_nt = cur_nt;
            if (last_c) {
                if (more) { nc = 0; nxt_nt = tile_nt(t_next); }
                else nc = c;                                     // clamp: the weight prefetch stays unconditional
            }
            int wnxt[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) wnxt[j] = unit_w(nxt_nt, nc, j);
            const char* lbase[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) lbase[i] = smem + bi * BUF + abase[i] + (NTAP == 4 ? phase_off(unit_ph(cur_nt, c)) : 0);

            frag_t a[2][NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) a[0][i] = *reinterpret_cast<const frag_t*>(lbase[i]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < NTAP; ++tap) {
                const int toff = mx_tap_pix(S2, tap, TWD, HW2, SHP) * RS;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bool reads = ks == 0 || tap < NTAP - 1;
                    const int ph = tap * 2 + ks;
                    if (ks == 0) {
#pragma unroll
                        for (int i = 0; i < NI; ++i) a[1][i] = *reinterpret_cast<const frag_t*>(lbase[i] + toff + 32);
                    } else if (tap < NTAP - 1) {
                        const int toff1 = mx_tap_pix(S2, tap + 1, TWD, HW2, SHP) * RS;
#pragma unroll
                        for (int i = 0; i < NI; ++i) a[0][i] = *reinterpret_cast<const frag_t*>(lbase[i] + toff1);
                    }
                    if (tap == 0 && ks == 0 && c == 0) {
                        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < NI; ++i)
#pragma unroll
                            for (int j = 0; j < NJ; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[ph % RD][j], a[ks][i], zero, 0, 0, 0);
                    } else if (VQK_MX_JOUTER) {
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int i = 0; i < NI; ++i)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[ph % RD][j], a[ks][i], acc[i][j], 0, 0, 0);
                    } else {
#pragma unroll
                        for (int i = 0; i < NI; ++i)
#pragma unroll
                            for (int j = 0; j < NJ; ++j)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bw[ph % RD][j], a[ks][i], acc[i][j], 0, 0, 0);
                    }
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        const int soff = (ph + RD >= NPH) ? wnxt[j] + (ph + RD - NPH) * 1024 : wcur[j] + (ph + RD) * 1024;
                        // VQK_MXABL & 32 (timing only, round 4): the two pixel-half waves of a cout half load the SAME weight
                        // fragments -- here the second wave keeps its stale ring instead: the kernel's L2 -> VGPR weight stream is
                        // halved (L2 side) at NO cost for whatever would share it, i.e. an upper bound on what a shared stream can win
                        // (same instruction stream: the second wave re-reads ONE fragment -- an L1 hit -- instead of skipping the load;
                        // skipping it put a branch into the pinned phase and cost 7-10 %)
                        bw[ph % RD][j] = wload(((VQK_MXABL & 32) && g.n > 0 && wm == 1) ? (soff & 1023) : soff);
                    }
                    if (VQK_MX_PIN) {
                        if (reads) {
#pragma unroll
                            for (int i = 0; i < NI; ++i) { SGB(0x008, 1); SGB(0x100, 1); }
                            if constexpr (NI * NJ - NI > 0) SGB(0x008, NI * NJ - NI);
                        } else {
                            SGB(0x008, NI * NJ);
                        }
                        SGB(0x020, NJ);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const unsigned long long t_a = (VQK_MXABL & 64) ? __builtin_amdgcn_s_memtime() : 0;
            if (VQK_MXABL & 64) { if (c == 0) t_c0 += t_a - t_top; if (c == nun - 1) t_cl += t_a - t_top; }
            if (c == nun - 1 && !((VQK_MXABL & 4) && g.n > 0)) {
                // park the tile: lane (pixel p, half kg) holds couts j*32 + 8*rq + 4*kg + e of its wave's 64 -> 8-byte pieces
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const u32x2 o = {pack_bf16x2(acc[i][j][4 * rq], acc[i][j][4 * rq + 1]),
                                             pack_bf16x2(acc[i][j][4 * rq + 2], acc[i][j][4 * rq + 3])};
                            *reinterpret_cast<u32x2*>(smem + sbase[i] + (j * 32 + 8 * rq) * 2) = o;
                        }
            }
            if (VQK_MXABL & 64) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the park's ds_writes are issued, not retired: count them as park)
                const unsigned long long t_b = __builtin_amdgcn_s_memtime();
                unit_barrier();
                const unsigned long long t_c = __builtin_amdgcn_s_memtime();
                t_park += t_b - t_a; t_bar += t_c - t_b; ++n_units;
            } else
            unit_barrier();
            if (last_c) { if (!more) break; ++tj; }
            cur_nt = nxt_nt; c = nc;
            bi = bi == NBUF - 1 ? 0 : bi + 1;
#pragma unroll
            for (int j = 0; j < NJ; ++j) wcur[j] = wnxt[j];
        }
        if ((VQK_MXABL & 64) && dbg && lane == 0) {
            unsigned long long* o = dbg + ((int64_t)blockIdx.x * 4 + wave) * 8;
            o[0] = __builtin_amdgcn_s_memtime() - t_begin; o[1] = t_bar; o[2] = t_park; o[3] = n_units;
            o[4] = t_c0; o[5] = t_cl; o[6] = (unsigned long long)nun; o[7] = 0;
        }
        return;
    }

    // ===================================================================== X waves: HBM traffic only
    const int xt = tid - 256, xw = wave - 4;
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<bf16_raw*>(x), 0, (int)((int64_t)g.n * g.h_in * g.w_in * g.cin * 2), 0x00020000);
    const int out_bytes = (int)((int64_t)g.n * g.dst_h * g.dst_w * g.cout * 2);    // dst_h x dst_w output pixels (dst_s h x dst_s w, or odd: +1)
    const __amdgpu_buffer_rsrc_t rsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw*>(res), 0, res ? out_bytes : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ysrd = __builtin_amdgcn_make_buffer_rsrc(y, 0, out_bytes >> (POOL ? 2 : 0), 0x00020000);

    // halo pieces of this wave: piece xw + 4*sl, lane -> 16-byte slot s = piece*64 + lane = halo pixel s/5, chunk s%5
    // (chunk 4 = pad).  rel = byte offset against the tile origin (input resolution), flg = border classes of the pixel
    // (1 top row, 2 bottom row, 4 left column, 8 right column of the halo; 16 = pad / beyond the halo: never fetched)
    int rel[XS], flg[XS];
#pragma unroll
    for (int sl = 0; sl < XS; ++sl) {
        const int s = (xw + 4 * sl) * 64 + lane;
        const int hp = s / 5, cp = s - hp * 5;
        if constexpr (S2) {
            // sub-image (a, b) of the patch: input pixel (2 hy + a, 2 hx + b); the odd sub-images have no row TH / column TW
            const int sh = hp / SHP, q = hp - sh * SHP;
            const int hy = q / HW2, hx = q - hy * HW2, pa = sh >> 1, pb = sh & 1;
            rel[sl] = (((2 * hy + pa) * g.w_in + 2 * hx + pb) * g.cin + cp * 8) * 2;
            flg[sl] = (cp == 4 || hp >= HROWS || (pa && hy == TH) || (pb && hx == TW)) ? 16 : 0;
        } else {
            const int hy = hp / HW2, hx = hp - hy * HW2;
            rel[sl] = (((((hy - HM) * g.src_s) >> g.ups) * g.w_in + (((hx - HM) * g.src_s) >> g.ups)) * g.cin + cp * 8) * 2;
            flg[sl] = (HM && hy == 0 ? 1 : 0) | (HM && hy == TH + 1 ? 2 : 0) | (HM && hx == 0 ? 4 : 0) | (HM && hx == TW + 1 ? 8 : 0) |
                      ((cp == 4 || hp >= HROWS) ? 16 : 0);
        }
    }
    auto issue_halo = [&](const TilePos& tp, int cu, int bufi) {
        // pmode 2: unit cu = phase * nch + chunk reads the phase (a, b) of the full-resolution gradient
        const int uph = pmode == 2 ? cu / nch : 0, c = pmode == 2 ? cu - uph * nch : cu;
        const int sa = pmode == 2 ? (uph >> 1) : g.src_a, sb = pmode == 2 ? (uph & 1) : g.src_b;
        const int base = (((tp.img * g.h_in + ((tp.py0 * g.src_s + sa) >> g.ups)) * g.w_in + ((tp.px0 * g.src_s + sb) >> g.ups)) * g.cin + c * 32) * 2;
        const int tb = S2 ? 16 : ((tp.py0 == 0 ? 1 : 0) | (tp.py0 + TH == g.h ? 2 : 0) | (tp.px0 == 0 ? 4 : 0) | (tp.px0 + TW == g.w ? 8 : 0) | 16);
#pragma unroll
        for (int sl = 0; sl < XS; ++sl) {
            if (xw + 4 * sl < PIECES) {                          // wave-uniform
                const int off = (flg[sl] & tb) ? OOB : base + rel[sl];
                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                    xsrd, (VQK_LDS void*)(smem + bufi * BUF + (xw + 4 * sl) * 1024), 16,
                    (VQK_MXABL & 8) ? (off & 0x8003fff0) : off, 0, 0, VQK_MX_HALO_AUX);
            }
        }
    };
    auto unpack8 = [&](const u32x4& r, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(r[e] << 16); v[2 * e + 1] = __uint_as_float(r[e] & 0xffff0000u); }
    };
    auto pack8 = [&](const float (&v)[8]) -> u32x4 {
        const u32x4 o = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
        return o;
    };
    const int slot = xt & 15;                                    // 8 couts = one 16-byte piece of the 256-byte pixel row
    constexpr int NP = POOL ? 4 : NQ;                            // output pieces per thread and tile
    constexpr int NR = NQ;                                       // residual pieces per thread and tile
    struct OutPos { int pix0, ppix0, co, img, tile; };
    auto out_pos = [&](const TilePos& tp) -> OutPos {
        OutPos o;
        o.tile = (tp.py0 / TH) * tiles_x + (tp.px0 >> TWLOG) + (pmode == 1 ? tp.ph * tiles_y * tiles_x : 0);   // tile of its image (deterministic GroupNorm sums: one slot per tile)
        const int da = pmode == 1 ? (tp.ph >> 1) : g.dst_a, db = pmode == 1 ? (tp.ph & 1) : g.dst_b;
        o.pix0 = (tp.img * g.dst_h + tp.py0 * g.dst_s + da) * g.dst_w + tp.px0 * g.dst_s + db;
        o.ppix0 = (tp.img * (g.h >> 1) + (tp.py0 >> 1)) * (g.w >> 1) + (tp.px0 >> 1);
        o.co = tp.nt * 128 + slot * 8;
        o.img = tp.img;
        return o;
    };
    auto res_off = [&](const OutPos& o, int k) -> int {          // byte offset of residual piece k (un-pooled resolution)
        int ty, tx;
        if constexpr (!POOL) { const int pp = k * 16 + (xt >> 4); ty = pp >> TWLOG; tx = pp & (TW - 1); }
        else { const int opp = (k >> 2) * 16 + (xt >> 4); ty = 2 * (opp >> (TWLOG - 1)) + ((k >> 1) & 1); tx = 2 * (opp & (TW / 2 - 1)) + (k & 1); }
        return ((o.pix0 + (ty * g.dst_w + tx) * g.dst_s) * g.cout + o.co) * 2;
    };
    auto load_res = [&](const OutPos& o, u32x4 (&rv)[NR]) {
#pragma unroll
        for (int k = 0; k < NR; ++k)
            rv[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrd, res_off(o, k), 0, 0));
    };
    int stat_idx = -1;                                           // workspace slot of this lane's statistic (drained tile)
    auto flush_stats = [&]() {                                   // first X wave, one barrier after the drain that parked them
        if (xw == 0) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) v += *reinterpret_cast<const float*>(smem + SCR + (k * 64 + lane) * 4);
            if (g.gn_part_nblk) {
                // deterministic mode: this tile's own slot.  Groups wider than a lane's 8 channels (16 / 32 channels per group: 2 / 4
                // neighbouring lanes hold parts of one group) are folded in lane order first; the first lane of a group stores
                const int span = g.gn_cpg >> 3;
                if (span >= 2) v += __shfl_xor(v, 1, 64);
                if (span >= 4) v += __shfl_xor(v, 2, 64);
                if (stat_idx >= 0 && (span < 2 || (lane & (span - 1)) == 0)) gnws[stat_idx] = (double)v;
            } else if (stat_idx >= 0) {
                atomicAdd(gnws + stat_idx, (double)v);
            }
        }
    };
    // k0 .. k1: the output pieces of this call (round 6: a tile's drain is PACED over the next tile's units -- measured with the
    // instrumented build, tools/mx_phase_probe.py: the matrix waves' unit beside a whole-tile drain took 6993 cycles against 5222 for a
    // unit with quiet auxiliary waves, 128 -> 128 @256^2); `first` clears the statistics accumulators, `last` folds them
    float ga = 0.f, qa = 0.f, gb = 0.f, qb = 0.f;                // GroupNorm sums of the tile being drained: channels 0-3 / 4-7 of the thread's eight
    auto drain = [&](const OutPos& o, const u32x4 (&rv)[NR], int k0, int k1, bool first, bool last) {
        if (first) { ga = 0.f; qa = 0.f; gb = 0.f; qb = 0.f; }
        float bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (bias) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + o.co), b1 = *reinterpret_cast<const f32x4*>(bias + o.co + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bv[e] = b0[e]; bv[4 + e] = b1[e]; }
        }
        // GroupNorm statistics of the stored output (wave-uniform switch): channel sums over this thread's pieces
        const bool want_stats = gnws != nullptr;
        // (v_dot2c_f32_bf16 sums a packed pair -- against (1, 1) for the sum, against itself for the squares -- straight
        // from the stored bf16 words: 8 VALU per 16-byte piece instead of 24 for unpack / add / fma; every X-wave VALU
        // instruction competes with the matrix wave of its SIMD for the issue port)
        // (inline asm: with __builtin_amdgcn_fdot2_f32_bf16 on the elements of a 16-byte vector hipcc 7.2 folded all four
        // dwords of a piece onto the first register.  hipcc pads no hazards for asm: a DOT result needs 3 wait states before
        // a different VALU instruction reads it and 1 before the same opcode accumulates into it again -- the eight dot
        // products of a piece go out as ONE statement with the four accumulators interleaved, and `settle` closes the
        // chain before the sums are read; without it the last products of a tile were dropped now and then)
        const unsigned ones2 = 0x3f803f80u;
        auto tally = [&](const u32x4& o) {
            const unsigned d0 = o[0], d1 = o[1], d2 = o[2], d3 = o[3];
            asm("v_dot2c_f32_bf16 %0, %4, %8\n\tv_dot2c_f32_bf16 %1, %4, %4\n\tv_dot2c_f32_bf16 %2, %6, %8\n\t"
                "v_dot2c_f32_bf16 %3, %6, %6\n\tv_dot2c_f32_bf16 %0, %5, %8\n\tv_dot2c_f32_bf16 %1, %5, %5\n\t"
                "v_dot2c_f32_bf16 %2, %7, %8\n\tv_dot2c_f32_bf16 %3, %7, %7\n\ts_nop 0"
                : "+v"(ga), "+v"(qa), "+v"(gb), "+v"(qb) : "v"(d0), "v"(d1), "v"(d2), "v"(d3), "v"(ones2));
        };
        auto settle = [&]() { asm volatile("s_nop 3" : "+v"(ga), "+v"(qa), "+v"(gb), "+v"(qb)); };
        u32x4 t[NQ];                                             // the thread's staging pieces, all requested up front
        if constexpr (!POOL) {
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                if (k < k0 || k >= k1) continue;                 // (wave-uniform)
                t[k] = *reinterpret_cast<const u32x4*>(smem + STG + (k * 16 + (xt >> 4)) * SPITCH + slot * 16);
            }
        } else {
#pragma unroll
            for (int k = 0; k < NQ; ++k) {
                if ((k >> 2) < k0 || (k >> 2) >= k1) continue;
                const int opp = (k >> 2) * 16 + (xt >> 4);
                const int ty = 2 * (opp >> (TWLOG - 1)) + ((k >> 1) & 1), tx = 2 * (opp & (TW / 2 - 1)) + (k & 1);
                t[k] = *reinterpret_cast<const u32x4*>(smem + STG + (ty * TW + tx) * SPITCH + slot * 16);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!POOL) {
            if (!bias && !res && g.act == 0 && g.acc_scale == 1.0f && g.out_gain == 1.0f) {     // plain copy (wave-uniform)
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    if (k < k0 || k >= k1) continue;
                    __builtin_amdgcn_raw_buffer_store_b128(t[k], ysrd, res_off(o, k), 0, VQK_MX_NT);
                    if (want_stats) tally(t[k]);
                }
            } else {
                const bool general = g.act != 0 || g.acc_scale != 1.0f || g.out_gain != 1.0f;     // wave-uniform
                float abl_b[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, abl_g[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    if (k < k0 || k >= k1) continue;
                    float f[8], r[8];
                    unpack8(t[k], f);
                    unpack8(rv[k], r);
                    if ((VQK_MXABL & 16) && g.n > 0) {
                        // TIMING-ONLY experiment (profiles/round3_gn_bwd_fusion_ab.txt): the arithmetic a fused GroupNorm+SiLU
                        // BACKWARD reduction would add to a data-gradient drain -- the residual operand stands in for the
                        // GroupNorm input x (same bytes), per element: x_hat, u = gamma x_hat + beta, sigmoid, SiLU', t = dy SiLU',
                        // two per-channel sums (d beta, d gamma).  Constants instead of per-group statistics; sums folded into
                        // the statistics accumulators so that nothing is dead code.
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xh = __fmaf_rn(r[e], 1.3f, -0.13f);
                            const float u = __fmaf_rn(xh, 1.1f, 0.05f);
                            const float sg = __frcp_rn(1.0f + __expf(-u));
                            const float t = f[e] * (sg * (1.0f + u * (1.0f - sg)));
                            abl_b[e] += t;
                            abl_g[e] = __fmaf_rn(t, xh, abl_g[e]);
                        }
                    }
                    if (general) {                               // y = out_gain * act(acc * acc_scale + bias) + residual
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            float v = __fmaf_rn(f[e], g.acc_scale, bv[e]);
                            v = g.act == 2 ? fmaxf(v, 0.0f) : g.act == 3 ? fmaxf(v, 0.2f * v) : v;
                            f[e] = __fmaf_rn(v, g.out_gain, r[e]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) f[e] = f[e] + bv[e] + r[e];
                    }
                    const u32x4 ov = pack8(f);
                    __builtin_amdgcn_raw_buffer_store_b128(ov, ysrd, res_off(o, k), 0, VQK_MX_NT);
                    if (want_stats) tally(ov);
                }
                if ((VQK_MXABL & 16) && g.n > 0) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { ga += abl_b[e]; qa += abl_g[e]; }
                }
            }
        } else {
            // y = pool_scale * sum over the 2x2 window of (conv + bias + residual), written at half resolution
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                if (k < k0 || k >= k1) continue;
                const int opp = k * 16 + (xt >> 4);              // pooled pixel 0..63 of the (TH/2) x (TW/2) output patch
                const int oty = opp >> (TWLOG - 1), otx = opp & (TW / 2 - 1);
                float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    float f[8], r[8];
                    unpack8(t[4 * k + d], f);
                    unpack8(rv[4 * k + d], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) s[e] += f[e] + r[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] = (s[e] + 4.0f * bv[e]) * g.pool_scale;
                const u32x4 ov = pack8(s);
                __builtin_amdgcn_raw_buffer_store_b128(ov, ysrd, ((o.ppix0 + oty * (g.w >> 1) + otx) * g.cout + o.co) * 2, 0, VQK_MX_NT);
                if (want_stats) tally(ov);
            }
        }
        if (want_stats && last) {
            // thread: 8 channels = two 4-channel halves (one group each when cpg == 4, the same group otherwise); the four
            // lanes l, l+16, l+32, l+48 of a wave own the same channels -> sum over them, then copy q = lane >> 4 of a slot
            // keeps ONE of the (up to) four values.  The four X waves' 64 values are parked in LDS and combined by the first
            // X wave after the next barrier (flush_stats): ONE wave-wide fp64 atomic instruction per tile -- with one per
            // wave the 256 CUs, which all work on the same image at a time, queued up on that image's four cache lines.
            settle();
            const bool split = g.gn_cpg == 4;
            if (!split) { ga += gb; qa += qb; }
            ga += __shfl_xor(ga, 16, 64); ga += __shfl_xor(ga, 32, 64);
            qa += __shfl_xor(qa, 16, 64); qa += __shfl_xor(qa, 32, 64);
            if (split) {
                gb += __shfl_xor(gb, 16, 64); gb += __shfl_xor(gb, 32, 64);
                qb += __shfl_xor(qb, 16, 64); qb += __shfl_xor(qb, 32, 64);
            }
            const int q = (xt >> 4) & 3;
            const float val = q == 0 ? ga : q == 1 ? qa : q == 2 ? gb : qb;
            *reinterpret_cast<float*>(smem + SCR + (xw * 64 + lane) * 4) = val;
            const int grp = (o.co + (q >> 1) * 4) / g.gn_cpg;
            const int gslot = o.img * (g.cout / g.gn_cpg) + grp;
            stat_idx = (split || q < 2) ? ((g.gn_part_nblk ? gslot * g.gn_part_nblk + g.gn_part_base + o.tile : gslot) * 2 + (q & 1)) : -1;
        }
    };

    // ---- dynamic tile queue: the first X wave fetches, lane 0 holds the atomic's result until the next interval
    int nissued = tqm == 2 ? 0 : 1, exhausted = 0, pend = 0, fv = 0;     // tiles fetched or static so far (wave-uniform)
    const int qbase = tqm == 2 ? 0 : cnt;                        // mode 1: the XCD's first cnt tiles are the blocks' static ones
    auto fetch_issue = [&]() {
        if (lane == 0) fv = __hip_atomic_fetch_add(g.tq + xcd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        pend = 1;
    };
    auto fetch_land = [&]() {
        const int t = tile_of(__builtin_amdgcn_readfirstlane(fv) + qbase);
        if (lane == 0) ring[nissued & 7] = t;
        ++nissued; pend = 0;
        if (t >= total_tiles) exhausted = 1;                     // the ONE fetch beyond the end: never fetch again
    };
    auto census = [&]() {                                        // the last block to finish zeroes the queue words again
        if (tqm && xw == 0 && lane == 0) {
            const int old = __hip_atomic_fetch_add(g.tq + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == grid - 1) {
#pragma unroll
                for (int i = 0; i < 9; ++i) __hip_atomic_exchange(g.tq + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    };
    if (tqm && xw == 0) {
        // what the first intervals need before any fetch of the loop can land: tile 0 (mode 2) and, for tiles of fewer than
        // four units, the tile of unit 3
        const int need = 3 / nun;
        while (nissued <= need && !exhausted) { fetch_issue(); fetch_land(); }
    }
    if (pro_sync) unit_barrier();
    const int t_first = tqm == 2 ? __builtin_amdgcn_readfirstlane(ring[0]) : vbid;
    if (t_first >= total_tiles) { census(); return; }            // queue mode 2, a late block: nothing left

    // position of the unit whose halo is requested next (NBUF - 1 units ahead of the M waves).  A new tile's id is looked up
    // right before its first halo is requested (lnew), one barrier after the interval that published it
    int ltj = 0, lc = 0, lbuf = 0, lvalid = 1, lnew = 0;
    TilePos ltp = tile_pos(t_first);
    auto look = [&]() {
        if (lnew) {
            lnew = 0;
            if (lvalid) {
                const int t = next_id(ltj);
                lvalid = t < total_tiles;
                if (lvalid) ltp = tile_pos(t);
            }
        }
    };
    auto bump = [&]() {
        if (++lc == nun) { lc = 0; ++ltj; lnew = 1; }
        lbuf = lbuf == NBUF - 1 ? 0 : lbuf + 1;
    };
    issue_halo(ltp, lc, lbuf); bump();                           // unit 0
    if constexpr (NBUF == 3) { look(); if (lvalid) issue_halo(ltp, lc, lbuf); bump(); }     // unit 1
    int tj = 0, c = 0;
    TilePos cur = tile_pos(t_first);
    OutPos done = out_pos(cur);
    int fu = 4 % nun, ford = 4 / nun;                            // (u + 4) % nun, (u + 4) / nun: the tile of unit u + 4
    bool pending = false, flush = false;
    int dk = 0;                                                  // next piece of the tile being drained
    u32x4 rv[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) rv[k] = u32x4{0u, 0u, 0u, 0u};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unit_barrier();

    for (;;) {
        // every vector-memory operation of the previous interval has completed: the halo of unit u+1 is in LDS (published
        // to the M waves by the barrier that ends this interval), the residual pieces are in registers
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tqm && xw == 0) {
            // the fetch issued in the last interval has landed: into the ring (visible to the other waves after this interval's
            // barrier); the tile of unit u + 4 is requested now -- two intervals before its first halo is
            if (pend) fetch_land();
            if (!exhausted && ford >= nissued) fetch_issue();
        }
        if (++fu == nun) { fu = 0; ++ford; }
        if constexpr (NBUF == 2) {
            // two halo buffers: the buffer of unit u+1 is the one unit u-1 was computed from -- free since the last barrier.
            // Requested first, it has this whole interval to land and is awaited before the barrier that publishes it
            look();
            if (lvalid) issue_halo(ltp, lc, lbuf);
            bump();
        }
        if (flush) { flush_stats(); flush = false; }
        if (pending && !((VQK_MXABL & 2) && g.n > 0)) {          // a tile parked during unit u-1 (or earlier: paced) waits in the staging tile
            // paced: ceil(NP / (nun - 1)) pieces per interval over the next tile's units 0 .. nun-2 -- it has to be gone before that
            // tile is parked at the end of its unit nun-1 (the residual registers are reloaded in that interval, too)
            const int per = (VQK_MX_PACE && nun > 2) ? (NP + nun - 2) / (nun - 1) : NP;
            const int k1 = dk + per < NP ? dk + per : NP;
            drain(done, rv, dk, k1, dk == 0, k1 == NP);
            dk = k1;
            if (dk == NP) { pending = false; dk = 0; flush = gnws != nullptr; }
        }
        const bool last_c = c == nun - 1;
        int t_next = 0;
        if (last_c) {                                            // unit u ends a tile: its residual is requested now,
            pending = true;                                      // the tile itself is drained from the next interval on
            done = out_pos(cur);
            if (res) load_res(done, rv);
            t_next = next_id(tj + 1);
        }
        if constexpr (NBUF == 3) {
            look();
            if (lvalid && !((VQK_MXABL & 1) && g.n > 0)) issue_halo(ltp, lc, lbuf);      // unit u+2
            bump();
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        unit_barrier();
        if (last_c) {
            if (t_next >= total_tiles) break;
            c = 0; ++tj; cur = tile_pos(t_next);
        } else {
            ++c;
        }
    }
    if (flush) flush_stats();
    if (pending) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        drain(done, rv, dk, NP, dk == 0, true);
        if (gnws) {
            unit_barrier();                                      // the M waves have ended: this joins the X waves only
            flush_stats();
        }
    }
    census();
}

}  // namespace

namespace vqkd {

namespace {
// one instantiation per kernel symbol: the dynamic-LDS opt-in is set (and checked) for EVERY variant, once
template <auto Kern>
int mx_launch(dim3 grid, int lds, hipStream_t st, const void* x, const void* w, const float* bias, const void* res, void* y,
              const ConvGeom& g) {
    static const hipError_t attr = hipFuncSetAttribute((const void*)Kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (attr != hipSuccess) return VQK_ERR_LAUNCH;
    hipLaunchKernelGGL(Kern, grid, dim3(512), lds, st, (const bf16_raw*)x, (const bf16_raw*)w, bias, (const bf16_raw*)res,
                       (bf16_raw*)y, g);
    return hipGetLastError() == hipSuccess ? VQK_OK : VQK_ERR_LAUNCH;
}

int device_cus() {                                               // persistent grid: one 512-thread block per CU
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return cus;
}
}  // namespace

int launch_conv3x3_mx(const void* x, const void* w, const float* bias, const void* res, void* y, const void* zeros,
                      const ConvGeom& g_in, int twlog, hipStream_t st) {
    (void)zeros;
    ConvGeom g = g_in;
    if (!g.dst_h) { g.dst_h = g.h * g.dst_s; g.dst_w = g.w * g.dst_s; }
    // 128-pixel half tiles for the 16x16 maps: a function of the image size only (never of the batch size: per-tile
    // statistics partials group differently), they double the blocks of launches whose 256-pixel tiles fill half the chip
    const int half_on = VQK_TUNE("MX_HALF", 1);
    const int half_hw = VQK_TUNE("MX_HALF_HW", 256);
    const bool half = half_on && g.h * g.w <= half_hw && !g.pool && g.ntap == 9 && (twlog == 4 ? (g.h % 8) == 0 : (g.h % 4) == 0);
    // 64-pixel quarter tiles (round 4): the 16x16 maps with <= 256 output channels (256 -> 256, 512 -> 256 @16^2: 128 half tiles for
    // 256 CUs at bs 32, 412-513 TF) -- again a function of the layer shape only
    const int qmode = VQK_TUNE("MX_QUARTER", 1);   // 1: layers with <= 2 output-channel tiles, 2: every half-tile layer
    const bool quarter = half && qmode && (g.tiles_n <= 2 || qmode >= 2) && (twlog == 4 ? (g.h % 4) == 0 : (g.h % 2) == 0);
    const int th = (quarter ? 64 : half ? 128 : 256) >> twlog;
    const int total = g.n * (g.h / th) * (g.w >> twlog) * g.tiles_n * ((g.ntap == 4 && g.phase_mode == 1) ? 4 : 1);
    // COMM_CUS (data parallel, world > 1): CUs left to the collective's kernels.  A persistent grid of one block per CU that
    // finds some CUs taken runs its last blocks as a SECOND wave (up to 2x the kernel time); a grid of cus - COMM_CUS blocks
    // runs as one wave on what is free (tools/comm_probe.py, profiles/round4_comm_probe.txt)
    int cus = device_cus() - VQK_TUNE("COMM_CUS", 0);
    if (cus < 1) cus = 1;
    const dim3 grid((unsigned)(total < cus ? total : cus));
    // dynamic tile queue (tuning slot TILE_QUEUE: 0 static share, 1 first tile static, 2 every tile from the queue) on the
    // queue words of the calling thread's current stream (vqk_set_tile_queue); a grid of one tile per block has nothing to draw
    const int tqm = VQK_TUNE("TILE_QUEUE", 0);
    const DetState& tqs = tile_queue_state();
    auto set_queue = [&](int tiles, int blocks) {
        const bool on = tqm > 0 && tqs.ws != nullptr && tqs.bytes >= 64 && tiles > blocks && (g.cpt >> 2) >= 2;
        g.tq = on ? reinterpret_cast<int*>(tqs.ws) : nullptr;
        g.tq_mode = on ? (tqm >= 2 ? 2 : 1) : 0;
    };
    set_queue(total, (int)grid.x);
    constexpr int TQB = 64;                                      // the kernel's tile ring
    constexpr int lds5 = 3 * (((256 / 32 + 2) * 34 * 5 + 63) / 64) * 1024 + 256 * 272 + 1024 + TQB;
    constexpr int lds4 = 3 * (((256 / 16 + 2) * 18 * 5 + 63) / 64) * 1024 + 256 * 272 + 1024 + TQB;
    constexpr int lds4h = 3 * (((128 / 16 + 2) * 18 * 5 + 63) / 64) * 1024 + 128 * 272 + 1024 + TQB;
    constexpr int lds5h = 3 * (((128 / 32 + 2) * 34 * 5 + 63) / 64) * 1024 + 128 * 272 + 1024 + TQB;
#define MXL(K, L) return mx_launch<K>(grid, L, st, x, w, bias, res, y, g)
    constexpr int lds4q = 3 * (((64 / 16 + 2) * 18 * 5 + 63) / 64) * 1024 + 64 * 272 + 1024 + TQB;
    constexpr int lds5q = 3 * (((64 / 32 + 2) * 34 * 5 + 63) / 64) * 1024 + 64 * 272 + 1024 + TQB;
    if (g.s2) {
        // stride-2 3x3 conv: 128-pixel tiles, TWO halo buffers of four (TH+1) x (TW+1) parity sub-images
        constexpr int lds5s = 2 * ((4 * (128 / 32 + 1) * 33 * 5 + 63) / 64) * 1024 + 128 * 272 + 1024 + TQB;
        constexpr int lds4s = 2 * ((4 * (128 / 16 + 1) * 17 * 5 + 63) / 64) * 1024 + 128 * 272 + 1024 + TQB;
        if (g.pool || g.ntap != 9 || (twlog == 4 ? (g.h % 8) : (g.h % 4)) != 0) return VQK_ERR_ARG;
        const int tot2 = g.n * (g.h / (128 >> twlog)) * (g.w >> twlog) * g.tiles_n;
        const dim3 grid2((unsigned)(tot2 < cus ? tot2 : cus));
        set_queue(tot2, (int)grid2.x);
        if (twlog == 5) return mx_launch<conv3x3_mx_kernel<5, false, 9, 128, 0, true>>(grid2, lds5s, st, x, w, bias, res, y, g);
        return mx_launch<conv3x3_mx_kernel<4, false, 9, 128, 0, true>>(grid2, lds4s, st, x, w, bias, res, y, g);
    }
    if (g.ntap == 2) {
        if (g.pool) return VQK_ERR_ARG;
        if (g.tapw == 2) { if (twlog == 5) MXL((conv3x3_mx_kernel<5, false, 2, 256, 2>), lds5); else MXL((conv3x3_mx_kernel<4, false, 2, 256, 2>), lds4); }
        if (g.tapw == 1) { if (twlog == 5) MXL((conv3x3_mx_kernel<5, false, 2, 256, 1>), lds5); else MXL((conv3x3_mx_kernel<4, false, 2, 256, 1>), lds4); }
        return VQK_ERR_ARG;
    }
    if (quarter) {
        if (twlog == 5) MXL((conv3x3_mx_kernel<5, false, 9, 64>), lds5q); else MXL((conv3x3_mx_kernel<4, false, 9, 64>), lds4q);
    } else if (half) {
        if (twlog == 5) MXL((conv3x3_mx_kernel<5, false, 9, 128>), lds5h); else MXL((conv3x3_mx_kernel<4, false, 9, 128>), lds4h);
    } else if (g.ntap == 4) {
        if (g.pool) return VQK_ERR_ARG;
        if (twlog == 5) MXL((conv3x3_mx_kernel<5, false, 4>), lds5); else MXL((conv3x3_mx_kernel<4, false, 4>), lds4);
    } else if (g.ntap == 1) {                                    // 1x1 conv: no halo, 2 phases per unit
        if (g.pool) return VQK_ERR_ARG;
        constexpr int lds1 = 3 * ((256 * 5 + 63) / 64) * 1024 + 256 * 272 + 1024 + TQB;
        if (twlog == 5) MXL((conv3x3_mx_kernel<5, false, 1>), lds1); else MXL((conv3x3_mx_kernel<4, false, 1>), lds1);
    } else if (twlog == 5) {
        if (g.pool) MXL((conv3x3_mx_kernel<5, true, 9>), lds5); else MXL((conv3x3_mx_kernel<5, false, 9>), lds5);
    } else {
        if (g.pool) MXL((conv3x3_mx_kernel<4, true, 9>), lds4); else MXL((conv3x3_mx_kernel<4, false, 9>), lds4);
    }
#undef MXL
}

}  // namespace vqkd
