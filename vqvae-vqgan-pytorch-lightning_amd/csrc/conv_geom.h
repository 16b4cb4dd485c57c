// Geometry descriptor and small device helpers shared by the conv translation units (conv.hip, conv_mx.hip).
#pragma once
#include "common.h"

namespace vqkd {

struct ConvGeom {
    int n, h_in, w_in, h, w, cin, cout, ks, ups;
    // general (im2col kernels only): output pixel (oh, ow) reads virtual-input pixel (oh*stride + kh - pad, ...);
    // the virtual input is x itself (ups = 0), its nearest x2 upsample (ups = 1) or x zero-stuffed x2 (ups = 1,
    // zs = 1: only even coordinates carry data -- the dgrad of a stride-2 conv); vh/vw = its extent.
    int stride, pad, zs, vh, vw;
    float acc_scale, out_gain;      // epilogue: y = out_gain * act(acc * acc_scale + bias) + residual
    int pool;                       // stream kernel: y = pool_scale * (2x2 sum of the above), written at half resolution
    float pool_scale;
    int m;          // n*h*w output pixels
    // im2col kernel, zero-stuffed input (dgrad of a stride-2 conv): ONE output-parity class per launch.  Output pixels
    // (2a + sub_py, 2b + sub_px), a < sub_h, b < sub_w, and only the taps that land on real (even) input positions:
    // kh in khl[0..nkh), kw in kwl[0..nkw) -- a quarter of the MFMA work of multiplying the stuffed zeros.
    int sub, sub_py, sub_px, sub_h, sub_w, nkh, nkw, khl[2], kwl[2];
    int wrow_chunks;   // 16-byte chunks per weight row (= ks*ks*cpt; kchunks counts only the taps of the class)
    int cpt;        // 16-byte chunks per tap  (cin / elems-per-16B)
    int kchunks;    // ks*ks*cpt
    int tiles_m, tiles_n;
    // matrix/auxiliary-wave kernel only: GroupNorm statistics of the OUTPUT fused into the drain -- the per-(sample, group)
    // sums of y and y*y (y as stored, i.e. rounded to bf16) are added to gn_ws[n][group][2] (vqk_gn_forward's workspace)
    double* gn_ws;
    int gn_cpg;     // channels per group (cout / groups): a multiple of 4
    // deterministic mode: gn_part_nblk > 0 -- the tile's sums are STORED at gn_ws[((n * groups + g) * gn_part_nblk + gn_part_base +
    // tile_in_image) * 2 + j] (one slot per tile: no atomics, no zero-on-entry protocol); the consumer adds the slots in order
    int gn_part_nblk, gn_part_base;
    // matrix/auxiliary-wave kernel, nearest-x2 upsample convs in PHASE form (conv_mx.hip): output pixel (2i+a, 2j+b) of a 3x3
    // conv over the upsampled image sees only a 2x2 window of the low-resolution input, with pre-summed weights -- four
    // 2x2-tap launches (16 tap-GEMMs per low-resolution pixel instead of 36).  ntap = 4: the taps (r, s), r, s in {0, 1},
    // sit at halo rows tap_oy + r / columns tap_ox + s of the ordinary (TH+2)x(TW+2) halo; the output is written at
    // pixel (dst_s*i + dst_a, dst_s*j + dst_b) of a (dst_s*h) x (dst_s*w) tensor (fprop: dst_s = 2); the input is read at
    // pixel (src_s*i + src_a, src_s*j + src_b) of an h_in x w_in tensor (the data gradient reads the phase of dy: src_s = 2).
    int ntap, tap_oy, tap_ox, src_s, src_a, src_b, dst_s, dst_a, dst_b;
    // ntap = 2: tapw names the window (2: 1x2, 1: 2x1).  dst_h x dst_w: extent of the destination tensor (0: dst_s*h x dst_s*w;
    // the data gradient of a stride-2 conv without padding writes the phases of a (2h+1) x (2w+1) tensor).
    // s2 = 1: the stride-2 3x3 conv itself (input (2h+1) x (2w+1), output h x w): conv3x3_mx_kernel<..., S2 = true>.
    int tapw, dst_h, dst_w, s2;
    int phase_mode; // ntap = 4, the four phases of an upsample conv in one launch: 1 forward (phase = a tile dimension), 2 data
                    // gradient (phase = a unit dimension: the tile accumulates all four in registers); weights: the four blocks of layout 2
    int phase_rev;  // phase_mode 1 only: the weight block of output phase ph is block 3 - ph of the operand.  With the data-gradient
                    // operand of a conv (layout 2, transpose: the pre-summed phase weights of W^T, taps mirrored) this makes the
                    // forward-type phase launch compute  nearest-x2(dy) * flip(W)^T  = the data gradient of a conv FOLLOWED BY a
                    // 2x2 average pool from the pooled gradient (vqk_conv2d_pooled_dgrad_phase): phase (a, b) of the mirrored
                    // kernel is phase (1-a, 1-b) of the unmirrored one with its 2x2 window mirrored
    int dy_pool;    // weight-gradient mx kernel: dy is given at HALF resolution (the gradient of a fused 2x2 average pool: every
                    // pooled pixel stands for its 2x2 block), dW is scaled by acc_scale
    int act;        // matrix/auxiliary-wave kernel: epilogue activation (0 none, 2 relu, 3 leaky relu 0.2), with acc_scale / out_gain
    // matrix/auxiliary-wave kernel: DYNAMIC TILE QUEUE (vqk_set_tile_queue + tuning slot TILE_QUEUE).  tq = nullptr: the static
    // share (tile j of block b = b + j * grid).  Otherwise tq[0..7] are the per-XCD counters and tq[8] the census of finished
    // blocks (all zero on entry; the last block to finish leaves them zero again): the blocks of an XCD draw the tiles of the
    // XCD's static sequence in order -- same tiles, same order, same L2 neighbourhoods, but a block that starts late (its CU was
    // held by a collective's kernel or by a kernel of another stream) finds less work instead of a fixed share.
    // tq_mode 1: a block's first tile is its static one (no start-up latency), 2: every tile comes from the queue.
    int* tq;
    int tq_mode;
    // weight-gradient mx kernel, split-product mode (conv_x3.hip): x and dy are (hi | lo) bf16 PAIR tensors, cin / cout are their
    // pixel pitches (TWICE the true channel counts).  The grid carries three classes of 64 x 64 tiles -- dy_hi^T x_hi, dy_hi^T x_lo,
    // dy_lo^T x_hi -- that the final atomic pass adds onto the same dW[cout / 2][9][cin / 2] tile.
    int fold;
};

__device__ __forceinline__ int xcd_remap(int bid, int total) {
    // contiguous chunk of tiles per XCD (block b runs on XCD b % 8); bijective for any total
    const int q = total >> 3, r = total & 7, xcd = bid & 7, k = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// two fp32 -> packed bf16 pair (v_cvt_pk_bf16_f32: hardware round-to-nearest-even)
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    const f32x2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}

// matrix-wave / auxiliary-wave 3x3 kernel (conv_mx.hip): bf16, whole 128-cout tiles, 256-pixel patches of width 1 << twlog
int launch_conv3x3_mx(const void* x, const void* w, const float* bias, const void* res, void* y, const void* zeros,
                      const ConvGeom& g, int twlog, hipStream_t st);

// matrix-wave / auxiliary-wave 3x3 weight-gradient kernel (conv_wgmx.hip): bf16, Cin % 64 == 0, Cout % 64 == 0, W % 16 == 0;
// grid = tiles x splits blocks of 512 threads, `pps` 8x16-pixel patches per split
// `part` (deterministic mode): workspace of splits * tiles * 64*9*64 floats -- the splits' partial tiles are stored there and
// summed in split order by a second launch instead of being accumulated with fp32 atomics
// conv.hip <-> conv_wgrad.hip: the geometry builder and the per-host-thread launch state (test hook vqk_conv_set_variant, the
// weight-gradient grid cap of vqk_conv_set_block_caps)
int conv_make_geom(ConvGeom& g, int dtype, int n, int h_in, int w_in, int cin, int cout, int ksize, int ups);
int& conv_force_variant();
int& conv_wgrad_blocks();

// split-product 3x3 conv (conv_x3.hip): fp32 in / out, three bf16 products per multiply-add; weights in layout 5
int launch_conv3x3_x3(const void* x, const void* w, const float* bias, const void* res, void* y, const void* zeros,
                      const ConvGeom& g, int act, int blocks_cap, hipStream_t st);
// ... and its weight gradient straight from the fp32 tensors (no pair tensors): fp32 atomics into dw
int launch_conv3x3_wgrad_x3(const void* x, const void* dy, float* dw, const ConvGeom& g, int blocks_cap, hipStream_t st);

// exact-fp32 kernels of the two edge convs (conv_thin_f32.hip): one side of the GEMM is the 4-channel (padded 3-channel) tensor
int launch_conv3x3_thin_out_f32(const float* x, const float* w, const float* bias, const float* res, float* y, int n, int h, int wd,
                                int cin, int act, float acc_scale, float out_gain, hipStream_t st);
int launch_conv3x3_thin_in_f32(const float* x, const float* w, const float* bias, float* y, int n, int h, int wd, int cout, hipStream_t st);
int launch_conv3x3_wgrad_thin_f32(int mode, const float* wide, const float* thin, float* dw, int n, int h, int w, int cw, float scale,
                                  hipStream_t st);

int launch_conv3x3_wgrad_mx(const void* x, const void* dy, float* dw, const void* zeros, const ConvGeom& g, int tiles,
                            int splits, int pps, hipStream_t st, float* part = nullptr);

}  // namespace vqkd
