/* vqk -- C-ABI of the MI355X (gfx950) VQ-VAE / VQ-GAN train-step kernels.
 *
 * Boundary rules:
 *   - plain C: raw device pointers + sizes + scalars, no torch types;
 *   - the CALLER owns every buffer (outputs, workspaces); nothing is allocated, nothing is
 *     synchronised, and a call's RESULT depends on its arguments only.  The library reads no environment variable.
 *     What state it does keep, all of it launch POLICY (which kernel / grid / accumulation order serves a call), and its scope:
 *       THREAD-LOCAL (applies to the launches the calling host thread issues afterwards; another thread -- e.g. autograd's
 *       backward worker -- arms its own): vqk_set_deterministic (+ its workspace), vqk_set_scratch, vqk_set_tile_queue, vqk_conv_set_block_caps,
 *       vqk_conv_set_variant.  These act at LAUNCH (= graph capture) time: a hipGraph captured under them replays the
 *       captured kernels and workspace pointers from any thread, whatever that thread's own settings are
 *       (tests/test_gpu_deterministic.py::test_graph_captured_on_one_thread_replays_identically_from_another);
 *       PROCESS-WIDE: the tuning slots of vqk_set_tuning (grid sizes, kernel-choice thresholds: never the arithmetic of a
 *       kernel, except where a slot's description says it selects a differently rounding kernel);
 *   - every launch goes to the `stream` argument (a hipStream_t passed as void*; NULL = the
 *     null stream);
 *   - return value: 0 = ok, negative = vqk_status (see vqk_status_str).  There is NO fallback
 *     path: an unsupported shape/dtype is an error, never a silent detour.
 *
 * Layouts: activations are NHWC (pixel-major, channels contiguous); conv weights are
 * [Cout][kh][kw][Cin] ("KRSC" = the physical layout of a channels_last OIHW torch tensor).
 * dtype codes: VQK_F32 = 0 (parity mode), VQK_BF16 = 1 (throughput mode, fp32 accumulate).
 *
 * What each entry point replaces in the reference (paths relative to the reference checkout):
 *   vqk_bias_act / vqk_upfirdn2d : the reference's own native plugin ABI,
 *        vqvae/modules/loss/stylegan2_discriminator/utils/ops/bias_act.cpp:32-90 and
 *        .../upfirdn2d.cpp:16-94 (pybind functions `bias_act`, `upfirdn2d`);
 *   everything else: ATen/cuDNN calls made from
 *        vqvae/modules/autoencoder.py:25-39 (GroupNorm), :63-77 (ResBlock), :89-91, :104-106,
 *        vqvae/modules/vector_quantizers.py:23-61, :128-180, :290-356 (quantizers),
 *        vqvae/model.py:272 (MSE), :428 (AdamW).
 */
#ifndef VQK_H_
#define VQK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VQK_F32 0
#define VQK_BF16 1

enum vqk_status {
    VQK_OK = 0,
    VQK_ERR_SHAPE = -1,       /* unsupported / inconsistent shape */
    VQK_ERR_DTYPE = -2,       /* unsupported dtype code */
    VQK_ERR_ALIGN = -3,       /* pointer not 16-byte aligned */
    VQK_ERR_LAUNCH = -4,      /* hipGetLastError() after launch */
    VQK_ERR_ARG = -5,         /* NULL / out-of-range argument */
    VQK_ERR_WORKSPACE = -6    /* workspace too small */
};

const char* vqk_status_str(int status);
int vqk_version(void);
/* name of the code-object architecture this library was built for ("gfx950") */
const char* vqk_arch(void);

/* ---------------------------------------------------------------- vector quantizer ---------
 * vector_quantizers.py:33-44 / :337-343.  z[N][D], e[K][D] fp32 row-major, D % 8 == 0.
 * assoc 0: d = (|z|^2 + |e|^2) - 2 z.e (Standard/EMA); assoc 1: d = (|z|^2 - 2 z.e) + |e|^2 (Entropy).
 * Products are exact fp32 (v_mfma_f32_32x32x2_f32), first minimum wins. */
int vqk_row_sqnorm_f32(const float* x, int64_t rows, int d, float* out, void* stream);
int vqk_vq_assign_f32(const float* z, const float* e, const float* z2, const float* e2,
                      int64_t n, int k, int d, int assoc, int64_t* idx, void* stream);
/* vqk_vq_assign_f32 with the same result (bit-identical indices) computed as a bf16-MFMA candidate filter + an exact fp32
 * re-rank of the candidates (csrc/vq_filter.hip states the error bound that keeps the exact winner inside the candidate
 * set); d == 256, k % 32 == 0.  ws: >= vqk_vq_filter_ws_bytes(k, d) bytes of device scratch (the bf16 codebook and the
 * per-code filter margins, rebuilt by every call).  VQK_ERR_SHAPE when not served (nothing launched). */
int64_t vqk_vq_filter_ws_bytes(int k, int d);
int vqk_vq_assign_filtered_f32(const float* z, const float* e, const float* z2, const float* e2, int64_t n, int k, int d,
                               int assoc, int64_t* idx, void* ws, int64_t ws_bytes, void* stream);
/* The quantizer FORWARD as one kernel (vector_quantizers.py:37-56): the filtered assignment above with |z|^2 computed in the
 * kernel (same bits as vqk_row_sqnorm_f32) and the gather fused into its epilogue -- idx[N]; optionally q = e[idx] as fp32
 * (q) and / or bf16 (q_lo), sse[0] += sum (q - z)^2, hist[idx] += 1 (sse / hist pre-zeroed by the caller).
 * ws = what vqk_vq_prepare_f32 built from THIS codebook (bf16 fragment-major copy, |e|^2, filter margins, max |e|^2:
 * vqk_vq_filter_ws_bytes(k, d) bytes): call it once per codebook CHANGE (optimizer step, EMA update), not per step.
 * d == 256, k % 32 == 0 (VQK_ERR_SHAPE otherwise, nothing launched). */
int vqk_vq_prepare_f32(const float* e, int k, int d, void* ws, int64_t ws_bytes, void* stream);
int vqk_vq_forward_f32(const float* z, const float* e, const void* ws, int64_t ws_bytes, int64_t n, int k, int d, int assoc,
                       int64_t* idx, float* q /* optional */, void* q_lo /* optional */, float* sse /* optional */,
                       int32_t* hist /* optional */, void* stream);
/* Same search, additionally writing the full fp32 distance matrix dmat[N][K] (Entropy quantizer).  d == 256, N % 128 == 0, K % 32 == 0
 * and a stream scratch of >= 5 * parts * N * 4 bytes (vqk_set_scratch; parts <= 16): the code tiles go through LDS once per 128 rows
 * (vq_assign_lds_kernel, blocks = N/128 x parts) and a second small launch folds the parts' (min, argmin) and log-sum-exp states;
 * otherwise every wave streams the codebook itself.  Same distances and indices, bit for bit, either way. */
int vqk_vq_distances_f32(const float* z, const float* e, const float* z2, const float* e2,
                         int64_t n, int k, int d, int assoc, int64_t* idx, float* dmat, void* stream);
/* vqk_vq_distances_f32 (d == 256) that also leaves the softmax row statistics of a = -d / temperature: lse[N], hrow[N] (sample
 * entropies) and hsum[0] += sum_i h_i (pre-zeroed) -- an online log-sum-exp under the distance MFMAs; follow it with
 * vqk_entropy_forward_presummed_f32 (column means + finalize) instead of vqk_entropy_forward_f32. */
int vqk_vq_distances_stats_f32(const float* z, const float* e, const float* z2, const float* e2, int64_t n, int k, int d,
                               int assoc, int64_t* idx, float* dmat, float temperature, float* lse, float* hrow, float* hsum,
                               void* stream);
int vqk_entropy_forward_presummed_f32(const float* dmat, int64_t n, int k, float temperature, const float* lse, float* psum,
                                      float* u, float* avg_term, void* stream);
/* Entropy loss on dmat (vector_quantizers.py:296-328, 'softmax' type): per row lse[N], hrow[N] (sample entropies),
 * hsum[0] += sum_i h_i, psum[K] += sum_i p_ik, u[K] = log(pbar+1e-5) + pbar/(pbar+1e-5), avg_term[0] += sum_k pbar
 * log(pbar+1e-5) (pbar = psum/N).  hsum, psum, avg_term pre-zeroed.  loss_ent = ratio * (hsum/N + avg_term). */
int vqk_entropy_forward_f32(const float* dmat, int64_t n, int k, float temperature, float* lse, float* hrow,
                            float* hsum, float* psum, float* u, float* avg_term, void* stream);
/* In place dmat <- dL_ent/dd (scaled by *gscale_dev): the cotangent that the two GEMMs turn into dz and dE. */
int vqk_entropy_backward_f32(float* dmat, const float* lse, const float* hrow, const float* u, int64_t n, int k,
                             float temperature, float ratio, const float* gscale_dev, void* stream);
/* The same cotangent as TWO bf16 matrices hi, lo [N][K] (hi = bf16(dd), lo = bf16(dd - hi); dmat is left untouched; K % 4 == 0):
 * the operands of split-product GEMMs on the bf16 MFMA kernels (dd E ~ hi E_hi + hi E_lo + lo E_hi, relative error ~2^-16) where the
 * reference multiplies in fp32 (vector_quantizers.py:296-356 differentiated; throughput mode only). */
int vqk_entropy_backward_split_f32(const float* dmat, const float* lse, const float* hrow, const float* u, int64_t n, int k,
                                   float temperature, float ratio, const float* gscale_dev, void* hi, void* lo, void* stream);
/* ent_loss_type == 'argmax' (vector_quantizers.py:311-315): the targets are one_hot(argmax_k a) with the gradient of p
 * (straight-through).  idx = the assignment of vqk_vq_distances_f32 (argmax a == argmin d, first wins), hist = its code
 * histogram (vqk_vq_gather_f32).  Forward: lse / hrow as above, ssum += sum_i (lse_i - a_i[idx_i]), mbuf[k] scratch,
 * u / avg_term from m = hist / N;  loss_ent = ratio * (ssum / N + avg_term).  Backward overwrites dmat with dL/dd. */
int vqk_entropy_argmax_forward_f32(const float* dmat, const int64_t* idx, const int32_t* hist, int64_t n, int k,
                                   float temperature, float* lse, float* hrow, float* hsum, float* ssum, float* mbuf,
                                   float* u, float* avg_term, void* stream);
int vqk_entropy_argmax_backward_f32(float* dmat, const int64_t* idx, const float* lse, const float* hrow, const float* u,
                                    int64_t n, int k, float temperature, float ratio, const float* gscale_dev,
                                    void* stream);
/* out[r][c] += a * scale[r] * m[r][c] */
int vqk_row_scale_add_f32(float* out, const float* m, const float* scale, int64_t rows, int c, float a, void* stream);
/* Gumbel-softmax rows (vector_quantizers.py:232-243): y = softmax((logits - log(noise))/tau) [hard: one-hot of its
 * argmax] written as `dtype`, idx = argmax y, klsum[0] += sum_i sum_n qy log(qy*K + 1e-10), qy = softmax(logits).
 * noise ~ Exp(1) is drawn by the caller (the reference's F.gumbel_softmax draws the same tensor first). */
int vqk_gumbel_forward(int dtype, const float* logits, const float* noise, int64_t n, int k, float tau, int hard,
                       void* y, int64_t* idx, float* klsum, int32_t* hist /* optional */,
                       const float* sched_dev /* optional */, void* stream);
/* dlogits = y (dy - <y,dy>)/tau + s*(kl_cost/n) qy (r - <qy,r>), s = *gscale_dev (soft sample path).
 * sched_dev (both calls, optional): device floats {tau, kl_cost} that REPLACE the scalar arguments -- the reference schedules
 * both per step (vqvae/model.py:218-225); reading them on the device lets a captured hipGraph follow the schedule. */
int vqk_gumbel_backward(int dtype, const float* logits, const float* noise, const void* dy, int64_t n, int k,
                        float tau, float kl_cost, const float* gscale_dev, float* dlogits,
                        const float* sched_dev /* optional */, void* stream);
/* q = e[idx] (written as fp32 and, if q_lo != NULL, also as bf16), sse[0] += sum (q - z)^2,
 * hist[idx] += 1 (int32, optional).  sse must be zeroed by the caller. */
int vqk_vq_gather_f32(const float* z, const float* e, const int64_t* idx, int64_t n, int k, int d,
                      float* q, void* q_lo, float* sse, int32_t* hist, void* stream);
/* dz = dq + s*cz * (z - q)   ;   de[idx] += s*ce * (q - z)   (de optional, pre-zeroed by caller)
 * s = *gscale_dev (a device scalar: the upstream gradient of the loss; NULL = 1), dq optional (NULL = 0),
 * fp32 or bf16 (dq_dtype).  vector_quantizers.py:52-56 differentiated. */
int vqk_vq_backward_f32(const float* z, const float* e, const int64_t* idx, const void* dq, int dq_dtype,
                        int64_t n, int k, int d, float cz, float ce, const float* gscale_dev, float* dz, float* de,
                        void* stream);
/* vqk_vq_backward_f32 as ONE kernel (d == 256): the rows of a 32-row block that share a code are summed in LDS, one
 * coalesced fp32 atomic row per distinct code and block goes to de (arrival order: not for deterministic mode). */
int vqk_vq_backward_fused_f32(const float* z, const float* e, const int64_t* idx, const void* dq, int dq_dtype,
                              int64_t n, int k, int d, float cz, float ce, const float* gscale_dev, float* dz, float* de,
                              void* stream);
/* EMA statistics (vector_quantizers.py:159-169): counts[k] += 1, dw[idx] += z (both pre-zeroed) ... */
int vqk_ema_stats_f32(const float* z, const int64_t* idx, int64_t n, int k, int d,
                      float* counts, float* dw, void* stream);
/* vqk_ema_stats_f32 with the rows of a 32-row block that share a code summed in LDS first (d == 256): one coalesced atomic row
 * per distinct code and block instead of one atomic per (row, channel). */
int vqk_ema_stats_fused_f32(const float* z, const int64_t* idx, int64_t n, int k, int d, float* counts, float* dw, void* stream);
/* ... and the update: count' = smooth(decay*count + (1-decay)*n_k), weight' = decay*weight + (1-decay)*dw,
 * codebook = weight'/count'.  `batch` is the smoothing constant b (global image batch). */
int vqk_ema_update_f32(float* ema_count, float* ema_weight, float* codebook, const float* counts, const float* dw,
                       int k, int d, float decay, float eps, float batch, void* stream);

/* ---------------------------------------------------------------- convolution --------------
 * Implicit-GEMM conv, stride 1, 'same' zero padding, ksize in {1,3}.
 *   x [N][Hin][Win][Cin], w [Cout][ks][ks][Cin], y [N][H][W][Cout]; (H,W) = (Hin,Win) << ups.
 *   ups = 1 fuses a nearest x2 upsample of x into the input addressing (autoencoder.py:104-106).
 *   bias (fp32, optional), residual (same dtype/shape as y, optional) are fused into the epilogue;
 *   act: 0 none, 1 tanh, 2 relu, 3 leaky-relu(0.2).  in/w dtype = `dtype`; y/residual dtype = out_dtype.
 *   Cin must be a multiple of 16 bytes worth of elements (4 fp32 / 8 bf16).
 * dgrad is the same entry point with weights repacked by vqk_conv_pack_dgrad. */
int vqk_conv2d_fprop(int dtype, const void* x, const void* w, const float* bias, const void* residual, void* y,
                     int out_dtype, int n, int h_in, int w_in, int cin, int cout, int ksize, int ups, int act,
                     int wlayout, const void* zeros, void* stream);
/* The same conv with a 2x2 pooling of its output fused into the epilogue: y[N][H/2][W/2][Cout] = pool_scale * (sum over
 * each 2x2 block of conv(x) + bias + residual) -- pool_scale 0.25: the avg_pool2d that follows a level's last ResBlock
 * (autoencoder.py:89-91, residual = the block's skip input at full resolution); pool_scale 1: the backward of the nearest
 * x2 upsample in front of an Upsample conv (:104-106).  bf16 only, fragment-major weights (layout 1), Cout % 128 == 0;
 * VQK_ERR_SHAPE when the problem is not eligible (callers then use vqk_conv2d_fprop + vqk_pool2x2). */
int vqk_conv2d_fprop_pooled(int dtype, const void* x, const void* w, const float* bias, const void* residual, void* y,
                            int n, int h_in, int w_in, int cin, int cout, int ksize, int ups, float pool_scale,
                            const void* zeros, void* stream);
/* The 3x3 conv (optionally with the fused 2x2 pooling) that ALSO accumulates the GroupNorm statistics of its output: the
 * per-(sample, group) sums of y and y*y (y as stored: rounded to bf16) are added to gn_ws[n][group][2] (doubles), the
 * workspace of vqk_gn_forward -- the consumer then calls vqk_gn_forward_presummed and the statistics pass over y
 * (autoencoder.py:25-39 re-reads its input once for mean / variance) disappears.  bf16, fragment-major weights, Cout % 128
 * == 0, (Cout / groups) % 4 == 0; VQK_ERR_SHAPE when the problem is not served by the matrix/auxiliary-wave kernel
 * (nothing has been launched: callers fall back to vqk_conv2d_fprop[_pooled] + vqk_gn_forward). */
int vqk_conv2d_fprop_gnstats(int dtype, const void* x, const void* w, const float* bias, const void* residual, void* y,
                             int n, int h_in, int w_in, int cin, int cout, int ksize, int ups, int pool, float pool_scale,
                             double* gn_ws, int groups, const void* zeros, void* stream);
/* The 3x3 conv on the padded 3-channel image (the encoder's first conv, autoencoder.py:132; x [N][H][W][8], w [Cout][3][3][8] in the compute
 * dtype, layout 0) with the GroupNorm sums of its output left in gn_ws as vqk_conv2d_fprop_gnstats leaves them.  Served: bf16,
 * Cout = 128 in 32 groups, W % 32 == 0, not in deterministic mode; VQK_ERR_SHAPE otherwise (nothing launched: callers run
 * vqk_conv2d_fprop and vqk_gn_forward). */
int vqk_conv2d_thin_in_gnstats(int dtype, const void* x, const void* w, const float* bias, void* y, int n, int h, int wd, int cout,
                               double* gn_ws, int groups, void* stream);
/* The nearest-x2 upsample + 3x3 conv of autoencoder.py:104-106 in PHASE form: output pixel (2i+a, 2j+b) only sees a 2x2
 * window of the low-resolution input with pre-summed weights, so the conv runs as four 2x2-tap launches (4/9 of the
 * multiply-adds).  w4: the operand of vqk_conv_pack_weights(..., layout 2) -- transpose = 0 for backward = 0 (x [N][h][w][Cin]
 * -> y [N][2h][2w][Cout] (+ bias; gn_ws as in vqk_conv2d_fprop_gnstats, or NULL)), transpose = 1 for backward = 1 (x = dy
 * [N][2h][2w][Cin := conv's Cout] -> y = dx [N][h][w][Cout := conv's Cin], the four phases accumulate in place).  bf16,
 * Cout % 128 == 0, Cin % 64 == 0; VQK_ERR_SHAPE when the matrix/auxiliary-wave kernel does not serve the problem (nothing
 * launched: callers use vqk_conv2d_fprop with ups = 1 / the pooled data gradient).
 * dtype VQK_F32 (this entry and the two pooled forms below): the SPLIT-PRODUCT mode -- fp32 tensors, w4 in layout 6, every product as
 * three bf16 products (csrc/conv_x3.hip, NTAP = 4), one launch for the four phases; Cout % 128 == 0, Cin % 32 == 0, h % 8 == 0,
 * w % 16 == 0 (h, w: the LOW resolution); gn_ws: groups of 4 / 8 / 16 channels, not in deterministic mode. */
int vqk_conv2d_ups_phase(int dtype, const void* x, const void* w4, const float* bias, void* y, int n, int h, int w,
                         int cin, int cout, int backward, double* gn_ws, int groups, const void* zeros, void* stream);
/* Data gradient of a 3x3 conv that is FOLLOWED by a 2x2 average pool (the encoder's Downsample in a ResBlock's last conv,
 * vqvae/modules/autoencoder.py:89-91), from the POOLED gradient, in phase form: dx [n, 2h, 2w, cout] = scale * (nearest-x2(dy_pooled)
 * conv flip(W)^T) -- every full-resolution pixel of a 2x2 block sees the same pooled gradient, so phase (a, b) of dx reads a 2x2
 * window of dy_pooled [n, h, w, cin] with pre-summed weights: 4/9 of the multiply-adds of the tap form (vqk_conv2d_general with the
 * nearest-x2 addressing).  w4t = vqk_conv_pack_weights(..., transpose = 1, layout = 2) of the conv's weight (the operand of
 * vqk_conv2d_ups_phase(backward = 1)); scale = the pool's 0.25.  Same shape rules as vqk_conv2d_ups_phase; VQK_ERR_SHAPE when not served. */
int vqk_conv2d_pooled_dgrad_phase(int dtype, const void* dy_pooled, const void* w4t, void* dx, int n, int h, int w, int cin,
                                  int cout, float scale, const void* zeros, void* stream);
/* FORWARD of a 3x3 conv followed by a 2x2 average pool, as the 4x4 stride-2 conv it is: y [n, h, w, cout] = scale * (sum over each
 * 2x2 block of conv3x3(x, W)) + res_pooled, x [n, 2h, 2w, cin].  One data-gradient-type phase launch (the tile accumulates the four
 * input phases in registers and is stored once) with the conv's FORWARD phase operand w4 = vqk_conv_pack_weights(..., transpose = 0,
 * layout = 2), phase blocks reversed: 4/9 of the multiply-adds of the conv + pooling-drain form (vqk_conv2d_fprop_pooled).
 * res_pooled (may be NULL): the POOLED residual, [n, h, w, cout]; gn_ws / groups as in vqk_conv2d_fprop_gnstats (sums of y).
 * Same shape rules as vqk_conv2d_ups_phase; VQK_ERR_SHAPE when not served. */
int vqk_conv2d_pooled_fprop_phase(int dtype, const void* x, const void* w4, const void* res_pooled, void* y, int n, int h, int w,
                                  int cin, int cout, float scale, double* gn_ws, int groups, const void* zeros, void* stream);
/* General form (im2col kernel when not plain): stride in {1,2}, explicit zero padding `pad`, explicit output size;
 * mode 0: x as is, 1: nearest x2 upsample of x, 2: x zero-stuffed x2 (the dgrad of a stride-2 conv, with flipped /
 * transposed weights and pad = ks-1-pad_fwd).  Epilogue: y = out_gain * act(acc * acc_scale + bias) + residual, act 0
 * none, 1 tanh, 2 relu, 3 leaky-relu(0.2) -- acc_scale is the StyleGAN2 runtime weight gain (discriminator.py:165),
 * out_gain the bias_act gain (bias_act.py:55).  Strided/padded problems need weight layout 0. */
int vqk_conv2d_general(int dtype, const void* x, const void* w, const float* bias, const void* residual, void* y,
                       int out_dtype, int n, int h_in, int w_in, int cin, int cout, int ksize, int stride, int pad,
                       int mode, int h_out, int w_out, int act, float acc_scale, float out_gain, int wlayout,
                       const void* zeros, void* stream);
/* The 3x3 weight gradient when dy is the gradient of a FUSED 2x2 average pool (vqk_conv2d_fprop_pooled): dy_pooled is
 * [N][h/2][w/2][Cout], every pooled pixel stands for its 2x2 block, dW += scale * sum over the h x w pixels (scale = the
 * pool's 0.25): the full-resolution copy of the gradient is never written.  bf16, h % 8 == 0, w % 16 == 0, Cin % 64 == 0,
 * Cout % 64 == 0; VQK_ERR_SHAPE when not served (nothing launched: callers unpool and call vqk_conv2d_wgrad). */
int vqk_conv2d_wgrad_pooled_dy(int dtype, const void* x, const void* dy_pooled, float* dw, int n, int h, int w, int cin,
                               int cout, float scale, const void* zeros, void* stream);
/* Weight gradient of a nearest-x2 UPSAMPLE + 3x3 conv (vqvae/modules/autoencoder.py:102-105) in PHASE form (csrc/conv_wgmx.hip):
 * x [n, h, w, cin] is the LOW-resolution input, dy [n, 2h, 2w, cout] the gradient of the conv's output;
 * dw[Cout][3][3][Cin] += scale * the gradient, as four launches of a 2x2-window kernel (one per output phase: 4/9 of the
 * multiply-adds of the tap form vqk_conv2d_wgrad(..., ups = 1)).  bf16, h % 8 == 0, w % 16 == 0, cin % 64 == 0, cout % 64 == 0;
 * VQK_ERR_SHAPE when not served (deterministic mode included: nothing launched, callers use the tap form). */
int vqk_conv2d_wgrad_ups_phase(int dtype, const void* x, const void* dy, float* dw, int n, int h, int w, int cin, int cout,
                               float scale, const void* zeros, void* stream);
/* Weight gradient of a 3x3 conv FOLLOWED by a 2x2 average pool from the pooled gradient, in phase form (the mirror image of
 * vqk_conv2d_wgrad_ups_phase: operands' roles swapped, csrc/conv_wgmx.hip): x [n, 2h, 2w, cin] the conv's input, dy_pooled
 * [n, h, w, cout]; dw[Cout][3][3][Cin] += scale * wgrad(x, unpool(dy_pooled)) with 4/9 of the multiply-adds of
 * vqk_conv2d_wgrad_pooled_dy.  Same shape rules; VQK_ERR_SHAPE when not served (deterministic mode included). */
int vqk_conv2d_wgrad_pooled_dy_phase(int dtype, const void* x, const void* dy_pooled, float* dw, int n, int h, int w, int cin,
                                     int cout, float scale, const void* zeros, void* stream);
int vqk_conv2d_wgrad_general(int dtype, const void* x, const void* dy, float* dw, int n, int h_in, int w_in, int cin,
                             int cout, int ksize, int stride, int pad, int mode, int h_out, int w_out,
                             const void* zeros, void* stream);
/* Split-product mode (layout 5 above), weight gradient of a 3x3 conv (optionally behind a nearest-x2 upsample, ups = 1):
 * vqk_split_pair_f32 turns an fp32 [rows][c] activation / gradient into its bf16 (hi | lo) PAIR tensor [rows][2c]
 * (hi = bf16(v), lo = bf16(v - hi); c % 8 == 0); vqk_conv2d_wgrad_x3 takes x_pair [n][h_in][w_in][2 cin] and dy_pair
 * [n][h][w][2 cout] and adds scale * (dy_hi^T x_hi + dy_hi^T x_lo + dy_lo^T x_hi) to dw fp32 [Cout][3][3][Cin]: three tile classes
 * of ONE launch of the bf16 matrix/auxiliary-wave weight-gradient kernel (csrc/conv_wgmx.hip), folded by its atomic pass.
 * cin % 64 == 0, cout % 64 == 0, h % 8 == 0, w % 16 == 0; VQK_ERR_SHAPE when not served (deterministic mode included:
 * nothing launched, callers use the exact-fp32 vqk_conv2d_wgrad). */
/* vqk_conv2d_wgrad_x3_f32: the same gradient straight from the fp32 tensors x [n][h_in][w_in][cin], dy [n][h][w][cout] -- both are
 * split in registers on their way into LDS and all three products come from one staged patch (csrc/conv_x3.hip:
 * conv3x3_wgrad_x3_kernel; no pair tensors, no split passes).  cin % 64 == 0, cout % 64 == 0, h % 8 == 0, w % 8 == 0.
 * ups = 1 with h_in % 8 == 0, w_in % 8 == 0 runs in PHASE form (four 2x2-window launches' worth of blocks on the low-resolution grid,
 * 4/9 of the MFMAs; tuning slot X3_WGRAD_PHASE = 0: the tap form).  ups = 2: the gradient of a conv that is FOLLOWED by a 2x2 average
 * pool, from the POOLED gradient -- dy [n][h_in / 2][w_in / 2][cout], x [n][h_in][w_in][cin], scale = the pool's 0.25: phase form only
 * (h_in % 16 == 0, w_in % 16 == 0; VQK_ERR_SHAPE otherwise, nothing launched). */
/* vqk_conv2d_fprop_x3_gnstats: the layout-5 3x3 conv with the GroupNorm sums of its OUTPUT (sum, sum of squares per (sample, group),
 * fp64 atomics) left in gn_ws as vqk_conv2d_fprop_gnstats leaves them for the bf16 convs -- the consuming vqk_gn_forward_presummed
 * skips its statistics pass.  Cout % 128 == 0, Cout / groups in {4, 8, 16}; VQK_ERR_SHAPE otherwise and in deterministic mode. */
int vqk_conv2d_fprop_x3_gnstats(const float* x, const void* w5, const float* bias, const float* residual, float* y, int n, int h_in,
                                int w_in, int cin, int cout, int ups, double* gn_ws, int groups, const void* zeros, void* stream);
int vqk_split_pair_f32(const float* src, void* dst, int64_t rows, int c, void* stream);
int vqk_conv2d_wgrad_x3_f32(const float* x, const float* dy, float* dw, int n, int h_in, int w_in, int cin, int cout, int ups,
                            float scale, void* stream);
int vqk_conv2d_wgrad_x3(const void* x_pair, const void* dy_pair, float* dw, int n, int h_in, int w_in, int cin, int cout, int ups,
                        float scale, const void* zeros, void* stream);
/* the same with dW += scale * (the gradient): a layer whose backward carries a scalar gain (the discriminator's linear skip convs:
 * weight gain x output gain) accumulates straight into an optimizer's gradient arena -- no zeroed temporary, scale pass and add */
int vqk_conv2d_wgrad_general_scaled(int dtype, const void* x, const void* dy, float* dw, int n, int h_in, int w_in, int cin,
                                    int cout, int ksize, int stride, int pad, int mode, int h_out, int w_out, float scale,
                                    const void* zeros, void* stream);
/* The STRIDE-2 3x3 conv without padding of the StyleGAN2 discriminator's down-sampling layers (the reference's
 * conv2d_resample.py:119-122: upfirdn2d blur with padding, then F.conv2d(stride=2)) and its data gradient on the
 * matrix/auxiliary-wave kernel (bf16).  vqk_conv2d_s2_supported: 1 when the shape is served (backward = 0: the conv, Cin % 64
 * == 0, Cout % 128 == 0, W_out % 32 == 0 && H_out % 4 == 0 or W_out % 16 == 0 && H_out % 8 == 0; backward = 1: its data gradient,
 * Cout % 64 == 0, Cin % 128 == 0, maps of 32x32 and more), 0 otherwise (callers use vqk_conv2d_general).
 * vqk_conv2d_s2_fprop: x [N][2 H_out + 1][2 W_out + 1][Cin] -> y [N][H_out][W_out][Cout] = out_gain * act(acc_scale * conv + bias);
 * wq = vqk_conv_pack_weights(..., transpose 0, layout 1).  The four parity sub-images of an input patch are staged side by
 * side in LDS, so every tap reads at unit stride.
 * vqk_conv2d_s2_dgrad: dy [N][H_out][W_out][Cout] -> dx [N][2 H_out + 1][2 W_out + 1][Cin] = acc_scale * gradient, EVERY element
 * written; w3 = vqk_conv_pack_weights(..., transpose 1, layout 3) (the four output parities' 4 / 2 / 2 / 1 taps), wt0 = the same
 * weights with transpose 1, layout 0 (the last row and the last column of dx run as parity classes of the im2col kernel). */
int vqk_conv2d_s2_supported(int dtype, int n, int h_out, int w_out, int cin, int cout, int backward);
int vqk_conv2d_s2_fprop(int dtype, const void* x, const void* wq, const float* bias, void* y, int n, int h_out, int w_out,
                        int cin, int cout, int act, float acc_scale, float out_gain, const void* zeros, void* stream);
int vqk_conv2d_s2_dgrad(int dtype, const void* dy, const void* w3, const void* wt0, void* dx, int n, int h_out, int w_out,
                        int cin, int cout, float acc_scale, const void* zeros, void* stream);
/* Weight operand layouts.  0: [Cout][ks][ks][Cin] (any shape).  1: "fragment-major" for the register-weight halo
 * kernel (3x3, Cin a whole 128-byte chunk, W%32==0 && H%8==0 or W%16==0 && H%16==0): Cout padded to a multiple of
 * 128, element order [Cout/32][64-byte Cin chunk][tap][k-substep 0..1][lane 0..63][16 bytes]: every MFMA operand
 * fragment is one coalesced 1 KiB load and the 18 fragments of a (cout tile, chunk) are contiguous.  vqk_conv_weight_layout() returns the layout the fprop launcher wants for
 * a problem (>= 0) or a negative status; vqk_conv_packed_elems() the element count of the packed buffer;
 * vqk_conv_pack_weights() builds it from the fp32 [Cout][ks][ks][Cin] master (transpose = 1: the dgrad operand,
 * i.e. Cin/Cout swapped and both taps flipped).  2: the upsample-phase operand of vqk_conv2d_ups_phase (bf16, 3x3): four
 * phases (a, b) of fragment-major blocks with FOUR taps each, tap (r, s) of a phase = the sum of the 3x3 taps that fall
 * on the same low-resolution pixel (rows {0},{1,2} for a = 0 and {0,1},{2} for a = 1; columns alike); transpose = 1:
 * channels swapped and the 2x2 taps mirrored.  5: the SPLIT-PRODUCT operand (csrc/conv_x3.hip; dtype VQK_F32, 3x3 or 1x1, Cin % 32 == 0):
 * layout 1 in bf16 with every fragment twice, hi = bf16(w) then lo = bf16(w - hi) -- [Cout/32][32-channel chunk][tap][k-substep]
 * [hi | lo][lane][16 bytes], 4 bytes per weight like the fp32 operands.  vqk_conv2d_fprop(VQK_F32, ..., out VQK_F32, wlayout 5)
 * then evaluates every product as x_hi w_hi + x_lo w_hi + x_hi w_lo on the bf16 matrix pipe with fp32 accumulation (the fp32
 * activations are split on the fly; ~2^-17 relative per product): H % 8 == 0, W % 16 == 0 at the OUTPUT resolution, Cout % 4 == 0;
 * callers choose it (vqk_conv_weight_layout never returns 5; layout 1 / dtype VQK_F32 stays the exact v_mfma_f32_32x32x2_f32 mode).
 * 6: layout 2's phase-summed four-tap operand (the sums taken in fp32) stored as layout 5's (hi | lo) fragment pairs -- the operand of the
 * phase entry points with dtype VQK_F32: [phase][Cout/32][32-channel chunk][tap 0..3][k-substep][hi | lo][lane][16 bytes]. */
int vqk_conv_weight_layout(int dtype, int n, int h_in, int w_in, int cin, int cout, int ksize, int ups);
int64_t vqk_conv_packed_elems(int cout, int cin, int ksize, int layout);
int vqk_conv_pack_weights(const float* w, void* out, int dtype, int cout, int cin, int ksize, int transpose,
                          int layout, void* stream);
/* Every conv operand of a model in ONE launch: descs_dev is a device array of ndesc descriptors of eight int64 words
 * {src, dst, dtype, cout, cin, ksize, transpose, layout}, each with the meaning of the vqk_conv_pack_weights
 * arguments.  Issued once per optimizer step (after vqk_adamw) instead of one pack launch per conv call; replaces
 * the per-call weight casts autocast performs in the reference's conv calls (vqvae/modules/autoencoder.py:57-60). */
int vqk_conv_pack_multi(const int64_t* descs_dev, int ndesc, int blocks_per_desc, void* stream);
/* test / tuning hook for the fprop kernel choice: -1 automatic, 0 im2col kernel only, 1 halo kernels when
 * eligible, 2 halo kernel with LDS-staged weights (layout 0) instead of register weights, 3 one-tile-per-block
 * register-weight halo kernel instead of the persistent stream kernel (bf16), 4 stream kernel without its half-tile
 * (128-pixel) form for maps with fewer 256-pixel tiles than CUs */
int vqk_conv_set_variant(int variant);
/* Deterministic mode (the reference trains with `deterministic=True`, vqvae/train.py:130).  on = 1: the float accumulations of
 * the train step that are otherwise combined with atomics in arrival order -- split-K partials of the weight gradients, column
 * sums (bias gradients), the cross-block sums of the GroupNorm statistics / backward reductions and of d gamma / d beta, the
 * per-code sums of the codebook gradient -- go through `ws` (>= 64 MiB recommended; VQK_ERR_WORKSPACE when a call needs more)
 * as per-block partials that are added in index order, or run unsplit.  THREAD-LOCAL like the block caps; the workspace must
 * belong to the stream the following launches go to (two streams = two workspaces, re-armed on every switch).  The scalar
 * loss sums and the EMA statistics keep their atomics (values, not gradients). */
int vqk_set_deterministic(int on, void* ws, int64_t ws_bytes);
/* Optional fp32 scratch of the CURRENT stream (thread-local like the block caps; NULL = none; contents need not be
 * initialised).  With it the general conv kernel splits K for problems whose pixel x cout tile grid would leave most of the
 * chip idle (the discriminator's 4x4 / 8x8 convs and fully connected layers: 8-32 tiles with K up to 8192): every split
 * stores its partial tile into its own slice and an epilogue kernel adds the slices in split order (no atomics: the same
 * bits every run, so the split is also taken in deterministic mode).  Needs >= 2 * pixels * cout * 4 bytes to be used.
 * ws must be 16-byte aligned (VQK_ERR_ALIGN), ws_bytes >= 0 (VQK_ERR_ARG); the same checks apply to vqk_set_deterministic. */
int vqk_set_scratch(void* ws, int64_t ws_bytes);
/* DYNAMIC TILE QUEUE of the persistent matrix/auxiliary-wave conv kernel (csrc/conv_mx.hip; tuning slot TILE_QUEUE: 0 = off,
 * the static share; 1 = a block's first tile is its static one, the rest are drawn; 2 = every tile is drawn).  ws: >= 64 bytes
 * of int32 words, ZERO on the first use; the kernels leave them zero again (the last block of a launch to run out of tiles
 * resets them), so launches that follow each other on ONE stream may share them -- launches that can run CONCURRENTLY (other
 * streams) need words of their own: the pointer is THREAD-LOCAL like vqk_set_scratch and is meant to follow the caller's
 * current stream.  ws = NULL: static share.  Results do not depend on the mode (a tile's arithmetic and its destination are
 * functions of the tile index only).  VQK_ERR_WORKSPACE below 64 bytes, VQK_ERR_ALIGN unless 16-byte aligned.
 * A kernel that is KILLED mid-way leaves the words dirty: re-zero them before the next launch. */
int vqk_set_tile_queue(void* ws, int64_t ws_bytes);
/* WORKSPACE CONTEXTS (round 6): the three workspaces above as fields of an object instead of three thread-local pointers.
 * A context is plain host memory (no device allocation, no stream): create one per (device, stream, host thread) that launches,
 * give it its workspaces ONCE, and name it before a batch of launches -- vqk_ctx_make_current is one pointer store, the only
 * thread-local state left on this path (the analogue of hipSetDevice).  ctx = NULL in a vqk_ctx_set_* call addresses the calling
 * thread's current context, which is what vqk_set_scratch / vqk_set_tile_queue / vqk_set_deterministic do: they are wrappers.
 * vqk_ctx_make_current(NULL) returns to the thread's own default context.  Two models stepped from two host threads, or one
 * thread alternating between two streams, hold one context each (ops.py::_stream).  A context must not be destroyed while a
 * thread still has it current (the destroying thread's own current pointer is reset). */
typedef struct vqk_ctx vqk_ctx;
int vqk_ctx_create(vqk_ctx** ctx);
int vqk_ctx_destroy(vqk_ctx* ctx);
int vqk_ctx_make_current(vqk_ctx* ctx);
int vqk_ctx_set_scratch(vqk_ctx* ctx, void* ws, int64_t ws_bytes);
int vqk_ctx_set_tile_queue(vqk_ctx* ctx, void* ws, int64_t ws_bytes);
int vqk_ctx_set_deterministic(vqk_ctx* ctx, int on, void* ws, int64_t ws_bytes);
/* Tuning slots: the launch heuristics that tools/ sweep (formerly read-once environment variables).  name = one of
 * vqk_tuning_name(0 .. vqk_tuning_count() - 1): MX, MX_1X1, TW16, STREAM_BLOCKS, MX_MIN_TILES, FPROP_SPLITK, SK_BLOCKS, SK_MINSTEPS, SK_MAXMB, UPS_PHASE, WGRAD_BLOCKS, WGMX, WGRAD_GEN_BLOCKS, WGRAD_NO_PW16, WGRAD_NO_P16K, MX_HALF, MX_HALF_HW, UPFIRDN_TILE, GN_BLOCKS_REDUCE, GN_BLOCKS_APPLY, GN_NT_MB, GN_NO_SMALL, WGMX_COEF_E4, GN_CLUSTER_MAX_HW, COMM_CUS, MX_QUARTER, MX_S2, MX_S2_DGRAD_MIN, UPS_MERGE, TILE_QUEUE.
 * A set slot overrides the built-in default at the next launch; vqk_reset_tuning returns every slot to its default.
 * Process-wide (relaxed atomics); VQK_ERR_ARG for an unknown name.  The Python host maps VQK_<NAME> environment variables
 * onto these calls when it loads the library (_native.py), so the A/B scripts keep their interface. */
int vqk_set_tuning(const char* name, int value);
int vqk_reset_tuning(void);
int vqk_tuning_count(void);
const char* vqk_tuning_name(int i);
/* caps on the persistent grids of the 3x3 fprop/dgrad kernel and of the all-taps wgrad kernel (0 = default: two
 * blocks per CU).  256 = one block per CU, leaving room for a kernel that runs concurrently on another stream
 * (the host overlaps a layer's wgrad with its dgrad and GroupNorm backward).  The caps are THREAD-LOCAL: they apply to the
 * launches the calling thread issues afterwards (set, launch, reset in one place), never to another thread or device
 * context.  vqk_conv_set_variant above is a process-wide TEST hook and not meant for product code. */
int vqk_conv_set_block_caps(int stream_blocks, int wgrad_blocks);
/* DIAGNOSTIC (tools/comm_probe.py): a stand-in for a collective's kernel -- `blocks` persistent 256-thread blocks stream
 * dst[i] += src[i] over `bytes` (16-byte aligned, a multiple of 16) `passes` times, sleeping `sleep` x 64 cycles between
 * 4-KiB pieces, so that a chosen number of CUs stays occupied for a chosen time while the train step runs on another stream. */
int vqk_probe_stream_add(const float* src, float* dst, int64_t bytes, int blocks, int passes, int sleep, void* stream);
/* BOX CALIBRATION (bench.py `box_calibration`; csrc/calib.hip) -- not part of the train step.  vqk_calib_fill: pseudo-random bf16
 * pairs of N(0,1)-like magnitude into w.  vqk_calib_mfma: `blocks` x 256 threads (one wave per SIMD) run `iters` x 18 phases of
 * the role-split conv kernel's matrix-wave instruction mix (4 LDS fragment reads + 2 L2 weight fragments + 8 bf16 32x32x16 MFMAs)
 * on the random operands of w (w_bytes: a power of two in [64 KiB, 1 GiB]); vqk_calib_mfma_flops = the multiply-add FLOPs of one
 * such launch.  vqk_calib_copy: dst = src, 16 bytes per lane, streaming (bytes % 16 == 0): HBM read + write bandwidth. */
int vqk_calib_fill(void* w, int64_t bytes, void* stream);
int vqk_calib_mfma(const void* w, int64_t w_bytes, float* sink, int iters, int blocks, void* stream);
int64_t vqk_calib_mfma_flops(int iters, int blocks);
int vqk_calib_copy(const void* src, void* dst, int64_t bytes, void* stream);
/* w [Cout][ks][ks][Cin] -> wt [Cin][ks][ks][Cout] with both taps flipped; src fp32, dst `dtype`. */
int vqk_conv_pack_dgrad(const float* w, void* wt, int dtype, int cout, int cin, int ksize, void* stream);
/* dw[Cout][ks][ks][Cin] (fp32) += sum_pix dy[pix][co] * x[pix (+) tap][ci].  dw must be pre-zeroed
 * (split-K partials are combined with fp32 atomics). */
int vqk_conv2d_wgrad(int dtype, const void* x, const void* dy, float* dw, int n, int h_in, int w_in, int cin,
                     int cout, int ksize, int ups, const void* zeros, void* stream);
/* The 3x3 weight gradient of the two EDGE convs (one 16-byte chunk of channels on one side: the padded 3-channel image /
 * reconstruction; vqvae/modules/autoencoder.py:114 and :170), bf16: (cin, cout) = (8, 128) or (128, 8), h*w % 128 == 0 and
 * (w % 128 == 0 or 128 % w == 0).
 * dw[Cout][3][3][Cin] += the gradient; split-K partials go through the caller's workspace `ws` (>=
 * vqk_conv2d_wgrad_edge_ws_bytes() bytes) and are summed in a fixed order: no atomics, run-to-run deterministic.
 * VQK_ERR_SHAPE when not served (nothing launched: callers use vqk_conv2d_wgrad). */
int64_t vqk_conv2d_wgrad_edge_ws_bytes(void);
int vqk_conv2d_wgrad_edge(int dtype, const void* x, const void* dy, float* dw, void* ws, int64_t ws_bytes, int n, int h,
                          int w, int cin, int cout, const void* zeros, void* stream);
/* vqk_conv2d_wgrad_edge with the TRUE channel count of the thin side (thin_true in 1..8; 3 for the image / reconstruction): dw is
 * the parameter's own unpadded gradient -- [128][3][3][thin_true] for (cin, cout) = (8, 128), [thin_true][3][3][128] for (128, 8) --
 * the sums of the zero-padded channels are dropped.  thin_true = 8: vqk_conv2d_wgrad_edge. */
int vqk_conv2d_wgrad_edge_true(int dtype, const void* x, const void* dy, float* dw, void* ws, int64_t ws_bytes, int n, int h,
                               int w, int cin, int cout, int thin_true, const void* zeros, void* stream);
/* The decoder's last conv (autoencoder.py:170): 3x3, stride 1, 'same', cin = 128 -> cout = 8 (the 3 image channels padded to
 * one 16-byte chunk), y = act(conv(x, w) + bias) with act 0 none / 1 tanh; bf16 in and out, h % 8 == 0, w % 32 == 0;
 * w: bf16 [8][3][3][128] (weight layout 0).  VQK_ERR_SHAPE when not served (nothing launched: callers use vqk_conv2d_fprop). */
int vqk_conv2d_thin_out(int dtype, const void* x, const void* w, const float* bias, void* y, int n, int h, int wd, int cin,
                        int cout, int act, const void* zeros, void* stream);
/* out[c] (+)= sum over rows of x[rows][c]  (bias gradients); out pre-zeroed. */
int vqk_colsum(int dtype, const void* x, int64_t rows, int c, float* out, void* stream);
/* out[i] += scale * sum_rows x[row][i] for the first c_out <= c columns only: `out` is the gradient of a bias whose layer carries
 * zero-padded output channels (the decoder's 3-channel head on 8) -- the sums go straight into the parameter's own (unpadded)
 * gradient; `scale` undoes a gain folded into x (the StyleGAN2 layers' runtime weight gain). */
int vqk_colsum_lead(int dtype, const void* x, int64_t rows, int c, int c_out, float scale, float* out, void* stream);
/* elementwise fp32 -> dtype cast (weight shadow copies) */
int vqk_cast(const float* src, void* dst, int dtype, int64_t n, void* stream);

/* ---------------------------------------------------------------- GroupNorm + SiLU ----------
 * autoencoder.py:25-39: per (sample, group) mean and UNBIASED variance; y = (x-mu)*rstd*w + b,
 * optionally followed by SiLU.  stats[n][g] = {mean, rstd} fp32.  acc = N*G*2 doubles scratch (zeroed
 * by the caller). */
int vqk_gn_stats(int dtype, const void* x, int n, int64_t hw, int c, int groups, float eps,
                 double* acc, float* stats, void* stream);
int vqk_gn_apply(int dtype, const void* x, const float* stats, const float* w, const float* b, void* y,
                 int n, int64_t hw, int c, int groups, int silu, void* stream);
/* statistics + normalisation (+SiLU) as one call: two launches (sums; finalize folded into the apply pass), stats
 * are also stored for the backward.  ws = N*G*2 doubles of sums + N 8-byte counter slots (N*G*2+N doubles), ZERO on entry and
 * left ZERO on exit (the last block of each sample in the apply pass clears that sample's slots), so one persistent workspace per stream serves every call
 * with no memset in between. */
int vqk_gn_forward(int dtype, const void* x, const float* w, const float* b, void* y, float* stats, double* ws, int n,
                   int64_t hw, int c, int groups, float eps, int silu, void* stream);
/* the same when the sums of x are already in ws (vqk_conv2d_fprop_gnstats of the producing conv): one launch */
int vqk_gn_forward_presummed(int dtype, const void* x, const float* w, const float* b, void* y, float* stats, double* ws,
                             int n, int64_t hw, int c, int groups, float eps, int silu, void* stream);
/* deterministic mode: the producing conv (vqk_conv2d_fprop_gnstats / vqk_conv2d_ups_phase while vqk_set_deterministic is on)
 * left the sums as ONE SLOT PER 256-pixel TILE, parts[((n * G + g) * nblk + tile) * 2 + j], nblk = (conv-resolution H * W) / 256
 * (plain stores, no zero-on-entry protocol: `ws` handed to the conv is `parts`, >= N*G*nblk*2 doubles).  The slots are added in
 * a fixed order into sums[N*G*2] (scratch), then the apply pass runs as in vqk_gn_forward_presummed. */
int vqk_gn_forward_presummed_parts(int dtype, const void* x, const float* w, const float* b, void* y, float* stats,
                                   const double* parts, int nblk, double* sums, int n, int64_t hw, int c, int groups, float eps,
                                   int silu, void* stream);
/* backward of y = act(GN(x)): needs x, stats, w, b and dy.  red = N*G*2+N doubles scratch with the vqk_gn_forward
 * workspace protocol (zero on entry, zero again on exit); dw/db [C] fp32 pre-zeroed (accumulated into).  accumulate != 0: dx += result; add != NULL: dx = result + add (the residual-branch
 * gradient of a ResBlock, fused instead of a separate add pass). */
int vqk_gn_backward(int dtype, const void* x, const float* stats, const float* w, const float* b, const void* dy,
                    void* dx, float* dw, float* db, double* red, int n, int64_t hw, int c, int groups, int silu,
                    int accumulate, const void* add, void* stream);
/* vqk_gn_backward / vqk_gn_backward_pooled_add (add_pooled != NULL) with the SIZE of the workspace stated.  With
 * ws_doubles >= N*G*2 + N + N*(C/32) the mid-size maps (H*W <= the GN_CLUSTER_MAX_HW slot, a multiple of 8 x 64 pixels in
 * bf16) run as ONE kernel: clusters of blocks per (sample, 32-channel slice) keep x and dy in registers, exchange their group
 * sums through the workspace (the extra N*(C/32) slots are the clusters' tickets) and apply from the registers -- x and dy are
 * read once.  Same workspace protocol (zero on entry, zero on exit); deterministic mode keeps the two-kernel form. */
int vqk_gn_backward_ws(int dtype, const void* x, const float* stats, const float* w, const float* b, const void* dy,
                       void* dx, float* dw, float* db, double* red, int64_t ws_doubles, int n, int h, int wd, int c, int groups,
                       int silu, int accumulate, const void* add, const void* add_pooled, float add_scale, void* stream);
/* The cluster form waits for its <= 8 partner blocks inside the kernel (an ordinary launch, no cooperative groups).  Forward
 * progress assumes that each XCD dispatches its share of a grid in block-index order (then the lowest waiting block's partners are
 * always resident or done -- csrc/norm.hip); the wait is BOUNDED (~0.2 s): a block that gives up is counted and falls through with
 * incomplete sums.  vqk_gn_cluster_timeouts reports the count (0 in a healthy run; synchronises the device).  A kernel killed
 * mid-way leaves sums / tickets dirty: re-zero the workspace.
 * CONCURRENCY RULE: the progress argument covers ONE cluster launch at a time per GPU.  Two of them in flight together (two streams, or
 * two processes sharing the device) can each occupy the slots the other's waiting blocks need -- measured: two ranks on one GPU, 600-870
 * timed-out blocks.  A caller that may run a second GroupNorm backward concurrently states a workspace WITHOUT the ticket region for it
 * (ws_doubles = N*G*2 + N: the two-kernel passes); the Python host gives the form to one host thread's models (ops.cluster_owner_ok);
 * processes that share a GPU set the GN_CLUSTER_MAX_HW slot to 0. */
int vqk_gn_cluster_timeouts(int* count);
/* vqk_gn_backward_ws (same-resolution addend) that also accumulates the per-channel sums of the dx it writes into
 * dx_colsum[C] (fp32): when x is the output of a conv with a bias, that is the conv's bias gradient -- no column-sum pass over
 * the gradient tensor.  Two-kernel form only; not in deterministic mode (VQK_ERR_ARG). */
int vqk_gn_backward_colsum(int dtype, const void* x, const float* stats, const float* w, const float* b, const void* dy,
                           void* dx, float* dw, float* db, double* red, int64_t ws_doubles, int n, int h, int wd, int c, int groups,
                           int silu, int accumulate, const void* add, float* dx_colsum, void* stream);
/* vqk_gn_backward with dx += add_scale * (add_pooled read at pixel (row/2, col/2)): the skip-branch gradient of a ResBlock
 * whose output went through a fused 2x2 average pool, still at half resolution ([N][h/2][wd/2][C]).  h * wd > 1024. */
int vqk_gn_backward_pooled_add(int dtype, const void* x, const float* stats, const float* w, const float* b, const void* dy,
                               void* dx, float* dw, float* db, double* red, int n, int h, int wd, int c, int groups, int silu,
                               const void* add_pooled, float add_scale, void* stream);

/* ---------------------------------------------------------------- pooling / pointwise -------
 * 2x2 stride-2 pooling with a scale: scale = 0.25 is avg_pool2d (autoencoder.py:89-91), scale = 1 is the
 * backward of the nearest x2 upsample.  x [N][H][W][C] -> y [N][H/2][W/2][C]. */
int vqk_pool2x2(int dtype, const void* x, void* y, int n, int h, int w, int c, float scale, void* stream);
/* y[N][2H][2W][C] = scale * x[N][H][W][C] replicated (backward of avg-pool with scale = 0.25). */
int vqk_unpool2x2(int dtype, const void* x, void* y, int n, int h, int w, int c, float scale, void* stream);
/* images [N][3][H][W] fp32 in [0,1] (NCHW) -> clamp, (x-0.5)/0.5, NHWC with C padded to cpad (zeros),
 * written as `dtype` and (optionally) as an fp32 NHWC-3 target for the loss.  base_autoencoder.py:31-50. */
int vqk_preprocess(const float* images, void* x_pad, int dtype, float* target, int n, int h, int w, int cpad,
                   void* stream);
/* The same with the reference's training augmentation (base_autoencoder.py:20-22: RandomResizedCrop(scale .7-1, ratio 1)
 * + RandomHorizontalFlip, per sample) fused in front: box[N][4] = {x0, y0, w, h} of the crop in source pixels (resampled
 * to H x W, bilinear, corner-aligned), flip[N] != 0 mirrors the output.  The random draws are the caller's (device
 * tensors: no host sync); the kornia generator itself is not in the reference tree, so its exact stream is unpinned. */
int vqk_augment_preprocess(const float* images, const float* box, const int32_t* flip, void* x_pad, int dtype, float* target,
                           int n, int h, int w, int cpad, void* stream);
/* loss[0] += sum (recon - target)^2 over n elements (recon dtype given; target fp32). */
int vqk_sse(int dtype, const void* recon, const float* target, int64_t n, float* loss, void* stream);
/* d = s * gscale * 2 (recon - target) * (through_tanh ? 1 - recon^2 : 1), s = *gscale_dev (NULL = 1)
 * (MSE backward, optionally through the decoder's tanh head; model.py:272, autoencoder.py:179) */
int vqk_mse_tanh_backward(int dtype, const void* recon, const float* target, int64_t n, float gscale,
                          const float* gscale_dev, int through_tanh, void* d, void* stream);
/* dx = dy * (1 - y^2) */
int vqk_tanh_backward(int dtype, const void* dy, const void* y, void* dx, int64_t n, void* stream);
/* Test-loop metrics on [0,1] NCHW fp32 images (vqvae/model.py:491-553; torchmetrics MeanSquaredError, PeakSignalNoiseRatio,
 * StructuralSimilarityIndexMeasure -- the library is not in the reference tree, its published algorithm is restated).
 * pair_stats: out5[0] += sum (pred - target)^2; out5[1] / out5[2] = running min / max of target, out5[3] / out5[4] of
 * pred (initialise to +inf / -inf).  ssim_sum: out[img] += sum over channels and the valid (h-k+1) x (w-k+1) region of the
 * SSIM map with window `window` [k*k] (k = 11 or 7), c1 = (k1 R)^2, c2 = (k2 R)^2, R = max(pred range, target range) read
 * from a pair_stats record of the same batch (`stats5`, device memory). */
int vqk_pair_stats(const float* pred, const float* target, int64_t n, float* out5, void* stream);
int vqk_ssim_sum(const float* pred, const float* target, int n, int c, int h, int w, const float* window, int ksize,
                 const float* stats5, float k1, float k2, float* out, void* stream);
/* generic elementwise: y = a*x + b*y2 (y2 optional) -- residual adds in backward */
int vqk_axpby(int dtype, const void* x, const void* y2, void* y, float a, float b, int64_t n, void* stream);

/* ---------------------------------------------------------------- optimizer ----------------
 * AdamW over a flat fp32 arena (model.py:428; torch.optim.AdamW semantics, decoupled decay).
 * seg_end[i] (exclusive, element index) / seg_wd[i]: weight decay per contiguous segment;
 * grad_scale multiplies g first (1/world after a sum all-reduce).  If shadow != NULL a bf16 copy of the
 * updated parameters is written in the same pass. */
int vqk_adamw(float* p, const float* g, float* m, float* v, int64_t n, const int64_t* seg_end, const float* seg_wd,
              int nseg, float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* shadow,
              void* stream);

/* ---------------------------------------------------------------- StyleGAN2 plugin ops ------
 * Same argument meaning as the reference's pybind functions (bias_act.cpp:32, upfirdn2d.cpp:16), with raw
 * pointers instead of tensors; NULL encodes the reference's "empty tensor".  x is contiguous NCHW fp32
 * with `inner` = elements per channel step (H*W) and `channels` = size of `dim`.
 * act: 1 linear, 3 lrelu (the reference's cuda_idx).  grad: 0 forward, 1 first-order backward. */
int vqk_bias_act(const float* x, const float* b, const float* xref, const float* yref, const float* dy, float* y,
                 int64_t numel, int64_t inner, int channels, int grad, int act, float alpha, float gain, float clamp,
                 void* stream);
int vqk_upfirdn2d(const float* x, const float* f, float* y, int n, int c, int in_h, int in_w, int fh, int fw,
                  int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip,
                  float gain, int out_h, int out_w, void* stream);

/* ---------------------------------------------------------------- VQ-GAN loss path (NHWC) ---
 * dx = dy * scale * act'(y), act' recovered from the saved OUTPUT y (bias_act.cu grad=1 semantics): 1 tanh, 2 relu,
 * 3 leaky-relu(0.2), 0 plain scaling.  `scale` carries the activation gain and the runtime weight gain. */
int vqk_act_backward(int dtype, const void* dy, const void* y, void* dx, int64_t n, int act, float scale, void* stream);
/* the same on [rows][c] tensors, plus colsum[c] += the column sums of dx (as stored): the bias gradient of the conv whose
 * output y is, without a second pass over dx (bias_act.py:196: db = dx summed over every dim but the channel).  c a whole
 * number of 16-byte vectors, at most 256 of them. */
int vqk_act_backward_colsum(int dtype, const void* dy, const void* y, void* dx, int64_t rows, int c, int act, float scale,
                            float* colsum, void* stream);
/* the same with the column sums scaled before they are ADDED to colsum (colsum += colsum_scale * sum_rows dx): the bias gradient
 * goes straight into an optimizer's gradient arena when dx carries a folded weight gain (colsum_scale = 1 / gain) */
int vqk_act_backward_colsum_scaled(int dtype, const void* dy, const void* y, void* dx, int64_t rows, int c, int act, float scale,
                                   float colsum_scale, float* colsum, void* stream);
/* upfirdn2d (upfirdn2d.cpp:16 argument meaning) on NHWC tensors of `dtype`, C a whole 16-byte chunk. */
int vqk_upfirdn2d_nhwc(int dtype, const void* x, const float* f, void* y, int n, int h, int w, int c, int fh, int fw,
                       int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip,
                       float gain, int out_h, int out_w, void* stream);
/* The backward of "relu / leaky-relu, then a 4x4 FIR blur" as ONE pass (up = down = 1; the reference runs bias_act's gradient
 * after upfirdn2d's, discriminator.py:104-120 / conv2d_resample.py:119): out = gain * act'(y_act) * upfirdn2d(x, f, pad, flip),
 * the slope taken from the sign of the activated tensor y_act [N][out_h][out_w][C] (act 2 = relu, 3 = leaky relu 0.2).
 * C a multiple of 8 sixteen-byte slots; VQK_ERR_SHAPE otherwise (callers run vqk_upfirdn2d_nhwc + vqk_act_backward). */
int vqk_upfirdn2d_act_backward(int dtype, const void* x, const float* f, const void* y_act, void* out, int n, int h, int w, int c,
                                int padx0, int padx1, int pady0, int pady1, int flip, float gain, int act, int out_h, int out_w,
                                void* stream);
/* 2x2/stride-2 max pool: backward = 0: out[N][H/2][W/2][C] = max ; backward = 1: out[N][H][W][C] = dy routed to the
 * first maximum of each window (x is the forward input). */
int vqk_maxpool2x2(int dtype, const void* x, const void* dy, void* out, int n, int h, int w, int c, int backward,
                   void* stream);
/* y[pix][c] = x[pix][c] * scale[c] + shift[c]   (LPIPS z-score, networks.py:51-52; shift may be NULL) */
int vqk_channel_affine(int dtype, const void* x, const float* scale, const float* shift, void* y, int64_t npix, int c,
                       void* stream);
/* LPIPS tap (lpips.py:31-38, utils.py:6-8): dfy == NULL: out[n] += (1/hw) sum_pix sum_c lin[c] (fx/(|fx|+1e-10) -
 * fy/(|fy|+1e-10))^2 (out pre-zeroed);  dfy != NULL: gradient w.r.t. fy for the upstream gradient gscale * gout[image]
 * (gout: n floats, one per image; NULL = 1). */
int vqk_lpips_tap(int dtype, const void* fx, const void* fy, const float* lin, int n, int64_t hw, int c, float* out,
                  const float* gout, float gscale, void* dfy, void* stream);
/* minibatch-stddev layer (discriminator.py:271-293), F = 1 feature: forward writes out[N][hw][cpad] = [x | stat | 0..]
 * and stat[N/group]; backward writes dx[N][hw][c] from dy[N][hw][cpad]. */
int vqk_mbstd(int dtype, const void* x, const void* dy, void* out, float* stat, int n, int64_t hw, int c, int cpad,
              int group, int backward, void* stream);
/* Second-order piece of the layer (R1 differentiates the backward pass): given v = cotangent of the first backward's
 * dx, writes ddy[N][hw][cpad] (cotangent of dy) and dxx[N][hw][c] (cotangent of x). */
int vqk_mbstd_double_backward(int dtype, const void* x, const void* dy, const void* v, void* ddy, void* dxx, int n,
                               int64_t hw, int c, int cpad, int group, void* stream);
/* out[0] += sum |target - recon| ;  d (+)= s * (a1 sign(recon-target) + a2 * 2 (recon-target)), s = *gscale_dev */
int vqk_l1_sum(int dtype, const void* recon, const float* target, int64_t n, float* out, void* stream);
int vqk_l1l2_backward(int dtype, const void* recon, const float* target, int64_t n, float a1, float a2,
                      const float* gscale_dev, void* d, int accumulate, void* stream);
/* loss.py:11-51 on [n] logits.  mode 0 hinge / 1 non-saturating; which 0 generator_loss(fake) / 1
 * discriminator_loss(real, fake).  loss[0] = value; dreal/dfake (optional) = gradients times *gscale_dev. */
int vqk_gan_loss(const float* logits_real, const float* logits_fake, int n, int mode, int which, float* loss,
                 float* dreal, float* dfake, const float* gscale_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VQK_H_ */
