/*
 * uclstm.h -- C ABI of libuclstm.so, the MI355X (gfx950) UNet-ConvLSTM hot path.
 *
 * The reference (dordanino12/unet-convlstm) has no FFI layer: its boundary is the
 * Python nn.Module surface of train/unet.py.  This header is what that surface binds
 * to underneath (see INTEGRATION.md for the ctypes stub): every entry point takes plain
 * device pointers, sizes and a hipStream_t (passed as void*), launches asynchronously
 * on that stream, allocates nothing, never synchronises, and returns 0 or a negative
 * UCLSTM_E_* code (the host mirror turns those into Python exceptions).
 *
 * Data layout in HBM:
 *   activations  bf16, NHWC, channel count padded to a multiple of 8 ("Cp"), pad = 0
 *   cell state   f32,  NHWC, Cp channels
 *   weights      repacked from the reference's f32 OIHW into bf16 [N][Ktot] panels,
 *                K ordered (tap, source, channel) with every (tap, source) segment padded
 *                to a multiple of 64 (uclstm_pack_weights)
 *
 * Each function cites the reference lines (relative to the reference checkout) whose
 * eager ATen calls it replaces.
 */
#ifndef UCLSTM_H
#define UCLSTM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UCLSTM_ABI_VERSION 15

#define UCLSTM_OK            0
#define UCLSTM_E_BADARG     -1   /* shape / alignment / null-pointer contract violated      */
#define UCLSTM_E_LAUNCH     -2   /* hipLaunchKernel reported an error (hipGetLastError)      */
#define UCLSTM_E_NODEVICE   -3   /* no gfx950 device visible                                 */

/* ------------------------------------------------------------------------------------ */
/* Implicit-GEMM convolution family                                                     */
/* ------------------------------------------------------------------------------------ */

/* One input tensor of a (possibly channel-concatenated) convolution:
 * bf16 [n_img][Hs][Ws][C], placed at (offY, offX) inside the output pixel frame
 * (the F.pad of train/unet.py:95-97; 0 when sizes match). */
typedef struct {
    const void* ptr;
    int32_t C;          /* padded channel count, multiple of 8 */
    int32_t Hs, Ws;
    int32_t offY, offX;
} uclstm_src;

/* One destination of the epilogue: columns [n_begin, n_end) of the GEMM go to channels
 * [c_off, c_off + n_end - n_begin) of bf16 [n_img][Hd][Wd][C] at pixel
 * (y*scale + oy, x*scale + ox); pixels falling outside are dropped.
 * The wgrad kernel READS its dY operand through the same table. */
typedef struct {
    void* ptr;
    int32_t n_begin, n_end;     /* multiples of 8 */
    int32_t C, c_off;
    int32_t Hd, Wd;
    int32_t scale, oy, ox;
} uclstm_seg;

#define UCLSTM_EPI_STORE 0   /* bias (+ per-column affine, ReLU) -> bf16 segments (+ BN partial sums) */
#define UCLSTM_EPI_LSTM  1   /* gate nonlinearities + cell update, train/unet.py:29-35                */
#define UCLSTM_EPI_ATOMIC 2  /* split-K: f32 atomic accumulation of the raw tile into acc_out (small-M GEMMs
                              * of the ConvLSTM recurrence, where 128x128 tiles alone cannot fill 256 CUs)    */

typedef struct {
    /* output pixel grid: n_img images of H x W; BatchNorm statistic groups (timesteps)
     * are `groups` equal runs of images (train/unet.py:179, :196 call BN once per t). */
    int32_t n_img, H, W, groups;
    /* tap geometry: source pixel = (y*scale + tap/ktap - pad - offY, x*scale + tap%ktap - pad - offX)
     *   3x3 pad 1 conv: ktap 3, scale 1, pad 1      1x1 conv / plain GEMM: ktap 1, scale 1, pad 0
     *   ConvTranspose2d(k2,s2) input-gradient gather: ktap 2, scale 2, pad 0 */
    int32_t ktap, scale, pad;
    int32_t nsrc;
    uclstm_src src[2];
    /* packed weights bf16 [N][Ktot], Ktot = ktap*ktap*(kseg[0]+kseg[1]), kseg[s] = roundup(src[s].C, 64) */
    const void* wp;
    int32_t N, Ktot;
    const float* bias;           /* [N] f32 or NULL */
    const float* col_scale;      /* [N] optional per-column scale  (eval-mode BatchNorm fold) */
    const float* col_shift;      /* [N] optional per-column shift  */
    int32_t relu;
    int32_t epi;
    /* UCLSTM_EPI_STORE */
    int32_t nseg;
    uclstm_seg seg[4];
    float* stats;                /* optional [groups][tiles_per_group][N][2] partial (sum, sumsq) of the STORED bf16 values */
    /* UCLSTM_EPI_LSTM: N = 64*ceil(Hd/16) gate-interleaved rows */
    int32_t Hd_p;                /* padded hidden channels, multiple of 8 */
    const float* c_prev;         /* f32 [pixels][Hd_p] or NULL (zero state, train/unet.py:23-25) */
    float* c_out;                /* f32 [pixels][Hd_p] */
    void* h_out;                 /* bf16 [pixels][Hd_p] */
    void* gates_out;             /* bf16 [pixels][4][Hd_p] post-activation i,f,g,o or NULL (inference) */
    const float* pre_add;        /* f32 [pixels][N] or NULL: pre-activations added to the GEMM result before the
                                  * nonlinearities, in panel-row order -- the x half W_x * x_t of the gate convolution when
                                  * it has been hoisted out of the recurrence (train/unet.py:55-57 recomputes it inside the
                                  * time loop; it does not depend on h) and this launch carries only W_h * h_{t-1} */
    /* UCLSTM_EPI_ATOMIC: acc_out[pixel*acc_ld + n] += tile (no bias); the caller zeroes / pre-loads acc_out.
     * acc_slab > 0: no atomics -- K range r STORES its tile into acc_out + r*acc_slab (floats); every element of each
     * of the uclstm_igemm_ksplit_used() slabs is written exactly once and the consumer adds the slabs
     * (global f32 atomics run at ~1.3 TB/s on MI355X, plain stores at ~6 TB/s). */
    float* acc_out;
    int32_t acc_ld;
    int32_t ksplit;              /* requested K ranges (>= 1) */
    int64_t acc_slab;
} uclstm_igemm_desc;

/* Rows of `stats` per group for a descriptor with N panel rows: ceil((n_img/groups)*H*W / tile_pixels),
 * tile_pixels = 256 when N <= 64 (64 x 256 block shape) else 128. */
int32_t uclstm_igemm_tiles_per_group(int32_t n_img, int32_t H, int32_t W, int32_t groups, int32_t N);

/* out = conv(src) as one MFMA implicit GEMM.  Replaces, depending on the descriptor:
 *   nn.Conv2d 3x3 of DoubleConv (train/unet.py:70-71) incl. the cat([skip, up]) of :98,
 *   its input gradient (flipped/transposed panel), ConvTranspose2d forward and input
 *   gradient (:90,:94), and with UCLSTM_EPI_LSTM the whole ConvLSTMCell.forward (:21-36:
 *   cat + conv + chunk + sigmoid/tanh + cell update in one kernel). */
int32_t uclstm_igemm_fwd(const uclstm_igemm_desc* d, void* stream);
/* Which kernel uclstm_igemm_fwd would run for this descriptor (same validation, nothing is launched): 0 = per-tap loop,
 * 128 x 128 tile; 1 = per-tap loop, 64 rows x 256 pixels (C_out <= 64); 2 = patch loop, 128 rows x 256 pixels (3x3 / pad 1 /
 * stride 1, every source on the output grid with C % 64 == 0, >= 2 channel chunks, 256 | pixels per group, patch <= 448
 * rows); 3 = the 64 -> 64-channel ring kernel.  Negative: UCLSTM_E_*.  Tests use it to assert that a parity case really
 * exercised the kernel it is meant for. */
int32_t uclstm_igemm_fwd_shape(const uclstm_igemm_desc* desc);
/* Number of non-empty K ranges uclstm_igemm_fwd will use for (Ktot, kernel width, requested ksplit): the slab count of acc_slab
 * mode.  K ranges are whole (source, 64-channel) chunks of ktap*ktap K-steps: steps per range = ktap^2 * ceil(chunks / ksplit). */
int32_t uclstm_igemm_ksplit_used(int32_t Ktot, int32_t ktap, int32_t ksplit);

/* n (1..4) INDEPENDENT descriptors as ONE launch: the per-timestep gate convolutions of the model's three ConvLSTMs
 * (bottleneck + two skip LSTMs, train/unet.py:185-191 runs them one after the other) each fill the chip for one or two rounds of
 * blocks only; as one grid -- members ordered longest block first -- their drains and fills overlap.  Every member must be a
 * descriptor uclstm_igemm_fwd would run on the patch shape (uclstm_igemm_fwd_shape == 2) with UCLSTM_EPI_LSTM or UCLSTM_EPI_ATOMIC
 * (slab mode), all with the same nsrc; anything else is UCLSTM_E_BADARG and the caller launches the members one by one.
 * Results are bit-identical to n separate uclstm_igemm_fwd launches. */
int32_t uclstm_igemm_fwd_group(const uclstm_igemm_desc* descs, int32_t n, void* stream);
/* Validation only: the block count of the launch uclstm_igemm_fwd_group would make (>= 1), or UCLSTM_E_*. */
int32_t uclstm_igemm_fwd_group_blocks(const uclstm_igemm_desc* descs, int32_t n);

/* Weight gradient  dWp[n][k] (+)= sum_pixels dY[pixel][n] * A[pixel][k]  (f32 [N][Ktot], same
 * K order as the forward panel).  dY is read through seg[] (nseg >= 1); `splits` pixel ranges
 * (0 = chosen by the library for its tile shape and the 256 CUs) accumulate with f32 atomics, so dWp
 * must be zeroed by the caller.  Replaces the autograd weight-gradient of the convolutions above. */
typedef struct {
    int32_t n_img, H, W;
    int32_t ktap, scale, pad;
    int32_t nsrc;
    uclstm_src src[2];
    int32_t N, Ktot;
    int32_t nseg;
    uclstm_seg seg[4];
    float* dwp;
    int32_t splits;
    int32_t accumulate;
    int32_t overlapped; /* != 0: the launch runs beside other GEMMs (side stream): the automatic range count (splits = 0) then
                         * minimises interference (few slabs, grid below the CU count) instead of the kernel's own duration */
    int32_t reserved_;
    int64_t slab;     /* 0: pixel ranges ADD into dwp with f32 atomics (dwp zeroed by the caller).  > 0 (>= N*Ktot): range r
                       * STORES its partial panel to dwp + r*slab (floats), nothing needs zeroing, and uclstm_unpack_wgrad adds
                       * the uclstm_igemm_wgrad_splits() slabs (float atomics ~1.3 TB/s, stores ~6 TB/s on MI355X). */
} uclstm_wgrad_desc;
int32_t uclstm_igemm_wgrad(const uclstm_wgrad_desc* d, void* stream);
/* Pixel ranges (= slabs in slab mode) the launch above will use for this descriptor; d->dwp may be NULL.  Call it with
 * splits = 0, size dwp as [result][N][Ktot], then launch with splits = result. */
int32_t uclstm_igemm_wgrad_splits(const uclstm_wgrad_desc* d);
/* Which kernel uclstm_igemm_wgrad would run for this descriptor (nothing is launched): 4 = ring-staged kernel of the 64-channel
 * full-resolution layers (slab mode only), 3 = 256 x 256 tile, 8-phase pipeline (C_out >= 256), 2 = 128 x 128 tile,
 * 1 = 64 x 256 / 64 x 192 tile (C_out <= 64), 0 = generic addressing (non-power-of-two images,
 * ConvTranspose, offset sources, kernels wider than 3x3).  Used by bench.py to price each kernel separately. */
int32_t uclstm_igemm_wgrad_shape(const uclstm_wgrad_desc* d);

/* ------------------------------------------------------------------------------------ */
/* Weight panels                                                                        */
/* ------------------------------------------------------------------------------------ */
#define UCLSTM_NMODE_IDENTITY 0   /* row n <-> entity n (valid n < n_valid)                               */
#define UCLSTM_NMODE_LSTM     1   /* gate-interleaved: n = hb*64 + gate*16 + j <-> gate*n_valid + hb*16 + j */
#define UCLSTM_NMODE_TAPMAJOR 2   /* n = tapn*n_cp + co (ConvTranspose forward), valid co < n_valid        */
#define UCLSTM_KMODE_IDENTITY 0   /* channel c of source s <-> choff[s] + c (valid c < cvalid[s])          */
#define UCLSTM_KMODE_GATES    1   /* c = gate*k_hdp + hc <-> gate*k_hd + hc (valid hc < k_hd)              */
#define UCLSTM_KMODE_IM2COL   2   /* c = tap*k_hd + ci (pre-gathered first layer), tap < k_hdp             */

typedef struct {
    int32_t N, Ktot;
    int32_t taps, nsrc;
    int32_t kseg[2], cvalid[2], choff[2];
    int32_t n_mode, n_valid, n_cp;
    int32_t k_mode, k_hdp, k_hd;
    int32_t tap_flip;
    /* element offset in the f32 source tensor = n_ent*stride_n + k_ent*stride_k
     *                                         + tap_eff*stride_tap + tapn*stride_ntap */
    int64_t stride_n, stride_k, stride_tap, stride_ntap;
} uclstm_pack_desc;

/* f32 reference layout (OIHW conv weight, train/unet.py:19,:70; [in,out,2,2] convT weight, :90)
 * -> bf16 panel [N][Ktot] (zero padded). */
int32_t uclstm_pack_weights(const uclstm_pack_desc* d, const float* w, void* wp, void* stream);
/* Batched packing: a training step repacks every panel after the optimiser step (~100 launches of 3-15 us: a latency chain the
 * step's first convolutions wait for).  Descriptors, pointers and block ranges are the same every step, so the host fills a
 * job table once (uclstm_pack_job_init, host side: returns the job's kernel family 0..4 and fills gx / nblocks / div; block0 =
 * running sum of nblocks over the jobs of ONE family), copies it to the device, and launches one kernel per family. */
typedef struct {
    uclstm_pack_desc d;
    const float* w;
    void* wp;
    int32_t block0, nblocks, gx, family;
    uint32_t div[15];
    int32_t pad_;
} uclstm_pack_job;
int32_t uclstm_pack_job_init(uclstm_pack_job* job /* HOST memory */, const uclstm_pack_desc* d, const float* w, void* wp, int32_t block0);
int32_t uclstm_pack_weights_batched(const uclstm_pack_job* jobs_dev /* DEVICE memory, jobs of one family ordered by block0 */,
                                    int32_t njobs, int32_t family, int32_t total_blocks, void* stream);

/* f32 panel gradient [N][Ktot] -> f32 gradient in the reference layout:
 * grad = (accumulate ? grad : 0) + dWp  on every valid element.  With more than 64 slabs the slabs are first folded into
 * slab 0 in place: dwp is scratch of the weight-gradient GEMM and is CONSUMED by this call. */
int32_t uclstm_unpack_wgrad(const uclstm_pack_desc* d, const float* dwp, int32_t nslab /* dWp = sum of nslab slabs */,
                            int64_t slab /* floats between slabs */, float* grad, int32_t accumulate, void* stream);
/* bias [n_valid*(4 if LSTM)] f32 -> panel-row order [N] f32 (zero padded). */
int32_t uclstm_pack_bias(const uclstm_pack_desc* d, const float* b, float* bp, void* stream);

/* ------------------------------------------------------------------------------------ */
/* BatchNorm2d + ReLU (train/unet.py:70-71), per-timestep statistics                    */
/* ------------------------------------------------------------------------------------ */
/* From the conv epilogue's partial sums build, per (group, channel): scale = gamma*rstd,
 * shift = beta - mean*scale, mean, rstd; then update running_mean/var with momentum 0.1 and
 * the unbiased variance, once per group IN ORDER (the reference calls BN once per timestep).
 * momentum < 0 means BatchNorm2d(momentum=None): cumulative average, group g uses the factor
 * 1/(-momentum + g), i.e. pass -(num_batches_tracked + 1).
 * `stats` is CONSUMED (its tile-0 slots are overwritten with mean/variance).  stats == NULL:
 * evaluation mode, scale/shift from the running statistics (groups = 1), nothing is updated. */
int32_t uclstm_bn_finalize(float* stats, int32_t groups, int32_t tiles_per_group, int32_t Cp, int32_t C,
                           int64_t count_per_group, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps,
                           float* scale, float* shift, float* mean, float* rstd, void* stream);
/* The same split in two, so that the sequential part leaves the critical path: uclstm_bn_stats_fwd is ONE launch that reduces the
 * partial sums and writes scale / shift / mean / rstd (`stats` is consumed as above); uclstm_bn_running_stats applies the in-order
 * momentum steps to the running statistics from what that launch left in `stats` -- nothing in the forward or backward pass waits
 * for it (the host layer runs it on its second stream). */
int32_t uclstm_bn_stats_fwd(float* stats, int32_t groups, int32_t tiles_per_group, int32_t Cp, int32_t C, int64_t count_per_group,
                            const float* gamma, const float* beta, float eps, float* scale, float* shift, float* mean, float* rstd,
                            void* stream);
int32_t uclstm_bn_running_stats(const float* stats, int32_t groups, int32_t tiles_per_group, int32_t Cp, int32_t C,
                                int64_t count_per_group, float* running_mean, float* running_var, float momentum, void* stream);
/* a = relu(z*scale[g] + shift[g]),  g = pixel / pixels_per_group;  z, a bf16 [pixels][Cp]. */
int32_t uclstm_bn_apply_relu(const void* z, void* a, const float* scale, const float* shift,
                             int64_t pixels, int64_t pixels_per_group, int32_t Cp, void* stream);
/* Backward pass 1: sums[g][c] = (sum g_, sum g_*xhat), g_ = dA * [scale*z+shift > 0].  Deterministic (no atomics): every
 * block writes its partial sums into `partials` ([uclstm_bn_bwd_reduce_rows()][Cp][2] floats, caller-allocated, need not
 * be initialised), a second kernel adds them in a fixed order into `sums` ([groups][Cp][2], overwritten). */
int64_t uclstm_bn_bwd_reduce_rows(int64_t pixels, int64_t pixels_per_group);
int32_t uclstm_bn_bwd_reduce(const void* z, const void* da, const float* scale, const float* shift,
                             const float* mean, const float* rstd, float* partials, float* sums,
                             int64_t pixels, int64_t pixels_per_group, int32_t Cp, void* stream);
/* Backward pass 2: dz = scale*(g_ - s1/n - xhat*s2/n)  (bf16). */
int32_t uclstm_bn_bwd_apply(const void* z, const void* da, const float* scale, const float* shift,
                            const float* mean, const float* rstd, const float* sums, void* dz,
                            int64_t pixels, int64_t pixels_per_group, int32_t Cp, void* stream);

/* MaxPool2d(2) fused into the BatchNorm stage that feeds it (the second stage of inc / down1..3; train/unet.py:81, :166-169: every
 * encoder block output goes to the next block's pooling AND to the decoder's skip connection).  H and W even.
 *   uclstm_bn_apply_relu_pool:  a = bf16(relu(z*scale + shift)) as uclstm_bn_apply_relu, and p = max over each 2x2 window of a.
 *   backward: the gradient of a is dskip (bf16, same shape as a; NULL = none) + scatter(dp) to the window's first maximum in scan
 *   order (ATen's rule, uclstm_maxpool2_bwd), rounded to bf16 as that kernel would have stored it -- formed on the fly inside the
 *   BatchNorm backward reduction / apply (same partials / sums / dz contract as uclstm_bn_bwd_reduce / _apply; `partials` has
 *   uclstm_bn_pool_bwd_rows() rows), with the arg-max recomputed from z. */
int64_t uclstm_bn_pool_bwd_rows(int64_t n_img, int32_t H, int32_t W, int32_t Cp, int32_t groups);
int32_t uclstm_bn_apply_relu_pool(const void* z, void* a, void* p, const float* scale, const float* shift, int64_t n_img, int32_t H,
                                  int32_t W, int32_t Cp, int32_t groups, void* stream);
int32_t uclstm_bn_pool_bwd_reduce(const void* z, const void* dskip, const void* dp, const float* scale, const float* shift,
                                  const float* mean, const float* rstd, float* partials, float* sums, int64_t n_img, int32_t H,
                                  int32_t W, int32_t Cp, int32_t groups, void* stream);
int32_t uclstm_bn_pool_bwd_apply(const void* z, const void* dskip, const void* dp, const float* scale, const float* shift,
                                 const float* mean, const float* rstd, const float* sums, void* dz, int64_t n_img, int32_t H,
                                 int32_t W, int32_t Cp, int32_t groups, void* stream);

/* The model's output head fused into its last BatchNorm stage (up0's second conv -> BatchNorm -> ReLU -> OutConv 1x1 with ONE
 * output channel; train/unet.py:70-71, :101-107, :196-199).  That activation feeds only the output convolution: unfused, a training
 * step makes six passes over the largest tensor of the model for it and its gradient; fused, neither exists in memory.
 *   forward:   y[p] = b + sum_c w[c] * bf16(relu(z[p][c]*scale[g][c] + shift[g][c]))            (y f32 [pixels] = NCHW with C = 1)
 *   backward:  da[p][c] = bf16(dy[p] * w[c]) computed on the fly inside the BatchNorm backward reduction / apply (same partials /
 *              sums / dz contract as uclstm_bn_bwd_reduce / _apply); dw[c] += sum_p dy[p] * a[p][c], db += sum_p dy[p].
 * Cp / 8 must be a power of two <= 64; `partials` has uclstm_bn_bwd_reduce_rows() rows. */
int32_t uclstm_bn_head_fwd(const void* z, const float* scale, const float* shift, const float* w, const float* b, float* y,
                           int64_t pixels, int64_t pixels_per_group, int32_t Cp, int32_t C, void* stream);
int32_t uclstm_bn_head_bwd_reduce(const void* z, const float* dy, const float* scale, const float* shift, const float* mean,
                                  const float* rstd, const float* w, float* partials, float* sums, float* dw, float* db,
                                  int64_t pixels, int64_t pixels_per_group, int32_t Cp, int32_t C, void* stream);
int32_t uclstm_bn_head_bwd_apply(const void* z, const float* dy, const float* scale, const float* shift, const float* mean,
                                 const float* rstd, const float* sums, const float* w, void* dz, int64_t pixels,
                                 int64_t pixels_per_group, int32_t Cp, int32_t C, void* stream);

/* Parameter gradients of the BatchNorm (train/unet.py:70) from the sums of pass 1, summed over the groups in order:
 * dbeta[c] = (accumulate ? dbeta[c] : 0) + sum_g sums[g][c][0],  dgamma likewise from sums[g][c][1],  c < C. */
int32_t uclstm_bn_bwd_param_grads(const float* sums, int32_t groups, int32_t Cp, int32_t C, float* dgamma, float* dbeta,
                                  int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------ */
/* MaxPool2d(2) (train/unet.py:81)                                                      */
/* ------------------------------------------------------------------------------------ */
int32_t uclstm_maxpool2_fwd(const void* a, void* p, int32_t n_img, int32_t H, int32_t W, int32_t Cp, void* stream);
/* da must be zero-filled by the caller when H or W is odd. First maximum in window scan order wins (ATen rule).
 * add (may be NULL; H and W even): a second gradient of the same tensor, bf16 [n_img][H][W][Cp] -- da = add + scatter(dp).
 * The UNet feeds every encoder output both to the next Down and to the decoder (train/unet.py:166-169, :188-196), so its
 * gradient is always such a sum. */
int32_t uclstm_maxpool2_bwd(const void* a, const void* dp, const void* add, void* da, int32_t n_img, int32_t H, int32_t W, int32_t Cp,
                            void* stream);

/* Finish of a split-K convolution with the STORE epilogue: out[pixel][c] = act16(relu?((sum of the nslab f32 slabs
 * pre[s*slab + pixel*ld + c] + bias[c]) * col_scale[c] + col_shift[c])), c < C (C % 8 == 0, ld >= C) -- the expression of
 * uclstm_igemm_fwd's own epilogue.  For inference convolutions on few pixels (a 32 x 32 bottleneck is 32 tiles on 256 CUs): the
 * GEMM runs as UCLSTM_EPI_ATOMIC K ranges in slab mode and this kernel applies bias / folded BatchNorm / ReLU
 * (train/unet.py:70-71 in eval mode). */
int32_t uclstm_splitk_finish(const float* pre, int32_t nslab, int64_t slab, int32_t ld, const float* bias, const float* col_scale,
                             const float* col_shift, int32_t relu, void* out, int64_t pixels, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------ */
/* ConvLSTM backward point-wise part (autograd of train/unet.py:29-35)                  */
/* ------------------------------------------------------------------------------------ */
/* dh = dh_a (+ dh_b); dc = dc_io + dh*o*(1-tanh(c)^2); dgates = pre-activation gradients
 * bf16 [pixels][4][Hd_p] (i,f,g,o); dc_io <- dc*f (gradient w.r.t. c_prev). */
int32_t uclstm_lstm_bwd_pointwise(const void* gates, const float* c_prev, const float* c_new,
                                  const void* dh_a, const void* dh_b, int32_t dh_b_is_f32 /* 0 bf16, 1 f32, 2 f32 and cleared after reading */,
                                  int32_t dh_b_nslab /* f32 only: dh_b is the sum of this many slabs */, int64_t dh_b_slab /* floats between slabs */,
                                  float* dc_io, int32_t dc_is_zero, void* dgates, int64_t pixels, int32_t Hd_p, void* stream);
/* Gate nonlinearities + cell update (train/unet.py:29-35) for the split-K form of the cell: `pre` is the f32
 * pre-activation [pixels][N] in the gate-interleaved panel-row order (N = 64*ceil(Hd/16)), bias in the same order. */
int32_t uclstm_lstm_fwd_pointwise(float* pre, int32_t nslab /* pre-activation = sum of nslab slabs (0 allowed with pre_add) */,
                                  int64_t slab /* floats between slabs */,
                                  int32_t clear /* != 0: zero what was read (atomic accumulator reused by the next step) */,
                                  const float* pre_add /* f32 [pixels][N] or NULL: added to the slabs (hoisted W_x * x_t) */,
                                  const float* bias, const float* c_prev, float* c_out, void* h_out,
                                  void* gates_out, int64_t pixels, int32_t Hd_p, void* stream);
/* n (1..4) independent cells in one launch (the model's three ConvLSTMs advance in lockstep in the forward pass,
 * uclstm_igemm_fwd_group); fields as the arguments above. */
typedef struct {
    float* pre;
    const float* pre_add;
    const float* bias;
    const float* c_prev;
    float* c_out;
    void* h_out;
    void* gates_out;
    int64_t slab;
    int64_t pixels;
    int32_t nslab, clear, Hd_p, reserved_;
} uclstm_lstm_fwd_pw_args;
int32_t uclstm_lstm_fwd_pointwise_group(const uclstm_lstm_fwd_pw_args* args, int32_t n, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Layout / boundary kernels                                                            */
/* ------------------------------------------------------------------------------------ */
/* f32 NCHW [..] -> bf16 NHWC with channel padding. Image i of the output reads image
 * (i % inner)*outer_stride + (i / inner)*inner_stride of the input (elements), which lets
 * [B,T,C,H,W] be read time-major (train/unet.py:180 indexes x_seq[:, t]). */
int32_t uclstm_nchw_to_nhwc(const float* x, void* out, int32_t n_img, int32_t C, int32_t Cp, int32_t H, int32_t W,
                            int32_t inner, int64_t inner_stride, int64_t outer_stride, void* stream);
int32_t uclstm_nhwc_to_nchw(const void* a, float* out, int32_t n_img, int32_t C, int32_t Cp, int32_t H, int32_t W, void* stream);
/* Gradient of the above (f32 NCHW -> bf16 NHWC is linear): */
int32_t uclstm_nchw_grad_to_nhwc(const float* g, void* out, int32_t n_img, int32_t C, int32_t Cp, int32_t H, int32_t W, void* stream);
/* First-layer gather: out[img][y][x][tap*C + c] = x[img'][c][y+dy-1][x+dx-1] (zero outside), Kp = padded 9*C. */
int32_t uclstm_im2col3x3_first(const float* x, void* out, int32_t n_img, int32_t C, int32_t Kp, int32_t H, int32_t W,
                               int32_t inner, int64_t inner_stride, int64_t outer_stride, void* stream);
/* f32 [pixels][Cp] <-> NCHW for the cell state at module boundaries. */
int32_t uclstm_nchw_to_nhwc_f32(const float* x, float* out, int32_t n_img, int32_t C, int32_t Cp, int32_t H, int32_t W, void* stream);
int32_t uclstm_nhwc_to_nchw_f32(const float* a, float* out, int32_t n_img, int32_t C, int32_t Cp, int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------------------------ */
/* OutConv 1x1 (train/unet.py:101-107): bf16 NHWC in, f32 NCHW out                      */
/* ------------------------------------------------------------------------------------ */
int32_t uclstm_outconv_fwd(const void* a, const float* w, const float* b, float* y,
                           int64_t n_img, int32_t HW, int32_t Cp, int32_t C, int32_t Co, void* stream);
/* da bf16 [pixels][Cp]; dw [Co][C], db [Co] accumulate with atomics (caller zeroes). */
int32_t uclstm_outconv_bwd(const void* a, const float* w, const float* dy, void* da, float* dw, float* db,
                           int64_t n_img, int32_t HW, int32_t Cp, int32_t C, int32_t Co, void* stream);

/* ------------------------------------------------------------------------------------ */
/* SpatialAttention (train/unet.py:113-125): mean & max over channels -> k x k conv (2 -> 1, no bias) -> sigmoid -> scale */
/* ------------------------------------------------------------------------------------ */
/* x, out bf16 [n_img][H][W][Cp] (C valid channels); w f32 [2][k][k] (the reference's [1,2,k,k] conv weight, k odd <= 15).
 * Outputs kept for the backward pass: att f32 [pixels] (the sigmoid map), desc f32 [pixels][2] (mean, max), argmax int32
 * [pixels] (first channel attaining the max: where torch.max routes the gradient). */
int32_t uclstm_attention_fwd(const void* x, const float* w, void* out, float* att, float* desc, int32_t* argmax, int32_t n_img,
                             int32_t H, int32_t W, int32_t Cp, int32_t C, int32_t k, void* stream);
/* dx bf16 = gradient w.r.t. x; dw f32 [2][k][k] = (dw_accumulate ? dw : 0) + weight gradient (deterministic block reductions);
 * scratch: f32 [3 * pixels + 2] work space. */
int32_t uclstm_attention_bwd(const void* x, const void* dout, const float* w, const float* att, const float* desc,
                             const int32_t* argmax, void* dx, float* dw, int32_t dw_accumulate, float* scratch, int32_t n_img,
                             int32_t H, int32_t W, int32_t Cp, int32_t C, int32_t k, void* stream);

/* column sums of a bf16 [pixels][Cp] tensor into f32 [Cp] (bias gradients); out must be zeroed. */
int32_t uclstm_colsum(const void* a, float* out, int64_t pixels, int32_t Cp, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Loss (main.py:28-72) -- weighted L1 + 0.005 * gradient L1, f32 [n][H][W] planes      */
/* ------------------------------------------------------------------------------------ */
/* sums[0..3] = sum(ad*w*m), sum(w*m), sum(gd*mc), sum(mc)  (m = 1 when mask == NULL); caller zeroes sums. */
int32_t uclstm_loss_fwd(const float* y_pred, const float* y, const float* mask, double* sums,
                        int64_t planes, int32_t H, int32_t W, void* stream);
/* grad = coefs[0] * d(sum ad*w*m)/dy_pred + coefs[1] * d(sum gd*mc)/dy_pred; coefs is a DEVICE f32[2]
 * (1/denominators times the upstream gradient), so the step never syncs with the host. */
int32_t uclstm_loss_bwd(const float* y_pred, const float* y, const float* mask, const float* coefs,
                        float* grad, int64_t planes, int32_t H, int32_t W, void* stream);

/* ------------------------------------------------------------------------------------ */
/* Optimiser (main.py:106-108): global-norm clip + AdamW on flat f32 buffers            */
/* ------------------------------------------------------------------------------------ */
int32_t uclstm_sumsq(const float* g, int64_t n, double* out /* accumulates, caller zeroes */, void* stream);
/* p,m,v,g flat f32 [n]; grad is scaled by min(1, max_norm/(sqrt(*sumsq)+1e-6)) read on device (no host sync). */
int32_t uclstm_adamw_step(float* p, float* m, float* v, const float* g, int64_t n, const double* sumsq, float max_norm,
                          float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream);

/* uclstm_adamw_step with every hyper-parameter and the step count in DEVICE memory: hyper = f32[8] {lr, beta1, beta2, eps,
 * weight_decay, max_norm (<= 0: no clipping), optimiser steps done so far, reserved}; the count is incremented on the device
 * after the update.  Nothing about the launch depends on host state, so it can be captured in a HIP graph and replayed while a
 * scheduler changes lr between replays (one small host-to-device copy). */
int32_t uclstm_adamw_step_dev(float* p, float* m, float* v, const float* g, int64_t n, const double* sumsq, float* hyper, void* stream);

/* fp16 training (the _f16 twins below): the backward pass runs on loss * scale so that fp16 activation gradients stay out of
 * the subnormal range, and g holds scale x the true gradient.  scale_state = DEVICE f32[3] {scale, growth tracker, successful
 * steps}.  uclstm_adamw_step_scaled is uclstm_adamw_step on g / scale (clip on the unscaled norm when max_norm > 0, Adam's
 * bias correction from the device-side count of successful steps) and does NOTHING when *sumsq (of the scaled gradients;
 * required) is not finite; uclstm_loss_scale_update then halves the scale (x backoff) after such a step, or counts a good
 * step and multiplies the scale by `growth` every `interval` good steps in a row.  No host synchronisation. */
int32_t uclstm_adamw_step_scaled(float* p, float* m, float* v, const float* g, int64_t n, const double* sumsq, float max_norm,
                                 float lr, float beta1, float beta2, float eps, float weight_decay, const float* scale_state,
                                 void* stream);
int32_t uclstm_loss_scale_update(float* scale_state, const double* sumsq, float growth, float backoff, int32_t interval,
                                 void* stream);

/* ------------------------------------------------------------------------------------ */
/* Data path around the step                              */
/* ------------------------------------------------------------------------------------ */
/* NPZSequenceDataset.__getitem__ (train/unet.py:273-304) for n_frames = batch*T frames on device: mask = raw x channel 0
 * > 1.1 (BEFORE scaling, :279), x = x_raw / norm_const (:283), y = 2*(asinh(clip(y_raw)/y_scale) - trans_min)/(trans_max -
 * trans_min) - 1 (:287-299).  x_raw/x: f32 [n_frames][C][HW]; y_raw/y/mask: f32 [n_frames][HW]. */
int32_t uclstm_dataset_transform(const float* x_raw, const float* y_raw, float* x, float* y, float* mask, int64_t n_frames,
                                 int32_t C, int32_t HW, float norm_const, float min_vel, float max_vel, int32_t clip,
                                 float y_scale, float trans_min, float trans_max, void* stream);
/* Epoch metric block of main.py:114-142 as running sums: sums[0..3] += sum|d|m, sum d^2 m, sum d m, sum m with
 * d = denormalize(y_pred) - denormalize(y) (train/unet.py:316-319, asinh transform); mask may be NULL. */
int32_t uclstm_metric_sums(const float* y_pred, const float* y, const float* mask, double* sums, int64_t n, float y_scale,
                           float trans_min, float trans_max, void* stream);

/* Scheduling aid, no reference counterpart: one wavefront that busy-waits `microseconds` of the constant 100 MHz wall clock on
 * `stream`.  The host layer uses it to find out whether two HIP streams really execute concurrently (streams that the
 * runtime multiplexes onto the same hardware queue serialise; which streams collide changes when e.g. an RCCL
 * communicator has created streams of its own first). */
int32_t uclstm_stream_spin(int32_t microseconds, void* stream);

/* ------------------------------------------------------------------------------------ */
/* fp16 twins (BASELINE.json configs[3]: "fp16 MFMA")                                    */
/* ------------------------------------------------------------------------------------ */
/* Every entry point above that reads or writes 16-bit activations or weight panels exists a second time with the suffix
 * _f16 and the IDENTICAL signature, where every "bf16" in the comments above reads "IEEE binary16": activations, h, gates,
 * gate gradients and panels are _Float16 and the GEMMs issue v_mfma_f32_16x16x32_f16.  Accumulators, cell state, BatchNorm
 * statistics, weight gradients and everything the optimiser touches are f32 in both families, so the remaining entry
 * points (finalize / parameter-gradient / unpack / loss / optimiser ...) are shared.  fp16 gradients need loss scaling
 * (uclstm_loss_scale_update below).  The same sources are compiled twice (csrc/common.h); this is a storage-type twin, not
 * a second backend. */
#define UCLSTM_F16_TWIN(fn) __typeof__(fn) fn##_f16;
UCLSTM_F16_TWIN(uclstm_igemm_fwd)
UCLSTM_F16_TWIN(uclstm_igemm_fwd_group)
UCLSTM_F16_TWIN(uclstm_igemm_wgrad)
UCLSTM_F16_TWIN(uclstm_pack_weights)
UCLSTM_F16_TWIN(uclstm_pack_weights_batched)
UCLSTM_F16_TWIN(uclstm_bn_apply_relu)
UCLSTM_F16_TWIN(uclstm_bn_bwd_reduce)
UCLSTM_F16_TWIN(uclstm_bn_bwd_apply)
UCLSTM_F16_TWIN(uclstm_bn_apply_relu_pool)
UCLSTM_F16_TWIN(uclstm_bn_pool_bwd_reduce)
UCLSTM_F16_TWIN(uclstm_bn_pool_bwd_apply)
UCLSTM_F16_TWIN(uclstm_bn_head_fwd)
UCLSTM_F16_TWIN(uclstm_bn_head_bwd_reduce)
UCLSTM_F16_TWIN(uclstm_bn_head_bwd_apply)
UCLSTM_F16_TWIN(uclstm_maxpool2_fwd)
UCLSTM_F16_TWIN(uclstm_maxpool2_bwd)
UCLSTM_F16_TWIN(uclstm_lstm_bwd_pointwise)
UCLSTM_F16_TWIN(uclstm_lstm_fwd_pointwise)
UCLSTM_F16_TWIN(uclstm_lstm_fwd_pointwise_group)
UCLSTM_F16_TWIN(uclstm_splitk_finish)
UCLSTM_F16_TWIN(uclstm_nchw_to_nhwc)
UCLSTM_F16_TWIN(uclstm_nhwc_to_nchw)
UCLSTM_F16_TWIN(uclstm_nchw_grad_to_nhwc)
UCLSTM_F16_TWIN(uclstm_im2col3x3_first)
UCLSTM_F16_TWIN(uclstm_outconv_fwd)
UCLSTM_F16_TWIN(uclstm_outconv_bwd)
UCLSTM_F16_TWIN(uclstm_colsum)
UCLSTM_F16_TWIN(uclstm_attention_fwd)
UCLSTM_F16_TWIN(uclstm_attention_bwd)

/* Library self-description (used by the loader to check the build). */
int32_t uclstm_abi_version(void);
const char* uclstm_build_arch(void);
/* SHA-256 (hex) of the sources the library was built from: csrc/*, this header and the compiler flags
 * (unet-convlstm_amd/build.py: source_hash).  The loader refuses a library whose hash differs from the tree's. */
const char* uclstm_source_hash(void);
/* hipGetErrorString of the most recent launch that returned UCLSTM_E_LAUNCH. */
const char* uclstm_last_error_string(void);

#ifdef __cplusplus
}
#endif
#endif /* UCLSTM_H */
