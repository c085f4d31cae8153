/*
 * ofasr.h -- C ABI of libofasr_hip.so: the MI355X (gfx950) kernels of the OFA-SR supernet hot
 * path (DynamicMBConvLayer stack + PixelShuffle upsampler).
 *
 * The reference (twice154/ofa-for-super-resolution) has NO native boundary: its hot path is
 * Python modules calling ATen.  Each entry point below replaces one ATen
 * call site of the reference; the host-side mirror of the reference's Python operator surface
 * (ofa-for-super-resolution_amd/elastic_nn/modules/dynamic_op.py, ...) binds these through
 * ctypes.  INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: device pointers + sizes, no torch types; every call is asynchronous on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream).
 *   - returns OFASR_OK (0) or a negative ofasr_status; never throws, never allocates device
 *     memory, never synchronises (the one exception is the measurement call ofasr_profile_read).
 *   - process-wide state, all of it behind mutexes / atomics: the thread-local last-error string;
 *     the per-kernel launch counters and the optional event profile (section "Diagnostics");
 *     and, used only by ofasr_mbconv_bwd, ONE side stream with its fork/join events, the list of
 *     unjoined deferred calls and the ofasr_mbconv_defer_join switch (one process drives one GPU:
 *     two host threads calling ofasr_mbconv_bwd on two streams share that side stream and its
 *     pending list -- correct, since every call orders itself by events, but not independent).
 *     Every other entry point is re-entrant: safe from several host threads on different
 *     streams, and inside hipGraph capture.
 *   - activations are NCHW-contiguous, `dtype` selects their element type (f32 / f16 / bf16);
 *     weights / filters / gradients of weights are ALWAYS fp32 (master weights), accumulation
 *     is fp32.  16-bit activation paths round once, on store.
 *   - weight SLICES are read in place: `ldw` is the row stride (in elements) of the max-size
 *     parameter, so `weight[:out, :in]` (dynamic_op.py:108) costs no `.contiguous()` copy.
 *   - workspace: caller-allocated device scratch, size from the matching *_workspace() query;
 *     contents need not be initialised.
 */
#ifndef OFASR_H
#define OFASR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* major*100 + minor: the minor number moves whenever the exported set below changes (tests/test_abi.py reads it here) */
#define OFASR_VERSION 300 /* round 3 */

typedef enum {
    OFASR_OK = 0,
    OFASR_ERR_INVALID_ARG = -1, /* null pointer, non-positive size, bad enum */
    OFASR_ERR_UNSUPPORTED = -2, /* shape/dtype outside what the kernels implement */
    OFASR_ERR_WORKSPACE = -3,   /* workspace missing or too small */
    OFASR_ERR_LAUNCH = -4       /* hipLaunchKernel reported an error */
} ofasr_status;

typedef enum { OFASR_F32 = 0, OFASR_F16 = 1, OFASR_BF16 = 2 } ofasr_dtype;

int ofasr_version(void);
/* message of the last failing call made by THIS host thread ("" if none) */
const char* ofasr_last_error_string(void);
const char* ofasr_status_string(int status);

/* ---------------------------------------------------------------------------------------------
 * PixelShuffle / PixelUnshuffle  -- replaces nn.PixelShuffle(2) (reference ofa/utils.py:309-310,
 * used by ConvLayer act 'pixelshuffle', ofa_mbs4.py:120) and pixel_unshuffle's one-hot strided
 * conv (ofa/utils.py:383-397).  Pure byte permutation, bit-exact, elem_size in {1,2,4,8}.
 *   shuffle:   x [N, C*r*r, H, W] -> y [N, C, H*r, W*r],  y[n,c,h*r+i,w*r+j] = x[n,c*r*r+i*r+j,h,w]
 *   unshuffle: x [N, C, H*r, W*r] -> y [N, C*r*r, H, W]   (inverse map)
 * Each is the other's gradient map.
 * ------------------------------------------------------------------------------------------- */
int ofasr_pixel_shuffle(const void* x, void* y, int64_t N, int64_t C, int64_t H, int64_t W,
                        int r, int elem_size, void* stream);
int ofasr_pixel_unshuffle(const void* x, void* y, int64_t N, int64_t C, int64_t H, int64_t W,
                          int r, int elem_size, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Elastic-kernel filter  -- replaces DynamicSeparableConv2d.get_active_filter
 * (reference ofa/elastic_nn/modules/dynamic_op.py:46-71 + sub_filter_start_end,
 * ofa/imagenet_codebase/utils/__init__.py:89-94): centre crop of the max-size depthwise weight,
 * optionally passed through the learned '%dto%d_matrix' chain (F.linear => f . M^T).
 *   w_max  [Cmax, kmax, kmax] fp32 (rows c < C are used)
 *   ks     HOST array ks[0] > ks[1] > ... > ks[nsteps]; ks[0] = kmax, ks[nsteps] = active K
 *   mats   HOST array of nsteps DEVICE pointers, mats[s] = [ks[s+1]^2, ks[s+1]^2] fp32
 *   transform 0: KERNEL_TRANSFORM_MODE None -> plain centre crop (mats may be NULL)
 *   f      [C, K, K] fp32 (out)
 * kmax <= 9, nsteps <= 3.
 * bwd: df [C,K,K] -> dw_max [Cmax,kmax,kmax] rows c<C FULLY written (zeros outside the crop
 *      window; rows >= C untouched -- caller pre-zeroes them), dmats[s] written for every
 *      walked step (host array of device pointers; ignored when transform == 0).
 * ------------------------------------------------------------------------------------------- */
int ofasr_ktransform_fwd(const float* w_max, const int* ks, int nsteps, const float* const* mats,
                         int transform, float* f, int64_t C, void* stream);
size_t ofasr_ktransform_bwd_workspace(const int* ks, int nsteps, int64_t C);
int ofasr_ktransform_bwd(const float* w_max, const int* ks, int nsteps, const float* const* mats,
                         int transform, const float* df, float* dw_max, float* const* dmats,
                         int64_t C, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Depthwise KxK convolution, stride 1, dilation 1, zero padding K/2  -- replaces the
 * F.conv2d(groups=C) of DynamicSeparableConv2d.forward (dynamic_op.py:73-84) and its autograd.
 *   x, y, dy, dx [N, C, H, W] (`dtype`);  f, df [C, K, K] fp32;  K in {1,3,5,7}
 * ------------------------------------------------------------------------------------------- */
int ofasr_dwconv_fwd(const void* x, const float* f, void* y, int64_t N, int64_t C, int64_t H,
                     int64_t W, int K, int dtype, void* stream);
int ofasr_dwconv_dgrad(const void* dy, const float* f, void* dx, int64_t N, int64_t C, int64_t H,
                       int64_t W, int K, int dtype, void* stream);
size_t ofasr_dwconv_wgrad_workspace(int64_t N, int64_t C, int64_t H, int64_t W, int K);
int ofasr_dwconv_wgrad(const void* dy, const void* x, float* df, int64_t N, int64_t C, int64_t H,
                       int64_t W, int K, int dtype, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Pointwise (1x1) convolution on an in-place weight slice (MFMA)  -- replaces
 * DynamicPointConv2d.forward (dynamic_op.py:104-112: weight[:out,:in].contiguous() + F.conv2d)
 * and its autograd.
 *   x  [N, Cin, HW]   y [N, Cout, HW]   (`dtype`)
 *   w  fp32, element (co, ci) at w[co*ldw + ci]  (co < Cout, ci < Cin)
 *   fwd:   y[n,co,p]  = sum_ci w[co,ci] * x[n,ci,p]
 *   dgrad: dx[n,ci,p] = sum_co w[co,ci] * dy[n,co,p]
 *   wgrad: dw[co*ldw+ci] = sum_{n,p} dy[n,co,p] * x[n,ci,p]   (only the slice is written)
 * ------------------------------------------------------------------------------------------- */
int ofasr_pwconv_fwd(const void* x, const float* w, int64_t ldw, void* y, int64_t N, int64_t Cin,
                     int64_t Cout, int64_t HW, int dtype, void* stream);
int ofasr_pwconv_dgrad(const void* dy, const float* w, int64_t ldw, void* dx, int64_t N,
                       int64_t Cin, int64_t Cout, int64_t HW, int dtype, void* stream);
size_t ofasr_pwconv_wgrad_workspace(int64_t N, int64_t Cin, int64_t Cout, int64_t HW);
int ofasr_pwconv_wgrad(const void* dy, const void* x, float* dw, int64_t ldw, int64_t N,
                       int64_t Cin, int64_t Cout, int64_t HW, int dtype, void* workspace,
                       size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Sliced BatchNorm2d fused with its activation and the residual add  -- replaces, per call site of the
 * MB block, F.batch_norm on [:C] slices (DynamicBatchNorm2d.bn_forward, dynamic_op.py:148-167) + the
 * in-place ReLU6 (dynamic_layers.py:44,56) + the shortcut add (proxyless_nets.py:50), forward and
 * backward.  x, residual, y, dy, dx: [N, C, HW] (`dtype`); all per-channel vectors fp32 of length >= C.
 *
 *   bn_stats     per-channel (sum, sum of squares) partials of x into `workspace`
 *   bn_finalize  training=1: batch mean / biased variance from the partials, running stats updated in
 *                place (EMA with `momentum`, unbiased variance); training=0: running stats are used.
 *                Emits mean, invstd, scale = gamma*invstd, shift = beta - mean*scale.
 *   bn_act_fwd   y = act((x-mean)*scale + beta (+ residual)), beta = shift + mean*scale;  act: 0 none, 1 ReLU6
 *   bn_act_bwd   dz = dy masked by the open ReLU6 window (recomputed from x); dgamma = sum dz*xhat,
 *                dbeta = sum dz; training=1: dx = scale*(dz - dbeta/M - xhat*dgamma/M), training=0:
 *                dx = scale*dz; dresidual (optional, may be NULL) = dz.
 * ------------------------------------------------------------------------------------------- */
size_t ofasr_bn_workspace(int64_t N, int64_t C);
int ofasr_bn_partials(int64_t N, int64_t C); /* number of partial slabs ofasr_bn_stats writes */
int ofasr_bn_stats(const void* x, int64_t N, int64_t C, int64_t HW, int dtype, void* workspace,
                   size_t workspace_bytes, void* stream);
int ofasr_bn_finalize(const void* workspace, int64_t n_partials, int64_t C, double count, const float* gamma,
                      const float* beta, float* running_mean, float* running_var, double momentum, double eps,
                      int training, float* mean, float* invstd, float* scale, float* shift, void* stream);
int ofasr_bn_act_fwd(const void* x, const void* residual, void* y, const float* scale, const float* shift,
                     const float* mean, int64_t N, int64_t C, int64_t HW, int act, int dtype, void* stream);
/* statistics pass (training only) + apply pass that folds the finalize in: `stats` [4*C] receives
 * mean | invstd | scale | shift (kept for backward); workspace >= ofasr_bn_workspace(N, C). */
int ofasr_bn_fwd(const void* x, const void* residual, void* y, const float* gamma, const float* beta,
                 float* running_mean, float* running_var, double momentum, double eps, int training, float* stats,
                 int64_t N, int64_t C, int64_t HW, int act, int dtype, void* workspace, size_t workspace_bytes,
                 void* stream);
size_t ofasr_bn_act_bwd_workspace(int64_t N, int64_t C);
int ofasr_bn_act_bwd(const void* dy, const void* x, const void* residual, void* dx, void* dresidual,
                     const float* scale, const float* shift, const float* mean, const float* invstd,
                     float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t HW, int act, int training,
                     int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Composite: one DynamicMBConvLayer (+ identity shortcut) per call  -- DynamicMBConvLayer.forward
 * (reference ofa/elastic_nn/modules/dynamic_layers.py:70-84) + MobileInvertedResidualBlock.forward
 * (ofa/imagenet_codebase/networks/proxyless_nets.py:44-51) and their autograd, as ONE host call that
 * enqueues the kernels above in order (expand 1x1 -> BN+ReLU6 -> kernel transform -> depthwise -> BN+ReLU6 ->
 * project 1x1 -> BN (+ x)).  Exists to keep the host (Python) cost per block at one FFI call: at the
 * MB stack's sizes the step is otherwise bound by ~100 Python-side launches per block, not by the GPU.
 *
 * act_buf  (activation dtype, 16-byte aligned)  ofasr_mbconv_act_elems(d) elements, kept for backward:
 *          [y1 | y2] 2*N*mid*HW then [y3 | out] 2*N*Cout*HW when the block takes the fused 16-bit path (BN + ReLU6 applied
 *          in the consumers' loads: the activated tensors are never written and get no room), otherwise
 *          [y1 | a1 | y2 | a2] 4*N*mid*HW then [y3 | out].  y = pre-BN conv outputs, a = activated tensors; `out`, the
 *          block output, is always the last N*Cout*HW elements.  The choice is a function of the descriptor alone.
 * stat_buf (fp32) per BN i in {expand, depthwise, project}: mean | invstd | scale | shift (4*C_i, C = mid, mid,
 *          Cout), then the active depthwise filter f [mid*K*K].
 * bwd: tmp_buf (activation dtype) 3*N*mid*HW + N*Cout*HW elements of scratch; every gradient tensor is
 *          FULLY written (dense max-size parameter gradients, zeros outside the active slice).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int64_t N, Cin, mid, Cout, H, W;
    int K;            /* active depthwise kernel size */
    int ks[4];        /* kernel sizes walked: ks[0] = kmax > ... > ks[chain_len-1] = K (as for ofasr_ktransform_*) */
    int chain_len;    /* number of valid entries in ks (1 when K == kmax) */
    int transform;    /* 1: apply mats[s] along the chain (KERNEL_TRANSFORM_MODE set and K < kmax); 0: plain crop */
    int dtype;        /* ofasr_dtype of the activations */
    int residual;     /* add x to the output (identity shortcut) */
    int bn_training[3];
    double bn_momentum[3];
    double bn_eps[3];
    int64_t Cmid_max, Cout_max;      /* row counts of the max-size parameters */
    int64_t ldw1, ldw2;              /* row strides of the 1x1 weights */
    const float* w1;                 /* [Cmid_max, ldw1] expand */
    const float* w2;                 /* [Cout_max, ldw2] project */
    const float* wdw_max;            /* [Cmid_max, kmax, kmax] */
    const float* mats[3];
    const float* gamma[3];
    const float* beta[3];
    float* running_mean[3];
    float* running_var[3];
    int64_t* num_batches_tracked[3]; /* incremented when bn_training[i] (may be NULL) */
} ofasr_mbconv_desc;

typedef struct {
    float* dw1;        /* [Cmid_max, ldw1] */
    float* dw2;        /* [Cout_max, ldw2] */
    float* dwdw_max;   /* [Cmid_max, kmax, kmax] */
    float* dmats[3];   /* gradients of the walked matrices (NULL for the others) */
    float* dgamma[3];  /* lengths Cmid_max, Cmid_max, Cout_max */
    float* dbeta[3];
} ofasr_mbconv_grads;

size_t ofasr_mbconv_workspace(const ofasr_mbconv_desc* d);
size_t ofasr_mbconv_stat_floats(const ofasr_mbconv_desc* d);
size_t ofasr_mbconv_act_elems(const ofasr_mbconv_desc* d);
int ofasr_mbconv_fwd(const ofasr_mbconv_desc* d, const void* x, void* act_buf, float* stat_buf, void* workspace,
                     size_t workspace_bytes, void* stream);
int ofasr_mbconv_bwd(const ofasr_mbconv_desc* d, const void* x, const void* act_buf, const float* stat_buf,
                     const void* dout, void* dx, void* tmp_buf, const ofasr_mbconv_grads* g, void* workspace,
                     size_t workspace_bytes, void* stream);
/* The MB stack -- all active blocks of OFAMobileNetS4.forward's stage loop (reference ofa_mbs4.py:147-151) in one call per
 * direction.  items[i] carries block i's descriptor and buffers exactly as ofasr_mbconv_fwd / _bwd take them; block i
 * reads the output of block i - 1 (the tail of its act_buf), and in the backward block i's dout is items[i + 1].dx
 * (items[n - 1] takes `dout`, items[0].dx is the gradient of x).  dx buffers are read on the caller's stream only, so
 * two alternating buffers suffice.  tmp_buf, grads and dx are unused (may be NULL) in the forward. */
typedef struct {
    const ofasr_mbconv_desc* desc;
    void* act_buf;
    float* stat_buf;
    void* workspace;
    size_t workspace_bytes;
    void* tmp_buf;                      /* backward */
    const ofasr_mbconv_grads* grads;    /* backward */
    void* dx;                           /* backward: N*Cin*HW elements */
} ofasr_mbstack_item;
int ofasr_mbstack_fwd(const ofasr_mbstack_item* items, int n, const void* x, void* stream);
int ofasr_mbstack_bwd(const ofasr_mbstack_item* items, int n, const void* x, const void* dout, void* stream);
/* ofasr_mbconv_bwd runs the weight-gradient kernels on a library-owned side stream beside the input-gradient chain
 * and, by default, ends by ordering them before whatever the caller enqueues next on `stream`.
 * ofasr_mbconv_defer_join(1) (process-wide; returns the previous setting) drops that per-call join: on return dx,
 * dgamma[] and dbeta[] are final in stream order, while dw1, dw2, dwdw_max and dmats[] are final only after
 * ofasr_mbconv_join(stream).  Until that join the caller must keep every buffer passed to the deferred calls valid
 * and must not read those four gradients; a later call whose tmp_buf / workspace overlaps an unjoined call's waits for
 * it.  This mirrors how the reference's autograd consumes them (torch AccumulateGrad at the end of backward(), read by
 * the optimizer step, progressive_shrinking.py:199-203).  Not for use inside hipGraph capture (the side stream must
 * re-join before a capture ends).  ofasr_mbconv_join is a no-op when nothing is pending. */
int ofasr_mbconv_defer_join(int enable);
int ofasr_mbconv_join(void* stream);
/* The library's side stream (a hipStream_t; NULL when OFASR_MBCONV_SIDE_STREAM=0), for callers that enqueue further
 * independent work beside the composite calls -- the host mirror puts the static convs' weight gradients there.  Such
 * work is ordered by the caller (events); ofasr_mbconv_join does not know about it. */
void* ofasr_side_stream(void);

/* ---------------------------------------------------------------------------------------------
 * Fused MB block, eval-mode / frozen BatchNorm, forward only (csrc/mbfused.hip): with bn_training[] all 0 the three
 * BNs are affine maps (running statistics) and fold into their convolutions, so the whole block
 *     out = x + BN3(W2 . relu6(BN2(dw_k(relu6(BN1(W1 . x))))))         (residual = 0: without the x +)
 * is ONE kernel that reads x once and writes out once; the mid tensor never leaves the CU.  The regime of
 * SRRunManager.validate / eval_ofa_net_sr.py (reference sr_run_manager.py:323-393, net.eval()) and of the frozen-BN
 * teacher's forward (:417-420).  Supported: f16 / bf16 activations, Cin = Cout = 64, mid % 32 == 0, K in {3,5,7}, any
 * N, H, W (ofasr_mbconv_infer_supported; otherwise OFASR_ERR_UNSUPPORTED and the caller uses ofasr_mbconv_fwd).
 * The descriptor is ofasr_mbconv_fwd's (running statistics are only read).  x and out must not overlap.
 * Two steps, so that a serving loop prepares once per set of weights:
 *   ofasr_mbconv_infer_prepare  kernel transform + BN folding -> the 16-bit operand images (a function of the weights,
 *                               BN tensors, K and mid only; ofasr_mbconv_infer_operand_bytes(d) bytes, kept by the caller)
 *   ofasr_mbconv_infer_run      the block on x with prepared operands; scratch (ofasr_mbconv_infer_scratch_bytes(d),
 *                               0 for launches with at least half as many tiles as the GPU has CUs) holds the partial
 *                               projections when a small launch spreads a tile's mid channels over several workgroups
 * ofasr_mbconv_infer = prepare + run on one workspace of ofasr_mbconv_infer_workspace(d) bytes.
 * ------------------------------------------------------------------------------------------- */
int ofasr_mbconv_infer_supported(const ofasr_mbconv_desc* d);
size_t ofasr_mbconv_infer_workspace(const ofasr_mbconv_desc* d);
size_t ofasr_mbconv_infer_operand_bytes(const ofasr_mbconv_desc* d);
size_t ofasr_mbconv_infer_scratch_bytes(const ofasr_mbconv_desc* d);
int ofasr_mbconv_infer_prepare(const ofasr_mbconv_desc* d, void* operands, size_t operand_bytes, void* stream);
int ofasr_mbconv_infer_run(const ofasr_mbconv_desc* d, const void* x, void* out, const void* operands,
                           size_t operand_bytes, void* scratch, size_t scratch_bytes, void* stream);
int ofasr_mbconv_infer(const ofasr_mbconv_desc* d, const void* x, void* out, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense KxK convolution (K in {3,5}, stride 1, zero padding K/2, no bias) of the static ConvLayers as an
 * implicit GEMM on the matrix cores  -- replaces nn.Conv2d in ConvLayer (reference ofa/layers.py:131-151) for
 * 16-bit activations: forward, input gradient and weight gradient.
 *   x [N,Cin,H,W], y [N,Cout,H,W] (f16 / bf16), w / dw [Cout,Cin,K,K] fp32; needs W % 8 == 0.
 * workspace: fwd / dgrad: the per-call 16-bit weight image, ofasr_conv2d_workspace(Cin, Cout, K, dgrad) bytes;
 *            wgrad: the split-K partial slabs, ofasr_conv2d_wgrad_workspace(N, Cin, Cout, H, W, K) bytes.
 * ofasr_conv2d_wgrad writes every element of dw (no accumulation into it); the summation order is fixed.
 * Returns OFASR_ERR_UNSUPPORTED for fp32 (use ofasr_conv2d_f32_*), other K and W % 8 != 0 (the host mirror zero-pads
 * ragged widths on the right, which is the convolution's own padding, and drops the extra output columns).
 * ------------------------------------------------------------------------------------------- */
size_t ofasr_conv2d_workspace(int64_t Cin, int64_t Cout, int K, int dgrad);
/* Training form of ConvLayer's conv -> BatchNorm: the forward also leaves the BatchNorm statistics partials of what it
 * stores ([Cout][units] (sum, sum of squares) in fp32, units = ofasr_conv2d_stat_units(...)), so no pass reads the conv
 * output just for statistics; fold them with ofasr_bn_fwd_cp (apply in the same call) or ofasr_bn_finalize_cp
 * (+ ofasr_pixel_shuffle2_bn for the decoder stages: BN apply and PixelShuffle(2) in one pass). */
int ofasr_conv2d_stat_units(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int K);
int ofasr_conv2d_fwd_stat(const void* x, const float* w, void* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                          int64_t W, int K, int dtype, void* partial, int64_t units, void* workspace,
                          size_t workspace_bytes, void* stream);
int ofasr_bn_fwd_cp(const void* x, const void* residual, void* y, const void* partial, int64_t P, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, double momentum, double eps, int training,
                    float* stats, int64_t N, int64_t C, int64_t HW, int act, int dtype, void* stream);
int ofasr_bn_finalize_cp(const void* partial, int64_t P, int64_t C, double count, const float* gamma, const float* beta,
                         float* running_mean, float* running_var, double momentum, double eps, int training, float* stats,
                         void* stream);
int ofasr_pixel_shuffle2_bn(const void* x, void* y, const float* stats, int64_t N, int64_t C, int64_t H, int64_t W,
                            int dtype, void* stream);
/* ... and its backward: dout arrives in the SHUFFLED layout [N, C/4, 2H, 2W]; both BatchNorm-backward passes read it
 * through the inverse shuffle (no un-shuffle pass).  dx [N, C, H, W], dgamma / dbeta [C]; act none, no residual. */
size_t ofasr_bn_bwd_ps2_workspace(int64_t N, int64_t C);
int ofasr_bn_bwd_ps2(const void* dout, const void* x, void* dx, const float* scale, const float* mean, const float* invstd,
                     float* dgamma, float* dbeta, int64_t N, int64_t C, int64_t H, int64_t W, int training, int dtype,
                     void* workspace, size_t workspace_bytes, void* stream);
/* Inference form of a whole ConvLayer (reference ofa/layers.py:120-151 in eval mode; the decoder's conv -> BN ->
 * PixelShuffle(2) stages, ofa_mbs4.py:111-123): y = act(BN_eval(conv(x))) as ONE kernel -- the BatchNorm's affine map
 * (scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale) is applied to the fp32 accumulators
 * and the result rounded once; act: 0 none, 1 ReLU6, 2 PixelShuffle(2) done by the store (y is [N, Cout/4, 2H, 2W]).
 *   ofasr_conv2d_infer_prepare  16-bit weight image + scale | shift -> operands (ofasr_conv2d_infer_operand_bytes; a
 *                               function of the weights and BN tensors only, kept by the caller); gamma == NULL: no BN
 *   ofasr_conv2d_infer_run      the conv on x with prepared operands; W % 8 == 0 as for ofasr_conv2d_fwd */
size_t ofasr_conv2d_infer_operand_bytes(int64_t Cin, int64_t Cout, int K);
int ofasr_conv2d_infer_prepare(const float* w, const float* gamma, const float* beta, const float* running_mean,
                               const float* running_var, double eps, int64_t Cin, int64_t Cout, int K, int dtype,
                               void* operands, size_t operand_bytes, void* stream);
int ofasr_conv2d_infer_run(const void* x, void* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int K,
                           int dtype, int act, const void* operands, size_t operand_bytes, void* stream);
int ofasr_conv2d_fwd(const void* x, const float* w, void* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                     int K, int dtype, void* workspace, size_t workspace_bytes, void* stream);
int ofasr_conv2d_dgrad(const void* dy, const float* w, void* dx, int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                       int64_t W, int K, int dtype, void* workspace, size_t workspace_bytes, void* stream);
size_t ofasr_conv2d_wgrad_workspace(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int K);
int ofasr_conv2d_wgrad(const void* dy, const void* x, float* dw, int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                       int64_t W, int K, int dtype, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The same dense KxK convolution with fp32 activations (the reference's own arithmetic), on the fp32-input matrix
 * instruction (csrc/conv2d_f32.hip): exact fp32 fma chains, any H / W, no alignment requirement.  x, y, dy, dx fp32
 * NCHW; w / dw [Cout,Cin,K,K] fp32.  Workspaces: the per-call weight image (fwd / dgrad) and the split-K partial slabs
 * (wgrad) from the matching queries; the wgrad summation order is fixed.
 * ------------------------------------------------------------------------------------------- */
size_t ofasr_conv2d_f32_workspace(int64_t Cin, int64_t Cout, int K, int dgrad);
int ofasr_conv2d_f32_fwd(const void* x, const float* w, void* y, int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W,
                         int K, void* workspace, size_t workspace_bytes, void* stream);
int ofasr_conv2d_f32_dgrad(const void* dy, const float* w, void* dx, int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                           int64_t W, int K, void* workspace, size_t workspace_bytes, void* stream);
size_t ofasr_conv2d_f32_wgrad_workspace(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W, int K);
int ofasr_conv2d_f32_wgrad(const void* dy, const void* x, float* dw, int64_t N, int64_t Cin, int64_t Cout, int64_t H,
                           int64_t W, int K, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * PIL-exact 8-bit bicubic resize  -- replaces the host-side img.resize(size, Image.BICUBIC) that makes the
 * reference's LR images (Scale(1/2), Scale(1/4): ofa/imagenet_codebase/data_providers/div2k_setxx.py:354-380, called
 * per sample at :288-298).  src / dst are uint8 PLANES ([planes, in_h, in_w] -> [planes, out_h, out_w], planes = N*3
 * for CHW images); the result equals Pillow's fixed-point resampling bit for bit (horizontal pass, then vertical;
 * 22-bit coefficients computed on the device in double).  Down-scale factors up to 8; workspace from the query.
 * ------------------------------------------------------------------------------------------- */
size_t ofasr_bicubic_resize_u8_workspace(int64_t planes, int64_t in_h, int64_t in_w, int64_t out_h, int64_t out_w);
int ofasr_bicubic_resize_u8(const void* src, void* dst, int64_t planes, int64_t in_h, int64_t in_w, int64_t out_h,
                            int64_t out_w, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Diagnostics (tests and bench.py; nothing on the product path calls these).
 *   Every kernel launch of the library is counted per kernel symbol (template arguments resolved, e.g.
 *   "dw_mfma_kernel<ofasr::bf16_t, 7, false, true, true>"): ofasr_debug_launch_count(substr) sums the counters of
 *   the symbols containing `substr` (NULL / "" = all) since the last ofasr_debug_reset_launch_counts(), so a parity
 *   test can assert WHICH kernel variant served a call; ofasr_debug_launch_table() lists "count<TAB>symbol" lines.
 *   ofasr_profile_enable(1) brackets every following launch with two events recorded on the stream the kernel is
 *   launched on (the caller's or the library's side stream); ofasr_profile_read() waits for them -- the library's only
 *   synchronising call -- and returns "symbol<TAB>launches<TAB>total_us<TAB>algorithmic_bytes<TAB>flops" lines for the
 *   launches bracketed since the previous read (bytes / flops per DESIGN.md section 3; 0 where not annotated).
 *   Returned strings stay valid until the next call of the same function.
 *   ofasr_debug_mbfused_tile(w) forces the tile width of ofasr_mbconv_infer's kernel (16, 32 or 64; 0 = choose per
 *   image size, the default; also settable at load time by OFASR_MBFUSED_TILE) and returns the previous setting;
 *   ofasr_debug_mbfused_split(0) keeps launches with fewer tiles than CUs from spreading a tile's mid-channel chunks
 *   over several workgroups (default 1; OFASR_MBFUSED_SPLIT=0 at load time), returns the previous setting.
 * ------------------------------------------------------------------------------------------- */
int ofasr_debug_mbfused_tile(int width);
int ofasr_debug_mbfused_split(int enable);
/*   ofasr_debug_mbconv_bn_bwd_stat(1): ofasr_mbconv_bwd lets the project input gradient take the BN2-backward sums of
 *   what it writes instead of running the reduction pass (default 0: measured 1 % slower in the training step;
 *   OFASR_MBCONV_BN_BWD_STAT=1 at load time); returns the previous setting. */
int ofasr_debug_mbconv_bn_bwd_stat(int enable);
long long ofasr_debug_launch_count(const char* substr);
void ofasr_debug_reset_launch_counts(void);
const char* ofasr_debug_launch_table(void);
int ofasr_profile_enable(int on);
const char* ofasr_profile_read(void);

#ifdef __cplusplus
}
#endif
#endif /* OFASR_H */
