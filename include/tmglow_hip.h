/* tmglow_hip.h -- C ABI of libtmglow_hip.so: the gfx950 (MI355X) kernels behind the TM-Glow
 * invertible hot path.  Plain pointers and sizes only: device pointers are owned by the caller
 * (PyTorch's caching allocator in the shipped host code), nothing here allocates or synchronises,
 * every launch goes to the HIP stream passed in.  Return value: 0 on success, a positive
 * hipError_t if the launch failed, a negative value for a rejected argument.
 *
 * The reference (zabaras/deep-turbulence, pure Python / PyTorch) has no FFI of its own; each
 * entry point below replaces the stock torch-op call sites cited next to it (paths relative to
 * /root/reference/tmglow/nn/).  INTEGRATION.md shows the ctypes binding the host side uses.
 *
 * Conventions
 *   - activations: fp32, NHWC.  A tensor argument is (pointer, d = {pixel stride in floats,
 *     channel offset}) so a channel slice of a wider buffer is addressed in place.
 *   - "segments": a list of (pointer, {stride, offset, nchannels}) that together stand for the
 *     channel concatenation torch.cat would build (flowAffine.py:74, convLSTM.py:72,151,
 *     denseBlock.py:152); at most TMG_MAX_IN_SEG inputs / TMG_MAX_OUT_SEG outputs.
 *   - `dims` arrays are host int64; their layout is documented per function.
 */
#ifndef TMGLOW_HIP_H
#define TMGLOW_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef TMG_MAX_IN_SEG
#define TMG_MAX_IN_SEG 3
#define TMG_MAX_OUT_SEG 3
#endif

typedef struct ihipStream_t* tmg_stream_t; /* == hipStream_t */

/* ---- dense contractions on the fp32 matrix cores (tmg_conv.hip) -------------------------------- */

/* Re-layout torch weights W[Cout][Cin][k][k] into the MFMA operand order
 * wpk[k*k][Kpad/16][Npad][16].  mode 0: forward operand (K=Cin, N=Cout).  mode 1: input-gradient
 * operand (K=Cout, N=Cin, taps flipped).  wpk must hold k*k*Kpad*Npad floats (Kpad, Npad = K, N
 * rounded up to 16).  cin_eff >= Cin builds the operand for cin_eff input channels (extra ones zero). */
int tmg_conv_pack(const void* w, void* wpk, int64_t Cout, int64_t Cin, int64_t cin_eff, int64_t ksize, int64_t mode,
                  tmg_stream_t st);

/* tmg_conv_pack with an input-channel map {cvalid, csplit, cgap}: operand channel c < cvalid reads source channel
 * c (+ cgap when c >= csplit), channels in [cvalid, cin_eff) are zero.  Lets a conv over the segments (x1 | D) use the
 * matching rows of a weight stored for cat(x1, cond, d1, d2) without re-laying it out on the host. */
int tmg_conv_pack_map(const void* w, void* wpk, int64_t Cout, int64_t Cin, int64_t cin_eff, int64_t ksize, int64_t mode,
                      const int64_t* map, tmg_stream_t st);

/* out = [relu]( (conv_k(pad(act(in)); wpk) + bias) * exp(clamp(kappa,-4,ln4)) ), ksize 1 or 3,
 * stride 1 or 2, zero or replicate padding, act = optional per-channel affine then optional ReLU.
 * Replaces F.conv2d at flowUtils.py:246-247 (Conv2dZeros), convLSTM.py:74 and :152, glowConv.py:194
 * and :220 (1x1), tmGlow.py:88-95,148-156,180-182 (encoder) and, with mode-1 packed weights, the
 * autograd input-gradient of each.
 * dims = {B,Hin,Win,Hout,Wout,ksize,stride,Cin,Cout,relu_in,pad_replicate,relu_out,accumulate} */
int tmg_conv_fwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* wpk, const void* bias,
                 const void* kappa, const void* in_scale, const void* in_shift, void* const* out_ptrs,
                 const int64_t* out_desc, int64_t nout, const int64_t* dims, tmg_stream_t st);

/* tmg_conv_fwd with an extra tensor `add` ({stride, off}, Cout channels) summed before bias and scale:
 * out = [relu]((conv + add + bias) * exp(clamp(kappa))).  Used to inject the conditioning map's pre-computed
 * contribution to a coupling network (flowAffine.py:74: cat(x1, cond) is linear in the conv). */
int tmg_conv_fwd_add(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* wpk, const void* bias,
                     const void* kappa, const void* in_scale, const void* in_shift, const void* add, const int64_t* add_desc,
                     void* const* out_ptrs, const int64_t* out_desc, int64_t nout, const int64_t* dims, tmg_stream_t st);

/* The same 3x3 / stride-1 convolution for contractions with many output channels (Cout >= 64: the ConvLSTM gate conv,
 * convLSTM.py:72-74, and the level-wide conditioning contraction) as Winograd F(2x2, 3x3): 2.25x fewer matrix-core operations,
 * fp32 throughout.  tmg_conv_wino_pack builds the operand U = G g G^T, [16][Cin_pad/16][Cout_pad][16] floats, from the torch-layout
 * weight [Cout][Cin][3][3]; tmg_conv_wino_fwd computes out = conv(pad(act(cat(in)))) + bias.
 * in_desc / out_desc = {stride, off, n} per segment (<= 3 each), dims = {B,H,W,Cin,Cout,relu_in,pad_replicate}.
 * Returns -100 (nothing launched) outside its envelope: float4-addressable operands, Cin % 4 == 0, Cout % 4 == 0, Cout >= 64. */
int tmg_conv_wino_pack(const void* w, void* U, int64_t Cout, int64_t Cin, int64_t mode, int64_t nvalid, tmg_stream_t st);
int tmg_conv_wino_fwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* U, const void* bias,
                      void* const* out_ptrs, const int64_t* out_desc, int64_t nout, const int64_t* dims, tmg_stream_t st);
/* OPT-IN variant of the same wide contraction (round 5): the 16 position GEMMs on the bf16 matrix pipe at fp32 accuracy - every fp32
 * operand split exactly into three bf16 parts (truncation), six of the nine part products (v_mfma_f32_16x16x32_bf16, fp32 accumulate),
 * both Winograd transforms in fp32 before the split.  tmg_conv_wino_pack3: U = G g G^T split, [16][Kpad32/32][Npad/16][3][64][8] bf16
 * (16 * Kpad32 * Npad * 3 two-byte values; modes as tmg_conv_wino_pack); tmg_conv_wino_fwd3: tmg_conv_wino_fwd's arguments and envelope.
 * Replaces nothing of the reference by itself: the same F.conv2d call sites (convLSTM.py:72-74); selected by
 * tmg_ops.set_winograd_precision("bf16x3") / TMG_WINO_BF3=1, the default stays the fp32 MFMA kernel. */
int tmg_conv_wino_pack3(const void* w, void* U, int64_t Cout, int64_t Cin, int64_t mode, int64_t nvalid, tmg_stream_t st);
int tmg_conv_wino_fwd3(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* U, const void* bias,
                       void* const* out_ptrs, const int64_t* out_desc, int64_t nout, const int64_t* dims, tmg_stream_t st);
/* tmg_conv_wino_pack: mode 0 = forward operand (K = Cin, N = Cout); mode 1 = operand of the input gradient w.r.t. the first
 * `nvalid` input channels (0: all), K = Cout, N = nvalid, taps flipped - what autograd's conv2d backward contracts with.
 * tmg_conv_wino_narrow: the same Winograd contraction for FEW output channels (Cout <= 48, Cin >= 64): the input gradients of the
 * wide contractions and the ConvLSTM block's narrow convs (convLSTM.py:150-152).  Up to 3 output segments (out_desc = {stride,
 * off, n} each); dims = {B,H,W,Cin,Cout,relu_in,pad_replicate,relu_out}.  -100 outside the envelope. */
int tmg_conv_wino_narrow(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* U, const void* bias,
                         void* const* out_ptrs, const int64_t* out_desc, int64_t nout, const int64_t* dims, tmg_stream_t st);

/* The large 3x3 / stride-1 weight gradients (>= 32 input and output channels) as Winograd F(3x3, 2x2): dW[Cout][cin_dst][3][3] +=
 * sum_pixels act(in)(p + tap) (x) dy(p), dbias += sum dy.  dims = {B,H,W,Cin,Cout,relu_in,pad_replicate,cin_dst,cin_valid,ci_split,
 * ci_off0,ci_off1} (destination mapping as tmg_conv_wgrad), dy_desc = {stride, off}; ws = scratch of at least
 * tmg_conv_wino_wgrad_ws_floats(dims) floats (0: shape not eligible).  Returns -100 outside the envelope (nothing launched). */
int64_t tmg_conv_wino_wgrad_ws_floats(const int64_t* dims);
int tmg_conv_wino_wgrad(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* dy, const int64_t* dy_desc,
                        void* dW, void* dbias, void* ws, int64_t ws_floats, const int64_t* dims, tmg_stream_t st);
/* `ngroups` identically shaped Winograd weight gradients in one launch (the per-layer zero-conv weight gradients of a wide flow level):
 * gtab = device int64 table [ngroups][4][4], rows 0-2 = {pointer, pixel stride, channel offset, channels} of the group's input segments;
 * group g reads dy channels [g * gdims[0], + Cout), accumulates into dW + g * gdims[1] floats and dbias + g * gdims[2] floats. */
int64_t tmg_conv_wino_wgrad_grouped_ws_floats(const int64_t* dims, int64_t ngroups);
int tmg_conv_wino_wgrad_grouped(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* gtab, int64_t ngroups,
                                const int64_t* gdims, const void* dy, const int64_t* dy_desc, void* dW, void* dbias, void* ws,
                                int64_t ws_floats, const int64_t* dims, tmg_stream_t st);

/* dW[Cout][Cin][k][k] += scale * sum_pixels act(in)(p*s+tap) (x) dy(p) ; dbias += scale * sum dy.
 * (accumulating: caller zero-fills; per-block partial sums go through the scratch `ws` and a reduce kernel.)  Replaces the autograd weight-gradient of the convs above.
 * dims = {B,Hin,Win,Hout,Wout,ksize,stride,Cin,Cout,relu_in,pad_replicate,cin_dst,cin_valid,ci_split,ci_off0,ci_off1};
 * dy_desc = {stride, off}; dW rows have cin_dst entries, source channel ci < cin_valid lands at
 * ci + (ci < ci_split ? ci_off0 : ci_off1) (all zero: dense [Cout][Cin][k*k]) */
int tmg_conv_wgrad(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* in_scale,
                   const void* in_shift, const void* dy, const int64_t* dy_desc, void* dW, void* dbias, const void* kappa,
                   void* ws, int64_t ws_floats, const int64_t* dims, tmg_stream_t st);
/* Scratch (in floats) the slab path of tmg_conv_wgrad wants for `dims`; with ws == NULL the kernel falls back to
 * direct float atomics on dW. */
int64_t tmg_conv_wgrad_ws_floats(const int64_t* dims);

/* Adds the contribution of the replicate-padded ring to the border pixels of a 3x3 input gradient
 * (adjoint of F.pad(mode='replicate'), flowUtils.py:246).  dims = {B,H,W,Cdy,Cx}; w = the conv's mode-1 packed weights. */
int tmg_conv_rep_border_fix(const void* dy, const int64_t* dy_desc, const void* w, const void* kappa, void* const* out_ptrs,
                            const int64_t* out_desc, int64_t nout, const int64_t* dims, tmg_stream_t st);

/* Direct input gradient for strided convs (encoder stride-2 convs, tmGlow.py:154-156,180-182).
 * dims = {B,Hin,Win,Hout,Wout,Cin,Cout,ksize,stride,accumulate} */
int tmg_conv_dgrad_direct(const void* dy, const int64_t* dy_desc, const void* w, void* dx, const int64_t* dx_desc,
                          const int64_t* dims, tmg_stream_t st);

/* Optional per-launch HIP-event timing on the launch stream (bench.py's roofline / bandwidth lines).  enable(1): time the
 * matrix-core kernels; enable(2): also the bandwidth-bound kernel classes; enable(100 + k): only kernel id k; enable(0): stop.  Enabling clears the records; the
 * event pool is created on the first enable, never inside a timed region.
 * collect: out[kid*3+{0,1,2}] = {launches, total ms, total algorithmic work}: flops for the matrix-core kernels, HBM bytes for
 * the classes whose tmg_prof_name starts with "hbm:"; returns #kernel ids. */
int tmg_prof_enable(int64_t on);
int tmg_prof_collect(double* out, int64_t nk);

/* ---- bandwidth-bound kernels (tmg_pointwise.hip) ------------------------------------------------ */

/* Affine coupling apply + per-sample log-det (flowAffine.py:76-83 forward, :102-109 reverse).
 * hh = coupling-network output, channels interleaved (shift, r).  logdet[b] += sum 2*softsign(r).
 * dims = {B, pixels per image, C/2, reverse} */
int tmg_affine_apply(const void* hh, const int64_t* hh_d, const void* x2, const int64_t* x_d, void* y2, const int64_t* y_d,
                     void* rsave, void* logdet, const int64_t* dims, tmg_stream_t st);
/* As tmg_affine_apply; also copies the Ch pass-through channels x1 -> y1 (x1_d / y1_d = {pixel stride, 0}) in the same pass
 * (the reference's cat(x1, x2') at flowAffine.py:83,109); x1 == NULL: no copy. */
int tmg_affine_apply_pass(const void* hh, const int64_t* hh_d, const void* x2, const int64_t* x_d, void* y2, const int64_t* y_d,
                          void* rsave, void* logdet, const void* x1, const int64_t* x1_d, void* y1, const int64_t* y1_d,
                          const int64_t* dims, tmg_stream_t st);
int tmg_affine_bwd(const void* gout, const int64_t* go_d, const void* yref, const int64_t* yr_d, const void* rsave,
                   const void* g, void* gin, const int64_t* gi_d, void* dhh, const int64_t* dh_d, const int64_t* dims,
                   tmg_stream_t st);

/* tmg_affine_bwd with dhh pre-multiplied by exp(clamp(*kappa)) (the Conv2dZeros output scale, flowUtils.py:247). */
int tmg_affine_bwd_scaled(const void* gout, const int64_t* go_d, const void* yref, const int64_t* yr_d, const void* rsave,
                          const void* g, void* gin, const int64_t* gi_d, void* dhh, const int64_t* dh_d, const void* kappa,
                          const int64_t* dims, tmg_stream_t st);

/* ConvLSTM gates (convLSTM.py:76-83): gates [npix][4R] pre-activation in order i,f,o,g.  The forward call only READS them
 * (c_next, h_next out); the backward call takes the same pre-activation tensor as `acts`, evaluates the gate activations on it
 * again and overwrites it in place with the pre-activation gradients; dh, dc_in and dc_prev may each be null (no gradient arriving /
 * none wanted for the previous cell state).  dims = {npix, R} */
int tmg_lstm_pointwise_fwd(void* gates, const void* c_prev, const int64_t* cprev_d, void* c_next, void* h_next,
                           const int64_t* dims, tmg_stream_t st);
int tmg_lstm_pointwise_bwd(void* acts, const void* c_prev, const int64_t* cprev_d, const void* c_next, const void* dh,
                           const void* dc_in, void* dc_prev, const int64_t* dims, tmg_stream_t st);

/* Diagonal Gaussian prior (flowUtils.py:176-209,274-275,307-334; top prior tmGlow.py:399-412).
 * dims = {B, pixels per image, Ch, mode(0: log-prob of z2 [+eps out], 1: sample from eps), clip_mean}
 * fl   = {mean_lo, mean_hi, logstd_lo, logstd_hi} */
int tmg_gauss_fwd(const void* hz, const int64_t* hz_d, const void* zin, const int64_t* zi_d, void* zout, const int64_t* zo_d,
                  void* logp, const int64_t* dims, const float* fl, tmg_stream_t st);
int tmg_gauss_bwd(const void* hz, const int64_t* hz_d, const void* zin, const int64_t* zi_d, const void* dzin,
                  const int64_t* dzi_d, const void* g, void* dzout, const int64_t* dzo_d, void* dhz, const int64_t* dh_d,
                  const int64_t* dims, const float* fl, tmg_stream_t st);

/* Checker squeeze / un-squeeze (flowUtils.py:114-122,137-145). dims = {B,h,w,C,to_small} */
int tmg_checker(const void* src, const int64_t* s_d, void* dst, const int64_t* d_d, const int64_t* dims, tmg_stream_t st);

/* Zero-padded channel halves for fields whose channel half is not a multiple of 4 (the reference's 3-channel data sets; the
 * halves are the chunk(2, 1) of flowAffine.py:73 / :98): compact [npix][2 ch] <-> [x1 | 0.. | x2 | 0..] with ch + pad channels
 * per half; to_padded writes the padding zeros itself.  dims = {npix, ch, pad, to_padded}; s_d / d_d = {pixel stride, offset} */
int tmg_pad_halves(const void* src, const int64_t* s_d, void* dst, const int64_t* d_d, const int64_t* dims, tmg_stream_t st);

/* n host int64 values -> device memory through kernel arguments (256 per launch): the pointer tables of the grouped launches
 * when the stream is being captured into a hipGraph (a pageable host-to-device copy cannot be recorded; a kernel node keeps its
 * arguments).  No reference counterpart: the reference has no captured training window (trainFlowParallel.py:256-297 is eager). */
int tmg_fill_i64(void* dst, const int64_t* vals, int64_t n, tmg_stream_t st);

/* Bilinear align_corners=True resize (misc.py:34-35) and its adjoint. dims = {B,hi,wi,ho,wo,C} */
int tmg_upsample_fwd(const void* src, void* dst, const int64_t* dims, tmg_stream_t st);
int tmg_upsample_bwd(const void* dout, void* din, const int64_t* dims, tmg_stream_t st);

/* Per-channel sums over pixels: BatchNorm batch moments (mode 0) and backward sums (mode 1)
 * (nn.BatchNorm2d at denseBlock.py:49).  dims = {npix, C, mode, divisor}: mode 0 subtracts v0[c]/divisor before summing
 * (divisor 0: v0 as is) - the centred second pass takes the first pass's sums directly.  tmg_bn_bwd_apply: dims = {npix, C,
 * accumulate, divisor} with m0, m1 divided by divisor. */
int tmg_chan_reduce(const void* x, const int64_t* x_d, const void* g, const int64_t* g_d, const void* v0, const void* v1,
                    const void* v2, const void* v3, void* s0, void* s1, const int64_t* dims, tmg_stream_t st);
int tmg_bn_bwd_apply(const void* x, const int64_t* x_d, const void* g, const int64_t* g_d, const void* a, const void* bsh,
                     const void* mean, const void* rstd, const void* gamma, const void* m0, const void* m1, void* dx,
                     const int64_t* dx_d, const int64_t* dims, tmg_stream_t st);
/* Batch sums -> mean, biased var, rstd, a = gamma*rstd, bsh = beta - mean*a (out[5][C]) and the in-place momentum update of the
 * running statistics (unbiased variance), one launch.  dims = {C, n}; fl = {eps, momentum}; rmean / rvar may be NULL. */
int tmg_bn_finalize(const void* sum, const void* csq, const void* gamma, const void* beta, void* rmean, void* rvar, void* out,
                    const int64_t* dims, const float* fl, tmg_stream_t st);
/* BatchNorm batch moments in one pass (denseBlock.py:49 in training mode): per-channel sum and sum of squares of an NHWC tensor /
 * channel-slice view accumulated in fp64 into acc (double [2][C], zeroed by the caller; dims = {pixels, C}, x_d = {pixel stride, offset}),
 * and the finalize step on those moments (arguments as tmg_bn_finalize; dims = {C, n, address of the module's int64
 * num_batches_tracked device scalar or 0: incremented by one in the same launch}). */
int tmg_chan_moments(const void* x, const int64_t* x_d, void* acc, const int64_t* dims, tmg_stream_t st);
int tmg_bn_finalize64(const void* acc, const void* gamma, const void* beta, void* rmean, void* rvar, void* out, const int64_t* dims,
                      const float* fl, tmg_stream_t st);

/* dst (+)= src * [ref > 0] + add  over n channels: ReLU-mask / concat adjoints. dims = {npix,n,accumulate} */
int tmg_masked_add(const void* src, const int64_t* s_d, const void* ref, const int64_t* r_d, const void* add,
                   const int64_t* a_d, void* dst, const int64_t* d_d, const int64_t* dims, tmg_stream_t st);

/* Both growth-1 layers of a coupling network in one launch: d1 = conv(relu(t0); w1) + add1, d2 = conv(relu(cat(t0, d1)); w2) +
 * add2, out = (d1, d2, 0, 0) per pixel (denseBlock.py:135-152 with two layers of growth 1).  dims = {B,H,W,Cin,relu_in,w_rows,
 * w_split,w_gap,w2_d1_row}: input channel c reads weight row c (+w_gap if c >= w_split) when c < w_rows; the d1 channel of the
 * second layer reads row w2_d1_row of w2. */
int tmg_c1x2_fwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* w1, const void* w2, const void* add1,
                 const int64_t* add1_d, const void* add2, const int64_t* add2_d, void* out, const int64_t* out_d, const int64_t* dims,
                 tmg_stream_t st);
/* Growth-1 dense layer of the coupling network, C_out = 1 (denseBlock.py:135-138), forward and
 * backward (input gradient accumulated into g segments, weight gradient accumulated atomically).
 * dims = {B,H,W,Cin,relu_in,w_rows,fill4,w_split,w_gap} */
int tmg_c1_fwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* w, void* out, const int64_t* out_d,
               const int64_t* dims, tmg_stream_t st);
/* tmg_c1_fwd plus a per-pixel scalar `add` ({stride, off}) summed onto the result. */
int tmg_c1_fwd_add(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* w, const void* add,
                   const int64_t* add_d, void* out, const int64_t* out_d, const int64_t* dims, tmg_stream_t st);
int tmg_c1_bwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* w, void* dW, const void* dd,
               const int64_t* dd_d, const void* dref, const int64_t* dref_d, void* const* g_ptrs, const int64_t* g_desc,
               int64_t ng, const int64_t* dims, tmg_stream_t st);

/* Fused backward of both growth-1 layers of a coupling network (denseBlock.py:135-152 x2) incl. the ReLU masks
 * and the concat adjoint: one pass over the network input.  in segments = nn inputs followed by the 4-channel D
 * buffer; dims = {B,H,W,Cin_total(incl. D),cin_nn,rows1,rows2,dd1_out,dd2_out,dd_stride,split2,gap2,dd_quad} (dd*_out: optional device
 * pointers, passed as integers, receiving the masked gradients w.r.t. d1 / d2; dd_quad = 1: they are channels 0, 1 of a 16-byte aligned
 * float4 slot per pixel and the kernel stores (dd1, dd2, 0, 0) whole - the level-wide stash then needs no zero fill). */
int tmg_dense2_bwd(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* w1, const void* w2, void* dW1,
                   void* dW2, const void* GD, int64_t gd_stride, const void* Dp, int64_t d_stride, const void* const* g0_ptrs,
                   const int64_t* g0_desc, void* const* out_ptrs, const int64_t* out_desc, int64_t ng, const void* add0,
                   int64_t add0_stride, const int64_t* dims, tmg_stream_t st);

/* d(kappa) of a Conv2dZeros from its parameter gradients: <W,dW> + <b,db>, zero outside the clamp range
 * (flowUtils.py:247). */
int tmg_dkappa(const void* w, const void* dw, int64_t nw, const void* b, const void* db, int64_t nb, const void* kappa, void* dk,
               tmg_stream_t st);

/* `nbatch` equally shaped weight tensors (w + b*Cout*Cin*k*k floats) packed in one launch into wpk + b*taps*Kpad*Npad
 * floats; map = {cvalid, csplit, cgap} as tmg_conv_pack_map. */
int tmg_conv_pack_batched(const void* w, void* wpk, int64_t nbatch, int64_t Cout, int64_t Cin, int64_t cin_eff, int64_t ksize,
                          int64_t mode, const int64_t* map, tmg_stream_t st);

/* njobs <= 48 independent packing jobs (different tensors, shapes, modes) in one launch: w[i] / wpk[i] = source / destination of job
 * i, jobs = njobs x {Cout, Cin, cin_eff, ksize, mode, cvalid, csplit, cgap} as tmg_conv_pack_map (identity map: cvalid = Cin,
 * csplit = INT_MAX, cgap = 0).  The forward and input-gradient operands of all layers of a dense block (denseBlock.py:69-100). */
int tmg_conv_pack_many(const void* const* w, void* const* wpk, const int64_t* jobs, int64_t njobs, tmg_stream_t st);

/* `ngroups` identically shaped weight-gradient contractions in one launch (the per-layer coupling convolutions of a flow
 * level, flowAffine.py:49-55 under autograd: 15 small launches per level otherwise).  in_ptrs / in_desc / dims as
 * tmg_conv_wgrad, describing group 0; gtab: DEVICE int64 table [ngroups][4][4]: rows 0-2 = {pointer, pixel stride, channel
 * offset, channels} of the group's input segments, row 3 = {dy pointer, dy pixel stride, 0, 0} or zeros (shared dy); gdims = {dy channel offset between groups, dW floats between groups, dbias
 * floats between groups}.  -100: this shape cannot be grouped, issue per-group tmg_conv_wgrad calls instead. */
int tmg_conv_wgrad_grouped(const void* const* in_ptrs, const int64_t* in_desc, int64_t nseg, const void* gtab, int64_t ngroups,
                           const int64_t* gdims, const void* dy, const int64_t* dy_desc, void* dW, void* dbias, void* ws,
                           int64_t ws_floats, const int64_t* dims, tmg_stream_t st);
int64_t tmg_conv_wgrad_grouped_ws_floats(const int64_t* dims, int64_t ngroups);

/* ---- fused affine coupling layer (tmg_coupling.hip) ---------------------------------------------------------------- */

/* The traffic-heavy part of one affine coupling layer in ONE launch, for the narrow levels (8 <= C <= 32, C/2 a multiple of 4):
 *   Conv2dZeros over relu(x1 | d1, d2) on the matrix cores (flowUtils.py:246-247, replicate padding) -> affine coupling +
 *   per-sample log-det (flowAffine.py:76-83 forward, :102-109 reverse) -> optional trailing ActNorm + invertible 1x1 mix
 *   out = Wm [x1; y2] + bm (glowConv.py:207-222 + actNorm.py:71-85, the reverse direction's order); Wm == NULL: out = [x1 | y2].
 * D [npix][4] = raw (d1, d2, 0, 0) from tmg_c1x2_fwd.  The conditioning map's share of the zero conv is passed pre-computed
 * (hc: C channels, before bias and scale; a conv is linear in its input channels and every layer of a level sees the same map).
 * Saved for the backward pass: rsave [npix][C/2] (softsign arguments), y2save [npix][C/2] (transformed half; may be NULL).
 * logdet[b] is accumulated.
 * dims = {B, H, W, C, reverse, x pixel stride, out pixel stride, hc pixel stride, row length of wz, column of d1 in wz};
 * wz in torch layout [C][rows][3][3].  Returns -100 when the shape is outside the kernel's envelope: use the per-op entry
 * points instead. */
int tmg_coupling_fwd(const void* x, void* out, void* rsave, void* y2save, const void* D, const void* hc, const void* wz,
                     const void* bz, const void* kappa, const void* Wm, const void* bm, void* logdet, const int64_t* dims,
                     tmg_stream_t st);

/* Backward of tmg_coupling_fwd's generative-direction layer up to the coupling network's input gradients, one launch:
 *   dto = Wm^T dout (input gradient of the trailing mix, glowConv.py:207-222 under autograd) -> affine-coupling backward
 *   (flowAffine.py:102-109) -> e^kappa dhh written to DH (C channels) -> input gradient of Conv2dZeros w.r.t. (x1 | d1, d2) with the
 *   exact adjoint of its replicate padding (flowUtils.py:246-247): G0 [npix][C/2], GD [npix][4] = (d d1, d d2, 0, 0).
 * dtin (C channels): second half = gradient w.r.t. the transformed input half, first half = dto1 (the pass-through gradient;
 * tmg_dense2_bwd adds the coupling network's share on top).  g: [B] gradient arriving on the log-det (may be NULL).
 * dims = {B, H, W, C, dout pixel stride, x pixel stride, DH pixel stride, dtin pixel stride, row length of wz, column of d1 in wz}.
 * Returns -100 outside the envelope (8 <= C <= 32, C/2 a multiple of 4). */
int tmg_coupling_bwd(const void* dout, const void* x, const void* r, const void* g, const void* Wm, const void* wz, const void* kappa,
                     void* DH, void* dtin, void* G0, void* GD, const int64_t* dims, tmg_stream_t st);

/* tmg_coupling_fwd / tmg_coupling_bwd with the two channel halves of every [npix][C] activation addressed separately (the halves
 * are the chunk(2, 1) of flowAffine.py:73 / :98; the narrow flow levels keep them in tensors of their own so that the kernels
 * reading x1 alone fetch whole cache lines of what they use).  The entry points above are these with half 2 = half 1 + C/2.
 * fwd dims = {B,H,W,C,reverse, x1 stride, out1 stride, hc stride, wz row length, d1 column, x2 stride, out2 stride}
 * bwd dims = {B,H,W,C, dout1 stride, x2 stride, DH stride, dtin1 stride, wz row length, d1 column, dout2 stride, dtin2 stride, fwd};
 * x2 = the second half of the layer input (the only part of it the backward pass reads).  fwd = 1: backward of the DENSITY
 * direction's layer (flowAffine.py:76-83: mix first, coupling second): dout is the gradient w.r.t. the coupling output (no mix
 * in front; Wm is not read), x2 = the second half of the coupling OUTPUT, dtin = the gradient w.r.t. the coupling input. */
int tmg_coupling_fwd_halves(const void* x1, const void* x2, void* out1, void* out2, void* rsave, void* y2save, const void* D,
                            const void* hc, const void* wz, const void* bz, const void* kappa, const void* Wm, const void* bm,
                            void* logdet, const int64_t* dims, tmg_stream_t st);
int tmg_coupling_bwd_halves(const void* dout1, const void* dout2, const void* x2, const void* r, const void* g, const void* Wm,
                            const void* wz, const void* kappa, void* DH, void* dtin1, void* dtin2, void* G0, void* GD,
                            const int64_t* dims, tmg_stream_t st);

/* ---- reduced-precision 1x1 channel mix (tmg_mix16.hip) ---------------------------------------------------------- */

/* y = fp16(W) . fp16(x) + bias per pixel with fp32 accumulation on v_mfma_f32_16x16x16_f16: the "fp16 MFMA 1x1 conv" variant
 * of BASELINE.json configs[4] for F.conv2d(x, W[C,C,1,1]) at glowConv.py:193-194 / :219-220 (activations stay fp32 in HBM and
 * are rounded to fp16 in registers).  Opt-in; the default mix is tmg_conv_fwd with ksize 1 (fp32 MFMA).
 * x_d / y_d = {pixel stride, channel offset}; W = fp32 [C][C] row-major; dims = {npix, C, transposed}: transposed != 0 applies
 * W^T (the input gradient of the same mix).  C % 4 == 0, C <= 256. */
int tmg_mix_f16(const void* x, const int64_t* x_d, const void* W, const void* bias, void* y, const int64_t* y_d, const int64_t* dims,
                tmg_stream_t st);
/* Inverses of K folded channel mixes [K][C][C] (and binv = -Winv b) in FP64, rounded once: the recompute-from-output backward of the
 * plain coupling layers rebuilds a layer's input from its output, out = Wm [x1; y2] + bm (glowConv.py:207-222 with actNorm.py:71-85
 * folded in; the reference's own forward-direction matrix, glowConv.py:164-174, is NOT this inverse).  dims = {K, C}; C <= 64. */
int tmg_mat_inverse(const void* W, const void* b, void* Winv, void* binv, const int64_t* dims, tmg_stream_t st);
/* The trainer's optimizer step (main.py:78: Adam, weight decay 1e-8, amsgrad) for all parameters in one launch.  tab: device int64 [n][5] =
 * pointers (param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq); chunks: device int32 [nchunks][3] = (tensor, first element, elements <= 4096);
 * dims = {nchunks, amsgrad}; fl = {lr, beta1, beta2, eps, weight_decay, 1 - beta1^t, sqrt(1 - beta2^t), 1 - beta1, 1 - beta2}.  Arithmetic and order of
 * torch.optim.Adam's single-tensor form. */
int tmg_adam_step(const void* tab, const void* chunks, const int64_t* dims, const float* fl, tmg_stream_t st);

/* Parameter-side folding of one flow level: ActNorm (actNorm.py:66-83) + PLU-parameterised invertible 1x1 conv (glowConv.py:151-161) of all
 * K layers -> mix matrices Wm [K,C,C], biases bm [K,C], the unfolded W [K,C,C] as FP64 scratch (kept for the backward) and the scalar
 * log-det ld of all K mixes; and the backward of that map (dl, du zero outside their triangular masks).  C <= 256.  The fold is evaluated in
 * fp64 and rounded once on store: a rounding error in a mix matrix is coherent over every pixel it is applied to.  tab: device int64 [K][5] = pointers to the layers'
 * own l, u, log_s, ActNorm weight, ActNorm bias tensors (the last two null: no ActNorm); sign_s [K][C]; perm / iperm: int32 [K][C], the row
 * permutation of P and its inverse.  dims = {K, C, reverse}; fl = {sign of the log_s term of the log-det, pixels per image}. */
int tmg_lu_fold_fwd(const void* tab, const void* sign_s, const void* perm, const void* iperm, void* W, void* Wm, void* bm, void* ld,
                    const int64_t* dims, const float* fl, tmg_stream_t st);
int tmg_lu_fold_bwd(const void* tab, const void* sign_s, const void* perm, const void* iperm, const void* W, const void* dWm, const void* dbm,
                    const void* dld, void* dl, void* du, void* dlogs, void* da, void* db, const int64_t* dims, const float* fl, tmg_stream_t st);
/* The same with the LAST layer's upstream gradients in tensors of their own (dWm_tail [C,C], dbm_tail [C]; dWm / dbm then hold layers
 * 0..K-2): the level-fused coupling node consumes the first K-1 mixes as one slice, the ConvLSTM layer the last one
 * (flowLSTMBlock.py:137-160), and autograd would otherwise zero-fill and add two full-size gradients per level. */
int tmg_lu_fold_bwd_split(const void* tab, const void* sign_s, const void* perm, const void* iperm, const void* W, const void* dWm,
                          const void* dbm, const void* dWm_tail, const void* dbm_tail, const void* dld, void* dl, void* du, void* dlogs,
                          void* da, void* db, const int64_t* dims, const float* fl, tmg_stream_t st);

/* Grouped 3x3 weight gradient with FOUR output channels per group (the growth-1 convs of the coupling networks, denseBlock.py:18-36, all
 * layers of a level in one launch) on v_mfma_f32_4x4x1 blocks - a (tap, input-channel quad) pair per block, one pixel per instruction -
 * instead of 16x16 tiles that would be 2/16 used.  gtab: device int64 [G][16] as for tmg_conv_wgrad_grouped; seg_channels[nseg]: channels of
 * the input segments; dy: shared upstream gradient, group g at channels [c g, c g + c), c = dims[5] = 4 or 2 (2: the compact stash of
 * (dd1, dd2) per layer - rows 2, 3 of a group's dW stay untouched), pixel stride dy_stride; dW [G][4][Cin][3][3] is accumulated into
 * (atomics).  dims = {B, H, W, Cin, relu_in, dy channels per group}; zero padding, stride 1.  -100: shape outside the envelope (Cin not
 * in {12, 20, 36, 68} or unaligned segments), nothing launched. */
int tmg_conv_wgrad_thin_grouped(const void* gtab, int64_t G, const int64_t* seg_channels, int64_t nseg, const void* dy, int64_t dy_stride,
                                void* dW, const int64_t* dims, tmg_stream_t st);

/* Grouped weight gradient of the 1x1 channel mixes (glowConv.py:193-222 + actNorm.py:71-85 under autograd), all layers of a level in
 * one launch: dW[g][o][i] += sum_px dout_g[px][o] y_g[px][i], db[g][o] += sum_px dout_g[px][o] - a streaming GEMM over the pixels with
 * both operands read from global memory in MFMA fragment order.  gtab: device int64 [G][16] as for tmg_conv_wgrad_grouped, every
 * group with its own dout (entries 12 / 13: pointer, pixel stride); tensors pixel-linear NHWC.  dims = {npix, C}; C in {16, 32},
 * otherwise -100.  dW [G][C][C], db [G][C] (nullable) are accumulated into. */
int tmg_mix_wgrad_grouped(const void* gtab, int64_t G, void* dW, void* db, const int64_t* dims, tmg_stream_t st);

/* [npix][CP] -> [CP/2][npix][2]: the level-wide conditioning addends of the growth-1 convs (channel 2k / 2k+1 = coupling layer k,
 * flowAffine.py:73-75 with the conditioning part of the dense block's input split off) as one pixel-contiguous float2 plane per layer, so
 * that each layer's launch reads 8 bytes per pixel instead of a whole line of the interleaved tensor.  CP a multiple of 4. */
int tmg_layer_planes(const void* src, void* dst, int64_t npix, int64_t CP, tmg_stream_t st);

/* Parameter-gradient epilogue of a level's NL plain coupling layers, one launch: d(kappa_k) = (<Wz_k, dWz_k> + <bz_k, dBz_k>) inside the
 * clamp range of the zero conv's log-scale (flowUtils.py:104-106; fp64 accumulation), and the scatter-add of the grouped 4-row
 * weight-gradient results tmpX [NL,4,ch+4,3,3] (x1 | d1 columns) and tmpC [NL,4,Cc,3,3] (conditioning columns) into the native
 * dW1 [NL,1,ch+Cc,3,3] / dW2 [NL,1,ch+Cc+1,3,3] of the two growth-1 convs (denseBlock.py:18-36).  Wz, dWz [NL,C,ch+Cc+2,3,3]; Bz, dBz [NL,C];
 * Kp, dK [NL]; tmpX / tmpC may be null; ws: 4*NL zero-initialised floats (8-byte aligned).  dims = {NL, C, ch, Cc, rows per layer in
 * tmpC: 4, or 2 for a compact tmpC [NL,2,Cc,3,3]}. */
int tmg_level_finish(const void* Wz, const void* dWz, const void* Bz, const void* dBz, const void* Kp, const void* tmpX, const void* tmpC,
                     void* dW1, void* dW2, void* dK, void* ws, const int64_t* dims, tmg_stream_t st);

/* The same mix in full fp32 (v_mfma_f32_16x16x4_f32) for C <= 128: the stand-alone 1x1 mixes (wide flow levels, ConvLSTM blocks)
 * without the general conv kernel's patch staging and operand-packing launch.  Arguments as tmg_mix_f16. */
int tmg_mix_f32(const void* x, const int64_t* x_d, const void* W, const void* bias, void* y, const int64_t* y_d,
                const int64_t* dims, tmg_stream_t st);
/* The same mix with the affine coupling of the generative direction fused in (the per-op chain of the 64- / 128-channel levels:
 * coupling -> mix, flowAffine.py:102-109 + glowConv.py:207-222 + actNorm.py:71-85).  _fwd: x [npix][C] = (x1 | x2), hh = zero-conv output
 * ((shift, r) interleaved): y2 = x2 exp(-2 softsign(r)) - shift, y = W [x1 ; y2] + bias; r and y2 ([npix][C/2] dense) are stored for
 * backward, logdet[image] += sum 2 softsign(r).  _bwd: dy -> dto1 = (W^T dy)[: C/2] ([npix][C/2] dense), dtin2 = gradient w.r.t. x2,
 * dhh = exp(clamp(kappa)) x gradient w.r.t. hh.  dims = {npix, C, pixels per image}; *_d = {pixel stride, channel offset}.  C = 64 or
 * 128; -100 outside the envelope (nothing launched). */
int tmg_mix_f32_affine_fwd(const void* x, const int64_t* x_d, const void* hh, const int64_t* hh_d, const void* W, const void* bias,
                           void* y, const int64_t* y_d, void* r, void* y2, void* logdet, const int64_t* dims, tmg_stream_t st);
int tmg_mix_f32_affine_bwd(const void* dy, const int64_t* dy_d, const void* W, const void* r, const void* t2, const int64_t* t2_d,
                           const void* g, const void* kappa, void* dto1, void* dtin2, const int64_t* dtin2_d, void* dhh,
                           const int64_t* dhh_d, const int64_t* dims, tmg_stream_t st);

/* ---- round 6: the glue around the flow kernels as launches of this library (tmg_glue.hip) ------------------------- */

#define TMG_SUM_TERMS_MAX 8

/* Split.reverse / GaussianDiag.sample (flowUtils.py:194-209, :325-335): z2 = mean + exp(log-std) eps with (mean | log-std) = hz
 * [npix][2 Ch] clipped as in tmg_gauss_fwd, written to out + o_d[1] at pixel stride o_d[0] - i.e. straight into the second half of
 * the [.., 2 Ch] tensor the reference builds with torch.cat((z1, z2), 1) (:334); pass (optional, [npix][Ch] at stride p_d[0]) is
 * copied to out + pass_off beside it.  eps_in given: reconstruct's latents; NULL: eps ~ N(0, 1) is drawn in the kernel
 * (Philox4x32-10, counter = (element quad, site), key = the two int64 at `nonce`, device memory - the host draws the nonce with
 * torch's generator once per model call, so the latents follow torch.manual_seed and stay fresh under hipGraph replay) and stored to
 * eps_out [npix][Ch] (optional) for the backward pass (tmg_gauss_bwd, mode 1).  logp[image] += sum -0.5 (ln 2 pi + 2 lsd + eps^2).
 * dims = {B, pixels per image, Ch, clip_mean, site}; fl = {mean_lo, mean_hi, logstd_lo, logstd_hi}. */
int tmg_gauss_sample(const void* hz, const int64_t* hz_d, const void* eps_in, const int64_t* ei_d, const void* pass,
                     const int64_t* p_d, void* out, const int64_t* o_d, int64_t pass_off, void* eps_out, void* logp, const void* nonce,
                     const int64_t* dims, const float* fl, tmg_stream_t st);

/* The benchmark loss of SURVEY 8-D, generative direction: *loss += fl[0] sum(y^2) + fl[1] sum(logdet) (the caller zeroes *loss;
 * fl = {1 / numel(y), 1 / (B noc H W)}), and its gradient dy = 2 fl[0] *g y, dld[b] = fl[1] *g with the upstream gradient read from
 * device memory.  y, dy: n contiguous floats, 16-byte aligned.  dims = {n, B}.  Replaces the pow / mean / div / add chain the
 * reference-side harness would build from torch ops (main.py trains through TMGLowLoss, trainFlowParallel.py:104-177 - tmg_phys_*). */
int tmg_reverse_loss_fwd(const void* y, const void* ld, void* loss, const int64_t* dims, const float* fl, tmg_stream_t st);
int tmg_reverse_loss_bwd(const void* y, const void* g, void* dy, void* dld, const int64_t* dims, const float* fl, tmg_stream_t st);

/* Log-det bookkeeping (tmGlow.py:412-414, :438-440; flowLSTMBlock.py:314-318, :345-359 sum the per-layer / per-level terms with one
 * `+` each): out[b] = sum_k terms[k][lens[k] == 1 ? 0 : b], n <= TMG_SUM_TERMS_MAX terms of length B or 1 (a broadcast scalar).
 * tmg_vec_sum: out[0] = sum_b g[b] - the gradient of a broadcast term. */
int tmg_sum_terms(const void* const* terms, const int64_t* lens, int64_t n, void* out, int64_t B, tmg_stream_t st);
int tmg_vec_sum(const void* g, int64_t B, void* out, tmg_stream_t st);

/* Parameter-side operands of a level's NL plain coupling layers gathered through a device table tab[NL][5] of the modules' own
 * tensors (w1 [1][ch+Cc][3][3], w2 [1][ch+Cc+1][3][3], wz [C][ch+Cc+2][3][3], bz [C], kappa [1]; denseBlock.py:135-152,
 * flowUtils.py:211-247): Wz [NL][C][ch+Cc+2][3][3] (stack), Wcat [NL C + 2 NLp][Cc][3][3] (rows k C + o: conditioning columns of
 * wz_k; rows NL C + 2k / + 2k+1: conditioning columns of w1_k / w2_k, zero for the NLp - NL padding layers), Bz [NL][C], Kp [NL].
 * dims = {NL, NLp, C, ch, Cc}. */
int tmg_level_pack(const void* tab, void* Wz, void* Wcat, void* Bz, void* Kp, const int64_t* dims, tmg_stream_t st);

/* up[b][y][x][c] = dy[b][y/2][x/2][c] where y and x are even, 0 elsewhere: the operand of the stride-2 input gradient of the encoder's
 * down-sampling convs (tmGlow.py:75-77, :177-181: conv3x3 stride 2) as a stride-1 contraction on the matrix cores.  up: contiguous
 * [B][H][W][C]; dims = {B, H, W, h, w, C}; dy_d = {pixel stride, channel offset}; C % 4 == 0, else -100 (nothing launched). */
int tmg_spread2(const void* dy, const int64_t* dy_d, void* up, const int64_t* dims, tmg_stream_t st);

/* ---- physics-constrained reverse-KL loss (tmg_physics.hip) ------------------------------ */

/* Residual sums of TMGLowLoss (trainFlowParallel.py:121-177 / physicsConstrained.py:42-94): y, target = [N,3,H,W]
 * planar (u_x,u_y,p); sums[3] += {sum pstar^2 over the interior, sum ustar^2 over interior rows, sum (y-target)^2};
 * optional residual fields pstar [N,1,H,W], ustar [N,1,H,W+2].  dims = {N,H,W}; fl = {sd[3], mu[3], dx, dy, rho}. */
int tmg_phys_fwd(const void* y, const void* target, void* sums, void* pstar_out, void* ustar_out, const int64_t* dims,
                 const float* fl, tmg_stream_t st);
/* The two residual fields on their own, for every stencil pair and scaling the reference's API accepts
 * (PhysConstrainedLES.calcDivergence / calcPressurePoisson, physicsConstrained.py:42-94; Grad1Filter2d / Grad2Filter2d with
 * kernel_size 3 or 5, grad1Filter.py:37-88, grad2Filter.py:28-101; scale = True / False): u [N][2][H][W] planar velocity,
 * p [N][1][H][W] (may be NULL when pstar is NULL); ustar [N][1][H][W+2] and / or pstar [N][1][H][W], clamped to [-1, 1].
 * dims = {N, H, W, k1, k2, scale}; fl = {dx, dy, rho}.  k1, k2 outside {3, 5}: -100, nothing launched. */
int tmg_phys_fields(const void* u, const void* p, void* ustar, void* pstar, const int64_t* dims, const float* fl, tmg_stream_t st);
/* Per-pixel RMS over the T steps of y = [B,T,3,H,W] against target_rms [B,3,H,W] (trainFlowParallel.py:143-144);
 * writes mean / coefficient maps for the backward pass; sum_out += sum (rms - target_rms)^2.  dims = {B,T,3*H*W}. */
int tmg_phys_rms(const void* y, const void* trms, void* mean_out, void* coef_out, void* sum_out, const int64_t* dims,
                 tmg_stream_t st);
/* d loss / d y for all four data terms.  dims = {N,T,H,W}; fl = {sd[3], mu[3], dx, dy, rho, cp, cd, cl, cr} with
 * c* = upstream * beta * 2 / (element count of the term). */
int tmg_phys_bwd(const void* y, const void* target, const void* mean, const void* coef, void* dy, const int64_t* dims,
                 const float* fl, tmg_stream_t st);
/* tmg_phys_bwd with c* additionally multiplied by the DEVICE scalar *upstream (the gradient arriving on the loss value), so the
 * host never reads it back (NULL: 1). */
int tmg_phys_bwd_dev(const void* y, const void* target, const void* mean, const void* coef, void* dy, const void* upstream,
                     const int64_t* dims, const float* fl, tmg_stream_t st);

#ifdef __cplusplus
}
#endif
#endif /* TMGLOW_HIP_H */
