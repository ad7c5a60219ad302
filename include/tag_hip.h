/*
 * tag_hip.h -- C ABI of libtag_hip.so: the MI355X (gfx950) hot path of text-to-audio grounding.
 *
 * The reference (wsntxxn/TextToAudioGrounding) is pure Python/PyTorch: it has no FFI of its own.
 * The replaceable seam is the nn.Module contract of SURVEY.md section 8(b); every entry point below
 * names the reference call site whose arithmetic it replaces (paths relative to the reference
 * root).  The Python host side (texttoaudiogrounding_amd/) binds these with ctypes and mirrors the
 * reference's module names and forward() signatures; INTEGRATION.md shows the stub a maintainer of
 * the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter comment says "host";
 *   - all floating tensors are contiguous fp32 unless a stride is given; indices are int64;
 *   - image tensors are channels-last: (B, H=time, W=freq, C);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are async;
 *   - return value: 0 on success, otherwise a negative TAG_E* code or a positive hipError_t;
 *     tag_last_error() returns a static, thread-local message for the last failure;
 *   - no call allocates device memory: scratch is passed in by the caller (`ws`), with the size
 *     given by the matching *_ws_bytes() query.
 */
#ifndef TAG_HIP_H
#define TAG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TAG_ABI_VERSION 3   /* bump whenever an entry point is added or changes its arguments / data layout (lib.py checks it FIRST) */
#define TAG_EINVAL (-1) /* bad argument (shape not supported, null pointer, ...) */
#define TAG_ELAUNCH (-2)

int tag_abi_version(void);
/* sha256 (hex) over the kernel sources the library was compiled from, as lib.csrc_sha256() forms it: the BINARY attests its
 * sources, so a stale libtag_hip.so beside edited sources is refused by lib.load() and a PMC profile is only quoted by
 * bench.py for the binary that actually ran (the reference has no native code: nothing is replaced) */
const char* tag_build_id(void);
const char* tag_last_error(void);
/* Developer switch for A/B timing by tools/ and tests/ (csrc/tag_lib.hip lists the names: "halo_bn256", "gru_coop", ...): set before
 * the first launch that reads it.  The launchers read no environment; texttoaudiogrounding_amd/lib.py forwards the TAG_* variables
 * of the tool scripts here at load time.  Nothing of the reference is replaced.  TAG_EINVAL for an unknown name. */
int tag_set_option(const char* name, int value);
/* Measurement aid of bench.py (csrc/probe.hip; nothing of the reference is replaced): `workgroups` x 4 waves of register-resident
 * MFMA work for `iters` iterations.  kind 0: bf16 32x32x16 on RANDOM operands (what a real kernel's toggling costs: the part's
 * power limit), 1: bf16 constant operands (datasheet conditions), 2 / 3: the same for the exact-fp32 32x32x2.  clocks: 3 x u64,
 * device; [0] shader clocks and [1] 100 MHz reference ticks workgroup 0 ran for.  tag_mfma_probe_flop: the FLOP of such a launch. */
int tag_mfma_probe(int kind, int iters, int workgroups, unsigned seed, void* clocks, void* stream);
double tag_mfma_probe_flop(int kind, int iters, int workgroups);
/* the VALU member of the probe family (tools/hybrid_probe.py): `workgroups` x 4 waves of v_pk_fma_f32 on register operands,
 * iters x 32 instructions x 256 FLOP per wave; clocks as above */
int tag_valu_probe(int iters, int workgroups, unsigned seed, void* clocks, void* stream);
/* number of CUs of the current device (used by the host to size split-K workspaces) */
int tag_device_cu_count(void);

/* ---------------------------------------------------------------------------------------------
 * F1 + F2: torchaudio MelSpectrogram(power=2, center=True, reflect) + AmplitudeToDB(power)
 * models/audio_encoder.py:113-124,183-184 (Cnn8Rnn) and :29-37,68-69 (CrnnEncoder).
 * wave (B,S) -> out (B, F=S/hop+1, n_mels) in dB, time-major (= the (B,1,F,n_mels) image the CNN reads).
 * window: (win_length) hann; fb: (n_fft/2+1, n_mels).  n_fft in {1024, 2048}; n_mels <= 64.
 * power_out (nullable): (B,F,n_mels) mel power before the dB step (parity is checked in the power domain).
 * ------------------------------------------------------------------------------------------- */
int tag_logmel_forward(const float* wave, int B, int S, int n_fft, int win_length, int hop,
                       const float* window, const float* fb, int n_mels, float* out_db,
                       float* power_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Input side of F1: waveform packs hold float16 samples (utils/data/pack_waveform.py:46-52); the reference's loaders widen
 * them to float32 on the host (datasets/single_phrase_dataset.py:44-45) and zero-pad the batch to its longest clip
 * (utils/train_util.py:211-216 via datasets/collate_function.py:24-28).  Device form: packed_f16 = the B ragged clips
 * back to back (what crosses PCIe), offsets (B+1) int64 sample offsets, out (B,S) float32 zero-padded (clips longer than S
 * are cut), len_out (B) int64 = clip lengths (nullable).  Exact: every float16 is a float32.
 * ------------------------------------------------------------------------------------------- */
int tag_waveform_f16_to_f32_padded(const void* packed_f16, const long* offsets, int B, int S, float* out,
                                   long* len_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * F3 / A1: BatchNorm2d statistics over a channels-last tensor x (rows, C)  (nn.BatchNorm2d, train)
 * models/audio_encoder.py:133,188-190 (bn0 over the mel axis) and models/panns.py:35-36,49-50.
 * Produces batch mean / invstd (biased var, eps) and the fused affine  scale = gamma*invstd,
 * shift = beta - mean*scale; updates running_mean/var (momentum, unbiased var) when non-null.
 * pre_op: 0 = stats of x; 1 = stats of leaky_relu(x, 0.1) (CrnnEncoder cdur_block chain).
 * ------------------------------------------------------------------------------------------- */
size_t tag_bn_stats_ws_bytes(long rows, int C);
int tag_bn_stats(const float* x, long rows, int C, int pre_op, const float* gamma, const float* beta,
                 float eps, float momentum, float* running_mean, float* running_var, float* mean,
                 float* invstd, float* scale, float* shift, void* ws, void* stream);
/* the same outputs from the partial statistics a conv kernel wrote in its epilogue (P rows; layout: see
 * tag_conv3x3_forward) */
size_t tag_bn_stats_from_partials_ws_bytes(int P, int C);
int tag_bn_stats_from_partials(const float* partials, int P, int C, const float* gamma, const float* beta, float eps,
                               float momentum, float* running_mean, float* running_var, float* mean,
                               float* invstd, float* scale, float* shift, void* ws, void* stream);
/* eval mode: scale/shift from running statistics */
int tag_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int C, float* scale, float* shift,
                       void* stream);
/* y = x*scale[c] + shift[c] over (rows, C) (bn0 apply) */
int tag_affine_forward(const float* x, long rows, int C, const float* scale, const float* shift,
                       float* y, void* stream);
/* dgamma[c] = sum dy*xhat, dbeta[c] = sum dy  for a plain affine BN (bn0): xhat=(x-mean)*invstd */
int tag_bn_param_grad(const float* x, const float* dy, long rows, int C, const float* mean,
                      const float* invstd, float* dgamma, float* dbeta, void* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A1: 3x3 convolution, stride 1, pad 1, no bias (models/panns.py:25-33,49-50), channels-last,
 * implicit GEMM on the fp32 MFMA (v_mfma_f32_32x32x2_f32).
 *   wpack: (9, Cin, Cout) produced by tag_pack_conv_weight.
 *   prologue applied to x on load (padding stays zero):
 *     0 none | 1 relu(x*in_scale[c]+in_shift[c]) | 2 leaky(x,.1)*in_scale+in_shift | 3 x*in_scale+in_shift
 * Requires Cin % 32 == 0 (use tag_conv3x3_c1_* for Cin == 1).
 * The same entry computes dgrad when given the flipped/transposed pack (wdgrad) and dy as x.
 * ------------------------------------------------------------------------------------------- */
int tag_pack_conv_weight(const float* w /*(Cout,Cin,3,3)*/, float* wfwd /*(9,Cin,Cout)*/,
                         float* wdgrad /*(9,Cout,Cin), nullable*/, int Cin, int Cout, void* stream);
/* stats (nullable): when given, the kernel also writes BatchNorm partial statistics of y in its epilogue --
 * tag_conv3x3_stats_rows(B,H,W,Cout) rows of (3*Cout) floats {tile pivot mu, sum(y-mu), sum((y-mu)^2)} followed
 * by one pixel count per row, i.e. rows*(3*Cout+1) floats; tag_bn_stats_from_partials turns them into
 * the batch statistics (saves the separate pass over y).  rows == 0: not available for this shape. */
int tag_conv3x3_stats_rows(int B, int H, int W, int Cout);
int tag_conv3x3_forward(const float* x, const float* wpack, int prologue, const float* in_scale,
                        const float* in_shift, float* y, float* stats, int B, int H, int W, int Cin, int Cout,
                        void* stream);

/* Alternative arithmetic for the same convolution (forward and dgrad): every fp32 operand is split exactly into
 * three bf16 terms and the products run on the bf16 MFMA (v_mfma_f32_32x32x16_bf16) with fp32 accumulation --
 * 6 of the 9 partial products by default (error of the dropped terms <= 2^-23 per product), TAG_X3_PRODUCTS=9
 * for all of them, =1 for plain bf16.  Opt-in; tag_conv3x3_forward above stays the exact-fp32 default.
 *   wfwd / wdgrad: tag_conv3x3_x3_pack_bytes(Cin, Cout) bytes each (pre-split, MFMA-fragment order).
 *   Requires W in {8,16,32,64}, Cin % 32 == 0, Cout % 64 == 0. */
size_t tag_conv3x3_x3_pack_bytes(int Cin, int Cout);
/* products: partial products per fp32 multiply -- 6 (default), 9 (all), 1 (plain bf16: operands rounded to nearest
 * bf16, one product; BASELINE configs[2] arithmetic), 0 = process default (TAG_X3_PRODUCTS, else 6).  A weight pack
 * made with products == 1 must only be used with products == 1 (its hi plane is rounded, not truncated). */
int tag_pack_conv_weight_x3(const float* w /*(Cout,Cin,3,3)*/, void* wfwd, void* wdgrad, int Cin, int Cout,
                            int products, void* stream);
int tag_conv3x3_x3_stats_rows(int B, int H, int W, int Cout);
/* the same for the bf16-storage entry points (tag_conv3x3_forward_x3_bf16 with this Cin / prologue; tag_conv3x3_dgrad_bnsums_bf16:
 * prologue 0).  Layers with Cin <= 128 run on the row-streaming kernel (csrc/conv_rows.hip: weights stationary in registers,
 * input rows by LDS-DMA into a ring) whose statistics accumulate over a whole strip of rows: one partial row per (strip, wave
 * M-group); the other layers write one row per tile M-group of the tile kernel.  (ABI 2: Cin and prologue were added.) */
int tag_conv3x3_x3_bf16_stats_rows(int B, int H, int W, int Cin, int Cout, int prologue);
/* 0: keep every bf16-storage layer on the tile kernel (A/B timing, tools/conv_rows_bench.py); returns the previous setting.
 * The environment variable TAG_CONV_ROWS=0 is the process-wide form. */
int tag_conv_rows_enable(int on);
/* the same switch for the bf16-storage weight gradient: 1 (default) = the layers without a producer prologue run on
 * csrc/conv_wgrad_dma.hip (operands by LDS-DMA), 2 = the prologue-1 layers too, 0 = every layer on the register-staged kernel
 * of conv_x3.hip; returns the previous setting.  Environment TAG_WGRAD_DMA is the process-wide form. */
int tag_wgrad_dma_enable(int on);
int tag_conv3x3_forward_x3(const float* x, const void* wpack, int prologue, const float* in_scale,
                           const float* in_shift, float* y, float* stats, int B, int H, int W, int Cin, int Cout,
                           int products, void* stream);
/* wgrad with the same arithmetic (K = pixels; operands fetched with the transposing LDS read ds_read_b64_tr_b16).
 * Requires W in {8,16,32,64}, Cin % 64 == 0, Cout % 64 == 0; ws: tag_conv3x3_wgrad_x3_ws_bytes. */
size_t tag_conv3x3_wgrad_x3_ws_bytes(int B, int H, int W, int Cin, int Cout);
int tag_conv3x3_wgrad_x3(const float* x, int prologue, const float* in_scale, const float* in_shift,
                         const float* dy, float* dw /*(Cout,Cin,3,3)*/, int B, int H, int W, int Cin, int Cout,
                         int products, void* ws, void* stream);
/* dw (Cout,Cin,3,3) = sum over pixels of prologue(x)[shifted] * dy ; ws from *_ws_bytes */
/* dgrad convolution fused with the REDUCTION half of the BatchNorm+ReLU backward its result flows into
 * (models/panns.py:49-50 backward: relu_(bn(conv(x)))).  da = conv(dy, wpack_dgrad) is written as by
 * tag_conv3x3_forward; in the epilogue every 64-pixel wave tile also reads yref (the raw conv output the forward
 * pass saved = the BatchNorm input at the same pixels/channels) and writes the partial sums over the tile of
 *   g = da * [yref*bn_scale + bn_shift > 0]    and    g * (yref - bn_mean) * bn_invstd
 * to bnpart [P][2][Cout] (P = tag_conv3x3_stats_rows(B,H,W,Cout) > 0: halo-tile shapes only).  The MFMA-bound conv
 * hides that read; the separate two-tensor reduction pass of tag_bnrelu_backward disappears.
 * tag_bn_grad_from_partials folds the rows (fp64, fixed order) into dgamma / dbeta; tag_bnrelu_backward_apply is the
 * apply half: dy = gamma*invstd*(g - dbeta/N - xhat*dgamma/N) (bn_train) or gamma*invstd*g. */
int tag_conv3x3_dgrad_bnsums(const float* dy, const float* wpack, float* da, const float* yref,
                             const float* bn_scale, const float* bn_shift, const float* bn_mean,
                             const float* bn_invstd, float* bnpart, int B, int H, int W, int Cin, int Cout,
                             void* stream);
/* Winograd F(2x2,3x3) form of the same convolution, all fp32: the A1 ConvBlock convs of models/panns.py:29-38,49-50 with 2.25 x
 * fewer MFMA FLOP than tag_conv3x3_forward.  Drop-in for tag_conv3x3_forward(stats) / tag_conv3x3_dgrad_bnsums / _dgrad_poolsums /
 * _forward_bnrelu_pool_eval / tag_conv3x3_wgrad.  Error vs an fp64 convolution: at or below the direct fp32 kernel's.
 *   Channel counts that are multiples of 64 (every Cnn8Rnn layer but the Cin = 1 conv): ONE kernel per launch (round 6,
 *   csrc/conv_wino_fused.hip) -- the input transform B^T d B (producer BatchNorm+ReLU prologue and zero padding applied on load) is
 *   formed on the way into LDS, the 16 products (tiles x Cin).(Cin x Cout) run on the exact-fp32 MFMA into 16 accumulator sets that
 *   stay in registers, the output transform A^T m A and the epilogue (BatchNorm statistics rows [K | r | q][Cout] + counts for
 *   tag_bn_stats_from_partials / rows [sum g | sum g*xhat][Cout] for tag_bn_grad_from_partials / pooled inference output) work from
 *   a parked output tile.  No workspace (ws is ignored, tag_conv3x3_wino_ws_bytes = 64); one partial row per 64-tile block.
 *   Other channel counts (32..1024 whose quarter divides 256, Cin % 32 == 0): round 5's plane form (csrc/conv_wino.hip): input
 *   transform -> 16 batched products -> output transform through planes in ws.
 *   ufwd / udgrad (16 * Cin * Cout floats each): G g G^T of the filter / of the tap-flipped, channel-swapped filter -- an OPAQUE
 *   pack of tag_pack_conv_weight_wino (fused form: per (64-cout block, 8-channel K chunk) the kernel's LDS image [xi][n 64][k 8];
 *   plane form: [16][Cin][Cout] / [16][Cout][Cin]); P = tag_conv3x3_wino_stats_rows; tag_conv3x3_wino_ok: 1 when the shape is served
 *   (every tensor below 2^31 bytes: a caller cuts larger inference batches, any cut gives the same rows). */
int tag_conv3x3_wino_ok(int B, int H, int W, int Cin, int Cout);
int tag_pack_conv_weight_wino(const float* w /*(Cout,Cin,3,3)*/, float* ufwd, float* udgrad, int Cin, int Cout, void* stream);
size_t tag_conv3x3_wino_ws_bytes(int B, int H, int W, int Cin, int Cout);
int tag_conv3x3_wino_stats_rows(int B, int H, int W, int Cout);
/* weight gradient in the Winograd domain: dw (Cout,Cin,3,3) = G^T [ sum over tiles (A dY A^T) (.) (B^T prologue(x) B) ] G -- the
 * adjoint of the forward form.  Fused form (channel counts multiples of 64): both transforms at staging, 64 ci x 64 co x 16 xi
 * accumulators per workgroup over one slice of the tile axis, ws = 2 S partial filters folded in a fixed order.  Plane form: 16 x S
 * batched products over K slices.  Drop-in for tag_conv3x3_wgrad on the shapes tag_conv3x3_wino_ok accepts. */
size_t tag_conv3x3_wino_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout);
/* v_saved (optional, plane form only): the transformed input the forward launch of the same convolution left in its v_keep
 * buffer -- the input transform is then not repeated (x may be NULL); allowed when tag_conv3x3_wino_wgrad_can_reuse_v (0 for the
 * fused form, which has no planes). */
int tag_conv3x3_wino_wgrad_can_reuse_v(int B, int H, int W, int Cin, int Cout);
int tag_conv3x3_wino_wgrad(const float* x, int prologue, const float* in_scale, const float* in_shift, const float* dy,
                           float* dw /*(Cout,Cin,3,3)*/, int B, int H, int W, int Cin, int Cout, void* ws, const float* v_saved,
                           void* stream);
/* v_keep (optional, plane form): 16 * T * Cin floats that receive the transformed input instead of ws (which then
 * needs the product planes only: 16 * T * Cout floats) -- kept by the caller for tag_conv3x3_wino_wgrad(v_saved). */
int tag_conv3x3_wino_forward(const float* x, const float* ufwd, int prologue, const float* in_scale, const float* in_shift,
                             float* y, float* stats, int B, int H, int W, int Cin, int Cout, void* ws, float* v_keep,
                             void* stream);
/* Winograd twin of tag_conv3x3_dgrad_poolsums (same arguments + ws): dx = conv(dy, udgrad) and, from the output transform, the
 * partial sums [sum dz | sum dz*xhat][Cout] of the BatchNorm+ReLU+pool(ph x 2)+dropout backward of the block below (yref unpooled
 * (B,Hf,Wf,Cout)); P = tag_conv3x3_wino_stats_rows rows for tag_bn_grad_from_partials. */
int tag_conv3x3_wino_dgrad_poolsums(const float* dy, const float* udgrad, float* dx, const float* yref, const float* bn_scale,
                                    const float* bn_shift, const float* bn_mean, const float* bn_invstd, float* bnpart, int B,
                                    int H, int W, int Cin, int Cout, int Hf, int Wf, int ph, int pw, int pool, float drop_p,
                                    uint64_t seed, void* ws, void* stream);
/* inference twin of tag_conv3x3_forward_bnrelu_pool_eval: the output transform pools its own 2 x 2 tile (one 2 x 2 window or two
 * 1 x 2 windows) after BatchNorm(eval) + ReLU; the raw conv output is never written. */
int tag_conv3x3_wino_forward_bnrelu_pool_eval(const float* x, const float* ufwd, int prologue, const float* in_scale,
                                              const float* in_shift, float* out, const float* bn_scale, const float* bn_shift,
                                              int B, int H, int W, int Cin, int Cout, int ph, int pw, int pool, void* ws,
                                              void* stream);
int tag_conv3x3_wino_dgrad_bnsums(const float* dy, const float* udgrad, float* da, const float* yref, const float* bn_scale,
                                  const float* bn_shift, const float* bn_mean, const float* bn_invstd, float* bnpart, int B,
                                  int H, int W, int Cin, int Cout, void* ws, void* stream);
size_t tag_bn_grad_from_partials_ws_bytes(int P, int C);
int tag_bn_grad_from_partials(const float* bnpart, int P, int C, float* dgamma, float* dbeta, void* ws, void* stream);
int tag_bnrelu_backward_apply(const float* y, const float* scale, const float* shift, const float* mean,
                              const float* invstd, const float* gamma, const float* da, float* dy,
                              const float* dgamma, const float* dbeta, long rows, int C, int bn_train, void* stream);
size_t tag_conv3x3_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout);
int tag_conv3x3_wgrad(const float* x, int prologue, const float* in_scale, const float* in_shift,
                      const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, void* ws,
                      void* stream);
/* Cin == 1 (conv_block1.conv1): x (B,H,W) with optional per-column affine x*col_scale[w]+col_shift[w]
 * (bn0 folded: the BatchNorm2d(64) over the mel axis), w (Cout,1,3,3) */
int tag_conv3x3_c1_forward(const float* x, const float* col_scale, const float* col_shift,
                           const float* w, float* y, int B, int H, int W, int Cout, void* stream);
/* the same with the BatchNorm partial statistics of y written by the kernel (W == 64, Cout == 64: rows > 0):
 * stats = [P][3][Cout] rows (pivot, sum(y - pivot), sum((y - pivot)^2)) + [P] pixel counts -> tag_bn_stats_from_partials */
int tag_conv3x3_c1_stats_rows(int B, int H, int W, int Cout);
int tag_conv3x3_c1_forward_stats(const float* x, const float* col_scale, const float* col_shift, const float* w,
                                 float* y, float* stats, int B, int H, int W, int Cout, void* stream);
size_t tag_conv3x3_c1_wgrad_ws_bytes(int B, int H, int W, int Cout);
int tag_conv3x3_c1_wgrad(const float* x, const float* col_scale, const float* col_shift,
                         const float* dy, float* dw, int B, int H, int W, int Cout, void* ws,
                         void* stream);
int tag_conv3x3_c1_dgrad(const float* dy, const float* w, float* dx, int B, int H, int W, int Cout,
                         void* stream);
/* fused wgrad + dgrad of the Cin = 1 convolution: ONE pass over dy (W == 64, Cout == 64 only) */
size_t tag_conv3x3_c1_backward_ws_bytes(int B, int H, int W, int Cout);
int tag_conv3x3_c1_backward(const float* x, const float* col_scale, const float* col_shift, const float* dy,
                            const float* w /*(Cout,1,3,3)*/, float* dw /*(Cout,1,3,3)*/, float* dx /*(B,H,W)*/,
                            int B, int H, int W, int Cout, void* ws, void* stream);
/* The same pass fed the RAW dgrad output of block 1's second conv: `da` = dL/d relu(bn1(yref)), and the backward of that
 * BatchNorm + ReLU (models/panns.py:46-48, x = F.relu_(self.bn1(self.conv1(x)))) is applied as the values are loaded --
 * dy = gamma*invstd * (dz - dbeta/N - xhat * dgamma/N), dz = da where bn1(yref) > 0, N = B*H*W (bn_train = 0: the two
 * mean terms are dropped) -- instead of in a separate tag_bnrelu_backward_apply pass over the largest activation.
 * dgamma / dbeta are INPUTS here: sum(dz * xhat) / sum(dz), as tag_bn_grad_from_partials leaves them. */
int tag_conv3x3_c1_backward_bnrelu(const float* x, const float* col_scale, const float* col_shift, const float* da,
                                   const float* yref, const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                   const float* bn_invstd, const float* gamma, const float* dgamma, const float* dbeta,
                                   int bn_train, const float* w, float* dw, float* dx, int B, int H, int W, int Cout,
                                   void* ws, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A1 + A2: relu(bn(y)) -> avg_pool + max_pool (kernel = stride = (ph,pw), floor) -> dropout
 * models/panns.py:49-58 with pool_type 'avg+max'; F.dropout models/audio_encoder.py:203-210.
 * y raw conv output (B,H,W,C); out (B,H/ph,W/pw,C).  Dropout keep-mask = counter-based hash of
 * (seed, flat output index); p = 0 disables.  act: 1 relu(bn) | 2 leaky(.1) without bn (scale=NULL),
 * pool: 0 avg+max | 1 LPPool(norm 4) | 2 avg | 3 max (the three pool_type values of ConvBlock.forward, models/panns.py:51-60).
 * ------------------------------------------------------------------------------------------- */
int tag_bnact_pool_forward(const float* y, const float* scale, const float* shift, float* out, int B,
                           int H, int W, int C, int ph, int pw, int act, int pool, float drop_p,
                           uint64_t seed, void* stream);
/* backward of the above + the BatchNorm backward, two passes:
 *   reduce: dgamma/dbeta (needs ws), apply: dy = invstd*gamma*(dz - dbeta/N - xhat*dgamma/N)
 * dout is the gradient wrt `out`.  */
size_t tag_bn_backward_ws_bytes(long rows, int C);
int tag_bnrelu_pool_backward(const float* y, const float* scale, const float* shift,
                             const float* mean, const float* invstd, const float* gamma,
                             const float* dout, float* dy, float* dgamma, float* dbeta, int B, int H,
                             int W, int C, int ph, int pw, int pool /* 0 avg+max | 2 avg | 3 max */, float drop_p,
                             uint64_t seed, int bn_train, void* ws, void* stream);
/* Pool backward without its own reduction pass (round 5; models/panns.py:46-62 backward).  The dgrad convolution of the NEXT block's first
 * conv produces dout = dL/d(pooled, dropped-out output) -- tag_conv3x3_dgrad_poolsums writes it as tag_conv3x3_forward would AND, from its
 * output tile (parked in LDS at the end of the kernel), the partial sums over the tile of dz and dz * xhat, where dz is the gradient at the
 * BatchNorm output of yref (B,Hf,Wf,Cout): dropout undone with the forward's counter-based mask, ReLU mask and first-maximum arg-max
 * recomputed from the ph x 2 window of yref.  bnpart [P][2][Cout], P = tag_conv3x3_stats_rows(B,H,W,Cout) (row 2t carries m-tile t,
 * row 2t + 1 is zero), H = Hf / ph, W = Wf / pw; windows ph x 2 with ph 1 or 2.  tag_bn_grad_from_partials folds the rows into
 * dgamma / dbeta; tag_bnrelu_pool_backward_apply is the apply pass of tag_bnrelu_pool_backward alone. */
int tag_conv3x3_dgrad_poolsums(const float* dy, const float* wpack, float* dx, const float* yref, const float* bn_scale,
                               const float* bn_shift, const float* bn_mean, const float* bn_invstd, float* bnpart, int B, int H,
                               int W, int Cin, int Cout, int Hf, int Wf, int ph, int pw, int pool, float drop_p, uint64_t seed,
                               void* stream);
/* Inference forward of one ConvBlock stage (models/panns.py:49-60 with BatchNorm in eval mode; models/hf_modeling_grounding.py:97-180
 * runs the encoder that way): out (B, H/ph, W/pw, Cout) = pool(relu(conv3x3(prologue(x)) * bn_scale + bn_shift)), window ph x 2 (ph 1 | 2,
 * floor), pool 0 avg+max | 2 avg | 3 max.  The raw conv output -- the largest tensor of the block -- is never written: the kernel
 * pools its own output tile.  Bit-identical to tag_conv3x3_forward followed by tag_bnact_pool_forward (act 1, no dropout).
 * Halo-tile shapes only (W 8/16/32/64), producer prologue 0 | 1. */
int tag_conv3x3_forward_bnrelu_pool_eval(const float* x, const float* wpack, int prologue, const float* in_scale,
                                         const float* in_shift, float* out, const float* bn_scale, const float* bn_shift, int B,
                                         int H, int W, int Cin, int Cout, int ph, int pw, int pool, void* stream);
/* bf16-storage twin (BASELINE configs[2] mode): the sums are taken from the bf16 values of dx the apply pass will read back; rows:
 * tag_conv3x3_dgrad_poolsums_bf16_rows (0 = shape not served: keep tag_bnrelu_pool_backward_bf16) */
int tag_conv3x3_dgrad_poolsums_bf16_rows(int B, int H, int W, int Cin, int Cout);
int tag_conv3x3_dgrad_poolsums_bf16(const void* dy, const void* wpack, void* dx, const void* yref, const float* bn_scale,
                                    const float* bn_shift, const float* bn_mean, const float* bn_invstd, float* bnpart, int B,
                                    int H, int W, int Cin, int Cout, int Hf, int Wf, int ph, int pw, int pool, float drop_p,
                                    uint64_t seed, void* stream);
int tag_bnrelu_pool_backward_apply(const float* y, const float* scale, const float* shift, const float* mean,
                                   const float* invstd, const float* gamma, const float* dout, float* dy, const float* dgamma,
                                   const float* dbeta, int B, int H, int W, int C, int ph, int pw, int pool, float drop_p,
                                   uint64_t seed, int bn_train, void* stream);
int tag_bnrelu_pool_backward_apply_bf16(const void* y, const float* scale, const float* shift, const float* mean,
                                        const float* invstd, const float* gamma, const void* dout, void* dy, const float* dgamma,
                                        const float* dbeta, int B, int H, int W, int C, int ph, int pw, int pool, float drop_p,
                                        uint64_t seed, int bn_train, void* stream);
/* same without pooling: upstream gradient da (rows,C) wrt relu(bn(y)); dy may alias da */
int tag_bnrelu_backward(const float* y, const float* scale, const float* shift, const float* mean,
                        const float* invstd, const float* gamma, const float* da, float* dy,
                        float* dgamma, float* dbeta, long rows, int C, int bn_train, void* ws,
                        void* stream);
/* CrnnEncoder (cdur_block, models/audio_encoder.py:16-22,39-49): BatchNorm in FRONT of the conv, applied to
 * v = pre(x), pre_op 0 identity | 1 leaky_relu(x, 0.1).  du = gradient wrt bn(v) (from the conv dgrad);
 * dx = gradient wrt x; dgamma/dbeta the BN parameter gradients.  gamma nullable (= 1). */
int tag_bn_act_backward(const float* x, int pre_op, const float* mean, const float* invstd,
                        const float* gamma, const float* du, float* dx, float* dgamma, float* dbeta,
                        long rows, int C, int bn_train, void* ws, void* stream);
/* backward of dropout(LPPool2d(4,(ph,pw))(leaky_relu(y,0.1))) (forward: tag_bnact_pool_forward act=2 pool=1) */
int tag_lppool_leaky_backward(const float* y, const float* dout, float* dy, int B, int H, int W, int C,
                              int ph, int pw, float drop_p, uint64_t seed, void* stream);
/* materialise the dropout keep-masks (0/1 bytes) the kernels use, for parity tests.  tag_dropout_mask: one splitmix64 per
 * element, 24-bit uniform (tag_mean_w_*, tag_dropout_*, the attention heads); tag_dropout_mask_pooled: the mask of the pooled
 * activations (tag_bnact_pool_forward, tag_bnrelu_pool_backward, tag_lppool_leaky_backward): one splitmix64 per 4 consecutive
 * elements, 16-bit uniform each (csrc/tag_common.h tag_keep4; F.dropout in models/audio_encoder.py:203-210 draws from
 * torch's Philox stream instead -- masks are replayed by seed, never compared with torch's) */
int tag_dropout_mask(uint64_t seed, long n, float p, uint8_t* mask, void* stream);
int tag_dropout_mask_pooled(uint64_t seed, long n, float p, uint8_t* mask, void* stream);

/* A3: mean over W then dropout: x (rows, W, C) -> (rows, C)   models/audio_encoder.py:212-215 */
int tag_mean_w_forward(const float* x, long rows, int W, int C, float drop_p, uint64_t seed, float* out,
                       void* stream);
int tag_mean_w_backward(const float* dout, long rows, int W, int C, float drop_p, uint64_t seed,
                        float* dx, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Dense fp32 GEMM on the MFMA, row-major:  C = act(op(A) op(B) + bias) [+ C if accumulate]
 *   op(A): transA ? A^T : A, A stored (M,K) ld=lda or (K,M) when transA; same for B (K,N)/(N,K).
 * Serves nn.Linear fc1 / audio_proj / text_proj (models/audio_encoder.py:140,216;
 * models/audio_text_model.py:45-46,78-87), the GRU input projections and all their backward GEMMs.
 * act: 0 none, 1 relu, 3 gelu (erf form), 4 tanh, 5 sigmoid.  bias (N) nullable.
 * ------------------------------------------------------------------------------------------- */
/* ws (nullable): scratch of tag_gemm_ws_bytes(M,N,K) bytes enabling a deterministic split-K for problems whose
 * MxN tile count cannot fill the chip (weight gradients: K = B*T); 0 bytes = not needed. */
size_t tag_gemm_ws_bytes(int M, int N, int K);
int tag_gemm(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C,
             int ldc, int M, int N, int K, const float* bias, int act, int accumulate, void* ws,
             void* stream);
/* tag_gemm with both operands rounded to bf16 (nearest-even) on v_mfma_f32_32x32x16_bf16, fp32 accumulate and fp32 tensors:
 * what autocast does to nn.Linear / the GRU projections in BASELINE configs[2] (bf16).  Same arguments, same workspace. */
int tag_gemm_bf16(const float* A, int lda, int transA, const float* B, int ldb, int transB, float* C,
             int ldc, int M, int N, int K, const float* bias, int act, int accumulate, void* ws,
             void* stream);
/* out[n] = sum_m x[m, n] (bias gradients); x (M,N) ld; ws >= tag_colsum_ws_bytes */
size_t tag_colsum_ws_bytes(long M, int N);
int tag_colsum(const float* x, int ld, long M, int N, float* out, void* ws, void* stream);
/* dx = dy * (y > 0) elementwise (relu_ backward), dx may alias dy */
int tag_relu_backward(const float* y, const float* dy, float* dx, long n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * A4: bidirectional GRU recurrence, PyTorch gate order (r,z,n), h0 = 0, ALL T steps
 * nn.GRU(512,256,bidirectional=True,batch_first=True) models/audio_encoder.py:141,217.
 *   gi   (B,T,2,3H): x W_ih^T + b_ih for both directions (a tag_gemm)
 *   w_hh (2,3H,H), b_hh (2,3H)
 *   y    (B,T,2H)   hidden states, forward direction in [:H], reverse in [H:]
 *   gates(B,T,2,4H) saved r,z,n and (W_hn h + b_hn) for the backward pass (nullable in inference)
 * backward: dy (B,T,2H) -> dgi (B,T,2,3H), dgh (B,T,2,3H); hprev (B,T,2,H) is also written
 * (h_{t-1} per direction) so that dW_hh = dgh^T hprev is a plain GEMM for the caller.
 * When the grid (H/16)*ceil(B/16)*2 fits the device (occupancy query x CUs) the whole sequence is ONE persistent
 * launch (weights in registers, state exchanged between workgroups through tagged 8-byte granules), issued with
 * hipLaunchCooperativeKernel so that co-residency of the spinning workgroups is guaranteed by the runtime; if the
 * grid does not fit or the cooperative launch is refused, one launch per step (same arithmetic).
 * ------------------------------------------------------------------------------------------- */
/* scratch for both directions of either pass: transposed weights + the exchange granules of the persistent
 * kernels + a STICKY error word (last 256 bytes): the CALLER zero-fills the scratch once when it allocates it; a
 * persistent kernel whose bounded spin ran out sets the word (and poisons its outputs with NaN, which makes
 * tag_adam_step skip the step); the library never clears it.  Copy the last 256 bytes to the host after a
 * synchronisation and pass them to tag_gru_timed_out. */
size_t tag_gru_ws_bytes(int B, int T, int H);
int tag_gru_forward(const float* gi, const float* w_hh, const float* b_hh, float* y, float* gates,
                    void* ws /* tag_gru_ws_bytes */, int B, int T, int H, void* stream);
int tag_gru_backward(const float* dy, const float* y, const float* gates, const float* w_hh,
                     float* dgi, float* dgh, float* hprev, void* scratch /* tag_gru_ws_bytes */, int B, int T,
                     int H, void* stream);
/* A HIP stream restricted to the compute units set in mask (bit i = CU i; `words` 32-bit words) -- hipExtStreamCreateWithCUMask.
 * The host side keeps the weight-gradient side stream off a share of the CUs with it (ops.py, TAG_WGRAD_CU_SKIP). */
int tag_stream_create_cu_mask(const unsigned* mask, int words, void** stream_out);
int tag_gru_timed_out(const void* host_copy_of_err_word);
/* After a timeout: make every later persistent GRU launch of this process publish its exchange granules write-through
 * (agent scope) instead of L2-resident on one XCD -- the fast form depends on gfx950 cache behaviour (SPX partition mode,
 * MTYPE_RW) beyond the HIP memory model; a stale granule can only show as a timeout.  Returns the previous setting. */
int tag_gru_disable_xcd_fast(void);

/* ---------------------------------------------------------------------------------------------
 * T1 + T2: nn.Embedding gather + mean over valid tokens
 * models/text_encoder.py:39-43,79-88; models/utils.py:33-58.
 * text (B,L) int64, text_len (B) int64, table (V,D); token_emb (B,L,D) nullable; seq_emb (B,D).
 * backward accumulates into dtable (V,D), which the caller zero-fills; it is DETERMINISTIC (no atomics: the
 * first occurrence of a token id owns its table row and sums all occurrences in a fixed order).
 * nn.Embedding raises on ids outside [0,V): tag_embed_check_ids sets *err_flag (device int, sticky, caller-zeroed)
 * to 1 when any id is out of range -- the host reads it at its next synchronisation; the gather/scatter kernels
 * clamp (forward) or skip (backward) such ids so that nothing is read or written out of bounds.
 * ------------------------------------------------------------------------------------------- */
int tag_embed_check_ids(const int64_t* text, long n, int V, int* err_flag, void* stream);
int tag_embed_mean_forward(const int64_t* text, const int64_t* text_len, const float* table,
                           float* token_emb, float* seq_emb, int B, int L, int D, int V, void* stream);
int tag_embed_mean_backward(const float* dseq, const int64_t* text, const int64_t* text_len,
                            float* dtable, int B, int L, int D, int V, void* stream);

/* ---------------------------------------------------------------------------------------------
 * M1 / M2: frame x phrase heads, text_level 'seq'   models/match.py:16-33 (ExpNegL2), :43-60 (DotProduct)
 * kind 0: sigmoid(a.t [/sqrt(D)]).clamp(1e-7,1)   kind 1: exp(-||a - t||)   (l2norm: normalise both first)
 * audio (B,T,D), text (B,D) -> sim (B,T).  backward -> daudio (B,T,D), dtext (B,D).
 * ------------------------------------------------------------------------------------------- */
int tag_match_forward(const float* audio, const float* text, float* sim, int kind, int l2norm,
                      int scale, int B, int T, int D, void* stream);
int tag_match_backward(const float* audio, const float* text, const float* sim, const float* dsim,
                       float* daudio, float* dtext, int kind, int l2norm, int scale, int B, int T,
                       int D, void* stream);

/* M3: align.DotProduct models/align.py:14-31: audio (B,T,D), text (B,N,D) -> (B,B,T,N) */
int tag_align_dot_forward(const float* audio, const float* text, float* out, int l2norm, int scaled,
                          int B, int T, int N, int D, float* ws /* (B*T + B*N)*D floats when l2norm */,
                          void* stream);
/* backward pieces of M3 (composed by the host with two tag_gemm calls: daudio = ds x text, dtext = ds^T x audio):
 * ds (B*T, B*N) = dout * p(1-p) * [1/sqrt(D)] through the clamp; F.normalize forward / backward on rows */
int tag_align_dot_dscore(const float* out, const float* dout, float* ds, int scaled, int B, int T, int N,
                         int D, void* stream);
int tag_l2norm_rows_forward(const float* x, float* y, long rows, int D, void* stream);
int tag_l2norm_rows_backward(const float* x, const float* du, float* dx, long rows, int D, void* stream);

/* ---------------------------------------------------------------------------------------------
 * R1 + L1: FrameBceLoss losses.py:12-24 after the label alignment of run_strong.py:107-118.
 * sim (B, >=Tt) row stride ld_sim, label (B, >=Tt) row stride ld_label, length (B) int64 (unclamped:
 * clamped to [1,Tt] inside).  loss: 1 float.  backward: dsim (B,ld_sim) zero outside the mask.
 * ------------------------------------------------------------------------------------------- */
int tag_frame_bce_forward(const float* sim, int ld_sim, const float* label, int ld_label,
                          const int64_t* length, int B, int Tt, float* loss, void* stream);
int tag_frame_bce_backward(const float* sim, int ld_sim, const float* label, int ld_label,
                           const int64_t* length, int B, int Tt, const float* dloss, float* dsim,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * P1: post-processing utils/eval_util.py:18-116 driven as in run_strong.py:203-252:
 * binarize (strict >, float64 compare) -> median filter (window, scipy 'reflect') -> connect
 * clusters (gap <= n_connect) -> contiguous regions.  One (clip, threshold) pair per thread.
 * sim (B,T) ld; thresholds (NT) double; regions (B,NT,max_regions,2) int64 rows [onset, offset);
 * counts (B,NT) int32.
 * ------------------------------------------------------------------------------------------- */
int tag_segments(const float* sim, int ld, int B, int T, const double* thresholds, int NT, int window,
                 int n_connect, int64_t* regions, int32_t* counts, int max_regions, void* stream);

/* ---------------------------------------------------------------------------------------------
 * O1: clip_grad_norm_ + Adam on a flat fp32 buffer (run_strong.py:143-145; torch.optim.Adam).
 * tag_sumsq: out[0] (double) = sum g^2 (zero-filled by the call).  tag_adam_step reads the squared
 * norm from device memory: coef = min(1, max_norm / (sqrt(gnorm_sq) + 1e-6)); max_norm <= 0 disables.
 * grad_scale multiplies g first (1/world_size for DP averaging).  A non-finite squared norm makes tag_adam_step a
 * no-op (parameters and moments untouched): one poisoned step cannot destroy the optimiser state.
 * ------------------------------------------------------------------------------------------- */
size_t tag_sumsq_ws_bytes(long n);
int tag_sumsq(const float* g, long n, double* out, void* ws, void* stream);
int tag_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1,
                  float beta2, float eps, int step, const double* gnorm_sq, float max_norm,
                  float grad_scale, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Inference path, text tower (next row of the scope table: models/hf_modeling_grounding.py:183-199 LaionClapEncoder =
 * Hugging Face ClapTextModel + ClapProjectionLayer, RoBERTa-style post-LayerNorm encoder), forward only.  Dense
 * layers are tag_gemm calls (bias + GELU / tanh / ReLU epilogues); these are the row kernels in between.
 * ------------------------------------------------------------------------------------------- */
/* out[b,l,:] = LayerNorm(word[ids] + type0 + pos[position_id]), position_id = cumsum(ids != pad)*(ids != pad) + pad */
int tag_roberta_embed_ln(const long* ids /*(B,L)*/, const float* word, const float* type0, const float* pos,
                         const float* gamma, const float* beta, float eps, float* out /*(B*L,D)*/, int B, int L,
                         int D, int pad_id, void* stream);
/* out = LayerNorm(x + res) * gamma + beta over rows of D (res nullable) */
int tag_add_layernorm(const float* x, const float* res, const float* gamma, const float* beta, float eps,
                      float* out, long rows, int D, void* stream);
/* softmax(q k^T / sqrt(dh) + key mask) v per (sequence, head); qkv (B*L, 3*heads*dh) = [q|k|v]; L <= 64;
 * dh in {16,32,64}; mask (B,L) int64 (0 = padded key) */
int tag_mha_small(const float* qkv, const long* mask, float* out /*(B*L, heads*dh)*/, int B, int L, int heads,
                  int dh, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BASELINE configs[3]: cross-encoder (models/cross_encoder.py:5-79) + token-level DotProduct (models/match.py:43-60).
 * Seq2SeqAttention's Linear over the concatenation [query ; kv] is applied as two tag_gemm calls
 * (aq = query Wq^T (B,T,Da), ak = kv Wk^T + b (B,L,Da)); these entries do the rest.  L <= 32 tokens.
 * ------------------------------------------------------------------------------------------- */
/* score = v . tanh(aq[b,q] + ak[b,k]); rows q >= qlen[b] and columns k >= klen[b] filled with -1e10; attn = softmax_k;
 * ctx = attn @ kv (B,T,Dk) */
int tag_addattn_forward(const float* aq, const float* ak, const float* v, const float* kv /*(B,L,Dk)*/,
                        const long* qlen, const long* klen, float* attn /*(B,T,L)*/, float* ctx, int B, int T, int L,
                        int Da, int Dk, void* stream);
size_t tag_addattn_backward_ws_bytes(int B, int T, int L, int Da, int Dk);
/* given dctx: daq (B,T,Da), dak (B,L,Da), dkv (B,L,Dk) (the attn @ kv term only), dv (Da); deterministic */
int tag_addattn_backward(const float* aq, const float* ak, const float* v, const float* kv, const float* attn,
                         const float* dctx, const long* qlen, const long* klen, float* daq, float* dak, float* dkv,
                         float* dv, int B, int T, int L, int Da, int Dk, void* ws, void* stream);
/* CrossGating pieces (models/cross_encoder.py:51-57): out = a * b; backward of out = x * g, g = sigmoid(z):
 * dx (+)= dout * g, dz = dout * x * g * (1 - g).  n % 4 == 0 */
int tag_mul(const float* a, const float* b, float* out, long n, void* stream);
int tag_gate_backward(const float* dout, const float* x, const float* g, float* dx, int accumulate, float* dz, long n,
                      void* stream);
/* DotProduct with text_level="token" after a cross-encoder: sim[r] = sigmoid(a[r].b[r] [/sqrt(D)]).clamp(1e-7, 1) */
int tag_rowdot_sigmoid_forward(const float* a, const float* b, float* sim, long rows, int D, int scale, void* stream);
int tag_rowdot_sigmoid_backward(const float* a, const float* b, const float* dsim, float* da, float* db, long rows,
                                int D, int scale, void* stream);
/* Both heads with text_level="token" in general (models/match.py:16-33, 43-60): rows r = (clip, frame), one text vector per
 * frame.  kind 0 = DotProduct (sigmoid(u.w [/sqrt(D)]).clamp(1e-7,1)), kind 1 = ExpNegL2 (exp(-||u - w||)); l2norm:
 * u = a/max(||a||,1e-12), w likewise (F.normalize). */
int tag_rowpair_forward(const float* a, const float* b, float* sim, long rows, int D, int kind, int l2norm, int scale,
                        void* stream);
int tag_rowpair_backward(const float* a, const float* b, const float* dsim, float* da, float* db, long rows, int D,
                         int kind, int l2norm, int scale, void* stream);
/* dtable[text[b,l]] += dtok[b,l,:] (token_emb gradient of EmbeddingLayer, models/text_encoder.py:37-43) */
int tag_embed_tokens_backward(const float* dtok, const long* text, float* dtable, int B, int L, int D, int V,
                              void* stream);

/* ---------------------------------------------------------------------------------------------
 * Weak supervision (next row of the scope table): MultiTextBiEncoder (models/audio_text_model.py:101-229) -- N phrases
 * per clip scored against the SAME audio embedding (no (B*N,T,D) expansion), pooled by linear_softmax_with_lens
 * (models/utils.py:75-76); ClipBceLoss (losses.py:38-43) is tag_frame_bce_* over the (B,N) clip matrix.
 * ------------------------------------------------------------------------------------------- */
/* sim[(b*N+n), t] = sigmoid(audio[b,t] . text[b*N+n] [/sqrt(D)]).clamp(1e-7,1);  N <= 16 in backward */
int tag_match_group_forward(const float* audio /*(B,T,D)*/, const float* text /*(B*N,D)*/, float* sim /*(B*N,T)*/,
                            int scale, int B, int N, int T, int D, void* stream);
int tag_match_group_backward(const float* audio, const float* text, const float* dsim, float* daudio /*(B,T,D)*/,
                             float* dtext /*(B*N,D)*/, int scale, int B, int N, int T, int D, void* stream);
/* clip[r] = sum_{t<len} f^2 / sum_{t<len} f over rows of frame probabilities (rows, T); len = length[r / group] */
int tag_linear_softmax_pool_forward(const float* fs, const long* length, float* clip, long rows, int T, int group,
                                    void* stream);
int tag_linear_softmax_pool_backward(const float* fs, const long* length, const float* dclip, float* dfs, long rows,
                                     int T, int group, void* stream);

/* align-by-phrase (models/audio_text_model.py:907-976): sim_pooling.AudioMeanTextMean over the (B,B,T,N) matrix of
 * tag_align_dot_forward (models/sim_pooling.py:6-22) and MaxMarginRankingLoss(fix_norm=True) (losses.py:226-264) */
int tag_meanmean_pool_forward(const float* sim, const long* audio_len, const long* text_len, float* out /*(B,B)*/, int B,
                              int T, int N, void* stream);
int tag_meanmean_pool_backward(const float* dout, const long* audio_len, const long* text_len, float* dsim, int B, int T,
                               int N, void* stream);
/* fix_norm != 0 (the reference's default): mean over the 2 n (n-1) off-diagonal pairs; 0: the diagonal pairs stay in
 * (losses.py:249-264) */
int tag_maxmargin_forward(const float* x /*(n,n)*/, int n, float margin, float lamda1, int fix_norm, float* loss,
                          void* stream);
int tag_maxmargin_backward(const float* x, int n, float margin, float lamda1, int fix_norm, const float* dloss, float* dx,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * EmbeddingAgg(aggregation="attention") = AttentionPooling, models/text_encoder.py:46-58,79-86: x (B,L,D) token embeddings,
 * lens (B), fc.weight w (D), fc.bias (1) -> weight (B,L) softmax over the valid tokens (-1e10 fill), out (B,D).
 * backward: dx (B,L,D) (to be ADDED to any other gradient of the tokens by the caller), gw (B,D) / gb (B) = per-phrase terms
 * of d fc.weight / d fc.bias (fold with tag_colsum).
 * BiEncoder(upsample=True), models/audio_text_model.py:90-97: F.interpolate(mode="linear", align_corners=False) of the
 * frame scores x (R,T) -> (R, T*ratio); backward gathers (no atomics).
 * ------------------------------------------------------------------------------------------- */
int tag_attnpool_forward(const float* x, const long* lens, const float* w, const float* bias, float* weight, float* out,
                         int B, int L, int D, void* stream);
int tag_attnpool_backward(const float* x, const float* w, const float* weight, const float* dout, float* dx, float* gw,
                          float* gb, int B, int L, int D, void* stream);
int tag_upsample_linear_forward(const float* x, float* out, long R, int T, int ratio, void* stream);
int tag_upsample_linear_backward(const float* dout, float* dx, long R, int T, int ratio, void* stream);
/* MultiTextBiEncoder with a cross-encoder (models/audio_text_model.py:165-168, audio_emb.unsqueeze(1).expand(-1, text_num,
 * -1, -1).reshape): out[(b*N + n)][0..R) = x[b][0..R); backward = sum over the N copies in ascending order. */
int tag_group_expand_forward(const float* x, float* out, long B, int N, long R, void* stream);
int tag_group_expand_backward(const float* dout, float* dx, long B, int N, long R, void* stream);

/* ---------------------------------------------------------------------------------------------
 * General similarity pooling: models/utils.py:22-105 (mean/max/linear_softmax/exp_softmax _with_lens), all twelve reducers
 * of models/sim_pooling.py:6-204 and the pooling modes of MultiTextBiEncoder (models/audio_text_model.py:205-215).
 * sim (R,T,N) fp32, T frames, N tokens/phrases innermost; alen indexed by r / a_div, tlen by r % t_mod.
 * amode (frames < alen): 0 mean, 1 max (first maximum), 2 linear softmax sum f^2 / sum f, 3 exp softmax sum softmax(f) f.
 * tmode (tokens < tlen): 0 mean, 1 sum, 2 max (first), 3 mean + sum; -1 = none: out is (R,N), else (R).
 * backward writes the whole dsim (R,T,N) (zeros outside the valid region).
 * ------------------------------------------------------------------------------------------- */
int tag_sim_pool_forward(const float* sim, const long* alen, const long* tlen /* nullable iff tmode < 0 */, float* out,
                         long R, int T, int N, int a_div, int t_mod, int amode, int tmode, void* stream);
int tag_sim_pool_backward(const float* sim, const long* alen, const long* tlen, const float* dout, float* dsim, long R,
                          int T, int N, int a_div, int t_mod, int amode, int tmode, void* stream);

/* ---------------------------------------------------------------------------------------------
 * BASELINE configs[3], the head named in the config text: match.CrossAttention (models/match.py:63-88) =
 * nn.MultiheadAttention(embed_dim E, heads H, dropout p, batch_first, kdim = vdim = kvdim) of every audio frame over the
 * phrase tokens -> audio + dropout(out) -> LayerNorm(E) -> Linear(E,1) -> sigmoid.  The four projections are tag_gemm
 * calls; these entry points are the attention core and the fused residual/LayerNorm/head, forward and backward.
 *   q (B,T,E) projected frames; k, v (B,L,E) projected tokens (L <= 32); klen (B) int64 valid tokens (key_padding_mask);
 *   attn (B,T,H,L) softmax weights BEFORE dropout (saved for backward); ctx (B,T,E).  head_dim = E/H in {16, 32} or a
 *   multiple of 64, E <= 1024.  drop_p / seed: dropout on the attention weights (train), counter-based like A2.
 * backward: dq (B,T,E), dk, dv (B,L,E) (sums over frames folded from per-tile partials in a fixed order).
 * tag_resln_head_*: x = audio, r = out_proj(ctx); sim (rows) = sigmoid(LayerNorm(x + dropout(r)) . w + b); mu / rstd (rows)
 * saved.  backward: dx, dr (rows,E) and the per-row terms gw = d logit * n, gg = dn * xhat, gb = dn, ds = d logit, whose
 * column sums (tag_colsum) are the gradients of linear.weight, norm.weight, norm.bias and linear.bias.
 * ------------------------------------------------------------------------------------------- */
int tag_mha_cross_forward(const float* q, const float* k, const float* v, const long* klen, float* attn, float* ctx,
                          int B, int T, int L, int E, int H, float drop_p, uint64_t seed, void* stream);
size_t tag_mha_cross_backward_ws_bytes(int B, int T, int L, int E);
int tag_mha_cross_backward(const float* q, const float* k, const float* v, const float* attn, const float* dctx,
                           const long* klen, float* dq, float* dk, float* dv, int B, int T, int L, int E, int H,
                           float drop_p, uint64_t seed, void* ws, void* stream);
int tag_resln_head_forward(const float* x, const float* r, const float* gamma, const float* beta, const float* w,
                           const float* bias, float* sim, float* mu, float* rstd, long rows, int E, float eps,
                           float drop_p, uint64_t seed, void* stream);
int tag_resln_head_backward(const float* x, const float* r, const float* gamma, const float* beta, const float* w,
                            const float* mu, const float* rstd, const float* sim, const float* dsim, float* dx, float* dr,
                            float* gw, float* gg, float* gb, float* ds, long rows, int E, float drop_p, uint64_t seed,
                            void* stream);

/* ---------------------------------------------------------------------------------------------
 * BASELINE configs[2]: bf16 ACTIVATION STORAGE.  The big tensors of the conv stack -- raw conv outputs, pooled block
 * outputs and the gradients of both -- are bf16 in HBM (void* here: raw 16-bit patterns, channels-last as before);
 * every product is bf16 x bf16 on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; BatchNorm statistics (from the fp32
 * accumulators / values before rounding), scale/shift, the per-channel gradient sums (fp64), weight gradients, the
 * GRU, the heads, the loss and the master weights stay fp32.  Outputs are rounded to nearest-even.  Same argument
 * meaning as the fp32 entry points of the same name (models/panns.py:46-62, models/audio_encoder.py:202-212).
 * tag_conv3x3_forward_x3_bf16 / tag_conv3x3_wgrad_x3_bf16 take the ONE-product weight pack
 * (tag_pack_conv_weight_x3(..., products = 1)).
 * ------------------------------------------------------------------------------------------- */
int tag_conv3x3_c1_forward_stats_bf16(const float* x, const float* col_scale, const float* col_shift, const float* w,
                                      void* y, float* stats /* nullable */, int B, int H, int W, int Cout, void* stream);
int tag_conv3x3_c1_backward_bf16(const float* x, const float* col_scale, const float* col_shift, const void* dy,
                                 const float* w, float* dw, float* dx, int B, int H, int W, int Cout, void* ws,
                                 void* stream);
int tag_conv3x3_c1_backward_bnrelu_bf16(const float* x, const float* col_scale, const float* col_shift, const void* da,
                                        const void* yref, const float* bn_scale, const float* bn_shift,
                                        const float* bn_mean, const float* bn_invstd, const float* gamma,
                                        const float* dgamma, const float* dbeta, int bn_train, const float* w, float* dw,
                                        float* dx, int B, int H, int W, int Cout, void* ws, void* stream);
int tag_conv3x3_forward_x3_bf16(const void* x, const void* wpack, int prologue, const float* in_scale,
                                const float* in_shift, void* y, float* stats, int B, int H, int W, int Cin, int Cout,
                                void* stream);
int tag_conv3x3_wgrad_x3_bf16(const void* x, int prologue, const float* in_scale, const float* in_shift, const void* dy,
                              float* dw, int B, int H, int W, int Cin, int Cout, void* ws, void* stream);
int tag_bnact_pool_forward_bf16(const void* y, const float* scale, const float* shift, void* out, int B, int H, int W,
                                int C, int ph, int pw, int act, int pool, float drop_p, uint64_t seed, void* stream);
int tag_bnrelu_pool_backward_bf16(const void* y, const float* scale, const float* shift, const float* mean,
                                  const float* invstd, const float* gamma, const void* dout, void* dy, float* dgamma,
                                  float* dbeta, int B, int H, int W, int C, int ph, int pw, int pool, float drop_p,
                                  uint64_t seed, int bn_train, void* ws, void* stream);
/* bf16 twins of tag_conv3x3_dgrad_bnsums / tag_bnrelu_backward_apply (BASELINE configs[2] mode): the sums are taken from the
 * fp32 accumulators of the dgrad conv before its output is rounded to bf16; bnpart rows [P][2][Cout],
 * P = tag_conv3x3_x3_stats_rows(B,H,W,Cout). */
int tag_conv3x3_dgrad_bnsums_bf16(const void* dy, const void* wpack, void* da, const void* yref, const float* bn_scale,
                                  const float* bn_shift, const float* bn_mean, const float* bn_invstd, float* bnpart,
                                  int B, int H, int W, int Cin, int Cout, void* stream);
int tag_bnrelu_backward_apply_bf16(const void* y, const float* scale, const float* shift, const float* mean,
                                   const float* invstd, const float* gamma, const void* da, void* dy,
                                   const float* dgamma, const float* dbeta, long rows, int C, int bn_train, void* stream);
int tag_bnrelu_backward_bf16(const void* y, const float* scale, const float* shift, const float* mean,
                             const float* invstd, const float* gamma, const void* da, void* dy, float* dgamma,
                             float* dbeta, long rows, int C, int bn_train, void* ws, void* stream);
int tag_mean_w_forward_bf16(const void* x, long rows, int W, int C, float drop_p, uint64_t seed, float* out, void* stream);
int tag_mean_w_backward_bf16(const float* dout, long rows, int W, int C, float drop_p, uint64_t seed, void* dx,
                             void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TAG_HIP_H */
