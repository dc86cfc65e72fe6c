/* dusty_gan_hip.h -- C ABI of libdustygan_hip.so (MI355X / gfx950 only).
 *
 * The drop-in boundary of the dusty-gan training hot path.  The reference (kazuto1011/dusty-gan) has no native
 * code on this path: every entry point below replaces a PyTorch module call (or its autograd backward) that the
 * reference's `Trainer.step` (trainers/dcgan_amp.py:162-325) makes.  The file:line next to each declaration is the
 * reference interface it replaces.  A maintainer binds these with ctypes; PyTorch is only the
 * owner of device memory and streams.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; nothing is allocated inside the library,
 *     workspaces are caller-owned; `stream` is a hipStream_t passed as void*;
 *   - return value: DG_OK (0) or a DG_E* code; kernels are launched asynchronously on `stream`;
 *   - images are fp32 [B,1,H,W]; feature maps are pixel-major/channel-minor ("NHWC") in `dtype` (DG_F32|DG_BF16);
 *   - conv weights: engine master fp32 [ky][kx][ci][co] ("cico"); the reference's (Cout,Cin,4,4) / (Cin,Cout,4,4)
 *     parameter tensors are strided VIEWS of that storage (no copy), so state_dict()/checkpoints keep the
 *     reference's shapes.
 */
#ifndef DUSTY_GAN_HIP_H
#define DUSTY_GAN_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DG_OK 0
#define DG_EINVAL 1
#define DG_EUNSUPPORTED 2
#define DG_EHIP 3

#define DG_F32 0
#define DG_BF16 1
/* DG_BF16X2 (round 5, the storage form of the "fp32x3" precision mode on the bf16 kernels): every element x is the PAIR
 * hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits kept), 4 bytes per element like DG_F32 and addressed with the same element
 * strides, laid out per 64 channels as 128 bytes of hi followed by 128 bytes of lo: element i (flat element offset from a
 * 256-byte aligned base, channel-minor, channel count a multiple of 64) has hi at bf16 index 2 i - i % 64 and lo 64 further.
 * A matrix-core kernel then contracts x . w as x_hi w_hi + x_lo w_hi + x_hi w_lo with the bf16 kernels' own 128-byte channel
 * chunks (three K steps per real one) - no splitting in registers, no fp32 staging.  Taken by: the ping-pong conv (in, w, out,
 * aux all DG_BF16X2), the LDS-DMA weight-gradient kernel (a and g), the direct and the thin VALU kernels (any operand),
 * dg_cast / dg_uncast.  Everything else answers DG_EUNSUPPORTED. */
#define DG_BF16X2 2

/* conv-like op geometry: a COARSE grid Hc x Wc and a FINE grid 2Hc x 2Wc */
#define DG_MODE_S2 0   /* out on coarse, in on fine : Down forward, Up backward-data          */
#define DG_MODE_UP 1   /* out on fine, in on coarse : Up/Head forward, Down backward-data     */
#define DG_MODE_GEMM 2 /* plain rows (Proj)                                                   */

#define DG_EPI_LINEAR 0 /* out = scale*acc + bias                                             */
#define DG_EPI_LRELU 1  /* FusedLeakyReLU: leaky_relu(scale*acc + bias, 0.2) * sqrt(2)        */
#define DG_EPI_MASK 2   /* out = scale*acc * lrelu'(aux) * sqrt(2)  (backward / R1 tangent)   */

/* Parameter block of dg_conv.  Strides are in ELEMENTS. */
typedef struct DgConv {
  int mode, adj, ring;       /* adj: 0 = the reference's Pad (reflect rows, circular|reflect cols), 1 = its adjoint */
  int B, Hc, Wc;
  int K, N;                  /* contraction channels, output channels */
  const void* in;
  long in_sb, in_sp, in_sk;  /* batch, pixel (row*W+col), channel */
  void* out;
  long out_sb, out_sp, out_sn;
  const void* w;
  long w_st, w_sk, w_sn;     /* tap (ky*4+kx), k, n */
  float scale;               /* EqualLR runtime scale, models/ops/common.py:124-125,133 */
  int epi;
  const float* bias;         /* fp32 [bias_mod] or NULL; index n % bias_mod */
  int bias_mod;
  const void* aux;           /* DG_EPI_MASK: saved activation with out's layout/dtype */
  float* dbias;              /* optional: dbias[n % bias_mod] += rowscale[b] * sum_pixels out */
  const float* rowscale;     /* optional per-sample weight of the dbias sum */
  int in_dtype, out_dtype, w_dtype;
  const float* nscale;       /* optional per-output-channel scale (Head: one EqualLR scale per head) */
  const void* up_frag;       /* optional, 16-byte aligned, DG_UP_FRAG_BYTES: the weight fragments of the thin matrix-core
                              * MODE_UP kernel (Head forward, Down1 backward-data) kept current by the caller with
                              * dg_transpose_shadow_multi_frags; NULL: built by a launch in front of the kernel */
  float* dbias_ws;           /* optional scratch of DG_DBIAS_SLOTS x DG_DBIAS_SLOT_FLOATS floats, zero on entry and left zero:
                              * per-slot staging of the bias-gradient rows of the thin matrix-core MODE_S2 kernel (hundreds of
                              * atomic rows on the same two lines serialise memory-side) and, since round 6, the
                              * cross-workgroup staging of the direct and one-tile MFMA kernels' bias-gradient sums - two 64-bit
                              * fixed-point words per channel + a ticket, bias_mod <= 8191 - which makes those sums independent
                              * of the arrival order (without it they are float atomics); other kernels ignore it.  One launch
                              * at a time per scratch: launches of one stream may share it */
  /* Saved leaky-relu masks, 1 bit per element (models/ops/common.py:99-106 under autograd saves the sign of the
   * pre-activation; the backward / R1 tangent passes only ever need that sign).  Bit e of a mask buffer belongs to the
   * element at offset e of the tensor it describes (bit e % 8 of byte e / 8): the mask of `out` is indexed like `out`, the
   * mask of `aux` like `aux`.  Needs out_sn == 1 and N, out_sp, out_sb multiples of 8 (16 for the matrix-core kernels). */
  float* dbias_part;         /* optional, with dbias (round 5, bit-reproducible bias gradients): kernels that take it
                              * (DgConvPlan.dbias_rows > 0: the ping-pong conv, the thin matrix-core MODE_S2 kernel, the lock-step
                              * persistent conv at fp32) store ONE
                              * partial row of N floats per workgroup at dbias_part + row * N - summed inside the workgroup in a
                              * fixed order - and add nothing to dbias; the caller sums the rows with
                              * dg_wgrad_reduce(ws = dbias_part, dw = dbias, numel = N, splits = dbias_rows).  Kernels that do
                              * not take it ignore it and add onto dbias with atomics as before.  The rows are NOT folded:
                              * row element n is the sum for output channel n, so the form needs bias_mod == N - the kernels that take
                              * rows refuse (DG_EUNSUPPORTED) a dbias launch with bias_mod < N, with or without dbias_part. */
  void* mask_out;            /* optional, DG_EPI_LRELU: also store bit = (out element > 0) for every element written.  The
                              * ping-pong conv and the thin matrix-core MODE_S2 kernel write it from their epilogues, behind any
                              * other kernel the library adds one packing launch: the bits are there when dg_conv returns OK */
  const void* mask_in;       /* optional, DG_EPI_MASK: the bits of `aux` (written through mask_out by the launch that produced
                              * aux); kernels that take bits (DgConvPlan.mask_bits & 2) read 1/16 of the bytes, the others use aux,
                              * which stays mandatory */
  float* tanh_sum_parts;     /* optional (round 6), the thin matrix-core MODE_UP kernel only (DgConvPlan.sum_parts > 0: N == 1,
                              * DG_EPI_LINEAR, fp32 output - the depth head, models/gans/dcgan_eqlr.py:29-46): the launch writes
                              * tanh(out) instead of out (Generator.forward's torch.tanh, :71) and every workgroup STORES the sum
                              * of what it wrote at tanh_sum_parts[its index]; sample b's workgroups are the sum_parts indices
                              * from b * sum_parts: the per-sample sums DiffAugment's contrast needs (DgAugSet.xsum_parts), made
                              * where the image is made - no head post-processing launch for the baseline generator.  Kernels
                              * that do not take it (sum_parts == 0) ignore it: the caller then runs dg_head_post_fwd[_sum]. */
} DgConv;
#define DG_UP_FRAG_BYTES (3 * 18 * 1024)
#define DG_DBIAS_SLOTS 32
#define DG_DBIAS_SLOT_FLOATS 1024

/* Parameter block of dg_wgrad:  dw[tap][ci][co] += scale * sum_b rowscale[b] * sum_pixels a[..][ci] * g[..][co] */
typedef struct DgWgrad {
  int wmode, ring;           /* 0 = Down, 1 = Up/Head, 2 = plain rows (Proj: set B=1,Hc=1,Wc=batch) */
  int B, Hc, Wc;
  int Ci, Co;
  const void* a;             /* the layer's forward input */
  long a_sb, a_sp, a_sc;
  const void* g;             /* gradient w.r.t. the layer's pre-activation output */
  long g_sb, g_sp, g_sc;
  float* dw;                 /* fp32 [tap][Ci][Co] */
  float scale;
  const float* rowscale;     /* optional [B] */
  int a_dtype, g_dtype;
  /* Optional (zero = off).  ws: caller-owned workspace for the split-K partial tiles of the MFMA LDS-DMA kernel: split s
   * stores its [16][Ci][Co] partial at ws + s * 16 Ci Co with plain stores and NOTHING is added to dw - the caller sums
   * the partials with dg_wgrad_reduce (sizes from dg_wgrad_plan).  Without it the partial tiles meet in dw through fp32
   * atomics, which the memory side executes at ~1.3 TB/s against ~6 TB/s for stores (8x the bytes of dw per launch) - and in
   * arrival order.  Since round 6 every kernel that splits the reduction over workgroups has the form (the thin kernels, the
   * register-staged MFMA kernel of the fp32 modes, the direct kernel: numel = Ci Co for wmode 2); dg_wgrad_plan says which
   * launch takes it (splits > 1), a workspace handed to a launch that does not is DG_EUNSUPPORTED.
   * g_mod: the gradient's sample index is b % g_mod, so ONE launch over 3B input samples (real | fake | R1 tangent) can
   * pair the tangent rows with the real batch's gradient chain again (trainers/dcgan_amp.py:229-235). */
  float* ws;
  int g_mod;
} DgWgrad;

/* ---- conv-like passes ------------------------------------------------------------------------------------
 * dg_conv replaces, depending on (mode, adj, epi):
 *   Down.forward            models/gans/dcgan_eqlr.py:75-82  (Pad :77 + EqualLR(Conv2d 4,2,0) :80 + FusedLeakyReLU :81)
 *   Up.forward              models/gans/dcgan_eqlr.py:19-26  (Pad + EqualLR(ConvTranspose2d 4,2,3) + FusedLeakyReLU)
 *   Head.forward            models/gans/dcgan_eqlr.py:29-46
 *   Proj.forward            models/gans/dcgan_eqlr.py:6-16
 *   their autograd backward-data passes (loss.backward() at trainers/dcgan_amp.py:235,309 and
 *   torch.autograd.grad(create_graph=True) at :218-223), and the R1 tangent pass (double backward, :229-235).
 * force: 0 = pick (MFMA implicit GEMM when the shape allows, else the thin LDS/VALU kernel for <=4-channel sides,
 * else the general direct kernel), 1 = direct, 2 = MFMA or error, 3 = thin or error, 4 / 5 = a persistent large-tile
 * MFMA kernel or error (4 the lock-step one, 5 the bf16 ping-pong one; what 0 / 2 pick for layers that fill the chip
 * with such tiles - the forced forms let parity tests run either family on small problems).
 */
int dg_conv(const DgConv* p, int force, void* stream);
/* "fp32x3" (precision contract: fp32 storage with fp32 or split-bf16 x 3 MFMA): a flag bit in the
 * `force` argument of dg_conv / dg_conv_ex / dg_conv_plan / dg_wgrad / dg_wgrad_plan, OR-ed onto the kernel-family code, for
 * DG_F32 operands on the matrix-core kernels.  Clear (default): exact fp32 products (v_mfma_f32_32x32x2_f32, 1/16 of the bf16
 * rate); set: every operand is split into bf16 hi + lo in registers and a product is a_hi b_hi + a_lo b_hi + a_hi b_lo on
 * the bf16 matrix instructions with fp32 accumulation (relative error ~2^-16 per product; the mode autocast-free fp32
 * training of trainers/dcgan_amp.py would want on this hardware).  Per call - nothing process-wide: two engines of
 * different precision in one process do not share a setting. */
#define DG_FORCE_FP32X3 0x100
/* What a dg_conv call launches (introspection for the parity tests and the benchmark: which kernel family / tile ran,
 * and how many tiles each persistent workgroup walks).  family: 1 direct, 2 one-tile-per-workgroup MFMA, 3 thin,
 * 4 persistent large-tile MFMA (lock step: fp32, small layers), 5 persistent ping-pong MFMA (bf16 fat layers),
 * 6 weight-streaming Proj forward (bf16 DG_MODE_GEMM with K = 512, B <= 32; dg_conv force 10 asks for it).  dg_conv_ex = dg_conv with a cap on the persistent kernel's workgroup count
 * (wg_cap <= 0: one residency wave of the device); dg_conv_plan fills `plan` for the same arguments and launches
 * nothing. */
typedef struct DgConvPlan {
  int family;
  int bm, bn;        /* output tile (pixels x channels), 0 for the non-MFMA kernels */
  int tiles;         /* tiles of the launch */
  int workgroups;    /* grid size */
  int tiles_per_wg;  /* most tiles any workgroup walks */
  int thin_mfma;     /* family 3: 1 = thin_s2_mfma, 2 = thin_up_mfma (matrix cores), 0 = the VALU kernels */
  int mask_bits;     /* 1: the kernel writes DgConv.mask_out itself (else a packing launch follows it), 2: it reads mask_in */
  int dbias_rows;    /* > 0: the kernel takes DgConv.dbias_part and writes this many partial rows of N floats; 0: it does not */
  int sum_parts;     /* > 0: the kernel takes DgConv.tanh_sum_parts and stores this many partial sums per sample; 0: it does not */
} DgConvPlan;
int dg_conv_ex(const DgConv* p, int force, int wg_cap, void* stream);
int dg_conv_plan(const DgConv* p, int force, int wg_cap, DgConvPlan* plan);
int dg_conv_mfma_supported(const DgConv* p);
int dg_conv_kernel_choice(const DgConv* p);   /* what force == 0 launches: 2 MFMA, 3 thin, 1 direct */

/* dg_wgrad replaces the weight-gradient half of the same autograd calls.  accumulate: 1 = atomically add onto dw
 * (dw zeroed by the caller at step start: optim.zero_grad, trainers/dcgan_amp.py:177,246), 0 = overwrite. */
int dg_wgrad(const DgWgrad* p, int accumulate, int force, void* stream);
/* What dg_wgrad launches for these arguments, and the split-K workspace it can use: `splits` partial tiles of
 * 16 Ci Co floats (ws_floats = splits * 16 Ci Co; 0 when the kernel that runs has no workspace form).  force 7 / 8:
 * the LDS-DMA kernel with / without W-tap pairs (A/B measurements; 0 / 2 choose by the K range per workgroup). */
typedef struct DgWgradPlan {
  int variant;      /* as dg_wgrad_kernel_variant */
  int splits;
  long ws_floats;
  int tap_pairs;    /* 1: one workgroup computes the W taps (kx, kx + 2) from one staged image */
} DgWgradPlan;
int dg_wgrad_plan(const DgWgrad* p, int accumulate, int force, DgWgradPlan* plan);
/* Up to 4 weight-gradient GEMMs as ONE launch (the layers of one network: independent of each other, all reading finished
 * activations and gradient chains - loss.backward() at trainers/dcgan_amp.py:235,309 produces them in one sweep too): every
 * item must run on the MFMA LDS-DMA kernel (dg_wgrad_plan: variant 5) and bring its split-K workspace `ws`, sized by
 * dg_wgrad_group_plan for the same (items, force, rounds).  One grid instead of n residency rounds: the ring fill and the
 * partial-tile stores of one layer's workgroups run under the matrix work of its neighbours'.  rounds > 0: the group as a
 * whole aims at rounds x 512 workgroups, shared among the items by their FLOPs - fewer, longer K ranges per item than a launch
 * of its own would use and proportionally fewer partial tiles; rounds <= 0: every item keeps the geometry (splits, ws_floats)
 * of its own dg_wgrad_plan.  The caller sums the partials with dg_wgrad_reduce as for single launches.  DG_EUNSUPPORTED
 * (nothing launched) if an item does not qualify. */
int dg_wgrad_group(const DgWgrad* items, int n, int force, int rounds, void* stream);
int dg_wgrad_group_plan(const DgWgrad* items, int n, int force, int rounds, DgWgradPlan* plans);
/* dw[i] (+)= sum_s ws[s * numel + i] for up to 16 layers in one launch (fixed summation order: deterministic gradients) */
typedef struct DgWgradReduce {
  const float* ws;
  float* dw;
  long numel;       /* 16 Ci Co, a multiple of 4 */
  int splits;
  int accumulate;   /* 1: add onto dw, 0: overwrite */
} DgWgradReduce;
int dg_wgrad_reduce(const DgWgradReduce* items, int n, void* stream);
int dg_wgrad_mfma_supported(const DgWgrad* p);
int dg_wgrad_kernel_choice(const DgWgrad* p);
/* which kernel `force` launches: 5 MFMA on the LDS-DMA ring (bf16 Down/Up layers), 2 register-staged MFMA (also
 * force == 6), 7 thin on the matrix cores (bf16 Down1 / Head), 3 thin (VALU), 1 direct, 0 unsupported */
int dg_wgrad_kernel_variant(const DgWgrad* p, int force);
/* 1 when the kernel `force` launches honours DgWgrad.g_mod (the LDS-DMA kernel; Down1's thin matrix-core kernel) */
int dg_wgrad_has_sample_map(const DgWgrad* p, int force);

/* ---- BlurVH  models/ops/common.py:74-88 (forward) and its adjoint --------------------------------------- */
int dg_blur_fwd(const float* x, void* out, int dtype, int B, int H, int W, int ring, void* stream);
/* + mean_acc[0] += mean(mean_src[0..mean_n)) (dg_mean_acc) in the same launch: the R1 penalty's logged value
 * (trainers/dcgan_amp.py:229) rides on the tangent's BlurVH pass.  DG_EUNSUPPORTED unless W % 4 == 0 and 16-byte aligned */
int dg_blur_fwd_mean(const float* x, void* out, int dtype, int B, int H, int W, int ring, const float* mean_src, int mean_n,
                     float* mean_acc, void* stream);
int dg_blur_bwd(const void* d, int dtype, float* dx, int B, int H, int W, int ring, void* stream);
/* the same adjoint at the end of the R1 chain (trainers/dcgan_amp.py:218-235): dx = oscale * g and ssq[b] += sum of g_b^2
 * (ssq zeroed by the caller) in one pass instead of dg_blur_bwd + dg_sample_sum + dg_scale; DG_EUNSUPPORTED unless W % 4 == 0
 * and H W % 1024 == 0 */
int dg_blur_bwd_r1(const void* d, int dtype, float* dx, float oscale, float* ssq, int B, int H, int W, int ring, void* stream);
/* ... and with the R1 tangent's BlurVH in the SAME launch (round 6): out[b] = BlurVH(oscale * g_b) [B,H,W,2] in `dtype`,
 * g_b = BlurVH^T(d[b]) never written; ssq[b] += |g_b|^2 and (mean_acc non-NULL) mean_acc[0] += sum_b |g_b|^2 / mean_n - the
 * logged penalty, trainers/dcgan_amp.py:229 - both zeroed by the caller.  The two launches it replaces: dg_blur_bwd_r1 +
 * dg_blur_fwd_mean; same arithmetic in the same order, so out is bit-identical to theirs.  DG_EUNSUPPORTED (nothing launched)
 * unless W % 4 == 0, H % 4 == 0 and six image rows fit 60 KB of LDS. */
int dg_blur_r1_tangent(const void* d, int dtype, void* out, float oscale, float* ssq, float* mean_acc, int mean_n, int B, int H,
                       int W, int ring, void* stream);

/* ---- final EqualLR(Conv2d(C,1,(h0,w0)))  models/gans/dcgan_eqlr.py:95 ------------------------------------ */
int dg_final_fwd(const void* d4, int dtype, const float* wf, const float* bias, float scale, int B, long n, float* y,
                 void* stream);
/* `_acc` forms (dg_final_fwd_acc, dg_sample_sum_acc, dg_diffaug_fwd_acc, dg_diffaug_bwd_acc): the fp32 accumulator the
 * call adds into (y / out / xsum / gsum) was zeroed by the CALLER - a step that carves all of them from one buffer zero-fills
 * once per step instead of once per call */
int dg_final_fwd_acc(const void* d4, int dtype, const float* wf, const float* bias, float scale, int B, long n, float* y,
                 void* stream);
/* dd4[b] = up[b]*scale*wf*lrelu'(d4[b])*sqrt2 ; dbias4[i%C] += rowscale[b]*dd4[b][i] */
int dg_final_bwd_data(const void* d4, int dtype, const float* wf, const float* up, const float* rowscale, float scale,
                      int B, long n, int C, void* dd4, float* dbias, void* stream);
/* out[i] += scale * sum_b coef[b]*src[b][i] */
int dg_batch_wsum(const void* src, int dtype, const float* coef, float scale, int B, long n, float* out, void* stream);

/* ---- Generator tanh + DUSty maskout  models/gans/dcgan_eqlr.py:71; models/dusty.py:45-59,77-91,107-127 ---- */
/* gout [B,1+k,H,W] planar fp32 (ch0 raw depth -> tanh in place, ch1.. logits); arch 0 none, 1 dusty1, 2 dusty2 */
int dg_head_post_fwd(float* gout, const float* noise_pixel, const float* noise_image, int arch, int training,
                     float tau, float drop_const, int B, long HW, float* mask, float* depth, void* stream);
/* + dsum[b] += sum of depth[b] (dsum zeroed by the caller; HW % 256 == 0 or DG_EUNSUPPORTED): the per-sample sums DiffAugment's
 * contrast needs of its input, produced where the image is produced (dg_diffaug_fwd_pre then skips its own pass) */
int dg_head_post_fwd_sum(float* gout, const float* noise_pixel, const float* noise_image, int arch, int training,
                         float tau, float drop_const, int B, long HW, float* mask, float* depth, float* dsum, void* stream);
/* draw[n] = s_n * d(loss)/d(head output n) (s_depth for ch0, s_conf for the logits: the EqualLR scale of each head,
 * pre-multiplied so the head's backward-data / weight-gradient passes run with scale 1); dbias[n] += unscaled sums.
 * `draw` (planar fp32) may be NULL when the pixel-major bf16 copy is the only consumer (cp 2 / 4, HW % 4 == 0, 16-byte
 * aligned planes; DG_EUNSUPPORTED otherwise, DG_EINVAL when both are NULL).  `bias_ws` (optional, 1024 floats per sample,
 * zero on entry, left zero): per-sample staging of the bias sums - a thousand atomics on one address serialise at
 * ~10 ns each, B slots take them in parallel and the last block of each sample folds its slot into dbias */
int dg_head_post_bwd(const float* gout, const float* noise_pixel, const float* noise_image, const float* mask,
                     const float* ddepth, int arch, float tau, float drop_const, int B, long HW, float s_depth,
                     float s_conf, float* draw, float* dbias, void* draw_pm /* optional bf16 [B,H,W,cp] copy */, int cp,
                     float* bias_ws, void* stream);
/* GumbelSigmoid.logistic_noise  models/dusty.py:30-36 */
int dg_logistic_noise(const float* u1, const float* u2, float eps, long n, float* out, void* stream);

/* ---- DiffAugment.forward  utils/diff_augment.py:114-132 (p = 1) and its backward ------------------------- */
/* policy bits: 1 brightness, 2 saturation, 4 contrast, 8 translation, 16 cutout; per-sample draws are arguments */
int dg_diffaug_fwd(const float* x, const float* u_b, const float* u_c, const int* t_h, const int* t_w, const int* o_x,
                   const int* o_y, int policy, int B, int H, int W, float* xsum_ws, float* y, void* stream);
int dg_diffaug_fwd_acc(const float* x, const float* u_b, const float* u_c, const int* t_h, const int* t_w, const int* o_x,
                   const int* o_y, int policy, int B, int H, int W, float* xsum_ws, float* y, void* stream);
/* xsum already holds the per-sample sums of x */
int dg_diffaug_fwd_pre(const float* x, const float* u_b, const float* u_c, const int* t_h, const int* t_w, const int* o_x,
                       const int* o_y, int policy, int B, int H, int W, const float* xsum, float* y, void* stream);
int dg_diffaug_bwd(const float* gy, const float* u_b, const float* u_c, const int* t_h, const int* t_w, const int* o_x,
                   const int* o_y, int policy, int B, int H, int W, float* gsum_ws, float* gx, void* stream);
int dg_diffaug_bwd_acc(const float* gy, const float* u_b, const float* u_c, const int* t_h, const int* t_w, const int* o_x,
                   const int* o_y, int policy, int B, int H, int W, float* gsum_ws, float* gx, void* stream);
/* DiffAugment.forward + BlurVH.forward in one pass (utils/diff_augment.py:114-132 -> models/ops/common.py:74-88, the first
 * two modules every D(A(x)) call runs, trainers/dcgan_amp.py:199-204,255-260): the augmented image is only ever D's input,
 * so it is not written - out = BlurVH(A(x)) [nsets B, H, W, 2] in `dtype`.  One or two source sets per launch (the D
 * phase's real | fake halves); xsum = the per-sample sums of x (dg_fetch_reals_sum / dg_head_post_fwd_sum), read by the
 * contrast stage.  DG_EUNSUPPORTED unless W % 4 == 0, H % 4 == 0 (a workgroup owns a band of four output rows) and six image
 * rows fit 60 KB of LDS. */
#define DG_XSUM_PARTS 8       /* partial sums per sample where a producer leaves them un-summed (dg_step_prologue_fetch) */
typedef struct DgAugSet {
  const float* x;            /* [B,1,H,W] fp32 */
  const float* xsum;         /* [B], or with xsum_parts = DG_XSUM_PARTS: [B][DG_XSUM_PARTS] partial sums, added in index order */
  int xsum_parts;            /* 0 / 1: one float per sample */
  const float *u_b, *u_c;    /* [B] the uniform(-1,1) draws of brightness / contrast */
  const int *t_h, *t_w, *o_x, *o_y;
} DgAugSet;
int dg_diffaug_blur_fwd(const DgAugSet* sets, int nsets, int policy, int B, int H, int W, int ring, void* out, int dtype,
                        void* stream);
/* The adjoint pair BlurVH^T -> DiffAugment^T of the G phase (loss_G.backward() through :256-260) in two launches instead
 * of three: dg_blur_bwd_augsum is dg_blur_bwd that also accumulates gsum[b] += the sum of dx[b] over the rows / columns
 * whose gradient reaches the source image (gsum zeroed by the caller; DG_EUNSUPPORTED unless W % 4 == 0), and
 * dg_diffaug_bwd_pre is dg_diffaug_bwd with that sum given. */
int dg_blur_bwd_augsum(const void* d, int dtype, float* dx, const int* t_h, const int* o_x, const int* o_y, int policy,
                       float* gsum, int B, int H, int W, int ring, void* stream);
int dg_diffaug_bwd_pre(const float* gy, const float* u_b, const float* u_c, const int* t_h, const int* t_w, const int* o_x,
                       const int* o_y, int policy, int B, int H, int W, const float* gsum, float* gx, void* stream);
/* dg_diffaug_bwd_pre + dg_head_post_bwd as one launch: the generator's upstream gradient (DiffAugment's adjoint gather of
 * gy) is evaluated per pixel quad inside the head post-processing's backward and never written
 * (loss_G.backward() through utils/diff_augment.py:114-132 into models/dusty.py:77-91,107-127).  DG_EUNSUPPORTED - nothing
 * launched - unless W % 4 == 0, the planes are 16-byte aligned and cp is 2 / 4 (or draw_pm is NULL). */
int dg_head_post_bwd_aug(const float* gout, const float* noise_pixel, const float* noise_image, const float* mask,
                         const float* gy, const float* u_b, const float* u_c, const int* t_h, const int* t_w,
                         const int* o_x, const int* o_y, int policy, const float* gsum, int arch, float tau,
                         float drop_const, int B, int H, int W, float s_depth, float s_conf, float* draw, float* dbias,
                         void* draw_pm, int cp, float* bias_ws, void* stream);

/* ---- GANLoss(nsgan)  models/loss.py:39-41,68-69 + gradient w.r.t. the logits ----------------------------- */
/* scal[0]=mean(y_real) scal[1]=mean(y_fake) scal[2]=loss_D */
int dg_nsgan_d(const float* y_real, const float* y_fake, int B, float w_gan, float* dy_real, float* dy_fake,
               float* scal, void* stream);
int dg_nsgan_g(const float* y_fake, int B, float w_gan, float* dy, float* scal, void* stream);
/* The same losses with the step's bookkeeping folded in (trainers/dcgan_amp.py:203-238,305-309): dy [2B] = dLoss/dy
 * (real | fake); up [2B] = [1..1 | dy_fake] and rs [2B] = [dy_real | 1..1] (the R1 schedule's per-sample vectors, either
 * may be NULL); acc[0..2] += (mean y_real, mean y_fake, loss_D); *dfinal_b += sum dy (the final conv's bias gradient).
 * dg_nsgan_g_step: acc[0] += loss_G.  dg_mean_acc: acc[0] += mean(x[0..n)). */
int dg_nsgan_d_step(const float* y_real, const float* y_fake, int B, float w_gan, float* dy, float* up, float* rs,
                    float* acc, float* dfinal_b, void* stream);
int dg_nsgan_g_step(const float* y_fake, int B, float w_gan, float* dy, float* acc, void* stream);

/* ---- GANLoss, all metrics  models/loss.py:39-61 (loss_D), :66-85 (loss_G) ------------------------------------
 * metric: DG_GAN_* below (the reference's `solver.gan_mode` strings in models/loss.py order).  Same outputs as the
 * nsgan entries above: D step writes dy = [d/dy_real | d/dy_fake] of w_gan * loss_D, the R1 schedule's `up` / `rs`
 * vectors (nullable), acc[0..2] += (mean y_real, mean y_fake, loss_D), dfinal_b += sum(dy) (nullable).  G step writes
 * dy = d(w_gan * loss_G)/dy_fake and acc[0] += loss_G; y_real = D(real) logits, required by the relativistic
 * metrics (ragan / rahinge / ralsgan, average_diff models/loss.py:11-18) and ignored (may be NULL) otherwise.
 * `smoothing` = GANLoss.smoothing (lsgan's real label, models/loss.py:46).  DG_EUNSUPPORTED for an unknown metric. */
enum { DG_GAN_NSGAN = 0, DG_GAN_WGAN = 1, DG_GAN_LSGAN = 2, DG_GAN_HINGE = 3, DG_GAN_RAGAN = 4, DG_GAN_RAHINGE = 5,
       DG_GAN_RALSGAN = 6 };
int dg_gan_d_step(int metric, float smoothing, const float* y_real, const float* y_fake, int B, float w_gan, float* dy,
                  float* up, float* rs, float* acc, float* dfinal_b, void* stream);
int dg_gan_g_step(int metric, const float* y_real, const float* y_fake, int B, float w_gan, float* dy, float* acc,
                  void* stream);
/* The loss step and the final conv's backward as ONE launch: dg_gan_d_step (mode_g = 0; r1 = 1: the R1 schedule's up / rs
 * vectors drive the chain, r1 = 0: upstream dy, unit weights) or dg_gan_g_step (mode_g = 1), then dg_final_bwd_data over
 * the 2B (B) samples with those vectors, and - when dwf is given - the final conv's weight gradient
 * dwf[i] += scale * sum_b dy[b] d4[b][i] (dg_batch_wsum) from the same read of d4: `loss.backward()`'s first links,
 * trainers/dcgan_amp.py:203-235 and :259-309.  DG_EUNSUPPORTED (nothing launched) when the vector forms do not apply or
 * more than 256 samples are in the pass.  dbias_part (optional, n floats, with dbias): instead of atomics onto dbias the launch
 * stores one partial per element of the map (summed over the samples in a fixed order) and the caller sums the n / C rows
 * per channel: dg_wgrad_reduce(ws = dbias_part, dw = dbias, numel = C, splits = n / C) - bit-reproducible bias gradients. */
int dg_final_gan_bwd(int metric, int mode_g, float smoothing, const float* y_real, const float* y_fake, int B, float w_gan,
                     int r1, float* dy, float* up, float* rs, float* acc, float* dfinal_b, const void* d4, int dtype,
                     const float* wf, float scale, long n, int C, void* dd4, float* dbias, float* dwf, float* dbias_part,
                     void* stream);
int dg_mean_acc(const float* x, int n, float* acc, void* stream);
/* Bit-reproducible cross-block sums (round 5).  Registers, for the current device, the arena of small fp32 accumulators the
 * step zero-fills once (per-sample image sums, logits, the augment adjoint's window sums: dg_*_acc / dg_*_sum entry points
 * accumulate into slices of it) together with a shadow of 128 bytes per arena float (a cache line each: adds to one line
 * serialise memory-side), zero at rest: kernels whose blocks add
 * partial sums into an arena slot then add them as 32.32 fixed point to the slot's shadow word (integer addition commutes)
 * and the last block to arrive converts the total - instead of float atomics, whose result depends on the arrival order.
 * arena = shadow = NULL unregisters (float atomics again).  Nothing in the reference to replace: torch's reductions are
 * deterministic on the CPU and the reference never asks for determinism on the GPU. */
int dg_det_arena(float* arena, long n, void* shadow);

/* ---- path-length regularisation  trainers/dcgan_amp.py:268-306 (the parts that are not convolutions) ------------
 * The penalty needs d/dtheta of |d(sum x y)/dz|: a forward-over-reverse pass built from dg_conv / dg_wgrad (the
 * reverse pass ends with dz^T = W^T dp0^T, a dg_wgrad of wmode 2) plus these two.  dg_pl_penalty: lengths |dz_b|, the running baseline pl_ema (device scalar,
 * updated: lerp(.., 0.01) :297), the penalty (:300) and v [B][K] = w * d penalty / d dz; acc[0] += baseline,
 * acc[1] += penalty.  dg_head_post_bwd2: tangent of dg_head_post_bwd along the head tangent `thead` [B,1+k,H,W] (the
 * Hessian of tanh / Gumbel straight-through maskout, models/dusty.py:77-91,107-127), same outputs as dg_head_post_bwd. */
int dg_pl_penalty(const float* dz, int B, int K, float w, float* pl_ema, float* v, float* acc, void* stream);
int dg_head_post_bwd2(const float* gout, const float* noise_pixel, const float* noise_image, const float* mask,
                      const float* ddepth, const float* thead, int arch, float tau, float drop_const, int B, long HW,
                      float s_depth, float s_conf, float* draw, float* dbias, void* draw_pm, int cp, void* stream);

/* ---- Trainer.fetch_reals  trainers/dcgan_amp.py:154-160 (utils/lidar.py:31-36, utils/__init__.py:70-73) -- */
int dg_fetch_reals(const float* pol, const float* mask, float min_depth, float max_depth, float drop_const, long n,
                   float* out, void* stream);
int dg_fetch_reals_sum(const float* pol, const float* mask, float min_depth, float max_depth, float drop_const, int B,
                       long HW, float* out, float* xsum, void* stream); /* + xsum[b] += sum of out[b] (as dg_head_post_fwd_sum) */
/* the same from a device-resident pool of `npool` batches (the synthetic loader): the batch index is
 * *pool_ctr % npool, read ON THE DEVICE, so a training step captured in a hipGraph replays on the next batch without a
 * host-side copy into a static buffer (next(loader) at trainers/dcgan_amp.py:186) */
int dg_fetch_reals_pool_sum(const float* pol_pool, const float* mask_pool, const unsigned long long* pool_ctr, int npool,
                            float min_depth, float max_depth, float drop_const, int B, long HW, float* out, float* xsum,
                            void* stream);

/* ---- data formats either side of the step ---------------------------------------------
 * dg_scan_to_polar: KITTIOdometry.preprocess + .transform  datasets/kitti.py:54-77, optionally fused with
 * Trainer.fetch_reals (trainers/dcgan_amp.py:154-160).  scan [B,Hs,Ws,C] fp32 = the projected scans written by
 * process_kitti.py:76-118 (x, y, z[, reflectance] per cell, C >= 3; C == 4 is read with 16-byte loads); flip [B]
 * bytes (NULL = no horizontal flip; TF.hflip precedes the resize); H x W = cfg.dataset.shape, NEAREST resize =
 * torch's legacy nearest (src = min(floor(dst * in / out), in - 1)).  Outputs (NCHW fp32): pol [B,1,H,W] =
 * (|xyz| - min) / (max - min), mask [B,1,H,W] in {0,1} = |xyz| > 0 & > min & < max, invalid cells zeroed;
 * xyz [B,3,H,W] = xyz / max_depth (nullable); x_real [B,1,H,W] = the network input fetch_reals would make of
 * (pol, mask) with drop_const (nullable).
 * dg_inv_to_xyz: utils.postprocess's depth branch + Coordinate.inv_to_xyz  utils/__init__.py:163-178,
 * utils/lidar.py:38-68.  in [B,1,H,W]: generator depth in [-1,1] (from_tanh = 1: tanh_to_sigmoid + clamp first) or
 * inverse depth in [0,1]; angle [2,H,W] = (elevation, azimuth) grid of LiDAR.init_coordmap (:127-130);
 * drop_const / tol as Coordinate (0 / 1e-8 by default).  depth01 [B,1,H,W] (nullable) receives the [0,1] inverse
 * depth, points [B,3,H,W] the unit-space point map. */
int dg_scan_to_polar(const float* scan, int B, int Hs, int Ws, int C, int H, int W, const unsigned char* flip,
                     double min_depth, double max_depth, float drop_const, float* pol, float* mask, float* xyz,
                     float* x_real, void* stream);
int dg_inv_to_xyz(const float* in, const float* angle, int B, int H, int W, int from_tanh, float min_depth,
                  float max_depth, float drop_const, float tol, float* depth01, float* points, void* stream);
/* utils.postprocess's other branches (utils/__init__.py:169-172): mode 0 = tanh_to_sigmoid(x).clamp(0,1)
 * (depth_orig), mode 1 = sigmoid(x) (confidence) */
int dg_unit_map(const float* x, long n, int mode, float* y, void* stream);
/* xyz_to_normal(points, mode="closest")  utils/__init__.py:215-219 -> estimate_surface_normal utils/geometry.py:38-127
 * (d = 2): points [B,3,H,W] -> normal image [B,3,H,W] in [0,1] */
int dg_normals(const float* points, int B, int H, int W, int d, float* out, void* stream);

/* ---- validation metrics (the reference's CUDA extensions and their torch drivers) -------
 * dg_fps: furthest point sampling  utils/sampling/fps/furthest_point_sampling.cu:97-207 (+ gather_points :38-60 when
 * `out` is given).  xyz [B,n,3] fp32, m <= n samples per cloud, temp [B,n] fp32 workspace, idx [B,m] int32, out
 * [B,m,3] (nullable) = the sampled points (downsample_point_clouds, furthest_point_sampling.py:84-93).  Starts at
 * index 0, skips points with |p|^2 <= 1e-3, ties resolved in the reference launcher's thread order.
 * dg_chamfer_dir: all-pairs directed Chamfer means L[i][j] = mean_{p in A_i} min_{q in B_j} |p - q|^2 for A [Na,n,3],
 * B [Nb,m,3] -> L [Na,Nb]; compute_cd(A_i, B_j) of utils/metrics/cov_mmd_1nna.py:20-22 = L_AB[i][j] + L_BA[j][i]
 * (chamfer_distance.cu / nnsearch chamfer_distance.cpp:41-66).  One launch per matrix instead of the reference's
 * Python loop over i (:34-50).
 * dg_grid_vote: JSD occupancy histogram  utils/metrics/jsd.py:24-79: counters[argmin_k |p - grid_k|^2] += 1 for
 * every point (first index on ties); pts [P,3], grid [Ng,3] (unit_cube_grid_point_cloud :11-21), counters [Ng]
 * accumulate.  dg_jsd: _jensen_shannon_divergence :96-107 of two counter vectors -> out[0]. */
int dg_fps(const float* xyz, int B, int n, int m, float* temp, int* idx, float* out, void* stream);
int dg_chamfer_dir(const float* A, int Na, int n, const float* Bc, int Nb, int m, float* L, void* stream);
/* dg_emd: approximate earth mover's distance  utils/metrics/distance/emd/earth_mover_distance.cu (approxmatch :28-190
 * + matchcost :218-262 fused; the [m,n] match matrix is never stored).  paired = 1: out[i] = emd(A_i, B_i) (Na == Nb,
 * the extension's semantics); paired = 0: out [Na,Nb] for all pairs.  The value is the raw cost (compute_emd of
 * utils/metrics/cov_mmd_1nna.py:12-17 divides by the point count). */
int dg_emd(const float* A, int Na, int n, const float* Bc, int Nb, int m, int paired, float* out, void* stream);
int dg_grid_vote(const float* pts, long P, const float* grid, int Ng, float* counters, void* stream);
int dg_jsd(const float* P, const float* Q, int n, float* out, void* stream);
/* SWD descriptors  utils/metrics/swd.py:16-62.  dg_pyr_down: pyramid_down (:24-30) on `planes` = B*C images
 * [H,W] -> [H/2,W/2] (H, W even).  dg_pyr_up_sub: fine -= pyramid_up(coarse) (:33-50), in place.
 * dg_extract_patches: extract_patches (:53-62) for given positions: out [B,NP,C,ph,pw], inds [NP] int64 =
 * y * (W - pw + 1) + x of each patch's top-left corner (the reference draws them with randperm). */
int dg_pyr_down(const float* in, long planes, int H, int W, float* out, void* stream);
int dg_pyr_up_sub(float* fine, const float* coarse, long planes, int H, int W, void* stream);
int dg_extract_patches(const float* img, int B, int C, int H, int W, int ph, int pw, const long* inds, int NP,
                       float* out, void* stream);

/* ---- small reductions / helpers --------------------------------------------------------------------------- */
int dg_sample_sum(const float* x, int B, long n, int sq, float* out, void* stream); /* out[b] = sum x or x^2 */
int dg_scale(const float* x, float a, long n, float* y, void* stream);
int dg_sample_sum_acc(const float* x, int B, long n, int sq, float* out, void* stream); /* out[b] = sum x or x^2 */
int dg_scale(const float* x, float a, long n, float* y, void* stream);
/* p[0..n) = 0 as a kernel launch (optim.zero_grad, trainers/dcgan_amp.py:177,246, and the step's accumulators).  The
 * library never uses hipMemsetAsync: its small fills replayed wrong from a hipGraph after a host-side synchronize. */
int dg_zero(float* p, long n, void* stream);
/* k <= 4 fp32 buffers (16-byte aligned, counts multiples of 4) zero-filled by ONE launch: the step's accumulator arena and
 * the two networks' gradient buffers (optim.zero_grad, trainers/dcgan_amp.py:177,246) as one graph node */
int dg_zero_multi(float* const* ptrs, const long* counts, int k, void* stream);

/* ---- optim.Adam.step + ema_inplace  trainers/dcgan_amp.py:116-125,238,312,316 and :30-35 ------------------ */
int dg_adam_ema_step(float* p, const float* grad, float* m, float* v, float* ema, void* shadow, int shadow_dtype,
                     long n, float gscale, float lr, float beta1, float beta2, float eps, int step, float ema_decay,
                     void* stream);
int dg_cast(const float* src, void* dst, int dtype, long n, void* stream);
/* the way back: dst[i] = (float)src[i] for DG_BF16 / DG_BF16X2 sources (hi + lo), a copy for DG_F32 */
int dg_uncast(const void* src, int dtype, float* dst, long n, void* stream);
/* up to 16 fp32 buffers (n[i] elements, multiples of 64) -> DG_BF16X2 copies in ONE launch: the split-bf16 twins of the fat
 * layers' weight shadows, rebuilt behind every optimizer step in the fp32x3 mode */
int dg_cast_x2_multi(const float* const* src, void* const* dst, const long* n, int count, void* stream);
int dg_transpose_shadow(const float* master, void* dst, int dtype, int Ci, int Co, void* stream);
/* every conv segment of a network in one launch: desc_dev[5 i + (0..4)] = (source element offset in master, destination
 * pointer, Ci, Co, first tile index), tiles = 16 taps x ceil(Ci/32) x ceil(Co/32) per segment (device memory) */
int dg_transpose_shadow_multi(const float* master, const long long* desc_dev, int nseg, int total_tiles, int dtype,
                              void* stream);
/* the same launch also rebuilds up to 4 DgConv.up_frag buffers from the fp32 master (bf16 only): weight element
 * (tap, n, k) of fragment set i is master[off + tap * m_st + n * m_sn + k * m_sk], rounded to bf16 like the shadow; K = 64.
 * One launch per optimizer step keeps every low-precision copy of a network current (Head / Down1 weights change with
 * optim.step, trainers/dcgan_amp.py:238,314) */
typedef struct DgUpFrag {
  long long off;             /* element offset of the layer's weights in `master` */
  void* frag;                /* DG_UP_FRAG_BYTES */
  long long m_st, m_sn, m_sk;
  int N, Hc, adj;            /* output channels (<= 4), coarse rows, DgConv.adj of the launches that use it */
} DgUpFrag;
int dg_transpose_shadow_multi_frags(const float* master, const long long* desc_dev, int nseg, int total_tiles, int dtype,
                                    const DgUpFrag* frags, int nfrag, void* stream);
/* the same launch with one more block doing what dg_counter_add_multi / dg_counter_add_multi_snap do (snap_idx < 0: no
 * snapshot): the last launch of a training step - the shadow refresh behind the generator's optimizer - also advances the
 * step's counters and files its scalars */
int dg_transpose_shadow_multi_tail(const float* master, const long long* desc_dev, int nseg, int total_tiles, int dtype,
                                   const DgUpFrag* frags, int nfrag, unsigned long long* const* counters,
                                   const unsigned long long* deltas, int k, int snap_idx, const float* src, int n,
                                   float* dst_ring, int ring, void* stream);

/* ---- Philox4x32-10 draws (the reference uses torch's device RNG: trainers/dcgan_amp.py:151-152, models/dusty.py:33-34,
 *      utils/diff_augment.py:27-28,59-60,86-87) --------------------------------------------------------------- */
int dg_philox_bits(uint64_t seed, uint64_t stream_id, uint64_t offset, long n4, uint32_t* out, void* stream);
/* kind 0 uniform[0,1), 1 normal, 2 uniform(lo,hi), 3 int32 in [ilo,ihi) */
int dg_philox_fill(uint64_t seed, uint64_t stream_id, uint64_t offset, int kind, float lo, float hi, int ilo, int ihi,
                   long n, void* out, void* stream);

/* one DiffAugment parameter set per sample: uf [3][B] = u_b,u_s,u_c in (-1,1); qi [4][B] = t_h,t_w,o_x,o_y
 * (ranges of utils/diff_augment.py:58-60,85-87); consumes 2*B Philox counters starting at `offset` */
int dg_aug_draw(uint64_t seed, uint64_t stream_id, uint64_t offset, int B, int H, int W, float* uf, int* qi,
                void* stream);

/* ---- device-resident counters (hipGraph-friendly variants): Philox offset / Adam step count read from device memory;
 *      dg_counter_add advances a counter after its consumers.  A step captured once replays with fresh draws. */
int dg_counter_add(unsigned long long* counter, unsigned long long delta, void* stream);
/* k <= 8 distinct counters advanced by one launch (DG_EINVAL on duplicates) */
int dg_counter_add_multi(unsigned long long* const* counters, const unsigned long long* deltas, int k, void* stream);
/* the same, and src[0..n) (n <= 64) is filed in slot (OLD value of counters[snap_idx]) % ring of dst_ring (n floats per slot;
 * device memory or mapped pinned host memory): the logged scalars of Trainer.step (trainers/dcgan_amp.py:319-323, five
 * blocking all_reduce + .item() there) leave the device with the step's last launch and are read by the host behind an
 * event, without a copy node or a blocking read on the launch stream */
int dg_counter_add_multi_snap(unsigned long long* const* counters, const unsigned long long* deltas, int k, int snap_idx,
                              const float* src, int n, float* dst_ring, int ring, void* stream);
int dg_philox_fill_dev(uint64_t seed, uint64_t stream_id, const unsigned long long* offset_dev, int kind, float lo,
                       float hi, int ilo, int ihi, long n, void* out, void* stream);
/* GumbelSigmoid.logistic_noise (models/dusty.py:30-36) in one launch: the same numbers as two dg_philox_fill_dev uniform
 * fills of n elements (U1, then U2) followed by dg_logistic_noise; the caller advances the counter by 2 ((n + 3) / 4) */
int dg_philox_logistic_dev(uint64_t seed, uint64_t stream_id, const unsigned long long* offset_dev, float eps, long n,
                           float* out, void* stream);
/* One launch in front of a training step: the zero-fill of k <= 4 fp32 buffers (dg_zero_multi: the accumulator arena and
 * the gradient buffers, optim.zero_grad at trainers/dcgan_amp.py:177,246) plus up to 6 draws - the numbers
 * dg_philox_fill_dev / dg_philox_logistic_dev / dg_aug_draw_dev produce for the same (seed, stream_id, *offset_dev + base):
 * sample_latents (trainers/dcgan_amp.py:151-152), GumbelSigmoid.logistic_noise (models/dusty.py:30-36) and DiffAugment's
 * parameters (utils/diff_augment.py:27-28,59-60,86-87).  The caller advances the counters as for the single draws. */
typedef struct DgDraw {
  int kind;                  /* 0 dg_philox_fill_dev, 1 dg_philox_logistic_dev, 2 dg_aug_draw_dev */
  int fill_kind;             /* kind 0: 0 U[0,1), 1 N(0,1), 2 U[lo,hi), 3 integers in [ilo, ihi) */
  uint64_t seed, stream_id;
  const unsigned long long* offset_dev;
  unsigned long long base;   /* added to *offset_dev: what earlier draws of this launch take from the same generator */
  float lo, hi, eps;
  int ilo, ihi;
  long n;                    /* kind 0 / 1: elements */
  void* out;                 /* kind 0 / 1 */
  void* out_bf16;            /* kind 0, optional: a bfloat16 copy of a float fill (the generator's latent operand) */
  int B, H, W;               /* kind 2 */
  float* uf;                 /* kind 2: [3][B] */
  int* qi;                   /* kind 2: [4][B] */
} DgDraw;
int dg_step_prologue(float* const* zero_ptrs, const long* zero_counts, int k, const DgDraw* draws, int ndraw, void* stream);
/* ... and fetch_reals of the step's batch as more workgroups of the SAME launch (round 6; trainers/dcgan_amp.py:154-160 with
 * Coordinate.invert_depth, utils/lidar.py:31-36, and sigmoid_to_tanh, utils/__init__.py:70-73; same arithmetic as
 * dg_fetch_reals_pool_sum): out[B,1,H,W] = the inverse-depth image in [-1, 1] of batch (*pool_ctr % npool) of the device-resident
 * pool (pool_ctr NULL: pol / mask ARE the batch), and parts[B][DG_XSUM_PARTS] = the partial sums of out per sample - block j owns
 * pixels [j HW / DG_XSUM_PARTS, (j + 1) HW / DG_XSUM_PARTS) of the batch and STORES its sum (no accumulator this launch would
 * have to zero first, no atomics: bit-reproducible); DgAugSet.xsum_parts tells the reader.  DG_EUNSUPPORTED (nothing launched)
 * unless HW % (1024 DG_XSUM_PARTS) == 0 and the images are 16-byte aligned. */
typedef struct DgFetch {
  const float *pol, *mask;              /* [npool][B][HW] (or [B][HW]) polar depth in [0, 1] and validity {0, 1} */
  const unsigned long long* pool_ctr;   /* device-resident loader position, or NULL */
  int npool;
  float min_depth, max_depth, drop_const;
  int B;
  long HW;
  float* out;                           /* [B][HW] */
  float* parts;                         /* [B][DG_XSUM_PARTS] */
} DgFetch;
int dg_step_prologue_fetch(float* const* zero_ptrs, const long* zero_counts, int k, const DgDraw* draws, int ndraw,
                           const DgFetch* fetch, void* stream);
int dg_aug_draw_dev(uint64_t seed, uint64_t stream_id, const unsigned long long* offset_dev, int B, int H, int W,
                    float* uf, int* qi, void* stream);
/* Adam with the (0-based, already-completed) step count in device memory: this call is step *step_dev + 1 */
int dg_adam_ema_step_dev(float* p, const float* grad, float* m, float* v, float* ema, void* shadow, int shadow_dtype,
                         long n, float gscale, float lr, float beta1, float beta2, float eps,
                         const unsigned long long* step_dev, float ema_decay, void* stream);

/* The optimizer of a network as ONE launch that forms the gradients it consumes (round 6; trainers/dcgan_amp.py:238, :312-316
 * behind loss.backward()'s last reductions): per piece of the flat buffers, the sum of the piece's split-K partial rows
 * (`part`, [splits][numel], fixed order - what dg_wgrad_reduce would add to the gradient; `accumulate`: + the gradient buffer's
 * content), optionally + ws_scale * sum_b ws_coef[b] * ws_src[b][i] (dg_batch_wsum: the final conv's weight gradient from the
 * R1 tangent; ws_coef NULL = 1), then Adam with beta1 = 0 + the EMA (`ema` NULL: none) exactly as dg_adam_ema_step_dev, and
 * every store: master, exp_avg_sq, EMA, the summed gradient (kept), the compute-type shadow, and for kind 1 - a [16][ci][co]
 * conv segment, co == 64 or co % 128 == 0, ci % 16 == 0 - the [16][co][ci] shadow `shadow_t` (dg_transpose_shadow_multi's output).  Replaces
 * dg_batch_wsum + dg_wgrad_reduce + dg_adam_ema_step_dev + the fat layers' share of dg_transpose_shadow_multi; the results are
 * bit-identical to theirs (same order of additions) except where ws_src folds dg_batch_wsum in.  Pieces must not overlap;
 * anything of the buffers no piece covers is left untouched.  `first_block` is filled by the library. */
#define DG_OPT_MAX_SEG 20
typedef struct DgOptSeg {
  long long off, numel;          /* piece [off, off + numel) of p / grad / v / ema / shadow, multiples of 4 */
  const float* part;             /* partial rows or NULL */
  int splits, accumulate;
  int kind, ci, co;              /* kind 0: flat; 1: conv tile walk (needs shadow_t) */
  void* shadow_t;
  const void* ws_src; const float* ws_coef; long long ws_stride; int ws_n, ws_bf16; float ws_scale;
  int first_block;
} DgOptSeg;
int dg_adam_fused(float* p, float* grad, float* v, float* ema, void* shadow, int shadow_dtype, const DgOptSeg* segs, int nseg,
                  float gscale, float lr, float beta2, float eps, const unsigned long long* step_dev, float ema_decay,
                  void* stream);
/* Adam (beta1 = 0) + EMA of Proj.weight [Np][K] with its weight gradient wscale * dp0^T z (trainers/dcgan_amp.py:309,312:
 * loss_G.backward() + optim_G.step() for that one tensor) formed inside the kernel from the bf16 operands dp0 [nb][Np]
 * and zT [nb][K]: the 268 MB gradient is never written.  Two kernels behind it: an LDS-resident VALU kernel for the
 * per-GPU batch (nb even, <= 64, operands fit 64 KB of LDS) and, for larger nb (the all-gathered global batch of a
 * data-parallel run), the MFMA gradient GEMM with the optimizer as its epilogue (Np % 128 == 0, K % 128 == 0).
 * Round 6: op_dtype DG_F32 (fp32 operands on the fp32 matrix instructions: the exact-fp32 mode) and DG_F32 | DG_FORCE_FP32X3
 * (fp32 operands split into bf16 pairs in registers: the fp32x3 mode) take the matrix-core path too - those modes ran the
 * gradient GEMM, its reduce and the plain optimizer over the 268 MB gradient.  shadow may be NULL (fp32 modes: the master IS
 * what the forward pass reads).  DG_EUNSUPPORTED for anything else (other element types, other shapes) - the caller then uses
 * dg_wgrad + dg_adam_ema_step_dev. */
int dg_adam_proj_fused(float* p, float* v, float* ema, void* shadow, int shadow_dtype, const void* dp0, const void* zT,
                       int op_dtype, int nb, long Np, int K, float wscale, float gscale, float lr, float beta2,
                       float eps, const unsigned long long* step_dev, float ema_decay, void* stream);

const char* dg_version(void);

#ifdef __cplusplus
}
#endif
#endif
