// extern "C" dispatchers of the conv-like passes (include/dusty_gan_hip.h).
#include "common.h"


int dg_conv_direct_launch(const ConvP* p, hipStream_t stream);
int dg_lrelu_bits_launch(const ConvP* p, hipStream_t stream);
int dg_conv_mfma_launch(const ConvP* p, hipStream_t stream, int wg_cap, DgConvPlan* plan, int x3);
int dg_conv_mfma_big_launch(const ConvP* p, hipStream_t stream, int family, int wg_cap, DgConvPlan* plan);
int dg_wgrad_mfma_dma_supported(const WgradP* p);
int dg_wgrad_mfma_dma_launch(const WgradP* p, int accumulate, int pairs, hipStream_t stream, DgWgradPlan* plan);
int dg_wgrad_mfma_dma_group_launch(const WgradP* items, int n, int pairs, int rounds, hipStream_t stream, DgWgradPlan* plans);
int dg_conv_thin_launch(const ConvP* p, hipStream_t stream);
int dg_conv_thin_supported(const ConvP* p);
int dg_proj_stream_supported(const ConvP* p);
int dg_proj_stream_launch(const ConvP* p, hipStream_t stream, DgConvPlan* plan);
int dg_conv_thin_mfma_variant(const ConvP* p);
int dg_conv_up_mfma_sum_parts(const ConvP* p);
int dg_conv_s2_mfma_blocks(const ConvP* p);
int dg_wgrad_thin_mfma_variant(const WgradP* p);
int dg_wgrad_thin_ws_splits(const WgradP* p);
int dg_wgrad_mfma_ws_splits(const WgradP* p, int accumulate);
int dg_wgrad_direct_ws_splits(const WgradP* p);
int dg_wgrad_direct_launch(const WgradP* p, hipStream_t stream);
int dg_wgrad_mfma_launch(const WgradP* p, int accumulate, hipStream_t stream, int x3);
int dg_wgrad_thin_launch(const WgradP* p, hipStream_t stream);
int dg_wgrad_thin_supported(const WgradP* p);

// `force` carries one flag bit beside the kernel-family code: DG_FORCE_FP32X3 (include/dusty_gan_hip.h) - fp32 operands through
// split-bf16 matrix instructions (mfma_common.h).  Per CALL, so two engines of different precision in one process do not
// share a setting (rounds 3-4 had a process-wide dg_set_fp32_split).

extern "C" {

const char* dg_version(void) { return "dusty_gan_hip 0.1 (gfx950)"; }

// force: 0 auto (MFMA implicit GEMM -> thin LDS/VALU kernel -> direct), 1 direct, 2 MFMA or error, 3 thin or error,
//        4 lock-step persistent large-tile MFMA kernel or error, 5 ping-pong persistent kernel or error (9: without its
//        both-parities tile for 64-channel MODE_UP layers), 10 the weight-streaming Proj forward or error;
//        | DG_FORCE_FP32X3: DG_F32 operands through split-bf16 matrix instructions (the one-tile-per-workgroup MFMA kernels).
//        plan != NULL: describe the launch instead of making it.
static int conv_dispatch0(const DgConv* p, int force_flags, int wg_cap, hipStream_t s, DgConvPlan* plan) {
  const int force = force_flags & ~DG_FORCE_FP32X3, x3 = (force_flags & DG_FORCE_FP32X3) ? 1 : 0;
  if (!p || !p->in || !p->out || !p->w) return DG_EINVAL;
  if (p->B <= 0 || p->K <= 0 || p->N <= 0) return DG_EINVAL;
  if (p->mode != MODE_GEMM && (p->Hc < 2 || p->Wc < 2)) return DG_EINVAL;
  if (p->epi == EPI_MASK && !p->aux) return DG_EINVAL;
  if (p->bias && p->bias_mod <= 0) return DG_EINVAL;
  if (p->dbias && p->bias_mod <= 0) return DG_EINVAL;
  const bool mfma_ok = !p->nscale && dg_conv_mfma_supported(p);
  const bool thin_ok = dg_conv_thin_supported(p);
  if (plan) { plan->family = 0; plan->bm = plan->bn = 0; plan->tiles = plan->workgroups = plan->tiles_per_wg = 0; plan->thin_mfma = 0; plan->mask_bits = 0; plan->dbias_rows = 0; plan->sum_parts = 0; }
  // Proj forward (bf16, K = 512, B <= 32): the weight-streaming kernel (proj_stream.hip); force 10 asks for it, 2 for the
  // general MFMA kernel it replaces
  if ((force == 0 || force == 10) && dg_proj_stream_supported(p)) return dg_proj_stream_launch(p, s, plan);
  if (force == 10) return DG_EUNSUPPORTED;
  if (force == 2) return mfma_ok ? dg_conv_mfma_launch(p, s, wg_cap, plan, x3) : DG_EUNSUPPORTED;
  if (force == 4 || force == 5 || force == 9) return mfma_ok ? dg_conv_mfma_big_launch(p, s, force, wg_cap, plan) : DG_EUNSUPPORTED;
  if (force == 0 && mfma_ok) return dg_conv_mfma_launch(p, s, wg_cap, plan, x3);
  if (force == 3 && !thin_ok) return DG_EUNSUPPORTED;
  if ((force == 3 || force == 0) && thin_ok) {
    if (plan) {
      plan->family = 3; plan->thin_mfma = dg_conv_thin_mfma_variant(p); plan->mask_bits = plan->thin_mfma == 1 ? 3 : 0;
      plan->dbias_rows = plan->thin_mfma == 1 ? dg_conv_s2_mfma_blocks(p) : 0;
      plan->sum_parts = plan->thin_mfma == 2 ? dg_conv_up_mfma_sum_parts(p) : 0;
      return DG_OK;
    }
    return dg_conv_thin_launch(p, s);
  }
  if (plan) { plan->family = 1; return DG_OK; }
  return dg_conv_direct_launch(p, s);
}

// The saved-mask fields around the dispatch: mask_out is honoured behind EVERY kernel (natively by the ping-pong conv and the
// thin matrix-core MODE_S2 kernel - DgConvPlan.mask_bits & 1 - otherwise by one packing launch over the output), mask_in
// only by kernels that take bits (the others read aux, which stays mandatory).
static int conv_dispatch(const DgConv* p, int force, int wg_cap, hipStream_t s, DgConvPlan* plan) {
  if (p && (p->mask_out || p->mask_in)) {
    if (p->mask_out && p->epi != EPI_LRELU) return DG_EINVAL;
    if (p->mask_in && p->epi != EPI_MASK) return DG_EINVAL;
    if (p->out_sn != 1 || p->N % 8 != 0 || p->out_sp % 8 != 0 || p->out_sb % 8 != 0) return DG_EUNSUPPORTED;
  }
  if (plan || !p || !p->mask_out) return conv_dispatch0(p, force, wg_cap, s, plan);
  DgConvPlan pl;
  int rc = conv_dispatch0(p, force, wg_cap, nullptr, &pl);
  if (rc) return rc;
  rc = conv_dispatch0(p, force, wg_cap, s, nullptr);
  if (rc == DG_OK && !(pl.mask_bits & 1)) rc = dg_lrelu_bits_launch(p, s);
  return rc;
}

int dg_conv(const DgConv* p, int force, void* stream) { return conv_dispatch(p, force, 0, (hipStream_t)stream, nullptr); }

int dg_conv_ex(const DgConv* p, int force, int wg_cap, void* stream) {
  return conv_dispatch(p, force, wg_cap, (hipStream_t)stream, nullptr);
}

int dg_conv_plan(const DgConv* p, int force, int wg_cap, DgConvPlan* plan) {
  if (!plan) return DG_EINVAL;
  return conv_dispatch(p, force, wg_cap, nullptr, plan);
}

int dg_conv_kernel_choice(const DgConv* p) {  // 2 = MFMA, 3 = thin, 1 = direct (what force == 0 would pick)
  if (dg_proj_stream_supported(p)) return 2;
  if (!p->nscale && dg_conv_mfma_supported(p)) return 2;
  if (dg_conv_thin_supported(p)) return 3;
  return 1;
}

// force: 0 auto, 1 direct, 2 MFMA (the LDS-DMA kernel where the shape allows), 3 thin, 6 the register-staged MFMA kernel,
//        7 / 8 the LDS-DMA kernel with / without W-tap pairs
static int wgrad_dispatch(const DgWgrad* p, int accumulate, int force_flags, hipStream_t s, DgWgradPlan* plan) {
  const int force = force_flags & ~DG_FORCE_FP32X3, x3 = (force_flags & DG_FORCE_FP32X3) ? 1 : 0;
  if (!p || !p->a || !p->g || !p->dw) return DG_EINVAL;
  if (p->B <= 0 || p->Ci <= 0 || p->Co <= 0 || p->Hc <= 0 || p->Wc <= 0) return DG_EINVAL;
  const bool mfma_ok = dg_wgrad_mfma_supported(p);
  const bool thin_ok = dg_wgrad_thin_supported(p);
  if (plan) { plan->variant = dg_wgrad_kernel_variant(p, force); plan->splits = 1; plan->ws_floats = 0; plan->tap_pairs = 0; }
  // bf16 Down / Up layers: LDS-DMA ring version (wgrad_mfma_dma.hip); force == 6 asks for the register-staged kernel
  if ((force == 0 || force == 2 || force == 7 || force == 8) && mfma_ok && dg_wgrad_mfma_dma_supported(p))
    return dg_wgrad_mfma_dma_launch(p, accumulate, force == 7 ? 1 : (force == 8 ? 2 : 0), s, plan);
  if (force == 7 || force == 8) return DG_EUNSUPPORTED;
  // the thin matrix-core kernels (Down1, Head) also have the workspace form: one partial tile per block
  const bool thin_runs = thin_ok && (force == 3 || (force == 0 && !mfma_ok));
  const int thin_splits = thin_runs ? dg_wgrad_thin_ws_splits(p) : 0;
  // ... and so have the register-staged MFMA kernel (the fp32 modes' fat layers) and the direct kernel (narrow nets) wherever
  // they split K over workgroups (round 6: their fp32 atomics summed in arrival order)
  const bool mfma_runs = mfma_ok && (force == 0 || force == 2 || force == 6);
  const bool direct_runs = !mfma_runs && !thin_runs && (force == 0 || force == 1) && !p->g_mod;
  const int other_splits = mfma_runs ? dg_wgrad_mfma_ws_splits(p, accumulate) : (direct_runs ? dg_wgrad_direct_ws_splits(p) : 0);
  if (plan) {
    const long numel = (long)(p->wmode == 2 ? 1 : 16) * p->Ci * p->Co;
    if (thin_splits) { plan->splits = thin_splits; plan->ws_floats = (long)thin_splits * 16 * p->Ci * p->Co; }
    else if (other_splits) { plan->splits = other_splits; plan->ws_floats = other_splits * numel; }
    return plan->variant ? DG_OK : DG_EUNSUPPORTED;
  }
  if (p->ws && !thin_splits && !other_splits) return DG_EUNSUPPORTED;
  // ... and the gradient-sample map exists there and in the thin matrix-core kernel of Down1 (checked by its launcher)
  const bool thin_map = p->g_mod && thin_ok && !mfma_ok && (force == 0 || force == 3);
  if (p->g_mod && !thin_map) return DG_EUNSUPPORTED;
  if (force == 2 || force == 6) return mfma_ok ? dg_wgrad_mfma_launch(p, accumulate, s, x3) : DG_EUNSUPPORTED;
  if (force == 0 && mfma_ok) return dg_wgrad_mfma_launch(p, accumulate, s, x3);
  if (!accumulate && !p->ws) {                       // (workspace form: the reduce launch overwrites dw)
    const long n = (long)(p->wmode == 2 ? 1 : 16) * p->Ci * p->Co;
    { const int zrc = dg_zero_f32(p->dw, n, s); if (zrc) return zrc; }
  }
  if (force == 3) return thin_ok ? dg_wgrad_thin_launch(p, s) : DG_EUNSUPPORTED;
  if (force == 0 && thin_ok) return dg_wgrad_thin_launch(p, s);
  return dg_wgrad_direct_launch(p, s);
}

int dg_wgrad(const DgWgrad* p, int accumulate, int force, void* stream) {
  return wgrad_dispatch(p, accumulate, force, (hipStream_t)stream, nullptr);
}

static int wgrad_group_dispatch(const DgWgrad* items, int n, int force_flags, int rounds, hipStream_t stream, DgWgradPlan* plans) {
  const int force = force_flags & ~DG_FORCE_FP32X3;
  if (!items || n < 1) return DG_EINVAL;
  if (force != 0 && force != 2 && force != 7 && force != 8) return DG_EUNSUPPORTED;
  for (int i = 0; i < n; ++i) {
    const DgWgrad* p = &items[i];
    if (!p->a || !p->g || !p->dw || p->B <= 0 || p->Ci <= 0 || p->Co <= 0 || p->Hc <= 0 || p->Wc <= 0) return DG_EINVAL;
    if (!dg_wgrad_mfma_supported(p)) return DG_EUNSUPPORTED;
  }
  return dg_wgrad_mfma_dma_group_launch(items, n, force == 7 ? 1 : (force == 8 ? 2 : 0), rounds, stream, plans);
}

int dg_wgrad_group(const DgWgrad* items, int n, int force, int rounds, void* stream) {
  return wgrad_group_dispatch(items, n, force, rounds, (hipStream_t)stream, nullptr);
}

int dg_wgrad_group_plan(const DgWgrad* items, int n, int force, int rounds, DgWgradPlan* plans) {
  if (!plans) return DG_EINVAL;
  return wgrad_group_dispatch(items, n, force, rounds, nullptr, plans);
}

int dg_wgrad_plan(const DgWgrad* p, int accumulate, int force, DgWgradPlan* plan) {
  if (!plan) return DG_EINVAL;
  return wgrad_dispatch(p, accumulate, force, nullptr, plan);
}

// 5 = MFMA on the LDS-DMA ring (wgrad_mfma_dma.hip), 2 = register-staged MFMA, 7 = thin on the matrix cores, 3 = thin
// (VALU), 1 = direct: what `force` launches
int dg_wgrad_kernel_variant(const DgWgrad* p, int force_flags) {
  const int force = force_flags & ~DG_FORCE_FP32X3;
  const bool mfma_ok = dg_wgrad_mfma_supported(p);
  if ((force == 0 || force == 2 || force == 7 || force == 8) && mfma_ok && dg_wgrad_mfma_dma_supported(p)) return 5;
  if (force == 7 || force == 8) return 0;
  if (force == 2 || force == 6) return mfma_ok ? 2 : 0;  // (6: the register-staged kernel, as dg_wgrad dispatches it)
  if (force == 0 && mfma_ok) return 2;
  const bool thin_ok = dg_wgrad_thin_supported(p);
  if ((force == 3 || force == 0) && thin_ok) return dg_wgrad_thin_mfma_variant(p) ? 7 : 3;
  if (force == 3) return 0;
  return 1;
}

// 1 when the kernel dg_wgrad(p, force) launches honours DgWgrad.g_mod (the LDS-DMA kernel, Down1's thin matrix-core kernel)
int dg_wgrad_has_sample_map(const DgWgrad* p, int force) {
  const int v = dg_wgrad_kernel_variant(p, force);
  if (v == 5) return 1;
  return v == 7 && p->wmode == 0 && dg_wgrad_thin_mfma_variant(p) == 1;
}

int dg_zero(float* p, long n, void* stream) { return p ? dg_zero_f32(p, n, (hipStream_t)stream) : DG_EINVAL; }

int dg_wgrad_kernel_choice(const DgWgrad* p) {
  if (dg_wgrad_mfma_supported(p)) return 2;
  if (dg_wgrad_thin_supported(p)) return 3;
  return 1;
}

}  // extern "C"
