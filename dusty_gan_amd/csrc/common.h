// Shared definitions for the dusty-gan HIP kernels (gfx950 / MI355X only).
//
// Layout conventions (DESIGN.md "Data layout in HBM"):
//   * activations: pixel-major / channel-minor ("NHWC"), element type T = float or __bf16
//   * conv weights (engine master, fp32):   M[ky][kx][ci][co]         ("cico")
//     shadows in T:                         S_cico[tap][ci][co], S_coci[tap][co][ci]
//   * every conv-like op is expressed on a COARSE grid (Hc x Wc) and a FINE grid (2Hc x 2Wc):
//       MODE_S2: out on the coarse grid, in on the fine grid  (Down forward, Up backward-data)
//       MODE_UP: out on the fine grid,  in on the coarse grid (Up forward, Down backward-data)
//     with boundary flavour adj=0 (the reference's Pad: reflect rows, circular/reflect columns;
//     models/ops/common.py:9-20) or adj=1 (its adjoint: zero fill + reflect-adjoint extra taps).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;

#include "../../include/dusty_gan_hip.h"

#define MODE_S2 0
#define MODE_UP 1
#define MODE_GEMM 2  // plain row-major GEMM rows (Proj): no taps

#define EPI_LINEAR 0
#define EPI_LRELU 1
#define EPI_MASK 2

#define LRELU_SLOPE 0.2f
#define SQRT2 1.4142135623730951f

#define HIP_CHECK_RET(x)                  \
  do {                                    \
    hipError_t _e = (x);                  \
    if (_e != hipSuccess) return DG_EHIP; \
  } while (0)

// ---------------------------------------------------------------------------------------------
// 1-D tap enumeration shared by every conv-like kernel (host+device; unit-tested on the host).
//
// For output index P on an axis whose coarse size is Nc (fine size 2Nc), tap i in [0,6):
// returns false if the tap does not exist, else the source index `src` on the INPUT axis and the
// 4-tap kernel index k.  Derivation: SURVEY.md §2.1 (probe-verified against F.conv2d /
// F.conv_transpose2d), boundary rules of models/ops/common.py:9-20.
//   circ=1: circular axis (W with ring=True): the adjoint of a circular conv is circular, adj is ignored.
//   circ=0: reflect axis (H always; W when ring=False).
__host__ __device__ inline bool dg_tap1d(int mode, int adj, int circ, int P, int Nc, int i, int& src, int& k) {
  const int Nf = 2 * Nc;
  if (mode == MODE_S2) {
    // out[P] = sum_k w[k] * in_f[2P-1+k]
    if (i < 4) {
      int r = 2 * P - 1 + i;
      k = i;
      if (circ) {
        if (r < 0) r += Nf;
        if (r >= Nf) r -= Nf;
        src = r;
        return true;
      }
      if (!adj) {  // reflect: -1 -> 1, Nf -> Nf-2
        if (r < 0) r = -r;
        if (r >= Nf) r = 2 * Nf - 2 - r;
        src = r;
        return true;
      }
      src = r;
      return r >= 0 && r < Nf;
    }
    if (circ || !adj) return false;
    // adjoint of Up's reflect padding: out_up[0] read x[1] through w[3]; out_up[Nf-1] read x[Nc-2] through w[0]
    if (i == 4) { src = 0; k = 3; return P == 1; }
    if (i == 5) { src = Nf - 1; k = 0; return P == Nc - 2; }
    return false;
  }
  // MODE_UP: P on the fine axis. even: w[1]*x[m] + w[3]*x[m-1]; odd: w[0]*x[m+1] + w[2]*x[m]
  const int m = P >> 1, par = P & 1;
  if (i < 2) {
    int r;
    if (i == 0) { r = par ? m + 1 : m; k = par ? 0 : 1; }
    else        { r = par ? m : m - 1; k = par ? 2 : 3; }
    if (circ) {
      if (r < 0) r += Nc;
      if (r >= Nc) r -= Nc;
      src = r;
      return true;
    }
    if (!adj) {
      if (r < 0) r = -r;
      if (r >= Nc) r = 2 * Nc - 2 - r;
      src = r;
      return true;
    }
    src = r;
    return r >= 0 && r < Nc;
  }
  if (circ || !adj) return false;
  // adjoint of Down's reflect padding: out_down[0] read x[1] through w[0]; out_down[Nc-1] read x[Nf-2] through w[3]
  if (i == 2) { src = 0; k = 0; return P == 1; }
  if (i == 3) { src = Nc - 1; k = 3; return P == Nf - 2; }
  return false;
}

// Weight-gradient index maps.  For coarse index m and kernel index k (0..3) along one axis:
//   wmode 0 (Down): input on the fine axis at  map(2m+k-1), gradient on the coarse axis at m
//   wmode 1 (Up):   input on the coarse axis at map(m+d_k), gradient on the fine axis at 2m+par_k
__host__ __device__ inline void dg_wgrad1d(int wmode, int circ, int m, int Nc, int k, int& src_in, int& src_g) {
  if (wmode == 0) {
    const int Nf = 2 * Nc;
    int r = 2 * m - 1 + k;
    if (circ) { if (r < 0) r += Nf; if (r >= Nf) r -= Nf; }
    else      { if (r < 0) r = -r;  if (r >= Nf) r = 2 * Nf - 2 - r; }
    src_in = r;
    src_g = m;
  } else {
    const int par = (k == 0 || k == 2) ? 1 : 0;
    const int d = (k == 0) ? 1 : (k == 3 ? -1 : 0);
    int r = m + d;
    if (circ) { if (r < 0) r += Nc; if (r >= Nc) r -= Nc; }
    else      { if (r < 0) r = -r;  if (r >= Nc) r = 2 * Nc - 2 - r; }
    src_in = r;
    src_g = 2 * m + par;
  }
}

// Zero-fill as a KERNEL launch.  hipMemsetAsync is not used anywhere in this library: captured into a hipGraph its small
// fills (32-byte per-sample accumulators) came back wrong on ROCm 7.2 / gfx950 in the replay that follows a host-side
// hipStreamSynchronize - the accumulators kept garbage, DiffAugment's contrast mean blew up and the step trained on
// 1e24-sized images (scripts/debug_seg_sync5.py; tests/test_gpu_step.py::test_graph_replay_survives_host_sync).  A kernel
// node carries its arguments by value and replays correctly.
int dg_zero_f32(float* p, long n, hipStream_t s);

// Parameter blocks are part of the C ABI: include/dusty_gan_hip.h
typedef DgConv ConvP;
typedef DgWgrad WgradP;

// Optional optimizer epilogue (ADAM = true; Proj.weight in data-parallel runs): the finished gradient tile never goes to
// memory -- Adam (beta1 = 0, so no first moment), the EMA of G_ema and the bf16/fp32 shadow are applied to the
// parameter tile in place.  Reference: optim.Adam.step + ema_inplace, trainers/dcgan_amp.py:312,316,30-35.
struct AdamEpi {
  float *p, *v, *ema;
  void* shadow;
  int shadow_bf16;
  float gscale, lr, b2, eps, ema_decay;
  const unsigned long long* stepp;
};

int dg_wgrad_mfma_adam_launch(const WgradP* p, const AdamEpi* ad, hipStream_t stream, int fp32x3);

// DG_BF16X2 (include/dusty_gan_hip.h): element i = the bf16 pair (hi, lo) at bf16 index 2 i - i % 64 and 64 further
__host__ __device__ __forceinline__ long dg_x2_index(long i) { return 2 * i - (i & 63); }
__device__ __forceinline__ float dg_ld(const void* p, long i, int dtype) {
  if (dtype == DG_BF16X2) {
    const bf16* q = (const bf16*)p + dg_x2_index(i);
    return (float)q[0] + (float)q[64];
  }
  return dtype == DG_BF16 ? (float)((const bf16*)p)[i] : ((const float*)p)[i];
}
__device__ __forceinline__ void dg_st(void* p, long i, int dtype, float v) {
  if (dtype == DG_BF16X2) {
    bf16* q = (bf16*)p + dg_x2_index(i);
    const bf16 hi = (bf16)v;
    q[0] = hi;
    q[64] = (bf16)(v - (float)hi);
  } else if (dtype == DG_BF16) ((bf16*)p)[i] = (bf16)v;
  else ((float*)p)[i] = v;
}

__device__ __forceinline__ float dg_epilogue(float acc, float scale, int epi, float bias, float auxv) {
  float v = acc * scale + bias;
  if (epi == EPI_LRELU) v = (v > 0.f ? v : LRELU_SLOPE * v) * SQRT2;
  else if (epi == EPI_MASK) v *= (auxv > 0.f ? SQRT2 : LRELU_SLOPE * SQRT2);
  return v;
}

__device__ __forceinline__ float dg_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- deterministic cross-block sums (round 5).  A float atomicAdd from many blocks sums in arrival order: the last bits of
// the total - a per-sample image sum, a logit - differ from run to run, and everything downstream with them.  For
// accumulators inside the registered arena (dg_det_arena: the step's AccArena, plus a shadow of DG_DET_STRIDE bytes per float,
// zero at rest) a block instead adds its partial as 32.32 FIXED POINT to the slot's 64-bit shadow word - integer addition is
// associative, the order no longer matters - and takes a ticket; the block that draws the last ticket converts the total and
// adds it to the float ONCE, leaving the shadow zero.  The integer add is acknowledged (s_waitcnt vmcnt(0): atomics execute
// memory-side) before the ticket is taken, so the last ticket holder reads every contribution.  Range +-2^31, resolution
// 2.3e-10 per contribution; a non-finite or larger partial reaches the float as it is (a NaN stays a NaN).  Anything outside
// the arena, or with no arena registered, falls back to the float atomic.
// What the parity modes rely on (round-5 advice): the window.  Each CONTRIBUTION is range-checked, the running total is not -
// it wraps beyond +-2^31, so a slot's contributors must stay below that in sum: the step's slots hold per-sample image sums
// (<= H W = 2.6e5 at 128x2048), logits (O(1)), R1's per-sample |g|^2 (O(1)) and the augment adjoint's window sums: eight
// orders of magnitude of head-room.  The resolution is ABSOLUTE (2^-32 per contribution): a sum whose true value is far below
// 1e-6 - R1's |g|^2 late in a collapsed training run - keeps fewer significant bits here than a float atomic would; the
// logged penalty then reads a few 1e-10 off, its gradient is untouched (the tangent is formed from g itself, not from ssq).
// Ordering: the integer add is a memory-side atomic; `s_waitcnt vmcnt(0)` returns when it has been performed, and the ticket is
// a second memory-side atomic on the same line issued behind it - the last ticket holder's exchange follows every add.
// One 128-byte line per slot (round 6): device-scope atomics execute memory-side and adds to ONE line serialise - with 16 bytes
// per slot the 32 per-sample sums of a batch shared four lines, and every image-sized kernel that sums per sample paid 4-5 us
// for its 512-1536 adds (scripts/bench_pointwise.py: head_post_fwd 10.6 us with sums, 6.9 without).
#define DG_DET_STRIDE 128
struct DgDet { float* base; unsigned long long* shadow; long n; };
__device__ __forceinline__ void dg_acc_add(float* dst, float v, unsigned contributors, const DgDet d) {
  const long k = dst - d.base;
  if (d.shadow == nullptr || k < 0 || k >= d.n || contributors <= 1) { atomicAdd(dst, v); return; }
  unsigned long long* acc = d.shadow + (DG_DET_STRIDE / 8) * k;
  unsigned* ticket = (unsigned*)(acc + 1);
  if (!(fabsf(v) < 2147483000.f)) { atomicAdd(dst, v); v = 0.f; }
  const long long q = __double2ll_rn((double)v * 4294967296.0);
  atomicAdd(acc, (unsigned long long)q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (atomicAdd(ticket, 1u) == contributors - 1) {
    const long long tot = (long long)atomicExch(acc, 0ull);
    atomicExch(ticket, 0u);
    atomicAdd(dst, (float)((double)tot * (1.0 / 4294967296.0)));
  }
}

// dg_acc_add that also tells its caller whether it drew the LAST ticket: 1 = last (total = the sum of all `contributors`
// partials of this round, already added to *dst), 0 = not last, -1 = dst is outside the registered arena (a float atomic was
// issued; nobody knows who is last).  For a second-level sum by the last contributors only (blur_r1_tangent_kernel's batch mean:
// 32 adds to one word instead of 512).
__device__ __forceinline__ int dg_acc_add_last(float* dst, float v, unsigned contributors, const DgDet d, float& total) {
  const long k = dst - d.base;
  if (d.shadow == nullptr || k < 0 || k >= d.n) { atomicAdd(dst, v); return -1; }
  unsigned long long* acc = d.shadow + (DG_DET_STRIDE / 8) * k;
  unsigned* ticket = (unsigned*)(acc + 1);
  if (!(fabsf(v) < 2147483000.f)) { atomicAdd(dst, v); v = 0.f; }
  const long long q = __double2ll_rn((double)v * 4294967296.0);
  atomicAdd(acc, (unsigned long long)q);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (atomicAdd(ticket, 1u) == contributors - 1) {
    const long long tot = (long long)atomicExch(acc, 0ull);
    atomicExch(ticket, 0u);
    total = (float)((double)tot * (1.0 / 4294967296.0));
    atomicAdd(dst, total);
    return 1;
  }
  return 0;
}

// ---- two-word fixed point (round 6): float sums that do not depend on the order of their terms, without the one-word form's
// absolute resolution.  v = hi 2^-20 + lo 2^-60 with hi = rint(v 2^20) and lo = the residual, which float arithmetic forms exactly
// (|v| < 16: Sterbenz; above, v is a multiple of 2^-20 already and the residual is zero): every float down to 2^-36 in magnitude
// enters the sum EXACTLY, smaller ones rounded to 2^-60; range +-4e12 per term (larger or non-finite: the caller's float path),
// +-8.8e12 for the total.  Integer adds are associative, so the total is one pair of integers whatever the arrival order; it is
// rounded to float once.  Used by the bias-gradient sums of the kernels outside the timed path (conv_direct, conv_mfma).
__device__ __forceinline__ bool dg_fix2(float v, long long& hi, long long& lo) {
  if (!(fabsf(v) < 4.0e12f)) return false;
  const float t = rintf(v * 1048576.f);
  hi = (long long)t;
  const float r = __fmaf_rn(-t, 1.f / 1048576.f, v);
  lo = __double2ll_rn((double)r * 1152921504606846976.0);
  return true;
}
__device__ __forceinline__ float dg_fix2_value(long long hi, long long lo) {
  return (float)((double)hi * (1.0 / 1048576.0) + (double)lo * (1.0 / 1152921504606846976.0));
}
// ... across the workgroups of one launch, through the caller's staging scratch (DgConv.dbias_ws: zero on entry, left zero): word
// pair i belongs to bias channel i, the LAST u64 of the scratch is the ticket.  Every workgroup adds its channel sums
// (dg_dbias_ws_add), then ALL its threads call dg_dbias_ws_finish once: the adds are acknowledged (memory-side atomics), a ticket
// is drawn, and the workgroup that draws the last one adds the totals onto dbias - one float add per channel and launch.
#define DG_DBIAS_WS_WORDS (DG_DBIAS_SLOTS * DG_DBIAS_SLOT_FLOATS / 2)
__device__ __forceinline__ bool dg_dbias_ws_ok(const float* ws, int bias_mod) {
  return ws != nullptr && 2 * bias_mod + 2 <= DG_DBIAS_WS_WORDS;
}
__device__ __forceinline__ void dg_dbias_ws_add(float* ws, int ch, long long hi, long long lo) {
  unsigned long long* w = (unsigned long long*)ws;
  if (hi) atomicAdd(&w[2 * ch], (unsigned long long)hi);
  if (lo) atomicAdd(&w[2 * ch + 1], (unsigned long long)lo);
}
__device__ __forceinline__ void dg_dbias_ws_finish(float* ws, int bias_mod, float* dbias) {
  unsigned long long* w = (unsigned long long*)ws;
  __shared__ unsigned s_dbias_ticket;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned nblk = gridDim.x * gridDim.y * gridDim.z;
  const unsigned nthr = blockDim.x * blockDim.y * blockDim.z;
  const unsigned t = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
  if (t == 0) s_dbias_ticket = atomicAdd((unsigned*)&w[DG_DBIAS_WS_WORDS - 1], 1u);
  __syncthreads();
  if (s_dbias_ticket != nblk - 1) return;
  for (int i = (int)t; i < bias_mod; i += (int)nthr) {
    const long long hi = (long long)atomicExch(&w[2 * i], 0ull), lo = (long long)atomicExch(&w[2 * i + 1], 0ull);
    if (hi | lo) atomicAdd(&dbias[i], dg_fix2_value(hi, lo));
  }
  if (t == 0) atomicExch((unsigned*)&w[DG_DBIAS_WS_WORDS - 1], 0u);
}

// tanh for the depth head (Generator.forward, models/gans/dcgan_eqlr.py:71) in ~17 VALU instructions: libm's tanhf made
// head_post_fwd4_kernel VALU-bound (8.4 M pixels x ~45 instructions = the whole 11 us of the launch, round 6).
//   |x| >= 0.25: (1 - e) / (1 + e) with e = exp(-2 |x|) in (0, 0.61]: no cancellation, ~2 ulp
//   |x| <  0.25: the odd Taylor polynomial through x^9 (next term < 9e-9 relative at 0.25)
__device__ __forceinline__ float dg_tanh(float x) {
  const float ax = fabsf(x);
  const float e = __expf(-2.f * ax);
  const float big = (1.f - e) * __frcp_rn(1.f + e);
  const float x2 = x * x;
  const float small = ax + ax * x2 * (-0.33333333333f + x2 * (0.13333333333f + x2 * (-0.05396825397f + x2 * 0.02186948854f)));
  return copysignf(ax < 0.25f ? small : big, x);
}

// Block-wide sum (blockDim.x a multiple of 64, <= 1024); result valid in thread 0.
__device__ __forceinline__ float dg_block_sum(float v, float* red /* >= 16 floats of LDS */) {
  v = dg_wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) red[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 0; i < nw; ++i) r += red[i];
  }
  return r;
}
