// Optimizer / parameter-state kernels: fused Adam (+EMA, + low-precision shadow write), the transposed weight
// shadow used by the MFMA forward kernels, and the Philox4x32-10 generator for z / Gumbel / DiffAugment draws.
#include "common.h"
#include "thin_up_frag.h"
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4_opt;

// torch.optim.Adam (no weight decay / amsgrad) as configured at trainers/dcgan_amp.py:116-125, fused with
// ema_inplace (:30-35) and with the T-typed shadow copy the conv kernels read.  All buffers are flat.
//   g <- grad * gscale (gscale = 1/world_size after a SUM all-reduce)
//   m <- b1 m + (1-b1) g ; v <- b2 v + (1-b2) g^2 ; p <- p - (lr/bc1) m / (sqrt(v)/sqrt(bc2) + eps)
//   ema <- d ema + (1-d) p   (only where ema != null)
template <typename T>
__global__ __launch_bounds__(256) void adam_ema_kernel(float* __restrict__ p, const float* __restrict__ grad,
                                                       float* __restrict__ m, float* __restrict__ v,
                                                       float* __restrict__ ema, T* __restrict__ shadow, long n4,
                                                       float gscale, float lr_bc1, float inv_sqrt_bc2, float b1,
                                                       float b2, float eps, float ema_decay,
                                                       const unsigned long long* __restrict__ stepp, float lr) {
  if (stepp) {  // step count in device memory (graph replay): bias corrections computed here
    const float t = (float)(*stepp + 1ull);
    lr_bc1 = lr / (1.f - (b1 > 0.f ? powf(b1, t) : 0.f));
    inv_sqrt_bc2 = rsqrtf(1.f - powf(b2, t));
  }
  // 16 bytes per lane per stream, grid-stride.  m == nullptr (beta1 == 0: exp_avg IS the scaled gradient) drops two
  // of the nine fp32 streams; the exported optimizer state rebuilds exp_avg from the gradient buffer.
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 g4 = ((const float4*)grad)[i];
    const float4 v4 = ((const float4*)v)[i];
    float4 p4 = ((const float4*)p)[i];
    float4 m4 = m ? ((const float4*)m)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 e4 = ema ? ((const float4*)ema)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float gg[4] = {g4.x * gscale, g4.y * gscale, g4.z * gscale, g4.w * gscale};
    float pv[4] = {p4.x, p4.y, p4.z, p4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w};
    float ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float mi = b1 * mv[k] + (1.f - b1) * gg[k];
      const float vi = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
      mv[k] = mi;
      vv[k] = vi;
      pv[k] = pv[k] - lr_bc1 * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
      ev[k] = ema_decay * ev[k] + (1.f - ema_decay) * pv[k];
    }
    ((float4*)p)[i] = make_float4(pv[0], pv[1], pv[2], pv[3]);
    ((float4*)v)[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (m) ((float4*)m)[i] = make_float4(mv[0], mv[1], mv[2], mv[3]);
    if (ema) ((float4*)ema)[i] = make_float4(ev[0], ev[1], ev[2], ev[3]);
    if (shadow) {  // one 8- or 16-byte store per lane (four 2-byte stores were a quarter of this kernel's store instructions)
      if constexpr (sizeof(T) == 2) {
        const T h[4] = {(T)pv[0], (T)pv[1], (T)pv[2], (T)pv[3]};
        uint2 pk;
        pk.x = (unsigned)__builtin_bit_cast(unsigned short, h[0]) | ((unsigned)__builtin_bit_cast(unsigned short, h[1]) << 16);
        pk.y = (unsigned)__builtin_bit_cast(unsigned short, h[2]) | ((unsigned)__builtin_bit_cast(unsigned short, h[3]) << 16);
        ((uint2*)shadow)[i] = pk;
      } else {
        ((float4*)shadow)[i] = make_float4(pv[0], pv[1], pv[2], pv[3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Round 6: the optimizer of a network as ONE launch that also FORMS the gradients it consumes (reference:
// trainers/dcgan_amp.py:238 / :312-316 - scaler.step(optim) + ema_inplace - behind loss.backward()'s last reductions).
// Before: dg_batch_wsum (final conv's R1 term) -> dg_wgrad_reduce (sums the split-K partial tiles and the bias-gradient rows
// into the gradient buffer) -> dg_adam_ema_step_dev (reads it back) -> dg_transpose_shadow_multi (re-reads the updated
// master for the [tap][co][ci] shadows): four dependent launches, the gradient written and read once for nothing.
// Here a workgroup owns a piece of the parameter buffer, sums that piece's partial tiles in a fixed order (bit-reproducible),
// applies Adam (beta1 = 0: exp_avg is the gradient itself) + the EMA, and writes master, exp_avg_sq, EMA, the gradient (kept:
// checkpoints rebuild exp_avg from it, tests read it), the compute-type shadow and - for the fat conv layers, whose piece is a
// (tap, 32 ci x 32 co) tile - the transposed shadow through an LDS tile.
//   kind 0  flat piece of 1024 elements, <= 64 partial rows summed per thread (8 loads in flight); optional extra term
//           g += ws_scale * sum_b ws_coef[b] * ws_src[b][i] (the final conv's weight gradient from the R1 tangent, dg_batch_wsum)
//   kind 2  flat piece of 64 elements whose > 64 partial rows are split over 16 thread groups and meet in LDS
//   kind 1  conv tile: 1024 consecutive elements of a [16][Ci][Co] segment as (R ci rows) x (CW co columns), Co = 64 (R 16) or a
//           multiple of 128 (R 8), Ci % R == 0; also writes the [16][Co][Ci] shadow
typedef DgOptSeg OptSeg;               // (include/dusty_gan_hip.h)
#define OPT_MAX_SEG DG_OPT_MAX_SEG
struct OptSegs { OptSeg s[OPT_MAX_SEG]; int n; int total_blocks; };

template <typename ST>
__global__ __launch_bounds__(256) void adam_fused_kernel(float* __restrict__ p, float* __restrict__ grad, float* __restrict__ v,
                                                         float* __restrict__ ema, ST* __restrict__ shadow, OptSegs segs,
                                                         float gscale, float lr, float b2, float eps, float ema_decay,
                                                         const unsigned long long* __restrict__ stepp) {
  __shared__ f32x4_opt s_part[16][17];
  __shared__ float tile[16 * 65 > 8 * 129 ? 16 * 65 : 8 * 129];   // kind 1: [R][CW + 1], (R, CW) = (16, 64) or (8, 128)
  int k = 0;
  for (int i = 1; i < segs.n; ++i)
    if ((int)blockIdx.x >= segs.s[i].first_block) k = i;
  const OptSeg& sg = segs.s[k];
  const int blk = (int)blockIdx.x - sg.first_block;
  const float t = (float)(*stepp + 1ull);
  const float inv_sqrt_bc2 = rsqrtf(1.f - powf(b2, t));
  // Adam (beta1 = 0) + EMA + every store of four consecutive parameters at flat float4 index i4, given their gradient sum.
  // The parameter-state loads are issued by `state_load` BEFORE the partial rows are summed (they do not depend on the sum:
  // more bytes in flight, a shorter dependent chain per workgroup).
  struct State { float4 v4, p4, e4; f32x4_opt g0; };
  auto state_load = [&](long i4, bool acc) {
    State st;
    st.v4 = ((const float4*)v)[i4];
    st.p4 = ((const float4*)p)[i4];
    st.e4 = ema ? ((const float4*)ema)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
    st.g0 = acc ? ((const f32x4_opt*)grad)[i4] : f32x4_opt{0.f, 0.f, 0.f, 0.f};
    return st;
  };
  auto update4 = [&](long i4, f32x4_opt g, const State& st, float (&pn)[4]) {
    const float4 v4 = st.v4, p4 = st.p4, e4 = st.e4;
    ((f32x4_opt*)grad)[i4] = g;
    const float gg[4] = {g[0] * gscale, g[1] * gscale, g[2] * gscale, g[3] * gscale};
    float vv[4] = {v4.x, v4.y, v4.z, v4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
    pn[0] = p4.x; pn[1] = p4.y; pn[2] = p4.z; pn[3] = p4.w;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float vi = b2 * vv[q] + (1.f - b2) * gg[q] * gg[q];
      vv[q] = vi;
      pn[q] = pn[q] - lr * (gg[q] / (sqrtf(vi) * inv_sqrt_bc2 + eps));
      ev[q] = ema_decay * ev[q] + (1.f - ema_decay) * pn[q];
    }
    ((float4*)p)[i4] = make_float4(pn[0], pn[1], pn[2], pn[3]);
    ((float4*)v)[i4] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (ema) ((float4*)ema)[i4] = make_float4(ev[0], ev[1], ev[2], ev[3]);
    if (shadow) {
      if constexpr (sizeof(ST) == 2) {
        const ST h[4] = {(ST)pn[0], (ST)pn[1], (ST)pn[2], (ST)pn[3]};
        uint2 pk;
        pk.x = (unsigned)__builtin_bit_cast(unsigned short, h[0]) | ((unsigned)__builtin_bit_cast(unsigned short, h[1]) << 16);
        pk.y = (unsigned)__builtin_bit_cast(unsigned short, h[2]) | ((unsigned)__builtin_bit_cast(unsigned short, h[3]) << 16);
        ((uint2*)shadow)[i4] = pk;
      } else {
        ((float4*)shadow)[i4] = make_float4(pn[0], pn[1], pn[2], pn[3]);
      }
    }
  };
  // the sum of a piece's partial rows for the float4 at index e4 of the SEGMENT, rows sp0, sp0 + step, ... (8 loads in flight)
  auto part_sum = [&](long e4, int sp0, int step) {
    f32x4_opt acc = {0.f, 0.f, 0.f, 0.f};
    const f32x4_opt* src = (const f32x4_opt*)sg.part + e4;
    const long stride = sg.numel / 4;
    int sp = sp0;
    for (; sp + 7 * step < sg.splits; sp += 8 * step) {
      f32x4_opt r[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) r[j] = __builtin_nontemporal_load(src + (long)(sp + j * step) * stride);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += r[j];
    }
    for (; sp < sg.splits; sp += step) acc += __builtin_nontemporal_load(src + (long)sp * stride);
    return acc;
  };
  if (sg.kind == 2) {
    const int el = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const long e4 = (long)blk * 16 + el;
    const bool in = 4 * e4 < sg.numel;
    s_part[grp][el] = in ? part_sum(e4, grp, 16) : f32x4_opt{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (grp == 0 && in) {
      f32x4_opt g = s_part[0][el];
#pragma unroll
      for (int j = 1; j < 16; ++j) g += s_part[j][el];
      const long i4 = sg.off / 4 + e4;
      const State st = state_load(i4, sg.accumulate != 0);
      g += st.g0;
      float pn[4];
      update4(i4, g, st, pn);
    }
    return;
  }
  if (sg.kind == 0) {
    const long e4 = (long)blk * 256 + threadIdx.x;
    if (4 * e4 >= sg.numel) return;
    const long i4 = sg.off / 4 + e4;
    const State st = state_load(i4, sg.accumulate != 0);
    f32x4_opt g = {0.f, 0.f, 0.f, 0.f};
    if (sg.part) g = part_sum(e4, 0, 1);
    g += st.g0;
    if (sg.ws_src) {                               // + ws_scale * sum_b coef[b] * src[b][4 e4 .. 4 e4 + 3]
      f32x4_opt w = {0.f, 0.f, 0.f, 0.f};
      for (int b0 = 0; b0 < sg.ws_n; b0 += 8) {
        f32x4_opt r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int b = b0 + j;
          r[j] = f32x4_opt{0.f, 0.f, 0.f, 0.f};
          if (b < sg.ws_n) {
            if (sg.ws_bf16) {
              const uint2 q = *(const uint2*)((const bf16*)sg.ws_src + (long)b * sg.ws_stride + 4 * e4);
              r[j] = f32x4_opt{__builtin_bit_cast(float, q.x << 16), __builtin_bit_cast(float, q.x & 0xffff0000u),
                               __builtin_bit_cast(float, q.y << 16), __builtin_bit_cast(float, q.y & 0xffff0000u)};
            } else {
              r[j] = *(const f32x4_opt*)((const float*)sg.ws_src + (long)b * sg.ws_stride + 4 * e4);
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (b0 + j < sg.ws_n) w += (sg.ws_coef ? sg.ws_coef[b0 + j] : 1.f) * r[j];
      }
      g += sg.ws_scale * w;
    }
    float pn[4];
    update4(i4, g, st, pn);
    return;
  }
  // kind 1: a tile of 1024 elements = R rows (ci) x CW columns (co) of one tap, CW = min(Co, 128), R = 1024 / CW: CONTIGUOUS
  // runs of 4 CW bytes in the master and in every partial row (a first version walked 32 x 32 tiles - 128-byte runs - and
  // summed the D network's 168 MB of partial tiles no faster than the reduce launch alone).  Thread = (row ty, columns 4 tx..).
  const int Ci = sg.ci, Co = sg.co;
  const int CW = Co < 128 ? Co : 128, R = 1024 / CW, tcw = Co / CW, tr = Ci / R;
  int tt = blk;
  const int tap = tt / (tr * tcw);
  tt -= tap * tr * tcw;
  const int ci0 = (tt / tcw) * R, co0 = (tt % tcw) * CW;
  const int tx = threadIdx.x % (CW / 4), ty = threadIdx.x / (CW / 4);
  const long e4 = (((long)tap * Ci + ci0 + ty) * Co + co0) / 4 + tx;
  const long i4 = sg.off / 4 + e4;
  const State st = state_load(i4, sg.accumulate != 0);
  f32x4_opt g = {0.f, 0.f, 0.f, 0.f};
  if (sg.part) g = part_sum(e4, 0, 1);
  g += st.g0;
  float pn[4];
  update4(i4, g, st, pn);
  float* tl = (float*)tile;                        // [R][CW + 1]
#pragma unroll
  for (int q = 0; q < 4; ++q) tl[ty * (CW + 1) + 4 * tx + q] = pn[q];
  __syncthreads();
  // transposed shadow: row co0 + c of [co][ci] gets the R consecutive ci of this tile (R = 8: 16 bytes of bf16 per row)
  for (int c = threadIdx.x; c < CW; c += 256) {
    ST* dst = (ST*)sg.shadow_t + ((long)tap * Co + co0 + c) * Ci + ci0;
    for (int r0 = 0; r0 < R; r0 += 4) {
      const float o[4] = {tl[(r0 + 0) * (CW + 1) + c], tl[(r0 + 1) * (CW + 1) + c], tl[(r0 + 2) * (CW + 1) + c],
                          tl[(r0 + 3) * (CW + 1) + c]};
      if constexpr (sizeof(ST) == 2) {
        const ST h[4] = {(ST)o[0], (ST)o[1], (ST)o[2], (ST)o[3]};
        uint2 pk;
        pk.x = (unsigned)__builtin_bit_cast(unsigned short, h[0]) | ((unsigned)__builtin_bit_cast(unsigned short, h[1]) << 16);
        pk.y = (unsigned)__builtin_bit_cast(unsigned short, h[2]) | ((unsigned)__builtin_bit_cast(unsigned short, h[3]) << 16);
        *(uint2*)(dst + r0) = pk;
      } else {
        *(float4*)(dst + r0) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Adam + EMA for Proj.weight with its gradient computed IN the kernel.  Proj is a linear layer (nz -> h0*w0*C,
// models/gans/dcgan_eqlr.py:6-13) whose [Np][K] fp32 gradient is 96 % of G's gradient bytes (268 MB at 64x1024):
//     g[n'][k] = wscale * sum_b dp0[b][n'] * z[b][k]          (b < nb <= 64 samples)
// costs 16 dot2 per element from LDS-resident operands, so forming it here instead of writing it with the wgrad GEMM
// and reading it back removes 2 x 268 MB of HBM traffic from the step (beta1 == 0: exp_avg is never stored).
// Operands bf16: z2[bp][k] = (z[2bp][k], z[2bp+1][k]) and dT[row][bp] = (dp0[2bp][row], dp0[2bp+1][row]) packed pairs.
typedef __attribute__((ext_vector_type(2))) __bf16 ad_bf16x2;
#define APF_ROWS 64

template <typename ST>
__global__ __launch_bounds__(256) void adam_proj_fused_kernel(float* __restrict__ p, float* __restrict__ v,
                                                              float* __restrict__ ema, ST* __restrict__ shadow,
                                                              const bf16* __restrict__ dp0, const bf16* __restrict__ zT,
                                                              int nb, long Np, int K, float wscale, float gscale, float lr,
                                                              float b2, float eps, float ema_decay,
                                                              const unsigned long long* __restrict__ stepp) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int nbp = nb / 2;
  unsigned* z2 = (unsigned*)smem;                    // [nbp][K]
  unsigned* dT = z2 + (long)nbp * K;                 // [APF_ROWS][nbp]
  const int tid = threadIdx.x;
  const long r0 = (long)blockIdx.x * APF_ROWS;
  const unsigned short* zr = (const unsigned short*)zT;
  const unsigned short* dr = (const unsigned short*)dp0;
  for (int i = tid; i < nbp * K; i += 256) {
    const int bp = i / K, k = i - bp * K;
    z2[i] = (unsigned)zr[(long)(2 * bp) * K + k] | ((unsigned)zr[(long)(2 * bp + 1) * K + k] << 16);
  }
  for (int i = tid; i < APF_ROWS * nbp; i += 256) {
    const int row = i % APF_ROWS, bp = i / APF_ROWS;
    unsigned d = 0;
    if (r0 + row < Np) d = (unsigned)dr[(long)(2 * bp) * Np + r0 + row] | ((unsigned)dr[(long)(2 * bp + 1) * Np + r0 + row] << 16);
    dT[row * nbp + bp] = d;
  }
  __syncthreads();
  const float t = (float)(*stepp + 1ull);
  const float inv_sqrt_bc2 = rsqrtf(1.f - powf(b2, t));
  const int quads = K / 4, kq = tid % quads, rpp = 256 / quads;
  for (int row = tid / quads; row < APF_ROWS && r0 + row < Np; row += rpp) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int bp = 0; bp < nbp; ++bp) {
      const ad_bf16x2 d = __builtin_bit_cast(ad_bf16x2, dT[row * nbp + bp]);
      const uint4 z4 = *(const uint4*)(z2 + (long)bp * K + 4 * kq);
      acc[0] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ad_bf16x2, z4.x), d, acc[0], false);
      acc[1] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ad_bf16x2, z4.y), d, acc[1], false);
      acc[2] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ad_bf16x2, z4.z), d, acc[2], false);
      acc[3] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(ad_bf16x2, z4.w), d, acc[3], false);
    }
    const long i4 = ((r0 + row) * K) / 4 + kq;
    const float4 v4 = ((const float4*)v)[i4];
    const float4 p4 = ((const float4*)p)[i4];
    const float4 e4 = ema ? ((const float4*)ema)[i4] : make_float4(0.f, 0.f, 0.f, 0.f);
    float pv[4] = {p4.x, p4.y, p4.z, p4.w}, vv[4] = {v4.x, v4.y, v4.z, v4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float g = acc[k] * wscale * gscale;
      const float vi = b2 * vv[k] + (1.f - b2) * g * g;
      vv[k] = vi;
      pv[k] = pv[k] - lr * (g / (sqrtf(vi) * inv_sqrt_bc2 + eps));
      ev[k] = ema_decay * ev[k] + (1.f - ema_decay) * pv[k];
    }
    ((float4*)p)[i4] = make_float4(pv[0], pv[1], pv[2], pv[3]);
    ((float4*)v)[i4] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    if (ema) ((float4*)ema)[i4] = make_float4(ev[0], ev[1], ev[2], ev[3]);
    if (shadow) {
      if constexpr (sizeof(ST) == 2) {
        const ST h[4] = {(ST)pv[0], (ST)pv[1], (ST)pv[2], (ST)pv[3]};
        uint2 pk;
        pk.x = (unsigned)__builtin_bit_cast(unsigned short, h[0]) | ((unsigned)__builtin_bit_cast(unsigned short, h[1]) << 16);
        pk.y = (unsigned)__builtin_bit_cast(unsigned short, h[2]) | ((unsigned)__builtin_bit_cast(unsigned short, h[3]) << 16);
        ((uint2*)shadow)[i4] = pk;
      } else {
        ((float4*)shadow)[i4] = make_float4(pv[0], pv[1], pv[2], pv[3]);
      }
    }
  }
}

// all conv segments of a network in one launch: desc[5 i + (0..4)] = (source element offset in `master`, destination
// pointer, Ci, Co, first tile index); a tile = (tap, 32 x 32 block of [ci][co]) as in transpose_shadow_kernel
struct CounterAdds {
  unsigned long long* c[8]; unsigned long long d[8]; int k;
  int snap_idx, snap_n, snap_ring; const float* snap_src; float* snap_dst;   // snap_idx < 0: no snapshot
};
__device__ __forceinline__ void counter_adds_body(const CounterAdds& a) {   // (one block's threads 0 .. k-1)
  if ((int)threadIdx.x >= a.k) return;
  const unsigned long long v = *a.c[threadIdx.x];
  *a.c[threadIdx.x] = v + a.d[threadIdx.x];
  if ((int)threadIdx.x == a.snap_idx) {          // the thread that advances the counter also files the snapshot under its old value
    float* dst = a.snap_dst + (long)(v % (unsigned long long)a.snap_ring) * a.snap_n;
    for (int i = 0; i < a.snap_n; ++i) dst[i] = a.snap_src[i];
    __threadfence_system();                      // (dst may be mapped host memory)
  }
}
struct UpFrags { DgUpFrag f[4]; int n; int first_block; };   // blocks >= first_block: 3 x UP_FRAG_BLOCKS per fragment set
// (+ optionally one last block that advances the step's counters and files its scalars: dg_transpose_shadow_multi_tail)
template <typename T>
__global__ __launch_bounds__(256) void transpose_shadow_multi_kernel(const float* __restrict__ master,
                                                                     const long long* __restrict__ desc, int nseg,
                                                                     UpFrags uf, CounterAdds ca, int ca_block) {
  __shared__ float tile[32][33];
  if ((int)blockIdx.x == ca_block) { counter_adds_body(ca); return; }
  if (uf.n > 0 && (int)blockIdx.x >= uf.first_block) {
    const int job = (int)blockIdx.x - uf.first_block, per = 3 * UP_FRAG_BLOCKS;
    const DgUpFrag& f = uf.f[job / per];
    const int cls = (job % per) / UP_FRAG_BLOCKS, e = ((job % per) % UP_FRAG_BLOCKS) * 256 + threadIdx.x;
    const float* w = master + f.off;
    up_frag_element(cls, e, f.N, f.Hc, f.adj,
                    [&](int tap, int n, int k) { return (float)(bf16)w[tap * f.m_st + n * f.m_sn + k * f.m_sk]; },
                    (unsigned char*)f.frag);
    return;
  }
  int sidx = 0;
  for (int i = 1; i < nseg; ++i)
    if ((long long)blockIdx.x >= desc[5 * i + 4]) sidx = i;
  const long long* d = desc + 5 * sidx;
  const int Ci = (int)d[2], Co = (int)d[3];
  const int tci = (Ci + 31) / 32, tco = (Co + 31) / 32;
  int t = (int)(blockIdx.x - d[4]);
  const int tap = t / (tci * tco);
  t -= tap * tci * tco;
  const int ci0 = (t / tco) * 32, co0 = (t % tco) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* src = master + d[0] + (long)tap * Ci * Co;
  T* dst = (T*)d[1] + (long)tap * Ci * Co;
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    tile[r][tx] = (ci < Ci && co < Co) ? src[(long)ci * Co + co] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    if (ci < Ci && co < Co) dst[(long)co * Ci + ci] = (T)tile[tx][r];
  }
}

template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (T)src[i];
}

// fp32 <-> DG_BF16X2 (include/dusty_gan_hip.h: per 64 elements 128 bytes of hi = bf16(x), then 128 bytes of lo = bf16(x - hi)):
// a thread owns 8 consecutive elements = one 16-byte piece of each half
struct CastItems { const float* src[16]; void* dst[16]; long first8[17]; int n; };
__device__ __forceinline__ void x2_pack8(const float* __restrict__ src, unsigned short* __restrict__ dst, long i8) {
  const float4 a = ((const float4*)src)[2 * i8], b = ((const float4*)src)[2 * i8 + 1];
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  unsigned hi[4], lo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bf16 h0 = (bf16)v[2 * e], h1 = (bf16)v[2 * e + 1];
    const bf16 l0 = (bf16)(v[2 * e] - (float)h0), l1 = (bf16)(v[2 * e + 1] - (float)h1);
    hi[e] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    lo[e] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
  }
  unsigned short* q = dst + dg_x2_index(8 * i8);
  *(uint4*)q = make_uint4(hi[0], hi[1], hi[2], hi[3]);
  *(uint4*)(q + 64) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}
__global__ __launch_bounds__(256) void cast_x2_kernel(CastItems it) {
  const long i8 = (long)blockIdx.x * 256 + threadIdx.x;
  int k = 0;
#pragma unroll
  for (int i = 1; i < 16; ++i)
    if (i < it.n && i8 >= it.first8[i]) k = i;
  if (i8 >= it.first8[it.n]) return;
  x2_pack8(it.src[k], (unsigned short*)it.dst[k], i8 - it.first8[k]);
}
__global__ __launch_bounds__(256) void uncast_x2_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, long n8) {
  const long i8 = (long)blockIdx.x * 256 + threadIdx.x;
  if (i8 >= n8) return;
  const unsigned short* q = src + dg_x2_index(8 * i8);
  const uint4 h = *(const uint4*)q, l = *(const uint4*)(q + 64);
  const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
  float v[8];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[2 * e] = __builtin_bit_cast(float, hw[e] << 16) + __builtin_bit_cast(float, lw[e] << 16);
    v[2 * e + 1] = __builtin_bit_cast(float, hw[e] & 0xffff0000u) + __builtin_bit_cast(float, lw[e] & 0xffff0000u);
  }
  ((float4*)dst)[2 * i8] = make_float4(v[0], v[1], v[2], v[3]);
  ((float4*)dst)[2 * i8 + 1] = make_float4(v[4], v[5], v[6], v[7]);
}
template <typename T>
__global__ void uncast_kernel(const T* __restrict__ src, float* __restrict__ dst, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (float)src[i];
}

// master [16][ci][co] fp32 -> shadow [16][co][ci] T, through a 32x32 LDS tile per (tap, ci-tile, co-tile)
template <typename T>
__global__ __launch_bounds__(256) void transpose_shadow_kernel(const float* __restrict__ src, T* __restrict__ dst,
                                                               int Ci, int Co) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int ci0 = blockIdx.y * 32, co0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* s = src + (long)tap * Ci * Co;
  T* d = dst + (long)tap * Ci * Co;
  for (int r = ty; r < 32; r += 8) {
    const int ci = ci0 + r, co = co0 + tx;
    tile[r][tx] = (ci < Ci && co < Co) ? s[(long)ci * Co + co] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int co = co0 + r, ci = ci0 + tx;
    if (ci < Ci && co < Co) d[(long)co * Ci + ci] = (T)tile[tx][r];
  }
}

// ----------------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11; the counter-based generator torch's device RNG also uses).  One call of
// philox(counter=(i,0,0,0) + offset, key=seed) yields 4 x 32 random bits for element group i.
__host__ __device__ inline void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t k0,
                                             uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c0 = n0; c1 = n1; c2 = n2; c3 = n3;
}

__host__ __device__ inline void philox4x32_10(uint64_t seed, uint64_t ctr_lo, uint64_t ctr_hi, uint32_t out[4]) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32), c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c0, c1, c2, c3, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// raw bits: out[4*i + j] = philox(seed, counter = (offset + i, stream))[j]
__global__ void philox_bits_kernel(uint64_t seed, uint64_t stream, uint64_t offset, long n4, uint32_t* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  uint32_t r[4];
  philox4x32_10(seed, offset + (uint64_t)i, stream, r);
  out[4 * i + 0] = r[0]; out[4 * i + 1] = r[1]; out[4 * i + 2] = r[2]; out[4 * i + 3] = r[3];
}

// kind 0: uniform in [0,1) with 24 bits (bits >> 8) * 2^-24
// kind 1: standard normal by Box-Muller on pairs (u in (0,1]: 1 - uniform)
// kind 2: uniform(lo,hi)
// kind 3: integer in [ilo, ihi) as int32 (modulo of the 32-bit draw; range << 2^32 so the bias is < 2^-20)
__global__ void philox_fill_kernel(uint64_t seed, uint64_t stream, uint64_t offset, int kind, float lo, float hi,
                                   int ilo, int ihi, long n, void* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;  // group of 4 outputs
  if (4 * i >= n) return;
  uint32_t r[4];
  philox4x32_10(seed, offset + (uint64_t)i, stream, r);
  float f[4];
  int q[4];
  const float s24 = 1.f / 16777216.f;
  if (kind == 1) {
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      const float u1 = 1.f - (float)(r[j] >> 8) * s24;  // (0,1]
      const float u2 = (float)(r[j + 1] >> 8) * s24;
      const float rad = sqrtf(-2.f * logf(u1));
      f[j] = rad * cosf(6.283185307179586f * u2);
      f[j + 1] = rad * sinf(6.283185307179586f * u2);
    }
  } else if (kind == 3) {
    const uint32_t range = (uint32_t)(ihi - ilo);
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = ilo + (int)(r[j] % range);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u = (float)(r[j] >> 8) * s24;
      f[j] = kind == 2 ? lo + (hi - lo) * u : u;
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long o = 4 * i + j;
    if (o < n) {
      if (kind == 3) ((int*)out)[o] = q[j];
      else ((float*)out)[o] = f[j];
    }
  }
}

// One DiffAugment parameter set per sample (utils/diff_augment.py:27-28,36-38,46-48,59-60,86-87): three
// uniform(-1,1) draws and four integer draws, from two Philox counters per sample.
__global__ void aug_draw_kernel(uint64_t seed, uint64_t stream, uint64_t offset, int B, int sh, int sw, int nx, int ny,
                                float* __restrict__ uf, int* __restrict__ qi) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  uint32_t r0[4], r1[4];
  philox4x32_10(seed, offset + 2 * (uint64_t)b, stream, r0);
  philox4x32_10(seed, offset + 2 * (uint64_t)b + 1, stream, r1);
  const float s24 = 1.f / 16777216.f;
  for (int j = 0; j < 3; ++j) uf[j * B + b] = -1.f + 2.f * (float)(r0[j] >> 8) * s24;
  qi[0 * B + b] = -sh + (int)(r1[0] % (uint32_t)(2 * sh + 1));
  qi[1 * B + b] = -sw + (int)(r1[1] % (uint32_t)(2 * sw + 1));
  qi[2 * B + b] = (int)(r1[2] % (uint32_t)nx);
  qi[3 * B + b] = (int)(r1[3] % (uint32_t)ny);
}

// ---- device-resident counters: the same draws with the Philox offset / Adam step count read from device memory, so
//      a whole training step can be captured once in a hipGraph and replayed (a host-side offset would be frozen in
//      the captured kernel arguments).  dg_counter_add advances a counter after its consumers (stream order).
__global__ void counter_add_kernel(unsigned long long* c, unsigned long long delta) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *c += delta;
}
__global__ void counter_add_multi_kernel(CounterAdds a) {
  if (blockIdx.x == 0) counter_adds_body(a);
}
// (the bodies are device functions: dg_step_prologue's kernel runs the same draws as extra blocks of one launch)
__device__ __forceinline__ void philox_fill_body(uint64_t seed, uint64_t stream, uint64_t offset, int kind, float lo, float hi,
                                                 int ilo, int ihi, long n, void* __restrict__ out, bf16* __restrict__ out_bf16,
                                                 long i) {
  if (4 * i >= n) return;
  uint32_t r[4];
  philox4x32_10(seed, offset + (uint64_t)i, stream, r);
  const float s24 = 1.f / 16777216.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long o = 4 * i + j;
    if (o >= n) break;
    if (kind == 3) {
      ((int*)out)[o] = ilo + (int)(r[j] % (uint32_t)(ihi - ilo));
    } else if (kind == 1) {
      const int a = j & ~1;
      const float u1 = 1.f - (float)(r[a] >> 8) * s24, u2 = (float)(r[a + 1] >> 8) * s24;
      const float rad = sqrtf(-2.f * logf(u1));
      const float v = (j & 1) ? rad * sinf(6.283185307179586f * u2) : rad * cosf(6.283185307179586f * u2);
      ((float*)out)[o] = v;
      if (out_bf16) out_bf16[o] = (bf16)v;
    } else {
      const float u = (float)(r[j] >> 8) * s24;
      const float v = kind == 2 ? lo + (hi - lo) * u : u;
      ((float*)out)[o] = v;
      if (out_bf16) out_bf16[o] = (bf16)v;
    }
  }
}
__global__ void philox_fill_dev_kernel(uint64_t seed, uint64_t stream, const unsigned long long* __restrict__ offp,
                                       int kind, float lo, float hi, int ilo, int ihi, long n, void* __restrict__ out) {
  philox_fill_body(seed, stream, *offp, kind, lo, hi, ilo, ihi, n, out, nullptr, (long)blockIdx.x * blockDim.x + threadIdx.x);
}
// GumbelSigmoid.logistic_noise (models/dusty.py:30-36) straight from the generator: element o of U1 is word o & 3 of Philox
// counter offset + o / 4, U2 the same (n + 3) / 4 counters further - exactly what two uniform fills of n elements followed
// by dg_logistic_noise produce, in one launch and without the two 4 n-byte round trips
__device__ __forceinline__ void philox_logistic_body(uint64_t seed, uint64_t stream, uint64_t offset, float eps, long n,
                                                     float* __restrict__ out, long i) {
  if (4 * i >= n) return;
  const uint64_t n4 = (uint64_t)((n + 3) / 4);
  uint32_t r1[4], r2[4];
  philox4x32_10(seed, offset + (uint64_t)i, stream, r1);
  philox4x32_10(seed, offset + n4 + (uint64_t)i, stream, r2);
  const float s24 = 1.f / 16777216.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long o = 4 * i + j;
    if (o >= n) break;
    const float u1 = (float)(r1[j] >> 8) * s24, u2 = (float)(r2[j] >> 8) * s24;
    out[o] = -logf(logf(u1 + eps) / logf(u2 + eps) + eps);
  }
}
__global__ void philox_logistic_dev_kernel(uint64_t seed, uint64_t stream, const unsigned long long* __restrict__ offp,
                                           float eps, long n, float* __restrict__ out) {
  philox_logistic_body(seed, stream, *offp, eps, n, out, (long)blockIdx.x * blockDim.x + threadIdx.x);
}
__device__ __forceinline__ void aug_draw_body(uint64_t seed, uint64_t stream, uint64_t offset, int B, int sh, int sw, int nx,
                                              int ny, float* __restrict__ uf, int* __restrict__ qi, int b) {
  if (b >= B) return;
  uint32_t r0[4], r1[4];
  philox4x32_10(seed, offset + 2 * (uint64_t)b, stream, r0);
  philox4x32_10(seed, offset + 2 * (uint64_t)b + 1, stream, r1);
  const float s24 = 1.f / 16777216.f;
  for (int j = 0; j < 3; ++j) uf[j * B + b] = -1.f + 2.f * (float)(r0[j] >> 8) * s24;
  qi[0 * B + b] = -sh + (int)(r1[0] % (uint32_t)(2 * sh + 1));
  qi[1 * B + b] = -sw + (int)(r1[1] % (uint32_t)(2 * sw + 1));
  qi[2 * B + b] = (int)(r1[2] % (uint32_t)nx);
  qi[3 * B + b] = (int)(r1[3] % (uint32_t)ny);
}
__global__ void aug_draw_dev_kernel(uint64_t seed, uint64_t stream, const unsigned long long* __restrict__ offp, int B,
                                    int sh, int sw, int nx, int ny, float* __restrict__ uf, int* __restrict__ qi) {
  aug_draw_body(seed, stream, *offp, B, sh, sw, nx, ny, uf, qi, blockIdx.x * blockDim.x + threadIdx.x);
}

// One launch for what a training step needs before its first real kernel: the zero-fill of the accumulator arena and the
// gradient buffers (dg_zero_multi) and every parameter draw of the step - latents (with their bfloat16 copy), Gumbel
// logistic noise, DiffAugment parameters - as extra blocks.  Four dependent launches of 4-8 us each otherwise.
struct PrologueZero { float* p[4]; long first[5]; int k; int blocks; };
struct PrologueDraws { DgDraw d[6]; int first_block[7]; int n; };
// fetch_reals as more blocks of the same launch (round 6; trainers/dcgan_amp.py:154-160, utils/lidar.py:31-36): block j of the
// job owns pixels [j chunk, (j + 1) chunk) of the batch - chunk = HW / DG_XSUM_PARTS, so a sample is DG_XSUM_PARTS blocks - and
// STORES its partial sum to parts[j]: the per-sample sums DiffAugment's contrast needs leave as DG_XSUM_PARTS partials per
// sample, summed by the reader in a fixed order.  No accumulator that this very launch would have to zero first, no atomics.
struct PrologueFetch { DgFetch f; int first_block; int blocks; long chunk; };
__device__ __forceinline__ float prologue_fetch_px(float pol, float m, float min_d, float max_d, float drop_const) {
  const float depth = pol * (max_d - min_d) + min_d;   // (pointwise.hip fetch_real_px: the same expressions)
  const float disp = 1.f / depth;
  float inv = (disp - 1.f / max_d) / (1.f / min_d - 1.f / max_d);
  inv = inv * 2.f - 1.f;
  return m * inv + (1.f - m) * drop_const;
}
__global__ __launch_bounds__(256) void step_prologue_kernel(PrologueZero z, PrologueDraws dr, PrologueFetch fe) {
  if (fe.blocks > 0 && (int)blockIdx.x >= fe.first_block) {
    __shared__ float red[16];
    const DgFetch& f = fe.f;
    const int j = (int)blockIdx.x - fe.first_block;
    const float* pol = f.pol;
    const float* mask = f.mask;
    if (f.pool_ctr) {
      const long off = (long)(*f.pool_ctr % (unsigned long long)f.npool) * ((long)f.B * f.HW);
      pol += off;
      mask += off;
    }
    const long i0 = (long)j * fe.chunk;
    float acc = 0.f;
    constexpr int U = 4;                                   // four trips' loads in flight per lane (8 x 16 bytes)
    for (long k0 = (long)threadIdx.x * 4; k0 < fe.chunk; k0 += U * 1024) {
      float4 p4[U], m4[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long k = k0 + u * 1024;
        if (k < fe.chunk) { p4[u] = *(const float4*)(pol + i0 + k); m4[u] = *(const float4*)(mask + i0 + k); }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long k = k0 + u * 1024;
        if (k >= fe.chunk) break;
        float4 o;
        o.x = prologue_fetch_px(p4[u].x, m4[u].x, f.min_depth, f.max_depth, f.drop_const);
        o.y = prologue_fetch_px(p4[u].y, m4[u].y, f.min_depth, f.max_depth, f.drop_const);
        o.z = prologue_fetch_px(p4[u].z, m4[u].z, f.min_depth, f.max_depth, f.drop_const);
        o.w = prologue_fetch_px(p4[u].w, m4[u].w, f.min_depth, f.max_depth, f.drop_const);
        *(float4*)(f.out + i0 + k) = o;
        acc += (o.x + o.y) + (o.z + o.w);
      }
    }
    const float sblk = dg_block_sum(acc, red);
    if (threadIdx.x == 0) f.parts[j] = sblk;
    return;
  }
  if ((int)blockIdx.x < z.blocks) {
    const long stride = (long)z.blocks * 256, total = z.first[z.k];
    for (long i4 = (long)blockIdx.x * 256 + threadIdx.x; 4 * i4 < total; i4 += stride) {
      const long i = 4 * i4;
      int j = 0;
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (q < z.k && i >= z.first[q]) j = q;
      *(float4*)(z.p[j] + (i - z.first[j])) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return;
  }
  const int blk = (int)blockIdx.x - z.blocks;
  int j = 0;
  for (int q = 1; q < dr.n; ++q)
    if (blk >= dr.first_block[q]) j = q;
  const DgDraw& d = dr.d[j];
  const long i = (long)(blk - dr.first_block[j]) * 256 + threadIdx.x;
  const uint64_t offset = *d.offset_dev + d.base;
  if (d.kind == 0) {
    philox_fill_body(d.seed, d.stream_id, offset, d.fill_kind, d.lo, d.hi, d.ilo, d.ihi, d.n, d.out, (bf16*)d.out_bf16, i);
  } else if (d.kind == 1) {
    philox_logistic_body(d.seed, d.stream_id, offset, d.eps, d.n, (float*)d.out, i);
  } else {
    const int sh = (int)(d.H * (1.0 / 8.0) / 2 + 0.5), sw = (int)(d.W * (1.0 / 8.0) / 2 + 0.5);
    const int ch = (int)(d.H * 0.5 + 0.5), cw = (int)(d.W * 0.5 + 0.5);
    aug_draw_body(d.seed, d.stream_id, offset, d.B, sh, sw, d.H + (1 - ch % 2), d.W + (1 - cw % 2), d.uf, d.qi, (int)i);
  }
}

static inline unsigned nblk(long n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }

extern "C" {

int dg_adam_ema_step(float* p, const float* grad, float* m, float* v, float* ema, void* shadow, int shadow_dtype,
                     long n, float gscale, float lr, float beta1, float beta2, float eps, int step, float ema_decay,
                     void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (n <= 0) return DG_OK;
  if (n % 4 != 0) return DG_EINVAL;  // flat stores are padded to 64 elements
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float lr_bc1 = (float)((double)lr / bc1);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const long n4 = n / 4;
  unsigned grid = nblk(n4);
  if (grid > 256 * 16) grid = 256 * 16;
  if (shadow && shadow_dtype == DG_BF16)
    adam_ema_kernel<bf16><<<grid, 256, 0, s>>>(p, grad, m, v, ema, (bf16*)shadow, n4, gscale, lr_bc1, inv_sqrt_bc2, beta1, beta2, eps, ema_decay, nullptr, lr);
  else
    adam_ema_kernel<float><<<grid, 256, 0, s>>>(p, grad, m, v, ema, (float*)shadow, n4, gscale, lr_bc1, inv_sqrt_bc2, beta1, beta2, eps, ema_decay, nullptr, lr);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// up to 16 fp32 buffers -> DG_BF16X2 twins in one launch (the split-bf16 copies of the fat layers' weight shadows)
int dg_cast_x2_multi(const float* const* src, void* const* dst, const long* n, int count, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!src || !dst || !n || count < 1 || count > 16) return DG_EINVAL;
  CastItems it{};
  long tot = 0;
  for (int i = 0; i < count; ++i) {
    if (!src[i] || !dst[i] || n[i] <= 0 || n[i] % 64 != 0 || ((size_t)src[i] & 15) || ((size_t)dst[i] & 255)) return DG_EINVAL;
    it.src[i] = src[i]; it.dst[i] = dst[i]; it.first8[i] = tot;
    tot += n[i] / 8;
  }
  it.first8[count] = tot;
  it.n = count;
  cast_x2_kernel<<<nblk(tot), 256, 0, s>>>(it);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_uncast(const void* src, int dtype, float* dst, long n, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!src || !dst || n <= 0) return DG_EINVAL;
  if (dtype == DG_BF16X2) {
    if (n % 64 != 0 || ((size_t)src & 255) || ((size_t)dst & 15)) return DG_EINVAL;
    uncast_x2_kernel<<<nblk(n / 8), 256, 0, s>>>((const unsigned short*)src, dst, n / 8);
  } else if (dtype == DG_BF16) uncast_kernel<bf16><<<nblk(n), 256, 0, s>>>((const bf16*)src, dst, n);
  else uncast_kernel<float><<<nblk(n), 256, 0, s>>>((const float*)src, dst, n);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_cast(const float* src, void* dst, int dtype, long n, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (dtype == DG_BF16X2) return dg_cast_x2_multi(&src, &dst, &n, 1, s_);
  if (dtype == DG_BF16) cast_kernel<bf16><<<nblk(n), 256, 0, s>>>(src, (bf16*)dst, n);
  else cast_kernel<float><<<nblk(n), 256, 0, s>>>(src, (float*)dst, n);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_transpose_shadow(const float* master, void* dst, int dtype, int Ci, int Co, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  dim3 grid((Co + 31) / 32, (Ci + 31) / 32, 16);
  if (dtype == DG_BF16) transpose_shadow_kernel<bf16><<<grid, 256, 0, s>>>(master, (bf16*)dst, Ci, Co);
  else transpose_shadow_kernel<float><<<grid, 256, 0, s>>>(master, (float*)dst, Ci, Co);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_philox_bits(uint64_t seed, uint64_t stream, uint64_t offset, long n4, uint32_t* out, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  philox_bits_kernel<<<nblk(n4), 256, 0, s>>>(seed, stream, offset, n4, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_philox_fill(uint64_t seed, uint64_t stream, uint64_t offset, int kind, float lo, float hi, int ilo, int ihi,
                   long n, void* out, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (kind < 0 || kind > 3) return DG_EINVAL;
  if (kind == 3 && ihi <= ilo) return DG_EINVAL;
  philox_fill_kernel<<<nblk((n + 3) / 4), 256, 0, s>>>(seed, stream, offset, kind, lo, hi, ilo, ihi, n, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_aug_draw(uint64_t seed, uint64_t stream, uint64_t offset, int B, int H, int W, float* uf, int* qi, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const int sh = (int)(H * (1.0 / 8.0) / 2 + 0.5), sw = (int)(W * (1.0 / 8.0) / 2 + 0.5);  // diff_augment.py:58
  const int ch = (int)(H * 0.5 + 0.5), cw = (int)(W * 0.5 + 0.5);                          // diff_augment.py:85
  aug_draw_kernel<<<nblk(B), 256, 0, s>>>(seed, stream, offset, B, sh, sw, H + (1 - ch % 2), W + (1 - cw % 2), uf, qi);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_counter_add(unsigned long long* counter, unsigned long long delta, void* s_) {
  counter_add_kernel<<<1, 64, 0, (hipStream_t)s_>>>(counter, delta);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// k <= 8 DISTINCT counters advanced by one launch (a step's Philox offsets and Adam step counts, queued by the caller
// behind their consumers: one graph node instead of one per counter)
static int fill_counter_adds(CounterAdds& a, unsigned long long* const* counters, const unsigned long long* deltas, int k,
                             int snap_idx, const float* src, int n, float* dst_ring, int ring) {
  if (k < 1 || k > 8 || !counters || !deltas) return DG_EINVAL;
  if (snap_idx >= 0 && (snap_idx >= k || !src || !dst_ring || n < 1 || n > 64 || ring < 1)) return DG_EINVAL;
  a = CounterAdds{};
  a.snap_idx = snap_idx; a.snap_n = n; a.snap_ring = ring; a.snap_src = src; a.snap_dst = dst_ring;
  for (int i = 0; i < k; ++i) {
    if (!counters[i]) return DG_EINVAL;
    for (int j = 0; j < i; ++j)
      if (counters[j] == counters[i]) return DG_EINVAL;
    a.c[i] = counters[i]; a.d[i] = deltas[i];
  }
  a.k = k;
  return DG_OK;
}
static int counter_add_multi(unsigned long long* const* counters, const unsigned long long* deltas, int k, int snap_idx,
                             const float* src, int n, float* dst_ring, int ring, void* s_) {
  CounterAdds a{};
  const int rc = fill_counter_adds(a, counters, deltas, k, snap_idx, src, n, dst_ring, ring);
  if (rc != DG_OK) return rc;
  counter_add_multi_kernel<<<1, 64, 0, (hipStream_t)s_>>>(a);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
int dg_counter_add_multi(unsigned long long* const* counters, const unsigned long long* deltas, int k, void* s_) {
  return counter_add_multi(counters, deltas, k, -1, nullptr, 0, nullptr, 1, s_);
}
// ... and slot (old value of counters[snap_idx]) % ring of `dst_ring` (n floats per slot; device memory or mapped pinned host
// memory) receives src[0..n): the step's logged scalars leave the device from the step's last launch - no copy node behind
// the graph, and the host reads slot i once an event recorded behind step i has completed, never blocking the launch stream
int dg_counter_add_multi_snap(unsigned long long* const* counters, const unsigned long long* deltas, int k, int snap_idx,
                              const float* src, int n, float* dst_ring, int ring, void* s_) {
  if (snap_idx < 0) return DG_EINVAL;
  return counter_add_multi(counters, deltas, k, snap_idx, src, n, dst_ring, ring, s_);
}

int dg_philox_fill_dev(uint64_t seed, uint64_t stream, const unsigned long long* offset_dev, int kind, float lo,
                       float hi, int ilo, int ihi, long n, void* out, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (kind < 0 || kind > 3) return DG_EINVAL;
  if (kind == 3 && ihi <= ilo) return DG_EINVAL;
  philox_fill_dev_kernel<<<nblk((n + 3) / 4), 256, 0, s>>>(seed, stream, offset_dev, kind, lo, hi, ilo, ihi, n, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// logistic noise of n elements from the device-resident Philox offset; the caller advances the counter by 2 ((n + 3) / 4)
int dg_philox_logistic_dev(uint64_t seed, uint64_t stream, const unsigned long long* offset_dev, float eps, long n,
                           float* out, void* s_) {
  if (!offset_dev || !out || n <= 0) return DG_EINVAL;
  philox_logistic_dev_kernel<<<nblk((n + 3) / 4), 256, 0, (hipStream_t)s_>>>(seed, stream, offset_dev, eps, n, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// zero-fill of k <= 4 fp32 buffers (as dg_zero_multi; k may be 0) + ndraw <= 6 draws (DgDraw) + optionally fetch_reals of one
// batch (DgFetch) in one launch
static int step_prologue_impl(float* const* ptrs, const long* counts, int k, const DgDraw* draws, int ndraw, const DgFetch* fetch,
                              void* s_) {
  if (k < 0 || k > 4 || ndraw < 0 || ndraw > 6 || (k && (!ptrs || !counts)) || (ndraw && !draws)) return DG_EINVAL;
  PrologueZero z{};
  long tot = 0;
  for (int i = 0; i < k; ++i) {
    if (!ptrs[i] || counts[i] < 0 || counts[i] % 4 != 0 || ((size_t)ptrs[i] & 15) != 0) return DG_EINVAL;
    z.p[i] = ptrs[i]; z.first[i] = tot;
    tot += counts[i];
  }
  z.first[k] = tot;
  z.k = k;
  unsigned zb = nblk(tot / 4);
  if (zb > 2048) zb = 2048;
  z.blocks = (int)zb;
  PrologueDraws dr{};
  long blocks = 0;
  for (int i = 0; i < ndraw; ++i) {
    const DgDraw& d = draws[i];
    if (!d.offset_dev) return DG_EINVAL;
    long threads;
    if (d.kind == 0) {
      if (!d.out || d.n <= 0 || d.fill_kind < 0 || d.fill_kind > 3 || (d.fill_kind == 3 && (d.ihi <= d.ilo || d.out_bf16)))
        return DG_EINVAL;
      threads = (d.n + 3) / 4;
    } else if (d.kind == 1) {
      if (!d.out || d.n <= 0) return DG_EINVAL;
      threads = (d.n + 3) / 4;
    } else if (d.kind == 2) {
      if (!d.uf || !d.qi || d.B <= 0 || d.H <= 0 || d.W <= 0) return DG_EINVAL;
      threads = d.B;
    } else {
      return DG_EINVAL;
    }
    dr.d[i] = d;
    dr.first_block[i] = (int)blocks;
    blocks += (threads + 255) / 256;
    if (blocks > (1L << 30)) return DG_EUNSUPPORTED;
  }
  dr.first_block[ndraw] = (int)blocks;
  dr.n = ndraw;
  PrologueFetch fe{};
  if (fetch) {
    const DgFetch& f = *fetch;
    if (!f.pol || !f.mask || !f.out || !f.parts || f.B <= 0 || f.HW <= 0 || (f.pool_ctr && f.npool <= 0)) return DG_EINVAL;
    // 16-byte accesses, whole 1024-pixel sweeps per block, DG_XSUM_PARTS blocks per sample
    if (f.HW % (1024L * DG_XSUM_PARTS) != 0 || (((size_t)f.pol | (size_t)f.mask | (size_t)f.out) & 15) != 0) return DG_EUNSUPPORTED;
    fe.f = f;
    fe.chunk = f.HW / DG_XSUM_PARTS;
    fe.blocks = f.B * DG_XSUM_PARTS;
    fe.first_block = (int)(zb + blocks);
  }
  if (zb + blocks + fe.blocks == 0) return DG_OK;
  step_prologue_kernel<<<(unsigned)(zb + blocks + fe.blocks), 256, 0, (hipStream_t)s_>>>(z, dr, fe);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
int dg_step_prologue(float* const* ptrs, const long* counts, int k, const DgDraw* draws, int ndraw, void* s_) {
  return step_prologue_impl(ptrs, counts, k, draws, ndraw, nullptr, s_);
}
int dg_step_prologue_fetch(float* const* ptrs, const long* counts, int k, const DgDraw* draws, int ndraw, const DgFetch* fetch,
                           void* s_) {
  if (!fetch) return DG_EINVAL;
  return step_prologue_impl(ptrs, counts, k, draws, ndraw, fetch, s_);
}

int dg_aug_draw_dev(uint64_t seed, uint64_t stream, const unsigned long long* offset_dev, int B, int H, int W,
                    float* uf, int* qi, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const int sh = (int)(H * (1.0 / 8.0) / 2 + 0.5), sw = (int)(W * (1.0 / 8.0) / 2 + 0.5);
  const int ch = (int)(H * 0.5 + 0.5), cw = (int)(W * 0.5 + 0.5);
  aug_draw_dev_kernel<<<nblk(B), 256, 0, s>>>(seed, stream, offset_dev, B, sh, sw, H + (1 - ch % 2), W + (1 - cw % 2), uf, qi);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_adam_ema_step_dev(float* p, const float* grad, float* m, float* v, float* ema, void* shadow, int shadow_dtype,
                         long n, float gscale, float lr, float beta1, float beta2, float eps,
                         const unsigned long long* step_dev, float ema_decay, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (n <= 0) return DG_OK;
  if (n % 4 != 0 || !step_dev) return DG_EINVAL;
  const long n4 = n / 4;
  unsigned grid = nblk(n4);
  if (grid > 256 * 16) grid = 256 * 16;
  if (shadow && shadow_dtype == DG_BF16)
    adam_ema_kernel<bf16><<<grid, 256, 0, s>>>(p, grad, m, v, ema, (bf16*)shadow, n4, gscale, 0.f, 0.f, beta1, beta2, eps, ema_decay, step_dev, lr);
  else
    adam_ema_kernel<float><<<grid, 256, 0, s>>>(p, grad, m, v, ema, (float*)shadow, n4, gscale, 0.f, 0.f, beta1, beta2, eps, ema_decay, step_dev, lr);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

/* The optimizer of a network as one launch that sums the pieces' partial gradient rows itself (adam_fused_kernel above) */
int dg_adam_fused(float* p, float* grad, float* v, float* ema, void* shadow, int shadow_dtype, const DgOptSeg* segs, int nseg,
                  float gscale, float lr, float beta2, float eps, const unsigned long long* step_dev, float ema_decay, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!p || !grad || !v || !segs || !step_dev || nseg < 1 || nseg > DG_OPT_MAX_SEG) return DG_EINVAL;
  if (shadow && shadow_dtype != DG_BF16 && shadow_dtype != DG_F32) return DG_EINVAL;
  if ((((size_t)p | (size_t)grad | (size_t)v | (size_t)ema | (size_t)shadow) & 15) != 0) return DG_EINVAL;
  OptSegs g{};
  long blocks = 0;
  for (int i = 0; i < nseg; ++i) {
    DgOptSeg q = segs[i];
    if (q.off < 0 || q.numel <= 0 || q.off % 4 != 0 || q.numel % 4 != 0 || q.splits < 0 || (q.splits > 0) != (q.part != nullptr))
      return DG_EINVAL;
    if (q.part && ((size_t)q.part & 15) != 0) return DG_EINVAL;
    if (q.kind == 1) {
      if (q.ci <= 0 || q.co <= 0 || !(q.co == 64 || q.co % 128 == 0) || q.ci % 16 != 0 || q.numel != 16LL * q.ci * q.co ||
          !q.shadow_t || !shadow || ((size_t)q.shadow_t & 15) != 0 || q.ws_src)
        return DG_EINVAL;
      q.first_block = (int)blocks;
      blocks += q.numel / 1024;
    } else if (q.kind == 0) {
      if (q.ws_src && (q.ws_n < 1 || q.ws_stride < q.numel || ((size_t)q.ws_src & 15) != 0 || q.ws_stride % 4 != 0)) return DG_EINVAL;
      if (q.splits > 64) {
        if (q.ws_src) return DG_EINVAL;
        q.kind = 2;                               // wide: 16 float4 per workgroup, 16 groups of partial rows
        q.first_block = (int)blocks;
        blocks += (q.numel / 4 + 15) / 16;
      } else {
        q.first_block = (int)blocks;
        blocks += (q.numel / 4 + 255) / 256;
      }
    } else {
      return DG_EINVAL;
    }
    if (blocks > (1L << 30)) return DG_EUNSUPPORTED;
    g.s[i] = q;
  }
  g.n = nseg;
  g.total_blocks = (int)blocks;
  if (shadow && shadow_dtype == DG_BF16)
    adam_fused_kernel<bf16><<<(unsigned)blocks, 256, 0, s>>>(p, grad, v, ema, (bf16*)shadow, g, gscale, lr, beta2, eps, ema_decay, step_dev);
  else
    adam_fused_kernel<float><<<(unsigned)blocks, 256, 0, s>>>(p, grad, v, ema, (float*)shadow, g, gscale, lr, beta2, eps, ema_decay, step_dev);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

/* Adam(beta1 = 0) + EMA of Proj.weight [Np][K] with its gradient wscale * dp0^T z formed in the kernel (bf16
 * operands, nb even and <= 64 rows).  DG_EUNSUPPORTED for other shapes: the caller then runs dg_wgrad + the plain Adam. */
int dg_adam_proj_fused(float* p, float* v, float* ema, void* shadow, int shadow_dtype, const void* dp0, const void* zT,
                       int op_dtype, int nb, long Np, int K, float wscale, float gscale, float lr, float beta2,
                       float eps, const unsigned long long* step_dev, float ema_decay, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!p || !v || !dp0 || !zT || !step_dev) return DG_EINVAL;
  // op_dtype: the operands' element type, DG_BF16 or (round 6) DG_F32, the latter optionally | DG_FORCE_FP32X3 - the fp32 rows
  // split into bf16 pairs in registers (the fp32x3 mode's arithmetic); fp32 operands run on the matrix-core path only
  const int x3 = (op_dtype & DG_FORCE_FP32X3) ? 1 : 0;
  op_dtype &= ~DG_FORCE_FP32X3;
  if ((op_dtype != DG_BF16 && op_dtype != DG_F32) || (x3 && op_dtype != DG_F32) || Np <= 0 || nb < 2) return DG_EUNSUPPORTED;
  const size_t lds = ((size_t)(nb / 2) * K + (size_t)APF_ROWS * (nb / 2)) * 4;
  const bool valu_ok = nb <= 64 && nb % 2 == 0 && K % 4 == 0 && K / 4 <= 256 && 256 % (K / 4) == 0 && lds <= 64 * 1024;
  // The MFMA gradient GEMM with the optimizer as its epilogue takes every batch size when Np and K are multiples of 128
  // and measured 6 % faster than the LDS-resident VALU kernel even at nb = 32 (scripts/bench_proj_adam.py: 311 vs
  // 331 us); the VALU kernel takes the shapes the GEMM does not tile.
  const bool mfma_ok = Np % 128 == 0 && K % 128 == 0;
  if (op_dtype == DG_F32 && !mfma_ok) return DG_EUNSUPPORTED;
  if (!valu_ok || mfma_ok) {
    // (wgrad_mfma.hip; also the only fused path for nb > 64, i.e. the all-gathered global batch of a multi-GPU run);
    // shapes neither kernel takes -> DG_EUNSUPPORTED -> caller's unfused path
    WgradP w{};
    w.wmode = 2; w.ring = 1; w.B = 1; w.Hc = 1; w.Wc = nb; w.Ci = (int)Np; w.Co = K;
    w.a = dp0; w.a_sb = 0; w.a_sp = Np; w.a_sc = 1;
    w.g = zT; w.g_sb = 0; w.g_sp = K; w.g_sc = 1;
    w.dw = nullptr; w.scale = wscale; w.rowscale = nullptr; w.a_dtype = op_dtype; w.g_dtype = op_dtype;
    AdamEpi ad{p, v, ema, shadow, shadow && shadow_dtype == DG_BF16 ? 1 : 0, gscale, lr, beta2, eps, ema_decay, step_dev};
    if (Np > 0x7fffffffL) return DG_EUNSUPPORTED;
    return dg_wgrad_mfma_adam_launch(&w, &ad, s, x3);
  }
  const unsigned grid = (unsigned)((Np + APF_ROWS - 1) / APF_ROWS);
  if (shadow && shadow_dtype == DG_BF16)
    adam_proj_fused_kernel<bf16><<<grid, 256, lds, s>>>(p, v, ema, (bf16*)shadow, (const bf16*)dp0, (const bf16*)zT, nb, Np, K, wscale, gscale, lr, beta2, eps, ema_decay, step_dev);
  else
    adam_proj_fused_kernel<float><<<grid, 256, lds, s>>>(p, v, ema, (float*)shadow, (const bf16*)dp0, (const bf16*)zT, nb, Np, K, wscale, gscale, lr, beta2, eps, ema_decay, step_dev);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_transpose_shadow_multi(const float* master, const long long* desc_dev, int nseg, int total_tiles, int dtype,
                              void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!master || !desc_dev || nseg <= 0 || total_tiles <= 0) return DG_EINVAL;
  UpFrags uf{};
  if (dtype == DG_BF16) transpose_shadow_multi_kernel<bf16><<<total_tiles, 256, 0, s>>>(master, desc_dev, nseg, uf, CounterAdds{}, -1);
  else transpose_shadow_multi_kernel<float><<<total_tiles, 256, 0, s>>>(master, desc_dev, nseg, uf, CounterAdds{}, -1);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

static int transpose_multi(const float* master, const long long* desc_dev, int nseg, int total_tiles, int dtype,
                           const DgUpFrag* frags, int nfrag, const CounterAdds* ca, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  // (nseg == 0, total_tiles == 0: no transposes - the launch only builds the fragments and / or advances the counters, round 6:
  //  dg_adam_fused writes the fat layers' transposed shadows itself)
  if (!master || nseg < 0 || total_tiles < 0 || (nseg > 0) != (total_tiles > 0) || (nseg && !desc_dev) || nfrag < 0 || nfrag > 4 ||
      (nfrag && !frags))
    return DG_EINVAL;
  if (total_tiles + nfrag == 0 && !ca) return DG_OK;
  if (nfrag && dtype != DG_BF16) return DG_EUNSUPPORTED;   // (the kernel that reads them is bf16 only)
  UpFrags uf{};
  uf.n = nfrag; uf.first_block = total_tiles;
  for (int i = 0; i < nfrag; ++i) {
    if (!frags[i].frag || ((size_t)frags[i].frag & 15) || frags[i].N < 1 || frags[i].N > 4 || frags[i].Hc < 1) return DG_EINVAL;
    uf.f[i] = frags[i];
  }
  const int work = total_tiles + nfrag * 3 * UP_FRAG_BLOCKS;
  const unsigned grid = (unsigned)(work + (ca ? 1 : 0));
  const CounterAdds cav = ca ? *ca : CounterAdds{};
  if (dtype == DG_BF16) transpose_shadow_multi_kernel<bf16><<<grid, 256, 0, s>>>(master, desc_dev, nseg, uf, cav, ca ? work : -1);
  else transpose_shadow_multi_kernel<float><<<grid, 256, 0, s>>>(master, desc_dev, nseg, uf, cav, ca ? work : -1);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
int dg_transpose_shadow_multi_frags(const float* master, const long long* desc_dev, int nseg, int total_tiles, int dtype,
                                    const DgUpFrag* frags, int nfrag, void* s_) {
  return transpose_multi(master, desc_dev, nseg, total_tiles, dtype, frags, nfrag, nullptr, s_);
}
// ... and one more block does what dg_counter_add_multi[_snap] does (snap_idx < 0: no snapshot): the LAST launch of a training
// step - the shadow refresh behind the generator's optimizer - also advances the step's counters and files its scalars,
// instead of a launch of five threads behind it
int dg_transpose_shadow_multi_tail(const float* master, const long long* desc_dev, int nseg, int total_tiles, int dtype,
                                   const DgUpFrag* frags, int nfrag, unsigned long long* const* counters,
                                   const unsigned long long* deltas, int k, int snap_idx, const float* src, int n,
                                   float* dst_ring, int ring, void* s_) {
  CounterAdds a{};
  const int rc = fill_counter_adds(a, counters, deltas, k, snap_idx, src, n, dst_ring, ring);
  if (rc != DG_OK) return rc;
  return transpose_multi(master, desc_dev, nseg, total_tiles, dtype, frags, nfrag, &a, s_);
}

}  // extern "C"
