// Direct (VALU) conv-like kernels.
//
// Two jobs:
//  1. the production kernels of the THIN layers, which are HBM-bound and have nothing for MFMA to chew on:
//     Head (Cout<=3, models/gans/dcgan_eqlr.py:29-46), Down1 (Cin=2 after BlurVH, :90), their backward-data and
//     weight-gradient passes;
//  2. the general-shape path (any channel count, ring=False) used for tiny nets such as the golden-vector cases,
//     and the on-GPU cross-check of the MFMA kernels.
#include "common.h"

// One thread computes NV consecutive output channels of one output pixel.
template <int NV>
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvP p) {
  // bias-gradient sums: two-word fixed point (common.h dg_fix2) in LDS and, through DgConv.dbias_ws, across the workgroups -
  // integer adds, the same bits whatever the order (round 6: float atomics at both levels summed in arrival order)
  __shared__ unsigned long long s_db[512][2];
  const bool want_db = p.dbias != nullptr;
  if (want_db) {
    for (int i = threadIdx.x; i < p.bias_mod; i += blockDim.x) s_db[i][0] = s_db[i][1] = 0ull;
    __syncthreads();
  }
  const int NG = (p.N + NV - 1) / NV;
  int Ho, Wo, Ws;
  if (p.mode == MODE_S2) { Ho = p.Hc; Wo = p.Wc; Ws = 2 * p.Wc; }
  else if (p.mode == MODE_UP) { Ho = 2 * p.Hc; Wo = 2 * p.Wc; Ws = p.Wc; }
  else { Ho = 1; Wo = 1; Ws = 1; }
  const long total = (long)p.B * Ho * Wo * NG;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < total) {
    const int ng = (int)(idx % NG);
    const long pix = idx / NG;
    const int X = (int)(pix % Wo);
    const int Y = (int)((pix / Wo) % Ho);
    const int b = (int)(pix / ((long)Wo * Ho));
    const int n0 = ng * NV;
    float acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
    const long in_b = (long)b * p.in_sb;
    if (p.mode == MODE_GEMM) {
      for (int k = 0; k < p.K; ++k) {
        const float a = dg_ld(p.in, in_b + (long)k * p.in_sk, p.in_dtype);
#pragma unroll
        for (int v = 0; v < NV; ++v)
          if (n0 + v < p.N) acc[v] += a * dg_ld(p.w, (long)k * p.w_sk + (long)(n0 + v) * p.w_sn, p.w_dtype);
      }
    } else {
      for (int i = 0; i < 6; ++i) {
        int r, ky;
        if (!dg_tap1d(p.mode, p.adj, 0, Y, p.Hc, i, r, ky)) continue;
        for (int j = 0; j < 6; ++j) {
          int c, kx;
          if (!dg_tap1d(p.mode, p.adj, p.ring, X, p.Wc, j, c, kx)) continue;
          const long ib = in_b + ((long)r * Ws + c) * p.in_sp;
          const long wb = (long)(ky * 4 + kx) * p.w_st;
          for (int k = 0; k < p.K; ++k) {
            const float a = dg_ld(p.in, ib + (long)k * p.in_sk, p.in_dtype);
#pragma unroll
            for (int v = 0; v < NV; ++v)
              if (n0 + v < p.N) acc[v] += a * dg_ld(p.w, wb + (long)k * p.w_sk + (long)(n0 + v) * p.w_sn, p.w_dtype);
          }
        }
      }
    }
    const long ob = (long)b * p.out_sb + ((long)Y * Wo + X) * p.out_sp;
    const float rs = p.rowscale ? p.rowscale[b] : 1.f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int n = n0 + v;
      if (n < p.N) {
        const long o = ob + (long)n * p.out_sn;
        const float bias = p.bias ? p.bias[n % p.bias_mod] : 0.f;
        const float auxv = p.epi == EPI_MASK ? dg_ld(p.aux, o, p.out_dtype) : 0.f;
        const float r = dg_epilogue(acc[v], p.nscale ? p.scale * p.nscale[n] : p.scale, p.epi, bias, auxv);
        dg_st(p.out, o, p.out_dtype, r);
        if (want_db) {
          long long hi, lo;
          if (dg_fix2(r * rs, hi, lo)) {
            if (hi) atomicAdd(&s_db[n % p.bias_mod][0], (unsigned long long)hi);
            if (lo) atomicAdd(&s_db[n % p.bias_mod][1], (unsigned long long)lo);
          } else atomicAdd(&p.dbias[n % p.bias_mod], r * rs);
        }
      }
    }
  }
  if (want_db) {
    __syncthreads();
    const bool ws = dg_dbias_ws_ok(p.dbias_ws, p.bias_mod);
    for (int i = threadIdx.x; i < p.bias_mod; i += blockDim.x) {
      const long long hi = (long long)s_db[i][0], lo = (long long)s_db[i][1];
      if (ws) dg_dbias_ws_add(p.dbias_ws, i, hi, lo);
      else if (hi | lo) atomicAdd(&p.dbias[i], dg_fix2_value(hi, lo));
    }
    if (ws) dg_dbias_ws_finish(p.dbias_ws, p.bias_mod, p.dbias);
  }
}

// Weight gradient, direct form.  One thread owns dw[tap][ci][co] for one pixel slab (gridDim.y slabs over the
// B*Hc coarse rows) and adds its partial sum with one fp32 atomic.
__global__ __launch_bounds__(256) void wgrad_direct_kernel(WgradP p) {
  const int ntap = p.wmode == 2 ? 1 : 16;
  const long total = (long)ntap * p.Ci * p.Co;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int co = (int)(idx % p.Co);
  const int ci = (int)((idx / p.Co) % p.Ci);
  const int tap = (int)(idx / ((long)p.Co * p.Ci));
  const int ky = tap >> 2, kx = tap & 3;
  const long rows = (long)p.B * p.Hc;
  const long per = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * per;
  const long r1 = r0 + per < rows ? r0 + per : rows;
  const int Wa = p.wmode == 0 ? 2 * p.Wc : p.Wc;
  const int Wg = p.wmode == 1 ? 2 * p.Wc : p.Wc;
  float tot = 0.f;
  for (long row = r0; row < r1; ++row) {
    const int b = (int)(row / p.Hc), m = (int)(row % p.Hc);
    float acc = 0.f;
    if (p.wmode == 2) {  // plain rows: "pixels" are the Wc rows of both operands, no taps
      const long ab = (long)b * p.a_sb + (long)ci * p.a_sc;
      const long gb = (long)b * p.g_sb + (long)co * p.g_sc;
      for (int x = 0; x < p.Wc; ++x)
        acc += dg_ld(p.a, ab + (long)x * p.a_sp, p.a_dtype) * dg_ld(p.g, gb + (long)x * p.g_sp, p.g_dtype);
    } else {
      int ra, rg;
      dg_wgrad1d(p.wmode, 0, m, p.Hc, ky, ra, rg);
      const long ab = (long)b * p.a_sb + (long)ci * p.a_sc;
      const long gb = (long)b * p.g_sb + (long)co * p.g_sc;
      for (int x = 0; x < p.Wc; ++x) {
        int ca, cg;
        dg_wgrad1d(p.wmode, p.ring, x, p.Wc, kx, ca, cg);
        acc += dg_ld(p.a, ab + ((long)ra * Wa + ca) * p.a_sp, p.a_dtype) *
               dg_ld(p.g, gb + ((long)rg * Wg + cg) * p.g_sp, p.g_dtype);
      }
    }
    tot += (p.rowscale ? p.rowscale[b] : 1.f) * acc;
  }
  // workspace form (DgWgrad.ws): slab y stores its partial at ws[y][idx], dg_wgrad_reduce sums the slabs in index order
  if (p.ws) p.ws[(long)blockIdx.y * total + idx] = tot * p.scale;
  else atomicAdd(&p.dw[idx], tot * p.scale);
}

int dg_conv_direct_launch(const ConvP* p, hipStream_t stream) {
  int Ho, Wo;
  if (p->mode == MODE_S2) { Ho = p->Hc; Wo = p->Wc; }
  else if (p->mode == MODE_UP) { Ho = 2 * p->Hc; Wo = 2 * p->Wc; }
  else { Ho = 1; Wo = 1; }
  if (p->dbias && p->bias_mod > 512) return DG_EINVAL;
  int nv = p->N >= 8 ? 8 : (p->N >= 4 ? 4 : p->N);
  const int NG = (p->N + nv - 1) / nv;
  const long total = (long)p->B * Ho * Wo * NG;
  const unsigned grid = (unsigned)((total + 255) / 256);
  if (grid == 0) return DG_OK;
  switch (nv) {
    case 1: conv_direct_kernel<1><<<grid, 256, 0, stream>>>(*p); break;
    case 2: conv_direct_kernel<2><<<grid, 256, 0, stream>>>(*p); break;
    case 3: conv_direct_kernel<3><<<grid, 256, 0, stream>>>(*p); break;
    case 4: conv_direct_kernel<4><<<grid, 256, 0, stream>>>(*p); break;
    default: conv_direct_kernel<8><<<grid, 256, 0, stream>>>(*p); break;
  }
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// enough slabs to fill the chip, never more than the rows there are
static long direct_slabs(const WgradP* p) {
  const long total = (long)(p->wmode == 2 ? 1 : 16) * p->Ci * p->Co;
  const long rows = (long)p->B * p->Hc;
  long slabs = (256L * 16 * 256) / (total > 0 ? total : 1);
  if (slabs < 1) slabs = 1;
  if (slabs > rows) slabs = rows;
  if (slabs > 4096) slabs = 4096;
  return slabs;
}

// > 1: the direct kernel's launch has that many slabs and takes DgWgrad.ws (slabs x numel floats) instead of atomics
int dg_wgrad_direct_ws_splits(const WgradP* p) {
  const long s = direct_slabs(p);
  return s > 1 ? (int)s : 0;
}

int dg_wgrad_direct_launch(const WgradP* p, hipStream_t stream) {
  const int ntap = p->wmode == 2 ? 1 : 16;
  const long total = (long)ntap * p->Ci * p->Co;
  const long slabs = direct_slabs(p);
  if (p->ws && slabs <= 1) return DG_EUNSUPPORTED;
  dim3 grid((unsigned)((total + 255) / 256), (unsigned)slabs);
  wgrad_direct_kernel<<<grid, 256, 0, stream>>>(*p);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}


// ---- DgConv.mask_out behind kernels that do not write it themselves: one pass over the output just written.
// Bit e of the mask = (out element at offset e) > 0, the test the EPI_MASK epilogues make on the saved activation
// (models/ops/common.py:99-106 under autograd).  One thread per 8 consecutive channels of one pixel = one mask byte.
__global__ __launch_bounds__(256) void lrelu_bits_kernel(const void* out, unsigned char* bits, int dtype, long total, int n8,
                                                         long pixels, long out_sb, long out_sp) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int g = (int)(idx % n8);
  const long pix = idx / n8;
  const long o = (pix / pixels) * out_sb + (pix % pixels) * out_sp + (long)g * 8;
  unsigned m = 0;
  if (dtype == DG_BF16) {
    const uint4 raw = *(const uint4*)((const bf16*)out + o);
    const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      m |= (unsigned)((short)(w[e] & 0xffffu) > 0) << (2 * e);
      m |= (unsigned)((int)w[e] > 0xffff) << (2 * e + 1);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) m |= (unsigned)(((const float*)out)[o + e] > 0.f) << e;
  }
  bits[o >> 3] = (unsigned char)m;
}

int dg_lrelu_bits_launch(const ConvP* p, hipStream_t s) {
  long pixels = 1;
  if (p->mode == MODE_S2) pixels = (long)p->Hc * p->Wc;
  else if (p->mode == MODE_UP) pixels = 4L * p->Hc * p->Wc;
  const int n8 = p->N / 8;
  const long total = (long)p->B * pixels * n8;
  const long blocks = (total + 255) / 256;
  lrelu_bits_kernel<<<(unsigned)blocks, 256, 0, s>>>(p->out, (unsigned char*)p->mask_out, p->out_dtype, total, n8, pixels,
                                                     p->out_sb, p->out_sp);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
