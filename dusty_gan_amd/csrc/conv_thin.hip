// Bandwidth-bound "thin" conv passes, VALU + LDS (nothing here is worth an MFMA: one side has <= 4 channels).
//
//   thin_smallk : MODE_S2, K <= 4 input channels -> N = 64*j output channels
//                 Down1 forward / R1 tangent (K = 2, models/gans/dcgan_eqlr.py:90) and Head backward-data (K = 1..3)
//   thin_smalln : MODE_UP, K = 64*j input channels -> N <= 4 output channels
//                 Head forward (dcgan_eqlr.py:29-46) and Down1 backward-data (N = 2)
//   thin_wgrad  : weight gradients of the same two layers
//
// Every kernel stages the input rows it needs in LDS once (coalesced), keeps the workgroup inside ONE output row so
// the reflect / reflect-adjoint tap list is uniform, and writes whole 128-B channel rows per pixel.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------
// thin_smallk: block = (b, coarse row Y, 64 output columns); thread = 4 pixels x 4 channels (N == 64 per pass).
#define SK_PX 64
template <int KMAX>
__global__ __launch_bounds__(256) void thin_smallk_kernel(ConvP p, int tiles_x, int n_base) {
  __shared__ float s_in[6][2 * SK_PX + 2][KMAX];  // up to 6 source rows x 130 fine columns x K
  __shared__ float s_w[16][KMAX][64];
  __shared__ int s_tap[1 + 2 * 6];
  __shared__ float s_db[64];
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int xt = bid % tiles_x; bid /= tiles_x;
  const int Y = bid % p.Hc, b = bid / p.Hc;
  const int n0 = xt * SK_PX;
  const int Wf = 2 * p.Wc;
  if (tid == 0) {
    int nt = 0;
    for (int i = 0; i < 6; ++i) {
      int r, ky;
      if (dg_tap1d(MODE_S2, p.adj, 0, Y, p.Hc, i, r, ky)) { s_tap[1 + 2 * nt] = r; s_tap[2 + 2 * nt] = ky; ++nt; }
    }
    s_tap[0] = nt;
  }
  if (tid < 64) s_db[tid] = 0.f;
  // weights [tap][k][n] for this pass's 64 output channels
  for (int i = tid; i < 16 * p.K * 64; i += 256) {
    const int n = i & 63, k = (i >> 6) % p.K, t = i / (64 * p.K);
    s_w[t][k][n] = dg_ld(p.w, (long)t * p.w_st + (long)k * p.w_sk + (long)(n_base + n) * p.w_sn, p.w_dtype);
  }
  __syncthreads();
  const int ntap = s_tap[0];
  const int ncol = 2 * SK_PX + 2;
  for (int i = tid; i < ntap * ncol * p.K; i += 256) {
    const int k = i % p.K, c = (i / p.K) % ncol, t = i / (p.K * ncol);
    int col = 2 * n0 - 1 + c;
    if (col < 0) col += Wf; else if (col >= Wf) col -= Wf;
    s_in[t][c][k] = dg_ld(p.in, (long)b * p.in_sb + ((long)s_tap[1 + 2 * t] * Wf + col) * p.in_sp + (long)k * p.in_sk,
                          p.in_dtype);
  }
  __syncthreads();
  const int cg = tid & 15, pg = tid >> 4;  // 4 channels, 4 pixels
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int t = 0; t < ntap; ++t) {
    const int ky = s_tap[2 + 2 * t];
#pragma unroll
    for (int kx = 0; kx < 4; ++kx) {
      for (int k = 0; k < p.K; ++k) {
        const float4 w = *(const float4*)&s_w[ky * 4 + kx][k][cg * 4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float a = s_in[t][2 * (pg * 4 + i) + kx][k];
          acc[i][0] += a * w.x; acc[i][1] += a * w.y; acc[i][2] += a * w.z; acc[i][3] += a * w.w;
        }
      }
    }
  }
  const int n = n_base + cg * 4;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias)
    for (int j = 0; j < 4; ++j) bias[j] = p.bias[(n + j) % p.bias_mod];
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int X = n0 + pg * 4 + i;
    const long o = (long)b * p.out_sb + ((long)Y * p.Wc + X) * p.out_sp + (long)n * p.out_sn;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float auxv = p.epi == EPI_MASK ? dg_ld(p.aux, o + j * p.out_sn, p.out_dtype) : 0.f;
      const float v = dg_epilogue(acc[i][j], p.scale, p.epi, bias[j], auxv);
      dg_st(p.out, o + j * p.out_sn, p.out_dtype, v);
      colsum[j] += v;
    }
  }
  if (p.dbias) {
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(&s_db[cg * 4 + j], colsum[j]);
    __syncthreads();
    if (tid < 64) atomicAdd(&p.dbias[(n_base + tid) % p.bias_mod], s_db[tid] * (p.rowscale ? p.rowscale[b] : 1.f));
  }
}

// ---------------------------------------------------------------------------------------------------------
// thin_smalln: block = (b, coarse row m, 64 coarse columns) -> the 2 x 128 fine outputs of that patch; each of the
// 4 waves owns one output parity (py,px), so its tap weights are wave-uniform (scalar loads).
#define SN_PX 64
template <typename T, int NMAX>
__global__ __launch_bounds__(256) void thin_smalln_kernel(ConvP p, int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // s_in[3 rows][66 cols][K] in T, row stride padded by 16 B
  const int K = p.K;
  const int rowb = K * (int)sizeof(T) + 16;
  T* s_in = (T*)smem;
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int xt = bid % tiles_x; bid /= tiles_x;
  const int m = bid % p.Hc, b = bid / p.Hc;
  const int n0 = xt * SN_PX;
  // stage rows m-1, m, m+1 (those that exist), columns n0-1 .. n0+64 (circular)
  const int cpr = K * (int)sizeof(T) / 16;  // 16-B chunks per pixel
  const T* in = (const T*)p.in;
  for (int i = tid; i < 3 * (SN_PX + 2) * cpr; i += 256) {
    const int ch = i % cpr, c = (i / cpr) % (SN_PX + 2), rr = i / (cpr * (SN_PX + 2));
    const int r = m - 1 + rr;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r >= 0 && r < p.Hc) {
      int col = n0 - 1 + c;
      if (col < 0) col += p.Wc; else if (col >= p.Wc) col -= p.Wc;
      v = *(const uint4*)(in + (long)b * p.in_sb + ((long)r * p.Wc + col) * p.in_sp + ch * (16 / (int)sizeof(T)));
    }
    *(uint4*)(smem + ((rr * (SN_PX + 2) + c) * rowb) + ch * 16) = v;
  }
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int py = wave >> 1, px = wave & 1;
  const int Y = 2 * m + py;
  float acc[NMAX];
#pragma unroll
  for (int j = 0; j < NMAX; ++j) acc[j] = 0.f;
  const float* __restrict__ w = (const float*)p.w;  // fp32 master weights [tap][k][n], uniform indices
  for (int i = 0; i < 4; ++i) {
    int r, ky;
    if (!dg_tap1d(MODE_UP, p.adj, 0, Y, p.Hc, i, r, ky)) continue;
    const int rr = r - (m - 1);
#pragma unroll
    for (int jx = 0; jx < 2; ++jx) {
      const int d = px == 0 ? (jx == 0 ? 0 : -1) : (jx == 0 ? 1 : 0);
      const int kx = px == 0 ? (jx == 0 ? 1 : 3) : (jx == 0 ? 0 : 2);
      const unsigned char* src = smem + ((rr * (SN_PX + 2) + lane + 1 + d) * rowb);
      const float* wt = w + (long)(ky * 4 + kx) * p.w_st;
      for (int k8 = 0; k8 < K; k8 += 16 / (int)sizeof(T)) {
        const uint4 raw = *(const uint4*)(src + k8 * sizeof(T));
        float a[16 / sizeof(T)];
        if (sizeof(T) == 2) {
          const bf16* h = (const bf16*)&raw;
#pragma unroll
          for (int e = 0; e < 8; ++e) a[e] = (float)h[e];
        } else {
          const float* h = (const float*)&raw;
#pragma unroll
          for (int e = 0; e < 4; ++e) a[e] = h[e];
        }
#pragma unroll
        for (int e = 0; e < (int)(16 / sizeof(T)); ++e) {
          const float* wk = wt + (long)(k8 + e) * p.w_sk;
#pragma unroll
          for (int j = 0; j < NMAX; ++j)
            if (j < p.N) acc[j] += a[e] * wk[j * p.w_sn];
        }
      }
    }
  }
  const int X = 2 * (n0 + lane) + px;
  const long o = (long)b * p.out_sb + ((long)Y * (2 * p.Wc) + X) * p.out_sp;
#pragma unroll
  for (int j = 0; j < NMAX; ++j) {
    if (j < p.N) {
      const float sc = p.nscale ? p.scale * p.nscale[j] : p.scale;
      const float v = acc[j] * sc + (p.bias ? p.bias[j % p.bias_mod] : 0.f);
      dg_st(p.out, o + (long)j * p.out_sn, p.out_dtype, v);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// thin_wgrad_down: wmode 0 with Ci <= 4 (Down1: Ci = 2), Co == 64 per pass.
// block = a range of (b, m) coarse rows; thread = (co = tid & 63, ky = tid >> 6): per coarse pixel it reads its
// gradient value once and the 4 x Ci input taps of its kernel row from LDS (wave-uniform address -> broadcast).
template <int CMAX>
__global__ __launch_bounds__(256) void thin_wgrad_down_kernel(WgradP p, int co_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_a = (float*)smem;  // [4 ky][2*Wc + 2][CMAX]
  const int tid = threadIdx.x;
  const int co = tid & 63, ky = tid >> 6;
  const int Wf = 2 * p.Wc, ncol = Wf + 2;
  const long units = (long)p.B * p.Hc;
  const long u0 = units * blockIdx.x / gridDim.x, u1 = units * (blockIdx.x + 1) / gridDim.x;
  float tot[4][CMAX];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < CMAX; ++c) tot[i][c] = 0.f;
  for (long u = u0; u < u1; ++u) {
    const int b = (int)(u / p.Hc), m = (int)(u % p.Hc);
    __syncthreads();
    for (int i = tid; i < 4 * ncol * p.Ci; i += 256) {
      const int c = i % p.Ci, col = (i / p.Ci) % ncol, kk = i / (p.Ci * ncol);
      int ra, rg;
      dg_wgrad1d(0, 0, m, p.Hc, kk, ra, rg);
      int cc = col - 1;
      if (cc < 0) cc += Wf; else if (cc >= Wf) cc -= Wf;
      s_a[(kk * ncol + col) * CMAX + c] =
          dg_ld(p.a, (long)b * p.a_sb + ((long)ra * Wf + cc) * p.a_sp + (long)c * p.a_sc, p.a_dtype);
    }
    __syncthreads();
    float acc[4][CMAX];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < CMAX; ++c) acc[i][c] = 0.f;
    const long gb = (long)b * p.g_sb + (long)m * p.Wc * p.g_sp + (long)(co_base + co) * p.g_sc;
    const float* row = s_a + (long)ky * ncol * CMAX;
#pragma unroll 4
    for (int x = 0; x < p.Wc; ++x) {
      const float g = dg_ld(p.g, gb + (long)x * p.g_sp, p.g_dtype);
      // input columns 2x-1 .. 2x+2 live at LDS columns 2x .. 2x+3
#pragma unroll
      for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int c = 0; c < CMAX; ++c) acc[kx][c] += g * row[(2 * x + kx) * CMAX + c];
    }
    const float rs = p.rowscale ? p.rowscale[b] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < CMAX; ++c) tot[i][c] += rs * acc[i][c];
  }
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < p.Ci) atomicAdd(&p.dw[((long)(ky * 4 + kx) * p.Ci + c) * p.Co + co_base + co], tot[kx][c] * p.scale);
}

// ---------------------------------------------------------------------------------------------------------
// thin_wgrad_up: wmode 1 with Co <= 4 (Head: Co = 1..3), Ci == 64 per pass.
// thread = (ci = tid & 63, ky = tid >> 6); the gradient rows (fine grid, <= 4 channels, any layout) go to LDS.
template <int NMAX>
__global__ __launch_bounds__(256) void thin_wgrad_up_kernel(WgradP p, int ci_base) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_g = (float*)smem;  // [2 py][2*Wc][NMAX]
  const int tid = threadIdx.x;
  const int ci = tid & 63, ky = tid >> 6;
  const int Wf = 2 * p.Wc;
  const long units = (long)p.B * p.Hc;
  const long u0 = units * blockIdx.x / gridDim.x, u1 = units * (blockIdx.x + 1) / gridDim.x;
  const int py = (ky == 0 || ky == 2) ? 1 : 0;
  float tot[4][NMAX];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < NMAX; ++c) tot[i][c] = 0.f;
  for (long u = u0; u < u1; ++u) {
    const int b = (int)(u / p.Hc), m = (int)(u % p.Hc);
    __syncthreads();
    for (int i = tid; i < 2 * Wf * p.Co; i += 256) {
      const int col = i % Wf, c = (i / Wf) % p.Co, pp = i / (Wf * p.Co);
      s_g[(pp * Wf + col) * NMAX + c] =
          dg_ld(p.g, (long)b * p.g_sb + ((long)(2 * m + pp) * Wf + col) * p.g_sp + (long)c * p.g_sc, p.g_dtype);
    }
    __syncthreads();
    int ra, rg;
    dg_wgrad1d(1, 0, m, p.Hc, ky, ra, rg);
    const long ab = (long)b * p.a_sb + (long)ra * p.Wc * p.a_sp + (long)(ci_base + ci) * p.a_sc;
    const float* grow = s_g + (long)py * Wf * NMAX;
    float acc[4][NMAX];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < NMAX; ++c) acc[i][c] = 0.f;
    // sliding window over the input row: a[x-1], a[x], a[x+1] (circular)
    float am = dg_ld(p.a, ab + (long)(p.Wc - 1) * p.a_sp, p.a_dtype);
    float a0 = dg_ld(p.a, ab, p.a_dtype);
#pragma unroll 4
    for (int x = 0; x < p.Wc; ++x) {
      const int xn = x + 1 == p.Wc ? 0 : x + 1;
      const float ap = dg_ld(p.a, ab + (long)xn * p.a_sp, p.a_dtype);
      // kx=1: (px 0, a[x]); kx=3: (px 0, a[x-1]); kx=0: (px 1, a[x+1]); kx=2: (px 1, a[x])
#pragma unroll
      for (int c = 0; c < NMAX; ++c) {
        const float g0 = grow[(2 * x) * NMAX + c], g1 = grow[(2 * x + 1) * NMAX + c];
        acc[1][c] += a0 * g0;
        acc[3][c] += am * g0;
        acc[0][c] += ap * g1;
        acc[2][c] += a0 * g1;
      }
      am = a0;
      a0 = ap;
    }
    const float rs = p.rowscale ? p.rowscale[b] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < NMAX; ++c) tot[i][c] += rs * acc[i][c];
  }
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
      if (c < p.Co) atomicAdd(&p.dw[((long)(ky * 4 + kx) * p.Ci + ci_base + ci) * p.Co + c], tot[kx][c] * p.scale);
}

// ---------------------------------------------------------------------------------------------------------
int dg_conv_thin_supported(const ConvP* p) {
  if (!p->ring || p->mode == MODE_GEMM) return 0;
  if (p->mode == MODE_S2)  // small K -> wide N
    return p->K <= 4 && p->N % 64 == 0 && p->Wc % SK_PX == 0 && !p->nscale && (!p->dbias || p->bias_mod >= p->N);
  // MODE_UP: wide K -> small N; fp32 master weights expected ([tap][k][n] strides given), no mask / bias-grad epilogue
  if (p->N > 4 || p->Wc % SN_PX != 0 || p->in_sk != 1) return 0;
  const int es = p->in_dtype == DG_BF16 ? 2 : 4;
  if ((p->K * es) % 16 != 0 || p->w_dtype != DG_F32) return 0;
  if (p->epi != EPI_LINEAR || p->dbias) return 0;
  return 1;
}

int dg_conv_thin_launch(const ConvP* p, hipStream_t s) {
  if (!dg_conv_thin_supported(p)) return DG_EUNSUPPORTED;
  if (p->mode == MODE_S2) {
    const int tiles_x = p->Wc / SK_PX;
    const unsigned grid = (unsigned)((long)p->B * p->Hc * tiles_x);
    for (int nb = 0; nb < p->N; nb += 64) {
      if (p->K <= 2) thin_smallk_kernel<2><<<grid, 256, 0, s>>>(*p, tiles_x, nb);
      else thin_smallk_kernel<4><<<grid, 256, 0, s>>>(*p, tiles_x, nb);
    }
  } else {
    const int tiles_x = p->Wc / SN_PX;
    const unsigned grid = (unsigned)((long)p->B * p->Hc * tiles_x);
    const int es = p->in_dtype == DG_BF16 ? 2 : 4;
    const size_t lds = (size_t)3 * (SN_PX + 2) * (p->K * es + 16);
    if (lds > 64 * 1024) return DG_EUNSUPPORTED;
    if (p->in_dtype == DG_BF16) thin_smalln_kernel<bf16, 4><<<grid, 256, lds, s>>>(*p, tiles_x);
    else thin_smalln_kernel<float, 4><<<grid, 256, lds, s>>>(*p, tiles_x);
  }
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_wgrad_thin_supported(const WgradP* p) {
  if (!p->ring) return 0;
  if (p->wmode == 0)
    return p->Ci <= 4 && p->Co % 64 == 0 &&
           (size_t)4 * (2 * p->Wc + 2) * (p->Ci <= 2 ? 2 : 4) * sizeof(float) <= 64 * 1024;
  if (p->wmode == 1) return p->Co <= 4 && p->Ci % 64 == 0 && (size_t)2 * 2 * p->Wc * 4 * sizeof(float) <= 64 * 1024;
  return 0;
}

int dg_wgrad_thin_launch(const WgradP* p, hipStream_t s) {
  if (!dg_wgrad_thin_supported(p)) return DG_EUNSUPPORTED;
  const long units = (long)p->B * p->Hc;
  unsigned grid = units < 1024 ? (unsigned)units : 1024u;
  if (p->wmode == 0) {
    for (int cb = 0; cb < p->Co; cb += 64) {
      if (p->Ci <= 2) thin_wgrad_down_kernel<2><<<grid, 256, (size_t)4 * (2 * p->Wc + 2) * 2 * sizeof(float), s>>>(*p, cb);
      else thin_wgrad_down_kernel<4><<<grid, 256, (size_t)4 * (2 * p->Wc + 2) * 4 * sizeof(float), s>>>(*p, cb);
    }
  } else {
    for (int cb = 0; cb < p->Ci; cb += 64)
      thin_wgrad_up_kernel<4><<<grid, 256, (size_t)2 * 2 * p->Wc * 4 * sizeof(float), s>>>(*p, cb);
  }
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
