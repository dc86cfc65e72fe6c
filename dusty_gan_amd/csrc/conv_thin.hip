// Bandwidth-bound "thin" conv passes of the two layers with <= 4 channels on one side (Down1, Head).
//
//   VALU + LDS fall-backs (fp32 mode, shapes the MFMA kernels refuse):
//   thin_smallk : MODE_S2, K <= 4 input channels -> N = 64*j output channels
//                 Down1 forward / R1 tangent (K = 2, models/gans/dcgan_eqlr.py:90) and Head backward-data (K = 1..3)
//   thin_smalln : MODE_UP, K = 64*j input channels -> N <= 4 output channels
//                 Head forward (dcgan_eqlr.py:29-46) and Down1 backward-data (N = 2)
//   thin_wgrad_down / thin_wgrad_up : weight gradients of the same two layers
//
//   bf16 on the matrix cores (what the benchmark step runs; each has its own header comment further down):
//   thin_s2_mfma, thin_up_mfma (+ thin_up_prep), thin_wgrad_down_mfma, thin_wgrad_up_mfma
//
// The fall-backs stage the input rows they need in LDS once (coalesced), keep the workgroup inside ONE output row so the
// reflect / reflect-adjoint tap list is uniform, and write whole 128-B channel rows per pixel.  The MFMA kernels'
// memory schedules are written out by hand (fixed-count unrolled load batches, the next row / group in flight in
// registers, nothing but the prefetch outstanding when a wait comes): hipcc does not unroll a staging loop with a
// run-time trip count, and a load inside an epilogue brings its own s_waitcnt vmcnt(0).
#include "common.h"
#include "mfma_common.h"
#include "thin_up_frag.h"

#include <stdlib.h>
#include <type_traits>

// ---------------------------------------------------------------------------------------------------------
// thin_smallk: block = (b, coarse row Y), walking the row's 64-column tiles; thread = 4 pixels x 4 channels (N == 64 per pass).
// (Round 5: one block per TILE re-loaded the pass's 16 x K x 64 weights and rebuilt the tap list for every 64 pixels and made
// three dependent round trips to memory per 16 KB of output - 141 us for Down1 forward at 64 samples in the fp32x3 mode.  A block
// now keeps weights and taps for the whole row and has the next tile's input window in flight, in registers, while it computes.)
#define SK_PX 64
template <int KMAX>
__global__ __launch_bounds__(256) void thin_smallk_kernel(ConvP p, int tiles_x, int n_base) {
  __shared__ float s_in[6][2 * SK_PX + 2][KMAX];  // up to 6 source rows x 130 fine columns x K
  __shared__ float s_w[16][KMAX][64];
  __shared__ int s_tap[1 + 2 * 6];
  __shared__ float s_db[64];
  const int tid = threadIdx.x;
  const int Y = blockIdx.x % p.Hc, b = blockIdx.x / p.Hc;
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
  // the window of a tile: element i = (tap row t, column c, channel k), NPRE per thread; decoded once (the tile only moves c)
  constexpr int NPRE = (6 * (2 * SK_PX + 2) * KMAX + 255) / 256;
  const int nst = ntap * ncol * p.K;
  int pc[NPRE], pl[NPRE];                         // window column, LDS index
  long pg_[NPRE];                                  // source offset without the column
  float pre[NPRE];
#pragma unroll
  for (int u = 0; u < NPRE; ++u) {
    const int i = tid + 256 * u;
    const int k = i % p.K, c = (i / p.K) % ncol, t = min(i / (p.K * ncol), 5);
    pc[u] = c;
    pl[u] = (t * ncol + c) * KMAX + k;
    pg_[u] = (long)b * p.in_sb + (long)s_tap[1 + 2 * (i < nst ? t : 0)] * Wf * p.in_sp + (long)k * p.in_sk;
  }
  auto fetch = [&](int xt) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < NPRE; ++u) {
      if (tid + 256 * u >= nst) continue;
      int col = 2 * xt * SK_PX - 1 + pc[u];
      if (col < 0) col += Wf; else if (col >= Wf) col -= Wf;
      pre[u] = dg_ld(p.in, pg_[u] + (long)col * p.in_sp, p.in_dtype);
    }
  };
  fetch(0);
  const int cg = tid & 15, pg = tid >> 4;  // 4 channels, 4 pixels
  const int n = n_base + cg * 4;
  float bias[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias)
    for (int j = 0; j < 4; ++j) bias[j] = p.bias[(n + j) % p.bias_mod];
  float colsum[4] = {0.f, 0.f, 0.f, 0.f};
  const bool x2fast = p.out_dtype == DG_BF16X2 && p.out_sn == 1;   // four consecutive channels: 8 bytes of hi, 8 bytes of lo
  for (int xt = 0; xt < tiles_x; ++xt) {
  const int n0 = xt * SK_PX;
  __syncthreads();                                // (the previous tile's reads of s_in are done)
#pragma unroll
  for (int u = 0; u < NPRE; ++u)
    if (tid + 256 * u < nst) (&s_in[0][0][0])[pl[u]] = pre[u];
  __syncthreads();
  if (xt + 1 < tiles_x) fetch(xt + 1);            // in flight during this tile's arithmetic and stores
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
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int X = n0 + pg * 4 + i;
    const long o = (long)b * p.out_sb + ((long)Y * p.Wc + X) * p.out_sp + (long)n * p.out_sn;
    if (x2fast) {
      const long q = dg_x2_index(o);
      uint2 ah = make_uint2(0, 0);
      if (p.epi == EPI_MASK) ah = *(const uint2*)((const unsigned short*)p.aux + q);   // (the sign lives in the hi half)
      const unsigned aw[4] = {ah.x << 16, ah.x & 0xffff0000u, ah.y << 16, ah.y & 0xffff0000u};
      unsigned hw[4], lw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = dg_epilogue(acc[i][j], p.scale, p.epi, bias[j], __builtin_bit_cast(float, aw[j]));
        const bf16 h = (bf16)v;
        hw[j] = __builtin_bit_cast(unsigned short, h);
        lw[j] = __builtin_bit_cast(unsigned short, (bf16)(v - (float)h));
        colsum[j] += v;
      }
      *(uint2*)((unsigned short*)p.out + q) = make_uint2(hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16));
      *(uint2*)((unsigned short*)p.out + q + 64) = make_uint2(lw[0] | (lw[1] << 16), lw[2] | (lw[3] << 16));
      continue;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float auxv = p.epi == EPI_MASK ? dg_ld(p.aux, o + j * p.out_sn, p.out_dtype) : 0.f;
      const float v = dg_epilogue(acc[i][j], p.scale, p.epi, bias[j], auxv);
      dg_st(p.out, o + j * p.out_sn, p.out_dtype, v);
      colsum[j] += v;
    }
  }
  }   // tiles of the row
  if (p.dbias) {
    // the block's 64 channel sums in a fixed order (16 pixel groups per channel through LDS), then - with the caller's staging
    // scratch (DgConv.dbias_ws, zero on entry and left zero) - order-independent across blocks: 32.32 fixed-point integer
    // adds onto 64 staging words, a ticket, and the LAST block adds the totals onto dbias once (round 5: the fp32 modes'
    // bias gradients of Down1 / Up3 were float atomics in arrival order)
    __syncthreads();                              // (s_in is dead: its first 16 x 64 floats hold the partial rows)
    float* part = &s_in[0][0][0];
#pragma unroll
    for (int j = 0; j < 4; ++j) part[pg * 64 + cg * 4 + j] = colsum[j];
    __syncthreads();
    if (tid < 64) {
      float v = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) v += part[r * 64 + tid];
      v *= p.rowscale ? p.rowscale[b] : 1.f;
      if (p.dbias_ws && fabsf(v) < 2147483000.f) {
        unsigned long long* w = (unsigned long long*)p.dbias_ws;
        atomicAdd(&w[tid], (unsigned long long)__double2ll_rn((double)v * 4294967296.0));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (memory-side atomics acknowledged before the ticket is drawn)
      } else {
        atomicAdd(&p.dbias[(n_base + tid) % p.bias_mod], v);
      }
    }
    if (p.dbias_ws) {
      __shared__ unsigned s_ticket;
      __syncthreads();
      if (tid == 0) s_ticket = atomicAdd((unsigned*)((unsigned long long*)p.dbias_ws + 64), 1u);
      __syncthreads();
      if (s_ticket == gridDim.x - 1 && tid < 64) {
        unsigned long long* w = (unsigned long long*)p.dbias_ws;
        const long long tot = (long long)atomicExch(&w[tid], 0ull);
        atomicAdd(&p.dbias[(n_base + tid) % p.bias_mod], (float)((double)tot * (1.0 / 4294967296.0)));
        if (tid == 0) atomicExch((unsigned*)(w + 64), 0u);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// thin_smalln: block = (b, coarse row m), looping over 64-column tiles -> the 2 x 128 fine outputs of each tile.
// Each of the 4 waves owns one output parity (py,px): its tap weights are wave-uniform (LDS broadcast reads in the
// bf16 build, where v_dot2c_f32_bf16 does 2 MACs per VALU instruction with no converts; plain loads + v_fmac in the
// fp32 build).  Input rows m-1, m, m+1 of the tile are staged in LDS once.
// Weights: the T shadow laid out [tap][n][k] (k contiguous).
#define SN_PX 64
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

// X2 (T = float): the input is DG_BF16X2 - the staged pixel rows are the same 4 K bytes, read as hi + lo pairs
template <typename T, int N, bool X2 = false>
__global__ __launch_bounds__(256) void thin_smalln_kernel(ConvP p) {
  static_assert(!X2 || sizeof(T) == 4, "DG_BF16X2 input: the fp32 build");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int ES = sizeof(T);
  const int K = p.K;
  const int rowb = K * ES + 16;                  // padded LDS pixel stride
  const int tid = threadIdx.x;
  const int m = blockIdx.x % p.Hc, b = blockIdx.x / p.Hc;
  const int cpr = K * ES / 16;                   // 16-B chunks per pixel
  const T* in = (const T*)p.in;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  const int py = wave >> 1, px = wave & 1;
  const int Y = 2 * m + py;
  // this wave's taps (wave-uniform): up to 3 row taps (2 regular + 1 reflect-adjoint extra) x 2 column taps
  int trow[3], tky[3], nrow = 0;
  for (int i = 0; i < 4; ++i) {
    int r, ky;
    if (nrow < 3 && dg_tap1d(MODE_UP, p.adj, 0, Y, p.Hc, i, r, ky)) { trow[nrow] = r - (m - 1); tky[nrow] = ky; ++nrow; }
  }
  const int dcol[2] = {px == 0 ? 0 : 1, px == 0 ? -1 : 0};
  const int kxs[2] = {px == 0 ? 1 : 0, px == 0 ? 3 : 2};
  // bf16: the whole [16][N][K] weight block lives in LDS behind the input strip (wave-uniform reads broadcast)
  unsigned char* s_w = smem + 3 * (SN_PX + 2) * rowb;
  if constexpr (ES == 2) {
    for (int i = tid; i < 16 * N * K / 8; i += 256) {
      const int k8 = i % (K / 8), j = (i / (K / 8)) % N, t = i / (K / 8 * N);
      *(uint4*)(s_w + ((t * N + j) * K + k8 * 8) * 2) =
          *(const uint4*)((const T*)p.w + (long)t * p.w_st + (long)j * p.w_sn + k8 * 8);
    }
  }
  for (int n0 = 0; n0 < p.Wc; n0 += SN_PX) {
    __syncthreads();
    for (int i = tid; i < 3 * (SN_PX + 2) * cpr; i += 256) {
      const int ch = i % cpr, c = (i / cpr) % (SN_PX + 2), rr = i / (cpr * (SN_PX + 2));
      const int r = m - 1 + rr;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r >= 0 && r < p.Hc) {
        int col = n0 - 1 + c;
        if (col < 0) col += p.Wc; else if (col >= p.Wc) col -= p.Wc;
        v = *(const uint4*)(in + (long)b * p.in_sb + ((long)r * p.Wc + col) * p.in_sp + ch * (16 / ES));
      }
      *(uint4*)(smem + ((rr * (SN_PX + 2) + c) * rowb) + ch * 16) = v;
    }
    __syncthreads();
    float acc[N];
#pragma unroll
    for (int j = 0; j < N; ++j) acc[j] = 0.f;
    for (int ti = 0; ti < nrow; ++ti) {
#pragma unroll
      for (int jx = 0; jx < 2; ++jx) {
        const unsigned char* src = smem + ((trow[ti] * (SN_PX + 2) + lane + 1 + dcol[jx]) * rowb);
        const T* wt = (const T*)p.w + (long)(tky[ti] * 4 + kxs[jx]) * p.w_st;  // [n][k] of this tap, uniform
        if constexpr (X2) {
          for (int k0 = 0; k0 < K; k0 += 8) {
            const unsigned char* q = src + (k0 >> 6) * 256 + (k0 & 63) * 2;
            const uint4 h = *(const uint4*)q, l = *(const uint4*)(q + 128);
            const unsigned hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
            float a8[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              a8[2 * e] = __builtin_bit_cast(float, hw[e] << 16) + __builtin_bit_cast(float, lw[e] << 16);
              a8[2 * e + 1] = __builtin_bit_cast(float, hw[e] & 0xffff0000u) + __builtin_bit_cast(float, lw[e] & 0xffff0000u);
            }
#pragma unroll
            for (int j = 0; j < N; ++j) {
              const float* wq = (const float*)(wt + (long)j * p.w_sn + k0);
#pragma unroll
              for (int e = 0; e < 8; ++e) acc[j] += a8[e] * wq[e];
            }
          }
        } else
        for (int k0 = 0; k0 < K; k0 += 16 / ES) {
          const uint4 raw = *(const uint4*)(src + k0 * ES);
          if constexpr (ES == 2) {
            const unsigned a4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int j = 0; j < N; ++j) {
              const uint4 wr = *(const uint4*)(s_w + (((tky[ti] * 4 + kxs[jx]) * N + j) * K + k0) * 2);
              const unsigned wq[4] = {wr.x, wr.y, wr.z, wr.w};
#pragma unroll
              for (int q = 0; q < 4; ++q)
                acc[j] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a4[q]),
                                                         __builtin_bit_cast(bf16x2, wq[q]), acc[j], false);
            }
          } else {
            const float a4[4] = {__builtin_bit_cast(float, raw.x), __builtin_bit_cast(float, raw.y),
                                 __builtin_bit_cast(float, raw.z), __builtin_bit_cast(float, raw.w)};
#pragma unroll
            for (int j = 0; j < N; ++j) {
              const float* wq = (const float*)(wt + (long)j * p.w_sn + k0);
#pragma unroll
              for (int q = 0; q < 4; ++q) acc[j] += a4[q] * wq[q];
            }
          }
        }
      }
    }
    const int X = 2 * (n0 + lane) + px;
    const long o = (long)b * p.out_sb + ((long)Y * (2 * p.Wc) + X) * p.out_sp;
#pragma unroll
    for (int j = 0; j < N; ++j) {
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
    if (CMAX == 2 && p.Ci == 2 && p.a_dtype == DG_F32 && p.a_sc == 1 && p.a_sp == 2 && p.a_sb % 4 == 0 && ((size_t)p.a & 15) == 0) {
      // the two-channel fp32 image (Down1 in the fp32 modes): a row is 2 Wf contiguous floats - 16-byte loads of two pixels
      // instead of a scalar load with two integer divisions per element (32 per thread and row set at Wf = 1024)
      const float* A = (const float*)p.a + (long)b * p.a_sb;
      for (int i = tid; i < 4 * (Wf / 2); i += 256) {
        const int j = i % (Wf / 2), kk = i / (Wf / 2);
        int ra, rg;
        dg_wgrad1d(0, 0, m, p.Hc, kk, ra, rg);
        const float4 v = *(const float4*)(A + ((long)ra * Wf + 2 * j) * 2);
        float* d = s_a + ((long)kk * ncol + 2 * j + 1) * 2;          // (column cc lives at LDS column cc + 1: 8-byte aligned)
        *(float2*)d = make_float2(v.x, v.y);
        *(float2*)(d + 2) = make_float2(v.z, v.w);
      }
      if (tid < 8) {                                                  // the circular halo: column -1 = Wf - 1, column Wf = 0
        const int kk = tid >> 1, hi = tid & 1;
        int ra, rg;
        dg_wgrad1d(0, 0, m, p.Hc, kk, ra, rg);
        const float2 v = *(const float2*)(A + ((long)ra * Wf + (hi ? 0 : Wf - 1)) * 2);
        *(float2*)(s_a + ((long)kk * ncol + (hi ? Wf + 1 : 0)) * 2) = v;
      }
    } else
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
    auto walk = [&](auto loadg) __attribute__((always_inline)) {
#pragma unroll 4
      for (int x = 0; x < p.Wc; ++x) {
        const float g = loadg(x);
        // input columns 2x-1 .. 2x+2 live at LDS columns 2x .. 2x+3
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
#pragma unroll
          for (int c = 0; c < CMAX; ++c) acc[kx][c] += g * row[(2 * x + kx) * CMAX + c];
      }
    };
    if (p.g_dtype == DG_BF16X2 && p.g_sc == 1 && p.g_sp % 64 == 0) {
      // split-bf16 gradient rows: the pixel stride is whole channel groups, so the (hi, lo) pair of this thread's channel
      // moves by a constant 2 g_sp halves per pixel (the generic dg_ld redoes the 64-bit index split per element)
      const unsigned short* gq = (const unsigned short*)p.g + dg_x2_index(gb);
      const long gs2 = 2 * p.g_sp;
      walk([&](int x) {
        const unsigned short* q = gq + (long)x * gs2;
        return __builtin_bit_cast(float, (unsigned)q[0] << 16) + __builtin_bit_cast(float, (unsigned)q[64] << 16);
      });
    } else {
      walk([&](int x) { return dg_ld(p.g, gb + (long)x * p.g_sp, p.g_dtype); });
    }
    const float rs = p.rowscale ? p.rowscale[b] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < CMAX; ++c) tot[i][c] += rs * acc[i][c];
  }
  // p.ws: the block's partial tile with plain stores (summed by dg_wgrad_reduce in a fixed order) instead of atomics on dw
  float* wsb = p.ws ? p.ws + (long)blockIdx.x * 16 * p.Ci * p.Co : nullptr;
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
#pragma unroll
    for (int c = 0; c < CMAX; ++c)
      if (c < p.Ci) {
        const long o = ((long)(ky * 4 + kx) * p.Ci + c) * p.Co + co_base + co;
        if (wsb) wsb[o] = tot[kx][c] * p.scale; else atomicAdd(&p.dw[o], tot[kx][c] * p.scale);
      }
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
    auto walk = [&](auto loada) __attribute__((always_inline)) {
      float am = loada(p.Wc - 1);
      float a0 = loada(0);
#pragma unroll 8   // (measured in the fp32x3 step: 4 -> 113 us, 8 -> 95 us, 16 -> 115 us; thin_wgrad_down: 2 / 4 / 8 -> 247 / 164 / 206 us)
      for (int x = 0; x < p.Wc; ++x) {
        const int xn = x + 1 == p.Wc ? 0 : x + 1;
        const float ap = loada(xn);
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
    };
    if (p.a_dtype == DG_BF16X2 && p.a_sc == 1 && p.a_sp % 64 == 0) {   // (as in thin_wgrad_down: constant stride between pairs)
      const unsigned short* aq = (const unsigned short*)p.a + dg_x2_index(ab);
      const long as2 = 2 * p.a_sp;
      walk([&](int x) {
        const unsigned short* q = aq + (long)x * as2;
        return __builtin_bit_cast(float, (unsigned)q[0] << 16) + __builtin_bit_cast(float, (unsigned)q[64] << 16);
      });
    } else {
      walk([&](int x) { return dg_ld(p.a, ab + (long)x * p.a_sp, p.a_dtype); });
    }
    const float rs = p.rowscale ? p.rowscale[b] : 1.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < NMAX; ++c) tot[i][c] += rs * acc[i][c];
  }
  float* wsb = p.ws ? p.ws + (long)blockIdx.x * 16 * p.Ci * p.Co : nullptr;   // (as thin_wgrad_down)
#pragma unroll
  for (int kx = 0; kx < 4; ++kx)
#pragma unroll
    for (int c = 0; c < NMAX; ++c)
      if (c < p.Co) {
        const long o = ((long)(ky * 4 + kx) * p.Ci + ci_base + ci) * p.Co + c;
        if (wsb) wsb[o] = tot[kx][c] * p.scale; else atomicAdd(&p.dw[o], tot[kx][c] * p.scale);
      }
}

// ---------------------------------------------------------------------------------------------------------
// thin_wgrad_down_mfma (bf16, Ci == 2, Co == 64): Down1's weight gradient on the matrix cores.
//   dW[(ky,kx,ci) = 32][co = 64] = sum_pixels A[pixel][(ky,kx,ci)] * G[pixel][co]
// GEMM view: M = 32 (one MFMA tile), N = 64 (two tiles), K = coarse pixels.  A block owns ROWS_PB consecutive
// coarse rows of one sample; the 4 input rows a coarse row touches are staged in LDS as (ci0,ci1) dwords and each
// wave walks a quarter of the row in 16-pixel K steps: the A fragment (8 consecutive pixels of one (tap,ci)) is
// gathered from the staged rows, the G fragment comes from a wave-private [16][64] LDS tile through the
// transposing read ds_read_b64_tr_b16.  Waves are reduced through LDS, then one fp32 atomic per element per block.
typedef __attribute__((ext_vector_type(8))) __bf16 tw_bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 tw_bf16x4;
typedef __attribute__((ext_vector_type(16))) float tw_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned tw_u32x4;
#define WG_ROWS_PB 2
#define WG_GS 4                                  // K steps (16 pixels each) whose gradient tiles a wave fetches at once

// Memory schedule (round 2; the first version made one global round trip per staged dword batch and per K step - loops
// hipcc does not unroll: load, s_waitcnt vmcnt(0), ds_write): the four input rows of a coarse row are fetched as NPT
// sixteen-byte pieces per thread, the NEXT row's pieces in flight while the current row is computed; a wave requests the
// gradient tiles of WG_GS K steps (2 x 16 B per lane each) in one batch before the row's barrier, so a row costs about one
// exposed round trip instead of ~24.  LDS row layout: pixel c at dword c + 4 (16-byte aligned pieces), the circular halo
// pixels -1 / Wf at dwords 3 / Wf + 4.  The cross-wave reduction buffer aliases the staging area (32 KB per block).
template <int NPT>
__global__ __launch_bounds__(256) void thin_wgrad_down_mfma_kernel(WgradP p, int rows_pb) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int Wf = 2 * p.Wc, ncol = Wf + 8;
  unsigned* s_a = (unsigned*)smem;                                   // [4 ky][ncol] dwords = (ci0, ci1)
  unsigned char* s_g = smem + (size_t)4 * ncol * 4;                  // [4 waves][16 px][144 B]
  float* s_red = (float*)smem;                                       // [4 waves][32][64] fp32 - after the row loop
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long units = (long)p.B * p.Hc;
  const long u0 = (long)blockIdx.x * rows_pb;
  const bf16* A = (const bf16*)p.a;
  const bf16* G = (const bf16*)p.g;
  // lane roles
  const int lr = lane & 31, lh = lane >> 5;
  const int m_ky = lr >> 3, m_kx = (lr >> 1) & 3, m_ci = lr & 1;     // A row of this lane: m = (ky*4+kx)*2+ci
  const int g16 = lane >> 4, i16 = lane & 15;
  const int kh = g16 >> 1, cb = g16 & 1, q = i16 >> 2, pp = i16 & 3; // transposing-read roles (see wgrad_mfma.hip)
  unsigned char* my_g = s_g + wave * 16 * 144;
  tw_f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  const int seg = p.Wc / 4;                                          // pixels per wave per row
  const int npc = Wf / 4;                                            // 16-byte pieces per staged row (4 rows: Wf pieces)
  tw_u32x4 sa[NPT];
  unsigned sh = 0;
  auto fetch_a = [&](long u) __attribute__((always_inline)) {
    const int b = (int)(u / p.Hc), Y = (int)(u % p.Hc);
    const bf16* img = A + (long)b * p.a_sb;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int i = tid + 256 * k;
      if (i < Wf) {
        const int kk = i / npc, pc = i % npc;
        int ra, rg;
        dg_wgrad1d(0, 0, Y, p.Hc, kk, ra, rg);
        sa[k] = *(const tw_u32x4*)(img + ((long)ra * Wf + 4 * pc) * 2);
      }
    }
    if (tid < 8) {                                                   // halo: pixel Wf - 1 in front, pixel 0 behind
      int ra, rg;
      dg_wgrad1d(0, 0, Y, p.Hc, tid >> 1, ra, rg);
      sh = *(const unsigned*)(img + ((long)ra * Wf + ((tid & 1) ? 0 : Wf - 1)) * 2);
    }
  };
  auto put_a = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int i = tid + 256 * k;
      if (i < Wf) *(tw_u32x4*)(s_a + (i / npc) * ncol + 4 + 4 * (i % npc)) = sa[k];
    }
    if (tid < 8) s_a[(tid >> 1) * ncol + ((tid & 1) ? Wf + 4 : 3)] = sh;
  };
  const long uend = u0 + rows_pb < units ? u0 + rows_pb : units;
  // the K steps of the block's rows in groups of WG_GS: group gq = (row u0 + gq / gpr, steps (gq % gpr) * WG_GS ...)
  const int spr = seg / 16, gpr = (spr + WG_GS - 1) / WG_GS, ngr = (int)(uend - u0) * gpr;
  auto load_group = [&](int gq, tw_u32x4 (&gt)[WG_GS][2]) __attribute__((always_inline)) {
    const long u = u0 + gq / gpr;
    const int s0 = (gq % gpr) * WG_GS;
    const int b = (int)(u / p.Hc), Y = (int)(u % p.Hc);
    const int bg = p.g_mod > 0 ? b % p.g_mod : b;                    // (one launch over real | fake | tangent input samples)
    const bf16* grow = G + (long)bg * p.g_sb + ((long)Y * p.Wc + wave * seg + 16 * s0) * p.g_sp;
    // the gradient tiles of the group's K steps: G[xb .. xb+15][0..63] (128 B per pixel) = 2 x (64 lanes x 16 B) each
#pragma unroll
    for (int st = 0; st < WG_GS; ++st)
      if (s0 + st < spr) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int c = lane + 64 * v, row = c >> 3, part = c & 7;
          gt[st][v] = *(const tw_u32x4*)(grow + (long)(16 * st + row) * p.g_sp + part * 8);
        }
      }
  };
  const bf16* s_a16 = (const bf16*)s_a;
  auto run_group = [&](int gq, const tw_u32x4 (&gt)[WG_GS][2]) __attribute__((always_inline)) {
    if (gq % gpr == 0) {                                             // first group of a row: its staged input rows
      __syncthreads();                                               // (the previous row's gathers are done)
      put_a();
      __syncthreads();
      if (u0 + gq / gpr + 1 < uend) fetch_a(u0 + gq / gpr + 1);      // in flight during this row's K steps
    }
    const int s0 = (gq % gpr) * WG_GS;
#pragma unroll
    for (int st = 0; st < WG_GS; ++st) {
      if (s0 + st >= spr) break;
      const int xb = wave * seg + 16 * (s0 + st);
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int c = lane + 64 * v, row = c >> 3, part = c & 7;
        *(tw_u32x4*)(my_g + row * 144 + part * 16) = gt[st][v];
      }
      // A fragment: pixels xb + 8 lh + j, j = 0..7, of this lane's (ky,kx,ci): fine column 2 x + kx - 1 -> dword + 4
      tw_bf16x8 fa;
#pragma unroll
      for (int j = 0; j < 8; ++j)
        fa[j] = s_a16[((long)m_ky * ncol + 2 * (xb + 8 * lh + j) + m_kx + 3) * 2 + m_ci];
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const unsigned char* ptr = my_g + (8 * kh + q) * 144 + (jt * 32 + 16 * cb + 4 * pp) * 2;
        const tw_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tw_bf16x4 __attribute__((address_space(3)))*)(ptr));
        const tw_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tw_bf16x4 __attribute__((address_space(3)))*)(ptr + 4 * 144));
        const tw_bf16x8 fg = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fg, acc[jt], 0, 0, 0);
      }
    }
  };
  // two register sets of gradient tiles: the next group's loads are in flight while the current group computes
  tw_u32x4 gta[WG_GS][2], gtb[WG_GS][2];
  fetch_a(u0);
  load_group(0, gta);
  for (int gq = 0; gq < ngr; gq += 2) {
    if (gq + 1 < ngr) load_group(gq + 1, gtb);
    run_group(gq, gta);
    if (gq + 2 < ngr) load_group(gq + 2, gta);
    if (gq + 1 < ngr) run_group(gq + 1, gtb);
  }
  // reduce the 4 waves, then one atomic per element.  D layout: col = lane & 31 (co), row = (e&3)+8(e>>2)+4 lh (m)
  const int b0 = (int)(u0 / p.Hc);
  const float sc = p.scale * (p.rowscale ? p.rowscale[b0] : 1.f);
  __syncthreads();
#pragma unroll
  for (int jt = 0; jt < 2; ++jt)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int mrow = (e & 3) + 8 * (e >> 2) + 4 * lh;
      s_red[(wave * 32 + mrow) * 64 + jt * 32 + lr] = acc[jt][e];
    }
  __syncthreads();
  // p.ws: the block's partial tile with plain stores (summed by dg_wgrad_reduce, fixed order) instead of 2048 atomics on
  // the 8 KB every block of the launch adds into
  float* wsb = p.ws ? p.ws + (long)blockIdx.x * 2048 : nullptr;
  for (int i = tid; i < 32 * 64; i += 256) {
    const float v = s_red[i] + s_red[2048 + i] + s_red[4096 + i] + s_red[6144 + i];
    if (wsb) wsb[i] = v * sc; else atomicAdd(&p.dw[i], v * sc);
  }
}

// ---------------------------------------------------------------------------------------------------------
// thin_wgrad_up_mfma (bf16, Ci == 64, Co <= 2, gradient pixel-major [fine pixel][2]): Head's weight gradient on the
// matrix cores.   dW[(ky,kx,co) = 32][ci = 64] = sum over input pixels (r, xi) of  Bm[(ky,kx,co)][r, xi] * a[r, xi][ci]
// The sum runs over INPUT pixels, so the 64-channel operand `a` is tap-independent and is read exactly once; the tap
// structure sits in the thin operand: for input row r a block builds the 32 "im2col" rows
//     Bm[(ky,kx,co)][xi] = g[fine row 2 (r - d_ky) + par_ky][fine col 2 ((xi - d_kx) mod Wc) + par_kx][co]
// in LDS (zero where r - d_ky leaves the grid; the two reflected rows of models/ops/common.py:9-20 add their mirror
// row: r = 1 takes fine row 0 through ky = 3, r = Hc-2 takes fine row 2Hc-1 through ky = 0 - the inverse of
// dg_wgrad1d(1, ...)).  Each wave then walks its share of the row in 16-pixel K steps: Bm fragment by one
// ds_read_b128, the `a` fragment from a wave-private [16][64] tile through the transposing read.  Waves are reduced
// through LDS, then one fp32 atomic per element per block.
#define WGU_ROWS_PB 2
#define WGU_GS 4                                 // K steps (16 pixels each) whose `a` tiles a wave fetches at once
#define WGU_TB 4                                 // im2col tasks per thread whose gradient dwords are fetched at once

// Memory schedule: the `a` tiles come in groups of WGU_GS K steps, the next group's loads in flight while the current
// one is computed, and the gradient dwords of WGU_TB im2col tasks per thread are requested in one batch (the first
// version made one global round trip per K step and per task: load, s_waitcnt vmcnt(0), ds_write).
// NP = 2 (gradient padded to four channels, Co = 3 or 4: the dusty2 head): both channel pairs in one pass - the 64-channel
// operand, its staging and its transposing reads are shared by the two pairs' MFMAs (a second pass read it again: 54 + 34 us
// for the three-head gradient against 39 us for one pair).
template <int NP>
__global__ __launch_bounds__(256) void thin_wgrad_up_mfma_kernel(WgradP p, int gpair) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TB = NP == 2 ? WGU_TB / 2 : WGU_TB;                  // (the same dwords in flight per thread)
  const int Wc = p.Wc, Wf = 2 * p.Wc;
  const int RSB = Wc * 2 + 16;                                       // im2col row stride (bytes), +16 B: bank spread
  unsigned char* s_b = smem;                                         // [NP][32 n][RSB]
  unsigned char* s_t = smem + (size_t)NP * 32 * RSB;                 // [4 waves][16 px][144 B]
  float* s_red = (float*)smem;                                       // [4 waves][NP 32][64] fp32 (aliases s_b at the end)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long u0 = (long)blockIdx.x * WGU_ROWS_PB;
  const bf16* A = (const bf16*)p.a;
  const unsigned* G = (const unsigned*)p.g + gpair;                  // one dword = channels (2 gpair, 2 gpair + 1) of a fine pixel
  const int gsd = (int)p.g_sp / 2;                                   // dwords per fine pixel (1: two channels, 2: four)
  const int lr = lane & 31, lh = lane >> 5;
  const int g16 = lane >> 4, i16 = lane & 15;
  const int kh = g16 >> 1, cb = g16 & 1, q = i16 >> 2, pp = i16 & 3; // transposing-read roles (see wgrad_mfma.hip)
  unsigned char* my_t = s_t + wave * 16 * 144;
  tw_f32x16 acc[NP][2];
#pragma unroll
  for (int pr = 0; pr < NP; ++pr)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[pr][j][e] = 0.f;
  const int nchunk = Wc / 8;
  const int b = (int)(u0 / p.Hc);
  // the K steps of the block's rows in groups of WGU_GS: a wave's step k of a row is the 16 pixels at wave * 16 + 64 k
  const int spr = Wc / 64, gpr = (spr + WGU_GS - 1) / WGU_GS, ngr = WGU_ROWS_PB * gpr;
  tw_u32x4 nxt[WGU_GS][2];
  auto load_group = [&](int gq) __attribute__((always_inline)) {
    const int r = (int)((u0 + gq / gpr) % p.Hc), s0 = (gq % gpr) * WGU_GS;
    const bf16* arow = A + (long)b * p.a_sb + (long)r * Wc * p.a_sp;
    // a[xb .. xb+15][0..63] (128 B per pixel) = 2 x (64 lanes x 16 B) per step
#pragma unroll
    for (int st = 0; st < WGU_GS; ++st)
      if (s0 + st < spr) {
        const int xb = wave * 16 + 64 * (s0 + st);
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          const int c = lane + 64 * v, row = c >> 3, part = c & 7;
          nxt[st][v] = *(const tw_u32x4*)(arow + (long)(xb + row) * p.a_sp + part * 8);
        }
      }
  };
  load_group(0);
  for (int gq = 0; gq < ngr; ++gq) {
    const int r = (int)((u0 + gq / gpr) % p.Hc), s0 = (gq % gpr) * WGU_GS;
    if (gq % gpr == 0) {
      __syncthreads();
      // ---- im2col rows of input row r: task = (tap, chunk of 8 input pixels), both co at once; TB tasks per thread
      //      and batch: all their gradient dwords are requested before the first is packed
      const unsigned* Gb = G + (long)b * (p.g_sb / 2);
      for (int tb = tid; tb < 16 * nchunk; tb += 256 * TB) {
        unsigned gv[TB][8][NP];
#pragma unroll
        for (int k = 0; k < TB; ++k) {
          const int t = tb + 256 * k;
          if (t < 16 * nchunk) {
            const int tap = t / nchunk, xi0 = (t % nchunk) * 8;
            const int ky = tap >> 2, kx = tap & 3;
            const int dky = ky == 0 ? 1 : (ky == 3 ? -1 : 0), pky = (ky == 0 || ky == 2) ? 1 : 0;
            const int dkx = kx == 0 ? 1 : (kx == 3 ? -1 : 0), pkx = (kx == 0 || kx == 2) ? 1 : 0;
            const int m = r - dky;
            int fr0 = (m >= 0 && m < p.Hc) ? 2 * m + pky : -1;           // fine row of the regular term
            if (fr0 < 0) {                                               // only the mirror term of a reflected row is left
              if (ky == 3 && r == 1) fr0 = 0;
              if (ky == 0 && r == p.Hc - 2) fr0 = 2 * p.Hc - 1;
            }
            const unsigned* g0 = Gb + (long)(fr0 < 0 ? 0 : fr0) * Wf * gsd;   // (fr0 < 0: loaded, not used)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              int x = xi0 + j - dkx;
              if (x < 0) x += Wc; else if (x >= Wc) x -= Wc;
              if (NP == 1) {
                gv[k][j][0] = g0[(2 * x + pkx) * gsd];
              } else {                                                   // (gsd == 2: the pixel's four channels in 8 bytes)
                const uint2 t2 = *(const uint2*)(g0 + (2 * x + pkx) * 2);
                gv[k][j][0] = t2.x; gv[k][j][NP - 1] = t2.y;
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < TB; ++k) {
          const int t = tb + 256 * k;
          if (t < 16 * nchunk) {
            const int tap = t / nchunk, xi0 = (t % nchunk) * 8;
            const int ky = tap >> 2, kx = tap & 3;
            const int dky = ky == 0 ? 1 : (ky == 3 ? -1 : 0);
            const int dkx = kx == 0 ? 1 : (kx == 3 ? -1 : 0), pkx = (kx == 0 || kx == 2) ? 1 : 0;
            const int m = r - dky;
            const bool reg_ok = m >= 0 && m < p.Hc;
            int fr1 = -1;                                                // mirror term of the reflected rows
            if (ky == 3 && r == 1) fr1 = 0;
            if (ky == 0 && r == p.Hc - 2) fr1 = 2 * p.Hc - 1;
            const bool any = reg_ok || fr1 >= 0;
            if (!reg_ok) fr1 = -1;                                       // (the mirror row then IS gv)
#pragma unroll
            for (int pr = 0; pr < NP; ++pr) {
              unsigned lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};       // co0 / co1 of the pair, 8 bf16 each
              if (any) {
                const unsigned* g1 = fr1 >= 0 ? Gb + (long)fr1 * Wf * gsd + pr : nullptr;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  unsigned v = gv[k][j][pr];
                  if (g1) {                                              // sum of two gradient rows, rounded once to bf16
                    int x = xi0 + j - dkx;
                    if (x < 0) x += Wc; else if (x >= Wc) x -= Wc;
                    const unsigned w2 = g1[(2 * x + pkx) * gsd];
                    const float s0f = __builtin_bit_cast(float, v << 16) + __builtin_bit_cast(float, w2 << 16);
                    const float s1f = __builtin_bit_cast(float, v & 0xffff0000u) + __builtin_bit_cast(float, w2 & 0xffff0000u);
                    const bf16 h0 = (bf16)s0f, h1 = (bf16)s1f;
                    v = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                  }
                  const unsigned c0 = v & 0xffffu, c1 = v >> 16;
                  if (j & 1) { lo[j >> 1] |= c0 << 16; hi[j >> 1] |= c1 << 16; }
                  else { lo[j >> 1] = c0; hi[j >> 1] = c1; }
                }
              }
              *(uint4*)(s_b + (size_t)(pr * 32 + tap * 2 + 0) * RSB + xi0 * 2) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
              *(uint4*)(s_b + (size_t)(pr * 32 + tap * 2 + 1) * RSB + xi0 * 2) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            }
          }
        }
      }
      __syncthreads();
    }
    tw_u32x4 cur[WGU_GS][2];
#pragma unroll
    for (int st = 0; st < WGU_GS; ++st)
#pragma unroll
      for (int v = 0; v < 2; ++v) cur[st][v] = nxt[st][v];
    if (gq + 1 < ngr) load_group(gq + 1);                            // in flight during this group's K steps
#pragma unroll
    for (int st = 0; st < WGU_GS; ++st) {
      if (s0 + st >= spr) break;
      const int xb = wave * 16 + 64 * (s0 + st);
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int c = lane + 64 * v, row = c >> 3, part = c & 7;
        *(tw_u32x4*)(my_t + row * 144 + part * 16) = cur[st][v];
      }
      tw_bf16x8 fa[NP];
#pragma unroll
      for (int pr = 0; pr < NP; ++pr) fa[pr] = *(const tw_bf16x8*)(s_b + (size_t)(pr * 32 + lr) * RSB + (xb + 8 * lh) * 2);
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        const unsigned char* ptr = my_t + (8 * kh + q) * 144 + (jt * 32 + 16 * cb + 4 * pp) * 2;
        const tw_bf16x4 l4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tw_bf16x4 __attribute__((address_space(3)))*)(ptr));
        const tw_bf16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((tw_bf16x4 __attribute__((address_space(3)))*)(ptr + 4 * 144));
        const tw_bf16x8 fg = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int pr = 0; pr < NP; ++pr) acc[pr][jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pr], fg, acc[pr][jt], 0, 0, 0);
      }
    }
  }
  // reduce the 4 waves, then one atomic per element.  D layout: col = lane & 31 (ci), row = (e&3)+8(e>>2)+4 lh (n)
  const float sc = p.scale * (p.rowscale ? p.rowscale[b] : 1.f);
  __syncthreads();
  constexpr int WS = NP * 32 * 64;                                   // a wave's partial tile
#pragma unroll
  for (int pr = 0; pr < NP; ++pr)
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int nrow = (e & 3) + 8 * (e >> 2) + 4 * lh;
        s_red[wave * WS + (pr * 32 + nrow) * 64 + jt * 32 + lr] = acc[pr][jt][e];
      }
  __syncthreads();
  float* wsb = p.ws ? p.ws + (long)blockIdx.x * 16 * 64 * p.Co : nullptr;   // (single-pass launches only: the partial tile)
  for (int i = tid; i < WS; i += 256) {
    const int n = i >> 6, ci = i & 63, tap = (n & 31) >> 1, co = 2 * (gpair + (n >> 5)) + (n & 1);
    if (co >= p.Co) continue;
    const float v = s_red[i] + s_red[WS + i] + s_red[2 * WS + i] + s_red[3 * WS + i];
    const long o = ((long)tap * p.Ci + ci) * p.Co + co;
    if (wsb) wsb[o] = v * sc; else atomicAdd(&p.dw[o], v * sc);
  }
}

// ---------------------------------------------------------------------------------------------------------
// thin_up_mfma (bf16 in, K = 64 input channels -> N <= 4 output channels, MODE_UP): Head forward and Down1
// backward-data on the matrix cores (models/gans/dcgan_eqlr.py:29-47 Head, :75-82 Down's input gradient).
// One coarse input pixel produces 2 x 2 fine outputs x N channels = at most 16 values, each a dot product over a
// subset of the 3 x 3 input neighbourhood x 64 channels.  That is a GEMM with
//   M' = 16 rows (py, n, px),   K = (row offset dr, column offset dc, ci) = 9 x 64,   N' = coarse pixels,
// run as v_mfma_f32_16x16x32_bf16 (18 per 16 pixels).  The weight operand - zero where a parity does not use a
// neighbour, summed where the reflected rows of models/ops/common.py:9-20 fold two taps onto one source row,
// adjoint extras included - only depends on the boundary class of the image row (interior / first / last): a prep
// launch builds the 3 x 18 fragments from dg_tap1d into a device-global table and every wave keeps its class's 18
// fragments in 72 VGPRs.  The activation rows m-1, m, m+1 of a 64-pixel tile are staged in LDS with full-line
// loads (the layout thin_smalln uses) and read back as B fragments, one ds_read_b128 per MFMA.
// Output row m' = (py * N + n) * 2 + px, so a lane's accumulator pairs are the two column parities of one output
// row: planar fp32 outputs are written as float2, 128 contiguous bytes per 16 lanes.
// The table is one per device: launches that use it must be ordered on one stream (they are: the step is one stream).
typedef __attribute__((ext_vector_type(4))) float tw_f32x4;
__device__ __attribute__((aligned(16))) unsigned char g_up_frag[UP_FRAG_BYTES];  // [class][frag][64 lanes][16 B]
__device__ __attribute__((aligned(16))) unsigned char g_up_frag_lo[UP_FRAG_BYTES];   // (X2: the lo halves of the folded fp32 weights)

// class 0 interior (built at m = 1), 1 first row, 2 last row (thin_up_frag.h)
__global__ __launch_bounds__(256) void thin_up_prep_kernel(ConvP p) {
  const bf16* w = (const bf16*)p.w;
  up_frag_element(blockIdx.y, blockIdx.x * 256 + threadIdx.x, p.N, p.Hc, p.adj,
                  [&](int tap, int n, int ci) { return (float)w[(long)tap * p.w_st + (long)n * p.w_sn + ci]; }, g_up_frag);
}
// X2 (fp32 weights, split-bf16 input): both tables, blockIdx.z = 0 hi / 1 lo
__global__ __launch_bounds__(256) void thin_up_prep_x2_kernel(ConvP p) {
  const float* w = (const float*)p.w;
  auto ld = [&](int tap, int n, int ci) { return w[(long)tap * p.w_st + (long)n * p.w_sn + ci]; };
  if (blockIdx.z == 0) up_frag_element<false>(blockIdx.y, blockIdx.x * 256 + threadIdx.x, p.N, p.Hc, p.adj, ld, g_up_frag);
  else up_frag_element<true>(blockIdx.y, blockIdx.x * 256 + threadIdx.x, p.N, p.Hc, p.adj, ld, g_up_frag_lo);
}

#define TU_PX 64
#define TU_RS 8
// One block = (sample, segment of TU_RS image rows, 64-pixel column tile) and walks DOWN its rows with a ring of four
// staged input rows in LDS: output row m reads rows m-1, m, m+1 from the ring while row m+2 is in flight in registers
// (3 sixteen-byte pieces per thread) and is written into the slot nobody reads - ONE barrier per row, every input row
// fetched once per segment (10 rows for 8) instead of three times, and everything the epilogue needs from global memory
// (scale, bias) fetched once in front of the loop.  The first version - one block per image row, all three rows staged
// per tile - made one round trip to memory PER PIECE (a loop the compiler did not unroll: load, s_waitcnt vmcnt(0),
// ds_write) plus two per epilogue, ~5 us per tile; pipelining that design took it from 40 to 30 us, and it stayed bound
// by the 3x re-read.
// X2 (round 5, the fp32x3 mode's Head forward / Down1 backward-data): the input is DG_BF16X2 (a pixel = 128 bytes of hi + 128 bytes
// of lo), the weights fp32: rows are staged with both halves, the weight fragments exist twice (hi / lo of the folded fp32
// weights, thin_up_prep_x2_kernel) and every k-step is three matrix instructions, w_hi x_hi + w_hi x_lo + w_lo x_hi.
template <bool X2>
__global__ __launch_bounds__(256) void thin_up_mfma_kernel(ConvP p, int tiles_x, int nseg) {
  constexpr int PPP = X2 ? 16 : 8;                                   // 16-byte pieces per pixel
  constexpr int RB = PPP * 16 + 16;                                  // LDS pixel stride: the pixel's bytes + 16 B
  constexpr int RPX = TU_PX + 2, ROWB = RPX * RB;                    // a staged row: the tile's pixels + halo
  constexpr int NLD = (RPX * PPP + 255) / 256;                       // 16-byte pieces per thread and row (the last partial)
  __shared__ __attribute__((aligned(16))) unsigned char s_in[4 * ROWB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = p.N, Wc = p.Wc, Hc = p.Hc;
  // XCD-aware, bijective block remap (blocks id and id+8 share an XCD): the column tiles and row segments of one sample
  // - which share halo columns / rows - land on ONE XCD's L2
  const int nwg = gridDim.x, id = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int xt = logical % tiles_x, sg = (logical / tiles_x) % nseg, b = logical / (tiles_x * nseg);
  const int m0 = sg * TU_RS, m1 = m0 + TU_RS < Hc ? m0 + TU_RS : Hc;
  const char* in = (const char*)p.in + (long)b * p.in_sb * (X2 ? 4 : 2);
  const int col = lane & 15, kg = lane >> 4;
  const unsigned spb = (unsigned)p.in_sp * (X2 ? 4u : 2u);           // bytes per pixel
  unsigned goff[NLD], loff[NLD];                                     // piece u of a row: pixel tid / PPP + (256 / PPP) u, piece tid % PPP
#pragma unroll
  for (int u = 0; u < NLD; ++u) {
    const int px = tid / PPP + (256 / PPP) * u;
    int cc = xt * TU_PX - 1 + px;
    if (cc < 0) cc += Wc; else if (cc >= Wc) cc -= Wc;
    goff[u] = (unsigned)cc * spb + (tid % PPP) * 16;
    loff[u] = px * RB + (tid % PPP) * 16;
  }
  const bool last_ok = tid / PPP + (256 / PPP) * (NLD - 1) < RPX;
  auto fetch_row = [&](int r, tw_u32x4 (&st)[NLD]) __attribute__((always_inline)) {
    r = r < 0 ? 0 : (r >= Hc ? Hc - 1 : r);                          // rows outside the grid carry zero weights
    const char* row = in + (unsigned)(r * Wc) * spb;
#pragma unroll
    for (int u = 0; u < NLD; ++u)
      if (u < NLD - 1 || last_ok) st[u] = *(const tw_u32x4*)(row + goff[u]);
  };
  auto put_row = [&](int r, const tw_u32x4 (&st)[NLD]) __attribute__((always_inline)) {   // row r lives in slot (r + 1) & 3
    unsigned char* dst = s_in + ((r + 1) & 3) * ROWB;
#pragma unroll
    for (int u = 0; u < NLD; ++u)
      if (u < NLD - 1 || last_ok) *(tw_u32x4*)(dst + loff[u]) = st[u];
  };
  tw_u32x4 st[NLD];
  {
    tw_u32x4 sa[NLD], sb[NLD];
    fetch_row(m0 - 1, sa); fetch_row(m0, sb); fetch_row(m0 + 1, st);   // first: the block's longest round trip
    put_row(m0 - 1, sa); put_row(m0, sb); put_row(m0 + 1, st);
  }
  // epilogue constants of this lane's two output rows q = 2 kg + h  (q = py * N + n)
  float e_sc[2], e_bias[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int q = 2 * kg + h, n = q < 2 * N ? q % N : 0;
    e_sc[h] = p.nscale ? p.scale * p.nscale[n] : p.scale;
    e_bias[h] = p.bias ? p.bias[n % p.bias_mod] : 0.f;
  }
  const bool tsum = p.tanh_sum_parts != nullptr;                     // (DgConv.tanh_sum_parts: launcher-checked N == 1, fp32 out)
  float lsum = 0.f;
  int cls = -1;
  tw_bf16x8 fa[18], fal[X2 ? 18 : 1];
  // the caller's fragments (kept current with its shadows) or the ones thin_up_prep_kernel has just built
  const unsigned char* frags = (p.up_frag && !X2) ? (const unsigned char*)p.up_frag : g_up_frag;
  __syncthreads();
  const int xl = wave * 16 + col;                                    // this lane's pixel inside the tile
  const int x = xt * TU_PX + xl;
  for (int m = m0; m < m1; ++m) {
    const bool more = m + 1 < m1;
    if (more) fetch_row(m + 2, st);                                  // in flight during the MFMAs and stores below
    const int mcls = m == 0 ? 1 : (m == Hc - 1 ? 2 : 0);             // boundary class of the row: its weight fragments
    if (mcls != cls) {                                               // (block-uniform; at most twice per block)
      cls = mcls;
#pragma unroll
      for (int f = 0; f < 18; ++f) fa[f] = *(const tw_bf16x8*)(frags + ((cls * 18 + f) * 64 + lane) * 16);
      if constexpr (X2) {
#pragma unroll
        for (int f = 0; f < 18; ++f) fal[f] = *(const tw_bf16x8*)(g_up_frag_lo + ((cls * 18 + f) * 64 + lane) * 16);
      }
    }
    tw_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const unsigned char* rowp = s_in + ((m + rr) & 3) * ROWB + xl * RB + kg * 16;   // row m - 1 + rr
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const tw_bf16x8 b0 = *(const tw_bf16x8*)(rowp + d * RB);
        const tw_bf16x8 b1 = *(const tw_bf16x8*)(rowp + d * RB + 64);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(rr * 3 + d) * 2 + 0], b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(rr * 3 + d) * 2 + 1], b1, acc, 0, 0, 0);
        if constexpr (X2) {
          const tw_bf16x8 l0 = *(const tw_bf16x8*)(rowp + d * RB + 128);
          const tw_bf16x8 l1 = *(const tw_bf16x8*)(rowp + d * RB + 192);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(rr * 3 + d) * 2 + 0], l0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(rr * 3 + d) * 2 + 1], l1, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[(rr * 3 + d) * 2 + 0], b0, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fal[(rr * 3 + d) * 2 + 1], b1, acc, 0, 0, 0);
        }
      }
    }
    // D: column = pixel (lane & 15), rows 4 kg + j  ->  m' = 4 kg + j = (py * N + n) * 2 + px
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = 2 * kg + h;
      if (q >= 2 * N) continue;
      const int py = q / N, n = q % N;
      float v0 = acc[2 * h] * e_sc[h] + e_bias[h], v1 = acc[2 * h + 1] * e_sc[h] + e_bias[h];
      if (tsum) { v0 = dg_tanh(v0); v1 = dg_tanh(v1); lsum += v0 + v1; }   // the depth head: tanh + the image's sum (N == 1)
      const long o = (long)b * p.out_sb + ((long)(2 * m + py) * (2 * Wc) + 2 * x) * p.out_sp + (long)n * p.out_sn;
      if (p.out_dtype == DG_F32 && p.out_sp == 1) {
        *(float2*)((float*)p.out + o) = make_float2(v0, v1);
      } else {
        dg_st(p.out, o, p.out_dtype, v0);
        dg_st(p.out, o + p.out_sp, p.out_dtype, v1);
      }
    }
    if (more) put_row(m + 2, st);                                    // slot (m + 3) & 3 = the slot of row m - 2: not read this step
    __syncthreads();                                                 // row m + 2 visible; row m - 1's slot free for row m + 3
  }
  if (tsum) {                                                        // this workgroup's share of sample b's image sum: stored, not
    const float t = dg_block_sum(lsum, (float*)s_in);                // added (a fixed order at the reader: bit-reproducible)
    if (tid == 0) p.tanh_sum_parts[logical] = t;
  }
}

int dg_conv_up_mfma_supported(const ConvP* p) {
  if (p->mode != MODE_UP || !p->ring) return 0;
  const bool x2 = p->in_dtype == DG_BF16X2;        // split-bf16 input, fp32 weights, fp32 / bf16 output (the fp32x3 mode)
  if (x2 ? (p->w_dtype != DG_F32 || p->out_dtype == DG_BF16X2 || p->in_sp % 64 != 0 || p->in_sb % 64 != 0 || ((size_t)p->in & 255))
         : (p->in_dtype != DG_BF16 || p->w_dtype != DG_BF16)) return 0;
  if (p->K != 64 || p->N < 1 || p->N > 4 || p->Wc % TU_PX != 0 || p->Hc < 2) return 0;
  if (p->in_sk != 1 || p->w_sk != 1 || p->in_sp % 8 != 0 || p->in_sb % 8 != 0) return 0;
  if (p->epi != EPI_LINEAR || p->dbias) return 0;
  return 1;
}

// partial sums per sample the kernel stores for DgConv.tanh_sum_parts (its workgroups per sample), 0 where it does not take it
int dg_conv_up_mfma_sum_parts(const ConvP* p) {
  if (!dg_conv_up_mfma_supported(p) || p->N != 1 || p->out_dtype != DG_F32 || p->out_sp != 1) return 0;
  return (p->Wc / TU_PX) * ((p->Hc + TU_RS - 1) / TU_RS);
}

int dg_conv_up_mfma_launch(const ConvP* p, hipStream_t s) {
  if (!dg_conv_up_mfma_supported(p)) return DG_EUNSUPPORTED;
  if (p->tanh_sum_parts && !dg_conv_up_mfma_sum_parts(p)) return DG_EINVAL;
  if (p->up_frag && ((size_t)p->up_frag & 15)) return DG_EINVAL;
  const bool x2 = p->in_dtype == DG_BF16X2;
  if (x2) thin_up_prep_x2_kernel<<<dim3(UP_FRAG_BLOCKS, 3, 2), 256, 0, s>>>(*p);
  else if (!p->up_frag) thin_up_prep_kernel<<<dim3(UP_FRAG_BLOCKS, 3), 256, 0, s>>>(*p);
  // (a column-walker variant with an LDS-DMA row ring that fetched every input row once instead of three times measured
  //  within noise of this kernel on the step - 0.277 vs 0.282 ms for the family - and was removed in round 2)
  const int tiles_x = p->Wc / TU_PX, nseg = (p->Hc + TU_RS - 1) / TU_RS;
  const long blocks = (long)p->B * nseg * tiles_x;
  if (blocks >= (1L << 31)) return DG_EUNSUPPORTED;
  if (x2) thin_up_mfma_kernel<true><<<(unsigned)blocks, 256, 0, s>>>(*p, tiles_x, nseg);
  else thin_up_mfma_kernel<false><<<(unsigned)blocks, 256, 0, s>>>(*p, tiles_x, nseg);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------------------------------------
int dg_conv_s2_mfma_supported(const ConvP* p);
int dg_conv_s2_mfma_launch(const ConvP* p, hipStream_t s);

int dg_conv_thin_supported(const ConvP* p) {
  if (!p->ring || p->mode == MODE_GEMM) return 0;
  if (p->mode == MODE_S2)  // small K -> wide N
    return p->K <= 4 && p->N % 64 == 0 && p->Wc % SK_PX == 0 && !p->nscale && (!p->dbias || p->bias_mod >= p->N);
  // MODE_UP: wide K -> small N; weights = the T shadow [tap][n][k]; no mask / bias-grad epilogue
  if (p->N > 3 || p->Wc % SN_PX != 0 || p->in_sk != 1 || p->w_sk != 1) return 0;
  const int es = p->in_dtype == DG_BF16 ? 2 : 4;
  const bool x2 = p->in_dtype == DG_BF16X2;       // (split-bf16 input rows, fp32 weights and output)
  if (x2 && (p->K % 64 != 0 || p->in_sp % 64 != 0 || p->in_sb % 64 != 0 || ((size_t)p->in & 255) || p->out_dtype == DG_BF16X2)) return 0;
  if ((p->K * es) % 16 != 0 || p->w_dtype != (x2 ? DG_F32 : p->in_dtype)) return 0;
  if (p->epi != EPI_LINEAR || p->dbias) return 0;
  return 1;
}

// 1 = thin_s2_mfma, 2 = thin_up_mfma (matrix cores), 0 = the VALU kernels (or unsupported)
int dg_conv_thin_mfma_variant(const ConvP* p) {
  if (!dg_conv_thin_supported(p)) return 0;
  return dg_conv_s2_mfma_supported(p) ? 1 : (dg_conv_up_mfma_supported(p) ? 2 : 0);
}

int dg_conv_thin_launch(const ConvP* p, hipStream_t s) {
  if (!dg_conv_thin_supported(p)) return DG_EUNSUPPORTED;
  if (dg_conv_s2_mfma_supported(p)) return dg_conv_s2_mfma_launch(p, s);
  if (dg_conv_up_mfma_supported(p)) return dg_conv_up_mfma_launch(p, s);
  if (p->mode == MODE_S2) {
    const int tiles_x = p->Wc / SK_PX;
    const unsigned grid = (unsigned)((long)p->B * p->Hc);   // a block walks the tiles of one output row
    for (int nb = 0; nb < p->N; nb += 64) {
      if (p->K <= 2) thin_smallk_kernel<2><<<grid, 256, 0, s>>>(*p, tiles_x, nb);
      else thin_smallk_kernel<4><<<grid, 256, 0, s>>>(*p, tiles_x, nb);
    }
  } else {
    const unsigned grid = (unsigned)((long)p->B * p->Hc);
    const int es = p->in_dtype == DG_BF16 ? 2 : 4;
    const size_t lds = (size_t)3 * (SN_PX + 2) * (p->K * es + 16) + (es == 2 ? (size_t)16 * p->N * p->K * 2 : 0);
    if (lds > 64 * 1024) return DG_EUNSUPPORTED;
    if (p->in_dtype == DG_BF16) {
      if (p->N == 1) thin_smalln_kernel<bf16, 1><<<grid, 256, lds, s>>>(*p);
      else if (p->N == 2) thin_smalln_kernel<bf16, 2><<<grid, 256, lds, s>>>(*p);
      else thin_smalln_kernel<bf16, 3><<<grid, 256, lds, s>>>(*p);
    } else if (p->in_dtype == DG_BF16X2) {
      if (p->N == 1) thin_smalln_kernel<float, 1, true><<<grid, 256, lds, s>>>(*p);
      else if (p->N == 2) thin_smalln_kernel<float, 2, true><<<grid, 256, lds, s>>>(*p);
      else thin_smalln_kernel<float, 3, true><<<grid, 256, lds, s>>>(*p);
    } else {
      if (p->N == 1) thin_smalln_kernel<float, 1><<<grid, 256, lds, s>>>(*p);
      else if (p->N == 2) thin_smalln_kernel<float, 2><<<grid, 256, lds, s>>>(*p);
      else thin_smalln_kernel<float, 3><<<grid, 256, lds, s>>>(*p);
    }
  }
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_wgrad_thin_supported(const WgradP* p) {
  if (!p->ring) return 0;
  if (p->wmode == 0)
    return p->Ci <= 4 && p->Co % 64 == 0 &&
           (size_t)4 * (2 * p->Wc + 2) * (p->Ci <= 2 ? 2 : 4) * sizeof(float) <= 160 * 1024;
  if (p->wmode == 1) return p->Co <= 4 && p->Ci % 64 == 0 && (size_t)2 * 2 * p->Wc * 4 * sizeof(float) <= 160 * 1024;
  return 0;
}

// shapes the two matrix-core weight-gradient kernels take (Down1: 2 -> 64 channels; Head: 64 -> <= 4 channels, pixel-major
// gradient padded to 2 or 4 channels)
static size_t wgrad_down_mfma_lds(const WgradP* p) {
  size_t lds = (size_t)4 * (2 * p->Wc + 8) * 4 + 4 * 16 * 144;      // staged rows + gradient tiles ...
  if (lds < (size_t)4 * 32 * 64 * 4) lds = (size_t)4 * 32 * 64 * 4;  // ... aliased by the cross-wave reduction
  return lds;
}
static bool wgrad_down_mfma_ok(const WgradP* p) {
  if (!(p->wmode == 0 && p->a_dtype == DG_BF16 && p->g_dtype == DG_BF16 && p->Ci == 2 && p->Co == 64 && p->a_sc == 1 &&
        p->g_sc == 1 && p->a_sp == 2 && p->g_sp == 64 && p->Wc % 64 == 0 && p->Hc % WG_ROWS_PB == 0))
    return false;
  return wgrad_down_mfma_lds(p) <= 160 * 1024 && 2 * p->Wc <= 4096 && p->a_sb % 8 == 0 && ((size_t)p->a & 15) == 0 &&
         p->g_sb % 8 == 0 && ((size_t)p->g & 15) == 0;
}
static int wgrad_up_pairs(const WgradP* p) { return (p->Co > 2 && p->g_sp == 4) ? 2 : 1; }   // channel pairs per pass
static size_t wgrad_up_mfma_lds_np(const WgradP* p, int np) {
  size_t lds = (size_t)np * 32 * (p->Wc * 2 + 16) + 4 * 16 * 144;
  if (lds < (size_t)4 * np * 32 * 64 * 4) lds = (size_t)4 * np * 32 * 64 * 4;
  return lds;
}
static size_t wgrad_up_mfma_lds(const WgradP* p) {
  const size_t two = wgrad_up_mfma_lds_np(p, wgrad_up_pairs(p));
  return two <= 160 * 1024 ? two : wgrad_up_mfma_lds_np(p, 1);       // (very wide maps: one pair per pass)
}
static bool wgrad_up_mfma_ok(const WgradP* p) {
  if (!(p->wmode == 1 && p->a_dtype == DG_BF16 && p->g_dtype == DG_BF16 && p->Ci == 64 && p->a_sc == 1 &&
        p->a_sp == 64 && p->g_sc == 1 && (p->g_sp == 2 || p->g_sp == 4) && p->Co <= p->g_sp && p->g_sb % 2 == 0 &&
        p->Wc % 64 == 0 && p->Hc >= 2 && p->Hc % WGU_ROWS_PB == 0 && p->a_sb % 8 == 0 && ((size_t)p->a & 15) == 0))
    return false;
  return wgrad_up_mfma_lds(p) <= 160 * 1024;
}
// 1 = thin_wgrad_down_mfma, 2 = thin_wgrad_up_mfma (matrix cores), 0 = the VALU kernels (or unsupported)
int dg_wgrad_thin_mfma_variant(const WgradP* p) {
  if (!dg_wgrad_thin_supported(p)) return 0;
  return wgrad_down_mfma_ok(p) ? 1 : (wgrad_up_mfma_ok(p) ? 2 : 0);
}

static int wgrad_down_rows_pb(const WgradP* p) {
  // rows per block: every block ends in 2048 partial sums for the same 8 KB; 4 rows once 2 rows give >= 1024 blocks
  return ((long)p->B * p->Hc >= 2048 && p->Hc % 4 == 0) ? 4 : WG_ROWS_PB;
}
// Partial tiles (= blocks) of the launch when the kernel that runs has the workspace form (DgWgrad.ws: plain-store partials
// of 16 Ci Co floats each, summed by dg_wgrad_reduce): the two matrix-core kernels (single-pass launches) and, round 5, the
// VALU kernels of the fp32 modes (one partial tile per block of their <= 1024-block grid).  0: no such form.
static unsigned wgrad_valu_grid(const WgradP* p) {
  const long units = (long)p->B * p->Hc;
  return units < 1024 ? (unsigned)units : 1024u;
}
int dg_wgrad_thin_ws_splits(const WgradP* p) {
  if (!dg_wgrad_thin_supported(p)) return 0;
  const long units = (long)p->B * p->Hc;
  long nb = 0;
  if (wgrad_down_mfma_ok(p)) nb = units / wgrad_down_rows_pb(p);
  else if (wgrad_up_mfma_ok(p)) {
    const bool one_pass = p->Co <= 2 || (wgrad_up_pairs(p) == 2 && wgrad_up_mfma_lds(p) == wgrad_up_mfma_lds_np(p, 2));
    if (one_pass) nb = units / WGU_ROWS_PB;
  } else if (!p->g_mod) nb = wgrad_valu_grid(p);
  return nb > 0 && nb <= 65536 ? (int)nb : 0;
}

int dg_wgrad_thin_launch(const WgradP* p, hipStream_t s) {
  if (!dg_wgrad_thin_supported(p)) return DG_EUNSUPPORTED;
  if (p->ws && !dg_wgrad_thin_ws_splits(p)) return DG_EUNSUPPORTED;
  if (p->g_mod && !wgrad_down_mfma_ok(p)) return DG_EUNSUPPORTED;   // only thin_wgrad_down_mfma has the sample map
  const long units = (long)p->B * p->Hc;
  const unsigned grid = wgrad_valu_grid(p);
  if (wgrad_down_mfma_ok(p)) {
    const size_t lds = wgrad_down_mfma_lds(p);
    const int Wf = 2 * p->Wc;
    {
      const void* fn = Wf <= 256 ? (const void*)thin_wgrad_down_mfma_kernel<1>
                                 : (Wf <= 1024 ? (const void*)thin_wgrad_down_mfma_kernel<4> : (const void*)thin_wgrad_down_mfma_kernel<16>);
      if (lds > 64 * 1024)  // opt in to more than the default 64 KiB of dynamic LDS
        HIP_CHECK_RET(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int rows_pb = wgrad_down_rows_pb(p);
      const unsigned nb = (unsigned)(units / rows_pb);
      if (Wf <= 256) thin_wgrad_down_mfma_kernel<1><<<nb, 256, lds, s>>>(*p, rows_pb);
      else if (Wf <= 1024) thin_wgrad_down_mfma_kernel<4><<<nb, 256, lds, s>>>(*p, rows_pb);
      else thin_wgrad_down_mfma_kernel<16><<<nb, 256, lds, s>>>(*p, rows_pb);
      HIP_CHECK_RET(hipGetLastError());
      return DG_OK;
    }
  }
  if (wgrad_up_mfma_ok(p)) {
    const size_t lds = wgrad_up_mfma_lds(p);
    {
      const bool both = wgrad_up_pairs(p) == 2 && lds == wgrad_up_mfma_lds_np(p, 2);
      if (lds > 64 * 1024)
        HIP_CHECK_RET(hipFuncSetAttribute(both ? (const void*)thin_wgrad_up_mfma_kernel<2> : (const void*)thin_wgrad_up_mfma_kernel<1>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      if (both) {                                      // Co = 3 / 4 on the four-channel copy: both channel pairs in one pass
        thin_wgrad_up_mfma_kernel<2><<<(unsigned)(units / WGU_ROWS_PB), 256, lds, s>>>(*p, 0);
      } else {
        for (int gpair = 0; 2 * gpair < p->Co; ++gpair)  // one pass per pair of gradient channels
          thin_wgrad_up_mfma_kernel<1><<<(unsigned)(units / WGU_ROWS_PB), 256, lds, s>>>(*p, gpair);
      }
      HIP_CHECK_RET(hipGetLastError());
      return DG_OK;
    }
  }
  if (p->wmode == 0) {
    const size_t lds = (size_t)4 * (2 * p->Wc + 2) * (p->Ci <= 2 ? 2 : 4) * sizeof(float);
    if (lds > 64 * 1024)
      HIP_CHECK_RET(hipFuncSetAttribute(p->Ci <= 2 ? (const void*)thin_wgrad_down_kernel<2> : (const void*)thin_wgrad_down_kernel<4>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int cb = 0; cb < p->Co; cb += 64) {
      if (p->Ci <= 2) thin_wgrad_down_kernel<2><<<grid, 256, lds, s>>>(*p, cb);
      else thin_wgrad_down_kernel<4><<<grid, 256, lds, s>>>(*p, cb);
    }
  } else {
    const size_t lds = (size_t)2 * 2 * p->Wc * 4 * sizeof(float);
    if (lds > 64 * 1024)
      HIP_CHECK_RET(hipFuncSetAttribute((const void*)thin_wgrad_up_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int cb = 0; cb < p->Ci; cb += 64)
      thin_wgrad_up_kernel<4><<<grid, 256, lds, s>>>(*p, cb);
  }
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// ---------------------------------------------------------------------------------------------------------
// thin_s2_mfma (bf16): MODE_S2 with a CP-channel input (CP = 2: Down1 forward / R1 tangent, Head backward-data with
// <= 2 heads; CP = 4: Head backward-data with 3 heads, channels zero-padded) -> 64 output channels, on the matrix
// cores with NO LDS staging of the operands: in pixel-major / channel-minor memory the 4 kx taps x CP channels of
// one kernel row of one output pixel are 16 (CP=2) or 32 (CP=4) CONTIGUOUS bytes, i.e. exactly the 8 consecutive k
// an MFMA lane needs, so every A fragment is one 16-byte global load.  K = 4 ky x 4 kx x CP.
//   CP = 2: 2 MFMA k-steps, lane half h <-> ky = 2 s + h, j <-> (kx = j >> 1, c = j & 1)
//   CP = 4: 4 MFMA k-steps, step s <-> ky, lane half h <-> kx in {2h, 2h+1}, j <-> (kx = 2h + (j >> 2), c = j & 3)
// One wave = one 32-pixel x 64-channel tile at a time (a contiguous range of tiles per wave), weights live in 16 / 32 VGPRs
// for the whole kernel.  Round 4: the product is formed as W x A^T (weights as the A operand), so a lane ends up with ONE
// pixel and runs of four consecutive channels - scale / leaky-relu / saved-mask select / bf16 packing happen on the
// accumulator layout (3.5 VALU instructions per element, 594 -> ~250 per tile: the kernel was VALU-bound, SQ_INSTS_VALU in
// profiles/r04a_pmc_sq_summary.txt), the tile goes through a wave-private 4.5 KB LDS patch as eight 8-byte writes and comes
// back as the 4096 CONTIGUOUS bytes it occupies in the pixel-major output; the bias rides in the accumulators' start value.
// The tile's inputs are prefetched two tiles ahead with counted vmcnt waits (see `prefetch` below), which lets the output
// stores of two tiles stay in flight: Down1 forward at batch 32 / 64 30.8 / 47.2 -> 24.9 / 39.3 us, Head backward-data
// 34.7 -> 30.9 us (same box, eager step).
// MB: the saved 1-bit leaky-relu masks (DgConv.mask_out / mask_in) as a compile-time flavour - 0 none, 1 the EPI_LRELU launch
// also writes them, 2 the EPI_MASK launch reads them instead of aux (a run-time choice kept both forms' registers live:
// 134 -> 172 VGPRs, 3 -> 2 waves per SIMD, Down1 forward 40 -> 64 us)
template <int CP, int MB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(CP == 2 ? 3 : 2))) void thin_s2_mfma_kernel(ConvP p, int tiles_x, long ntiles) {
  constexpr int NS = CP == 2 ? 2 : 4;            // MFMA k-steps per kernel (without adjoint extras)
  __shared__ __attribute__((aligned(16))) unsigned char s_t[4][32 * 144];
  __shared__ float s_db[64];
  __shared__ float s_dbw[4][64];                 // bias-gradient partial rows, one per wave (summed in a fixed order)
  __shared__ __attribute__((aligned(16))) float s_binit[64];   // bias / scale: what the accumulators start from
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int Wf = 2 * p.Wc, Hf = 2 * p.Hc;
  const bf16* in = (const bf16*)p.in;
  const bf16* w = (const bf16*)p.w;              // [tap][n][k = c] with strides w_st, w_sn, 1 ; k < p.K real channels
  if (tid < 64) { s_db[tid] = 0.f; s_binit[tid] = p.bias ? p.bias[tid % p.bias_mod] / p.scale : 0.f; }
  __syncthreads();

  // B fragments: element j of step s for output channel n = jt*32 + lr
  auto wval = [&](int ky, int kx, int c, int n) -> bf16 {
    return c < p.K ? w[(long)(ky * 4 + kx) * p.w_st + (long)n * p.w_sn + c] : (bf16)0.f;
  };
  tw_bf16x8 fb[NS][2], fbx[2];                   // fbx: ky = 3 placed in lane half 0 (reflect-adjoint extra tap)
#pragma unroll
  for (int jt = 0; jt < 2; ++jt) {
    const int n = jt * 32 + lr;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ky = CP == 2 ? 2 * s + lh : s;
        const int kx = CP == 2 ? (j >> 1) : 2 * lh + (j >> 2);
        const int c = CP == 2 ? (j & 1) : (j & 3);
        fb[s][jt][j] = wval(ky, kx, c, n);
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) fbx[jt][j] = CP == 2 ? wval(3, j >> 1, j & 1, n) : (bf16)0.f;
  }

  float csum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) csum[e] = 0.f;
  // each wave owns a CONTIGUOUS range of tiles and walks (x tile, row, sample) with a carry chain: the three 64-bit
  // divisions of a strided decode cost more than the four MFMAs of a tile
  const int gw = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  const int nt = (int)ntiles, tq = nt / nw, tr = nt % nw;
  const int t0 = gw * tq + (gw < tr ? gw : tr), tcnt = tq + (gw < tr ? 1 : 0);
  int xt = t0 % tiles_x, Y = (t0 / tiles_x) % p.Hc, b = t0 / (tiles_x * p.Hc);
  unsigned char* my = s_t[wave];
  // one 16-byte window of input row r of sample bb starting at fine column c0 (circular): 4 (CP=2) or 2 (CP=4) pixels
  auto window_of = [&](int bb, int r, int c0) -> uint4 {
    constexpr int NPX = 8 / CP;
    const bf16* img = in + (long)bb * p.in_sb;
    if (c0 >= 0 && c0 + NPX <= Wf) return *(const uint4*)(img + ((long)r * Wf + c0) * CP);
    unsigned d[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {              // dword q = pixel q (CP=2) or half pixel (CP=4)
      int cc = c0 + (CP == 2 ? q : (q >> 1));
      if (cc < 0) cc += Wf; else if (cc >= Wf) cc -= Wf;
      d[q] = *(const unsigned*)(img + ((long)r * Wf + cc) * CP + (CP == 2 ? 0 : (q & 1) * 2));
    }
    return make_uint4(d[0], d[1], d[2], d[3]);
  };
  // Everything a tile loads - its NS fragments (16 bytes per lane each), the 64 mask bits of the lane's pixel, the per-sample
  // weight of the bias-gradient sums - is requested TWO tiles ahead, by inline asm with counted waits.  vmcnt retires in
  // order: a wait for loads issued ONE tile ahead also waits for the output stores of the tile before (issued in between),
  // i.e. a store has one tile's time (~1.5 us at three-four waves per SIMD) to be acknowledged, against 2-3 us under a
  // 3 TB/s write stream; two tiles ahead the stores of tile j only have to be complete at the top of tile j + 3.  (The
  // same pipeline written in C++ does not survive the compiler's own waitcnt insertion: register copies of prefetched
  // values and zero-initialisations of conditionally loaded registers each became an s_waitcnt vmcnt(0) per tile.)
  // A window that wraps around the row (first lane of a row's first tile, last lane of its last) is the clamped window
  // shifted by one pixel plus that pixel from the other end of the row: a second small load issued for EVERY tile, so that
  // the number of loads per tile - what the counted waits count - is a constant.
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  constexpr int NPXD = CP / 2;                   // dwords per pixel
  constexpr int NF = 2 * NS + 1 + (MB == 2 ? 1 : 0);          // loads per prefetch
  constexpr int NSTO = 4 + (MB == 1 ? 4 : 0);                 // stores per tile
  v4u pa[2][NS];                               // the two prefetch register sets (indexed by compile-time constants only)
  typedef typename std::conditional<CP == 2, unsigned, v2u>::type wrap_t;   // the pixel from the other end of the row
  wrap_t pw[2][NS];
  v2u pm[2];
  float prs[2];
  const int Wd = Wf * NPXD;                      // dwords per input row
  const float* rs_src = (p.dbias && p.rowscale) ? p.rowscale : (const float*)p.w;   // (always a valid address)
  const unsigned zero_off = 0u;
  auto out_base = [&](int xt_, int Y_, int b_) -> long { return (long)b_ * p.out_sb + ((long)Y_ * p.Wc + xt_ * 32) * 64; };
  auto prefetch = [&](int xt_, int Y_, int b_, auto buf_tag) __attribute__((always_inline)) {
    constexpr int I = decltype(buf_tag)::value;
    const char* img = (const char*)(in + (long)b_ * p.in_sb);   // wave-uniform
    const int d0 = (2 * (xt_ * 32 + lr) - 1 + (CP == 2 ? 0 : 2 * lh)) * NPXD;
    const int d0c = min(max(d0, 0), Wd - 4);
    const int dw = d0 < 0 ? Wd - NPXD : 0;       // the pixel from the other end of the row (lanes that do not wrap: any)
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int ky = CP == 2 ? 2 * s + lh : s;
      int r = 2 * Y_ - 1 + ky;
      if (!p.adj) { if (r < 0) r = -r; if (r >= Hf) r = 2 * Hf - 2 - r; }
      else r = min(max(r, 0), Hf - 1);           // (rows outside the image: zeroed where the fragment is used)
      const unsigned ro = (unsigned)(r * Wd);
      const unsigned oa = (ro + (unsigned)d0c) * 4u, ow = (ro + (unsigned)dw) * 4u;
      v4u ta;
      wrap_t tw;                                 // (whole asm outputs only: building a register pair from a loaded dword is a
                                                 //  v_mov of a register whose load has not landed)
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ta) : "v"(oa), "s"(img) : "memory");
      if (CP == 2) asm volatile("global_load_dword %0, %1, %2" : "=v"(tw) : "v"(ow), "s"(img) : "memory");
      else asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(tw) : "v"(ow), "s"(img) : "memory");
      pa[I][s] = ta;
      pw[I][s] = tw;
    }
    const float* rsp = rs_src + ((p.dbias && p.rowscale) ? b_ : 0);
    float trs;
    asm volatile("global_load_dword %0, %1, %2" : "=v"(trs) : "v"(zero_off), "s"(rsp) : "memory");
    prs[I] = trs;
    if (MB == 2) {
      const char* mb = (const char*)p.mask_in + (out_base(xt_, Y_, b_) >> 3);
      const unsigned om = (unsigned)(lr * 8);
      v2u tm;
      asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(tm) : "v"(om), "s"(mb) : "memory");
      pm[I] = tm;
    }
  };
  auto advance = [&](int& xt_, int& Y_, int& b_) __attribute__((always_inline)) {
    if (++xt_ == tiles_x) { xt_ = 0; if (++Y_ == p.Hc) { Y_ = 0; ++b_; } }
  };
  // Two register sets used alternately by a loop unrolled twice, each refilled in place right behind the tile's MFMAs (the
  // mask bits and the weight are copied out first).
  prs[0] = prs[1] = 1.f;
  pm[0] = pm[1] = v2u{0u, 0u};
#pragma unroll
  for (int s = 0; s < NS; ++s) { pa[0][s] = pa[1][s] = v4u{0u, 0u, 0u, 0u}; pw[0][s] = pw[1][s] = wrap_t{}; }
  int xt2 = xt, Y2 = Y, b2 = b;                  // the tile two ahead of the one being computed
  if (tcnt > 0) prefetch(xt2, Y2, b2, std::integral_constant<int, 0>{});
  advance(xt2, Y2, b2);
  if (tcnt > 1) prefetch(xt2, Y2, b2, std::integral_constant<int, 1>{});
  advance(xt2, Y2, b2);
  // Epilogue constants.  D = W x A^T: a lane holds ONE pixel (lr) and the channels jt*32 + 8g + 4lh + r (g, r = 0..3), i.e.
  // runs of four consecutive channels = 8 bytes of the pixel-major output row.  sqrt(2) is folded into the scale (lrelu
  // commutes with a positive factor: max(v, 0.2 v)), the bias into the accumulators' start value (s_binit), the saved mask
  // into ONE select per element (the lane's pixel owns 64 mask bits = one 8-byte load).
  float c_pos = p.scale * SQRT2, c_neg = p.scale * (LRELU_SLOPE * SQRT2);
  asm volatile("" : "+v"(c_pos), "+v"(c_neg));   // (opaque: else the select is made between constants + a second multiply)
  const float c_lin = p.epi == EPI_LRELU ? c_pos : p.scale;
  const float slope = p.epi == EPI_LRELU ? LRELU_SLOPE : 1.f;          // max(v, 1 v) = v: no select per element
  unsigned char* my_w = my + lr * 144 + lh * 8;                         // phase 1: + jt*64 + g*16
  unsigned char* my_r = my + (lane >> 3) * 144 + (lane & 7) * 16;       // phase 2: + u * 8 * 144
  auto tile = [&](auto buf_tag, const int ti) __attribute__((always_inline)) {
    constexpr int I = decltype(buf_tag)::value;
    const bool more = ti + 2 < tcnt;
    // this tile's loads have landed: everything issued behind them may still be in flight - the stores of the two tiles
    // before and the next tile's prefetch (tiles 0 and 1: what exists of that)
#define S2_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
    if (ti >= 2) { if (ti + 1 < tcnt) S2_WAIT(2 * NSTO + NF); else S2_WAIT(2 * NSTO); }
    else if (ti == 1) { if (tcnt > 2) S2_WAIT(NSTO + NF); else S2_WAIT(NSTO); }
    else { if (tcnt > 1) S2_WAIT(NF); else S2_WAIT(0); }
#undef S2_WAIT
    v4u a_cur[NS];
    wrap_t w_cur[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      v4u ta = pa[I][s];
      wrap_t tw = pw[I][s];
      asm volatile("" : "+v"(ta), "+v"(tw));     // (uses pinned behind the wait)
      a_cur[s] = ta; w_cur[s] = tw;
    }
    float rs = prs[I];
    v2u mword = pm[I];
    asm volatile("" : "+v"(rs), "+v"(mword));
    if (!(p.dbias && p.rowscale)) rs = 1.f;      // (the load is issued regardless, from a valid address: constant load count)
    if (xt == 0 || xt == tiles_x - 1) {          // (wave-uniform) the wrapped windows: shift in the pixel from the other end
      const bool lo = xt == 0 && lr == 0 && (CP == 2 || lh == 0);
      const bool hi = xt == tiles_x - 1 && lr == 31 && (CP == 2 || lh == 1);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const v4u a = a_cur[s];
        if constexpr (CP == 2) {
          const unsigned w = w_cur[s];
          if (lo) a_cur[s] = v4u{w, a.x, a.y, a.z};
          if (hi) a_cur[s] = v4u{a.y, a.z, a.w, w};
        } else {
          const v2u w = w_cur[s];
          if (lo) a_cur[s] = v4u{w.x, w.y, a.x, a.y};
          if (hi) a_cur[s] = v4u{a.z, a.w, w.x, w.y};
        }
      }
    }
    if (p.adj && (Y == 0 || Y == p.Hc - 1)) {    // (adjoint: kernel rows outside the image contribute nothing)
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        const int ky = CP == 2 ? 2 * s + lh : s;
        const int r = 2 * Y - 1 + ky;
        if (r < 0 || r >= Hf) a_cur[s] = v4u{0u, 0u, 0u, 0u};
      }
    }
    int nxt = xt, nY = Y, nb = b;
    advance(nxt, nY, nb);
    const long obase = out_base(xt, Y, b);       // the tile = 4096 contiguous bytes from here
    // the fallback form of EPI_MASK without saved bits reads the lane's 8 x 4 activations themselves (8-byte pieces, 128 B
    // apart between lanes, waited for inside the tile: slow, unused by the training step)
    uint2 araw[(MB != 2) ? 8 : 1];
    if (MB == 0 && p.epi == EPI_MASK) {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        araw[MB != 2 ? q : 0] = *(const uint2*)((const bf16*)p.aux + obase + lr * 64 + (q >> 2) * 32 + (q & 3) * 8 + lh * 4);
    }
    const int X = xt * 32 + lr;                  // this lane's output pixel (as fragment column)
    auto window = [&](int r, int c0) -> uint4 { return window_of(b, r, c0); };
    tw_f32x16 acc[2];                            // start value: bias / scale of the lane's channels
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 b4 = *(const float4*)&s_binit[jt * 32 + g * 8 + lh * 4];
        acc[jt][4 * g] = b4.x; acc[jt][4 * g + 1] = b4.y; acc[jt][4 * g + 2] = b4.z; acc[jt][4 * g + 3] = b4.w;
      }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const tw_bf16x8 fa = __builtin_bit_cast(tw_bf16x8, a_cur[s]);
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s][jt], fa, acc[jt], 0, 0, 0);
    }
    if (more) prefetch(xt2, Y2, b2, buf_tag);    // (in place: everything of this register set has been copied out or consumed)
    advance(xt2, Y2, b2);
    if (p.adj && CP == 2 && (Y == 1 || Y == p.Hc - 2)) {
      // reflect-adjoint extras (common.h dg_tap1d, MODE_S2/adj): row 0 through ky = 3 into Y == 1,
      // row Hf-1 through ky = 0 into Y == Hc-2; the tap sits in lane half 0, half 1 multiplies zeros
      const int c0 = 2 * X - 1;
      if (Y == 1) {
        uint4 a = make_uint4(0, 0, 0, 0);
        if (lh == 0) a = window(0, c0);
        const tw_bf16x8 fa = __builtin_bit_cast(tw_bf16x8, a);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fbx[jt], fa, acc[jt], 0, 0, 0);
      }
      if (Y == p.Hc - 2) {
        uint4 a = make_uint4(0, 0, 0, 0);
        if (lh == 0) a = window(Hf - 1, c0);
        const tw_bf16x8 fa = __builtin_bit_cast(tw_bf16x8, a);
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][jt], fa, acc[jt], 0, 0, 0);
      }
    }
    if (p.adj && CP == 4 && (Y == 1 || Y == p.Hc - 2)) {
      // CP = 4: a k-step is one kernel row, so the two reflect-adjoint extras are one more k-step each with the
      // weights of ky = 3 (row 0 into Y == 1) / ky = 0 (row Hf-1 into Y == Hc-2)
      const int c0 = 2 * X - 1 + 2 * lh;
      if (Y == 1) {
        const tw_bf16x8 fa = __builtin_bit_cast(tw_bf16x8, window(0, c0));
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[NS - 1][jt], fa, acc[jt], 0, 0, 0);
      }
      if (Y == p.Hc - 2) {
        const tw_bf16x8 fa = __builtin_bit_cast(tw_bf16x8, window(Hf - 1, c0));
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][jt], fa, acc[jt], 0, 0, 0);
      }
    }
    // ---- epilogue, phase 1: scale / activation / mask on the accumulator layout, 4 channels -> one 8-byte LDS write.
    //      The form is chosen once per tile (0 linear / leaky-relu through c_lin and slope, 1 saved bits, 2 the activation
    //      itself as mask source): a run-time test of p.epi per element became a scalar branch per element.
    const unsigned msh0 = mword.x >> (4 * lh), msh1 = mword.y >> (4 * lh);   // bit 8g + r = channel jt*32 + 8g + 4lh + r
    auto phase1 = [&](auto form_tag) __attribute__((always_inline)) {
      constexpr int FORM = decltype(form_tag)::value;
#pragma unroll
      for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float a = acc[jt][4 * g + r];
            if (FORM == 1) {
              int sel;
              float kf;
              asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(sel) : "v"(jt ? msh1 : msh0), "n"(8 * g + r));
              asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(kf) : "v"(sel), "v"(c_pos), "v"(c_neg));
              v[r] = a * kf;
            } else if (FORM == 2) {
              const uint2 aw = araw[MB != 2 ? jt * 4 + g : 0];
              const int w32 = (int)(r < 2 ? aw.x : aw.y);
              const bool posv = (r & 1) ? w32 > 0xffff : (short)w32 > 0;    // bf16 > 0 <=> its bits as a signed integer > 0
              v[r] = a * (posv ? c_pos : c_neg);
            } else {
              const float t = a * c_lin;
              v[r] = fmaxf(t, slope * t);
            }
          }
          uint2 pk;
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk.x) : "v"(v[0]), "v"(v[1]));
          asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk.y) : "v"(v[2]), "v"(v[3]));
          *(uint2*)(my_w + jt * 64 + g * 16) = pk;
        }
    };
    if (MB == 2) phase1(std::integral_constant<int, 1>{});
    else if (MB == 0 && p.epi == EPI_MASK) phase1(std::integral_constant<int, 2>{});
    else phase1(std::integral_constant<int, 0>{});
    // ---- phase 2: the patch read back as whole 16-byte pieces of pixel rows = the tile's 4096 contiguous output bytes
    unsigned char* otile = (unsigned char*)((bf16*)p.out + obase) + lane * 16;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint4 raw = *(const uint4*)(my_r + u * (8 * 144));
      if (MB == 1) {                             // (EPI_LRELU) the saved mask of these 8 channels, from the rounded values:
        const unsigned w4[4] = {raw.x, raw.y, raw.z, raw.w};   // halves clamped to [0, 1], then shifted together (conv_mfma_pp.hip)
        unsigned tb[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          asm("v_pk_max_i16 %0, %1, 0\n\tv_pk_min_u16 %0, %0, 1 op_sel_hi:[1,0]" : "=&v"(tb[e]) : "v"(w4[e]));
        const unsigned mm = tb[0] | (tb[1] << 2) | (tb[2] << 4) | (tb[3] << 6);
        ((unsigned char*)p.mask_out)[(obase >> 3) + lane + 64 * u] = (unsigned char)((mm & 0x55u) | ((mm >> 15) & 0xAAu));
      }
      if (p.dbias) {
        const bf16* v = (const bf16*)&raw;
#pragma unroll
        for (int e = 0; e < 8; ++e) csum[e] += rs * (float)v[e];
      }
      *(uint4*)(otile + u * 1024) = raw;
    }
    xt = nxt; Y = nY; b = nb;
  };
  for (int ti = 0; ti < tcnt; ti += 2) {
    tile(std::integral_constant<int, 0>{}, ti);
    if (ti + 1 < tcnt) tile(std::integral_constant<int, 1>{}, ti + 1);
  }
  if (p.dbias) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = csum[e];
      for (int d = 8; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
      if (lane < 8) s_dbw[tid >> 6][lane * 8 + e] = v;          // (round 5: per-wave rows instead of LDS atomics)
    }
    __syncthreads();
    if (tid < 64) s_db[tid] = (s_dbw[0][tid] + s_dbw[1][tid]) + (s_dbw[2][tid] + s_dbw[3][tid]);
    if (tid < 64) {
      if (p.dbias_part) {
        // one partial row per block in the caller's workspace, summed by dg_wgrad_reduce in a fixed order: bit-reproducible
        p.dbias_part[(long)blockIdx.x * 64 + tid] = s_db[tid];
      } else if (p.dbias_ws && gridDim.x > DG_DBIAS_SLOTS) {
        // 768 blocks adding one 256-byte row each into the SAME two lines retire one after the other (memory-side, ~10-20 ns
        // each: 16 us behind the Head backward-data).  Staged: block j adds into slot j % 32 of the caller's scratch (4 KB
        // apart, zero on entry), the last block to arrive at a slot (ticket behind the row, lane 0) folds it into dbias
        // and leaves the slot zero: 24 adds per slot, 32 per dbias line.
        const int slot = blockIdx.x % DG_DBIAS_SLOTS;
        const unsigned mine = (gridDim.x - slot + DG_DBIAS_SLOTS - 1) / DG_DBIAS_SLOTS;   // blocks that use this slot
        float* w = p.dbias_ws + slot * DG_DBIAS_SLOT_FLOATS;
        atomicAdd(&w[tid], s_db[tid]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (acknowledged memory-side; a fence would write back L2)
        unsigned t = 0;
        if (tid == 0) t = atomicAdd((unsigned*)&w[64], 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t == mine - 1) {
          atomicAdd(&p.dbias[tid % p.bias_mod], atomicExch(&w[tid], 0.f));
          if (tid == 0) atomicExch((unsigned*)&w[64], 0u);
        }
      } else {
        atomicAdd(&p.dbias[tid % p.bias_mod], s_db[tid]);
      }
    }
  }
}

int dg_conv_s2_mfma_supported(const ConvP* p) {
  if (p->mode != MODE_S2 || !p->ring || p->nscale) return 0;
  if (p->in_dtype != DG_BF16 || p->out_dtype != DG_BF16 || p->w_dtype != DG_BF16) return 0;
  if (p->N != 64 || p->K > 4 || p->Wc % 32 != 0 || p->Hc < 4) return 0;
  const int cp = p->in_sp;  // padded channel count of the input tensor
  if ((cp != 2 && cp != 4) || p->K > cp || p->in_sk != 1 || p->w_sk != 1 || p->out_sn != 1 || p->out_sp != 64) return 0;
  if (p->dbias && p->bias_mod != 64) return 0;
  if (p->bias && p->scale == 0.f) return 0;      // (the bias is the accumulators' start value bias / scale: round-4 advice)
  if (p->out_sb % 64 != 0) return 0;             // (a tile's output and its mask bits are addressed as whole 64-channel pixels)
  return 1;
}

#ifndef S2_CAP
#define S2_CAP 768
#endif
// grid of the thin matrix-core MODE_S2 kernel = the partial rows it writes to DgConv.dbias_part (DgConvPlan.dbias_rows)
int dg_conv_s2_mfma_blocks(const ConvP* p) {
  if (!dg_conv_s2_mfma_supported(p)) return 0;
  const long ntiles = (long)p->B * p->Hc * (p->Wc / 32);
  const long blocks = (ntiles + 3) / 4;
  return (int)(blocks > S2_CAP ? S2_CAP : blocks);
}

int dg_conv_s2_mfma_launch(const ConvP* p, hipStream_t s) {
  if (!dg_conv_s2_mfma_supported(p)) return DG_EUNSUPPORTED;
  const int tiles_x = p->Wc / 32;
  const long ntiles = (long)p->B * p->Hc * tiles_x;
  long blocks = (ntiles + 3) / 4;
  const long cap = S2_CAP;  // 3 blocks per CU.  Round 3 (164 VGPRs, 3 resident): 256 -> 50 us, 512 -> 34 us, 768 -> 31 us, 1024 -> 38 us
                         // for Down1 forward at batch 32; round 4 (116-120 VGPRs, 4 resident): 512 / 768 / 1024 -> 26.6 / 25.5 / 26.3 us
                         // (43.0 / 38.3 / 41.9 at batch 64) - the per-wave weight preload amortises over the tiles
  if (blocks > cap) blocks = cap;
  const int mb = (p->epi == EPI_LRELU && p->mask_out) ? 1 : ((p->epi == EPI_MASK && p->mask_in) ? 2 : 0);
#define DG_S2_LAUNCH(CP_, MB_) thin_s2_mfma_kernel<CP_, MB_><<<(unsigned)blocks, 256, 0, s>>>(*p, tiles_x, ntiles)
  if (p->in_sp == 2) { if (mb == 1) DG_S2_LAUNCH(2, 1); else if (mb == 2) DG_S2_LAUNCH(2, 2); else DG_S2_LAUNCH(2, 0); }
  else { if (mb == 1) DG_S2_LAUNCH(4, 1); else if (mb == 2) DG_S2_LAUNCH(4, 2); else DG_S2_LAUNCH(4, 0); }
#undef DG_S2_LAUNCH
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
