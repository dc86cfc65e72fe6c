// Implicit-GEMM conv on the matrix cores (gfx950 MFMA), one kernel for every "fat" conv-like pass of the step:
//   Down forward / R1 tangent pass (MODE_S2, adj=0), Up forward (MODE_UP, adj=0),
//   Down backward-data (MODE_UP, adj=1), Up backward-data (MODE_S2, adj=1), and the Proj GEMM (MODE_GEMM).
// Reference ops: models/gans/dcgan_eqlr.py:6-26,75-82 with models/ops/common.py Pad/EqualLR/FusedLeakyReLU fused:
// padding is index arithmetic in the tile loader, the EqualLR scale, bias, leaky-relu (or its derivative mask) and
// the bias-gradient column sums live in the epilogue.
//
// GEMM view: M = output pixels, N = output channels, K = taps x input channels.
//   * one workgroup = 256 threads = 4 waves (2 x 2), tile BM x BN with BM,BN in {64,128}
//   * an M-tile lies inside ONE output row (and one column parity in MODE_UP) so the tap list - including the
//     reflect-adjoint extra taps - is workgroup-uniform (scalar control flow, no per-lane predication)
//   * K step = 128 bytes of channels (64 bf16 / 32 f32) per tap; tiles are staged global -> registers -> LDS with
//     the next tile's loads in flight during the MFMAs; LDS rows are padded to 144 B (conflict-free ds_read_b128)
//   * bf16: v_mfma_f32_32x32x16_bf16; f32: v_mfma_f32_32x32x2_f32 (exact fp32, the parity mode)
//   * blockIdx is remapped so each XCD (private L2) walks a contiguous range of M-tiles across all their N-tiles
#include "common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define RS 144  // LDS row stride in bytes (128 B of K + 16 B pad)

__device__ __forceinline__ void mma_tile(const bf16*, const uint4& a, const uint4& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&a, *(const bf16x8*)&b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_tile(const float*, const uint4& a, const uint4& b, f32x16& acc) {
  const f32x4 fa = *(const f32x4*)&a, fb = *(const f32x4*)&b;
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], acc, 0, 0, 0);
}

template <typename T, int BM, int BN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvP p, int tiles_n, int tiles_x) {
  constexpr int ES = sizeof(T);
  constexpr int BK = 128 / ES;   // channels per K step
  constexpr int EPC = 16 / ES;   // elements per 16-byte chunk
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int AU = BM / 32, BU = BN / 32;
  __shared__ __attribute__((aligned(16))) unsigned char lds[(BM + BN) * RS + 512];
  unsigned char* ldsA = lds;
  unsigned char* ldsB = lds + BM * RS;
  int* s_tap = (int*)(lds + (BM + BN) * RS);  // [0] = ntaps, then {src_row, col_offset, weight_tap} x ntaps

  // ---- XCD-aware, bijective block remap (blocks id and id+8 share an XCD)
  const int nwg = gridDim.x, id = blockIdx.x;
  const int q8 = nwg >> 3, r8 = nwg & 7, xcd = id & 7;
  const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (id >> 3);
  const int nt = logical % tiles_n;
  int mt = logical / tiles_n;
  const int nb0 = nt * BN;

  // ---- decode the M tile (all scalar)
  int b = 0, Y = 0, px = 0, n0 = 0, Ws = 1, cmul = 1, Wo = 1;
  if (p.mode == MODE_S2) {
    const int xt = mt % tiles_x; mt /= tiles_x;
    Y = mt % p.Hc; b = mt / p.Hc;
    n0 = xt * BM; Ws = 2 * p.Wc; cmul = 2; Wo = p.Wc;
  } else if (p.mode == MODE_UP) {
    const int xt = mt % tiles_x; mt /= tiles_x;
    px = mt & 1; mt >>= 1;
    Y = mt % (2 * p.Hc); b = mt / (2 * p.Hc);
    n0 = xt * BM; Ws = p.Wc; cmul = 1; Wo = 2 * p.Wc;
  } else {
    n0 = mt * BM;  // first batch row of this tile
  }

  const int tid = threadIdx.x;
  if (tid == 0) {
    int nt_ = 0;
    if (p.mode == MODE_GEMM) {
      s_tap[1] = 0; s_tap[2] = 0; s_tap[3] = 0; nt_ = 1;
    } else {
      for (int i = 0; i < 6; ++i) {
        int r, ky;
        if (!dg_tap1d(p.mode, p.adj, 0, Y, p.Hc, i, r, ky)) continue;
        const int nw = p.mode == MODE_S2 ? 4 : 2;
        for (int j = 0; j < nw; ++j) {
          int coff, kx;
          if (p.mode == MODE_S2) { coff = j - 1; kx = j; }
          else if (px == 0) { coff = j == 0 ? 0 : -1; kx = j == 0 ? 1 : 3; }
          else { coff = j == 0 ? 1 : 0; kx = j == 0 ? 0 : 2; }
          s_tap[1 + 3 * nt_ + 0] = r;
          s_tap[1 + 3 * nt_ + 1] = coff;
          s_tap[1 + 3 * nt_ + 2] = ky * 4 + kx;
          ++nt_;
        }
      }
    }
    s_tap[0] = nt_;
  }
  __syncthreads();
  const int ntaps = s_tap[0];
  const int KC = p.K / BK;
  const int nsteps = ntaps * KC;

  const int part = tid & 7, rbase = tid >> 3;
  const T* in = (const T*)p.in;
  const T* w = (const T*)p.w;

  uint4 ra[AU], rb[BU];
  auto load_tiles = [&](int tq, int kc) {
    const int r = s_tap[1 + 3 * tq + 0], coff = s_tap[1 + 3 * tq + 1], wt = s_tap[1 + 3 * tq + 2];
    const long koff = (long)kc * BK + part * EPC;
#pragma unroll
    for (int u = 0; u < AU; ++u) {
      const int row = rbase + 32 * u;
      const T* src;
      bool ok = true;
      if (p.mode == MODE_GEMM) {
        const int br = n0 + row;
        ok = br < p.B;
        src = in + (long)(ok ? br : 0) * p.in_sb + koff;
      } else {
        int c = cmul * (n0 + row) + coff;
        if (c < 0) c += Ws; else if (c >= Ws) c -= Ws;
        src = in + (long)b * p.in_sb + ((long)r * Ws + c) * p.in_sp + koff;
      }
      ra[u] = ok ? *(const uint4*)src : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < BU; ++u) {
      const int n = nb0 + rbase + 32 * u;
      const bool ok = n < p.N;
      const T* src = w + (long)wt * p.w_st + (long)(ok ? n : 0) * p.w_sn + koff;
      rb[u] = ok ? *(const uint4*)src : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tiles = [&]() {
#pragma unroll
    for (int u = 0; u < AU; ++u) *(uint4*)(ldsA + (rbase + 32 * u) * RS + part * 16) = ra[u];
#pragma unroll
    for (int u = 0; u < BU; ++u) *(uint4*)(ldsB + (rbase + 32 * u) * RS + part * 16) = rb[u];
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const unsigned char* fa = ldsA + (wm * (BM / 2) + lr) * RS + lh * 16;
  const unsigned char* fb = ldsB + (wn * (BN / 2) + lr) * RS + lh * 16;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  int tq = 0, kc = 0;
  load_tiles(0, 0);
  store_tiles();
  __syncthreads();
  for (int s = 0; s < nsteps; ++s) {
    if (++kc == KC) { kc = 0; ++tq; }
    const bool more = s + 1 < nsteps;
    if (more) load_tiles(tq, kc);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      uint4 a[TM], bb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *(const uint4*)(fa + i * 32 * RS + ks * 32);
#pragma unroll
      for (int j = 0; j < TN; ++j) bb[j] = *(const uint4*)(fb + j * 32 * RS + ks * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) mma_tile((const T*)nullptr, a[i], bb[j], acc[i][j]);
    }
    __syncthreads();
    if (more) {
      store_tiles();
      __syncthreads();
    }
  }

  // ---- epilogue: C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* s_db = (float*)lds;  // BN floats, LDS is free now
  const bool want_db = p.dbias != nullptr;
  if (want_db) {
    if (tid < BN) s_db[tid] = 0.f;
    __syncthreads();
  }
  T* out = (T*)p.out;
  const T* aux = (const T*)p.aux;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = nb0 + wn * (BN / 2) + j * 32 + lr;
    const bool nok = n < p.N;
    const float bias = (p.bias && nok) ? p.bias[n % p.bias_mod] : 0.f;
    float colsum = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        long o;
        bool ok = nok;
        if (p.mode == MODE_GEMM) {
          const int br = n0 + row;
          ok = ok && br < p.B;
          o = (long)br * p.out_sb + (long)n * p.out_sn;
        } else {
          const int X = p.mode == MODE_S2 ? n0 + row : 2 * (n0 + row) + px;
          o = (long)b * p.out_sb + ((long)Y * Wo + X) * p.out_sp + (long)n * p.out_sn;
        }
        if (ok) {
          const float auxv = p.epi == EPI_MASK ? (float)aux[o] : 0.f;
          const float v = dg_epilogue(acc[i][j][e], p.scale, p.epi, bias, auxv);
          out[o] = (T)v;
          colsum += v;
        }
      }
    }
    if (want_db) {
      colsum += __shfl_xor(colsum, 32, 64);
      if (lh == 0 && nok) atomicAdd(&s_db[wn * (BN / 2) + j * 32 + lr], colsum);
    }
  }
  if (want_db) {
    __syncthreads();
    if (tid < BN && nb0 + tid < p.N) {
      const float rs = p.rowscale ? p.rowscale[b] : 1.f;
      atomicAdd(&p.dbias[(nb0 + tid) % p.bias_mod], s_db[tid] * rs);
    }
  }
}

template <typename T, int BM, int BN>
static int launch_cfg(const ConvP* p, hipStream_t stream) {
  const int tiles_n = (p->N + BN - 1) / BN;
  int tiles_x = 1;
  long tiles_m;
  if (p->mode == MODE_S2) { tiles_x = p->Wc / BM; tiles_m = (long)p->B * p->Hc * tiles_x; }
  else if (p->mode == MODE_UP) { tiles_x = p->Wc / BM; tiles_m = (long)p->B * 2 * p->Hc * 2 * tiles_x; }
  else tiles_m = (p->B + BM - 1) / BM;
  const long nwg = tiles_m * tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffL) return DG_EINVAL;
  conv_mfma_kernel<T, BM, BN><<<(unsigned)nwg, 256, 0, stream>>>(*p, tiles_n, tiles_x);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// Shapes this kernel takes; everything else goes to the direct kernel (dg_conv in api.hip decides).
extern "C" int dg_conv_mfma_supported(const ConvP* p) {
  const int es = p->in_dtype == DG_BF16 ? 2 : 4;
  const int BK = 128 / es;
  if (p->in_dtype != p->out_dtype || p->in_dtype != p->w_dtype) return 0;
  if (p->K % BK != 0 || p->N % 64 != 0) return 0;
  if (p->in_sk != 1 || p->w_sk != 1) return 0;
  if (p->mode == MODE_GEMM) return p->dbias == nullptr;
  if (!p->ring) return 0;
  if (p->Wc % 64 != 0) return 0;
  if (p->dbias && p->bias_mod < p->N) return 0;
  return 1;
}

int dg_conv_mfma_launch(const ConvP* p, hipStream_t stream) {
  if (!dg_conv_mfma_supported(p)) return DG_EUNSUPPORTED;
  const bool m128 = p->mode == MODE_GEMM ? false : (p->Wc % 128 == 0);
  const bool n128 = p->N % 128 == 0;
  if (p->in_dtype == DG_BF16) {
    if (m128 && n128) return launch_cfg<bf16, 128, 128>(p, stream);
    if (m128) return launch_cfg<bf16, 128, 64>(p, stream);
    if (n128) return launch_cfg<bf16, 64, 128>(p, stream);
    return launch_cfg<bf16, 64, 64>(p, stream);
  }
  if (m128 && n128) return launch_cfg<float, 128, 128>(p, stream);
  if (m128) return launch_cfg<float, 128, 64>(p, stream);
  if (n128) return launch_cfg<float, 64, 128>(p, stream);
  return launch_cfg<float, 64, 64>(p, stream);
}
