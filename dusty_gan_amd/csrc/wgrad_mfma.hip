// Weight-gradient GEMM on the matrix cores: dW[tap][ci][co] += scale * sum_b rs[b] sum_pixels A[src_a][ci] G[src_g][co]
// for Down (wmode 0), Up (wmode 1) and plain row-major operands (wmode 2: Proj, pixels = batch rows).
// Autograd counterpart in the reference: the weight gradients of nn.Conv2d / nn.ConvTranspose2d inside EqualLR
// (models/gans/dcgan_eqlr.py:9,24,80; models/ops/common.py:132-133), incl. the R1 double-backward terms.
//
// GEMM view: M = ci, N = co, K = pixels.  Both operands are stored pixel-major / channel-minor, i.e. the reduction
// index is the ROW of both tiles, so the MFMA fragments (8 consecutive k per lane) are read with the gfx950
// transposing LDS load ds_read_b64_tr_b16 (bf16) or plain ds_read_b32 (f32 MFMA takes one k per lane).
//   * workgroup = 4 waves (2 x 2) on a BM x BN tile of one tap; grid.y = tap, grid.z = K split over (sample,row)
//     units; partial sums are added with fp32 atomics whose wave-instruction writes two 128-B row segments (the
//     full-rate shape), or stored when there is a single split and nothing to accumulate onto.
//   * the per-sample weight (dy_b for the real batch) costs no second accumulator set: the running sum is
//     kept divided by the current sample's weight and rescaled by rs_old / rs_new when the sample changes.
#include "mfma_common.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// coarse pixels per K chunk: 64 (bf16) / 32 (fp32)
#define BKP_OF(es) ((es) == 2 ? 64 : 32)

// AdamEpi (common.h): optional optimizer epilogue, ADAM = true (Proj.weight in data-parallel runs)
// X3 (T = float only): fp32x3 - fp32 tiles in LDS, operands split into bf16 hi / lo in registers (mfma_common.h)
template <typename T, int BM, int BN, bool ADAM = false, bool X3 = false>
__global__ __launch_bounds__(256) void wgrad_mfma_kernel(WgradP p, int tiles_n, int accumulate, AdamEpi ad = AdamEpi{}) {
  static_assert(!X3 || sizeof(T) == 4, "fp32x3: fp32 operands");
  constexpr int ES = sizeof(T);
  constexpr int EPC = 16 / ES;
  constexpr int BKP = BKP_OF(ES);
  // LDS row strides (bytes).  bf16: +64 B so that the 4 rows x 2 column blocks x 4 column quads a 32-lane half
  // touches in one ds_read_b64_tr_b16 fall on distinct bank pairs (row stride = 16 dwords mod 64); the +16 B
  // padding of the first version cost 8-10 % of the wave cycles in SQ_LDS_BANK_CONFLICT.
  constexpr int RSA = BM * ES + (ES == 2 ? 64 : 16), RSG = BN * ES + (ES == 2 ? 64 : 16);
  constexpr int CPA = BM * ES / 16, CPG = BN * ES / 16;  // 16-B chunks per row
  constexpr int UA = BKP * CPA / 256, UG = BKP * CPG / 256;  // chunks per thread
  constexpr int TM = BM / 64, TN = BN / 64;
  __shared__ __attribute__((aligned(16))) unsigned char lds[BKP * (RSA + RSG)];
  unsigned char* ldsA = lds;
  unsigned char* ldsG = lds + BKP * RSA;

  const int tid = threadIdx.x;
  const int ct = blockIdx.x % tiles_n, mt = blockIdx.x / tiles_n;
  const int ci0 = mt * BM, co0 = ct * BN;
  const int tap = blockIdx.y, ky = tap >> 2, kx = tap & 3;
  const long units = (long)p.B * p.Hc;
  const long u0 = units * blockIdx.z / gridDim.z, u1 = units * (blockIdx.z + 1) / gridDim.z;
  const int chunks_per_row = (p.Wc + BKP - 1) / BKP;
  const long nchunks = (u1 - u0) * chunks_per_row;
  const int Wa = p.wmode == 0 ? 2 * p.Wc : p.Wc;
  const int Wg = p.wmode == 1 ? 2 * p.Wc : p.Wc;
  const T* A = (const T*)p.a;
  const T* G = (const T*)p.g;

  // two register sets: the tiles of chunk j travel in set j & 1 and are issued TWO chunks before they are stored to
  // LDS (the kernel is bound by the exposed global-load latency of its short chunks, not by bytes or instructions)
  uint4 ra[2][UA], rg[2][UG];
  auto load_tiles = [&](uint4 (&ra)[UA], uint4 (&rg)[UG], long u, int xc) {
    const int b = (int)(u / p.Hc), m = (int)(u % p.Hc);
    int rowa = 0, rowg = 0;
    if (p.wmode != 2) dg_wgrad1d(p.wmode, 0, m, p.Hc, ky, rowa, rowg);
    const T* ab = A + (long)b * p.a_sb + ci0;
    const T* gb = G + (long)b * p.g_sb + co0;
#pragma unroll
    for (int v = 0; v < UA; ++v) {
      const int c = tid + 256 * v;
      const int row = c / CPA, part = c % CPA;
      const int x = xc * BKP + row;
      int ca = x, cg = x;
      if (p.wmode != 2 && x < p.Wc) dg_wgrad1d(p.wmode, 1, x, p.Wc, kx, ca, cg);
      ra[v] = x < p.Wc ? *(const uint4*)(ab + ((long)rowa * Wa + ca) * p.a_sp + part * EPC) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int v = 0; v < UG; ++v) {
      const int c = tid + 256 * v;
      const int row = c / CPG, part = c % CPG;
      const int x = xc * BKP + row;
      int ca = x, cg = x;
      if (p.wmode != 2 && x < p.Wc) dg_wgrad1d(p.wmode, 1, x, p.Wc, kx, ca, cg);
      rg[v] = x < p.Wc ? *(const uint4*)(gb + ((long)rowg * Wg + cg) * p.g_sp + part * EPC) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tiles = [&](const uint4 (&ra)[UA], const uint4 (&rg)[UG]) {
#pragma unroll
    for (int v = 0; v < UA; ++v) {
      const int c = tid + 256 * v;
      *(uint4*)(ldsA + (c / CPA) * RSA + (c % CPA) * 16) = ra[v];
    }
#pragma unroll
    for (int v = 0; v < UG; ++v) {
      const int c = tid + 256 * v;
      *(uint4*)(ldsG + (c / CPG) * RSG + (c % CPG) * 16) = rg[v];
    }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float cur_rs = 1.f;
  if (p.rowscale && nchunks > 0) {
    cur_rs = p.rowscale[(int)(u0 / p.Hc)];
    if (fabsf(cur_rs) < 1e-30f) cur_rs = cur_rs < 0.f ? -1e-30f : 1e-30f;
  }
  if (nchunks > 0) {
    long lu = u0;                              // loader position: the next chunk to ISSUE
    int lxc = 0;
    long issued = 0;
    auto issue = [&](uint4 (&qa)[UA], uint4 (&qg)[UG]) {
      if (issued < nchunks) {
        load_tiles(qa, qg, lu, lxc);
        if (++lxc == chunks_per_row) { lxc = 0; ++lu; }
      }
      ++issued;
    };
    long u = u0;                               // compute position
    int xc = 0;
    issue(ra[0], rg[0]);                       // chunk 0
    store_tiles(ra[0], rg[0]);
    issue(ra[1], rg[1]);                       // chunk 1
    issue(ra[0], rg[0]);                       // chunk 2
    __syncthreads();
    auto one_chunk = [&](long s, uint4 (&na)[UA], uint4 (&ng)[UG]) {
      // LDS holds chunk s; (na, ng) hold chunk s+1 (in flight or landed) and are refilled with chunk s+3
      const int cur_b = (int)(u / p.Hc);
      if (++xc == chunks_per_row) { xc = 0; ++u; }
      const bool more = s + 1 < nchunks;
      if constexpr (ES == 2) {
        // lane -> (row block q, column quad pp) inside its 16-lane group; group -> (k half, column block)
        const int g16 = lane >> 4, i16 = lane & 15;
        const int kh = g16 >> 1, cb = g16 & 1, q = i16 >> 2, pp = i16 & 3;
#pragma unroll
        for (int kq = 0; kq < BKP / 16; ++kq) {
          bf16x8 fa[TM], fg[TN];
          const int prow = kq * 16 + 8 * kh + q;
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            const unsigned char* ptr = ldsA + prow * RSA + (wm * (BM / 2) + i * 32 + 16 * cb + 4 * pp) * 2;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(ptr));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(ptr + 4 * RSA));
            fa[i] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const unsigned char* ptr = ldsG + prow * RSG + (wn * (BN / 2) + j * 32 + 16 * cb + 4 * pp) * 2;
            const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(ptr));
            const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bf16x4 __attribute__((address_space(3)))*)(ptr + 4 * RSG));
            fg[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fg[j], acc[i][j], 0, 0, 0);
        }
      } else if constexpr (X3) {
        // four pixel-row pairs per step: this lane's rows 2 (k8 + m) + lh, m = 0..3, are the 4 k values of its fragments
#pragma unroll 2
        for (int k8 = 0; k8 < BKP / 2; k8 += 4) {
          SplitA sa[TM];
          SplitB sg[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) {
            f32x4 f;
#pragma unroll
            for (int m = 0; m < 4; ++m) f[m] = *(const float*)(ldsA + (2 * (k8 + m) + lh) * RSA + (wm * (BM / 2) + i * 32 + lr) * 4);
            sa[i] = split_a(f);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            f32x4 f;
#pragma unroll
            for (int m = 0; m < 4; ++m) f[m] = *(const float*)(ldsG + (2 * (k8 + m) + lh) * RSG + (wn * (BN / 2) + j * 32 + lr) * 4);
            sg[j] = split_b(f);
          }
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) mma_tile_x3(sa[i], sg[j], acc[i][j]);
        }
      } else {
#pragma unroll 4
        for (int k2 = 0; k2 < BKP / 2; ++k2) {
          float fa[TM], fg[TN];
          const int prow = 2 * k2 + lh;
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[i] = *(const float*)(ldsA + prow * RSA + (wm * (BM / 2) + i * 32 + lr) * 4);
#pragma unroll
          for (int j = 0; j < TN; ++j) fg[j] = *(const float*)(ldsG + prow * RSG + (wn * (BN / 2) + j * 32 + lr) * 4);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i], fg[j], acc[i][j], 0, 0, 0);
        }
      }
      if (p.rowscale && more) {
        const int next_b = (int)(u / p.Hc);
        if (next_b != cur_b) {
          float rn = p.rowscale[next_b];
          if (fabsf(rn) < 1e-30f) rn = rn < 0.f ? -1e-30f : 1e-30f;
          const float ratio = cur_rs / rn;
          cur_rs = rn;
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] *= ratio;
        }
      }
      __syncthreads();
      if (more) {
        store_tiles(na, ng);
        issue(na, ng);
        __syncthreads();
      }
    };
    for (long s = 0; s < nchunks; s += 2) {
      one_chunk(s, ra[1], rg[1]);
      if (s + 1 < nchunks) one_chunk(s + 1, ra[0], rg[0]);
    }
  }

  // D layout: col = lane & 31 (co), row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) (ci)
  if constexpr (ADAM) {
    // The accumulator layout gives each lane 4-byte pieces of 16 different rows; Adam streams five arrays, so the
    // tile is first transposed through LDS (64 rows at a time, the rows of one wave-row) and then walked with 16-byte
    // accesses, 32 lanes covering one 512-byte row segment.
    static_assert(BM == 128 && BN == 128 && sizeof(lds) >= 64 * BN * 4, "ADAM epilogue: 128 x 128 tiles");
    const float t = (float)(*ad.stepp + 1ull);
    const float inv_sqrt_bc2 = rsqrtf(1.f - powf(ad.b2, t));
    const float gmul = p.scale * cur_rs * ad.gscale;
    float* stage = (float*)lds;
    __syncthreads();  // every wave is done reading the operand tiles
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (wm == half) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
              stage[(i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * BN + wn * (BN / 2) + j * 32 + lr] = acc[i][j][e];
      }
      __syncthreads();
      const int c4 = tid & 31;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) {
        const int row = rr * 8 + (tid >> 5);
        const int ci = ci0 + half * 64 + row, co = co0 + c4 * 4;
        if (ci < p.Ci && co < p.Co) {
          const long i4 = ((long)ci * p.Co + co) / 4;
          const float4 g4 = *(const float4*)(stage + row * BN + c4 * 4);
          // (every stream of the optimizer is touched once per step: non-temporal, so the 1.7 GB do not sweep L2 / the
          //  Infinity Cache of what the next kernels read)
          typedef __attribute__((ext_vector_type(4))) float nt4;
          const nt4 v4 = __builtin_nontemporal_load((const nt4*)ad.v + i4);
          const nt4 p4 = __builtin_nontemporal_load((const nt4*)ad.p + i4);
          const nt4 e4 = ad.ema ? __builtin_nontemporal_load((const nt4*)ad.ema + i4) : nt4{0.f, 0.f, 0.f, 0.f};
          float gv[4] = {g4.x, g4.y, g4.z, g4.w}, pv[4] = {p4.x, p4.y, p4.z, p4.w};
          float vv[4] = {v4.x, v4.y, v4.z, v4.w}, ev[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float g = gv[k] * gmul;
            const float vi = ad.b2 * vv[k] + (1.f - ad.b2) * g * g;
            vv[k] = vi;
            pv[k] = pv[k] - ad.lr * (g / (sqrtf(vi) * inv_sqrt_bc2 + ad.eps));
            ev[k] = ad.ema_decay * ev[k] + (1.f - ad.ema_decay) * pv[k];
          }
          __builtin_nontemporal_store(nt4{pv[0], pv[1], pv[2], pv[3]}, (nt4*)ad.p + i4);
          __builtin_nontemporal_store(nt4{vv[0], vv[1], vv[2], vv[3]}, (nt4*)ad.v + i4);
          if (ad.ema) __builtin_nontemporal_store(nt4{ev[0], ev[1], ev[2], ev[3]}, (nt4*)ad.ema + i4);
          if (ad.shadow) {
            if (ad.shadow_bf16) {
              const bf16 h[4] = {(bf16)pv[0], (bf16)pv[1], (bf16)pv[2], (bf16)pv[3]};
              uint2 pk;
              pk.x = (unsigned)__builtin_bit_cast(unsigned short, h[0]) | ((unsigned)__builtin_bit_cast(unsigned short, h[1]) << 16);
              pk.y = (unsigned)__builtin_bit_cast(unsigned short, h[2]) | ((unsigned)__builtin_bit_cast(unsigned short, h[3]) << 16);
              ((uint2*)ad.shadow)[i4] = pk;
            } else {
              ((float4*)ad.shadow)[i4] = make_float4(pv[0], pv[1], pv[2], pv[3]);
            }
          }
        }
      }
      __syncthreads();
    }
    return;
  }
  // workspace form (DgWgrad.ws, round 6): split z stores its partial tile plainly at ws[z][tap][ci][co]; dg_wgrad_reduce sums
  // the splits in index order - bit-reproducible, where the atomics below sum in arrival order
  float* dw = p.ws ? p.ws + ((long)blockIdx.z * gridDim.y + tap) * p.Ci * p.Co : p.dw + (long)tap * p.Ci * p.Co;
  if (p.ws) accumulate = 0;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int ci = ci0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int co = co0 + wn * (BN / 2) + j * 32 + lr;
        if (ci < p.Ci && co < p.Co) {
          const float v = acc[i][j][e] * (p.scale * cur_rs);
          float* dst = dw + (long)ci * p.Co + co;
          if (accumulate) atomicAdd(dst, v);
          else *dst = v;
        }
      }
}

// K split of a launch: none without `accumulate` (plain stores), else towards >= 2 workgroups per CU
static long k_split(const WgradP* p, int BM, int BN, int accumulate) {
  const int tiles_m = (p->Ci + BM - 1) / BM, tiles_n = (p->Co + BN - 1) / BN;
  const int ntap = p->wmode == 2 ? 1 : 16;
  const long units = (long)p->B * p->Hc;
  long tiles = (long)tiles_m * tiles_n * ntap;
  long split = 1;
  if (accumulate) {
    split = (512 + tiles - 1) / tiles;
    if (split > units) split = units;
    if (split < 1) split = 1;
  }
  return split;
}

// > 1: the launch splits K that many ways and takes DgWgrad.ws (splits x numel floats) instead of adding with atomics
int dg_wgrad_mfma_ws_splits(const WgradP* p, int accumulate) {
  if (!dg_wgrad_mfma_supported(p) || p->a_dtype == DG_BF16X2) return 0;
  const long s = k_split(p, p->Ci % 128 == 0 ? 128 : 64, p->Co % 128 == 0 ? 128 : 64, accumulate);
  return s > 1 ? (int)s : 0;
}

template <typename T, int BM, int BN, bool X3 = false>
static int launch_cfg(const WgradP* p, int accumulate, hipStream_t stream) {
  const int tiles_m = (p->Ci + BM - 1) / BM, tiles_n = (p->Co + BN - 1) / BN;
  const int ntap = p->wmode == 2 ? 1 : 16;
  const long split = k_split(p, BM, BN, accumulate);
  if (p->ws && split <= 1) return DG_EUNSUPPORTED;   // (a workspace is only taken by a launch that splits: dg_wgrad_plan says so)
  dim3 grid((unsigned)(tiles_m * tiles_n), (unsigned)ntap, (unsigned)split);
  wgrad_mfma_kernel<T, BM, BN, false, X3><<<grid, 256, 0, stream>>>(*p, tiles_n, accumulate);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// Proj.weight's gradient GEMM with the optimizer folded into its epilogue (optim.hip dg_adam_proj_fused dispatches here
// when the batch is too large for its LDS-resident VALU kernel, i.e. the all-gathered global batch of multi-GPU runs)
int dg_wgrad_mfma_adam_launch(const WgradP* p, const AdamEpi* ad, hipStream_t stream, int fp32x3) {
  if (p->wmode != 2 || p->a_dtype != p->g_dtype || (p->a_dtype != DG_BF16 && p->a_dtype != DG_F32) || p->a_sc != 1 || p->g_sc != 1)
    return DG_EUNSUPPORTED;
  if (p->Ci % 128 != 0 || p->Co % 128 != 0 || p->rowscale) return DG_EUNSUPPORTED;
  const int tiles_m = p->Ci / 128, tiles_n = p->Co / 128;
  dim3 grid((unsigned)(tiles_m * tiles_n), 1, 1);  // no K split: every tile sees the whole batch
  // fp32 operands (round 6: the parity-class modes' Proj.weight, which ran gradient GEMM + reduce + plain optimizer): on the
  // fp32 matrix instructions, or - fp32x3 - split into bf16 hi / lo in registers like every other GEMM of that mode
  if (p->a_dtype == DG_BF16) wgrad_mfma_kernel<bf16, 128, 128, true><<<grid, 256, 0, stream>>>(*p, tiles_n, 0, *ad);
  else if (fp32x3) wgrad_mfma_kernel<float, 128, 128, true, true><<<grid, 256, 0, stream>>>(*p, tiles_n, 0, *ad);
  else wgrad_mfma_kernel<float, 128, 128, true, false><<<grid, 256, 0, stream>>>(*p, tiles_n, 0, *ad);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_wgrad_mfma_dma_supported(const WgradP* p);

extern "C" int dg_wgrad_mfma_supported(const WgradP* p) {
  if (p->a_dtype == DG_BF16X2 || p->g_dtype == DG_BF16X2) return dg_wgrad_mfma_dma_supported(p);   // (that kernel or nothing)
  if (p->a_dtype != p->g_dtype) return 0;
  if (p->a_sc != 1 || p->g_sc != 1) return 0;
  if (p->Ci % 64 != 0 || p->Co % 64 != 0) return 0;
  if (p->wmode != 2) {
    if (!p->ring) return 0;
    if (p->Wc % BKP_OF(p->a_dtype == DG_BF16 ? 2 : 4) != 0) return 0;
  }
  return 1;
}

// accumulate = 1: dw += (atomics, K split over workgroups); accumulate = 0: dw = (single pass, plain stores)
int dg_wgrad_mfma_launch(const WgradP* p, int accumulate, hipStream_t stream, int fp32x3) {
  if (!dg_wgrad_mfma_supported(p) || p->a_dtype == DG_BF16X2) return DG_EUNSUPPORTED;
  const bool m128 = p->Ci % 128 == 0, n128 = p->Co % 128 == 0;
  if (p->a_dtype == DG_BF16) {
    if (m128 && n128) return launch_cfg<bf16, 128, 128>(p, accumulate, stream);
    if (m128) return launch_cfg<bf16, 128, 64>(p, accumulate, stream);
    if (n128) return launch_cfg<bf16, 64, 128>(p, accumulate, stream);
    return launch_cfg<bf16, 64, 64>(p, accumulate, stream);
  }
  if (fp32x3) {
    if (m128 && n128) return launch_cfg<float, 128, 128, true>(p, accumulate, stream);
    if (m128) return launch_cfg<float, 128, 64, true>(p, accumulate, stream);
    if (n128) return launch_cfg<float, 64, 128, true>(p, accumulate, stream);
    return launch_cfg<float, 64, 64, true>(p, accumulate, stream);
  }
  if (m128 && n128) return launch_cfg<float, 128, 128>(p, accumulate, stream);
  if (m128) return launch_cfg<float, 128, 64>(p, accumulate, stream);
  if (n128) return launch_cfg<float, 64, 128>(p, accumulate, stream);
  return launch_cfg<float, 64, 64>(p, accumulate, stream);
}
