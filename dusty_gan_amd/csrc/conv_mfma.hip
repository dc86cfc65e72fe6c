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
//   * K step = SB bytes of channels per tap (128: 64 bf16 / 32 f32), moved global -> LDS by LDS-DMA
//     (global_load_lds_dwordx4, no VGPR staging, no ds_write) into an NS-stage ring (default 128 B x 2), ONE raw
//     s_barrier per K step, counted vmcnt, the next step's DMA in flight during the MFMAs.  LDS rows are unpadded;
//     bank conflicts are removed by an XOR swizzle applied on the per-lane SOURCE address (lane -> chunk ^ ((row >> 1)
//     & 7)) and undone on the fragment reads (conflict-free for every ds_read_b128 16-lane group; SQ_LDS_BANK_CONFLICT
//     = 0).  Measured before the change: with every global load removed the register-staged kernel only went
//     394 -> 444 TFLOP/s, i.e. it was bound by ds_write_b128 (79 B/clk/CU) + two barriers per step.
//   * fragment reads are inline-asm ds_read_b128 with counted lgkmcnt: a compiler-visible LDS read behind an in-flight
//     LDS-DMA makes hipcc drain vmcnt(0) first, which serialised DMA and MFMA in the first DMA version; the tap
//     iterator is scalar code on the issue side only (no LDS tap table) for the same reason.
//   * s_setprio(1) around the MFMA groups: +2.5-3 % (same-box A/B, DG_CONV_DBG=8 switches it off).
//   * bf16: v_mfma_f32_32x32x16_bf16; f32: v_mfma_f32_32x32x2_f32 (exact fp32, the parity mode)
//   * blockIdx is remapped so each XCD (private L2) walks a contiguous range of M-tiles across all their N-tiles
#include "mfma_common.h"

#ifndef DG_CONV_SETPRIO
#define DG_CONV_SETPRIO 1
#endif

// X3 (T = float only): the fp32x3 form - fp32 storage and LDS images, operands split into bf16 hi / lo in registers, bf16
// matrix instructions (mfma_common.h)
template <typename T, int BM, int BN, int SB, int NS, bool X3 = false>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvP p, int tiles_n, int tiles_x, int dbg) {
  static_assert(!X3 || sizeof(T) == 4, "fp32x3: fp32 operands");
  // SB = bytes of K per row per pipeline stage (64 or 128), NS = LDS stages (prefetch distance NS-1)
  constexpr int ES = sizeof(T);
  constexpr bool PRIO = DG_CONV_SETPRIO;
  constexpr int BK = SB / ES;          // channels per K step
  constexpr int EPC = 16 / ES;         // elements per 16-byte chunk
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int RPI = 1024 / SB;       // tile rows moved by one DMA wave-instruction (1 KiB)
  constexpr int CPR = SB / 16;         // 16-byte chunks per row
  constexpr int KS = SB / 32;          // MFMA k-steps (16 bf16 / 8 f32) per stage
  constexpr int STAGE = (BM + BN) * SB;
  constexpr int IA = BM / RPI / 4, IB = BN / RPI / 4;  // DMA instructions per wave per tile (A, B)
  constexpr int IPT = IA + IB;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NS * STAGE];

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
  // ---- tap enumeration is scalar code on the ISSUE side only (no LDS table: a compiler-visible ds_read inside the
  //      loop would make hipcc drain the LDS-DMA queue with vmcnt(0) before it)
  const int nW = p.mode == MODE_S2 ? 4 : (p.mode == MODE_UP ? 2 : 1);
  int nH = 1;
  if (p.mode != MODE_GEMM) {
    nH = 0;
    for (int i = 0; i < 6; ++i) {
      int r, ky;
      nH += dg_tap1d(p.mode, p.adj, 0, Y, p.Hc, i, r, ky) ? 1 : 0;
    }
  }
  const int KC = p.K / BK;
  const int nsteps = nH * nW * KC;

  const T* in = (const T*)p.in;
  const T* w = (const T*)p.w;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int lrow = lane / CPR, pos = lane % CPR;

  // row swizzle: chunk position of global chunk g in tile row `row` is g ^ swz(row)
  auto swz = [](int row) { return SB == 128 ? ((row >> 1) & 7) : ((row >> 2) & 3); };

  // issue-side iterator over (H tap, W tap, k chunk)
  int it_i = -1, it_j = 0, it_kc = 0, it_r = 0, it_ky = 0;
  auto next_htap = [&]() {
    if (p.mode == MODE_GEMM) { it_i = 0; return; }
    for (++it_i; it_i < 6; ++it_i)
      if (dg_tap1d(p.mode, p.adj, 0, Y, p.Hc, it_i, it_r, it_ky)) break;
  };
  next_htap();
  auto advance = [&]() {
    if (++it_kc == KC) {
      it_kc = 0;
      if (++it_j == nW) { it_j = 0; next_htap(); }
    }
  };

  // ---------------- DMA of one K-step tile into stage `st`: wave `wave` moves row groups wave, wave+4, ...
  auto issue_dma = [&](int st) {
    int coff = 0, kx = 0;
    if (p.mode == MODE_S2) { coff = it_j - 1; kx = it_j; }
    else if (p.mode == MODE_UP) {
      if (px == 0) { coff = it_j == 0 ? 0 : -1; kx = it_j == 0 ? 1 : 3; }
      else { coff = it_j == 0 ? 1 : 0; kx = it_j == 0 ? 0 : 2; }
    }
    const int wt = p.mode == MODE_GEMM ? 0 : it_ky * 4 + kx;
    unsigned char* base = lds + st * STAGE;
    const long k0 = (long)it_kc * BK;
#pragma unroll
    for (int u = 0; u < IA; ++u) {
      const int grp = wave + 4 * u;
      const int row = grp * RPI + lrow;
      const long koff = k0 + (pos ^ swz(row)) * EPC;
      const T* src;
      if (p.mode == MODE_GEMM) {
        int br = n0 + row;
        if (br >= p.B) br = p.B - 1;  // rows past the batch: duplicate data, dropped in the epilogue
        src = in + (long)br * p.in_sb + koff;
      } else {
        int c = cmul * (n0 + row) + coff;
        if (c < 0) c += Ws; else if (c >= Ws) c -= Ws;
        src = in + (long)b * p.in_sb + ((long)it_r * Ws + c) * p.in_sp + koff;
      }
      dma16(src, base + grp * 1024);
    }
#pragma unroll
    for (int u = 0; u < IB; ++u) {
      const int grp = wave + 4 * u;
      const int row = grp * RPI + lrow;
      int n = nb0 + row;
      if (n >= p.N) n = p.N - 1;
      dma16(w + (long)wt * p.w_st + (long)n * p.w_sn + k0 + (pos ^ swz(row)) * EPC, base + BM * SB + grp * 1024);
    }
  };

  const int wm = wave >> 1, wn = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int sw = swz(lr);  // tile-row bases of the fragments are multiples of 32, so swz(row) == swz(lr)
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  unsigned fragA[KS], fragB[KS];  // LDS byte addresses (stage 0) of this lane's A / B fragment per MFMA k-step
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int off = ((2 * ks + lh) ^ sw) * 16;
    fragA[ks] = lds0 + (wm * (BM / 2) + lr) * SB + off;
    fragB[ks] = lds0 + (BM + wn * (BN / 2) + lr) * SB + off;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Fragment reads are inline asm: a compiler-visible LDS load after an LDS-DMA makes hipcc emit s_waitcnt vmcnt(0)
  // (it cannot prove the DMA writes another stage), which would serialise DMA and MFMA.  The reads of k-step ks+1
  // are issued before the MFMAs of k-step ks; counted lgkmcnt + sched_barrier keep the MFMAs behind their operands.
  i32x4 fa[2][TM], fb[2][TN];
  auto read_frags = [&](int set, unsigned a_addr, unsigned b_addr) {
    LDS_READ128(fa[set][0], a_addr, 0);
    if constexpr (TM == 2) LDS_READ128(fa[set][1], a_addr, 32 * SB);
    LDS_READ128(fb[set][0], b_addr, 0);
    if constexpr (TN == 2) LDS_READ128(fb[set][1], b_addr, 32 * SB);
  };
  auto compute = [&](unsigned st_off) {
    read_frags(0, fragA[0] + st_off, fragB[0] + st_off);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
        read_frags((ks + 1) & 1, fragA[ks + 1] + st_off, fragB[ks + 1] + st_off);
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(TM + TN) : "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
      if (PRIO && !(dbg & 8)) __builtin_amdgcn_s_setprio(1);
      if constexpr (X3) {
        SplitA sa[TM];
        SplitB sb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) sa[i] = split_a(__builtin_bit_cast(f32x4, fa[ks & 1][i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) sb[j] = split_b(__builtin_bit_cast(f32x4, fb[ks & 1][j]));
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) mma_tile_x3(sa[i], sb[j], acc[i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            mma_tile((const T*)nullptr, fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
      }
      if (PRIO && !(dbg & 8)) __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- NS-stage ring, prefetch distance D = NS-1, ONE raw barrier per K step, counted vmcnt (never a drain in
  //      steady state).  Tile s lives in stage s % NS.  At the top of step s: wait until tile s has landed
  //      (at most D-1 younger tiles still in flight), barrier (everyone's share landed AND everyone finished
  //      reading stage (s-1) % NS), then refill that stage with tile s+D, then compute tile s.
  constexpr int D = NS - 1;
  int issued = 0;
  for (; issued < D && issued < nsteps; ++issued) {
    issue_dma(issued % NS);
    advance();
  }
  for (int s = 0; s < nsteps; ++s) {
    const int younger = issued - 1 - s;  // tiles issued after tile s (0 .. D-1), uniform
    if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPT) : "memory");
    else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPT) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * IPT) : "memory");
    __builtin_amdgcn_s_barrier();
    if (issued < nsteps) {
      if (!(dbg & 1)) issue_dma(issued % NS);
      advance();
      ++issued;
    }
    if (!(dbg & 2)) compute((unsigned)((s % NS) * STAGE));
  }
  __syncthreads();

  if (dbg & 4) return;
  // ---- epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5),
  //      i.e. a lane owns ONE channel of 16 pixels: storing from there is 2-byte scattered traffic (measured: 35 %
  //      of the kernel).  Instead each half of the tile (BM/2 pixel rows) is transposed through LDS and written
  //      as 16-byte chunks, whole channel rows per pixel; the leaky-relu mask source (aux) is read the same way and
  //      the bias-gradient column sums are folded with two shuffles + one LDS atomic per 16 lanes.
  constexpr int ORS = BN * ES + 16;            // LDS row stride of the staged output tile (bytes)
  constexpr int OCH = BN * ES / 16;            // 16-byte chunks per output row
  constexpr bool ONEPASS = BM * ORS + BN * 4 <= NS * STAGE;  // whole tile staged at once when LDS allows
  constexpr int PASSES = ONEPASS ? 1 : 2;
  constexpr int PROWS = BM / PASSES;           // tile rows per pass
  constexpr int CPT = PROWS * OCH / 256;       // 16-byte chunks per thread per pass
  static_assert(PROWS * ORS + BN * 4 <= NS * STAGE, "epilogue staging does not fit the pipeline's LDS");
  unsigned char* s_out = lds;
  // bias-gradient column sums: one row per wave (each wave writes every channel of the tile once), summed in wave order, then
  // across the workgroups as two-word fixed point through DgConv.dbias_ws (common.h) - the same bits whatever the arrival
  // order (round 6: an LDS float atomic per wave and a global one per workgroup)
  __shared__ float s_dbw[4][BN];
  const bool want_db = p.dbias != nullptr;
  T* out = (T*)p.out;
  float csum[EPC];
#pragma unroll
  for (int e = 0; e < EPC; ++e) csum[e] = 0.f;
  const int part = tid % OCH;                  // this thread's 16-byte column chunk (fixed: 256 % OCH == 0)
  const bool nok = nb0 + part * EPC < p.N;
  auto out_off = [&](int trow, bool& ok) -> long {  // element offset of (tile row, this thread's chunk)
    ok = nok;
    if (p.mode == MODE_GEMM) {
      const int br = n0 + trow;
      ok = ok && br < p.B;
      return (long)br * p.out_sb + nb0 + part * EPC;
    }
    const int X = p.mode == MODE_S2 ? n0 + trow : 2 * (n0 + trow) + px;
    return (long)b * p.out_sb + ((long)Y * Wo + X) * p.out_sp + nb0 + part * EPC;
  };
  for (int pass = 0; pass < PASSES; ++pass) {
    // leaky-relu mask source of this pass: issue the loads now, they land while the tile is being staged
    uint4 araw[CPT];
    if (p.epi == EPI_MASK) {
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        bool ok;
        const long o = out_off(pass * PROWS + (tid + 256 * u) / OCH, ok);
        araw[u] = ok ? *(const uint4*)((const T*)p.aux + o) : make_uint4(0, 0, 0, 0);
      }
    }
    __syncthreads();                           // previous pass fully written out / main loop done
    if (ONEPASS || wm == pass) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = wn * (BN / 2) + j * 32 + lr;
        const int n = nb0 + col;
        const float bias = (p.bias && n < p.N) ? p.bias[n % p.bias_mod] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int row = (ONEPASS ? wm * (BM / 2) : 0) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            float v = acc[i][j][e] * p.scale + bias;
            if (p.epi == EPI_LRELU) v = (v > 0.f ? v : LRELU_SLOPE * v) * SQRT2;
            *(T*)(s_out + row * ORS + col * ES) = (T)v;
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int row = (tid + 256 * u) / OCH;   // (tid + 256 u) % OCH == part
      bool ok;
      const long o = out_off(pass * PROWS + row, ok);
      if (!ok) continue;
      uint4 raw = *(const uint4*)(s_out + row * ORS + part * 16);
      T* v = (T*)&raw;
      if (p.epi == EPI_MASK) {
        const T* av = (const T*)&araw[u];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
          const float f = (float)v[e] * ((float)av[e] > 0.f ? SQRT2 : LRELU_SLOPE * SQRT2);
          v[e] = (T)f;
        }
      }
      if (want_db) {
#pragma unroll
        for (int e = 0; e < EPC; ++e) csum[e] += (float)v[e];
      }
      *(uint4*)(out + o) = raw;
    }
  }
  if (want_db) {
    // threads with equal `part` inside a wave are OCH lanes apart (OCH = 8 or 16 for bf16, 16 or 32 for f32)
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
      float v = csum[e];
      for (int d = OCH; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
      if (lane < OCH) s_dbw[tid >> 6][part * EPC + e] = v;
    }
    __syncthreads();
    const bool ws = dg_dbias_ws_ok(p.dbias_ws, p.bias_mod);
    if (tid < BN && nb0 + tid < p.N) {
      const float rs = p.rowscale ? p.rowscale[b] : 1.f;
      const float v = (((s_dbw[0][tid] + s_dbw[1][tid]) + s_dbw[2][tid]) + s_dbw[3][tid]) * rs;
      const int ch = (nb0 + tid) % p.bias_mod;
      long long hi, lo;
      if (ws && dg_fix2(v, hi, lo)) dg_dbias_ws_add(p.dbias_ws, ch, hi, lo);
      else atomicAdd(&p.dbias[ch], v);
    }
    if (ws) dg_dbias_ws_finish(p.dbias_ws, p.bias_mod, p.dbias);
  }
}

template <typename T, int BM, int BN, bool X3 = false>
static int launch_cfg(const ConvP* p, hipStream_t stream, DgConvPlan* plan) {
  const int tiles_n = (p->N + BN - 1) / BN;
  int tiles_x = 1;
  long tiles_m;
  if (p->mode == MODE_S2) { tiles_x = p->Wc / BM; tiles_m = (long)p->B * p->Hc * tiles_x; }
  else if (p->mode == MODE_UP) { tiles_x = p->Wc / BM; tiles_m = (long)p->B * 2 * p->Hc * 2 * tiles_x; }
  else tiles_m = (p->B + BM - 1) / BM;
  const long nwg = tiles_m * tiles_n;
  if (nwg <= 0 || nwg > 0x7fffffffL) return DG_EINVAL;
  if (plan) {
    plan->family = 2; plan->bm = BM; plan->bn = BN; plan->tiles = (int)nwg; plan->workgroups = (int)nwg;
    plan->tiles_per_wg = 1;
    return DG_OK;
  }
  // 128-byte stages x 2 (64-byte stages x 3 / x 4 measured slower on every layer and are not instantiated)
  conv_mfma_kernel<T, BM, BN, 128, 2, X3><<<(unsigned)nwg, 256, 0, stream>>>(*p, tiles_n, tiles_x, 0);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// Shapes this kernel takes; everything else goes to the thin / direct kernels (dg_conv in api.hip decides).
int dg_conv_mfma_pp_launch(const ConvP* p, hipStream_t stream, int min_tiles, int wg_cap, DgConvPlan* plan, int dual);

extern "C" int dg_conv_mfma_supported(const ConvP* p) {
  if (p->in_dtype == DG_BF16X2 || p->out_dtype == DG_BF16X2 || p->w_dtype == DG_BF16X2) {
    // split-bf16 pairs: the ping-pong kernel or nothing (its launcher describes the launch without making it)
    DgConvPlan pl;
    return dg_conv_mfma_pp_launch(p, nullptr, 1, 0, &pl, 1) == DG_OK;
  }
  const int es = p->in_dtype == DG_BF16 ? 2 : 4;
  const int BK = 128 / es;  // K must be a multiple of the largest stage (128 B of channels)
  if (p->in_dtype != p->out_dtype || p->in_dtype != p->w_dtype) return 0;
  if (p->K % BK != 0 || p->N % 64 != 0) return 0;
  if (p->in_sk != 1 || p->w_sk != 1 || p->out_sn != 1) return 0;
  if (p->mode == MODE_GEMM) return p->dbias == nullptr;
  if (!p->ring) return 0;
  if (p->Wc % 64 != 0) return 0;
  if (p->dbias && p->bias_mod < p->N) return 0;
  return 1;
}

int dg_conv_mfma_persist_launch_bf16(const ConvP* p, hipStream_t stream, int auto_rule, int wg_cap, DgConvPlan* plan);
int dg_conv_mfma_persist_launch_f32(const ConvP* p, hipStream_t stream, int auto_rule, int wg_cap, DgConvPlan* plan);

int dg_conv_mfma_pp_launch(const ConvP* p, hipStream_t stream, int min_tiles, int wg_cap, DgConvPlan* plan, int dual);

// plan != NULL: fill it with what would be launched and launch nothing.
// Auto rule: bf16 layers with >= 256 tiles of 256 pixels -> ping-pong persistent kernel (conv_mfma_pp.hip, family 5);
// layers the lock-step persistent kernel tiles with every CU busy -> that one (conv_mfma_persist_impl.h, family 4: fp32,
// 128 x 128 tiles); everything else -> one tile per workgroup (below, family 2).
int dg_conv_mfma_launch(const ConvP* p, hipStream_t stream, int wg_cap, DgConvPlan* plan, int fp32x3) {
  if (!dg_conv_mfma_supported(p)) return DG_EUNSUPPORTED;
  if (p->in_dtype == DG_BF16X2) return dg_conv_mfma_pp_launch(p, stream, 1, wg_cap, plan, 1);
  if (p->in_dtype == DG_BF16) {
    const int rc = dg_conv_mfma_pp_launch(p, stream, 256, wg_cap, plan, 1);
    if (rc != DG_EUNSUPPORTED) return rc;
  }
  const bool x3 = p->in_dtype == DG_F32 && fp32x3;
  if (!x3) {   // (fp32x3: the one-tile-per-workgroup kernel below - the lock-step persistent fp32 kernel sits at 256 VGPRs)
    const int rc = p->in_dtype == DG_BF16 ? dg_conv_mfma_persist_launch_bf16(p, stream, 1, wg_cap, plan)
                                          : dg_conv_mfma_persist_launch_f32(p, stream, 1, wg_cap, plan);
    if (rc != DG_EUNSUPPORTED) return rc;
  }
  const bool m128 = p->mode == MODE_GEMM ? false : (p->Wc % 128 == 0);
  const bool n128 = p->N % 128 == 0;
  if (x3) {
    if (m128 && n128) return launch_cfg<float, 128, 128, true>(p, stream, plan);
    if (m128) return launch_cfg<float, 128, 64, true>(p, stream, plan);
    if (n128) return launch_cfg<float, 64, 128, true>(p, stream, plan);
    return launch_cfg<float, 64, 64, true>(p, stream, plan);
  }
  if (p->in_dtype == DG_BF16) {
    if (m128 && n128) return launch_cfg<bf16, 128, 128>(p, stream, plan);
    if (m128) return launch_cfg<bf16, 128, 64>(p, stream, plan);
    if (n128) return launch_cfg<bf16, 64, 128>(p, stream, plan);
    return launch_cfg<bf16, 64, 64>(p, stream, plan);
  }
  if (m128 && n128) return launch_cfg<float, 128, 128>(p, stream, plan);
  if (m128) return launch_cfg<float, 128, 64>(p, stream, plan);
  if (n128) return launch_cfg<float, 64, 128>(p, stream, plan);
  return launch_cfg<float, 64, 64>(p, stream, plan);
}

// a persistent large-tile kernel wherever its geometry allows, else DG_EUNSUPPORTED (parity tests of either family on
// small problems): dg_conv force == 4 -> the lock-step kernel, force == 5 -> the ping-pong kernel
int dg_conv_mfma_big_launch(const ConvP* p, hipStream_t stream, int family, int wg_cap, DgConvPlan* plan) {
  if (!dg_conv_mfma_supported(p)) return DG_EUNSUPPORTED;
  if (family == 5 || family == 9)   // 9: the ping-pong kernel without its both-parities tile (A/B, parity tests)
    return (p->in_dtype == DG_BF16 || p->in_dtype == DG_BF16X2) ? dg_conv_mfma_pp_launch(p, stream, 1, wg_cap, plan, family == 5)
                                                               : DG_EUNSUPPORTED;
  if (p->in_dtype == DG_BF16X2) return DG_EUNSUPPORTED;
  return p->in_dtype == DG_BF16 ? dg_conv_mfma_persist_launch_bf16(p, stream, 0, wg_cap, plan)
                                : dg_conv_mfma_persist_launch_f32(p, stream, 0, wg_cap, plan);
}
