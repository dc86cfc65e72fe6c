// Proj forward (models/gans/dcgan_eqlr.py:6-16: z [B,512] -> [B, h0 w0 C] through EqualLR(ConvTranspose2d) + FusedLeakyReLU)
// as a weight-streaming kernel.  The GEMM reads its 134 MB bf16 weight shadow once and does 4 GFLOP with it: HBM-bound.
// The general one-tile-per-workgroup MFMA kernel ran it at 2.7 TB/s inside the step (cold HBM, 1024 workgroups of four
// K steps each: the ring never reaches steady state, and a ring one or two stages deeper changed nothing).
// Here every WAVE is its own stream: it owns every (4 x gridDim)-th block of 16 weight rows, keeps the whole latent
// operand in registers (the MFMA B fragments of all 16 K steps), and moves its row blocks global -> LDS by LDS-DMA into a
// wave-private two-stage ring (16 KB per stage = 16 rows x 1 KB, one wave instruction per row, non-temporal: the shadow
// is read once per step) - no workgroup barrier in the loop, 32 KB in flight per wave, 128 KB per CU.
//   * MFMA: v_mfma_f32_16x16x32_bf16, A = 16 weight rows x 32 k, B = 32 k x 16 samples; D[row n'][sample b];
//   * LDS rows are unpadded (a DMA piece is lane-linear); the 16-byte chunk c of row r lives at chunk c ^ r (r = 0..15),
//     applied on the DMA source address and undone in the fragment read address: conflict-free for ds_read_b128's lane
//     groups (rows 0-3,12-15 with k group 0 and rows 4-11 with k group 1 share a group: their chunk sets are disjoint);
//   * epilogue: EqualLR scale, bias[n' % C], leaky-relu * sqrt2, four consecutive n' per lane -> one 8-byte store.
#include "mfma_common.h"

namespace {

constexpr int PS_K = 512;                        // latent width this kernel is built for (every shipped config)
constexpr int PS_ROWS = 16;                      // weight rows per stage
constexpr int PS_STAGE = PS_ROWS * PS_K * 2;     // 16 KB
constexpr int PS_NS = 2, PS_WAVES = 4;
constexpr int PS_MAXBIAS = 2048;

__device__ __forceinline__ void ps_dma16_nt(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 2 /* nt */);
}

#define PS_WAITV(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

template <int NB>  // blocks of 16 samples: B <= 16 NB
__global__ __launch_bounds__(256, 1) void proj_stream_kernel(ConvP p, int tiles) {
  __shared__ __attribute__((aligned(1024))) unsigned char lds[PS_WAVES * PS_NS * PS_STAGE];
  __shared__ float s_bias[PS_MAXBIAS];           // (bias reads through LDS: a global load per tile would sit in vmcnt between
                                                 //  the ring's pieces and force a full drain before its use)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kg = lane >> 4;
  const bf16* z = (const bf16*)p.in;
  const char* W = (const char*)p.w;
  bf16* out = (bf16*)p.out;

  // the latent operand, once: zf[s][nb] = z[16 nb + r16][32 s + 8 kg .. + 7]
  bf16x8 zf[16][NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int b = 16 * nb + r16;
    const bf16* zrow = z + (long)(b < p.B ? b : 0) * p.in_sb + 8 * kg;     // (clamped: 32 unconditional loads in flight)
#pragma unroll
    for (int s = 0; s < 16; ++s) zf[s][nb] = *(const bf16x8*)(zrow + 32 * s);
    if (b >= p.B) {
#pragma unroll
      for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) zf[s][nb][e] = (bf16)0.f;
    }
  }

  unsigned char* my = lds + wave * (PS_NS * PS_STAGE);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)my;
  // fragment read addresses: chunk (4 s + kg) ^ r16 of row r16 = ((s ^ (r16 >> 2)) << 2 | (kg ^ (r16 & 3)))
  unsigned ra[4];
#pragma unroll
  for (int q = 0; q < 4; ++q)
    ra[q] = lds0 + (unsigned)(r16 * (PS_K * 2) + (((q ^ (r16 >> 2)) << 6) | ((kg ^ (r16 & 3)) << 4)));
  const long wrow = (long)p.w_sn * 2;            // bytes per weight row
  const int GW = gridDim.x * PS_WAVES, gw = blockIdx.x * PS_WAVES + wave;

  auto issue = [&](int t, int st) __attribute__((always_inline)) {
    const char* src = W + (long)t * PS_ROWS * wrow;
#pragma unroll
    for (int r = 0; r < PS_ROWS; ++r)
      ps_dma16_nt(src + r * wrow + ((lane ^ r) << 4), my + st * PS_STAGE + r * (PS_K * 2));
  };

  const float c_lin = p.epi == EPI_LRELU ? p.scale * SQRT2 : p.scale;
  const float slope = p.epi == EPI_LRELU ? LRELU_SLOPE : 1.f;
  const float bmul = p.epi == EPI_LRELU ? SQRT2 : 1.f;
  const int bmod = p.bias ? p.bias_mod : 1;
  for (int i = tid; i < bmod; i += 256) s_bias[i] = p.bias ? p.bias[i] * bmul : 0.f;
  __syncthreads();                               // (the only workgroup barrier: before the streams start)

  int t = gw;
  if (t < tiles) issue(t, 0);
  if (t + GW < tiles) issue(t + GW, 1);
  for (int i = 0; t < tiles; ++i, t += GW) {
    const int st = i & 1;
    const bool next = t + GW < tiles;           // the other stage holds a tile in flight
    // loads retire in issue order: behind this tile's pieces sit at least the next tile's 16 pieces (first iteration:
    // exactly those - a count of 16 + NB there let the tile's last row still be in flight: NaNs from one row in 8192)
    if (next) PS_WAITV(16); else PS_WAITV(0);
    i32x4 af[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (st == 0) LDS_READ128(af[s], ra[s & 3], (s >> 2) * 256);
      else LDS_READ128(af[s], ra[s & 3], PS_STAGE + (s >> 2) * 256);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(af[s]));
    if (t + 2 * GW < tiles) issue(t + 2 * GW, st);        // the stage is free: its fragments are in registers
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[s]), zf[s][nb], acc[nb], 0, 0, 0);
    // D[row 4 kg + j][col r16]: four consecutive n' of sample 16 nb + r16
    const int n0 = t * PS_ROWS + 4 * kg;
    float bias[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bias[j] = s_bias[(n0 + j) % bmod];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int b = 16 * nb + r16;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        v[j] = fmaf(acc[nb][j], c_lin, bias[j]);
        v[j] = fmaxf(v[j], slope * v[j]);
      }
      unsigned lo, hi;
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(v[0]), "v"(v[1]));
      asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(v[2]), "v"(v[3]));
      if (b < p.B) *(uint2*)(out + (long)b * p.out_sb + n0) = make_uint2(lo, hi);
    }
  }
}

}  // namespace

// bf16 MODE_GEMM forward with K = 512, B <= 32, row-major [N][K] weights and the bias / leaky-relu (or linear) epilogue
int dg_proj_stream_supported(const ConvP* p) {
  if (p->mode != MODE_GEMM || p->in_dtype != DG_BF16 || p->out_dtype != DG_BF16 || p->w_dtype != DG_BF16) return 0;
  if (p->K != PS_K || p->B < 1 || p->B > 32 || p->N % PS_ROWS != 0 || p->N < PS_ROWS) return 0;
  if (p->epi == EPI_MASK || p->dbias || p->nscale || p->rowscale) return 0;
  if (p->in_sk != 1 || p->w_sk != 1 || p->out_sn != 1 || p->w_sn != PS_K) return 0;
  if (p->in_sb % 8 != 0 || p->out_sb % 4 != 0) return 0;
  if (((size_t)p->in & 15) != 0 || ((size_t)p->w & 15) != 0 || ((size_t)p->out & 7) != 0) return 0;
  if (p->bias && (p->bias_mod <= 0 || p->bias_mod > PS_MAXBIAS)) return 0;
  if ((long)p->N / PS_ROWS > 0x7fffffffL) return 0;
  return 1;
}

int dg_proj_stream_launch(const ConvP* p, hipStream_t stream, DgConvPlan* plan) {
  if (!dg_proj_stream_supported(p)) return DG_EUNSUPPORTED;
  const int tiles = p->N / PS_ROWS;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1)
      cus = 256;                                 // (a plan asked for without a device: MI355X)
  }
  int G = cus;                                   // 128 KB of LDS: one workgroup per CU
  if ((long)G * PS_WAVES > tiles) G = (tiles + PS_WAVES - 1) / PS_WAVES;
  if (plan) {
    plan->family = 6; plan->bm = 16 * ((p->B + 15) / 16); plan->bn = PS_ROWS; plan->tiles = tiles; plan->workgroups = G;
    plan->tiles_per_wg = (tiles + G - 1) / G;
    return DG_OK;
  }
  if (p->B <= 16) proj_stream_kernel<1><<<(unsigned)G, 256, 0, stream>>>(*p, tiles);
  else proj_stream_kernel<2><<<(unsigned)G, 256, 0, stream>>>(*p, tiles);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
