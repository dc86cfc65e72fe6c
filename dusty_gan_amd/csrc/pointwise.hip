// HBM-bound kernels of the step: BlurVH, the final (4 x w0) dot, head post-processing (tanh + Gumbel point-drop),
// DiffAugment, NSGAN losses, fetch_reals and small reductions.  Images are fp32 [B,1,H,W]; feature maps are T.
#include "common.h"

// the registered accumulator arena of this device (dg_det_arena; common.h dg_acc_add): read by the kernels of this file that
// sum across blocks into arena slots - per-sample image sums, logits, the augment adjoint's window sums
__device__ DgDet g_det = {nullptr, nullptr, 0};

// 16-byte loads / stores of feature-map elements as floats: V = 8 bf16 or 4 fp32 per access
template <typename T> struct Vec16;
template <> struct Vec16<bf16> {
  static constexpr int V = 8;
  static __device__ __forceinline__ void load(const bf16* p, float (&v)[8]) {
    const uint4 r = *(const uint4*)p;
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      v[2 * k] = __builtin_bit_cast(float, w[k] << 16);
      v[2 * k + 1] = __builtin_bit_cast(float, w[k] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(bf16* p, const float (&v)[8]) {
    unsigned w[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      w[k] = (unsigned)__builtin_bit_cast(unsigned short, (bf16)v[2 * k]) |
             ((unsigned)__builtin_bit_cast(unsigned short, (bf16)v[2 * k + 1]) << 16);
    *(uint4*)p = make_uint4(w[0], w[1], w[2], w[3]);
  }
};
template <> struct Vec16<float> {
  static constexpr int V = 4;
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 r = *(const float4*)p;
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};

// ----------------------------------------------------------------------------------------------------------
// BlurVH (models/ops/common.py:74-88): x [B,H,W] fp32 -> h0 [B,H,W,2] (ch0 = vertical [1,2,1]/4 with reflect rows,
// ch1 = horizontal [1,2,1]/4 with circular / reflect columns).
template <typename T>
__global__ void blur_fwd_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int H, int W, int ring) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * H * W;
  if (idx >= total) return;
  const int xx = (int)(idx % W), y = (int)((idx / W) % H);
  const long base = idx - (long)y * W - xx;  // b*H*W
  const int yu = y == 0 ? 1 : y - 1, yd = y == H - 1 ? H - 2 : y + 1;
  int xl = xx - 1, xr = xx + 1;
  if (ring) { if (xl < 0) xl += W; if (xr >= W) xr -= W; }
  else      { if (xl < 0) xl = 1;  if (xr >= W) xr = W - 2; }
  const float c = x[idx];
  const float v = 0.25f * x[base + (long)yu * W + xx] + 0.5f * c + 0.25f * x[base + (long)yd * W + xx];
  const float h = 0.25f * x[base + (long)y * W + xl] + 0.5f * c + 0.25f * x[base + (long)y * W + xr];
  out[idx * 2 + 0] = (T)v;
  out[idx * 2 + 1] = (T)h;
}

// Adjoint of BlurVH: d [B,H,W,2] -> dx [B,H,W] fp32.
template <typename T>
__global__ void blur_bwd_kernel(const T* __restrict__ d, float* __restrict__ dx, int B, int H, int W, int ring) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * H * W;
  if (idx >= total) return;
  const int xx = (int)(idx % W), y = (int)((idx / W) % H);
  const long base = idx - (long)y * W - xx;
  auto D0 = [&](int yy, int xq) { return (float)d[(base + (long)yy * W + xq) * 2 + 0]; };
  auto D1 = [&](int yy, int xq) { return (float)d[(base + (long)yy * W + xq) * 2 + 1]; };
  float v = 0.5f * D0(y, xx);
  if (y > 0) v += 0.25f * D0(y - 1, xx);
  if (y < H - 1) v += 0.25f * D0(y + 1, xx);
  if (y == 1) v += 0.25f * D0(0, xx);          // row 0 read x[1] as its reflected upper neighbour
  if (y == H - 2) v += 0.25f * D0(H - 1, xx);  // row H-1 read x[H-2] as its reflected lower neighbour
  float h = 0.5f * D1(y, xx);
  if (ring) {
    h += 0.25f * D1(y, xx == 0 ? W - 1 : xx - 1) + 0.25f * D1(y, xx == W - 1 ? 0 : xx + 1);
  } else {
    if (xx > 0) h += 0.25f * D1(y, xx - 1);
    if (xx < W - 1) h += 0.25f * D1(y, xx + 1);
    if (xx == 1) h += 0.25f * D1(y, 0);
    if (xx == W - 2) h += 0.25f * D1(y, W - 1);
  }
  dx[idx] = v + h;
}

// Four pixels per thread (W % 4 == 0): 16-byte loads of the three rows, one 16-byte (bf16) / two (fp32) stores; the same
// expressions as the scalar kernels above, which remain for other widths.
// Grid = (row, sample): the row's neighbours and boundary cases are block-uniform and the per-quad work is 32-bit (the
// first version decoded a flat 64-bit quad index per thread and, in the adjoint, loaded each neighbour row under its own
// condition - one global round trip after the other).
template <typename T>
__global__ __launch_bounds__(256) void blur_fwd4_kernel(const float* __restrict__ x, T* __restrict__ out, int B, int H,
                                                        int W, int ring, const float* __restrict__ mean_src, int mean_n,
                                                        float* __restrict__ mean_acc) {
  if (mean_src && blockIdx.x == 0 && blockIdx.y == 0) {          // rider of block (0, 0): mean_acc[0] += mean(mean_src[0..n))
    __shared__ float red[16];                                    // (dg_mean_acc: the R1 penalty of the micro-batch)
    float sm = 0.f;
    for (int i = threadIdx.x; i < mean_n; i += 256) sm += mean_src[i];
    const float t = dg_block_sum(sm, red);
    if (threadIdx.x == 0) mean_acc[0] += t / mean_n;
  }
  const int y = blockIdx.x, W4 = W >> 2;
  const long base = (long)blockIdx.y * H * W;                    // b*H*W
  const int yu = y == 0 ? 1 : y - 1, yd = y == H - 1 ? H - 2 : y + 1;
  const float* rc = x + base + (long)y * W;
  const float* ru = x + base + (long)yu * W;
  const float* rd = x + base + (long)yd * W;
  for (int q4 = threadIdx.x; q4 < W4; q4 += 256) {
    const int x0 = q4 * 4;
    const float4 c4 = *(const float4*)(rc + x0);
    const float4 u4 = *(const float4*)(ru + x0);
    const float4 d4 = *(const float4*)(rd + x0);
    int xl = x0 - 1, xr = x0 + 4;
    if (ring) { if (xl < 0) xl += W; if (xr >= W) xr -= W; }
    else      { if (xl < 0) xl = 1;  if (xr >= W) xr = W - 2; }
    const float c[6] = {rc[xl], c4.x, c4.y, c4.z, c4.w, rc[xr]};
    const float u[4] = {u4.x, u4.y, u4.z, u4.w}, d[4] = {d4.x, d4.y, d4.z, d4.w};
    float o[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      o[2 * k] = 0.25f * u[k] + 0.5f * c[k + 1] + 0.25f * d[k];
      o[2 * k + 1] = 0.25f * c[k] + 0.5f * c[k + 1] + 0.25f * c[k + 2];
    }
    T* op = out + (base + (long)y * W + x0) * 2;
    if constexpr (sizeof(T) == 2) {
      Vec16<bf16>::store((bf16*)op, o);
    } else {
      *(float4*)op = make_float4(o[0], o[1], o[2], o[3]);
      *(float4*)(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
    }
  }
}

// BlurVH's adjoint for NR consecutive rows ys .. ys + NR - 1 of one pixel quad (columns x0 .. x0 + 3): the NR + 2 source rows
// and the 2 NR ring neighbours of a thread are loaded TOGETHER, unconditionally (rows clamped into the image; what a clamped
// row contributes is never used) - row after row, each row's loads waited for before the next row's were issued: four to six
// dependent memory round trips per workgroup (round 6: blur_bwd4_kernel 8.9 us, 6.8 without its sums).  Same expressions in
// the same order as the per-row code: bit-identical results.  Rows outside [0, H) come back as garbage the caller skips.
template <typename T, int NR>
__device__ __forceinline__ void blur_adj_rows(const T* __restrict__ d, long base, int ys, int x0, int H, int W, int ring,
                                              float (&g)[NR][4]) {
  float rows[NR + 2][8];
  float el[NR], er[NR];
  auto clampy = [&](int yy) { return yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy); };
#pragma unroll
  for (int r = 0; r < NR + 2; ++r) {
    const T* p = d + (base + (long)clampy(ys - 1 + r) * W + x0) * 2;
    if constexpr (sizeof(T) == 2) {
      Vec16<bf16>::load((const bf16*)p, rows[r]);
    } else {
      const float4 a = *(const float4*)p, b2 = *(const float4*)(p + 4);
      rows[r][0] = a.x; rows[r][1] = a.y; rows[r][2] = a.z; rows[r][3] = a.w;
      rows[r][4] = b2.x; rows[r][5] = b2.y; rows[r][6] = b2.z; rows[r][7] = b2.w;
    }
  }
  // channel 1 of the pixels left and right of the quad (circular columns: wrapped; reflect columns: clamped - then unused
  // at the border): two 2- or 4-byte loads per row, unconditional like the rows'
  const int xl = ring ? (x0 == 0 ? W - 1 : x0 - 1) : (x0 == 0 ? 0 : x0 - 1);
  const int xr = ring ? (x0 + 3 == W - 1 ? 0 : x0 + 4) : (x0 + 4 > W - 1 ? W - 1 : x0 + 4);
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const long rowb = base + (long)clampy(ys + j) * W;
    el[j] = (float)d[(rowb + xl) * 2 + 1];
    er[j] = (float)d[(rowb + xr) * 2 + 1];
  }
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    const int y = ys + j;
    const float (&m)[8] = rows[j + 1];
    const float (&tu)[8] = rows[j];
    const float (&td)[8] = rows[j + 2];
    float v[4], h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = 0.5f * m[2 * k];
    if (y > 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += 0.25f * tu[2 * k]; }
    if (y < H - 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += 0.25f * td[2 * k]; }
    if (y == 1) {                                  // row 0 read x[1] as its reflected upper neighbour (row 0 IS tu here)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += 0.25f * tu[2 * k]; }
    if (y == H - 2) {                              // row H-1 read x[H-2] as its reflected lower neighbour (row H-1 IS td here)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] += 0.25f * td[2 * k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int xx = x0 + k;
      h[k] = 0.5f * m[2 * k + 1];
      const bool hasl = k > 0, hasr = k < 3;
      if (ring) {
        h[k] += 0.25f * (hasl ? m[2 * k - 1] : el[j]) + 0.25f * (hasr ? m[2 * k + 3] : er[j]);
      } else {
        if (xx > 0) h[k] += 0.25f * (hasl ? m[2 * k - 1] : el[j]);
        if (xx < W - 1) h[k] += 0.25f * (hasr ? m[2 * k + 3] : er[j]);
        if (xx == 1) h[k] += 0.25f * m[1];         // column 0 read x[1] as its reflected left neighbour (pixel 0 = this quad's first)
        if (xx == W - 2) h[k] += 0.25f * m[7];     // column W-1 read x[W-2] (pixel W-1 = this quad's last)
      }
      g[j][k] = v[k] + h[k];
    }
  }
}

// ssq != nullptr (R1): dx = oscale * g and ssq[b] += sum of g^2 over the sample, g = the adjoint's result - the R1
// penalty's per-sample |g|^2 and its tangent v = (gp / B) g in the pass that makes g.  A block owns `rows_pb` consecutive
// rows of one sample (one atomic per block).
// The rows / columns of sample b's augmented image whose gradient reaches the source image (DiffAugment's adjoint sums
// exactly these: diffaug_bwd_sum_kernel): rows y with 0 <= y + t_h < H, minus the cut-out box.
struct AugWin { const int *t_h, *o_x, *o_y; int policy, cut_h, cut_w; };

// ssq != nullptr && !win: R1 form (below).  win != nullptr: ssq[b] += sum of g over the sample's window `win` instead
// (the contrast term of DiffAugment's adjoint) - the pass that makes g also makes the sum its adjoint needs.
template <typename T>
__global__ __launch_bounds__(256) void blur_bwd4_kernel(const T* __restrict__ d, float* __restrict__ dx, int B, int H,
                                                        int W, int ring, float oscale, float* __restrict__ ssq, int rows_pb,
                                                        AugWin win, int use_win) {
  __shared__ float red[16];
  const int W4 = W >> 2, b = blockIdx.y;
  const long base = (long)b * H * W;
  float ssacc = 0.f;
  const int y0 = blockIdx.x * rows_pb, y1 = y0 + rows_pb < H ? y0 + rows_pb : H;
  int w_th = 0, w_r0 = 0, w_c0 = 0;
  if (use_win) {
    w_th = (win.policy & 8) ? win.t_h[b] : 0;
    w_r0 = (win.policy & 16) ? win.o_x[b] - win.cut_h / 2 : 0;
    w_c0 = (win.policy & 16) ? win.o_y[b] - win.cut_w / 2 : 0;
  }
  if (rows_pb == 4 && y0 + 4 <= H) {               // the band's loads batched (blur_adj_rows)
    for (int q4 = threadIdx.x; q4 < W4; q4 += 256) {
      const int x0 = q4 * 4;
      float g[4][4];
      blur_adj_rows<T, 4>(d, base, y0, x0, H, W, ring, g);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int y = y0 + j;
        const float g0 = g[j][0], g1 = g[j][1], g2 = g[j][2], g3 = g[j][3];
        *(float4*)(dx + base + (long)y * W + x0) = make_float4(oscale * g0, oscale * g1, oscale * g2, oscale * g3);
        if (!use_win) {
          ssacc += g0 * g0 + g1 * g1 + g2 * g2 + g3 * g3;
        } else if (y + w_th >= 0 && y + w_th < H) {
          const bool cutrow = (win.policy & 16) && y >= w_r0 && y < w_r0 + win.cut_h;
          const float gq[4] = {g0, g1, g2, g3};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (!(cutrow && x0 + k >= w_c0 && x0 + k < w_c0 + win.cut_w)) ssacc += gq[k];
        }
      }
    }
  } else
  for (int y = y0; y < y1; ++y)
  for (int q4 = threadIdx.x; q4 < W4; q4 += 256) {
  const int x0 = q4 * 4;
  auto row8 = [&](int yy, float (&v)[8]) {                      // (ch0, ch1) of pixels x0 .. x0+3 of row yy
    const T* p = d + (base + (long)yy * W + x0) * 2;
    if constexpr (sizeof(T) == 2) {
      Vec16<bf16>::load((const bf16*)p, v);
    } else {
      const float4 a = *(const float4*)p, b2 = *(const float4*)(p + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b2.x; v[5] = b2.y; v[6] = b2.z; v[7] = b2.w;
    }
  };
  auto D1 = [&](int yy, int xq) { return (float)d[(base + (long)yy * W + xq) * 2 + 1]; };
  // the row and its two vertical neighbours: three unconditional loads (clamped rows are loaded and not used)
  float m[8], tu[8], td[8];
  row8(y, m);
  row8(y > 0 ? y - 1 : y, tu);
  row8(y < H - 1 ? y + 1 : y, td);
  float el = 0.f, er = 0.f;                                      // ring: the horizontal neighbours outside the quad
  if (ring) { el = D1(y, x0 == 0 ? W - 1 : x0 - 1); er = D1(y, x0 + 3 == W - 1 ? 0 : x0 + 4); }
  float v[4], h[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = 0.5f * m[2 * k];
  if (y > 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += 0.25f * tu[2 * k]; }
  if (y < H - 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += 0.25f * td[2 * k]; }
  if (y == 1) { float t[8]; row8(0, t);          // row 0 read x[1] as its reflected upper neighbour
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += 0.25f * t[2 * k]; }
  if (y == H - 2) { float t[8]; row8(H - 1, t);  // row H-1 read x[H-2] as its reflected lower neighbour
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += 0.25f * t[2 * k]; }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int xx = x0 + k;
    h[k] = 0.5f * m[2 * k + 1];
    const bool hasl = k > 0, hasr = k < 3;       // neighbours inside the quad come from registers
    if (ring) {
      h[k] += 0.25f * (hasl ? m[2 * k - 1] : el) + 0.25f * (hasr ? m[2 * k + 3] : er);
    } else {
      if (xx > 0) h[k] += 0.25f * (hasl ? m[2 * k - 1] : D1(y, xx - 1));
      if (xx < W - 1) h[k] += 0.25f * (hasr ? m[2 * k + 3] : D1(y, xx + 1));
      if (xx == 1) h[k] += 0.25f * D1(y, 0);
      if (xx == W - 2) h[k] += 0.25f * D1(y, W - 1);
    }
  }
  const float g0 = v[0] + h[0], g1 = v[1] + h[1], g2 = v[2] + h[2], g3 = v[3] + h[3];
  *(float4*)(dx + base + (long)y * W + x0) = make_float4(oscale * g0, oscale * g1, oscale * g2, oscale * g3);
  if (!use_win) {
    ssacc += g0 * g0 + g1 * g1 + g2 * g2 + g3 * g3;
  } else if (y + w_th >= 0 && y + w_th < H) {
    const bool cutrow = (win.policy & 16) && y >= w_r0 && y < w_r0 + win.cut_h;
    const float gq[4] = {g0, g1, g2, g3};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (!(cutrow && x0 + k >= w_c0 && x0 + k < w_c0 + win.cut_w)) ssacc += gq[k];
  }
  }
  if (ssq) {
    const float sblk = dg_block_sum(ssacc, red);
    if (threadIdx.x == 0) dg_acc_add(&ssq[b], sblk, gridDim.x, g_det);
  }
}

// ----------------------------------------------------------------------------------------------------------
// R1's turn-around at the image in ONE launch (round 6; trainers/dcgan_amp.py:218-235): g = BlurVH^T(e0) is the gradient of
// sum(y_real) w.r.t. the augmented real image, the penalty reads |g_b|^2, and the double backward's tangent v = oscale g goes
// straight back up through BlurVH (models/ops/common.py:74-88) - so g never needs to exist in memory: a block owns a band
// of R1T_ROWS rows of one sample, forms oscale g for the band and one halo row on either side in LDS (the same expressions,
// in the same order, as blur_bwd4_kernel), then BlurVH of those rows (blur_fwd4_kernel's expressions) into the tangent
// slot of h0.  ssq[b] += |g_b|^2 over the band's own rows; mean_acc[0] += the same / mean_n (the logged penalty: its mean
// over the batch is the sum of all blocks' shares - no second pass over ssq).  W % 4 == 0, H % R1T_ROWS == 0.
#define R1T_ROWS 4
template <typename T>
__global__ __launch_bounds__(256) void blur_r1_tangent_kernel(const T* __restrict__ d, T* __restrict__ out, int H, int W,
                                                              int ring, float oscale, float* __restrict__ ssq,
                                                              float* __restrict__ mean_acc, int mean_n) {
  extern __shared__ float s_g[];                  // [R1T_ROWS + 2][W]: oscale * g of rows y0 - 1 .. y0 + R1T_ROWS
  __shared__ float red[16];
  const int W4 = W >> 2, b = blockIdx.y;
  const long base = (long)b * H * W;
  const int y0 = blockIdx.x * R1T_ROWS;
  float ssacc = 0.f;
  for (int q4 = threadIdx.x; q4 < W4; q4 += 256) {
    const int x0 = q4 * 4;
    float g[R1T_ROWS + 2][4];
    blur_adj_rows<T, R1T_ROWS + 2>(d, base, y0 - 1, x0, H, W, ring, g);   // rows y0 - 1 .. y0 + R1T_ROWS, their loads batched
#pragma unroll
    for (int r = 0; r < R1T_ROWS + 2; ++r) {
      const int y = y0 - 1 + r;
      if (y < 0 || y >= H) continue;              // (block-uniform: the rows beyond the image are never read below)
      const float g0 = g[r][0], g1 = g[r][1], g2 = g[r][2], g3 = g[r][3];
      *(float4*)(s_g + r * W + x0) = make_float4(oscale * g0, oscale * g1, oscale * g2, oscale * g3);
      if (r >= 1 && r <= R1T_ROWS) ssacc += g0 * g0 + g1 * g1 + g2 * g2 + g3 * g3;
    }
  }
  __syncthreads();
  for (int yy = 0; yy < R1T_ROWS; ++yy) {
    const int y = y0 + yy;
    const int yu = y == 0 ? 1 : y - 1, yd = y == H - 1 ? H - 2 : y + 1;
    const float* rc = s_g + (y - y0 + 1) * W;
    const float* ru = s_g + (yu - y0 + 1) * W;
    const float* rd = s_g + (yd - y0 + 1) * W;
    for (int q4 = threadIdx.x; q4 < W4; q4 += 256) {
      const int x0 = q4 * 4;
      const float4 c4 = *(const float4*)(rc + x0);
      const float4 u4 = *(const float4*)(ru + x0);
      const float4 d4 = *(const float4*)(rd + x0);
      int xl = x0 - 1, xr = x0 + 4;
      if (ring) { if (xl < 0) xl += W; if (xr >= W) xr -= W; }
      else      { if (xl < 0) xl = 1;  if (xr >= W) xr = W - 2; }
      const float c[6] = {rc[xl], c4.x, c4.y, c4.z, c4.w, rc[xr]};
      const float u[4] = {u4.x, u4.y, u4.z, u4.w}, dn[4] = {d4.x, d4.y, d4.z, d4.w};
      float o[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[2 * k] = 0.25f * u[k] + 0.5f * c[k + 1] + 0.25f * dn[k];
        o[2 * k + 1] = 0.25f * c[k] + 0.5f * c[k + 1] + 0.25f * c[k + 2];
      }
      T* op = out + (base + (long)y * W + x0) * 2;
      if constexpr (sizeof(T) == 2) {
        Vec16<bf16>::store((bf16*)op, o);
      } else {
        *(float4*)op = make_float4(o[0], o[1], o[2], o[3]);
        *(float4*)(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
  }
  const float sblk = dg_block_sum(ssacc, red);
  if (threadIdx.x == 0) {
    // the sample's last band adds the sample's total to the batch mean: gridDim.y adds to that word, not gridDim.x gridDim.y
    // (512 adds to ONE word serialise memory-side: 9 us of this launch, scripts/bench_pointwise.py)
    float tot = 0.f;
    const int last = dg_acc_add_last(&ssq[b], sblk, gridDim.x, g_det, tot);
    if (mean_acc) {
      if (last == 1) dg_acc_add(mean_acc, tot / (float)mean_n, gridDim.y, g_det);
      else if (last < 0) atomicAdd(mean_acc, sblk / (float)mean_n);
    }
  }
}

// ----------------------------------------------------------------------------------------------------------
// Final EqualLR(Conv2d(C,1,(h0,w0))) (models/gans/dcgan_eqlr.py:95): y[b] = scale * <d4[b], wf> + bias.
template <typename T>
__global__ __launch_bounds__(256) void final_fwd_kernel(const T* __restrict__ d4, const float* __restrict__ wf,
                                                        const float* __restrict__ bias, float scale, long n,
                                                        float* __restrict__ y) {
  // grid = (slabs, B): each block reduces one slab of one sample and adds it to y[b] (zeroed by the launcher);
  // slab 0 also adds the bias.  One block per sample left 7/8 of the chip idle (0.18 ms per call at B = 64).
  // 16-byte accesses (n % V == 0, checked by the launcher; else the scalar kernel below).
  __shared__ float red[16];
  constexpr int V = Vec16<T>::V;
  const int b = blockIdx.y;
  const T* row = d4 + (long)b * n;
  float acc = 0.f;
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * V; i < n; i += (long)gridDim.x * blockDim.x * V) {
    float a[V];
    Vec16<T>::load(row + i, a);
#pragma unroll
    for (int k = 0; k < V; k += 4) {
      const float4 w4 = *(const float4*)(wf + i + k);
      acc += a[k] * w4.x + a[k + 1] * w4.y + a[k + 2] * w4.z + a[k + 3] * w4.w;
    }
  }
  const float s = dg_block_sum(acc, red);
  if (threadIdx.x == 0) dg_acc_add(&y[b], s * scale + ((bias && blockIdx.x == 0) ? bias[0] : 0.f), gridDim.x, g_det);
}
template <typename T>
__global__ __launch_bounds__(256) void final_fwd_scalar_kernel(const T* __restrict__ d4, const float* __restrict__ wf,
                                                               const float* __restrict__ bias, float scale, long n,
                                                               float* __restrict__ y) {
  __shared__ float red[16];
  const int b = blockIdx.y;
  const T* row = d4 + (long)b * n;
  float acc = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    acc += (float)row[i] * wf[i];
  const float s = dg_block_sum(acc, red);
  if (threadIdx.x == 0) dg_acc_add(&y[b], s * scale + ((bias && blockIdx.x == 0) ? bias[0] : 0.f), gridDim.x, g_det);
}

// dd4[b][i] = up[b] * scale * wf[i] * lrelu'(d4[b][i]) * sqrt2 ; dbias4[i % C] += rowscale[b] * dd4[b][i]
// A block owns 64 V consecutive elements; its four waves split the samples (wave w: b = w, w + 4, ...), a lane owns V
// consecutive elements (V consecutive channels: C % V == 0): 16-byte loads and stores, four samples in flight per
// element tile, the waves' bias-gradient partials meet in LDS and leave as ONE atomic per element per block.  (One
// element per thread over all samples in turn: 2-byte accesses, 15 us for 34 MB.)
template <typename T>
__global__ __launch_bounds__(256) void final_bwd_data_kernel(const T* __restrict__ d4, const float* __restrict__ wf,
                                                             const float* __restrict__ up,
                                                             const float* __restrict__ rowscale, float scale, int B,
                                                             long n, int C, T* __restrict__ dd4,
                                                             float* __restrict__ dbias) {
  constexpr int V = Vec16<T>::V;
  __shared__ float part[4][64 * V];
  __shared__ float s_u[256], s_r[256];          // per-sample factors (B <= 256, checked by the launcher): LDS reads inside
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;   // the loop keep its global loads free of other waits
  for (int b = threadIdx.x; b < B; b += 256) { s_u[b] = up ? up[b] : 1.f; s_r[b] = rowscale ? rowscale[b] : 1.f; }
  const long i = ((long)blockIdx.x * 64 + lane) * V;
  float w[V], db[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { w[k] = 0.f; db[k] = 0.f; }
  if (i < n) {                                  // (n % V == 0: a thread's V elements are all inside)
#pragma unroll
    for (int k4 = 0; k4 < V; k4 += 4) {         // (wf 16-byte aligned: checked by the launcher)
      const float4 r = *(const float4*)(wf + i + k4);
      w[k4] = r.x * scale; w[k4 + 1] = r.y * scale; w[k4 + 2] = r.z * scale; w[k4 + 3] = r.w * scale;
    }
  }
  __syncthreads();
  if (i < n) {
#pragma unroll 4
    for (int b = wave; b < B; b += 4) {
      float a[V], g[V];
      Vec16<T>::load(d4 + (long)b * n + i, a);
      const float u = s_u[b], rs = s_r[b];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        g[k] = u * w[k] * (a[k] > 0.f ? SQRT2 : LRELU_SLOPE * SQRT2);
        db[k] += rs * g[k];
      }
      Vec16<T>::store(dd4 + (long)b * n + i, g);
    }
  }
  if (!dbias) return;
#pragma unroll
  for (int k = 0; k < V; ++k) part[wave][lane * V + k] = db[k];
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * V; e += 256) {
    const long ie = (long)blockIdx.x * 64 * V + e;
    if (ie < n) atomicAdd(&dbias[ie % C], part[0][e] + part[1][e] + part[2][e] + part[3][e]);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void final_bwd_data_scalar_kernel(const T* __restrict__ d4, const float* __restrict__ wf,
                                                                    const float* __restrict__ up,
                                                                    const float* __restrict__ rowscale, float scale,
                                                                    int B, long n, int C, T* __restrict__ dd4,
                                                                    float* __restrict__ dbias) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float w = wf[i] * scale;
  float db = 0.f;
  for (int b = 0; b < B; ++b) {
    const float a = (float)d4[(long)b * n + i];
    const float g = (up ? up[b] : 1.f) * w * (a > 0.f ? SQRT2 : LRELU_SLOPE * SQRT2);
    dd4[(long)b * n + i] = (T)g;
    db += (rowscale ? rowscale[b] : 1.f) * g;
  }
  if (dbias) atomicAdd(&dbias[i % C], db);
}

// out[i] += scale * sum_b coef[b] * src[b][i]   (coef null -> 1).  Vector form: a block owns 64 V consecutive elements,
// its four waves split the samples, partials meet in LDS, plain read-modify-write of out (no atomics).
template <typename T>
__global__ __launch_bounds__(256) void batch_wsum_kernel(const T* __restrict__ src, const float* __restrict__ coef,
                                                         float scale, int B, long n, float* __restrict__ out) {
  constexpr int V = Vec16<T>::V;
  __shared__ float part[4][64 * V];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long i = ((long)blockIdx.x * 64 + lane) * V;
  float acc[V];
#pragma unroll
  for (int k = 0; k < V; ++k) acc[k] = 0.f;
  if (i < n) {
#pragma unroll 8
    for (int b = wave; b < B; b += 4) {
      float a[V];
      Vec16<T>::load(src + (long)b * n + i, a);
      const float c = coef ? coef[b] : 1.f;
#pragma unroll
      for (int k = 0; k < V; ++k) acc[k] += c * a[k];
    }
  }
#pragma unroll
  for (int k = 0; k < V; ++k) part[wave][lane * V + k] = acc[k];
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * V; e += 256) {
    const long ie = (long)blockIdx.x * 64 * V + e;
    if (ie < n) out[ie] += (part[0][e] + part[1][e] + part[2][e] + part[3][e]) * scale;
  }
}
template <typename T>
__global__ void batch_wsum_scalar_kernel(const T* __restrict__ src, const float* __restrict__ coef, float scale, int B,
                                         long n, float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) acc += (coef ? coef[b] : 1.f) * (float)src[(long)b * n + i];
  out[i] += acc * scale;
}

// ----------------------------------------------------------------------------------------------------------
// Head post-processing: Generator.forward's tanh (models/gans/dcgan_eqlr.py:71) + DUSty maskout
// (models/dusty.py:77-91, 107-127).  gout [B,1+k,H,W] planar fp32: ch0 raw depth -> tanh in place (depth_orig),
// ch1.. confidence logits (kept).  arch: 0 none, 1 dusty1, 2 dusty2.  noise_pixel [B,H,W], noise_image [B].
// dsum != nullptr: dsum[b] += sum of depth[b] - the per-sample sum DiffAugment's contrast needs of its input, produced where
// the image is produced.  A block then owns `chunk` consecutive pixels of ONE sample (HW % chunk == 0) and issues one atomic:
// with one block per 256 pixels the 8192 atomics on 32 addresses cost 80 us (round 1 met the same in head_post_bwd).
template <int arch>
__device__ __forceinline__ float head_post_px(float* __restrict__ gout, const float* __restrict__ noise_pixel,
                                              const float* __restrict__ noise_image, int training,
                                              float inv_tau, float drop_const, long HW, float* __restrict__ mask,
                                              int b, long p) {
  const long idx = (long)b * HW + p;
  const int nch = 1 + (arch == 0 ? 0 : arch);
  float* g = gout + (long)b * nch * HW + p;
  const float t = dg_tanh(g[0]);
  g[0] = t;
  if (arch == 0) return t;
  const float sp = 1.f / (1.f + __expf(-(g[HW] + noise_pixel[idx]) * inv_tau));
  const float mp = sp > 0.5f ? 1.f : 0.f;
  float m = mp;
  if (arch == 1) {
    mask[idx] = mp;
  } else {
    float mi;
    if (training) {
      const float si = 1.f / (1.f + __expf(-(g[2 * HW] + noise_image[b]) * inv_tau));
      mi = si > 0.5f ? 1.f : 0.f;
    } else {
      mi = g[2 * HW] > 0.f ? 1.f : 0.f;
    }
    mask[(long)b * 2 * HW + p] = mp;
    mask[(long)b * 2 * HW + HW + p] = mi;
    m = mp * mi;
  }
  return m * t + (1.f - m) * drop_const;
}
// Four consecutive pixels per thread (HW % 1024 == 0, sums wanted): 16-byte loads / stores of every plane; same arithmetic
// as head_post_px (which remains for other sizes).
template <int arch>
__global__ __launch_bounds__(256) void head_post_fwd4_kernel(float* __restrict__ gout, const float* __restrict__ noise_pixel,
                                      const float* __restrict__ noise_image, int training, float inv_tau,
                                      float drop_const, int B, long HW, float* __restrict__ mask,
                                      float* __restrict__ depth, float* __restrict__ dsum, int chunk) {
  __shared__ float red[16];
  const long i0 = (long)blockIdx.x * chunk;
  const int b = (int)(i0 / HW);
  const long p0 = i0 - (long)b * HW;
  constexpr int nch = 1 + (arch == 0 ? 0 : arch);
  float* g = gout + (long)b * nch * HW + p0;
  const float ni = (arch == 2 && training) ? noise_image[b] : 0.f;
  float acc = 0.f;
  // the loads of four trips issued before the first store (gout is rewritten in place: the compiler cannot hoist a later
  // trip's loads over an earlier trip's stores, and one load in flight per wave left the launch at a third of the HBM rate)
  constexpr int U = 4;
  for (int k0 = threadIdx.x * 4; k0 < chunk; k0 += U * 1024) {
    float4 g0u[U], g1u[U], npu[U], g2u[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * 1024;
      if (k < chunk) {
        g0u[u] = *(const float4*)(g + k);
        if (arch >= 1) { g1u[u] = *(const float4*)(g + HW + k); npu[u] = *(const float4*)(noise_pixel + (long)b * HW + p0 + k); }
        if (arch == 2) g2u[u] = *(const float4*)(g + 2 * HW + k);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
    const int k = k0 + u * 1024;
    if (k >= chunk) break;
    const float4 g0 = g0u[u];
    float t[4] = {dg_tanh(g0.x), dg_tanh(g0.y), dg_tanh(g0.z), dg_tanh(g0.w)};
    *(float4*)(g + k) = make_float4(t[0], t[1], t[2], t[3]);
    float dv[4] = {t[0], t[1], t[2], t[3]};
    if (arch >= 1) {
      const float4 g1 = g1u[u], np = npu[u];
      const float l1[4] = {g1.x + np.x, g1.y + np.y, g1.z + np.z, g1.w + np.w};
      float mp[4], m[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float sp = 1.f / (1.f + __expf(-l1[q] * inv_tau));
        mp[q] = sp > 0.5f ? 1.f : 0.f;
        m[q] = mp[q];
      }
      if (arch == 1) {
        *(float4*)(mask + (long)b * HW + p0 + k) = make_float4(mp[0], mp[1], mp[2], mp[3]);
      } else {
        const float4 g2 = g2u[u];
        const float l2[4] = {g2.x, g2.y, g2.z, g2.w};
        float mi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (training) {
            const float si = 1.f / (1.f + __expf(-(l2[q] + ni) * inv_tau));
            mi[q] = si > 0.5f ? 1.f : 0.f;
          } else {
            mi[q] = l2[q] > 0.f ? 1.f : 0.f;
          }
          m[q] = mp[q] * mi[q];
        }
        *(float4*)(mask + (long)b * 2 * HW + p0 + k) = make_float4(mp[0], mp[1], mp[2], mp[3]);
        *(float4*)(mask + (long)b * 2 * HW + HW + p0 + k) = make_float4(mi[0], mi[1], mi[2], mi[3]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) dv[q] = m[q] * t[q] + (1.f - m[q]) * drop_const;
    }
    *(float4*)(depth + i0 + k) = make_float4(dv[0], dv[1], dv[2], dv[3]);
    acc += (dv[0] + dv[1]) + (dv[2] + dv[3]);
    }
  }
  const float sblk = dg_block_sum(acc, red);
  if (threadIdx.x == 0) dg_acc_add(&dsum[b], sblk, (unsigned)(HW / chunk), g_det);
}

template <int arch>   // compile-time: the pixel function is then straight-line code and the unrolled trips batch their loads
__global__ __launch_bounds__(256) void head_post_fwd_kernel(float* __restrict__ gout, const float* __restrict__ noise_pixel,
                                     const float* __restrict__ noise_image, int training, float inv_tau,
                                     float drop_const, int B, long HW, float* __restrict__ mask,
                                     float* __restrict__ depth, float* __restrict__ dsum, int chunk) {
  __shared__ float red[16];
  if (!dsum) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (long)B * HW) {
      const int b = (int)(idx / HW);
      depth[idx] = head_post_px<arch>(gout, noise_pixel, noise_image, training, inv_tau, drop_const, HW, mask, b, idx - (long)b * HW);
    }
    return;
  }
  // the block's pixels belong to ONE sample (HW % chunk == 0): the sample index is block-uniform, and the four pixels a
  // thread handles per trip are independent - unrolled so that their loads are in flight together (a 64-bit division per
  // pixel and one round trip per pixel made this 15 us for 25 MB)
  const long i0 = (long)blockIdx.x * chunk;
  const int b = (int)(i0 / HW);
  const long p0 = i0 - (long)b * HW;
  float acc = 0.f;
#pragma unroll 4
  for (int k = threadIdx.x; k < chunk; k += 256) {
    const float dv = head_post_px<arch>(gout, noise_pixel, noise_image, training, inv_tau, drop_const, HW, mask, b, p0 + k);
    depth[i0 + k] = dv;
    acc += dv;
  }
  const float sblk = dg_block_sum(acc, red);
  if (threadIdx.x == 0) dg_acc_add(&dsum[b], sblk, (unsigned)(HW / chunk), g_det);
}

// Backward of the above: ddepth [B,H,W] -> draw [B,1+k,H,W] planar (gradient w.r.t. the head conv outputs).
template <int arch, int CP>   // CP: 0 no pixel-major copy, 2 / 4 that padded channel count, 1 any other (`cp`)
__global__ __launch_bounds__(256) void head_post_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ noise_pixel,
                                     const float* __restrict__ noise_image, const float* __restrict__ mask,
                                     const float* __restrict__ ddepth, float inv_tau, float drop_const,
                                     int B, long HW, float s_depth, float s_conf, float* __restrict__ draw,
                                     float* __restrict__ dbias, bf16* __restrict__ draw_pm, int cp) {
  __shared__ float red[16];
  // grid-stride: a block covers many pixels so that the bias-gradient sums cost one atomic per block per head (one
  // pixel per thread meant 8192 atomics on the same address: 100 us of the 108 us this kernel took at B = 32)
  // blockIdx.y = sample (no 64-bit division per pixel); straight-line body (arch and the copy's layout are compile-time),
  // four independent pixels per trip in flight together
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  const int b = blockIdx.y;
#pragma unroll 4
  for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long)gridDim.x * blockDim.x) {
  const long idx = (long)b * HW + p;
  float d0 = 0.f, d1 = 0.f, d2 = 0.f;  // unscaled gradients w.r.t. the head outputs (= the head bias gradients)
  {
  const int nch = 1 + (arch == 0 ? 0 : arch);
  const float* g = gout + (long)b * nch * HW + p;
  float* d = draw + (long)b * nch * HW + p;
  const float t = g[0];
  const float dt = 1.f - t * t;
  const float go = ddepth[idx];
  if (arch == 0) {
    d0 = go * dt;
  } else {
    const float sp = 1.f / (1.f + __expf(-(g[HW] + noise_pixel[idx]) * inv_tau));
    const float dmask = go * (t - drop_const);
    if (arch == 1) {
      const float mp = mask[idx];
      d0 = mp * go * dt;
      d1 = dmask * sp * (1.f - sp) * inv_tau;
    } else {
      const float mp = mask[(long)b * 2 * HW + p], mi = mask[(long)b * 2 * HW + HW + p];
      const float si = 1.f / (1.f + __expf(-(g[2 * HW] + noise_image[b]) * inv_tau));
      d0 = mp * mi * go * dt;
      d1 = dmask * mi * sp * (1.f - sp) * inv_tau;
      d2 = dmask * mp * si * (1.f - si) * inv_tau;
      d[2 * HW] = d2 * s_conf;
    }
    d[HW] = d1 * s_conf;
  }
  d[0] = d0 * s_depth;
  if (CP != 0) {  // second copy, pixel-major / channel-minor bf16 [B,H,W,cp], channels zero-padded: the operand layout
                  // of the MFMA backward-data kernel (thin_s2_mfma); one store per pixel for cp = 2 / 4
    const unsigned short h0 = __builtin_bit_cast(unsigned short, (bf16)(d0 * s_depth));
    const unsigned short h1 = __builtin_bit_cast(unsigned short, (bf16)(arch >= 1 ? d1 * s_conf : 0.f));
    const unsigned short h2 = __builtin_bit_cast(unsigned short, (bf16)(arch >= 2 ? d2 * s_conf : 0.f));
    if (CP == 2) {
      *(unsigned*)(draw_pm + idx * 2) = (unsigned)h0 | ((unsigned)h1 << 16);
    } else if (CP == 4) {
      *(uint2*)(draw_pm + idx * 4) = make_uint2((unsigned)h0 | ((unsigned)h1 << 16), (unsigned)h2);
    } else {      // any other padded channel count
      bf16* q = draw_pm + idx * cp;
      q[0] = __builtin_bit_cast(bf16, h0);
      if (cp > 1) q[1] = __builtin_bit_cast(bf16, h1);
      if (cp > 2) q[2] = __builtin_bit_cast(bf16, h2);
      for (int c = 3; c < cp; ++c) q[c] = (bf16)0.f;
    }
  }
  }
  a0 += d0; a1 += d1; a2 += d2;
  }
  if (dbias) {  // head biases are outside EqualLR's input scaling: their gradient is the unscaled sum
    const float s0 = dg_block_sum(a0, red);
    if (threadIdx.x == 0) atomicAdd(&dbias[0], s0);
    if (arch >= 1) { const float s1 = dg_block_sum(a1, red); if (threadIdx.x == 0) atomicAdd(&dbias[1], s1); }
    if (arch >= 2) { const float s2 = dg_block_sum(a2, red); if (threadIdx.x == 0) atomicAdd(&dbias[2], s2); }
  }
}

struct AugP {
  const float *u_b, *u_c;
  const int *t_h, *t_w, *o_x, *o_y;
  int policy, B, H, W, cut_h, cut_w;
};

// Where head_post_bwd4_kernel gets d loss / d depth of a pixel quad from: the tensor itself, or - HeadGradAug - DiffAugment's
// adjoint gather applied on the fly to the BlurVH adjoint's output gy (diffaug_bwd_kernel's arithmetic for the four
// columns of a quad; W % 4 == 0, so a quad lies in one row): the generator's upstream gradient is then never written.
struct HeadGradPlain {
  const float* ddepth;
  __device__ __forceinline__ float4 operator()(int b, long p, long HW) const { return *(const float4*)(ddepth + (long)b * HW + p); }
};
struct HeadGradAug {
  AugP a;
  const float* gy;
  const float* gsum;
  __device__ __forceinline__ float4 operator()(int b, long p, long HW) const {
    const int W = a.W, Wm1 = a.W - 1;
    const int r = (int)(p / W), q0 = (int)(p - (long)r * W);
    int yy = r, tw = 0;
    if (a.policy & 8) {
      yy = r - a.t_h[b];
      tw = a.t_w[b] % Wm1;
      if (tw < 0) tw += Wm1;
    }
    const bool row_ok = yy >= 0 && yy < a.H;
    int c0 = 0, c1 = 0;
    if ((a.policy & 16) && row_ok) {
      const int r0 = a.o_x[b] - a.cut_h / 2;
      if (yy >= r0 && yy < r0 + a.cut_h) { c0 = a.o_y[b] - a.cut_w / 2; c1 = c0 + a.cut_w; }
    }
    float cc = 1.f, gm = 0.f;
    if (a.policy & 4) {
      const float u = a.u_c[b];
      cc = 1.f + 0.5f * u * u;
      gm = (1.f - cc) * gsum[b] / (float)HW;
    }
    const float* grow = gy + (long)b * HW + (long)(row_ok ? yy : 0) * W;
    float o[4];
    const float gb = grow[W - 1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = q0 + k;
      float g2 = 0.f;
      if (a.policy & 8) {
        int w1 = c - tw;
        if (w1 < 0) w1 += Wm1;
        const float ga = grow[w1];
        if (row_ok && c <= W - 2) {
          if (!(w1 >= c0 && w1 < c1)) g2 += ga;
          if (w1 == 0 && !(W - 1 >= c0 && W - 1 < c1)) g2 += gb;
        }
      } else {
        const float ga = grow[c];
        if (!(c >= c0 && c < c1)) g2 = ga;
      }
      o[k] = (a.policy & 4) ? cc * g2 + gm : g2;
    }
    return make_float4(o[0], o[1], o[2], o[3]);
  }
};

// Four consecutive pixels per thread (HW % 4 == 0): 16-byte loads of every plane, 16-byte stores; `draw` (the planar fp32
// copy) may be null - the bf16 path consumes only the pixel-major copy, and three 8 MB planes were written for nobody.
template <int arch, int CP, typename DD>   // CP: 2 / 4 padded channel count of the pixel-major copy, 0 none
__global__ __launch_bounds__(256) void head_post_bwd4_kernel(const float* __restrict__ gout, const float* __restrict__ noise_pixel,
                                      const float* __restrict__ noise_image, const float* __restrict__ mask,
                                      DD ddepth, float inv_tau, float drop_const, int B, long HW,
                                      float s_depth, float s_conf, float* __restrict__ draw, float* __restrict__ dbias,
                                      bf16* __restrict__ draw_pm, float* __restrict__ bias_ws) {
  __shared__ float red[16];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  const int b = blockIdx.y;
  constexpr int nch = 1 + (arch == 0 ? 0 : arch);
  const float* g = gout + (long)b * nch * HW;
  const float ni = arch == 2 ? noise_image[b] : 0.f;
  auto ld = [](const float* q) { const float4 v = *(const float4*)q; return v; };
  for (long p = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; p < HW; p += (long)gridDim.x * blockDim.x * 4) {
    const long idx = (long)b * HW + p;
    const float4 t4 = ld(g + p), go4 = ddepth(b, p, HW);
    const float t[4] = {t4.x, t4.y, t4.z, t4.w}, go[4] = {go4.x, go4.y, go4.z, go4.w};
    float d0[4], d1[4] = {0.f, 0.f, 0.f, 0.f}, d2[4] = {0.f, 0.f, 0.f, 0.f};
    if (arch == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) d0[q] = go[q] * (1.f - t[q] * t[q]);
    } else {
      const float4 g14 = ld(g + HW + p), np4 = ld(noise_pixel + idx);
      const float l1[4] = {g14.x + np4.x, g14.y + np4.y, g14.z + np4.z, g14.w + np4.w};
      float mp[4], mi[4] = {1.f, 1.f, 1.f, 1.f}, l2[4] = {0.f, 0.f, 0.f, 0.f};
      if (arch == 1) {
        const float4 m4 = ld(mask + idx);
        mp[0] = m4.x; mp[1] = m4.y; mp[2] = m4.z; mp[3] = m4.w;
      } else {
        const float4 m4 = ld(mask + (long)b * 2 * HW + p), i4 = ld(mask + (long)b * 2 * HW + HW + p), g24 = ld(g + 2 * HW + p);
        mp[0] = m4.x; mp[1] = m4.y; mp[2] = m4.z; mp[3] = m4.w;
        mi[0] = i4.x; mi[1] = i4.y; mi[2] = i4.z; mi[3] = i4.w;
        l2[0] = g24.x; l2[1] = g24.y; l2[2] = g24.z; l2[3] = g24.w;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float dt = 1.f - t[q] * t[q];
        const float sp = 1.f / (1.f + __expf(-l1[q] * inv_tau));
        const float dmask = go[q] * (t[q] - drop_const);
        if (arch == 1) {
          d0[q] = mp[q] * go[q] * dt;
          d1[q] = dmask * sp * (1.f - sp) * inv_tau;
        } else {
          const float si = 1.f / (1.f + __expf(-(l2[q] + ni) * inv_tau));
          d0[q] = mp[q] * mi[q] * go[q] * dt;
          d1[q] = dmask * mi[q] * sp * (1.f - sp) * inv_tau;
          d2[q] = dmask * mp[q] * si * (1.f - si) * inv_tau;
        }
      }
    }
    if (draw) {
      float* d = draw + (long)b * nch * HW + p;
      *(float4*)d = make_float4(d0[0] * s_depth, d0[1] * s_depth, d0[2] * s_depth, d0[3] * s_depth);
      if (arch >= 1) *(float4*)(d + HW) = make_float4(d1[0] * s_conf, d1[1] * s_conf, d1[2] * s_conf, d1[3] * s_conf);
      if (arch >= 2) *(float4*)(d + 2 * HW) = make_float4(d2[0] * s_conf, d2[1] * s_conf, d2[2] * s_conf, d2[3] * s_conf);
    }
    if (CP != 0) {
      unsigned w01[4], w2[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const unsigned short h0 = __builtin_bit_cast(unsigned short, (bf16)(d0[q] * s_depth));
        const unsigned short h1 = __builtin_bit_cast(unsigned short, (bf16)(arch >= 1 ? d1[q] * s_conf : 0.f));
        const unsigned short h2 = __builtin_bit_cast(unsigned short, (bf16)(arch >= 2 ? d2[q] * s_conf : 0.f));
        w01[q] = (unsigned)h0 | ((unsigned)h1 << 16);
        w2[q] = (unsigned)h2;
      }
      if (CP == 2) {
        *(uint4*)(draw_pm + idx * 2) = make_uint4(w01[0], w01[1], w01[2], w01[3]);
      } else {
        *(uint4*)(draw_pm + idx * 4) = make_uint4(w01[0], w2[0], w01[1], w2[1]);
        *(uint4*)(draw_pm + idx * 4 + 8) = make_uint4(w01[2], w2[2], w01[3], w2[3]);
      }
    }
    a0 += (d0[0] + d0[1]) + (d0[2] + d0[3]);
    a1 += (d1[0] + d1[1]) + (d1[2] + d1[3]);
    a2 += (d2[0] + d2[1]) + (d2[2] + d2[3]);
  }
  if (dbias) {
    // Atomics on ONE address retire at ~10 ns each (they execute memory-side): a thousand blocks adding straight into
    // dbias[n] cost 10 us per head - more than the pass over the data.  With `bias_ws` (4 KB per sample, zero on entry and
    // left zero) the blocks of a sample add into that sample's slot - B independent addresses - and the last one to
    // arrive (a ticket in the slot) folds the slot into dbias: gridDim.x adds per slot, B per dbias[n].
    const float s0 = dg_block_sum(a0, red);
    const float s1 = arch >= 1 ? dg_block_sum(a1, red) : 0.f;
    const float s2 = arch >= 2 ? dg_block_sum(a2, red) : 0.f;
    if (threadIdx.x == 0) {
      if (bias_ws && gridDim.x > 1) {
        // Round 5: the staged sums are 32.32 FIXED POINT (integer adds commute: the total no longer depends on the order in
        // which blocks and samples arrive - common.h dg_acc_add has the argument).  Two levels, as before: the blocks of a
        // sample add into that sample's slot; the last block of a sample (ticket) moves the slot's totals, still integers,
        // into the launch's accumulators (upper half of slot 0) and takes a second ticket; the last SAMPLE converts and adds
        // each head's total to dbias once.  Everything is left zero.
        unsigned long long* w = (unsigned long long*)(bias_ws + (long)b * 1024);   // 4 KB apart
        unsigned long long* gacc = (unsigned long long*)(bias_ws + 512);            // (slot 0, bytes 2048 ..)
        const float sv[3] = {s0, s1, s2};
        bool odd = false;
#pragma unroll
        for (int h = 0; h <= arch; ++h) {
          if (!(fabsf(sv[h]) < 2147483000.f)) { atomicAdd(&dbias[h], sv[h]); odd = true; }
          else atomicAdd(&w[h], (unsigned long long)__double2ll_rn((double)sv[h] * 4294967296.0));
        }
        (void)odd;
        // the adds above before the ticket: they are acknowledged from memory-side when vmcnt drains (a __threadfence()
        // here is buffer_wbl2 - a write-back of the megabytes of gradient this kernel has just stored, per block)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd((unsigned*)&w[3], 1u) == gridDim.x - 1) {
#pragma unroll
          for (int h = 0; h <= arch; ++h) atomicAdd(&gacc[h], atomicExch(&w[h], 0ull));
          atomicExch((unsigned*)&w[3], 0u);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (atomicAdd((unsigned*)&gacc[3], 1u) == gridDim.y - 1) {
#pragma unroll
            for (int h = 0; h <= arch; ++h)
              atomicAdd(&dbias[h], (float)((double)(long long)atomicExch(&gacc[h], 0ull) * (1.0 / 4294967296.0)));
            atomicExch((unsigned*)&gacc[3], 0u);
          }
        }
      } else {
        atomicAdd(&dbias[0], s0);
        if (arch >= 1) atomicAdd(&dbias[1], s1);
        if (arch >= 2) atomicAdd(&dbias[2], s2);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------------
// Path-length regularisation (trainers/dcgan_amp.py:268-306), the pieces that are not convolutions.
//
// head_post_bwd2_kernel: the SECOND-order part of the head post-processing.  With x = depth output, h = the head conv
// outputs (h0 raw depth, h1 / h2 confidence logits), y the upstream of x and th the forward-mode tangent of h along
// the latent direction v, the tangent of the first-order backward d x / d h_i * y is  y * sum_j H_ij th_j  with H the
// Hessian of x in h as autograd sees it (hard masks carry the straight-through derivative sp' = sp (1 - sp) / tau,
// which is itself differentiable: sp'' = sp' (1 - 2 sp) / tau):
//   H00 = m (-2 t)(1 - t^2)   H01 = mi sp' (1 - t^2)   H02 = mp si' (1 - t^2)
//   H11 = (t - c) mi sp''     H12 = (t - c) sp' si'    H22 = (t - c) mp si''        (t = tanh h0, c = drop_const)
// Outputs as head_post_bwd_kernel: draw2 (scaled by the head's EqualLR scale), the pixel-major bf16 copy, and the
// head-bias gradient sums.
__global__ void head_post_bwd2_kernel(const float* __restrict__ gout, const float* __restrict__ noise_pixel,
                                      const float* __restrict__ noise_image, const float* __restrict__ mask,
                                      const float* __restrict__ ddepth, const float* __restrict__ thead, int arch,
                                      float inv_tau, float drop_const, int B, long HW, float s_depth, float s_conf,
                                      float* __restrict__ draw, float* __restrict__ dbias, bf16* __restrict__ draw_pm,
                                      int cp) {
  __shared__ float red[16];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < (long)B * HW; idx += (long)gridDim.x * blockDim.x) {
    const int b = (int)(idx / HW);
    const long p = idx - (long)b * HW;
    const int nch = 1 + (arch == 0 ? 0 : arch);
    const float* g = gout + (long)b * nch * HW + p;
    const float* th = thead + (long)b * nch * HW + p;
    float* d = draw + (long)b * nch * HW + p;
    const float t = g[0], dt = 1.f - t * t, y = ddepth[idx];
    const float t0 = th[0];
    float d0, d1 = 0.f, d2 = 0.f;
    if (arch == 0) {
      d0 = y * (-2.f * t * dt) * t0;
    } else {
      const float sp = 1.f / (1.f + __expf(-(g[HW] + noise_pixel[idx]) * inv_tau));
      const float sp1 = sp * (1.f - sp) * inv_tau, sp2 = sp1 * (1.f - 2.f * sp) * inv_tau;
      const float t1 = th[HW], tc = t - drop_const;
      if (arch == 1) {
        const float mp = mask[idx];
        d0 = y * (mp * (-2.f * t * dt) * t0 + sp1 * dt * t1);
        d1 = y * (sp1 * dt * t0 + tc * sp2 * t1);
      } else {
        const float mp = mask[(long)b * 2 * HW + p], mi = mask[(long)b * 2 * HW + HW + p];
        const float si = 1.f / (1.f + __expf(-(g[2 * HW] + noise_image[b]) * inv_tau));
        const float si1 = si * (1.f - si) * inv_tau, si2 = si1 * (1.f - 2.f * si) * inv_tau;
        const float t2 = th[2 * HW];
        d0 = y * (mp * mi * (-2.f * t * dt) * t0 + mi * sp1 * dt * t1 + mp * si1 * dt * t2);
        d1 = y * (mi * sp1 * dt * t0 + tc * mi * sp2 * t1 + tc * sp1 * si1 * t2);
        d2 = y * (mp * si1 * dt * t0 + tc * sp1 * si1 * t1 + tc * mp * si2 * t2);
        d[2 * HW] = d2 * s_conf;
      }
      d[HW] = d1 * s_conf;
    }
    d[0] = d0 * s_depth;
    if (draw_pm) {
      bf16* q = draw_pm + idx * cp;
      q[0] = (bf16)(d0 * s_depth);
      if (cp > 1) q[1] = (bf16)(arch >= 1 ? d1 * s_conf : 0.f);
      if (cp > 2) q[2] = (bf16)(arch >= 2 ? d2 * s_conf : 0.f);
      if (cp > 3) q[3] = (bf16)0.f;
    }
    a0 += d0; a1 += d1; a2 += d2;
  }
  if (dbias) {
    const float s0 = dg_block_sum(a0, red);
    if (threadIdx.x == 0) atomicAdd(&dbias[0], s0);
    if (arch >= 1) { const float s1 = dg_block_sum(a1, red); if (threadIdx.x == 0) atomicAdd(&dbias[1], s1); }
    if (arch >= 2) { const float s2 = dg_block_sum(a2, red); if (threadIdx.x == 0) atomicAdd(&dbias[2], s2); }
  }
}

// |J^T y| per sample, the running baseline and the penalty (:294-300), and v = w * d penalty / d dz, the direction of
// the forward-over-reverse pass.  pl_ema (device scalar) is updated in place; acc[0] += baseline, acc[1] += penalty.
// The baseline a = ema + 0.01 (mean l - ema) stays in the graph in the reference (lerp of a live mean), hence the
// second term of  dP/dl_b = (2/B) [(l_b - a) - 0.01 mean_c (l_c - a)].
__global__ __launch_bounds__(256) void pl_penalty_kernel(const float* __restrict__ dz, int B, int K, float w,
                                                         float* __restrict__ pl_ema, float* __restrict__ v,
                                                         float* __restrict__ acc) {
  __shared__ float len[256];
  __shared__ float bc[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int b = wave; b < B; b += 4) {  // one wave per sample
    float s = 0.f;
    for (int k = lane; k < K; k += 64) { const float x = dz[(long)b * K + k]; s += x * x; }
    s = dg_wave_sum(s);
    if (lane == 0) len[b] = sqrtf(s);
  }
  __syncthreads();
  if (tid == 0) {
    float mu = 0.f;
    for (int b = 0; b < B; ++b) mu += len[b];
    mu /= (float)B;
    const float ema = pl_ema[0];
    const float a = ema + 0.01f * (mu - ema);
    float pen = 0.f, dev = 0.f;
    for (int b = 0; b < B; ++b) { const float e = len[b] - a; pen += e * e; dev += e; }
    pl_ema[0] = a;
    acc[0] += a;
    acc[1] += pen / (float)B;
    bc[0] = a;
    bc[1] = dev / (float)B;
  }
  __syncthreads();
  const float a = bc[0], mdev = bc[1];
  for (long i = tid; i < (long)B * K; i += 256) {
    const int b = (int)(i / K);
    const float l = len[b];
    const float dl = (2.f / (float)B) * ((l - a) - 0.01f * mdev);
    v[i] = l > 0.f ? w * dl * dz[i] / l : 0.f;
  }
}

// ----------------------------------------------------------------------------------------------------------
// Per-sample reductions: out[b] = sum_i f(x[b][i]) with f = identity (sq=0) or square (sq=1).  One block per
// (sample, slab); slabs are combined with atomics (out must be zeroed by the caller).
__global__ __launch_bounds__(256) void sample_sum_kernel(const float* __restrict__ x, long n, int sq,
                                                         float* __restrict__ out) {
  __shared__ float red[16];
  const int b = blockIdx.y;
  const float* row = x + (long)b * n;
  float acc = 0.f;
  if ((n & 3) == 0 && (((size_t)row) & 15) == 0) {             // 16-byte loads
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
      const float4 v = *(const float4*)(row + i);
      acc += sq ? v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w : v.x + v.y + v.z + v.w;
    }
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
      const float v = row[i];
      acc += sq ? v * v : v;
    }
  }
  const float s = dg_block_sum(acc, red);
  if (threadIdx.x == 0) dg_acc_add(&out[b], s, gridDim.x, g_det);
}

// ----------------------------------------------------------------------------------------------------------
// DiffAugment (utils/diff_augment.py:114-132, p = 1) on [B,1,H,W] fp32, one fused gather pass.
// policy bits: 1 brightness, 2 saturation (identity for one channel), 4 contrast, 8 translation, 16 cutout.
// Per-sample parameters: u_b,u_c (the uniform(-1,1) draws; the applied factor is u*u),
// t_h,t_w,o_x,o_y ints.  xsum[b] = sum of x[b] (needed by contrast: mean of x + brightness).

__device__ __forceinline__ bool aug_cut(const AugP& a, int b, int y, int x) {
  if (!(a.policy & 16)) return false;
  const int r0 = a.o_x[b] - a.cut_h / 2, c0 = a.o_y[b] - a.cut_w / 2;
  return y >= r0 && y < r0 + a.cut_h && x >= c0 && x < c0 + a.cut_w;
}

// One block per image row (blockIdx.x = row, blockIdx.y = sample): everything that depends on the sample or the row is
// block-uniform, the per-pixel work is 32-bit (the first version decoded a flat 64-bit index per pixel: three 64-bit
// divisions cost more than the pixel's memory traffic - 12.9 us for 16.8 MB).
__global__ __launch_bounds__(256) void diffaug_fwd_kernel(AugP a, const float* __restrict__ x, const float* __restrict__ xsum,
                                                          float* __restrict__ y) {
  const int yy = blockIdx.x, b = blockIdx.y, W = a.W, Wm1 = a.W - 1;
  const long HW = (long)a.H * W;
  float* yrow = y + (long)b * HW + (long)yy * W;
  int sy = yy, tw = 0;
  bool row_ok = true;
  if (a.policy & 8) {
    sy = yy + a.t_h[b];
    row_ok = sy >= 0 && sy < a.H;
    tw = a.t_w[b] % Wm1;
    if (tw < 0) tw += Wm1;                                           // (xx + t_w) mod (W - 1) = xx + tw, minus W - 1 once at most
  }
  int c0 = 0, c1 = 0;                                                // cut-out columns [c0, c1) of this row
  if (a.policy & 16) {
    const int r0 = a.o_x[b] - a.cut_h / 2;
    if (yy >= r0 && yy < r0 + a.cut_h) { c0 = a.o_y[b] - a.cut_w / 2; c1 = c0 + a.cut_w; }
  }
  float br = 0.f, cc = 1.f, mean = 0.f;
  if (a.policy & 1) { const float u = a.u_b[b]; br = 0.5f * u * u; }
  if (a.policy & 4) {
    const float u = a.u_c[b];
    cc = 1.f + 0.5f * u * u;
    mean = xsum[b] / (float)HW + br;
  }
  const float* xrow = x + (long)b * HW + (long)(row_ok ? sy : 0) * W;
#pragma unroll 4
  for (int xx = threadIdx.x; xx < W; xx += 256) {
    int sx = xx;
    if (a.policy & 8) { sx = xx + tw; if (sx >= Wm1) sx -= Wm1; }
    float v = xrow[sx];                                              // always a valid address: loads of the unrolled trips batch
    if (a.policy & 1) v += br;
    if (a.policy & 4) v = mean + cc * (v - mean);
    yrow[xx] = (row_ok && !(xx >= c0 && xx < c1)) ? v : 0.f;
  }
}

// DiffAugment + BlurVH in one pass (utils/diff_augment.py:114-132 -> models/ops/common.py:74-88): the augmented image is
// only ever the discriminator's input, so it is never written - every output pixel evaluates the augmentation at its five
// blur taps straight from the source image.  Up to two source sets in one launch (the D phase's real | fake halves,
// trainers/dcgan_amp.py:199-204): sample b < a[0].B reads set 0, the rest set 1.  Grid (row, sample), 4 pixels per thread.
struct AugSrc { AugP a; const float* x; const float* xsum; int parts; };
// The three source rows of an output row go through LDS with 16-byte loads (the translation wraps columns modulo W - 1,
// so the augmented row is a rotated copy: unaligned - read from LDS, not from global memory, where a first version with
// 14 scalar gathers per 4 pixels ran no faster than the two kernels it replaced).
// Round 6: a block owns a BAND of DAB_ROWS output rows of one sample and stages the DAB_ROWS + 2 source rows it needs once
// (one row per output row before: three staged rows per output row, 4096 short blocks per 64 images, 17 us for 34 MB).
#define DAB_ROWS 4
template <typename T>
__global__ __launch_bounds__(256) void diffaug_blur_fwd_kernel(AugSrc s0, AugSrc s1, T* __restrict__ out, int ring) {
  extern __shared__ float s_rows[];               // [DAB_ROWS + 2][W]: source rows of augmented rows y0 - 1 .. y0 + DAB_ROWS
  const int set = (int)blockIdx.y >= s0.a.B;
  const AugP& a = set ? s1.a : s0.a;
  const float* x = set ? s1.x : s0.x;
  const float* xsum = set ? s1.xsum : s0.xsum;
  const int parts = set ? s1.parts : s0.parts;
  const int b = (int)blockIdx.y - (set ? s0.a.B : 0);
  const int y0 = blockIdx.x * DAB_ROWS, H = a.H, W = a.W, Wm1 = a.W - 1, W4 = a.W >> 2;
  const long HW = (long)H * W;
  int th = 0, tw = 0;
  if (a.policy & 8) {
    th = a.t_h[b];
    tw = a.t_w[b] % Wm1;
    if (tw < 0) tw += Wm1;
  }
  const int r0 = (a.policy & 16) ? a.o_x[b] - a.cut_h / 2 : 0, cl = (a.policy & 16) ? a.o_y[b] - a.cut_w / 2 : 0;
  float br = 0.f, cc = 1.f, mean = 0.f;
  if (a.policy & 1) { const float u = a.u_b[b]; br = 0.5f * u * u; }
  if (a.policy & 4) {
    const float u = a.u_c[b];
    cc = 1.f + 0.5f * u * u;
    float sx = 0.f;
    if (parts > 1) {                              // the producer's partial sums, added in index order (dg_step_prologue_fetch)
      for (int j = 0; j < parts; ++j) sx += xsum[(long)b * parts + j];
    } else sx = xsum[b];
    mean = sx / (float)HW + br;
  }
  // stage: LDS row r holds the source row of augmented row ya = y0 - 1 + r (rows outside the image are never read below; a
  // row the translation moved out of the image is staged as zeros), with brightness and contrast applied where the pixel is
  // STAGED (once per source pixel, not once per tap that reads it): the same two expressions as diffaug_fwd_kernel.
  // The six rows' loads of a thread are issued TOGETHER (a first version staged row after row: hipcc kept each row's
  // load -> arithmetic -> LDS store a loop of its own, six dependent memory round trips per workgroup - 12 of the launch's 17 us).
  const float* srcr[DAB_ROWS + 2];
  bool okr[DAB_ROWS + 2], inr[DAB_ROWS + 2];
#pragma unroll
  for (int r = 0; r < DAB_ROWS + 2; ++r) {
    const int ya = y0 - 1 + r, sy = ya + th;
    inr[r] = ya >= 0 && ya < H;
    okr[r] = sy >= 0 && sy < H;
    srcr[r] = x + (long)b * HW + (long)(okr[r] ? sy : 0) * W;
  }
  for (int q4 = threadIdx.x; q4 < W4; q4 += 256) {
    float4 pq[DAB_ROWS + 2];
#pragma unroll
    for (int r = 0; r < DAB_ROWS + 2; ++r) pq[r] = *(const float4*)(srcr[r] + q4 * 4);   // (unconditional: srcr is always a valid
                                                                                            //  row - a predicated load costs a vmcnt(0))
#pragma unroll
    for (int r = 0; r < DAB_ROWS + 2; ++r) {
      if (!inr[r]) continue;
      float v[4] = {pq[r].x, pq[r].y, pq[r].z, pq[r].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t = v[q] + br;
        t = mean + cc * (t - mean);
        v[q] = okr[r] ? t : 0.f;
      }
      *(float4*)(s_rows + r * W + q4 * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  __syncthreads();
  for (int yy = 0; yy < DAB_ROWS; ++yy) {
    const int y = y0 + yy;
    // the three augmented rows of this output row (reflected at the border): staged row, validity, cut-out columns
    const int yr[3] = {y == 0 ? 1 : y - 1, y, y == H - 1 ? H - 2 : y + 1};
    int c0[3], c1[3];
    const float* rowp[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      rowp[k] = s_rows + (yr[k] - y0 + 1) * W;
      const bool cutrow = (a.policy & 16) && yr[k] >= r0 && yr[k] < r0 + a.cut_h;
      c0[k] = cutrow ? cl : 0;
      c1[k] = cutrow ? cl + a.cut_w : 0;
    }
    auto aug = [&](int k, int xx) {               // augmented image at (row k of the three, column xx)
      int sx = xx + tw;
      if (sx >= Wm1) sx -= Wm1;
      return (xx >= c0[k] && xx < c1[k]) ? 0.f : rowp[k][sx];
    };
    T* orow = out + ((long)blockIdx.y * HW + (long)y * W) * 2;
    for (int q4 = threadIdx.x; q4 < W4; q4 += 256) {
      const int x0 = q4 * 4;
      int xl = x0 - 1, xr = x0 + 4;
      if (ring) { if (xl < 0) xl += W; if (xr >= W) xr -= W; }
      else      { if (xl < 0) xl = 1;  if (xr >= W) xr = W - 2; }
      const float c[6] = {aug(1, xl), aug(1, x0), aug(1, x0 + 1), aug(1, x0 + 2), aug(1, x0 + 3), aug(1, xr)};
      float o[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        o[2 * k] = 0.25f * aug(0, x0 + k) + 0.5f * c[k + 1] + 0.25f * aug(2, x0 + k);
        o[2 * k + 1] = 0.25f * c[k] + 0.5f * c[k + 1] + 0.25f * c[k + 2];
      }
      T* op = orow + x0 * 2;
      if constexpr (sizeof(T) == 2) {
        Vec16<bf16>::store((bf16*)op, o);
      } else {
        *(float4*)op = make_float4(o[0], o[1], o[2], o[3]);
        *(float4*)(op + 4) = make_float4(o[4], o[5], o[6], o[7]);
      }
    }
  }
}

// Backward pass 1: gsum[b] = sum over the augmented image of the gradient that reaches x2 (pre-translation
// image): every (y,x) not cut out and with a valid source row contributes once.  blockIdx.x strides the rows.
__global__ __launch_bounds__(256) void diffaug_bwd_sum_kernel(AugP a, const float* __restrict__ gy,
                                                              float* __restrict__ gsum) {
  __shared__ float red[16];
  const int b = blockIdx.y, W = a.W;
  const long HW = (long)a.H * W;
  const int th = (a.policy & 8) ? a.t_h[b] : 0;
  const int r0 = (a.policy & 16) ? a.o_x[b] - a.cut_h / 2 : 0, cl = (a.policy & 16) ? a.o_y[b] - a.cut_w / 2 : 0;
  float acc = 0.f;
  for (int yy = blockIdx.x; yy < a.H; yy += gridDim.x) {
    if (yy + th < 0 || yy + th >= a.H) continue;
    int c0 = 0, c1 = 0;
    if ((a.policy & 16) && yy >= r0 && yy < r0 + a.cut_h) { c0 = cl; c1 = cl + a.cut_w; }
    const float* row = gy + (long)b * HW + (long)yy * W;
    for (int xx = threadIdx.x; xx < W; xx += 256)
      if (!(xx >= c0 && xx < c1)) acc += row[xx];
  }
  const float s = dg_block_sum(acc, red);
  if (threadIdx.x == 0) dg_acc_add(&gsum[b], s, gridDim.x, g_det);
}

// Backward pass 2 (gather form of the scatter): gx[b,r,c] from gy.  One block per image row, as the forward kernel.
__global__ __launch_bounds__(256) void diffaug_bwd_kernel(AugP a, const float* __restrict__ gy, const float* __restrict__ gsum,
                                                          float* __restrict__ gx) {
  const int r = blockIdx.x, b = blockIdx.y, W = a.W, Wm1 = a.W - 1;
  const long HW = (long)a.H * W;
  float* out = gx + (long)b * HW + (long)r * W;
  int yy = r, tw = 0;
  if (a.policy & 8) {
    yy = r - a.t_h[b];
    tw = a.t_w[b] % Wm1;
    if (tw < 0) tw += Wm1;
  }
  const bool row_ok = yy >= 0 && yy < a.H;
  int c0 = 0, c1 = 0;                                                // cut-out columns [c0, c1) of row yy of the augmented image
  if ((a.policy & 16) && row_ok) {
    const int r0 = a.o_x[b] - a.cut_h / 2;
    if (yy >= r0 && yy < r0 + a.cut_h) { c0 = a.o_y[b] - a.cut_w / 2; c1 = c0 + a.cut_w; }
  }
  float cc = 1.f, gm = 0.f;
  if (a.policy & 4) {
    const float u = a.u_c[b];
    cc = 1.f + 0.5f * u * u;
    gm = (1.f - cc) * gsum[b] / (float)HW;
  }
  const float* grow = gy + (long)b * HW + (long)(row_ok ? yy : 0) * W;
#pragma unroll 4
  for (int c = threadIdx.x; c < W; c += 256) {
    float g2 = 0.f;  // gradient w.r.t. the pre-translation image at (r,c)
    if (a.policy & 8) {
      int w1 = c - tw;                                               // (c - t_w) mod (W - 1)
      if (w1 < 0) w1 += Wm1;
      const float ga = grow[w1], gb = grow[W - 1];                   // (valid addresses whatever the predicates say)
      if (row_ok && c <= W - 2) {
        if (!(w1 >= c0 && w1 < c1)) g2 += ga;
        // columns 0 and W-1 of the output both read source column (t_w mod (W-1))
        if (w1 == 0 && !(W - 1 >= c0 && W - 1 < c1)) g2 += gb;
      }
    } else {
      const float ga = grow[c];
      if (!(c >= c0 && c < c1)) g2 = ga;
    }
    out[c] = (a.policy & 4) ? cc * g2 + gm : g2;
  }
}

// ----------------------------------------------------------------------------------------------------------
// NSGAN losses (models/loss.py:39-41, 68-69) + their gradients w.r.t. the logits; one block.
// scal[0]=mean(y_real) scal[1]=mean(y_fake) scal[2]=loss_D ; dy_* = d(w_gan*loss_D)/dy
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(__expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + __expf(-x)); }

__global__ __launch_bounds__(256) void nsgan_d_kernel(const float* __restrict__ y_real,
                                                      const float* __restrict__ y_fake, int B, float w_gan,
                                                      float* __restrict__ dy_real, float* __restrict__ dy_fake,
                                                      float* __restrict__ scal) {
  __shared__ float red[16];
  float sr = 0.f, sf = 0.f, lr = 0.f, lf = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const float r = y_real[i], f = y_fake[i];
    sr += r; sf += f;
    lr += softplus_f(-r); lf += softplus_f(f);
    dy_real[i] = -w_gan * sigmoid_f(-r) / (float)B;
    dy_fake[i] = w_gan * sigmoid_f(f) / (float)B;
  }
  const float a = dg_block_sum(sr, red), b = dg_block_sum(sf, red);
  const float c = dg_block_sum(lr, red), d = dg_block_sum(lf, red);
  if (threadIdx.x == 0) {
    scal[0] = a / B; scal[1] = b / B; scal[2] = c / B + d / B;
  }
}

// scal[0] = loss_G ; dy = d(w_gan*loss_G)/dy_fake
__global__ __launch_bounds__(256) void nsgan_g_kernel(const float* __restrict__ y_fake, int B, float w_gan,
                                                      float* __restrict__ dy, float* __restrict__ scal) {
  __shared__ float red[16];
  float l = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const float f = y_fake[i];
    l += softplus_f(-f);
    dy[i] = -w_gan * sigmoid_f(-f) / (float)B;
  }
  const float s = dg_block_sum(l, red);
  if (threadIdx.x == 0) scal[0] = s / B;
}

// The D-phase bookkeeping around the loss in one launch (trainers/dcgan_amp.py:203-238): loss + dLoss/dy as
// nsgan_d_kernel, plus the two per-sample vectors the R1 schedule feeds the backward with (up = [1 .. 1 | dy_fake],
// rs = [dy_real | 1 .. 1]; either may be null), the running sums of the logged scalars (acc[0..2] +=) and the
// final bias gradient (dfinal_b += sum dy).  Replaces eight tiny torch kernels per step.
__global__ __launch_bounds__(256) void nsgan_d_step_kernel(const float* __restrict__ y_real,
                                                           const float* __restrict__ y_fake, int B, float w_gan,
                                                           float* __restrict__ dy, float* __restrict__ up,
                                                           float* __restrict__ rs, float* __restrict__ acc,
                                                           float* __restrict__ dfinal_b) {
  __shared__ float red[16];
  float sr = 0.f, sf = 0.f, lr = 0.f, lf = 0.f, sd = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const float r = y_real[i], f = y_fake[i];
    sr += r; sf += f;
    lr += softplus_f(-r); lf += softplus_f(f);
    const float dr = -w_gan * sigmoid_f(-r) / (float)B, df = w_gan * sigmoid_f(f) / (float)B;
    dy[i] = dr; dy[B + i] = df;
    if (up) { up[i] = 1.f; up[B + i] = df; }
    if (rs) { rs[i] = dr; rs[B + i] = 1.f; }
    sd += dr + df;
  }
  const float a = dg_block_sum(sr, red), b = dg_block_sum(sf, red);
  const float c = dg_block_sum(lr, red), d = dg_block_sum(lf, red), e = dg_block_sum(sd, red);
  if (threadIdx.x == 0) {
    acc[0] += a / B; acc[1] += b / B; acc[2] += c / B + d / B;
    if (dfinal_b) dfinal_b[0] += e;
  }
}

// acc[0] += loss_G ; dy = d(w_gan*loss_G)/dy_fake
__global__ __launch_bounds__(256) void nsgan_g_step_kernel(const float* __restrict__ y_fake, int B, float w_gan,
                                                           float* __restrict__ dy, float* __restrict__ acc) {
  __shared__ float red[16];
  float l = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    const float f = y_fake[i];
    l += softplus_f(-f);
    dy[i] = -w_gan * sigmoid_f(-f) / (float)B;
  }
  const float s = dg_block_sum(l, red);
  if (threadIdx.x == 0) acc[0] += s / B;
}

// ----------------------------------------------------------------------------------------------------------
// All seven GANLoss metrics (models/loss.py:39-61 loss_D, :66-85 loss_G) as one kernel.  Every metric is
//   loss = mean_i phi_r(a_i) + mean_i phi_f(b_i),   a = y_real - [rel] mean(y_fake),  b = y_fake - [rel] mean(y_real)
// with phi from a small family; `rel` marks the relativistic-average metrics (average_diff, loss.py:11-18).
//   d loss / d y_real_i = phi_r'(a_i)/B - [rel] mean_j phi_f'(b_j) / B      (and symmetrically for y_fake)
enum { PHI_NONE = 0, PHI_SOFTPLUS = 1, PHI_LINEAR = 2, PHI_SQUARE = 3, PHI_HINGE = 4 };
struct GanForm {
  int kr, kf;        // phi family of the real / fake term
  float sr, sf;      // sign applied to the argument: softplus(s x), s x, relu(1 + s x)
  float cr, cf;      // target of the square: (x - c)^2
  int rel;
};

// value and derivative of phi(kind, s, c) at x
__device__ __forceinline__ void gan_phi(int kind, float s, float c, float x, float& v, float& d) {
  switch (kind) {
    case PHI_SOFTPLUS: v = softplus_f(s * x); d = s * sigmoid_f(s * x); break;
    case PHI_LINEAR: v = s * x; d = s; break;
    case PHI_SQUARE: v = (x - c) * (x - c); d = 2.f * (x - c); break;
    case PHI_HINGE: { const float t = 1.f + s * x; v = t > 0.f ? t : 0.f; d = t > 0.f ? s : 0.f; } break;
    default: v = 0.f; d = 0.f;
  }
}

static int gan_form(int metric, int mode_g, float smoothing, GanForm* f) {
  // rows follow models/loss.py:39-61 (D) and :66-85 (G)
  const GanForm D[7] = {
      {PHI_SOFTPLUS, PHI_SOFTPLUS, -1.f, 1.f, 0.f, 0.f, 0},      // nsgan
      {PHI_LINEAR, PHI_LINEAR, -1.f, 1.f, 0.f, 0.f, 0},          // wgan
      {PHI_SQUARE, PHI_SQUARE, 0.f, 0.f, smoothing, 0.f, 0},     // lsgan (label_real * smoothing, label_fake = 0)
      {PHI_HINGE, PHI_HINGE, -1.f, 1.f, 0.f, 0.f, 0},            // hinge
      {PHI_SOFTPLUS, PHI_SOFTPLUS, -1.f, 1.f, 0.f, 0.f, 1},      // ragan
      {PHI_HINGE, PHI_HINGE, -1.f, 1.f, 0.f, 0.f, 1},            // rahinge
      {PHI_SQUARE, PHI_SQUARE, 0.f, 0.f, 1.f, -1.f, 1}};         // ralsgan
  const GanForm G[7] = {
      {PHI_NONE, PHI_SOFTPLUS, 0.f, -1.f, 0.f, 0.f, 0},          // nsgan
      {PHI_NONE, PHI_LINEAR, 0.f, -1.f, 0.f, 0.f, 0},            // wgan
      {PHI_NONE, PHI_SQUARE, 0.f, 0.f, 0.f, 1.f, 0},             // lsgan (target 1, not smoothed: loss.py:71-72)
      {PHI_NONE, PHI_LINEAR, 0.f, -1.f, 0.f, 0.f, 0},            // hinge
      {PHI_SOFTPLUS, PHI_SOFTPLUS, 1.f, -1.f, 0.f, 0.f, 1},      // ragan
      {PHI_HINGE, PHI_HINGE, 1.f, -1.f, 0.f, 0.f, 1},            // rahinge
      {PHI_SQUARE, PHI_SQUARE, 0.f, 0.f, -1.f, 1.f, 1}};         // ralsgan
  if (metric < 0 || metric > 6) return DG_EUNSUPPORTED;
  *f = mode_g ? G[metric] : D[metric];
  return DG_OK;
}

// One block.  D mode (mode_g = 0): dy = [d/dy_real | d/dy_fake] of w_gan * loss, up / rs / dfinal_b / acc[0..2] as in
// nsgan_d_step_kernel.  G mode: only the fake half carries a gradient (D(real) is data, trainers/dcgan_amp.py:259);
// dy = d(w_gan * loss)/dy_fake, acc[0] += loss; y_real may be null unless the metric is relativistic.
// `red` >= 17 floats of LDS.  dy / up / rs may be LDS (the fused final-layer kernel: every block evaluates the step for
// itself) or global memory; `write_acc`: add the scalars / the final bias gradient (one block only).
__device__ __forceinline__ void gan_step_body(const GanForm& fm, int mode_g, const float* __restrict__ y_real,
                                              const float* __restrict__ y_fake, int B, float w_gan, float* dy, float* up,
                                              float* rs, bool write_acc, float* __restrict__ acc,
                                              float* __restrict__ dfinal_b, float* red) {
  auto bsum = [&](float v) {  // block sum, broadcast to every thread
    const float r = dg_block_sum(v, red);
    __syncthreads();
    if (threadIdx.x == 0) red[16] = r;
    __syncthreads();
    return red[16];
  };
  const bool has_r = (fm.kr != PHI_NONE) || !mode_g;
  float sr = 0.f, sf = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    if (has_r) sr += y_real[i];
    sf += y_fake[i];
  }
  const float mr = bsum(sr) / B, mf = bsum(sf) / B;
  const float offr = fm.rel ? mf : 0.f, offf = fm.rel ? mr : 0.f;
  float lsum = 0.f, dsr = 0.f, dsf = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    float v, d;
    if (fm.kr != PHI_NONE) {
      gan_phi(fm.kr, fm.sr, fm.cr, y_real[i] - offr, v, d);
      lsum += v; dsr += d;
    }
    gan_phi(fm.kf, fm.sf, fm.cf, y_fake[i] - offf, v, d);
    lsum += v; dsf += d;
  }
  const float loss = bsum(lsum) / B;
  const float mdr = bsum(dsr) / B, mdf = bsum(dsf) / B;
  const float k = w_gan / (float)B;
  float sd = 0.f;
  for (int i = threadIdx.x; i < B; i += blockDim.x) {
    float v, d, dr = 0.f;
    if (fm.kr != PHI_NONE) {
      gan_phi(fm.kr, fm.sr, fm.cr, y_real[i] - offr, v, d);
      dr = k * (d - (fm.rel ? mdf : 0.f));
    }
    gan_phi(fm.kf, fm.sf, fm.cf, y_fake[i] - offf, v, d);
    const float df = k * (d - (fm.rel ? mdr : 0.f));
    if (mode_g) {
      dy[i] = df;
    } else {
      dy[i] = dr; dy[B + i] = df;
      if (up) { up[i] = 1.f; up[B + i] = df; }
      if (rs) { rs[i] = dr; rs[B + i] = 1.f; }
      sd += dr + df;
    }
  }
  const float e = dg_block_sum(sd, red);
  if (threadIdx.x == 0 && write_acc) {
    if (mode_g) {
      acc[0] += loss;
    } else {
      acc[0] += mr; acc[1] += mf; acc[2] += loss;
      if (dfinal_b) dfinal_b[0] += e;
    }
  }
}
__global__ __launch_bounds__(256) void gan_step_kernel(GanForm fm, int mode_g, const float* __restrict__ y_real,
                                                       const float* __restrict__ y_fake, int B, float w_gan,
                                                       float* __restrict__ dy, float* __restrict__ up,
                                                       float* __restrict__ rs, float* __restrict__ acc,
                                                       float* __restrict__ dfinal_b) {
  __shared__ float red[17];
  gan_step_body(fm, mode_g, y_real, y_fake, B, w_gan, dy, up, rs, true, acc, dfinal_b, red);
}

// The loss step and the final conv's backward in ONE launch (three in the D phase, two in the G phase otherwise): every
// block evaluates the loss step on the 2B (or B) logits for itself - a few hundred flops - and keeps the per-sample
// vectors in LDS; block 0 also writes them out (later launches read dy / rs) and adds the scalars.  Then the block's
// element range of the final conv's backward-data pass (final_bwd_data_kernel) and, in the same sweep over the samples,
// of its weight gradient dwf[i] += scale * sum_b dy[b] d4[b][i] (dg_batch_wsum) - d4 is read once for both.
// D mode: ns = 2B samples [real | fake]; r1: chain upstream [1 | dy_fake], bias-gradient weights [dy_real | 1] (the R1
// schedule), else upstream dy, weights 1.  G mode: ns = B samples (the fake batch), upstream dy.
template <typename T>
__global__ __launch_bounds__(512) void final_gan_bwd_kernel(GanForm fm, int mode_g, const float* __restrict__ y_real,
                                                            const float* __restrict__ y_fake, int B, float w_gan, int r1,
                                                            float* __restrict__ dy, float* __restrict__ up,
                                                            float* __restrict__ rs, float* __restrict__ acc,
                                                            float* __restrict__ dfinal_b, const T* __restrict__ d4,
                                                            const float* __restrict__ wf, float scale, long n, int C,
                                                            T* __restrict__ dd4, float* __restrict__ dbias,
                                                            float* __restrict__ dwf, float* __restrict__ dbias_part) {
  constexpr int V = Vec16<T>::V;
  constexpr int NW = 8;                           // waves per workgroup (round 6: four left one workgroup per CU with 16-32 KB in flight)
  __shared__ float part[NW][64 * V];
  __shared__ float s_dy[256], s_u[256], s_r[256], red[17];
  const int ns = mode_g ? B : 2 * B;
  gan_step_body(fm, mode_g, y_real, y_fake, B, w_gan, s_dy, r1 ? s_u : nullptr, r1 ? s_r : nullptr, blockIdx.x == 0, acc,
                dfinal_b, red);
  __syncthreads();
  if (blockIdx.x == 0)
    for (int b = threadIdx.x; b < ns; b += 64 * NW) {
      dy[b] = s_dy[b];
      if (up && r1) up[b] = s_u[b];
      if (rs && r1) rs[b] = s_r[b];
    }
  if (!r1) {
    for (int b = threadIdx.x; b < ns; b += 64 * NW) { s_u[b] = s_dy[b]; s_r[b] = 1.f; }
    __syncthreads();
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long i = ((long)blockIdx.x * 64 + lane) * V;
  float w[V], db[V], dw[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { w[k] = 0.f; db[k] = 0.f; dw[k] = 0.f; }
  if (i < n) {
#pragma unroll
    for (int k4 = 0; k4 < V; k4 += 4) {
      const float4 r = *(const float4*)(wf + i + k4);
      w[k4] = r.x * scale; w[k4 + 1] = r.y * scale; w[k4 + 2] = r.z * scale; w[k4 + 3] = r.w * scale;
    }
    // (round 6) eight samples' loads in flight per lane: with four, one workgroup per CU kept 16 KB in flight and the launch
    // ran at 1.8 TB/s of its 34 MB
    constexpr int NB = 8;
    for (int b0 = wave; b0 < ns; b0 += NW * NB) {
      float a[NB][V];
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int b = b0 + NW * j;
        if (b < ns) Vec16<T>::load(d4 + (long)b * n + i, a[j]);
        else {
#pragma unroll
          for (int k = 0; k < V; ++k) a[j][k] = 0.f;
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int b = b0 + NW * j;
        if (b < ns) {
          float g[V];
          const float u = s_u[b], rsb = s_r[b], c = s_dy[b];
#pragma unroll
          for (int k = 0; k < V; ++k) {
            g[k] = u * w[k] * (a[j][k] > 0.f ? SQRT2 : LRELU_SLOPE * SQRT2);
            db[k] += rsb * g[k];
            dw[k] += c * a[j][k];
          }
          Vec16<T>::store(dd4 + (long)b * n + i, g);
        }
      }
    }
  }
  if (dbias) {
#pragma unroll
    for (int k = 0; k < V; ++k) part[wave][lane * V + k] = db[k];
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * V; e += 64 * NW) {
      const long ie = (long)blockIdx.x * 64 * V + e;
      // dbias_part: one partial per element of the map (= per pixel and channel, summed over the samples in a fixed order);
      // the caller sums the n / C pixel rows per channel with dg_wgrad_reduce - no atomics, bit-reproducible
      if (ie < n) {
        float v = part[0][e];
#pragma unroll
        for (int w8 = 1; w8 < NW; ++w8) v += part[w8][e];
        if (dbias_part) dbias_part[ie] = v; else atomicAdd(&dbias[ie % C], v);
      }
    }
  }
  if (dwf) {
    __syncthreads();
#pragma unroll
    for (int k = 0; k < V; ++k) part[wave][lane * V + k] = dw[k];
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * V; e += 64 * NW) {
      const long ie = (long)blockIdx.x * 64 * V + e;
      if (ie < n) {
        float v = part[0][e];
#pragma unroll
        for (int w8 = 1; w8 < NW; ++w8) v += part[w8][e];
        dwf[ie] += v * scale;
      }
    }
  }
}

// acc[0] += mean(x[0..n))   (R1 penalty of the micro-batch: mean of the per-sample squared-gradient sums)
__global__ __launch_bounds__(256) void mean_acc_kernel(const float* __restrict__ x, int n, float* __restrict__ acc) {
  __shared__ float red[16];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
  const float t = dg_block_sum(s, red);
  if (threadIdx.x == 0) acc[0] += t / n;
}

// fetch_reals (trainers/dcgan_amp.py:154-160; utils/lidar.py:31-36; utils/__init__.py:70-73)
__device__ __forceinline__ float fetch_real_px(float pol, float m, float min_d, float max_d, float drop_const) {
  const float depth = pol * (max_d - min_d) + min_d;
  const float disp = 1.f / depth;
  float inv = (disp - 1.f / max_d) / (1.f / min_d - 1.f / max_d);
  inv = inv * 2.f - 1.f;
  return m * inv + (1.f - m) * drop_const;
}
// xsum != nullptr: per-sample sums of the result, one atomic per block of `chunk` pixels (see head_post_fwd_kernel)
// pool_ctr != nullptr: `pol` / `mask` are pools of `npool` batches of n pixels and the batch is *pool_ctr % npool (a
// device-resident loader position: a captured training step replays on the next pooled batch without a copy)
__global__ __launch_bounds__(256) void fetch_reals_kernel(const float* __restrict__ pol, const float* __restrict__ mask,
                                                          float min_d, float max_d, float drop_const, long n,
                                                          float* __restrict__ out, float* __restrict__ xsum, long HW,
                                                          int chunk, const unsigned long long* __restrict__ pool_ctr,
                                                          int npool) {
  __shared__ float red[16];
  if (pool_ctr) {
    const long off = (long)(*pool_ctr % (unsigned long long)npool) * n;
    pol += off;
    mask += off;
  }
  if (!xsum) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fetch_real_px(pol[i], mask[i], min_d, max_d, drop_const);
    return;
  }
  const long i0 = (long)blockIdx.x * chunk;
  float acc = 0.f;
#pragma unroll 4
  for (int k = threadIdx.x; k < chunk; k += 256) {               // (independent pixels: their loads in flight together)
    const long i = i0 + k;
    const float v = fetch_real_px(pol[i], mask[i], min_d, max_d, drop_const);
    out[i] = v;
    acc += v;
  }
  const float sblk = dg_block_sum(acc, red);
  if (threadIdx.x == 0) dg_acc_add(&xsum[i0 / HW], sblk, (unsigned)(HW / chunk), g_det);
}

// y = a * x
__global__ void scale_kernel(const float* __restrict__ x, float a, long n, float* __restrict__ y) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = a * x[i];
}

// logistic noise of GumbelSigmoid (models/dusty.py:30-36) from two uniform fields
__global__ void logistic_noise_kernel(const float* __restrict__ u1, const float* __restrict__ u2, float eps, long n,
                                      float* __restrict__ out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = -logf(logf(u1[i] + eps) / logf(u2[i] + eps) + eps);
}

// ----------------------------------------------------------------------------------------------------------
static inline unsigned nblk(long n, int bs = 256) { return (unsigned)((n + bs - 1) / bs); }
// pixels per block of the kernels that also sum their output per sample: the largest power-of-two multiple of 256 that
// divides HW, at most 4096 (one atomic per block: 16 per 64x1024 sample)
static int sum_chunk(long HW) {
  int c = 256;
  while (c < 4096 && HW % (2 * c) == 0) c *= 2;
  return c;
}

__global__ void dg_zero_kernel(float* __restrict__ p, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = 0.f;
}

// several buffers in one launch (a step's accumulator arena and gradient buffers: one graph node instead of three)
struct ZeroItems { float* p[4]; long n[4]; long first[5]; int k; };
__global__ void dg_zero_multi_kernel(ZeroItems z) {
  const long stride = (long)gridDim.x * blockDim.x;
  const long total = z.first[z.k];
  for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; 4 * i4 < total; i4 += stride) {
    const long i = 4 * i4;
    int j = 0;
#pragma unroll
    for (int q = 1; q < 4; ++q)
      if (q < z.k && i >= z.first[q]) j = q;
    *(float4*)(z.p[j] + (i - z.first[j])) = make_float4(0.f, 0.f, 0.f, 0.f);   // (counts are multiples of 4: the launcher pads down)
  }
}

int dg_zero_f32(float* p, long n, hipStream_t s) {
  if (n <= 0) return DG_OK;
  unsigned g = nblk(n);
  if (g > 2048) g = 2048;
  dg_zero_kernel<<<g, 256, 0, s>>>(p, n);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

template <typename DD>
static int head_post_bwd4_launch(DD dd, const float* gout, const float* noise_pixel, const float* noise_image,
                                 const float* mask, int arch, float tau, float drop_const, int B, long HW, float s_depth,
                                 float s_conf, float* draw, float* dbias, void* draw_pm, int cpk, float* bias_ws,
                                 hipStream_t s) {
  const unsigned per = B >= 1024 ? 1u : (unsigned)(1024 / B);
  unsigned hb4 = nblk(HW / 4);
  if (hb4 > per) hb4 = per;
  const dim3 grid4(hb4, B);
#define DG_HPB4(A, C)                                                                                                    \
  head_post_bwd4_kernel<A, C, DD><<<grid4, 256, 0, s>>>(gout, noise_pixel, noise_image, mask, dd, 1.f / tau, drop_const,   \
                                                        B, HW, s_depth, s_conf, draw, dbias, (bf16*)draw_pm, bias_ws)
#define DG_HPB4_A(A) do { if (cpk == 0) DG_HPB4(A, 0); else if (cpk == 2) DG_HPB4(A, 2); else DG_HPB4(A, 4); } while (0)
  if (arch == 0) DG_HPB4_A(0); else if (arch == 1) DG_HPB4_A(1); else DG_HPB4_A(2);
#undef DG_HPB4_A
#undef DG_HPB4
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

extern "C" {

// k <= 4 fp32 buffers (16-byte aligned, counts multiples of 4) zero-filled by one launch
int dg_zero_multi(float* const* ptrs, const long* counts, int k, void* s_) {
  if (!ptrs || !counts || k < 1 || k > 4) return DG_EINVAL;
  ZeroItems z{};
  long tot = 0;
  for (int i = 0; i < k; ++i) {
    if (!ptrs[i] || counts[i] < 0 || counts[i] % 4 != 0 || ((size_t)ptrs[i] & 15) != 0) return DG_EINVAL;
    z.p[i] = ptrs[i]; z.n[i] = counts[i]; z.first[i] = tot;
    tot += counts[i];
  }
  z.first[k] = tot;
  z.k = k;
  if (tot == 0) return DG_OK;
  unsigned g = nblk(tot / 4);
  if (g > 2048) g = 2048;
  dg_zero_multi_kernel<<<g, 256, 0, (hipStream_t)s_>>>(z);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// dg_blur_fwd + dg_mean_acc(mean_src, mean_n, mean_acc) as one launch (the R1 block: the tangent's BlurVH pass follows the
// kernel that produced the per-sample |g|^2 sums whose mean is logged).  DG_EUNSUPPORTED - nothing launched - unless the
// four-pixel form applies.
int dg_blur_fwd_mean(const float* x, void* out, int dtype, int B, int H, int W, int ring, const float* mean_src, int mean_n,
                     float* mean_acc, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!mean_src || !mean_acc || mean_n < 1) return DG_EINVAL;
  if (!(W % 4 == 0 && W >= 8 && H >= 2 && ((size_t)x & 15) == 0 && ((size_t)out & 15) == 0)) return DG_EUNSUPPORTED;
  if (dtype == DG_BF16) blur_fwd4_kernel<bf16><<<dim3(H, B), 256, 0, s>>>(x, (bf16*)out, B, H, W, ring, mean_src, mean_n, mean_acc);
  else blur_fwd4_kernel<float><<<dim3(H, B), 256, 0, s>>>(x, (float*)out, B, H, W, ring, mean_src, mean_n, mean_acc);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_blur_fwd(const float* x, void* out, int dtype, int B, int H, int W, int ring, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const long n = (long)B * H * W;
  if (W % 4 == 0 && W >= 8 && H >= 2 && ((size_t)x & 15) == 0 && ((size_t)out & 15) == 0) {
    if (dtype == DG_BF16) blur_fwd4_kernel<bf16><<<dim3(H, B), 256, 0, s>>>(x, (bf16*)out, B, H, W, ring, nullptr, 0, nullptr);
    else blur_fwd4_kernel<float><<<dim3(H, B), 256, 0, s>>>(x, (float*)out, B, H, W, ring, nullptr, 0, nullptr);
  } else if (dtype == DG_BF16) blur_fwd_kernel<bf16><<<nblk(n), 256, 0, s>>>(x, (bf16*)out, B, H, W, ring);
  else blur_fwd_kernel<float><<<nblk(n), 256, 0, s>>>(x, (float*)out, B, H, W, ring);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_blur_bwd(const void* d, int dtype, float* dx, int B, int H, int W, int ring, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const long n = (long)B * H * W;
  if (W % 4 == 0 && W >= 8 && H >= 2 && ((size_t)d & 15) == 0 && ((size_t)dx & 15) == 0) {
    if (dtype == DG_BF16) blur_bwd4_kernel<bf16><<<dim3(H, B), 256, 0, s>>>((const bf16*)d, dx, B, H, W, ring, 1.f, nullptr, 1, AugWin{}, 0);
    else blur_bwd4_kernel<float><<<dim3(H, B), 256, 0, s>>>((const float*)d, dx, B, H, W, ring, 1.f, nullptr, 1, AugWin{}, 0);
  } else if (dtype == DG_BF16) blur_bwd_kernel<bf16><<<nblk(n), 256, 0, s>>>((const bf16*)d, dx, B, H, W, ring);
  else blur_bwd_kernel<float><<<nblk(n), 256, 0, s>>>((const float*)d, dx, B, H, W, ring);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

static inline bool vec_ok(const void* p, long n, int dtype) {
  const int V = dtype == DG_BF16 ? 8 : 4;
  return n % V == 0 && ((size_t)p & 15) == 0;
}

static int final_fwd_impl(const void* d4, int dtype, const float* wf, const float* bias, float scale, int B, long n, float* y,
                          bool zero, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (zero) { const int zrc = dg_zero_f32(y, B, s); if (zrc) return zrc; }
  const bool vec = vec_ok(d4, n, dtype) && ((size_t)wf & 15) == 0;
  const int V = vec ? (dtype == DG_BF16 ? 8 : 4) : 1;
  unsigned slabs = nblk(n, 256 * 4 * V);
  if (slabs > 32) slabs = 32;
  if (slabs < 1) slabs = 1;
  if (vec) {
    if (dtype == DG_BF16) final_fwd_kernel<bf16><<<dim3(slabs, B), 256, 0, s>>>((const bf16*)d4, wf, bias, scale, n, y);
    else final_fwd_kernel<float><<<dim3(slabs, B), 256, 0, s>>>((const float*)d4, wf, bias, scale, n, y);
  } else {
    if (dtype == DG_BF16) final_fwd_scalar_kernel<bf16><<<dim3(slabs, B), 256, 0, s>>>((const bf16*)d4, wf, bias, scale, n, y);
    else final_fwd_scalar_kernel<float><<<dim3(slabs, B), 256, 0, s>>>((const float*)d4, wf, bias, scale, n, y);
  }
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// BlurVH adjoint for the R1 chain: dx = oscale * g, ssq[b] += |g_b|^2 (ssq zeroed by the caller); DG_EUNSUPPORTED unless
// W % 4 == 0 and H W % 1024 == 0 (the caller then runs dg_blur_bwd + dg_sample_sum + dg_scale)
int dg_blur_bwd_r1(const void* d, int dtype, float* dx, float oscale, float* ssq, int B, int H, int W, int ring, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!ssq) return DG_EINVAL;
  if (W % 4 != 0 || W < 8 || H < 2 || ((long)H * W) % 1024 != 0 || ((size_t)d & 15) != 0 || ((size_t)dx & 15) != 0)
    return DG_EUNSUPPORTED;
  const int rows_pb = H % 4 == 0 ? 4 : (H % 2 == 0 ? 2 : 1);    // rows per block = per atomic on ssq[b]
  const dim3 grid((H + rows_pb - 1) / rows_pb, B);
  if (dtype == DG_BF16) blur_bwd4_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)d, dx, B, H, W, ring, oscale, ssq, rows_pb, AugWin{}, 0);
  else blur_bwd4_kernel<float><<<grid, 256, 0, s>>>((const float*)d, dx, B, H, W, ring, oscale, ssq, rows_pb, AugWin{}, 0);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// dg_blur_bwd_r1 + dg_blur_fwd_mean as ONE launch: out[b] = BlurVH(oscale * BlurVH^T(d[b])) in `dtype` (the R1 tangent's first
// feature map from the real chain's last gradient map), ssq[b] += |BlurVH^T(d[b])|^2, mean_acc[0] += sum_b of that / mean_n
// (mean_acc optional).  ssq / mean_acc zeroed by the caller.  DG_EUNSUPPORTED - nothing launched - unless W % 4 == 0,
// H % 4 == 0 and the band's six image rows fit 64 KB of LDS.
int dg_blur_r1_tangent(const void* d, int dtype, void* out, float oscale, float* ssq, float* mean_acc, int mean_n, int B, int H,
                       int W, int ring, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!d || !out || !ssq || B <= 0 || (mean_acc && mean_n < 1)) return DG_EINVAL;
  if (dtype != DG_BF16 && dtype != DG_F32) return DG_EINVAL;
  const size_t lds = (size_t)(R1T_ROWS + 2) * W * sizeof(float);
  if (W % 4 != 0 || W < 8 || H < 4 || H % R1T_ROWS != 0 || lds > 60 * 1024 || ((size_t)d & 15) != 0 || ((size_t)out & 15) != 0)
    return DG_EUNSUPPORTED;
  const dim3 grid(H / R1T_ROWS, B);
  if (dtype == DG_BF16) blur_r1_tangent_kernel<bf16><<<grid, 256, lds, s>>>((const bf16*)d, (bf16*)out, H, W, ring, oscale, ssq, mean_acc, mean_n);
  else blur_r1_tangent_kernel<float><<<grid, 256, lds, s>>>((const float*)d, (float*)out, H, W, ring, oscale, ssq, mean_acc, mean_n);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_final_fwd(const void* d4, int dtype, const float* wf, const float* bias, float scale, int B, long n, float* y,
                 void* s_) {
  return final_fwd_impl(d4, dtype, wf, bias, scale, B, n, y, true, s_);
}
// `_acc` forms: the small fp32 accumulator the kernel adds into (y / out / xsum / gsum) was zeroed by the CALLER - a step
// that carves all of them from one buffer zero-fills once instead of once per call (each fill is a ~5 us graph node)
int dg_final_fwd_acc(const void* d4, int dtype, const float* wf, const float* bias, float scale, int B, long n, float* y,
                     void* s_) {
  return final_fwd_impl(d4, dtype, wf, bias, scale, B, n, y, false, s_);
}

int dg_final_bwd_data(const void* d4, int dtype, const float* wf, const float* up, const float* rowscale, float scale,
                      int B, long n, int C, void* dd4, float* dbias, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const int V = dtype == DG_BF16 ? 8 : 4;
  if (vec_ok(d4, n, dtype) && vec_ok(dd4, n, dtype) && C % V == 0 && B <= 256 && ((size_t)wf & 15) == 0) {
    const unsigned grid = nblk(n / V, 64);
    if (dtype == DG_BF16)
      final_bwd_data_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)d4, wf, up, rowscale, scale, B, n, C, (bf16*)dd4, dbias);
    else
      final_bwd_data_kernel<float><<<grid, 256, 0, s>>>((const float*)d4, wf, up, rowscale, scale, B, n, C, (float*)dd4, dbias);
  } else if (dtype == DG_BF16) {
    final_bwd_data_scalar_kernel<bf16><<<nblk(n), 256, 0, s>>>((const bf16*)d4, wf, up, rowscale, scale, B, n, C, (bf16*)dd4, dbias);
  } else {
    final_bwd_data_scalar_kernel<float><<<nblk(n), 256, 0, s>>>((const float*)d4, wf, up, rowscale, scale, B, n, C, (float*)dd4, dbias);
  }
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_batch_wsum(const void* src, int dtype, const float* coef, float scale, int B, long n, float* out,
                  void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const int V = dtype == DG_BF16 ? 8 : 4;
  if (vec_ok(src, n, dtype)) {
    const unsigned grid = nblk(n / V, 64);
    if (dtype == DG_BF16) batch_wsum_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)src, coef, scale, B, n, out);
    else batch_wsum_kernel<float><<<grid, 256, 0, s>>>((const float*)src, coef, scale, B, n, out);
  } else if (dtype == DG_BF16) {
    batch_wsum_scalar_kernel<bf16><<<nblk(n), 256, 0, s>>>((const bf16*)src, coef, scale, B, n, out);
  } else {
    batch_wsum_scalar_kernel<float><<<nblk(n), 256, 0, s>>>((const float*)src, coef, scale, B, n, out);
  }
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

static int head_post_fwd_impl(float* gout, const float* noise_pixel, const float* noise_image, int arch, int training,
                              float tau, float drop_const, int B, long HW, float* mask, float* depth, float* dsum, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (arch < 0 || arch > 2) return DG_EINVAL;
  if (dsum && HW % 256 != 0) return DG_EUNSUPPORTED;
  const int chunk = dsum ? sum_chunk(HW) : 256;
  const unsigned nb = nblk((long)B * HW, chunk);
  if (dsum && chunk % 1024 == 0 && ((size_t)gout & 15) == 0 && ((size_t)depth & 15) == 0 && ((size_t)mask & 15) == 0 &&
      ((size_t)noise_pixel & 15) == 0) {
    if (arch == 0)
      head_post_fwd4_kernel<0><<<nb, 256, 0, s>>>(gout, noise_pixel, noise_image, training, 1.f / tau, drop_const, B, HW, mask, depth, dsum, chunk);
    else if (arch == 1)
      head_post_fwd4_kernel<1><<<nb, 256, 0, s>>>(gout, noise_pixel, noise_image, training, 1.f / tau, drop_const, B, HW, mask, depth, dsum, chunk);
    else
      head_post_fwd4_kernel<2><<<nb, 256, 0, s>>>(gout, noise_pixel, noise_image, training, 1.f / tau, drop_const, B, HW, mask, depth, dsum, chunk);
    HIP_CHECK_RET(hipGetLastError());
    return DG_OK;
  }
  if (arch == 0)
    head_post_fwd_kernel<0><<<nb, 256, 0, s>>>(gout, noise_pixel, noise_image, training, 1.f / tau, drop_const, B, HW, mask, depth, dsum, chunk);
  else if (arch == 1)
    head_post_fwd_kernel<1><<<nb, 256, 0, s>>>(gout, noise_pixel, noise_image, training, 1.f / tau, drop_const, B, HW, mask, depth, dsum, chunk);
  else
    head_post_fwd_kernel<2><<<nb, 256, 0, s>>>(gout, noise_pixel, noise_image, training, 1.f / tau, drop_const, B, HW, mask, depth, dsum, chunk);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
int dg_head_post_fwd(float* gout, const float* noise_pixel, const float* noise_image, int arch, int training,
                     float tau, float drop_const, int B, long HW, float* mask, float* depth, void* s_) {
  return head_post_fwd_impl(gout, noise_pixel, noise_image, arch, training, tau, drop_const, B, HW, mask, depth, nullptr, s_);
}
// ... + dsum[b] += sum of depth[b] (dsum zeroed by the caller; HW % 256 == 0 or DG_EUNSUPPORTED): the per-sample sums
// that dg_diffaug_fwd_pre takes instead of making its own pass over the image
int dg_head_post_fwd_sum(float* gout, const float* noise_pixel, const float* noise_image, int arch, int training,
                         float tau, float drop_const, int B, long HW, float* mask, float* depth, float* dsum, void* s_) {
  if (!dsum) return DG_EINVAL;
  return head_post_fwd_impl(gout, noise_pixel, noise_image, arch, training, tau, drop_const, B, HW, mask, depth, dsum, s_);
}

int dg_head_post_bwd(const float* gout, const float* noise_pixel, const float* noise_image, const float* mask,
                     const float* ddepth, int arch, float tau, float drop_const, int B, long HW, float s_depth,
                     float s_conf, float* draw, float* dbias, void* draw_pm, int cp, float* bias_ws,
                     void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (arch < 0 || arch > 2) return DG_EINVAL;
  unsigned hb = nblk(HW);                        // blocks per sample: ~1024 blocks in all, each one atomic per head
  const unsigned per = B >= 1024 ? 1u : (unsigned)(1024 / B);
  if (hb > per) hb = per;
  const dim3 grid(hb, B);
  const int cpk = !draw_pm ? 0 : (cp == 2 ? 2 : (cp == 4 ? 4 : 1));
  if (!draw && !draw_pm) return DG_EINVAL;
  auto al = [](const void* q) { return ((size_t)q & 15) == 0; };
  if (HW % 4 == 0 && cpk != 1 && al(gout) && al(ddepth) && al(draw) && al(draw_pm) && al(noise_pixel) && al(mask))
    return head_post_bwd4_launch(HeadGradPlain{ddepth}, gout, noise_pixel, noise_image, mask, arch, tau, drop_const, B, HW,
                                 s_depth, s_conf, draw, dbias, draw_pm, cpk, bias_ws, s);
  if (!draw) return DG_EUNSUPPORTED;   // (the scalar kernel always writes the planar copy)
#define DG_HPB(A, C)                                                                                                   \
  head_post_bwd_kernel<A, C><<<grid, 256, 0, s>>>(gout, noise_pixel, noise_image, mask, ddepth, 1.f / tau, drop_const, \
                                                  B, HW, s_depth, s_conf, draw, dbias, (bf16*)draw_pm, cp)
#define DG_HPB_A(A)                                                                                  \
  do {                                                                                               \
    if (cpk == 0) DG_HPB(A, 0); else if (cpk == 2) DG_HPB(A, 2); else if (cpk == 4) DG_HPB(A, 4); else DG_HPB(A, 1); \
  } while (0)
  if (arch == 0) DG_HPB_A(0); else if (arch == 1) DG_HPB_A(1); else DG_HPB_A(2);
#undef DG_HPB_A
#undef DG_HPB
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

static int sample_sum_impl(const float* x, int B, long n, int sq, float* out, bool zero, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (zero) { const int zrc = dg_zero_f32(out, B, s); if (zrc) return zrc; }
  unsigned gx = nblk(n, 256 * 8);
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  sample_sum_kernel<<<dim3(gx, B), 256, 0, s>>>(x, n, sq, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_sample_sum(const float* x, int B, long n, int sq, float* out, void* s_) {
  return sample_sum_impl(x, B, n, sq, out, true, s_);
}
int dg_sample_sum_acc(const float* x, int B, long n, int sq, float* out, void* s_) {
  return sample_sum_impl(x, B, n, sq, out, false, s_);
}

static AugP make_aug(const float* u_b, const float* u_c, const int* t_h, const int* t_w, const int* o_x,
                     const int* o_y, int policy, int B, int H, int W) {
  AugP a;
  a.u_b = u_b; a.u_c = u_c; a.t_h = t_h; a.t_w = t_w; a.o_x = o_x; a.o_y = o_y;
  a.policy = policy; a.B = B; a.H = H; a.W = W;
  a.cut_h = (int)(H * 0.5 + 0.5);  // utils/diff_augment.py:85
  a.cut_w = (int)(W * 0.5 + 0.5);
  return a;
}

// xsum: [B] workspace (per-sample sum of x), y: [B,H,W]
static int diffaug_fwd_impl(const float* x, const float* u_b, const float* u_c, const int* t_h, const int* t_w,
                            const int* o_x, const int* o_y, int policy, int B, int H, int W, float* xsum, float* y,
                            bool zero, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const AugP a = make_aug(u_b, u_c, t_h, t_w, o_x, o_y, policy, B, H, W);
  if (policy & 4) {
    const int rc = sample_sum_impl(x, B, (long)H * W, 0, xsum, zero, s);
    if (rc) return rc;
  }
  diffaug_fwd_kernel<<<dim3(H, B), 256, 0, s>>>(a, x, xsum, y);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_diffaug_fwd(const float* x, const float* u_b, const float* u_c, const int* t_h, const int* t_w,
                   const int* o_x, const int* o_y, int policy, int B, int H, int W, float* xsum, float* y,
                   void* s_) {
  return diffaug_fwd_impl(x, u_b, u_c, t_h, t_w, o_x, o_y, policy, B, H, W, xsum, y, true, s_);
}
int dg_diffaug_fwd_acc(const float* x, const float* u_b, const float* u_c, const int* t_h, const int* t_w,
                       const int* o_x, const int* o_y, int policy, int B, int H, int W, float* xsum, float* y,
                       void* s_) {
  return diffaug_fwd_impl(x, u_b, u_c, t_h, t_w, o_x, o_y, policy, B, H, W, xsum, y, false, s_);
}

// xsum already holds the per-sample sums of x (dg_fetch_reals_sum / dg_head_post_fwd_sum): no pass of its own
int dg_diffaug_fwd_pre(const float* x, const float* u_b, const float* u_c, const int* t_h, const int* t_w,
                       const int* o_x, const int* o_y, int policy, int B, int H, int W, const float* xsum, float* y,
                       void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const AugP a = make_aug(u_b, u_c, t_h, t_w, o_x, o_y, policy, B, H, W);
  diffaug_fwd_kernel<<<dim3(H, B), 256, 0, s>>>(a, x, xsum, y);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// DiffAugment + BlurVH forward for one or two source sets (set k fills samples [k B, (k + 1) B) of `out`); xsum_k = the
// per-sample sums of x_k (dg_fetch_reals_sum / dg_head_post_fwd_sum).  DG_EUNSUPPORTED unless W % 4 == 0.
int dg_diffaug_blur_fwd(const DgAugSet* sets, int nsets, int policy, int B, int H, int W, int ring, void* out, int dtype,
                        void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!sets || nsets < 1 || nsets > 2 || !out || B <= 0) return DG_EINVAL;
  if (W % 4 != 0 || W < 8 || H < 2 || ((size_t)out & 15) != 0) return DG_EUNSUPPORTED;
  for (int k = 0; k < nsets; ++k)
    if (((size_t)sets[k].x & 15) != 0) return DG_EUNSUPPORTED;       // 16-byte row loads
  AugSrc src[2];
  for (int k = 0; k < 2; ++k) {
    const DgAugSet& q = sets[k < nsets ? k : 0];
    if (!q.x || ((policy & 4) && !q.xsum)) return DG_EINVAL;
    src[k].a = make_aug(q.u_b, q.u_c, q.t_h, q.t_w, q.o_x, q.o_y, policy, B, H, W);
    src[k].x = q.x;
    src[k].xsum = q.xsum;
    src[k].parts = q.xsum_parts;
    if (q.xsum_parts < 0 || q.xsum_parts > 256) return DG_EINVAL;
  }
  if (H % DAB_ROWS != 0) return DG_EUNSUPPORTED;                     // bands of DAB_ROWS output rows
  const dim3 grid(H / DAB_ROWS, nsets * B);
  const size_t lds = (size_t)(DAB_ROWS + 2) * W * sizeof(float);
  if (lds > 60 * 1024) return DG_EUNSUPPORTED;
  if (dtype == DG_BF16) diffaug_blur_fwd_kernel<bf16><<<grid, 256, lds, s>>>(src[0], src[1], (bf16*)out, ring);
  else diffaug_blur_fwd_kernel<float><<<grid, 256, lds, s>>>(src[0], src[1], (float*)out, ring);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// BlurVH adjoint that also accumulates what DiffAugment's adjoint needs from its input: gsum[b] += sum of dx[b] over the
// rows / columns whose gradient reaches the source image (gsum zeroed by the caller); then dg_diffaug_bwd_pre.
int dg_blur_bwd_augsum(const void* d, int dtype, float* dx, const int* t_h, const int* o_x, const int* o_y, int policy,
                       float* gsum, int B, int H, int W, int ring, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!gsum) return DG_EINVAL;
  if (W % 4 != 0 || W < 8 || H < 2 || ((size_t)d & 15) != 0 || ((size_t)dx & 15) != 0) return DG_EUNSUPPORTED;
  AugWin w;
  w.t_h = t_h; w.o_x = o_x; w.o_y = o_y; w.policy = policy;
  w.cut_h = (int)(H * 0.5 + 0.5); w.cut_w = (int)(W * 0.5 + 0.5);
  const int rows_pb = H % 4 == 0 ? 4 : (H % 2 == 0 ? 2 : 1);
  const dim3 grid((H + rows_pb - 1) / rows_pb, B);
  if (dtype == DG_BF16) blur_bwd4_kernel<bf16><<<grid, 256, 0, s>>>((const bf16*)d, dx, B, H, W, ring, 1.f, gsum, rows_pb, w, 1);
  else blur_bwd4_kernel<float><<<grid, 256, 0, s>>>((const float*)d, dx, B, H, W, ring, 1.f, gsum, rows_pb, w, 1);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// DiffAugment's adjoint with gsum already made (dg_blur_bwd_augsum): the gather pass only
int dg_diffaug_bwd_pre(const float* gy, const float* u_b, const float* u_c, const int* t_h, const int* t_w,
                       const int* o_x, const int* o_y, int policy, int B, int H, int W, const float* gsum, float* gx,
                       void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const AugP a = make_aug(u_b, u_c, t_h, t_w, o_x, o_y, policy, B, H, W);
  diffaug_bwd_kernel<<<dim3(H, B), 256, 0, s>>>(a, gy, gsum, gx);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// dg_diffaug_bwd_pre + dg_head_post_bwd in one launch: d loss / d depth is DiffAugment's adjoint gather of gy (the BlurVH
// adjoint's output; gsum from dg_blur_bwd_augsum), evaluated per pixel quad where the head post-processing's backward
// needs it - the generator's upstream gradient [B,1,H,W] is never written.  DG_EUNSUPPORTED (nothing launched) unless the
// four-pixel form applies (W % 4 == 0, 16-byte aligned planes, cp 2 / 4 or no pixel-major copy).
int dg_head_post_bwd_aug(const float* gout, const float* noise_pixel, const float* noise_image, const float* mask,
                         const float* gy, const float* u_b, const float* u_c, const int* t_h, const int* t_w,
                         const int* o_x, const int* o_y, int policy, const float* gsum, int arch, float tau,
                         float drop_const, int B, int H, int W, float s_depth, float s_conf, float* draw, float* dbias,
                         void* draw_pm, int cp, float* bias_ws, void* s_) {
  if (arch < 0 || arch > 2 || !gy || B <= 0 || H <= 0 || W <= 1) return DG_EINVAL;
  if (!draw && !draw_pm) return DG_EINVAL;
  if ((policy & 4) && !gsum) return DG_EINVAL;
  const int cpk = !draw_pm ? 0 : (cp == 2 ? 2 : (cp == 4 ? 4 : 1));
  auto al = [](const void* q) { return ((size_t)q & 15) == 0; };
  if (!(W % 4 == 0 && cpk != 1 && al(gout) && al(draw) && al(draw_pm) && al(noise_pixel) && al(mask))) return DG_EUNSUPPORTED;
  HeadGradAug dd{make_aug(u_b, u_c, t_h, t_w, o_x, o_y, policy, B, H, W), gy, gsum};
  return head_post_bwd4_launch(dd, gout, noise_pixel, noise_image, mask, arch, tau, drop_const, B, (long)H * W, s_depth,
                               s_conf, draw, dbias, draw_pm, cpk, bias_ws, (hipStream_t)s_);
}

static int diffaug_bwd_impl(const float* gy, const float* u_b, const float* u_c, const int* t_h, const int* t_w,
                            const int* o_x, const int* o_y, int policy, int B, int H, int W, float* gsum, float* gx,
                            bool zero, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  const AugP a = make_aug(u_b, u_c, t_h, t_w, o_x, o_y, policy, B, H, W);
  if (policy & 4) {
    if (zero) { const int zrc = dg_zero_f32(gsum, B, s); if (zrc) return zrc; }
    unsigned gxn = (unsigned)((H + 3) / 4);                            // a block sums ~4 rows: one atomic per block
    if (gxn > 64) gxn = 64;
    diffaug_bwd_sum_kernel<<<dim3(gxn, B), 256, 0, s>>>(a, gy, gsum);
  }
  diffaug_bwd_kernel<<<dim3(H, B), 256, 0, s>>>(a, gy, gsum, gx);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
int dg_diffaug_bwd(const float* gy, const float* u_b, const float* u_c, const int* t_h, const int* t_w,
                   const int* o_x, const int* o_y, int policy, int B, int H, int W, float* gsum, float* gx,
                   void* s_) {
  return diffaug_bwd_impl(gy, u_b, u_c, t_h, t_w, o_x, o_y, policy, B, H, W, gsum, gx, true, s_);
}
int dg_diffaug_bwd_acc(const float* gy, const float* u_b, const float* u_c, const int* t_h, const int* t_w,
                       const int* o_x, const int* o_y, int policy, int B, int H, int W, float* gsum, float* gx,
                       void* s_) {
  return diffaug_bwd_impl(gy, u_b, u_c, t_h, t_w, o_x, o_y, policy, B, H, W, gsum, gx, false, s_);
}

int dg_nsgan_d(const float* y_real, const float* y_fake, int B, float w_gan, float* dy_real, float* dy_fake,
               float* scal, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  nsgan_d_kernel<<<1, 256, 0, s>>>(y_real, y_fake, B, w_gan, dy_real, dy_fake, scal);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_nsgan_g(const float* y_fake, int B, float w_gan, float* dy, float* scal, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  nsgan_g_kernel<<<1, 256, 0, s>>>(y_fake, B, w_gan, dy, scal);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_fetch_reals(const float* pol, const float* mask, float min_depth, float max_depth, float drop_const, long n,
                   float* out, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  fetch_reals_kernel<<<nblk(n), 256, 0, s>>>(pol, mask, min_depth, max_depth, drop_const, n, out, nullptr, 1, 256, nullptr, 1);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
// ... + xsum[b] += sum of out[b] over its HW pixels (xsum zeroed by the caller; HW % 256 == 0 or DG_EUNSUPPORTED)
int dg_fetch_reals_sum(const float* pol, const float* mask, float min_depth, float max_depth, float drop_const, int B,
                       long HW, float* out, float* xsum, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!xsum) return DG_EINVAL;
  if (HW % 256 != 0) return DG_EUNSUPPORTED;
  const int chunk = sum_chunk(HW);
  fetch_reals_kernel<<<nblk((long)B * HW, chunk), 256, 0, s>>>(pol, mask, min_depth, max_depth, drop_const, (long)B * HW, out,
                                                                xsum, HW, chunk, nullptr, 1);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}
// ... from a device-resident pool of `npool` batches: batch index = *pool_ctr % npool, read on the device
int dg_fetch_reals_pool_sum(const float* pol_pool, const float* mask_pool, const unsigned long long* pool_ctr, int npool,
                            float min_depth, float max_depth, float drop_const, int B, long HW, float* out, float* xsum,
                            void* s_) {
  hipStream_t s = (hipStream_t)s_;
  if (!xsum || !pool_ctr || npool < 1) return DG_EINVAL;
  if (HW % 256 != 0) return DG_EUNSUPPORTED;
  const int chunk = sum_chunk(HW);
  fetch_reals_kernel<<<nblk((long)B * HW, chunk), 256, 0, s>>>(pol_pool, mask_pool, min_depth, max_depth, drop_const,
                                                                (long)B * HW, out, xsum, HW, chunk, pool_ctr, npool);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_scale(const float* x, float a, long n, float* y, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  scale_kernel<<<nblk(n), 256, 0, s>>>(x, a, n, y);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_logistic_noise(const float* u1, const float* u2, float eps, long n, float* out, void* s_) {
  hipStream_t s = (hipStream_t)s_;
  logistic_noise_kernel<<<nblk(n), 256, 0, s>>>(u1, u2, eps, n, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_nsgan_d_step(const float* y_real, const float* y_fake, int B, float w_gan, float* dy, float* up, float* rs,
                    float* acc, float* dfinal_b, void* s_) {
  if (!y_real || !y_fake || !dy || !acc || B <= 0) return DG_EINVAL;
  nsgan_d_step_kernel<<<1, 256, 0, (hipStream_t)s_>>>(y_real, y_fake, B, w_gan, dy, up, rs, acc, dfinal_b);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_nsgan_g_step(const float* y_fake, int B, float w_gan, float* dy, float* acc, void* s_) {
  if (!y_fake || !dy || !acc || B <= 0) return DG_EINVAL;
  nsgan_g_step_kernel<<<1, 256, 0, (hipStream_t)s_>>>(y_fake, B, w_gan, dy, acc);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_gan_d_step(int metric, float smoothing, const float* y_real, const float* y_fake, int B, float w_gan,
                  float* dy, float* up, float* rs, float* acc, float* dfinal_b, void* s_) {
  if (!y_real || !y_fake || !dy || !acc || B <= 0) return DG_EINVAL;
  GanForm fm;
  const int rc = gan_form(metric, 0, smoothing, &fm);
  if (rc != DG_OK) return rc;
  gan_step_kernel<<<1, 256, 0, (hipStream_t)s_>>>(fm, 0, y_real, y_fake, B, w_gan, dy, up, rs, acc, dfinal_b);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_gan_g_step(int metric, const float* y_real, const float* y_fake, int B, float w_gan, float* dy, float* acc,
                  void* s_) {
  if (!y_fake || !dy || !acc || B <= 0) return DG_EINVAL;
  GanForm fm;
  const int rc = gan_form(metric, 1, 1.f, &fm);
  if (rc != DG_OK) return rc;
  if (fm.kr != PHI_NONE && !y_real) return DG_EINVAL;  // relativistic metrics read D(real) (models/loss.py:76-85)
  gan_step_kernel<<<1, 256, 0, (hipStream_t)s_>>>(fm, 1, y_real, y_fake, B, w_gan, dy, nullptr, nullptr, acc,
                                                  nullptr);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

// dg_gan_d_step / dg_gan_g_step + dg_final_bwd_data (+ dg_batch_wsum with coef = dy when dwf is given) in one launch.
// DG_EUNSUPPORTED (nothing launched) unless the 16-byte forms apply and the samples fit the kernel's per-sample tables:
// the caller then issues the separate calls.
int dg_final_gan_bwd(int metric, int mode_g, float smoothing, const float* y_real, const float* y_fake, int B, float w_gan,
                     int r1, float* dy, float* up, float* rs, float* acc, float* dfinal_b, const void* d4, int dtype,
                     const float* wf, float scale, long n, int C, void* dd4, float* dbias, float* dwf, float* dbias_part,
                     void* s_) {
  if (!y_fake || !dy || !acc || !d4 || !wf || !dd4 || B <= 0 || n <= 0) return DG_EINVAL;
  if (dbias_part && (!dbias || n % C != 0)) return DG_EINVAL;
  if (!mode_g && !y_real) return DG_EINVAL;
  GanForm fm;
  const int rc = gan_form(metric, mode_g ? 1 : 0, mode_g ? 1.f : smoothing, &fm);
  if (rc != DG_OK) return rc;
  if (mode_g && fm.kr != PHI_NONE && !y_real) return DG_EINVAL;
  if (mode_g && r1) return DG_EINVAL;
  const int ns = mode_g ? B : 2 * B;
  const int V = dtype == DG_BF16 ? 8 : 4;
  if (!(vec_ok(d4, n, dtype) && vec_ok(dd4, n, dtype) && C % V == 0 && ns <= 256 && ((size_t)wf & 15) == 0 &&
        (!dwf || ((size_t)dwf & 15) == 0)))
    return DG_EUNSUPPORTED;
  const unsigned grid = nblk(n / V, 64);
  hipStream_t s = (hipStream_t)s_;
  if (dtype == DG_BF16)
    final_gan_bwd_kernel<bf16><<<grid, 512, 0, s>>>(fm, mode_g ? 1 : 0, y_real, y_fake, B, w_gan, r1, dy, up, rs, acc, dfinal_b,
                                                    (const bf16*)d4, wf, scale, n, C, (bf16*)dd4, dbias, dwf, dbias_part);
  else
    final_gan_bwd_kernel<float><<<grid, 512, 0, s>>>(fm, mode_g ? 1 : 0, y_real, y_fake, B, w_gan, r1, dy, up, rs, acc, dfinal_b,
                                                     (const float*)d4, wf, scale, n, C, (float*)dd4, dbias, dwf, dbias_part);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_head_post_bwd2(const float* gout, const float* noise_pixel, const float* noise_image, const float* mask,
                      const float* ddepth, const float* thead, int arch, float tau, float drop_const, int B, long HW,
                      float s_depth, float s_conf, float* draw, float* dbias, void* draw_pm, int cp, void* s_) {
  if (!gout || !ddepth || !thead || !draw || B <= 0 || HW <= 0 || arch < 0 || arch > 2) return DG_EINVAL;
  if (arch >= 1 && (!noise_pixel || !mask)) return DG_EINVAL;
  if (arch == 2 && !noise_image) return DG_EINVAL;
  const long n = (long)B * HW;
  const int hb = (int)min((long)1024, (n + 255) / 256);
  head_post_bwd2_kernel<<<hb, 256, 0, (hipStream_t)s_>>>(gout, noise_pixel, noise_image, mask, ddepth, thead, arch,
                                                         1.f / tau, drop_const, B, HW, s_depth, s_conf, draw, dbias,
                                                         (bf16*)draw_pm, cp);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_pl_penalty(const float* dz, int B, int K, float w, float* pl_ema, float* v, float* acc, void* s_) {
  if (!dz || !pl_ema || !v || !acc || B <= 0 || B > 256 || K <= 0) return DG_EINVAL;
  pl_penalty_kernel<<<1, 256, 0, (hipStream_t)s_>>>(dz, B, K, w, pl_ema, v, acc);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_mean_acc(const float* x, int n, float* acc, void* s_) {
  if (!x || !acc || n <= 0) return DG_EINVAL;
  mean_acc_kernel<<<1, 256, 0, (hipStream_t)s_>>>(x, n, acc);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

}  // extern "C"


// ---- deterministic sums: the accumulator arena and its shadow (common.h dg_acc_add) ----------------------------------------
extern "C" int dg_det_arena(float* arena, long n, void* shadow) {
  if ((arena == nullptr) != (shadow == nullptr) || n < 0) return DG_EINVAL;
  if (shadow && ((size_t)shadow & 15) != 0) return DG_EINVAL;
  const DgDet d{arena, (unsigned long long*)shadow, arena ? n : 0};
  HIP_CHECK_RET(hipMemcpyToSymbol(HIP_SYMBOL(g_det), &d, sizeof(d), 0, hipMemcpyHostToDevice));
  return DG_OK;
}
