// Point-cloud kernels of the validation metrics (the reference's native CUDA extensions), gfx950:
//   * fps_kernel            : iterative furthest point sampling, one workgroup per cloud.
//                             Reference: utils/sampling/fps/furthest_point_sampling.cu:97-207 (+ gather :38-60).
//   * chamfer_dir_kernel    : ALL-PAIRS directed Chamfer means  L[i][j] = mean_{p in A_i} min_{q in B_j} |p - q|^2.
//                             Reference: utils/metrics/distance/cd/chamfer_distance.{cu,cpp} (nnsearch :41-66) as driven by
//                             utils/metrics/cov_mmd_1nna.py:20-52, which loops i in Python and calls the extension on
//                             (cloud i expanded) x (512 clouds) -- N^2 / 512 launches.  Here one launch fills a whole
//                             [Na,Nb] matrix; CD(i,j) = L_AB[i][j] + L_BA[j][i].
// Both are VALU-bound fp32 work (K = 3: no matrix-core shape): the Chamfer kernel keeps PT points of cloud A per lane in
// registers, streams cloud B through LDS with broadcast 16-byte reads and evaluates two points per packed instruction.
#include "common.h"

namespace {

typedef float f2 __attribute__((ext_vector_type(2)));

// --------------------------------------------------------------------------------------------------------------
// FPS.  Selection rule of the reference, including its tie-breaking: every thread t scans k = t, t + T, ... keeping the
// FIRST strictly-greater candidate (T = opt_n_threads(n), a power of two <= 512); the tree reduction then folds slot
// t + w into slot t for w = T/2 .. 1 keeping the lower SLOT on ties, so two threads meet at the lowest bit in which
// their ids differ and the one with that bit clear wins.  Among equal maxima the winner therefore minimises
// (bit-reversed (k mod T), k); `tie_mod` carries T.
// Points with |p|^2 <= 1e-3 (dropped returns at the origin) never become candidates (:132-134).
struct Cand {
  float v;
  int k;
};
__device__ __forceinline__ bool cand_better(const Cand& a, const Cand& b, int tie_mod) {
  if (a.v != b.v) return a.v > b.v;
  const unsigned ra = __brev((unsigned)(a.k & (tie_mod - 1))), rb = __brev((unsigned)(b.k & (tie_mod - 1)));
  return ra != rb ? ra < rb : a.k < b.k;
}

__global__ __launch_bounds__(1024) void fps_kernel(const float* __restrict__ xyz, int n, int m, int tie_mod,
                                                   float* __restrict__ temp, int* __restrict__ idx,
                                                   float* __restrict__ out) {
#pragma clang fp contract(off)
  __shared__ Cand red[16];
  __shared__ int s_old;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  xyz += (long)blockIdx.x * n * 3;
  temp += (long)blockIdx.x * n;
  idx += (long)blockIdx.x * m;
  if (out) out += (long)blockIdx.x * m * 3;
  for (int k = tid; k < n; k += blockDim.x) temp[k] = 1e10f;  // furthest_point_sampling.cpp: torch::full(1e10)
  int old = 0;
  if (tid == 0) {
    idx[0] = 0;
    if (out) { out[0] = xyz[0]; out[1] = xyz[1]; out[2] = xyz[2]; }
  }
  __syncthreads();
  for (int j = 1; j < m; ++j) {
    const float x1 = xyz[old * 3 + 0], y1 = xyz[old * 3 + 1], z1 = xyz[old * 3 + 2];
    Cand best = {-1.f, 0};
    for (int k = tid; k < n; k += blockDim.x) {
      const float x2 = xyz[k * 3 + 0], y2 = xyz[k * 3 + 1], z2 = xyz[k * 3 + 2];
      const float mag = (x2 * x2) + (y2 * y2) + (z2 * z2);
      if (mag <= 1e-3f) continue;
      const float d = (x2 - x1) * (x2 - x1) + (y2 - y1) * (y2 - y1) + (z2 - z1) * (z2 - z1);
      const float d2 = fminf(d, temp[k]);
      temp[k] = d2;
      const Cand c = {d2, k};
      if (cand_better(c, best, tie_mod)) best = c;
    }
    // wave reduction, then across waves
    for (int o = 32; o > 0; o >>= 1) {
      Cand other = {__shfl_xor(best.v, o, 64), __shfl_xor(best.k, o, 64)};
      if (cand_better(other, best, tie_mod)) best = other;
    }
    if (lane == 0) red[wave] = best;
    __syncthreads();
    if (tid == 0) {
      Cand b = red[0];
      for (int w = 1; w < nw; ++w)
        if (cand_better(red[w], b, tie_mod)) b = red[w];
      // every candidate skipped (all points at the origin): the reference's besti stays 0
      s_old = b.v < 0.f ? 0 : b.k;
      idx[j] = s_old;
    }
    __syncthreads();
    old = s_old;
    if (out && tid < 3) out[j * 3 + tid] = xyz[old * 3 + tid];
  }
}

// --------------------------------------------------------------------------------------------------------------
// Directed Chamfer means, all pairs.
// Work split: a WAVE owns 512 consecutive points of one cloud A_i (8 per lane, as 4 packed pairs, in registers for the
// whole kernel) and their running minima; a workgroup is 4 waves = 4 clouds (n <= 512), 2 clouds (n <= 1024) or a
// 2048-point slice of one cloud (larger n: blockIdx.z walks the slices and the partial means are added atomically).
// The workgroup streams TB clouds B_j through LDS in chunks of MC points (x, y, z as 16 bytes, double-buffered; the
// next chunk's global loads are issued before the math and stored to LDS after it); every lane reads the SAME LDS
// address (broadcast).  Per point pair: 6 packed fp32 instructions per two pairs + one v_min each = 4 lane-ops.
constexpr int CH_THREADS = 256;
constexpr int CH_PT = 8;                 // points of A per lane
constexpr int CH_WPTS = 64 * CH_PT;      // points of A per wave
constexpr int CH_MC = 512;               // points of B per LDS chunk (8 KB), two buffers
constexpr int CH_LD = CH_MC / CH_THREADS;  // chunk points loaded per thread

__global__ __launch_bounds__(CH_THREADS) void chamfer_dir_kernel(const float* __restrict__ A, int Na, int n, int wpc,
                                                                 const float* __restrict__ Bc, int Nb, int m, int TB,
                                                                 float* __restrict__ L) {
  __shared__ float4 qbuf[2][CH_MC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // wpc = waves per cloud (1, 2 or a multiple of 4).  Which cloud / slice does this wave own?
  int i, slice;
  if (wpc >= 4) { i = blockIdx.y; slice = blockIdx.z * 4 + wave; }
  else { i = blockIdx.y * (4 / wpc) + wave / wpc; slice = wave % wpc; }
  const bool live = i < Na;
  const int p0 = slice * CH_WPTS + lane * CH_PT;
  f2 px[CH_PT / 2], py[CH_PT / 2], pz[CH_PT / 2];
  const float* a = A + (long)(live ? i : 0) * n * 3;
#pragma unroll
  for (int h = 0; h < CH_PT / 2; ++h) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int p = p0 + 2 * h + e;
      const int pp = p < n ? p : 0;  // out-of-range slots replicate point 0; their minima are dropped from the sum
      px[h][e] = a[pp * 3 + 0];
      py[h][e] = a[pp * 3 + 1];
      pz[h][e] = a[pp * 3 + 2];
    }
  }
  const int j0 = blockIdx.x * TB, j1 = min(Nb, j0 + TB);
  const int nchunk = (m + CH_MC - 1) / CH_MC;
  const int total = (j1 - j0) * nchunk;  // (cloud, chunk) steps of this workgroup
  float r[CH_LD][3];
  auto fetch = [&](int step) {
    const int j = j0 + step / nchunk, c = step % nchunk;
    const float* q = Bc + (long)j * m * 3;
#pragma unroll
    for (int u = 0; u < CH_LD; ++u) {
      const int k = c * CH_MC + u * CH_THREADS + tid;
      const int kk = k < m ? k : 0;  // the tail of the last chunk repeats the cloud's first point: no effect on a minimum
      r[u][0] = q[kk * 3 + 0]; r[u][1] = q[kk * 3 + 1]; r[u][2] = q[kk * 3 + 2];
    }
  };
  auto commit = [&](int buf) {
#pragma unroll
    for (int u = 0; u < CH_LD; ++u) qbuf[buf][u * CH_THREADS + tid] = make_float4(r[u][0], r[u][1], r[u][2], 0.f);
  };
  f2 rmin[CH_PT / 2];
  if (total > 0) { fetch(0); commit(0); }
  __syncthreads();
  for (int step = 0; step < total; ++step) {
    const int buf = step & 1, c = step % nchunk;
    if (c == 0) {
#pragma unroll
      for (int h = 0; h < CH_PT / 2; ++h) rmin[h] = (f2){3.0e38f, 3.0e38f};
    }
    if (step + 1 < total) fetch(step + 1);
#pragma unroll 2
    for (int t = 0; t < CH_MC; ++t) {
      const float4 q = qbuf[buf][t];
      const f2 qx = (f2){q.x, q.x}, qy = (f2){q.y, q.y}, qz = (f2){q.z, q.z};
#pragma unroll
      for (int h = 0; h < CH_PT / 2; ++h) {
        const f2 dx = qx - px[h], dy = qy - py[h], dz = qz - pz[h];
        f2 d = dx * dx;
        d = __builtin_elementwise_fma(dy, dy, d);
        d = __builtin_elementwise_fma(dz, dz, d);
        rmin[h][0] = __builtin_fminf(rmin[h][0], d[0]);
        rmin[h][1] = __builtin_fminf(rmin[h][1], d[1]);
      }
    }
    if (c == nchunk - 1) {  // cloud B_j finished: this wave's share of mean_p min_q
      float s = 0.f;
#pragma unroll
      for (int h = 0; h < CH_PT / 2; ++h) {
        if (p0 + 2 * h < n) s += rmin[h][0];
        if (p0 + 2 * h + 1 < n) s += rmin[h][1];
      }
      s = dg_wave_sum(s);
      if (lane == 0 && live && slice * CH_WPTS < n) {
        float* dst = &L[(long)i * Nb + j0 + step / nchunk];
        if (wpc == 1) *dst = s / (float)n;
        else atomicAdd(dst, s / (float)n);
      }
    }
    if (step + 1 < total) commit(buf ^ 1);
    __syncthreads();
  }
}

// --------------------------------------------------------------------------------------------------------------
// JSD occupancy histogram (utils/metrics/jsd.py:24-79): every point votes for its nearest node of the in-sphere unit
// grid -- brute force like the reference (argmin over all nodes, first index on ties), nodes streamed through LDS.
constexpr int GV_CHUNK = 2048;

__global__ __launch_bounds__(256) void grid_vote_kernel(const float* __restrict__ pts, long P,
                                                        const float* __restrict__ grid, int Ng,
                                                        float* __restrict__ counters) {
#pragma clang fp contract(off)
  __shared__ float4 g[GV_CHUNK];
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = p < P;
  const float x = ok ? pts[p * 3 + 0] : 0.f, y = ok ? pts[p * 3 + 1] : 0.f, z = ok ? pts[p * 3 + 2] : 0.f;
  float best = 3.0e38f;
  int bi = 0;
  for (int c0 = 0; c0 < Ng; c0 += GV_CHUNK) {
    const int cn = min(GV_CHUNK, Ng - c0);
    __syncthreads();
    for (int t = threadIdx.x; t < cn; t += blockDim.x)
      g[t] = make_float4(grid[(c0 + t) * 3 + 0], grid[(c0 + t) * 3 + 1], grid[(c0 + t) * 3 + 2], 0.f);
    __syncthreads();
    for (int t = 0; t < cn; ++t) {
      const float4 q = g[t];
      const float dx = x - q.x, dy = y - q.y, dz = z - q.z;
      const float d = (dx * dx + dy * dy) + dz * dz;
      if (d < best) { best = d; bi = c0 + t; }  // strict: the first minimum wins, as torch.argmin
    }
  }
  if (ok) atomicAdd(&counters[bi], 1.f);
}

// _jensen_shannon_divergence (jsd.py:96-107), base-2 entropies with the reference's eps handling: `_entropy` adds 1e-8
// IN PLACE, so the normalised P and Q carry it into the mixture term, which then gets its own 1e-8.
__global__ __launch_bounds__(1024) void jsd_kernel(const float* __restrict__ P, const float* __restrict__ Q, int n,
                                                   float* __restrict__ out) {
  __shared__ float red[16];
  __shared__ float bc[2];
  float sp = 0.f, sq = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { sp += P[i]; sq += Q[i]; }
  const float tp = dg_block_sum(sp, red);
  const float tq = dg_block_sum(sq, red);
  if (threadIdx.x == 0) { bc[0] = tp; bc[1] = tq; }
  __syncthreads();
  const float SP = bc[0], SQ = bc[1];
  float e1 = 0.f, e2 = 0.f, es = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float p = P[i] / SP + 1e-8f, q = Q[i] / SQ + 1e-8f;
    const float mix = (p + q) / 2.f + 1e-8f;
    e1 -= p * log2f(p); e2 -= q * log2f(q); es -= mix * log2f(mix);
  }
  const float a = dg_block_sum(e1, red), b = dg_block_sum(e2, red), c = dg_block_sum(es, red);
  if (threadIdx.x == 0) out[0] = c - (a + b) / 2.f;
}

// --------------------------------------------------------------------------------------------------------------
// SWD descriptors (utils/metrics/swd.py:23-68): Laplacian pyramid levels and patch gathering.
// Gaussian taps [1,4,6,4,1] (x) [1,4,6,4,1] / 256 with reflect padding 2 (get_kernel :16-21, pyramid_down :24-30).
__device__ __forceinline__ int reflect_idx(int i, int n) {
  i = i < 0 ? -i : i;
  return i >= n ? 2 * n - 2 - i : i;
}
__device__ __constant__ float GAUSS5[5] = {1.f / 16.f, 4.f / 16.f, 6.f / 16.f, 4.f / 16.f, 1.f / 16.f};

// out [P,H/2,W/2] = gaussian 5x5, stride 2, of reflect-padded in [P,H,W]   (P = B*C planes)
__global__ __launch_bounds__(256) void pyr_down_kernel(const float* __restrict__ in, long planes, int H, int W,
                                                       float* __restrict__ out) {
  const int Ho = H / 2, Wo = W / 2;
  const long n = planes * Ho * Wo;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
  const float* src = in + (i / ((long)Wo * Ho)) * H * W;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    const int yy = reflect_idx(2 * y + a - 2, H);
    float row = 0.f;
#pragma unroll
    for (int b = 0; b < 5; ++b) row += GAUSS5[b] * src[(long)yy * W + reflect_idx(2 * x + b - 2, W)];
    acc += GAUSS5[a] * row;
  }
  out[i] = acc;
}

// fine [P,H,W] -= pyramid_up(coarse [P,H/2,W/2])   (pyramid_up :33-42, laplacian_pyramid :45-50): zero-insertion
// puts coarse(i,j) at the ODD position (2i+1, 2j+1) of an H x W plane (the transposed conv's centre tap, last row /
// column cropped), then reflect pad 2 and the 5x5 gaussian times 4.
__global__ __launch_bounds__(256) void pyr_up_sub_kernel(float* __restrict__ fine, const float* __restrict__ coarse,
                                                         long planes, int H, int W) {
  const long n = planes * H * W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W), y = (int)((i / W) % H);
  const int Hc = H / 2, Wc = W / 2;
  const float* src = coarse + (i / ((long)W * H)) * Hc * Wc;
  float acc = 0.f;
#pragma unroll
  for (int a = 0; a < 5; ++a) {
    const int yy = reflect_idx(y + a - 2, H);
    if (!(yy & 1)) continue;
    float row = 0.f;
#pragma unroll
    for (int b = 0; b < 5; ++b) {
      const int xx = reflect_idx(x + b - 2, W);
      if (xx & 1) row += GAUSS5[b] * src[(long)(yy >> 1) * Wc + (xx >> 1)];
    }
    acc += GAUSS5[a] * row;
  }
  fine[i] -= 4.f * acc;
}

// out [B,NP,C,ph,pw] = the patches of img [B,C,H,W] whose top-left corners are inds[k] = y * (W - pw + 1) + x
// (extract_patches :53-62: unfold + index_select; the same positions for every image of the minibatch)
__global__ __launch_bounds__(256) void patches_kernel(const float* __restrict__ img, int B, int C, int H, int W, int ph,
                                                      int pw, const long* __restrict__ inds, int NP,
                                                      float* __restrict__ out) {
  const long n = (long)B * NP * C * ph * pw;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int dx = (int)(i % pw), dy = (int)((i / pw) % ph);
  const int c = (int)((i / ((long)pw * ph)) % C);
  const int k = (int)((i / ((long)pw * ph * C)) % NP);
  const int b = (int)(i / ((long)pw * ph * C * NP));
  const long pos = inds[k];
  const int nW = W - pw + 1;
  const int y = (int)(pos / nW) + dy, x = (int)(pos % nW) + dx;
  out[i] = img[(((long)b * C + c) * H + y) * W + x];
}

// --------------------------------------------------------------------------------------------------------------
// Earth mover's distance, approximate matching (utils/metrics/distance/emd/earth_mover_distance.cu:28-190 approxmatch,
// :218-262 matchcost): ten annealing levels exp(-4^j d^2), j = 7 .. -1, then 0; per level the three sweeps of the
// reference (left ratios, right consumption, matched mass).  The match matrix is never stored: matchcost is linear in
// it, so the cost sum_{k,l} match[l][k] d^2(k,l) is accumulated while the mass is produced.  One workgroup per cloud
// pair, both clouds and the four mass vectors in LDS.  (i, j) = (blockIdx.y, blockIdx.x), or the diagonal pairing
// (i = j) when `paired`.
__global__ __launch_bounds__(512) void emd_kernel(const float* __restrict__ A, int n, const float* __restrict__ Bc,
                                                  int m, int Nb, int paired, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char emd_lds[];
  float4* p1 = (float4*)emd_lds;            // [n]
  float4* p2 = p1 + n;                      // [m]
  float* remainL = (float*)(p2 + m);        // [n]
  float* remainR = remainL + n;             // [m]
  float* ratioL = remainR + m;              // [n]
  float* ratioR = ratioL + n;               // [m]
  __shared__ float red[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int j = blockIdx.x, i = paired ? blockIdx.x : blockIdx.y;
  const float* a = A + (long)i * n * 3;
  const float* b = Bc + (long)j * m * 3;
  const float multiL = n >= m ? 1.f : (float)(m / n), multiR = n >= m ? (float)(n / m) : 1.f;
  for (int k = tid; k < n; k += nt) { p1[k] = make_float4(a[k * 3], a[k * 3 + 1], a[k * 3 + 2], 0.f); remainL[k] = multiL; }
  for (int l = tid; l < m; l += nt) { p2[l] = make_float4(b[l * 3], b[l * 3 + 1], b[l * 3 + 2], 0.f); remainR[l] = multiR; }
  __syncthreads();
  float cost = 0.f;
  for (int lev = 7; lev >= -2; --lev) {
    const float level = lev == -2 ? 0.f : -powf(4.0f, (float)lev);
    for (int k = tid; k < n; k += nt) {
      const float4 x = p1[k];
      float suml = 1e-9f;
      for (int l = 0; l < m; ++l) {
        const float4 q = p2[l];
        const float d = (q.x - x.x) * (q.x - x.x) + (q.y - x.y) * (q.y - x.y) + (q.z - x.z) * (q.z - x.z);
        suml += __expf(level * d) * remainR[l];
      }
      ratioL[k] = remainL[k] / suml;
    }
    __syncthreads();
    for (int l = tid; l < m; l += nt) {
      const float4 q = p2[l];
      float sumr = 0.f;
      for (int k = 0; k < n; ++k) {
        const float4 x = p1[k];
        const float d = (q.x - x.x) * (q.x - x.x) + (q.y - x.y) * (q.y - x.y) + (q.z - x.z) * (q.z - x.z);
        sumr += __expf(level * d) * ratioL[k];
      }
      const float r = remainR[l];
      sumr *= r;
      const float consumption = fminf(r / (sumr + 1e-9f), 1.0f);
      ratioR[l] = consumption * r;
      remainR[l] = fmaxf(0.0f, r - sumr);
    }
    __syncthreads();
    for (int k = tid; k < n; k += nt) {
      const float4 x = p1[k];
      const float rl = ratioL[k];
      float suml = 0.f;
      for (int l = 0; l < m; ++l) {
        const float4 q = p2[l];
        const float d = (q.x - x.x) * (q.x - x.x) + (q.y - x.y) * (q.y - x.y) + (q.z - x.z) * (q.z - x.z);
        const float w = __expf(level * d) * rl * ratioR[l];
        cost += w * d;       // matchcost (:218-262): match[l][k] * d, summed over levels
        suml += w;
      }
      remainL[k] = fmaxf(0.0f, remainL[k] - suml);
    }
    __syncthreads();
  }
  const float tot = dg_block_sum(cost, red);
  if (tid == 0) out[paired ? (long)i : (long)i * Nb + j] = tot;
}

}  // namespace

extern "C" {

int dg_fps(const float* xyz, int B, int n, int m, float* temp, int* idx, float* out, void* s_) {
  if (!xyz || !temp || !idx || B <= 0 || n <= 0 || m <= 0 || m > n) return DG_EINVAL;
  // opt_n_threads(n) of the reference launcher: the largest power of two <= min(n, 512) -- only its tie rule is kept
  int tie = 1;
  while (tie * 2 <= n && tie * 2 <= 512) tie *= 2;
  const int threads = n >= 1024 ? 1024 : (n >= 256 ? 256 : 64);
  fps_kernel<<<B, threads, 0, (hipStream_t)s_>>>(xyz, n, m, tie, temp, idx, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_chamfer_dir(const float* A, int Na, int n, const float* Bc, int Nb, int m, float* L, void* s_) {
  if (!A || !Bc || !L || Na <= 0 || Nb <= 0 || n <= 0 || m <= 0) return DG_EINVAL;
  hipStream_t s = (hipStream_t)s_;
  int wpc = (n + CH_WPTS - 1) / CH_WPTS;        // waves per cloud
  if (wpc > 2) wpc = (wpc + 3) / 4 * 4;         // 1, 2 or whole workgroups
  const int gy = wpc >= 4 ? Na : (Na + 4 / wpc - 1) / (4 / wpc);
  const int gz = wpc >= 4 ? wpc / 4 : 1;
  if (wpc > 1) { const int zrc = dg_zero_f32(L, (long)Na * Nb, s); if (zrc) return zrc; }
  // TB clouds of B per workgroup: long enough to amortise the A registers, short enough for >= ~8 workgroups per CU
  int TB = 32;
  while (TB > 1 && (long)((Nb + TB - 1) / TB) * gy * gz < 2048) TB >>= 1;
  const dim3 grid((Nb + TB - 1) / TB, gy, gz);
  if (grid.y > 65535 || grid.z > 65535) return DG_EUNSUPPORTED;
  chamfer_dir_kernel<<<grid, CH_THREADS, 0, s>>>(A, Na, n, wpc, Bc, Nb, m, TB, L);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_grid_vote(const float* pts, long P, const float* grid, int Ng, float* counters, void* s_) {
  if (!pts || !grid || !counters || P <= 0 || Ng <= 0) return DG_EINVAL;
  grid_vote_kernel<<<(unsigned)((P + 255) / 256), 256, 0, (hipStream_t)s_>>>(pts, P, grid, Ng, counters);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_jsd(const float* P, const float* Q, int n, float* out, void* s_) {
  if (!P || !Q || !out || n <= 0) return DG_EINVAL;
  jsd_kernel<<<1, 1024, 0, (hipStream_t)s_>>>(P, Q, n, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_pyr_down(const float* in, long planes, int H, int W, float* out, void* s_) {
  if (!in || !out || planes <= 0 || H < 4 || W < 4 || (H & 1) || (W & 1)) return DG_EINVAL;
  const long n = planes * (H / 2) * (W / 2);
  pyr_down_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)s_>>>(in, planes, H, W, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_pyr_up_sub(float* fine, const float* coarse, long planes, int H, int W, void* s_) {
  if (!fine || !coarse || planes <= 0 || H < 4 || W < 4 || (H & 1) || (W & 1)) return DG_EINVAL;
  const long n = planes * H * W;
  pyr_up_sub_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)s_>>>(fine, coarse, planes, H, W);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_extract_patches(const float* img, int B, int C, int H, int W, int ph, int pw, const long* inds, int NP,
                       float* out, void* s_) {
  if (!img || !inds || !out || B <= 0 || C <= 0 || ph <= 0 || pw <= 0 || ph > H || pw > W || NP <= 0) return DG_EINVAL;
  const long n = (long)B * NP * C * ph * pw;
  patches_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)s_>>>(img, B, C, H, W, ph, pw, inds, NP, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_emd(const float* A, int Na, int n, const float* Bc, int Nb, int m, int paired, float* out, void* s_) {
  if (!A || !Bc || !out || Na <= 0 || Nb <= 0 || n <= 0 || m <= 0) return DG_EINVAL;
  if (paired && Na != Nb) return DG_EINVAL;
  const size_t lds = (size_t)(n + m) * (16 + 8);
  if (lds > 150 * 1024) return DG_EUNSUPPORTED;
  static bool opted = false;
  if (!opted) {
    HIP_CHECK_RET(hipFuncSetAttribute((const void*)emd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    opted = true;
  }
  const dim3 grid(Nb, paired ? 1 : Na);
  if (grid.y > 65535) return DG_EUNSUPPORTED;
  emd_kernel<<<grid, 512, lds, (hipStream_t)s_>>>(A, n, Bc, m, Nb, paired, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

}  // extern "C"
