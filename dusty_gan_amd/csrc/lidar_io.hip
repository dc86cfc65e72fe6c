// The data formats either side of the training step, as streaming kernels (gfx950):
//   * scan_to_polar_kernel : a batch of projected KITTI scans [B,Hs,Ws,C>=3] (x,y,z[,reflectance] per cell, the
//     `.npy` files written by process_kitti.py:76-118) -> normalised polar depth, validity mask, unit-space xyz at
//     the training resolution, and optionally the network input itself.  Replaces KITTIOdometry.preprocess +
//     .transform (datasets/kitti.py:54-77: norm, range mask, min/max normalisation, zeroing, to_tensor, hflip,
//     NEAREST resize) and, fused behind it, Trainer.fetch_reals (trainers/dcgan_amp.py:154-160).
//   * inv_to_xyz_kernel    : generated inverse depth -> metric-normalised point map on the sensor's angle grid.
//     Replaces utils.postprocess's depth branch + Coordinate.inv_to_xyz / revert_depth / pol_to_xyz
//     (utils/__init__.py:163-178, utils/lidar.py:38-68).
// Both are HBM-bound: one 16-byte cell read per output pixel / one image-sized stream in, three out.
#include "common.h"

namespace {

inline int nblk(long n) { return (int)((n + 255) / 256); }

// torch's legacy "nearest" (what torchvision 0.9 TF.resize(tensor, size, NEAREST) calls: F.interpolate(mode="nearest")):
// src = min(floor(dst * float(in) / out), in - 1)
__device__ __forceinline__ int nearest_src(int dst, int in_size, int out_size) {
  const float scale = (float)in_size / (float)out_size;
  const int s = (int)floorf((float)dst * scale);
  return s < in_size - 1 ? s : in_size - 1;
}

__global__ __launch_bounds__(256) void scan_to_polar_kernel(
    const float* __restrict__ scan, int B, int Hs, int Ws, int C, int H, int W, const unsigned char* __restrict__ flip,
    float min_d, float max_d, float range_d, float drop_const, float* __restrict__ pol, float* __restrict__ mask,
    float* __restrict__ xyz, float* __restrict__ x_real) {
#pragma clang fp contract(off)  // plain IEEE operators, never fused: HIP's __fmul_rn / __fsqrt_rn wrappers are
                                // contractable / native-precision, so they are not used here
  const long n = (long)B * H * W;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int w = (int)(i % W), h = (int)((i / W) % H), b = (int)(i / ((long)W * H));
  const int hs = nearest_src(h, Hs, H);
  int ws = nearest_src(w, Ws, W);
  if (flip && flip[b]) ws = Ws - 1 - ws;  // TF.hflip happens before the resize (datasets/kitti.py:73-75)
  const float* cell = scan + (((long)b * Hs + hs) * Ws + ws) * C;
  float x, y, z;
  if (C == 4) {
    const float4 v = *reinterpret_cast<const float4*>(cell);
    x = v.x; y = v.y; z = v.z;
  } else {
    x = cell[0]; y = cell[1]; z = cell[2];
  }
  // np.linalg.norm(xyz, ord=2, axis=2) in float32: sqrt(add.reduce(x*x)); no fused multiply-add, so the range
  // mask below makes the same decisions as numpy at the min/max boundaries
  const float d = sqrtf((x * x + y * y) + z * z);  // correctly rounded (hipcc's default for fp32 sqrt and divide)
  const bool valid = d > 0.f && d > min_d && d < max_d;
  const float p = valid ? (d - min_d) / range_d : 0.f;
  pol[i] = p;
  mask[i] = valid ? 1.f : 0.f;
  if (xyz) {
    const long hw = (long)H * W, o = (long)b * 3 * hw + (long)h * W + w;
    xyz[o] = valid ? x / max_d : 0.f;
    xyz[o + hw] = valid ? y / max_d : 0.f;
    xyz[o + 2 * hw] = valid ? z / max_d : 0.f;
  }
  if (x_real) {  // fetch_reals: invert_depth -> [-1,1] -> dropped pixels = drop_const (same arithmetic as
                 // fetch_reals_kernel in pointwise.hip)
    const float depth = p * (max_d - min_d) + min_d;
    const float disp = 1.f / depth;
    float inv = (disp - 1.f / max_d) / (1.f / min_d - 1.f / max_d);
    inv = inv * 2.f - 1.f;
    x_real[i] = valid ? inv : drop_const;
  }
}

__global__ __launch_bounds__(256) void inv_to_xyz_kernel(const float* __restrict__ in, const float* __restrict__ angle,
                                                         int B, int H, int W, int from_tanh, float min_d, float max_d,
                                                         float drop_const, float tol, float* __restrict__ depth01,
                                                         float* __restrict__ points) {
  const long hw = (long)H * W, n = (long)B * hw;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long px = i % hw, b = i / hw;
  float inv = in[i];
  if (from_tanh) {  // tanh_to_sigmoid(.).clamp_(0, 1)  utils/__init__.py:168
    inv = (inv + 1.f) / 2.f;
    inv = fminf(fmaxf(inv, 0.f), 1.f);
  }
  if (depth01) depth01[i] = inv;
  const bool valid = fabsf(inv - drop_const) > tol;                           // lidar.py:59
  const float disp = inv * (1.f / min_d - 1.f / max_d) + 1.f / max_d;          // revert_depth :40-43
  float d = 1.f / disp;
  d = (d - min_d) / (max_d - min_d);                                           // :45-46
  d = d * (max_d - min_d) + min_d;                                             // :61
  d = d / max_d;                                                               // :62
  d = valid ? d : 0.f;                                                         // :63
  const float pitch = angle[px], yaw = angle[hw + px];
  const float cp = cosf(pitch), sp = sinf(pitch), cy = cosf(yaw), sy = sinf(yaw);
  const long o = b * 3 * hw + px;
  points[o] = d * cp * cy;                                                     // pol_to_xyz :49-56
  points[o + hw] = d * cp * sy;
  points[o + 2 * hw] = d * sp;
}

// mode 0: tanh_to_sigmoid(x).clamp(0,1)  (depth_orig, utils/__init__.py:169-170);  mode 1: sigmoid(x) (confidence, :171-172)
__global__ __launch_bounds__(256) void unit_map_kernel(const float* __restrict__ x, long n, int mode,
                                                       float* __restrict__ y) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  y[i] = mode == 0 ? fminf(fmaxf((v + 1.f) / 2.f, 0.f), 1.f) : 1.f / (1.f + __expf(-v));
}

// Surface normals of a point map, "closest" mode (utils/geometry.py:38-127 estimate_surface_normal with d = 2, then
// xyz_to_normal utils/__init__.py:215-219): the 8 neighbours at distance d (rows padded with +inf, columns circular);
// candidate k pairs neighbour k with neighbour k + 2; the pair with the smallest |p1 - a| + |p2 - a| (first on ties)
// gives n = (p1 - a) x (p2 - a) / (|.| + 1e-8); output = clamp((-n + 1) / 2, 0, 1) with NaN -> 0 before the map.
__global__ __launch_bounds__(256) void normals_kernel(const float* __restrict__ pts, int B, int H, int W, int d,
                                                      float* __restrict__ out) {
#pragma clang fp contract(off)
  const long hw = (long)H * W, n = (long)B * hw;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int x = (int)(i % W), y = (int)((i / W) % H);
  const float* base = pts + (i / hw) * 3 * hw;
  const int DH[8] = {-1, -1, 0, 1, 1, 1, 0, -1}, DW[8] = {0, 1, 1, 1, 0, -1, -1, -1};
  const float inf = __builtin_inff();
  float nb[8][3];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int yy = y + DH[k] * d;
    int xx = (x + DW[k] * d) % W;
    if (xx < 0) xx += W;
    const bool ok = yy >= 0 && yy < H;
    const long o = (long)(ok ? yy : 0) * W + xx;
#pragma unroll
    for (int c = 0; c < 3; ++c) nb[k][c] = ok ? base[c * hw + o] : inf;
  }
  const long oa = (long)y * W + x;
  const float a0 = base[oa], a1 = base[hw + oa], a2 = base[2 * hw + oa];
  float dist[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float u0 = nb[k][0] - a0, u1 = nb[k][1] - a1, u2 = nb[k][2] - a2;
    dist[k] = sqrtf((u0 * u0 + u1 * u1) + u2 * u2);
  }
  float best = 0.f;
  int bk = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float s = dist[k] + dist[(k + 2) & 7];
    if (k == 0 || s < best) { best = s; bk = k; }
  }
  float v1[3], v2[3];
#pragma unroll
  for (int k = 0; k < 8; ++k)  // register-indexed select (no dynamic indexing into nb)
    if (k == bk) {
      const int k2 = (k + 2) & 7;
      v1[0] = nb[k][0] - a0; v1[1] = nb[k][1] - a1; v1[2] = nb[k][2] - a2;
      v2[0] = nb[k2][0] - a0; v2[1] = nb[k2][1] - a1; v2[2] = nb[k2][2] - a2;
    }
  float c0 = v1[1] * v2[2] - v1[2] * v2[1], c1 = v1[2] * v2[0] - v1[0] * v2[2], c2 = v1[0] * v2[1] - v1[1] * v2[0];
  const float len = sqrtf((c0 * c0 + c1 * c1) + c2 * c2) + 1e-8f;
  float r[3] = {-(c0 / len), -(c1 / len), -(c2 / len)};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float v = r[c] != r[c] ? 0.f : r[c];
    v = (v + 1.f) / 2.f;
    out[(i / hw) * 3 * hw + c * hw + oa] = fminf(fmaxf(v, 0.f), 1.f);
  }
}

}  // namespace

extern "C" {

int dg_unit_map(const float* x, long n, int mode, float* y, void* s_) {
  if (!x || !y || n <= 0 || mode < 0 || mode > 1) return DG_EINVAL;
  unit_map_kernel<<<nblk(n), 256, 0, (hipStream_t)s_>>>(x, n, mode, y);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_scan_to_polar(const float* scan, int B, int Hs, int Ws, int C, int H, int W, const unsigned char* flip,
                     double min_depth, double max_depth, float drop_const, float* pol, float* mask, float* xyz,
                     float* x_real, void* s_) {
  if (!scan || !pol || !mask || B <= 0 || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0 || C < 3) return DG_EINVAL;
  if (!(max_depth > min_depth)) return DG_EINVAL;
  // `out["depth"] /= self.max_depth - self.min_depth`: the divisor is a Python double rounded once to float32
  // (the config values arrive as doubles for exactly that reason)
  const float range_d = (float)(max_depth - min_depth);
  const long n = (long)B * H * W;
  scan_to_polar_kernel<<<nblk(n), 256, 0, (hipStream_t)s_>>>(scan, B, Hs, Ws, C, H, W, flip, (float)min_depth,
                                                              (float)max_depth, range_d, drop_const, pol, mask, xyz, x_real);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_inv_to_xyz(const float* inv, const float* angle, int B, int H, int W, int from_tanh, float min_depth,
                  float max_depth, float drop_const, float tol, float* depth01, float* points, void* s_) {
  if (!inv || !angle || !points || B <= 0 || H <= 0 || W <= 0) return DG_EINVAL;
  const long n = (long)B * H * W;
  inv_to_xyz_kernel<<<nblk(n), 256, 0, (hipStream_t)s_>>>(inv, angle, B, H, W, from_tanh, min_depth, max_depth,
                                                           drop_const, tol, depth01, points);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

int dg_normals(const float* points, int B, int H, int W, int d, float* out, void* s_) {
  if (!points || !out || B <= 0 || H <= 0 || W <= 0 || d <= 0) return DG_EINVAL;
  const long n = (long)B * H * W;
  normals_kernel<<<nblk(n), 256, 0, (hipStream_t)s_>>>(points, B, H, W, d, out);
  HIP_CHECK_RET(hipGetLastError());
  return DG_OK;
}

}  // extern "C"
