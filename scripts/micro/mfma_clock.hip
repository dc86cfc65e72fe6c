// What the matrix pipe sustains on this box: every SIMD of every CU issues independent v_mfma_f32_16x16x32_bf16 back to back
// (WAVES waves per SIMD, 8 accumulators each) for N iterations; time by HIP events.  16 cycles per instruction and SIMD at
// full rate => the implied shader clock under a pure matrix load, and the TFLOP/s ceiling a kernel can be priced against.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_clock scripts/micro/mfma_clock.hip ; run: ./mfma_clock [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2;
  int cus = 256;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  float* out;
  hipMalloc(&out, 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    mfma_loop<<<cus, 64 * 4 * wps>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * wps;
    const double flops = mfma_per_simd * 4 * cus * 16384.0;
    printf("waves/SIMD %d: %.3f ms, %.1f TFLOP/s, implied clock at 16 cycles per MFMA: %.3f GHz\n", wps, ms, flops / ms / 1e9,
           mfma_per_simd * 16 / (ms * 1e-3) / 1e9);
  }
  return 0;
}
