// numerics of v_dot2c_f32_bf16 vs an fp32 fma chain on the same bf16 inputs
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__global__ void k(const unsigned* a, const unsigned* b, float* o1, float* o2, int n) {
  int t = threadIdx.x + blockIdx.x * blockDim.x;
  float acc1 = 0.f, acc2 = 0.f;
  for (int i = 0; i < n; ++i) {
    unsigned x = a[t * n + i], y = b[t * n + i];
    acc1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, x), __builtin_bit_cast(bf16x2, y), acc1, false);
    bf16x2 xv = __builtin_bit_cast(bf16x2, x), yv = __builtin_bit_cast(bf16x2, y);
    acc2 += (float)xv[0] * (float)yv[0];
    acc2 += (float)xv[1] * (float)yv[1];
  }
  o1[t] = acc1; o2[t] = acc2;
}
int main() {
  const int T = 256, n = 128;
  unsigned *a, *b; float *o1, *o2;
  hipMallocManaged(&a, T * n * 4); hipMallocManaged(&b, T * n * 4);
  hipMallocManaged(&o1, T * 4); hipMallocManaged(&o2, T * 4);
  srand(1);
  auto rb = []() { float f = (rand() / (float)RAND_MAX) * 2 - 1; unsigned u; memcpy(&u, &f, 4); return (u >> 16) & 0xffff; };
  for (int i = 0; i < T * n; ++i) { a[i] = rb() | (rb() << 16); b[i] = rb() | (rb() << 16); }
  k<<<1, T>>>(a, b, o1, o2, n);
  hipDeviceSynchronize();
  double num = 0, den = 0;
  for (int i = 0; i < T; ++i) { num += (o1[i] - o2[i]) * (double)(o1[i] - o2[i]); den += o2[i] * (double)o2[i]; }
  printf("dot2 vs fma rel l2 %.3e  (sample %f %f)\n", sqrt(num / den), o1[0], o2[0]);
  return 0;
}
