// What a dependent launch of the replayed step costs before it computes anything, and what a one-tile-per-CU kernel's store
// burst costs at its end: chains of N dependent kernel nodes in one hipGraph (as the training step is replayed), each node
//   empty          nothing
//   store KB       every workgroup (512 threads, one per CU) writes KB kilobytes with 16-byte stores (plain / nontemporal)
//   load+store     every workgroup first reads 128 KB (16-byte loads, summed), then writes 64 KB
//   spin US        every workgroup spins US microseconds (s_memrealtime), then writes 64 KB: is the burst hidden behind a
//                  long kernel or does it add to it?
// build: hipcc --offload-arch=gfx950 -O3 -o launch_tail scripts/micro/launch_tail.hip ; run: ./launch_tail
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) int i32x4;

__global__ __launch_bounds__(512) void k_empty(int* p) {
  if (p == nullptr) __builtin_trap();
}

template <bool NT>
__global__ __launch_bounds__(512) void k_store(i32x4* out, int per_thread, int spin_us, const i32x4* in, int loads) {
  i32x4 v = {(int)threadIdx.x, (int)blockIdx.x, 3, 4};
  if (loads > 0) {
    const i32x4* src = in + (long)blockIdx.x * loads * 512 + threadIdx.x;
    for (int i = 0; i < loads; ++i) {
      const i32x4 t = src[(long)i * 512];
      v += t;
    }
  }
  if (spin_us > 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(8);
  }
  i32x4* dst = out + (long)blockIdx.x * per_thread * 512 + threadIdx.x;
  for (int i = 0; i < per_thread; ++i) {
    if (NT) __builtin_nontemporal_store(v, dst + (long)i * 512);
    else dst[(long)i * 512] = v;
  }
}

// the ping-pong conv's output pattern: a 256-pixel x 128-channel bf16 tile per workgroup, every wave-instruction writes 8
// pixels x 128 B (its 64 channels) at the tensor's pixel stride (256 B x N tiles); the N tiles of a pixel block are
// neighbouring workgroups
__global__ __launch_bounds__(512) void k_store_tile(char* out, int stride, int spin_us) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int nt = stride / 256;
  const long base = (long)(blockIdx.x / nt) * 256 * stride + (long)(blockIdx.x % nt) * 256 + wn * 128 + (lane & 7) * 16;
  if (spin_us > 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100ull) __builtin_amdgcn_s_sleep(8);
  }
  const i32x4 v = {(int)threadIdx.x, (int)blockIdx.x, 3, 4};
  for (int i = 0; i < 8; ++i) {
    const int px = wm * 64 + i * 8 + (lane >> 3);
    *(i32x4*)(out + base + (long)px * stride) = v;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <class F>
static double time_chain(hipStream_t s, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) launch(i);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1e3 / n;   // microseconds per node
}

int main() {
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int N = 40;
  const size_t out_bytes = (size_t)cus * 512 * 1024;          // up to 512 KB per workgroup, a fresh region per node pair
  i32x4 *out[2], *in;
  CK(hipMalloc(&out[0], out_bytes)); CK(hipMalloc(&out[1], out_bytes)); CK(hipMalloc(&in, out_bytes));
  CK(hipMemset(in, 1, out_bytes));
  int* dummy; CK(hipMalloc(&dummy, 4));
  printf("CUs %d, %d dependent nodes per graph, microseconds per node\n", cus, N);
  printf("empty<<<1,64>>>            %7.2f\n", time_chain(s, N, [&](int) { k_empty<<<1, 64, 0, s>>>(dummy); }));
  printf("empty<<<CUs,512>>>         %7.2f\n", time_chain(s, N, [&](int) { k_empty<<<cus, 512, 0, s>>>(dummy); }));
  for (int kb : {8, 32, 64, 128, 256}) {
    const int pt = kb * 1024 / (512 * 16);
    const double a = time_chain(s, N, [&](int i) { k_store<false><<<cus, 512, 0, s>>>(out[i & 1], pt, 0, in, 0); });
    const double b = time_chain(s, N, [&](int i) { k_store<true><<<cus, 512, 0, s>>>(out[i & 1], pt, 0, in, 0); });
    printf("store %3d KB per CU (%5.1f MB)  plain %7.2f   nontemporal %7.2f\n", kb, kb * cus / 1024.0, a, b);
  }
  {
    const double a = time_chain(s, N, [&](int i) { k_store<false><<<cus, 512, 0, s>>>(out[i & 1], 8, 0, in, 16); });
    const double b = time_chain(s, N, [&](int i) { k_store<false><<<cus, 512, 0, s>>>(out[i & 1], 8, 0, out[(i + 1) & 1], 16); });
    printf("load 128 KB + store 64 KB per CU: constant input %7.2f   previous node's output (a chain) %7.2f\n", a, b);
  }
  for (int us : {10, 30}) {
    const double a = time_chain(s, N, [&](int i) { k_store<false><<<cus, 512, 0, s>>>(out[i & 1], 0, us, in, 0); });
    const double b = time_chain(s, N, [&](int i) { k_store<false><<<cus, 512, 0, s>>>(out[i & 1], 8, us, in, 0); });
    const double c = time_chain(s, N, [&](int i) { k_store<true><<<cus, 512, 0, s>>>(out[i & 1], 8, us, in, 0); });
    printf("spin %2d us: no store %7.2f   + 64 KB store at the end %7.2f (nontemporal %7.2f)\n", us, a, b, c);
  }
  for (int stride : {256, 512, 1024}) {
    const double a = time_chain(s, N, [&](int i) { k_store_tile<<<cus, 512, 0, s>>>((char*)out[i & 1], stride, 0); });
    const double b = time_chain(s, N, [&](int i) { k_store_tile<<<cus, 512, 0, s>>>((char*)out[i & 1], stride, 10); });
    printf("conv tile pattern, 64 KB per CU, pixel stride %4d B: %7.2f   after a 10 us spin %7.2f\n", stride, a, b);
  }
  // stream launches (no graph) of the empty kernel, for comparison
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) k_empty<<<cus, 512, 0, s>>>(dummy);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < 200; ++i) k_empty<<<cus, 512, 0, s>>>(dummy);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream launches of empty<<<CUs,512>>>: %7.2f us each\n", ms * 1e3 / 200);
  }
  return 0;
}
