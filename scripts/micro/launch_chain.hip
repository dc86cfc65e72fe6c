// Go / no-go for removing the ramp between consecutive persistent conv launches (round-5 review, item 2): a chain of
// N dependent "layers", each one workgroup of 512 threads per CU holding 130 KB of LDS (so two layers' workgroups can never
// share a CU, exactly like the ping-pong conv), shaped after the stamps of profiles/r05a_conv_lifetime_phases_b32.txt:
//   prologue (PRO us of ALU)  ->  weight fill (64 KB per CU of layer-constant bytes into LDS)
//   ->  [the dependency on the previous layer is needed from HERE]
//   ->  input fill (64 KB per CU of the PREVIOUS layer's output, written by another CU, checked word by word)
//   ->  K loop (BODY us)  ->  64 KB of output stores
// run three ways:
//   edges   N kernel nodes of one hipGraph, each depending on its predecessor (what the step does today)
//   flags   the same N nodes on TWO parallel branches of the graph (even / odd layers): a node depends on the node two
//           layers back by its graph edge and on its predecessor through a done-counter the predecessor's workgroups
//           release at exit and this layer's workgroups acquire after their prologue and weight fill; every spin is bounded
//   chain   ONE node: the layers as phases of one persistent launch, a counter barrier between phases, the next phase's
//           weight fill issued in front of the barrier
//   flags-wt / chain-wt   the same two with the hand-off in the guide's write-through form (MI355X_MICROARCH.md, "Valid forms",
//           table row 1): every output store `sc1`, no release fence (each wave drains its stores, workgroup barrier, one lane
//           adds to the counter), every load of the handed-off bytes `sc1`, no acquire fence (poll, workgroup barrier)
// Every consumed word is checked (stale reads are counted), every timeout is counted.
// build: hipcc --offload-arch=gfx950 -O3 -o launch_chain scripts/micro/launch_chain.hip ; run: ./launch_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int THREADS = 512;
constexpr int KB = 64;                                   // bytes per CU of each of: weights, input, output
constexpr int VEC_PER_THREAD = KB * 1024 / (THREADS * 16);   // 8 x 16 B per thread
constexpr unsigned SPIN_LIMIT = 200000;                  // bounded spins (~0.2 s), then the timeout word is set

struct Chain {
  u32x4* buf[2];            // layer l writes buf[l & 1], reads buf[(l + 1) & 1]
  const u32x4* weights;
  unsigned* done;           // done[l]: monotonic over replays, += 1 per workgroup of layer l
  unsigned* epoch;          // replay number (incremented by the head node)
  unsigned* err;            // [0] stale words, [1] timeouts
  int pro_us, body_us, nwg, wt;
};

__device__ __forceinline__ void spin_us(int us) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(2);
}

__device__ __forceinline__ unsigned word_of(unsigned epoch, int layer, int wg, int i) {
  return epoch * 2654435761u + (unsigned)layer * 40503u + (unsigned)wg * 97u + (unsigned)i;
}

// wait: 0 none (graph edge), 1 poll done[layer - 1]
__device__ __forceinline__ void layer_body(const Chain& c, int layer, int wg, unsigned epoch, u32x4* lds, int wait,
                                           bool skip_pro_fill) {
  const int t = threadIdx.x;
  if (!skip_pro_fill) {
    spin_us(c.pro_us);
    const u32x4* w = c.weights + (long)(layer & 7) * KB * 64 + t;     // layer-constant bytes (L2 / MALL resident)
#pragma unroll
    for (int i = 0; i < VEC_PER_THREAD; ++i) lds[i * THREADS + t] = w[(long)i * THREADS];
  }
  if (wait && layer > 0) {
    if (t == 0) {
      const unsigned want = epoch * (unsigned)c.nwg;
      unsigned n = 0;
      while (__hip_atomic_load(&c.done[layer - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(4);
        if (++n > SPIN_LIMIT) { atomicAdd(&c.err[1], 1u); break; }
      }
      if (!c.wt) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __syncthreads();
  }
  unsigned bad = 0;
  if (layer > 0) {                                                    // the previous layer's output, written by another CU
    const int src = (wg + 37) % c.nwg;
    const u32x4* in = c.buf[(layer + 1) & 1] + (long)src * KB * 64 + t;
    u32x4 v[VEC_PER_THREAD];
    if (c.wt) {                                                       // all loads in flight, ONE wait the values are tied to
#pragma unroll
      for (int i = 0; i < VEC_PER_THREAD; ++i)
        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(v[i]) : "v"(in + (long)i * THREADS) : "memory");
      static_assert(VEC_PER_THREAD == 8, "operand list below");
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) :: "memory");
    } else {
#pragma unroll
      for (int i = 0; i < VEC_PER_THREAD; ++i) v[i] = in[(long)i * THREADS];
    }
#pragma unroll
    for (int i = 0; i < VEC_PER_THREAD; ++i) {
      lds[(VEC_PER_THREAD + i) * THREADS + t] = v[i];
      bad += v[i].x != word_of(epoch, layer - 1, src, i * THREADS + t);
    }
  }
  if (bad) atomicAdd(&c.err[0], bad);
  spin_us(c.body_us);
  u32x4* out = c.buf[layer & 1] + (long)wg * KB * 64 + t;
#pragma unroll
  for (int i = 0; i < VEC_PER_THREAD; ++i) {
    u32x4 v = lds[i * THREADS + t];
    v.x = word_of(epoch, layer, wg, i * THREADS + t);
    if (c.wt) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(out + (long)i * THREADS), "v"(v) : "memory");
    else out[(long)i * THREADS] = v;
  }
}

__device__ __forceinline__ void publish(const Chain& c, int layer) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    if (!c.wt) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __hip_atomic_fetch_add(&c.done[layer], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

__global__ void k_head(Chain c) { if (threadIdx.x == 0 && blockIdx.x == 0) c.epoch[0] += 1; }

__global__ __launch_bounds__(THREADS) void k_layer(Chain c, int layer, int wait) {
  extern __shared__ u32x4 lds[];
  const unsigned epoch = __hip_atomic_load(c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  layer_body(c, layer, blockIdx.x, epoch, lds, wait, false);
  if (wait) publish(c, layer);
}

__global__ __launch_bounds__(THREADS) void k_chain(Chain c, int nlayers) {
  extern __shared__ u32x4 lds[];
  const unsigned epoch = __hip_atomic_load(c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  for (int l = 0; l < nlayers; ++l) {
    // (the weight fill of phase l > 0 was issued in front of the barrier below, at the end of phase l - 1)
    layer_body(c, l, blockIdx.x, epoch, lds, 1, l > 0);
    publish(c, l);
    if (l + 1 < nlayers) {
      spin_us(c.pro_us);
      const u32x4* w = c.weights + (long)((l + 1) & 7) * KB * 64 + threadIdx.x;
#pragma unroll
      for (int i = 0; i < VEC_PER_THREAD; ++i) lds[i * THREADS + threadIdx.x] = w[(long)i * THREADS];
    }
  }
}

static double replay_us(hipStream_t s, hipGraphExec_t ge, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, s));
    CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best * 1e3;
}

int main(int argc, char** argv) {
  int cus = 256;
  CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
  const int N = 12;
  const size_t lds_bytes = 130 * 1024;
  CK(hipFuncSetAttribute((const void*)k_layer, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  CK(hipFuncSetAttribute((const void*)k_chain, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
  hipStream_t s, s2; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&s2));
  Chain c;
  const size_t bytes = (size_t)cus * KB * 1024;
  CK(hipMalloc(&c.buf[0], bytes)); CK(hipMalloc(&c.buf[1], bytes));
  u32x4* w; CK(hipMalloc(&w, 8 * KB * 1024)); CK(hipMemset(w, 1, 8 * KB * 1024)); c.weights = w;
  CK(hipMalloc(&c.done, 64 * 4)); CK(hipMalloc(&c.epoch, 4)); CK(hipMalloc(&c.err, 8));
  c.nwg = cus;
  printf("CUs %d, %d layers per chain, 130 KB of LDS per workgroup, 64 KB weights + 64 KB input + 64 KB output per CU and layer\n", cus, N);
  printf("%-20s %9s %9s %9s %9s %9s   (us per layer; stale words / timeouts over all replays)\n", "prologue / body (us)", "edges", "flags", "chain", "flags-wt", "chain-wt");
  for (int body : {10, 30}) for (int pro : {0, 2}) {
    c.pro_us = pro; c.body_us = body;
    double res[5]; unsigned errs[5][2];
    for (int variant = 0; variant < 5; ++variant) {
      c.wt = variant >= 3;
      CK(hipMemset(c.done, 0, 64 * 4)); CK(hipMemset(c.epoch, 0, 4)); CK(hipMemset(c.err, 0, 8));
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      k_head<<<1, 64, 0, s>>>(c);
      if (variant == 0) {
        for (int l = 0; l < N; ++l) k_layer<<<cus, THREADS, lds_bytes, s>>>(c, l, 0);
      } else if (variant == 1 || variant == 3) {
        hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
        CK(hipEventRecord(fork, s)); CK(hipStreamWaitEvent(s2, fork, 0));
        for (int l = 0; l < N; ++l) k_layer<<<cus, THREADS, lds_bytes, (l & 1) ? s2 : s>>>(c, l, 1);
        CK(hipEventRecord(join, s2)); CK(hipStreamWaitEvent(s, join, 0));
      } else {
        k_chain<<<cus, THREADS, lds_bytes, s>>>(c, N);
      }
      CK(hipStreamEndCapture(s, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      res[variant] = replay_us(s, ge, 8) / N;
      CK(hipMemcpy(errs[variant], c.err, 8, hipMemcpyDeviceToHost));
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    char tag[64]; snprintf(tag, sizeof tag, "%d / %d", pro, body);
    printf("%-20s %9.2f %9.2f %9.2f %9.2f %9.2f   edges %u/%u  flags %u/%u  chain %u/%u  flags-wt %u/%u  chain-wt %u/%u\n", tag, res[0], res[1],
           res[2], res[3], res[4], errs[0][0], errs[0][1], errs[1][0], errs[1][1], errs[2][0], errs[2][1], errs[3][0], errs[3][1],
           errs[4][0], errs[4][1]);
  }
  return 0;
}
