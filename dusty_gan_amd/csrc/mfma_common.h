// Shared device helpers of the MFMA implicit-GEMM kernels (conv_mfma.hip, conv_mfma_persist.hip).
#pragma once
#include "common.h"

#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) int i32x4;

#define LDS_READ128(dst, addr, imm) \
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm) : "memory")

__device__ __forceinline__ void mma_tile(const bf16*, const i32x4& a, const i32x4& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_tile(const float*, const i32x4& a, const i32x4& b, f32x16& acc) {
  const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], acc, 0, 0, 0);
}

// ---- "fp32x3": fp32 operands on the bf16 matrix cores (SURVEY.md section 7 "Precision contract": fp32 storage with fp32 or
// split-bf16 x 3 MFMA).  Every fp32 operand x is split into hi = bf16(x) and lo = bf16(x - hi) (16 mantissa bits kept), and a
// product a b is formed as a_hi b_hi + a_lo b_hi + a_hi b_lo with fp32 accumulation (the a_lo b_lo term, <= 2^-16 of the
// product, is dropped).  The fragment a fp32 kernel holds per lane - 4 consecutive k of one row - becomes the 8-element
// bf16 fragments of v_mfma_f32_32x32x16_bf16 by putting (hi | lo) side by side:
//     MFMA 1:  A = [a_hi(4) | a_lo(4)]   B = [b_hi(4) | b_hi(4)]      -> a_hi b_hi + a_lo b_hi
//     MFMA 2:  A = [a_hi(4) |   0    ]   B = [b_lo(4) |   0    ]      -> a_hi b_lo
// i.e. 2 x 32 cycles where the exact path (4 x v_mfma_f32_32x32x2_f32) takes 4 x 64, with no change to the LDS images or
// the fragment reads.  Asked for per call: DG_FORCE_FP32X3 in the `force` argument of dg_conv / dg_wgrad (api.hip).

struct SplitA { i32x4 hl, h0; };   // [hi01, hi23, lo01, lo23], [hi01, hi23, 0, 0]
struct SplitB { i32x4 hh, l0; };   // [hi01, hi23, hi01, hi23], [lo01, lo23, 0, 0]

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ void split_f32x4(const f32x4& f, unsigned (&hi)[2], unsigned (&lo)[2]) {
  // plain casts, not inline asm: the packed conversions feed matrix instructions a few cycles later, and hipcc pads the
  // VALU-write -> MFMA-read wait states only behind instructions it can see (a first version with asm v_cvt_pk_bf16_f32
  // multiplied stale registers on one tile shape: 1e36-sized weight gradients)
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const bf16x2_t h2 = {(__bf16)f[2 * q], (__bf16)f[2 * q + 1]};
    const unsigned h = __builtin_bit_cast(unsigned, h2);
    const float l0 = f[2 * q] - __builtin_bit_cast(float, h << 16);
    const float l1 = f[2 * q + 1] - __builtin_bit_cast(float, h & 0xffff0000u);
    const bf16x2_t l2 = {(__bf16)l0, (__bf16)l1};
    hi[q] = h;
    lo[q] = __builtin_bit_cast(unsigned, l2);
  }
}
__device__ __forceinline__ SplitA split_a(const f32x4& f) {
  unsigned hi[2], lo[2];
  split_f32x4(f, hi, lo);
  SplitA r;
  r.hl = i32x4{(int)hi[0], (int)hi[1], (int)lo[0], (int)lo[1]};
  r.h0 = i32x4{(int)hi[0], (int)hi[1], 0, 0};
  return r;
}
__device__ __forceinline__ SplitB split_b(const f32x4& f) {
  unsigned hi[2], lo[2];
  split_f32x4(f, hi, lo);
  SplitB r;
  r.hh = i32x4{(int)hi[0], (int)hi[1], (int)hi[0], (int)hi[1]};
  r.l0 = i32x4{(int)lo[0], (int)lo[1], 0, 0};
  return r;
}
__device__ __forceinline__ void mma_tile_x3(const SplitA& a, const SplitB& b, f32x16& acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.hl), __builtin_bit_cast(bf16x8, b.hh), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a.h0), __builtin_bit_cast(bf16x8, b.l0), acc, 0, 0, 0);
}

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  // one wave instruction: lane l's 16 bytes land at lds_dst + 16*l (lds_dst is wave-uniform)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

