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

__device__ __forceinline__ void dma16(const void* gsrc, void* lds_dst) {
  // one wave instruction: lane l's 16 bytes land at lds_dst + 16*l (lds_dst is wave-uniform)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

