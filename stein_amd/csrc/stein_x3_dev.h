// stein_x3_dev.h -- device helpers shared by the split-precision kernels (stein_x3.hip, stein_dpanel.hip): vector types,
// the operand-plane geometry, the 16x16x32 products and the hand-counted streamed loads.
#pragma once
#include "stein_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // staging registers (native vector: stays in VGPRs)

constexpr int XROW = 64;                 // bytes per LDS row: 32 bf16
constexpr int XPLANE = 128 * XROW;       // one 128-row plane of a tile: 8192 B
constexpr int XTILE_E = 128 * 32;        // elements of one plane of a tile
// Plane of an operand tile (128 rows x 32 k), element offset of 16-byte chunk `chunk` (8 consecutive k starting at
// 8 chunk) of row `row` (0..127): fragment order [row / 16][chunk][row % 16][8], i.e. the operand fragment of a 16x16x32
// MFMA (16 rows x 32 k) is 1 KB contiguous and lane l = 16 chunk + row % 16 reads bytes 16 l .. 16 l + 15
__host__ __device__ __forceinline__ int vfrag_offset(int row, int chunk) {
  return (((row >> 4) * 4 + chunk) * 16 + (row & 15)) * 8;
}

// The products of one fragment pair, smallest first (plane 0 = most significant term).  NP is the split KIND = the
// number of planes: 2 -> three fp16 products, 1 -> one bf16 product.  Fragments are held
// as 32-bit vectors and bit-cast at the MFMA (loop-carried 16-bit vectors get scalarised by the compiler).
#define X3_BF(v) __builtin_bit_cast(bf16x8, v)
#define X3_HF(v) __builtin_bit_cast(f16x8, v)

// the same products on the 16x16x32 shape (one MFMA covers a whole 32-deep k tile)
template <int NP>
__device__ __forceinline__ f32x4 x3_products16(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x4 c) {
  if (NP == 2) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(X3_HF(a[1]), X3_HF(b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(X3_HF(a[0]), X3_HF(b[1]), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(X3_HF(a[0]), X3_HF(b[0]), c, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(X3_BF(a[0]), X3_BF(b[0]), c, 0, 0, 0);
}

// ---- streamed loads with hand-counted waits (the full story: stein_x3.hip, in front of k_phi_x3fs) -------------------
typedef float f32x4g __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void stream_load16(f32x4g& dst, const void* base /* wave-uniform */, u32 byte_off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base));
}
__device__ __forceinline__ void stream_load16(u32x4& dst, const void* base /* wave-uniform */, u32 byte_off) {
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(byte_off), "s"(base));
}
template <int N>
__device__ __forceinline__ void stream_wait() {
  asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
  __builtin_amdgcn_sched_barrier(0);
}
