// coissue.hip -- do matrix (MFMA) and vector (VALU) instructions of DIFFERENT waves on one SIMD overlap, per MFMA shape?
// One workgroup per CU; waves [0, NV) x 4 run a VALU loop shaped like the contraction's producers (fma, exp, add, two more
// plain ops per entry), the following waves run back-to-back MFMAs (16x16x32 or 32x32x16, f16).  Each wave times its own
// loop with s_memtime.  Modes: both kinds, MFMA waves only, VALU waves only.  Diagnostic probe, not part of the product.
//   hipcc --offload-arch=gfx950 -O3 coissue.hip -o coissue && ./coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

template <int SHAPE>   // 16 or 32
__device__ __forceinline__ u64 mfma_loop(int iters, float* sink) {
  f16x8 a, b;
  for (int k = 0; k < 8; ++k) { a[k] = (_Float16)(0.001f * (threadIdx.x + k)); b[k] = (_Float16)(0.002f * (threadIdx.x + 3 * k)); }
  u64 t0, t1;
  if (SHAPE == 16) {
    f32x4 c[8];
    for (int u = 0; u < 8; ++u) c[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 3; ++r)      // three dependent products per accumulator, as x3_products16 does
#pragma unroll
        for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[u], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0.f;
    for (int u = 0; u < 8; ++u) s += c[u][0] + c[u][3];
    if (s == 123.456f) *sink = s;
  } else {
    f32x16 c[4];
    for (int u = 0; u < 4; ++u) for (int e = 0; e < 16; ++e) c[u][e] = 0.f;
    t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u], 0, 0, 0);
    }
    t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0.f;
    for (int u = 0; u < 4; ++u) s += c[u][0] + c[u][7];
    if (s == 123.456f) *sink = s;
  }
  return t1 - t0;
}

__device__ __forceinline__ u64 valu_loop(int iters, float* sink) {
  float x[8];
  for (int k = 0; k < 8; ++k) x[k] = 0.01f * (threadIdx.x + k);
  const float c = -0.37f, d = 0.25f;
  float rs = 0.f;
  u64 t0 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {     // one "entry": fma, exp, add, two more plain ops (stand-ins for the split instructions)
      float p = __builtin_amdgcn_exp2f(__builtin_fmaf(c, x[k], d));
      rs += p;
      float h = __builtin_fmaf(p, 0.5f, x[k]);
      x[k] = __builtin_fmaf(h, 0.25f, p);
    }
  }
  u64 t1 = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F);
  float s = rs;
  for (int k = 0; k < 8; ++k) s += x[k];
  if (s == 123.456f) *sink = s;
  return t1 - t0;
}

// NV = VALU waves per SIMD (0 or 1), NM = MFMA waves per SIMD (0, 1 or 2); VALU waves come first (oldest), as the producers do
template <int SHAPE>
__global__ __launch_bounds__(768) void k_probe(int nv, int nm, int mf_iters, int va_iters, u64* out, float* sink) {
  const int wave = threadIdx.x >> 6;
  u64 dt = 0;
  int kind = -1;
  if (wave < 4 * nv) { dt = valu_loop(va_iters, sink); kind = 0; }
  else if (wave < 4 * (nv + nm)) { dt = mfma_loop<SHAPE>(mf_iters, sink); kind = 1; }
  if ((threadIdx.x & 63) == 0 && kind >= 0) atomicAdd(&out[kind * 2], dt), atomicAdd(&out[kind * 2 + 1], 1ull);
}

template <int SHAPE>
static void run(const char* name, int nv, int nm, int mf_iters, int va_iters, u64* dout, float* sink) {
  u64 h[4] = {0, 0, 0, 0};
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(dout, 0, sizeof(h)));
    hipLaunchKernelGGL(k_probe<SHAPE>, dim3(256), dim3(768), 0, 0, nv, nm, mf_iters, va_iters, dout, sink);
    CK(hipDeviceSynchronize());
  }
  CK(hipMemcpy(h, dout, sizeof(h), hipMemcpyDeviceToHost));
  const int per_it = SHAPE == 16 ? 24 : 12;
  printf("%-44s", name);
  if (h[3]) printf("  MFMA wave: %6.1f cycles per MFMA (%5.1f per 16x16x32-equivalent)", (double)h[2] / h[3] / ((double)mf_iters * per_it),
                   (double)h[2] / h[3] / ((double)mf_iters * per_it) / (SHAPE == 16 ? 1.0 : 2.0));
  if (h[1]) printf("  VALU wave: %6.1f cycles per entry (5 instructions, one of them exp)", (double)h[0] / h[1] / ((double)va_iters * 8));
  printf("\n");
}

int main() {
  u64* dout; float* sink;
  CK(hipMalloc(&dout, 64)); CK(hipMalloc(&sink, 64));
  // sized so that the two loops take about the same time when they do not disturb each other
  const int MI = 4000;
  printf("== 16x16x32 f16 ==\n");
  run<16>("MFMA only, 1 wave/SIMD", 0, 1, MI, 0, dout, sink);
  run<16>("MFMA only, 2 waves/SIMD", 0, 2, MI, 0, dout, sink);
  run<16>("VALU only, 1 wave/SIMD", 1, 0, 0, 4 * MI, dout, sink);
  run<16>("1 VALU + 1 MFMA wave/SIMD", 1, 1, MI, 4 * MI, dout, sink);
  run<16>("1 VALU + 2 MFMA waves/SIMD (the contraction)", 1, 2, MI, 8 * MI, dout, sink);
  printf("== 32x32x16 f16 ==\n");
  run<32>("MFMA only, 1 wave/SIMD", 0, 1, MI, 0, dout, sink);
  run<32>("MFMA only, 2 waves/SIMD", 0, 2, MI, 0, dout, sink);
  run<32>("1 VALU + 1 MFMA wave/SIMD", 1, 1, MI, 4 * MI, dout, sink);
  run<32>("1 VALU + 2 MFMA waves/SIMD", 1, 2, MI, 8 * MI, dout, sink);
  return 0;
}
