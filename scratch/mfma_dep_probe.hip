// mfma_dep_probe.hip -- how fast does one wave issue v_mfma_f32_16x16x32_f16 when consecutive MFMAs accumulate into the same
// registers?  CHAINS independent accumulators per wave, walked round-robin: CHAINS = 1 is a pure dependent chain
// (acc = mfma(a, b, acc) back to back), CHAINS = 2 what the 3-product split kernels do (two column blocks alternate),
// CHAINS = 4 / 8 what a reordering would give.  Prints MFMAs per microsecond and SIMD for 1 and 2 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 scratch/mfma_dep_probe.hip -o /tmp/mfma_dep ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f / (1 + i + threadIdx.x)); }
  f32x4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 24 / CHAINS; ++r)
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678f) out[threadIdx.x] = s;   // keep the work alive
}

template <int CHAINS>
static void run(float* out, int threads) {
  const int iters = 20000, grid = 256;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(threads), 0, 0, out, 100);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(k<CHAINS>, dim3(grid), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = 24.0 * iters;                         // MFMAs issued by one wave
  const int waves_per_simd = threads / 256;
  printf("chains %d, %d wave(s) per SIMD: %.3f ms -> %.1f MFMAs per us and SIMD (16 cycles each at 2.4 GHz = 150.0)\n", CHAINS,
         waves_per_simd, ms, per_wave * waves_per_simd / (ms * 1e3));
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  for (int threads : {256, 512}) {
    run<1>(out, threads); run<2>(out, threads); run<3>(out, threads); run<4>(out, threads); run<6>(out, threads); run<8>(out, threads);
  }
  return 0;
}
