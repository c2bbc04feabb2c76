// cu_ingest.hip -- how many bytes per clock one CU takes in from L2 / Infinity Cache / HBM with 1 KB-per-wave coalesced loads
// (16 B per lane), and puts out with 16 B-per-lane stores.  Diagnostic probe (scratch/), not part of the product.
//   hipcc --offload-arch=gfx950 -O3 cu_ingest.hip -o cu_ingest && ./cu_ingest
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// every workgroup walks `span` bytes starting at its own offset (wrapping inside `region` bytes), `iters` times 16 KB
template <int UNROLL>
__global__ __launch_bounds__(256) void k_read(const u32x4* __restrict__ buf, size_t region16, size_t stride16, int iters,
                                              u32x4* __restrict__ sink) {
  const size_t t = threadIdx.x;
  size_t at = ((size_t)blockIdx.x * stride16) % region16;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      size_t a = at + (size_t)u * 256 + t;
      if (a >= region16) a -= region16;
      v[u] = buf[a];
    }
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
    at += (size_t)UNROLL * 256;
    if (at >= region16) at -= region16;
  }
  if (acc.x == 0x12345678u) sink[t] = acc;   // never true: keeps the loads alive
}

__global__ __launch_bounds__(256) void k_write(u32x4* __restrict__ buf, size_t per_block16, int iters) {
  const size_t t = threadIdx.x;
  u32x4* p = buf + (size_t)blockIdx.x * per_block16;
  const u32x4 v = {1u, 2u, 3u, (unsigned)blockIdx.x};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int u = 0; u < 4; ++u) p[((size_t)it * 4 + u) * 256 + t] = v;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz_nom = 2.4;
  printf("device %s, %d CUs\n", prop.name, cus);
  const size_t big = (size_t)2 << 30;   // 2 GiB
  u32x4* buf; CK(hipMalloc(&buf, big));
  CK(hipMemset(buf, 1, big));
  u32x4* sink; CK(hipMalloc(&sink, 4096 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Case { const char* name; size_t region; size_t stride; };
  // region: bytes all workgroups wander in; stride: byte distance between the start offsets of consecutive workgroups
  const Case cases[] = {
      {"L2-resident   (every WG re-reads the same 1 MiB)", (size_t)1 << 20, 0},
      {"L2-shared     (2 MiB region, WG offsets 16 KiB apart)", (size_t)2 << 20, 16384},
      {"MALL          (24 MiB region, WG offsets 64 KiB apart)", (size_t)24 << 20, 65536},
      {"HBM streaming (2 GiB region, disjoint 1 MiB per WG)", big, (size_t)1 << 20},
  };
  for (const Case& c : cases) {
    for (int wgs_per_cu : {1, 2, 3, 4, 8}) {
      const int blocks = cus * wgs_per_cu;
      const int iters = c.region == big ? 64 / 4 * 4 : 256;   // x 8 unroll x 4 KB per WG-iteration
      const int reps = 5;
      float best = 1e30f;
      for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_read<8>, dim3(blocks), dim3(256), 0, 0, buf, c.region / 16, c.stride / 16, iters, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        const float ms = time_ms(e0, e1);
        if (ms < best) best = ms;
      }
      const double bytes = (double)blocks * iters * 8 * 4096;
      printf("read  %-58s %d WG/CU: %7.1f GB/s per CU = %5.1f B/clk @%.1f GHz, chip %6.2f TB/s (%.3f ms)\n", c.name,
             wgs_per_cu, bytes / best / 1e6 / cus, bytes / best / 1e6 / cus / ghz_nom, ghz_nom, bytes / best / 1e9, best);
    }
  }
  // is ~25 GB/s the limit of ONE CU for HBM-latency loads, or only the chip's HBM rate divided by 256?  few workgroups:
  for (int blocks : {8, 16, 32, 64, 128}) {
    const int iters = 256;
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_read<8>, dim3(blocks), dim3(256), 0, 0, buf, big / 16, ((size_t)8 << 20) / 16, iters, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      const float ms = time_ms(e0, e1);
      if (ms < best) best = ms;
    }
    const double bytes = (double)blocks * iters * 8 * 4096;
    printf("read  HBM streaming, %3d workgroups of 256 threads (8 x 16 B per lane in flight): %7.1f GB/s per workgroup, chip %6.2f TB/s\n",
           blocks, bytes / best / 1e6 / blocks, bytes / best / 1e9);
  }
  for (int wgs_per_cu : {1, 2, 3, 4, 8}) {
    const int blocks = cus * wgs_per_cu;
    const size_t per_block = big / blocks / 16384 * 16384;
    const int iters = (int)(per_block / 16384);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, buf, per_block / 16, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      const float ms = time_ms(e0, e1);
      if (ms < best) best = ms;
    }
    const double bytes = (double)blocks * iters * 16384;
    printf("write HBM streaming %d WG/CU: %7.1f GB/s per CU = %5.1f B/clk, chip %6.2f TB/s (%.3f ms)\n", wgs_per_cu,
           bytes / best / 1e6 / cus, bytes / best / 1e6 / cus / ghz_nom, bytes / best / 1e9, best);
  }
  return 0;
}
