// stein_ablate.h -- hooks of the diagnostic builds (scratch/build_variant.py ... -DSTEIN_STAMPS): per-phase cycle sums taken
// with s_memtime.  In the shipped library every macro here expands to nothing and no stamp executes; the stamp values
// leave a diagnostic kernel only through g_dp_stamps, which nothing else reads.
#pragma once
#ifdef STEIN_STAMPS
__device__ unsigned long long g_dp_stamps[10];   // k_distance_panel, summed over waves: [0] waiting for streamed loads,
                                                // [1] LDS fragment reads + MFMAs, [2] issuing the next requests,
                                                // [3] epilogue, [4] panel switches (barriers + load), [5] strips, [6] waves,
                                                // [7] shader-clock ticks and [8] 100 MHz real-time ticks of the waves' lifetimes
extern "C" int stein_debug_dp(unsigned long long* host_out, int reset) {
  if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dp_stamps), sizeof(g_dp_stamps)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[10] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_dp_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
__device__ unsigned long long g_dp_wg[3 * 256];   // per logical workgroup id: start, end (100 MHz ticks) of wave 0, its strips
extern "C" int stein_debug_dp_wg(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dp_wg), sizeof(g_dp_wg)) == hipSuccess ? 0 : -1;
}
#define DP_STAMP_WG(p, w, lane)                                            \
  do {                                                                     \
    if ((w) == 0 && (lane) == 0 && (p) < 256) {                            \
      const unsigned long long r1_ = __builtin_amdgcn_s_memrealtime();     \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                  \
      g_dp_wg[3 * (p)] = dpst_r0; g_dp_wg[3 * (p) + 1] = r1_; g_dp_wg[3 * (p) + 2] = dpst_acc[5]; \
    }                                                                      \
  } while (0)
#define DP_STAMP_DECL unsigned long long dpst_acc[6] = {0, 0, 0, 0, 0, 0}; unsigned long long dpst_last = __builtin_amdgcn_s_memtime(); const unsigned long long dpst_c0 = dpst_last, dpst_r0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#define DP_STAMP(k)                                                        \
  do {                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                     \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();          \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                    \
    dpst_acc[k] += now_ - dpst_last;                                       \
    dpst_last = now_;                                                      \
    __builtin_amdgcn_sched_barrier(0);                                     \
  } while (0)
#define DP_STAMP_COUNT(k) do { dpst_acc[k] += 1ull; } while (0)
#define DP_STAMP_FLUSH(lane)                                               \
  do {                                                                     \
    if ((lane) == 0) {                                                     \
      for (int k_ = 0; k_ < 6; ++k_) atomicAdd(&g_dp_stamps[k_], dpst_acc[k_]); \
      atomicAdd(&g_dp_stamps[6], 1ull);                                    \
      const unsigned long long c1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime(); \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                  \
      atomicAdd(&g_dp_stamps[7], c1_ - dpst_c0); atomicAdd(&g_dp_stamps[8], r1_ - dpst_r0); \
    }                                                                      \
  } while (0)
#else
#define DP_STAMP_DECL do {} while (0)
#define DP_STAMP(k) do {} while (0)
#define DP_STAMP_COUNT(k) do {} while (0)
#define DP_STAMP_FLUSH(lane) do {} while (0)
#define DP_STAMP_WG(p, w, lane) do {} while (0)
#endif
