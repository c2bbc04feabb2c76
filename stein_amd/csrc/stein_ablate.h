// stein_ablate.h -- hooks of the diagnostic builds (scratch/build_variant.py ... -DSTEIN_STAMPS): per-phase cycle sums taken
// with s_memtime.  In the shipped library every macro here expands to nothing and no stamp executes; the stamp values
// leave a diagnostic kernel only through g_dp_stamps, which nothing else reads.
#pragma once
// timing-only ablations of k_distance_panel (wrong results; -DSTEIN_DP_ABL_NOSTORE: the regular strips keep their values alive
// but store nothing)
#ifdef STEIN_DP_ABL_NOSTORE
#define DP_STORE16(ptr, val) asm volatile("" :: "v"((val).x), "v"((val).y), "v"((val).z), "v"((val).w), "v"(ptr))
#elif defined(STEIN_DP_NT_STORE)   // (experiment: streaming stores for D)
typedef float dp_f4v __attribute__((ext_vector_type(4)));
#define DP_STORE16(ptr, val) __builtin_nontemporal_store(dp_f4v{(val).x, (val).y, (val).z, (val).w}, reinterpret_cast<dp_f4v*>(ptr))
#else
#define DP_STORE16(ptr, val) (*reinterpret_cast<float4*>(ptr) = (val))
#endif
#ifdef STEIN_STAMPS
__device__ unsigned long long g_dp_stamps[12];   // k_distance_panel, summed over waves: [0] waiting for streamed loads,
                                                // [1] LDS fragment reads + MFMAs, [2] issuing the next requests,
                                                // [3] epilogue, [4] panel switches (barriers + load), [5] strips, [6] waves,
                                                // [7] shader-clock ticks and [8] 100 MHz real-time ticks of the waves' lifetimes
extern "C" int stein_debug_dp(unsigned long long* host_out, int reset) {
  if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dp_stamps), sizeof(g_dp_stamps)) != hipSuccess) return -1;
  if (reset) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_dp_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
__device__ unsigned long long g_dp_wg[6 * 256];   // per logical workgroup id: start, end (100 MHz ticks) of wave 0, its strips, its shader-clock ticks, the time its own segments were done, its strips up to then
extern "C" int stein_debug_dp_wg(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dp_wg), sizeof(g_dp_wg)) == hipSuccess ? 0 : -1;
}
#define DP_STAMP_WG(p, w, lane)                                            \
  do {                                                                     \
    if ((w) == 0 && (lane) == 0 && (p) < 256) {                            \
      const unsigned long long r1_ = __builtin_amdgcn_s_memrealtime();     \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                  \
      const unsigned long long c1w_ = __builtin_amdgcn_s_memtime();       \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                  \
      g_dp_wg[6 * (p)] = dpst_r0; g_dp_wg[6 * (p) + 1] = r1_; g_dp_wg[6 * (p) + 2] = dpst_acc[5]; g_dp_wg[6 * (p) + 3] = c1w_ - dpst_c0; \
      g_dp_wg[6 * (p) + 4] = dpst_own_t; g_dp_wg[6 * (p) + 5] = dpst_own_n; \
    }                                                                      \
  } while (0)
__device__ unsigned long long g_dp_slow[8 * 2048];   // per wave: its slowest strip: cycles total, waiting, MFMA phase, requests, epilogue, segment, strip, when (100 MHz)
extern "C" int stein_debug_dp_slow(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dp_slow), sizeof(g_dp_slow)) == hipSuccess ? 0 : -1;
}
#define DP_STRIP_BEGIN do { for (int k_ = 0; k_ < 4; ++k_) dpst_snap[k_] = dpst_acc[k_]; } while (0)
#define DP_STRIP_END(g, s)                                                  \
  do {                                                                     \
    unsigned long long d_[4], tot_ = 0;                                    \
    for (int k_ = 0; k_ < 4; ++k_) { d_[k_] = dpst_acc[k_] - dpst_snap[k_]; tot_ += d_[k_]; } \
    if (tot_ > dpst_slow[0]) {                                             \
      dpst_slow[0] = tot_; for (int k_ = 0; k_ < 4; ++k_) dpst_slow[1 + k_] = d_[k_];          \
      dpst_slow[5] = (unsigned long long)(g); dpst_slow[6] = (unsigned long long)(s);          \
      dpst_slow[7] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F);     \
    }                                                                      \
  } while (0)
#define DP_SLOW_FLUSH(p, w, lane)                                           \
  do {                                                                     \
    if ((lane) == 0 && (p) < 256) for (int k_ = 0; k_ < 8; ++k_) g_dp_slow[((p) * 8 + (w)) * 8 + k_] = dpst_slow[k_]; \
  } while (0)
#define DP_STAMP_OWN_DONE do { if (!dpst_own_t) { dpst_own_t = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); dpst_own_n = dpst_acc[5]; } } while (0)
#define DP_STAMP_DECL unsigned long long dpst_snap[4] = {0, 0, 0, 0}, dpst_slow[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long dpst_own_t = 0, dpst_own_n = 0; unsigned long long dpst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long dpst_last = __builtin_amdgcn_s_memtime(); const unsigned long long dpst_c0 = dpst_last, dpst_r0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
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
      atomicAdd(&g_dp_stamps[9], dpst_acc[6]); atomicAdd(&g_dp_stamps[10], dpst_acc[7]); \
    }                                                                      \
  } while (0)
#else
#define DP_STAMP_DECL do {} while (0)
#define DP_STAMP(k) do {} while (0)
#define DP_STAMP_COUNT(k) do {} while (0)
#define DP_STAMP_FLUSH(lane) do {} while (0)
#define DP_STAMP_WG(p, w, lane) do {} while (0)
#define DP_STAMP_OWN_DONE do {} while (0)
#define DP_STRIP_BEGIN do {} while (0)
#define DP_STRIP_END(g, s) do {} while (0)
#define DP_SLOW_FLUSH(p, w, lane) do {} while (0)
#endif
