// stein_ablate.h -- hooks of the diagnostic builds (scratch/build_variant.py ... -DSTEIN_STAMPS): per-phase cycle sums taken
// with s_memtime.  In the shipped library every macro here expands to nothing and no stamp executes; the stamp values
// leave a diagnostic kernel only through g_dp_stamps, which nothing else reads.
#pragma once
#ifdef STEIN_ABLATE_DPANEL   // ---- stein_dpanel.hip (k_distance_panel) ----------------------------------------------
// timing-only ablations of k_distance_panel (wrong results; -DSTEIN_DP_ABL_NOSTORE: the regular strips keep their values alive
// but store nothing)
#ifdef STEIN_DP_ABL_NOSTORE
#define DP_STORE16(ptr, val) asm volatile("" :: "v"((val).x), "v"((val).y), "v"((val).z), "v"((val).w))   /* (the address is not kept alive: inside the epilogue's lambda an asm operand does not capture it) */
#elif defined(STEIN_DP_SC1_STORE)   // (experiment: write-through stores for D)
typedef float dp_f4w __attribute__((ext_vector_type(4)));
#define DP_STORE16(ptr, val) do { const dp_f4w v_ = {(val).x, (val).y, (val).z, (val).w}; float* p_ = (ptr); asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p_), "v"(v_) : "memory"); } while (0)
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
__device__ unsigned long long g_dp_stamps_w[2048 * 12];   // the same per wave (workgroup * 8 + wave): what the kernel writes
extern "C" int stein_debug_dp(unsigned long long* host_out, int reset) {
  static unsigned long long all[2048 * 12];
  if (host_out) {
    if (hipMemcpyFromSymbol(all, HIP_SYMBOL(g_dp_stamps_w), sizeof(all)) != hipSuccess) return -1;
    for (int k = 0; k < 12; ++k) { host_out[k] = 0; for (int w = 0; w < 2048; ++w) host_out[k] += all[w * 12 + k]; }
  }
  if (reset) { for (int i = 0; i < 2048 * 12; ++i) all[i] = 0; if (hipMemcpyToSymbol(HIP_SYMBOL(g_dp_stamps_w), all, sizeof(all)) != hipSuccess) return -1; }
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
__device__ unsigned int g_dp_trace[2048 * 64];   // per wave: [0] strips done, [1 + (k & 62)] end of its k-th strip (100 MHz ticks since the wave began; a ring of the last 62)
extern "C" int stein_debug_dp_trace(unsigned int* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dp_trace), sizeof(g_dp_trace)) == hipSuccess ? 0 : -1;
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
    {                                                                      \
      const unsigned long long tr_ = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); \
      if ((threadIdx.x & 63) == 0) dpst_trace[(threadIdx.x >> 6) * 64 + 1 + (dpst_ntrace % 62u)] = (unsigned int)(tr_ - dpst_r0); \
      ++dpst_ntrace;                                                       \
    }                                                                      \
  } while (0)
#define DP_SLOW_FLUSH(p, w, lane)                                           \
  do {                                                                     \
    if ((lane) == 0 && (p) < 256) for (int k_ = 0; k_ < 8; ++k_) g_dp_slow[((p) * 8 + (w)) * 8 + k_] = dpst_slow[k_]; \
    if ((lane) == 0 && (p) < 256) {                                        \
      g_dp_trace[((p) * 8 + (w)) * 64] = dpst_ntrace;                      \
      for (int k_ = 1; k_ < 63; ++k_) g_dp_trace[((p) * 8 + (w)) * 64 + k_] = dpst_trace[(w) * 64 + k_]; \
    }                                                                      \
  } while (0)
#define DP_STAMP_OWN_DONE do { if (!dpst_own_t) { dpst_own_t = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); dpst_own_n = dpst_acc[5]; } } while (0)
#define DP_STAMP_DECL __shared__ unsigned int dpst_trace[8 * 64]; unsigned int dpst_ntrace = 0u; unsigned long long dpst_snap[4] = {0, 0, 0, 0}, dpst_slow[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long dpst_own_t = 0, dpst_own_n = 0; unsigned long long dpst_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long dpst_last = __builtin_amdgcn_s_memtime(); const unsigned long long dpst_c0 = dpst_last, dpst_r0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
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
// (plain stores into per-wave slots, summed by stein_debug_dp on the host: with atomicAdd onto twelve shared words, every wave's
// exit queued a dozen same-address atomics in front of the loads of the waves still running -- a 100-200 us tail that was the
// stamp code's, not the kernel's)
#define DP_STAMP_FLUSH(lane)                                               \
  do {                                                                     \
    if ((lane) == 0) {                                                     \
      unsigned long long* o_ = g_dp_stamps_w + (size_t)(blockIdx.x * 8 + (threadIdx.x >> 6)) * 12; \
      for (int k_ = 0; k_ < 6; ++k_) o_[k_] += dpst_acc[k_];               \
      o_[6] += 1ull;                                                       \
      const unsigned long long c1_ = __builtin_amdgcn_s_memtime(), r1_ = __builtin_amdgcn_s_memrealtime(); \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                  \
      o_[7] += c1_ - dpst_c0; o_[8] += r1_ - dpst_r0;                       \
      o_[9] += dpst_acc[6]; o_[10] += dpst_acc[7];                          \
    }                                                                      \
  } while (0)
#elif defined(STEIN_WGEND)
// the lightest probe: begin and end (100 MHz ticks) of every wave of the otherwise shipped kernel, one plain store per wave
__device__ unsigned long long g_dp_wave_span[2 * 2048];
extern "C" int stein_debug_dp_span(unsigned long long* host_out) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dp_wave_span), sizeof(g_dp_wave_span)) == hipSuccess ? 0 : -1;
}
#define DP_STAMP_DECL const unsigned long long dpst_r0 = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F)
#define DP_STAMP(k) do {} while (0)
#define DP_STAMP_COUNT(k) do {} while (0)
#define DP_STAMP_FLUSH(lane) do {} while (0)
#define DP_STAMP_WG(p, w, lane)                                            \
  do {                                                                     \
    if ((lane) == 0 && (p) < 256) {                                        \
      const unsigned long long r1_ = __builtin_amdgcn_s_memrealtime();     \
      __builtin_amdgcn_s_waitcnt(0xC07F);                                  \
      g_dp_wave_span[2 * ((p) * 8 + (w))] = dpst_r0; g_dp_wave_span[2 * ((p) * 8 + (w)) + 1] = r1_; \
    }                                                                      \
  } while (0)
#define DP_STAMP_OWN_DONE do {} while (0)
#define DP_STRIP_BEGIN do {} while (0)
#define DP_STRIP_END(g, s) do {} while (0)
#define DP_SLOW_FLUSH(p, w, lane) do {} while (0)
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

#endif   // STEIN_ABLATE_DPANEL

#ifdef STEIN_ABLATE_X3
// ---- stein_x3.hip (k_distance_x3, k_phi_x3fs): per-phase cycle sums of one wave per role and workgroup --------------------
#ifdef STEIN_STAMPS
__device__ u64 g_stamps[8];
#define STAMP(k)                                                                  \
  do {                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                            \
    const u64 now_ = __builtin_amdgcn_s_memtime();                                \
    __builtin_amdgcn_s_waitcnt(0xC07F);                                           \
    st_acc[k] += now_ - st_last;                                                  \
    st_last = now_;                                                               \
    __builtin_amdgcn_sched_barrier(0);                                            \
  } while (0)
__device__ u64 g_wave[24];   // contraction, per wave of the workgroup: [2 w] cycles working, [2 w + 1] cycles at the stage barrier
extern "C" int stein_debug_waves(u64* host_out, int reset) {
  if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wave), sizeof(u64) * 24) != hipSuccess) return -1;
  if (reset) { u64 z[24] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wave), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
__device__ u64 g_wg[3 * 1024];   // contraction, per workgroup: real-time ticks (100 MHz) at kernel entry, main-loop start, main-loop end
extern "C" int stein_debug_wg(u64* host_out, int reset) {
  if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wg), sizeof(u64) * 3 * 1024) != hipSuccess) return -1;
  if (reset) { static u64 z[3 * 1024]; if (hipMemcpyToSymbol(HIP_SYMBOL(g_wg), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
__device__ u64 g_xcd[16];   // distance pass: [x] latest workgroup end on XCD x (blockIdx % 8), [8 + x] earliest end; 100 MHz ticks
extern "C" int stein_debug_xcd(u64* host_out, int reset) {
  if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_xcd), sizeof(u64) * 16) != hipSuccess) return -1;
  if (reset) { u64 z[16]; for (int i = 0; i < 16; ++i) z[i] = i < 8 ? 0ull : ~0ull; if (hipMemcpyToSymbol(HIP_SYMBOL(g_xcd), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
__device__ u64 g_clock[2];   // sums over the sampled waves of (shader-clock ticks, 100 MHz real-time ticks) inside the contraction's main loop
extern "C" int stein_debug_clock(u64* host_out, int reset) {
  if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_clock), sizeof(u64) * 2) != hipSuccess) return -1;
  if (reset) { u64 z[2] = {0, 0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_clock), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
extern "C" int stein_debug_stamps(u64* host_out, int reset) {
  if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(u64) * 8) != hipSuccess) return -1;
  if (reset) { u64 z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#define X3_STAMP_DECL \
  u64 st_acc[6] = {0, 0, 0, 0, 0, 0}; \
  u64 st_last = __builtin_amdgcn_s_memtime();
#define X3_STAMP_FLUSH_DISTANCE \
  STAMP(3); \
  if (t == 0) { \
  for (int k = 0; k < 4; ++k) atomicAdd(&g_stamps[k], st_acc[k]); \
  atomicAdd(&g_stamps[7], 1ull); \
  const u64 rt = __builtin_amdgcn_s_memrealtime(); \
  __builtin_amdgcn_s_waitcnt(0xC07F); \
  atomicMax(reinterpret_cast<unsigned long long*>(&g_xcd[blockIdx.x & 7]), (unsigned long long)rt); \
  atomicMin(reinterpret_cast<unsigned long long*>(&g_xcd[8 + (blockIdx.x & 7)]), (unsigned long long)rt); \
  }
#define X3_STAMP_FLUSH_PRODUCER \
  if (t == 0) { \
  for (int k = 0; k < 3; ++k) atomicAdd(&g_stamps[k], st_acc[k]); \
  atomicAdd(&g_stamps[7], 1ull); \
  } \
  if ((t & 63) == 0) { atomicAdd(&g_wave[2 * (t >> 6)], st_acc[0] + st_acc[1] + st_acc[5]); atomicAdd(&g_wave[2 * (t >> 6) + 1], st_acc[2]); }
#define X3_STAMP_WG_ENTRY \
  if (t == 256 && blockIdx.x < 1024) { g_wg[3 * blockIdx.x] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); }
#define X3_STAMP_DECL_CONSUMER \
  u64 st_acc[6] = {0, 0, 0, 0, 0, 0}; \
  u64 st_last = __builtin_amdgcn_s_memtime(); \
  const u64 clk0 = st_last, rt0 = __builtin_amdgcn_s_memrealtime(); \
  __builtin_amdgcn_s_waitcnt(0xC07F); \
  if (t == 256 && blockIdx.x < 1024) g_wg[3 * blockIdx.x + 1] = rt0;
#define X3_STAMP_FLUSH_CONSUMER \
  if (t == 256) { \
  const u64 clk1 = __builtin_amdgcn_s_memtime(), rt1 = __builtin_amdgcn_s_memrealtime(); \
  __builtin_amdgcn_s_waitcnt(0xC07F); \
  atomicAdd(&g_clock[0], clk1 - clk0); \
  atomicAdd(&g_clock[1], rt1 - rt0); \
  if (blockIdx.x < 1024) g_wg[3 * blockIdx.x + 2] = rt1; \
  } \
  if ((t & 63) == 0) { atomicAdd(&g_wave[2 * (t >> 6)], st_acc[0] + st_acc[3] + st_acc[5]); atomicAdd(&g_wave[2 * (t >> 6) + 1], st_acc[4]); } \
  if (t == 256) \
  for (int k = 3; k < 5; ++k) atomicAdd(&g_stamps[k], st_acc[k]); \
  if (t == 512) { \
  atomicAdd(&g_stamps[5], st_acc[3] + st_acc[4]); \
  atomicAdd(&g_stamps[6], st_acc[0]); \
  }
#else
#define STAMP(k) do {} while (0)
#define X3_STAMP_DECL do {} while (0)
#define X3_STAMP_FLUSH_DISTANCE do {} while (0)
#define X3_STAMP_FLUSH_PRODUCER do {} while (0)
#define X3_STAMP_WG_ENTRY do {} while (0)
#define X3_STAMP_DECL_CONSUMER do {} while (0)
#define X3_STAMP_FLUSH_CONSUMER do {} while (0)
#endif
#endif   // STEIN_ABLATE_X3
