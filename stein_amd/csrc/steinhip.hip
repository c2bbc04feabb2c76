// steinhip.hip -- gfx950 (MI355X / CDNA4) kernels and C ABI for the SVGD particle update.
//
// Pipeline for one step on one rank (rows [row0, row0+n_local) of n particles, d parameters):
//   k_rownorms      r_i = |theta_i|^2                                   HBM-bound, 4nd bytes
//   k_distance      D = r_i + r_j - 2 theta theta^T  (fp32 MFMA 32x32x2, 128x128 tiles, LDS staged)
//   k_hist x3       3-level radix select over the fp32 bit patterns of D (exact median)
//   k_resolve x3    1-wave kernel that walks the histogram; last level -> median, h^2
//   k_phi_partial   P = exp(-D/2h^2) built on the fly from the D tile, O += P.[G|theta] (fp32 MFMA),
//                   rowsum(P) on the VALU -- K is never materialised
//   k_phi_finish    phi = (O_G + (rowsum*theta - O_T)/h^2)/n, per-block partial |phi|^2 (fp64)
//   k_sum_partials  deterministic reduction of the partials
//   k_apply_*       clip + Adagrad/Adam + theta += step, one streaming pass
// Reference formulae: see include/steinhip.h for the file:line of each stage.
//
// Everything is launched on the caller's stream; nothing here synchronises with the host.

#include <hip/hip_runtime.h>
#include <type_traits>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>
#include <math.h>

#include "stein_common.h"
#include "stein_x3.h"

#include <vector>

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int stein_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define fail stein_fail

extern "C" int stein_version(void) { return STEIN_VERSION; }
extern "C" const char* stein_last_error(void) { return g_err; }

constexpr int LDK = BK + 4;    // LDS row stride (floats) of a [rows][k] fp32 tile: +16 B keeps ds_read_b128 conflict-free

// Row-of-k tile loader: rows `rbase + lr + 32p`, k range [k0 + lc, +4).  VEC requires d % 4 == 0.
template <bool VEC>
__device__ __forceinline__ void load_rows_k(const float* __restrict__ M, int nrows, int d, int rbase, int k0,
                                            int lr, int lc, float4 (&v)[4]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = rbase + lr + 32 * p;
    const int k = k0 + lc;
    if (VEC) {
      v[p] = ld4_or_zero(M + (size_t)row * d + k, row < nrows && k < d);
    } else {
      const float* src = M + (size_t)row * d + k;
      const bool rok = row < nrows;
      v[p].x = (rok && k + 0 < d) ? src[0] : 0.f;
      v[p].y = (rok && k + 1 < d) ? src[1] : 0.f;
      v[p].z = (rok && k + 2 < d) ? src[2] : 0.f;
      v[p].w = (rok && k + 3 < d) ? src[3] : 0.f;
    }
  }
}

__device__ __forceinline__ void store_rows_k(float* S, int lr, int lc, const float4 (&v)[4]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) *reinterpret_cast<float4*>(S + (lr + 32 * p) * LDK + lc) = v[p];
}

__device__ __forceinline__ float comp(const float4& v, int t) {
  return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w));
}

// ------------------------------------------------------------------------------------------------
// k_rownorms: one wave per row
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float elem_f32(const float* p) { return *p; }
__device__ __forceinline__ float elem_f32(const unsigned short* p) { return __uint_as_float((u32)*p << 16); }  // bf16 bits

template <typename TIN>
__global__ __launch_bounds__(256) void k_rownorms(const TIN* __restrict__ T, int n, int d, float* __restrict__ r) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (wave >= n) return;
  const TIN* row = T + (size_t)wave * d;
  float s = 0.f;
  for (int k = lane; k < d; k += 64) {
    const float x = elem_f32(row + k);
    s = fmaf(x, x, s);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) r[wave] = s;
}

// ------------------------------------------------------------------------------------------------
// radix select (keys, state and hist_add live in stein_common.h)
// ------------------------------------------------------------------------------------------------
__global__ void k_sel_init(SelState* st, u64 total) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const u32 even = (total & 1ull) ? 0u : 1u;
    st->rank[0] = even ? total / 2 - 1 : total / 2;
    st->rank[1] = total / 2;
    st->prefix[0] = st->prefix[1] = 0u;
    st->diverged = 0u;
    st->even = even;
    st->median = st->h2 = st->lo = st->hi = 0.f;
  }
}

// ------------------------------------------------------------------------------------------------
// k_distance: D = r_i + r_j - 2 T T^T on the fp32 matrix cores.
//   A operand = rows of the row block, B operand = rows of the column block, both k-contiguous, so both
//   tiles live in LDS as [row][k] and a lane fetches 4 consecutive k with one ds_read_b128.  The k order
//   inside an MFMA step is permuted identically for A and B (lane half h, step t <-> k = 8kk + 4h + t).
//   Every D_ij runs the same k-ordered fma chain with the operands swapped for D_ji, so D is bitwise
//   symmetric.  SYM (the block is the whole n x n matrix): only tiles on or above the diagonal are
//   computed and each off-diagonal tile is also stored transposed -- half the MFMA work.
//   hist0 != NULL: the level-0 radix-select histogram (top 11 key bits) is taken from the accumulators
//   here instead of re-reading D; a mirrored tile counts twice.
// ------------------------------------------------------------------------------------------------
template <bool VEC, bool SYM>
__global__ __launch_bounds__(NTHREADS, 2) void k_distance(const float* __restrict__ T, const float* __restrict__ r,
                                                       float* __restrict__ D, int n, int d, int row0, int n_local,
                                                       long ldD, int tiles_m, int tiles_n, u64* __restrict__ hist0,
                                                       SpecState* __restrict__ spec, u64* __restrict__ spec_buf) {
  constexpr int kStage = (BM + BN) * LDK, kEpi = EPI_LDS_BYTES / 4;   // main loop tiles; the epilogue reuses the array
  __shared__ __attribute__((aligned(16))) float smem[kStage > kEpi ? kStage : kEpi];
  float* As = smem;
  float* Bs = smem + BM * LDK;

  int tile_m, tile_n;
  if (!distance_tile<SYM>(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tile_m, tile_n)) return;

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wy = wid >> 1, wx = wid & 1;
  const int lr = t >> 3, lc = (t & 7) * 4;
  const int arow0 = row0 + tile_m * BM;  // global particle index of the tile's first row
  const int brow0 = tile_n * BN;
  const EpiPrefetch pf = distance_epilogue_prefetch(r, n, row0, n_local, tile_m, tile_n, spec);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float4 ra[4], rb[4];
  load_rows_k<VEC>(T, n, d, arow0, 0, lr, lc, ra);
  load_rows_k<VEC>(T, n, d, brow0, 0, lr, lc, rb);

  const int l31 = lane & 31, h4 = (lane >> 5) * 4;
  for (int k0 = 0; k0 < d; k0 += BK) {
    store_rows_k(As, lr, lc, ra);
    store_rows_k(Bs, lr, lc, rb);
    __syncthreads();
    if (k0 + BK < d) {  // next tile's loads fly under this tile's MFMAs
      load_rows_k<VEC>(T, n, d, arow0, k0 + BK, lr, lc, ra);
      load_rows_k<VEC>(T, n, d, brow0, k0 + BK, lr, lc, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        a[i] = *reinterpret_cast<const float4*>(As + (wy * 64 + i * 32 + l31) * LDK + kk * 8 + h4);
        b[i] = *reinterpret_cast<const float4*>(Bs + (wx * 64 + i * 32 + l31) * LDK + kk * 8 + h4);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(a[i], s), comp(b[j], s), acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  // the staging tiles are dead (every wave is past the loop's last barrier): 8 KB of them hold the level-0 histogram
  distance_epilogue<SYM>(acc, reinterpret_cast<u32*>(smem), D, n, n_local, ldD, tile_m, tile_n, hist0, pf, 2.f,
                         spec, spec_buf);
}

// ------------------------------------------------------------------------------------------------
// radix select: histogram pass, resolve (keys, state and hist_add are defined above k_distance)
// ------------------------------------------------------------------------------------------------
// ---- chained form of the radix select (fused call only): the histogram passes resolve the earlier levels
// themselves, so the fused call launches no k_resolve between them (and nothing at all between them matters when the
// speculative window hit: every launch is a few microseconds even when it returns at once).
// Block-wide (256 threads): the select state after `levels` resolved levels, computed from the INITIAL state in
// *st (ranks set by k_median_init, prefixes 0 -- nothing writes *st until resolve_all_body) and the global histograms.
// FRESH: the histograms may hold atomics of THIS launch (the last workgroup out, the barrier of k_hist_all): device-scope
// loads.  Otherwise earlier launches wrote them and plain loads do (served by the L2: a few thousand workgroups reading
// the same 16 KB with device-scope loads queued on the handful of memory channels that hold it, ~50 us per pass at C2).
struct ChainState { u32 prefix[2]; u64 rank[2]; bool two; };
template <bool FRESH>
__device__ __attribute__((noinline)) ChainState chain_resolve(const u64* hist_all, int levels, const SelState* st) {
  __shared__ u64 c_part[256];
  __shared__ u32 c_bin[2];
  __shared__ u64 c_rest[2];
  const int t = threadIdx.x;
  ChainState cs;
  cs.prefix[0] = cs.prefix[1] = 0u;
  cs.rank[0] = st->rank[0]; cs.rank[1] = st->rank[1];
  cs.two = false;
  for (int level = 0; level < levels; ++level) {
    const int bits = level == 2 ? 10 : 11;
    const u64* hl = hist_all + (size_t)level * 2 * STEIN_HIST_BINS;
    for (int tg = 0; tg < 2; ++tg) {
      const u64* src = hl + ((cs.two && tg == 1) ? STEIN_HIST_BINS : 0);
      u64 mine[8], sum = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) { mine[k] = FRESH ? load_fresh(src + t * 8 + k) : src[t * 8 + k]; sum += mine[k]; }
      c_part[t] = sum;
      __syncthreads();
      for (int o = 1; o < 256; o <<= 1) {   // inclusive scan
        u64 v = 0;
        if (t >= o) v = c_part[t - o];
        __syncthreads();
        c_part[t] += v;
        __syncthreads();
      }
      const u64 excl = c_part[t] - sum, rank = cs.rank[tg];
      if (t == 255 && rank >= c_part[255]) { c_bin[tg] = (u32)((1 << bits) - 1); c_rest[tg] = 0; }   // cannot happen: counts cover the rank
      if (rank >= excl && rank < excl + sum) {
        u64 cum = excl;
        int k = 0;
        while (k < 7 && cum + mine[k] <= rank) cum += mine[k++];
        c_bin[tg] = (u32)(t * 8 + k);
        c_rest[tg] = rank - cum;
      }
      __syncthreads();
      cs.prefix[tg] = (cs.prefix[tg] << bits) | c_bin[tg];
      cs.rank[tg] = c_rest[tg];
      __syncthreads();
    }
    cs.two = cs.prefix[0] != cs.prefix[1];
  }
  return cs;
}

// SYM (square symmetric block): only columns >= row are read; an off-diagonal entry counts twice.
// Level 0 sees every value and a handful of bins hold them all -> wave-merged adds.  Levels 1-2 only see the
// values inside the selected bin, spread over up to 2048 digits -> plain LDS atomics are cheaper.
// k_hist (staged calls): `hist` is this level's histogram and *st holds the state left by k_resolve.
// k_hist_all (fused call): all levels in one launch; what its last resolver needs to finish the select:
struct HistFinal {
  SelState* st;
  SpecState* sp;
  float* h2_out;
  float ln_n;
};
__device__ __attribute__((noinline)) void resolve_all_body(u64* hist_all, SelState* st, SpecState* sp, float ln_n, float* h2_out);   // below

// one histogram pass of a workgroup over a share of the block (units vb, vb + nvb, ... of "virtual workgroup" vb of nvb):
// digits of LEVEL into the LDS histogram h[2][STEIN_HIST_BINS] (zeroed by the caller), given the prefixes the earlier levels fixed
template <int LEVEL, bool SYM>
__device__ __forceinline__ void hist_pass_body(const float* __restrict__ D, long ldD, int n_local, int n, u32* h, u32 pa, u32 pb,
                                               bool two, long vb, long nvb) {
  const int lane = threadIdx.x & 63;
  // one unit = one [128][32] tile of the tile-major block (16 KB, 4 x 16 B per thread); SYM skips the tiles that lie
  // entirely below the diagonal.  (Keeping the loads of two more units in flight -- three register sets in rotation --
  // changed nothing at C2 or C3: the pass is not bound by the latency of its loads.)
  const long ntc = ldD >> 5;
  const int ntr = (n_local + DT_ROWS - 1) / DT_ROWS, ncol_tiles = (n + DT_COLS - 1) / DT_COLS;
  // SYM: row tile ti only has the units tj >= 4 ti (the others lie wholly below the diagonal).  Enumerated row by row over
  // ALL units and skipped, the strided shares were as uneven as they can be: with nvb a multiple of ncol_tiles every share
  // keeps ONE tj -- the share of the last column strip had 32 units at C3, that of the first none, and the launch lasted as
  // long as the longest (round 3: 0.19 ms per pass, twice the balanced time).  So the rows are folded as in the distance
  // pass: virtual row v = row v, then row ntr - 1 - v: W = 2 ncol_tiles - 4 (ntr - 1) useful units whatever v (the middle
  // row of an odd ntr stands alone), and L = v W + x runs over useful units only.
  const int vrows = SYM ? (ntr + 1) / 2 : ntr;
  const int W = SYM ? 2 * ncol_tiles - 4 * (ntr - 1) : ncol_tiles;
  const long units = (long)vrows * W;
  for (long u = vb; u < units; u += nvb) {
    int ti, tj;
    if (SYM) {
      const int v = (int)(u / W), x = (int)(u - (long)v * W), len0 = ncol_tiles - 4 * v;
      if (x < len0) { ti = v; tj = 4 * v + x; }
      else {
        ti = ntr - 1 - v;
        if (ti == v) continue;                       // the middle row has no partner
        tj = 4 * ti + (x - len0);
      }
      if (tj >= ncol_tiles) continue;                // (cannot happen: len0 + the partner's length = W)
    } else {
      ti = (int)(u / ncol_tiles); tj = (int)(u - (long)ti * ncol_tiles);
    }
    const float* tile = D + ((size_t)ti * ntc + tj) * DT_ELEMS;
    float4 v4[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v4[q] = *reinterpret_cast<const float4*>(tile + (threadIdx.x + 256 * q) * 4);
    // Interior unit: every entry exists and (SYM) lies strictly above the diagonal -- no per-entry bounds, no per-entry
    // weight.  All but the O(n / 32) units along the edges and the diagonal take this path (C3, three passes: 0.64 -> 0.57 ms).
    const bool interior = ti * DT_ROWS + DT_ROWS <= n_local && tj * DT_COLS + DT_COLS <= n &&
                          (!SYM || tj * DT_COLS >= ti * DT_ROWS + DT_ROWS);
    if (interior) {
      constexpr u32 WT = SYM ? 2u : 1u;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float x[4] = {v4[q].x, v4[q].y, v4[q].z, v4[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const u32 key = f32_key(x[e]);
          if (LEVEL == 0) {
            hist_add(h, key >> 21, true, lane, WT);
          } else {
            const u32 digit = LEVEL == 1 ? ((key >> 10) & 2047u) : (key & 1023u);
            const u32 hi = LEVEL == 1 ? (key >> 21) : (key >> 10);
            if (hi == pa) atomicAdd(&h[digit], WT);
            if (two && hi == pb) atomicAdd(&h[STEIN_HIST_BINS + digit], WT);
          }
        }
      }
      continue;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int f = threadIdx.x + 256 * q;          // float4 index inside the tile: row f / 8, columns 4 (f & 7) ..
      const int row = ti * DT_ROWS + (f >> 3), c0 = tj * DT_COLS + (f & 7) * 4;
      const float x[4] = {v4[q].x, v4[q].y, v4[q].z, v4[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = c0 + e;
        const bool inb = row < n_local && col < n && (!SYM || col >= row);
        const u32 key = f32_key(x[e]);
        const u32 w = (SYM && col != row) ? 2u : 1u;
        if (LEVEL == 0) {
          if (SYM) {
            hist_add(h, key >> 21, inb && col != row, lane, 2u);
            if (inb && col == row) atomicAdd(&h[key >> 21], 1u);
          } else {
            hist_add(h, key >> 21, inb, lane);
          }
        } else {
          const u32 digit = LEVEL == 1 ? ((key >> 10) & 2047u) : (key & 1023u);
          const u32 hi = LEVEL == 1 ? (key >> 21) : (key >> 10);
          if (inb && hi == pa) atomicAdd(&h[digit], w);
          if (two && inb && hi == pb) atomicAdd(&h[STEIN_HIST_BINS + digit], w);
        }
      }
    }
  }
}

template <int LEVEL, bool SYM>
__global__ __launch_bounds__(256) void k_hist(const float* __restrict__ D, long ldD, int n_local, int n,
                                              const SelState* st, u64* hist, const u32* __restrict__ skip) {
  if (skip && *skip) return;   // the speculative window already produced this step's median
  __shared__ u32 h[2 * STEIN_HIST_BINS];
  for (int b = threadIdx.x; b < 2 * STEIN_HIST_BINS; b += 256) h[b] = 0u;
  __syncthreads();
  const u32 pa = st->prefix[0], pb = st->prefix[1];
  const bool two = st->diverged != 0u;
  hist_pass_body<LEVEL, SYM>(D, ldD, n_local, n, h, pa, pb, two, blockIdx.x, gridDim.x);
  __syncthreads();
  for (int b = threadIdx.x; b < (two ? 2 : 1) * STEIN_HIST_BINS; b += 256)
    if (h[b]) atomicAdd(&hist[b], (u64)h[b]);
}

// The whole chained radix select of the fused symmetric call in ONE launch (round 3: for n <= 4096 only, three chained
// launches above that; round 4: every size -- when the window hit, the usual case, three launches returned at once, a few
// microseconds each).  The workgroups of the one launch meet behind each level.
//
// No workgroup ever waits for one that has not started (round 3's form assumed that the whole grid was resident: two
// processes on a card, or a stream with a CU mask, could leave the resident workgroups spinning for absent ones).  The work
// of a level is cut into nvb "virtual workgroups" (virtual workgroup v takes units v, v + nvb, ... of hist_pass_body's
// enumeration).  Real workgroup b takes virtual workgroup b -- after CLAIMING it (atomic exchange on HistSync::claim, an
// address of its own) -- flushes its LDS histogram into the global one (device-scope atomics, acknowledged: s_waitcnt
// vmcnt(0)) and reports it done: one add on its class's leaf counter, and the add that completes a class adds to the top
// counter; the add that completes the top makes its workgroup the level's resolver: it walks the global histogram once
// (chain_resolve, device-scope loads), publishes the select state and the new generation in all 64 class lines.  The
// others poll the line of their class (32 pollers per line; a line carries the state too).  A workgroup that has waited
// HIST_PATIENCE polls without seeing the level complete starts looking for UNCLAIMED virtual workgroups (their owners have not
// started: the chip is shared, the stream has a CU mask, or the launch is larger than the chip) and takes them over, one
// after the other, before it goes back to waiting.  So a waiting workgroup only ever waits for virtual workgroups that
// somebody running has claimed, and every level completes with one resident workgroup as well as with two thousand; a
// workgroup that starts late finds its own virtual workgroup taken and every level published, and falls through.
// The wait is bounded all the same (a hardware fault is the only way to exhaust it): FuseState::gave_up turns the
// step's bandwidth into NaN -- a wrong median is never returned -- and raises the device's error word in page-locked host
// memory, which the next call of the C ABI on this device reports as STEIN_E_HIP (stein_take_device_error).
constexpr int HIST_SPIN_MAX = 1 << 20;
constexpr int HIST_PATIENCE = 160;     // polls (~0.3 ms in all, hs_wait) before a waiting workgroup looks for abandoned work
constexpr int HIST_BLOCKS = 2048;      // workgroups of a histogram pass (C3, every step a miss: 0.61 ms of select with 2048, 0.77 with 1024, 1.17 with 512)
constexpr int HIST_ALL_SMALL_N = 4096; // up to here k_hist_all runs HIST_ALL_VBLOCKS virtual workgroups, above HIST_BLOCKS
constexpr int HIST_ALL_VBLOCKS = 512;  // ... = its grid, 2 per CU (C2, a miss: 135 us with 512, 165 with 256, 180 with 1024)
static_assert(sizeof(HistSync) <= (size_t)SPEC_TABLE * 8, "HistSync lives in the window table");
static_assert(HIST_BLOCKS <= HS_NV, "HistSync::claim holds one flag per virtual workgroup");

// thread 0: virtual workgroup v of `level` is done (its counts have reached the global histogram) -> is this the last one?
__device__ __forceinline__ bool hs_report_done(HistSync* hs, int level, u32 v, u32 nvb) {
  return tree_report_done(hs->leaf[level], &hs->top[level], v, nvb);
}
// whole workgroup: an unclaimed virtual workgroup of `level`, claimed for the caller; nvb if there is none.  Every thief
// scans from a start of its own (workgroup id x a stride coprime to any nvb <= 2048, + the number of its attempt): a
// thousand thieves that all took the FIRST unclaimed entry fought over one virtual workgroup per round (first form: 19 ms
// for a level with 256 absent owners).
__device__ __attribute__((noinline)) u32 hs_steal(HistSync* hs, int level, u32 nvb) {
  __shared__ u32 s_first, s_got;
  for (u32 attempt = 0;; ++attempt) {
    if (threadIdx.x == 0) s_first = nvb;
    __syncthreads();
    const u32 start = (blockIdx.x * 1021u + attempt * 97u) % nvb;
    u32 mine = nvb;   // position in scan order (0 = start) of this thread's first unclaimed entry
    for (u32 i = threadIdx.x; i < nvb; i += 256) {
      u32 v = start + i;
      if (v >= nvb) v -= nvb;
      if (load_fresh(&hs->claim[level][v]) == 0u) { mine = i; break; }
    }
    if (mine < nvb) atomicMin(&s_first, mine);
    __syncthreads();
    const u32 pos = s_first;
    if (pos >= nvb) return nvb;
    u32 cand = start + pos;
    if (cand >= nvb) cand -= nvb;
    if (threadIdx.x == 0)
      s_got = __hip_atomic_exchange(&hs->claim[level][cand], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u ? 1u : 0u;
    __syncthreads();
    const bool got = s_got != 0u;
    __syncthreads();   // (s_first / s_got are rewritten by the next round)
    if (got) return cand;
  }
}

// thread 0: poll a class line until it carries generation `want`; false when max_polls ran out.  Naps of 0.25 us at first,
// 2 us after the first few dozen polls (a poll is a device-scope load: it goes to the memory side every time).
__device__ __forceinline__ bool hs_wait(const u32* gen, u32 want, int max_polls) {
  int nap = 1;
  for (int spin = 0; spin < max_polls; ++spin) {
    if (load_fresh(gen) >= want) return true;
    for (int k = 0; k < nap; ++k) __builtin_amdgcn_s_sleep(8);
    if (spin >= 32 && nap < 8) nap += nap;
  }
  return false;
}

// (not inlined, one function per level: see resolve_all_body; the three passes in one function took 70 registers, 36 apart)
template <int LEVEL>
__device__ __attribute__((noinline)) void hist_pass_sym(const float* __restrict__ D, long ldD, int n, u32* h, u32 pa, u32 pb,
                                                        bool two, u32 v, u32 nvb) {
  hist_pass_body<LEVEL, true>(D, ldD, n, n, h, pa, pb, two, v, nvb);
}

__global__ __launch_bounds__(256) void k_hist_all(const float* __restrict__ D, long ldD, int n, const SelState* st, u64* hist_all,
                                                  const u32* __restrict__ hit, const u32* __restrict__ skip_l0, HistFinal fin,
                                                  FuseState* fs, HistSync* hs /* zero at launch */, u32 nvb,
                                                  u32* errword /* page-locked host memory, or NULL */) {
  if (*hit) return;   // the speculative window already produced this step's median
  __shared__ u32 h[2 * STEIN_HIST_BINS];
  __shared__ u32 s_v, s_flag;
  const int first = *skip_l0 == 0u ? 0 : 1;   // level 0 may have been taken by the distance kernel (an earlier launch)
  ChainState cs = chain_resolve<false>(hist_all, first, st);
  const HistSync::Line* myline = &hs->line[blockIdx.x % HS_CLASSES];
  for (int level = first; level < STEIN_HIST_LEVELS; ++level) {
    const u32 want = (u32)(level - first + 1);
    // this workgroup's own virtual workgroup, unless somebody has taken it over
    if (threadIdx.x == 0)
      s_v = blockIdx.x < nvb && __hip_atomic_exchange(&hs->claim[level][blockIdx.x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u
                ? blockIdx.x : nvb;
    bool last = false, thief = false;
    for (;;) {
      for (int b = threadIdx.x; b < 2 * STEIN_HIST_BINS; b += 256) h[b] = 0u;
      __syncthreads();
      const u32 v = s_v;
      if (v < nvb) {
        if (level == 0) hist_pass_sym<0>(D, ldD, n, h, cs.prefix[0], cs.prefix[1], cs.two, v, nvb);
        else if (level == 1) hist_pass_sym<1>(D, ldD, n, h, cs.prefix[0], cs.prefix[1], cs.two, v, nvb);
        else hist_pass_sym<2>(D, ldD, n, h, cs.prefix[0], cs.prefix[1], cs.two, v, nvb);
        __syncthreads();
        u64* hl = hist_all + (size_t)level * 2 * STEIN_HIST_BINS;
        for (int b = threadIdx.x; b < (cs.two ? 2 : 1) * STEIN_HIST_BINS; b += 256)
          if (h[b]) atomicAdd(&hl[b], (u64)h[b]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's histogram atomics have been acknowledged
        __syncthreads();
        if (threadIdx.x == 0) s_flag = hs_report_done(hs, level, v, nvb) ? 1u : 0u;
        __syncthreads();
        last = s_flag != 0u;
        if (last) break;
      }
      if (!thief) {   // wait for the level to be published -- for a while
        __syncthreads();
        if (threadIdx.x == 0) s_flag = hs_wait(&myline->gen, want, HIST_PATIENCE) ? 1u : 0u;
        __syncthreads();
        if (s_flag) break;
        thief = true;   // out of patience: somebody's virtual workgroup may have no owner
      }
      const u32 more = hs_steal(hs, level, nvb);
      if (more < nvb) {   // an abandoned virtual workgroup, now ours
        if (threadIdx.x == 0) s_v = more;
        __syncthreads();
        continue;
      }
      // nothing is abandoned (any more): whoever claimed the rest is running and will report it
      if (threadIdx.x == 0) {
        const bool ok = hs_wait(&myline->gen, want, HIST_SPIN_MAX);
        if (!ok) __hip_atomic_store(&fs->gave_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      break;
    }
    const bool final_level = level + 1 == STEIN_HIST_LEVELS;
    if (last) {   // the level's resolver
      if (final_level) {
        resolve_all_body(hist_all, fin.st, fin.sp, fin.ln_n, fin.h2_out);
        if (threadIdx.x == 0 && load_fresh(&fs->gave_up)) {   // a wait ran out somewhere: no median, and loudly so
          fin.st->median = fin.st->h2 = __builtin_nanf("");
          if (fin.h2_out) *fin.h2_out = __builtin_nanf("");
          if (errword) __hip_atomic_store(errword, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      } else {
        cs = chain_resolve<true>(hist_all, level + 1, st);
      }
      if (threadIdx.x < HS_CLASSES) {   // one thread per class line: the state first, then (acknowledged) the generation
        HistSync::Line* ln = &hs->line[threadIdx.x];
        const u32 w[6] = {cs.prefix[0], cs.prefix[1], (u32)cs.rank[0], (u32)(cs.rank[0] >> 32), (u32)cs.rank[1], (u32)(cs.rank[1] >> 32)};
#pragma unroll
        for (int k = 0; k < 6; ++k) __hip_atomic_store(&ln->pub[k], w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&ln->gen, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      continue;   // (behind the final level the loop ends)
    }
    if (final_level) return;   // the select is complete (or, gave_up, declared failed); nobody needs the state any more
    // (a workgroup that started late may read the state of a LATER level here, or a mix of two: then that later level was
    // complete before the read, all of its virtual workgroups are claimed, and the state is never used)
    u32 w[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = load_fresh(&myline->pub[k]);
    cs.prefix[0] = w[0]; cs.prefix[1] = w[1];
    cs.rank[0] = (u64)w[2] | ((u64)w[3] << 32); cs.rank[1] = (u64)w[4] | ((u64)w[5] << 32);
    cs.two = w[0] != w[1];
  }
}

// one wave; hist points at this level's [2][STEIN_HIST_BINS] counters (already summed over ranks)
__global__ __launch_bounds__(64) void k_resolve(const u64* __restrict__ hist, int level, SelState* st, float ln_n,
                                                float* h2_out, float* median_out, const u32* __restrict__ skip) {
  if (skip && *skip) return;
  __shared__ u64 bins[STEIN_HIST_BINS];
  __shared__ u64 chunk[64];
  const int lane = threadIdx.x;
  const int bits = level == 2 ? 10 : 11;
  const bool div_in = st->diverged != 0u;
  u32 newp[2];
  u64 newr[2];
  for (int tg = 0; tg < 2; ++tg) {
    const u64* src = hist + ((div_in && tg == 1) ? STEIN_HIST_BINS : 0);
    u64 s = 0;
    for (int b = 0; b < 32; ++b) {
      const u64 c = src[lane * 32 + b];
      bins[lane * 32 + b] = c;
      s += c;
    }
    chunk[lane] = s;
    __syncthreads();
    if (lane == 0) {
      u64 rank = st->rank[tg], cum = 0;
      int c = 0;
      while (c < 63 && cum + chunk[c] <= rank) cum += chunk[c++];
      int b = c * 32;
      const int bend = b + 31;
      while (b < bend && cum + bins[b] <= rank) cum += bins[b++];
      newp[tg] = (st->prefix[tg] << bits) | (u32)b;
      newr[tg] = rank - cum;
    }
    __syncthreads();
  }
  if (lane == 0) {
    st->prefix[0] = newp[0]; st->prefix[1] = newp[1];
    st->rank[0] = newr[0]; st->rank[1] = newr[1];
    st->diverged = (newp[0] != newp[1]) ? 1u : 0u;
    if (level == 2) {
      const float lo = key_f32(newp[0]), hi = key_f32(newp[1]);
      const float med = st->even ? 0.5f * (lo + hi) : lo;
      const float bw = sqrtf(med / ln_n);      // abstract_kernel.py:40
      const float h2 = bw * bw;                // squared_exponential_kernel.py:22 squares it again
      st->lo = lo; st->hi = hi; st->median = med; st->h2 = h2;
      if (h2_out) *h2_out = h2;
      if (median_out) *median_out = med;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// speculative median window (SpecState in stein_common.h): begin / select / update, one launch each per step
__global__ __launch_bounds__(256) void k_median_init(SelState* st, SpecState* sp, u64 total, u64* __restrict__ hist,
                                                     u64* __restrict__ slots) {
  median_init_body(blockIdx.x * 256 + threadIdx.x, gridDim.x * 256, st, sp, total, hist, slots);
}

// First kernel of the fused call (fp32 inputs; bf16 inputs: the same work rides in k_split's launch, stein_x3.hip): the row
// norms and everything the later kernels expect to find zeroed or set up (prologue_body, stein_common.h).
template <typename TIN>
__global__ __launch_bounds__(256) void k_prologue(const TIN* __restrict__ T, PrologueArgs a) {
  prologue_body<TIN>(T, a, (int)blockIdx.x, (int)gridDim.x);
}

__device__ __attribute__((noinline)) void spec_update_dev(const SelState* st, SpecState* sp);   // below

// All 256 bins of an LDS histogram -> the bin holding 0-based rank `rank` and the rank inside it; *bin = 256 when the
// rank lies past the last bin.  Called by the whole workgroup (>= 256 threads); `scan` is 256 words of LDS scratch.
// Both targets at once: waves 0-3 scan histogram hA (256 bins) for rankA, waves 4-7 histogram hB for rankB, each with
// shuffles; `scan[0..7]` carries the wave totals (2 barriers in all; round 4: four separate locates cost 12 of them, ~1.4 k
// cycles each time, a quarter of this one-workgroup kernel).  bin[k] = 256: rank k lies beyond its histogram.
__device__ __forceinline__ void spec_locate2(const u32* hA, u32 rankA, const u32* hB, u32 rankB, u32* scan, u32* bin, u32* rest) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int side = t >> 8;                       // 0: target A (threads 0..255), 1: target B (256..511)
  u32 c = 0u, incl = 0u;
  if (t < 512) {
    c = (side ? hB : hA)[t & 255];
    incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const u32 v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) scan[wave] = incl;
  }
  if (t < 2) bin[t] = 256u;
  __syncthreads();
  if (t < 512) {
    u32 base = 0u;
    for (int w = side * 4; w < wave; ++w) base += scan[w];
    const u32 excl = base + incl - c, rank = side ? rankB : rankA;
    if (excl <= rank && rank < excl + c) { bin[side] = (u32)(t & 255); rest[side] = rank - excl; }
  }
  __syncthreads();
}

// One workgroup: exact weighted selection of the two median targets among the buffered window entries.
// Entry = key << 2 | weight, offset o = key - lo_key < 65536: pass 1 histograms o >> 8, pass 2 the low byte of the
// entries that share each target's high byte.  Returns (workgroup-uniform) whether the window held both targets.
#ifdef STEIN_SEL_STAMPS
__device__ unsigned long long g_sel_stamps[16];
extern "C" int stein_debug_sel_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sel_stamps), sizeof(g_sel_stamps)) == hipSuccess ? 0 : -1; }
#define SEL_STAMP(k) do { if (threadIdx.x == 0) { g_sel_stamps[k] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); } } while (0)
#else
#define SEL_STAMP(k) do {} while (0)
#endif
__device__ __forceinline__ bool spec_select_body(SelState* st, SpecState* sp, const u64* __restrict__ slots, float ln_n,
                                                 float* h2_out, int update) {
  const u64* __restrict__ buf = slots + SPEC_SLOTS * 8;
  __shared__ u32 h1[256], h2a[256], h2b[256], scan[256];
  __shared__ u32 sel[8];   // [0,1] high bytes, [2,3] ranks inside them, [4,5] low bytes, [6,7] scratch
  __shared__ u64 below_s;
  const int t = threadIdx.x;
  SEL_STAMP(0);
  const u32 cnt = sp->count, lo = sp->lo_key, width = sp->width;
  if (width == 0u || sp->overflow || cnt > SPEC_CAP || cnt == 0u) return false;   // miss: the radix select runs
  SEL_STAMP(1);
  // thread 0 asks now for what it will need at the very end (the result and the predictor update are a chain of dependent
  // loads otherwise: ~3 k cycles behind the last barrier)
  u32 even0 = 0u;
  if (t == 0) even0 = st->even;
  u64 mine = t < (int)SPEC_SLOTS ? slots[t * 8] : 0ull;
  const u64 total = sp->total;
  if (t == 0) below_s = 0ull;
  if (t < 256) { h1[t] = 0u; h2a[t] = 0u; h2b[t] = 0u; }
  __syncthreads();
  {
    // the weights below the window, one slot per thread: summed per wave first (256 same-address 64-bit LDS atomics took
    // 7 k cycles of this kernel's 28 k at C2)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((t & 63) == 0 && mine) atomicAdd(reinterpret_cast<unsigned long long*>(&below_s), (unsigned long long)mine);
  }
  __syncthreads();
  const u64 below = below_s;
  SEL_STAMP(2);
  const u64 r0 = (total & 1ull) ? total / 2 : total / 2 - 1, r1 = total / 2;
  if (r0 < below || r1 - below > 0xfffffff0ull) return false;   // the target lies below the window
  // Pass 1 histograms the HIGH byte of the offsets: a window of `width` keys occupies (width >> 8) + 1 bins, a handful, and
  // every entry of every wave lands in them.  Up to 16 entries per thread are fetched ONCE, all loads in flight together,
  // and both passes work from the registers (round 4: the loops below paid one memory latency per 1024 entries, twice).
  // With at most eight bins in play the counts are kept in eight registers per thread, summed over the wave with shuffles
  // and added with one atomic per wave and bin (LDS atomics merged by ballots, below, were 1 k cycles per 1024 entries:
  // 7 k of this kernel's 28 k cycles at C2, 15 k of 39 k at C3).
  // (the 8-byte entries come through ONE compute unit: 13 k of them are 108 KB, ~3 k cycles of its load path -- requested
  // any earlier they only delay the slot sums above, which wait behind them in the memory pipeline)
  constexpr int EPT = 16;
  const bool inreg = cnt <= 1024u * EPT;
  const u32 nb = (width >> 8) + 1u;
  u64 er[EPT];
  if (inreg) {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const u32 i = (u32)k * 1024u + (u32)t;
      er[k] = (u32)k * 1024u < cnt && i < cnt ? buf[i] : 0ull;   // (0: weight 0, counted nowhere)
    }
  }
  if (inreg && nb <= 8u) {
    u32 c[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if ((u32)k * 1024u < cnt) {                // workgroup-uniform
        const u64 e = er[k];
        const u32 hb = (((u32)(e >> 2) - lo) >> 8) & 255u, w = (u32)e & 3u;
#pragma unroll
        for (int b = 0; b < 8; ++b) c[b] += hb == (u32)b ? w : 0u;
      }
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      if ((u32)b < nb) {                         // workgroup-uniform
        u32 v = c[b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((t & 63) == 0 && v) atomicAdd(&h1[b], v);
      }
    }
  } else if (inreg) {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if ((u32)k * 1024u < cnt) {                // workgroup-uniform: the ballots need every lane of a wave
        const u64 e = er[k];
        const u32 hb = (((u32)(e >> 2) - lo) >> 8) & 255u, w = (u32)e & 3u;
        hist_add(h1, hb, w == 2u, t & 63, 2u);   // (lanes that share the leader's bin are merged into one atomic per wave)
        hist_add(h1, hb, w == 1u, t & 63, 1u);
        if (w == 3u) atomicAdd(&h1[hb], 3u);     // (no producer writes weight 3; kept exact all the same)
      }
    }
  } else {
    for (u32 i0 = 0; i0 < cnt; i0 += 1024) {   // wave-uniform trip count: the ballots need every lane
      const u32 i = i0 + (u32)t;
      const bool ok = i < cnt;
      const u64 e = ok ? buf[i] : 0ull;
      const u32 hb = (((u32)(e >> 2) - lo) >> 8) & 255u, w = (u32)e & 3u;
      hist_add(h1, hb, ok && w == 2u, t & 63, 2u);
      hist_add(h1, hb, ok && w == 1u, t & 63, 1u);
      if (ok && w == 3u) atomicAdd(&h1[hb], 3u);
    }
  }
  __syncthreads();
  SEL_STAMP(3);
  spec_locate2(h1, (u32)(r0 - below), h1, (u32)(r1 - below), scan, &sel[0], &sel[2]);
  SEL_STAMP(4);
  const u32 ba = sel[0], bb = sel[1];
  if (ba == 256u || bb == 256u) return false;   // a target lies above the window
  const bool two_hb = ba != bb;                 // (both targets in one high byte, the usual case: one low-byte histogram serves both)
  if (inreg) {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if ((u32)k * 1024u < cnt) {                // workgroup-uniform
        const u64 e = er[k];
        const u32 o = (u32)(e >> 2) - lo, w = (u32)e & 3u;
        if (w && (o >> 8) == ba) atomicAdd(&h2a[o & 255u], w);
        if (w && two_hb && (o >> 8) == bb) atomicAdd(&h2b[o & 255u], w);
      }
    }
  } else {
    for (u32 i = t; i < cnt; i += 1024) {
      const u64 e = buf[i];
      const u32 o = (u32)(e >> 2) - lo, w = (u32)e & 3u;
      if ((o >> 8) == ba) atomicAdd(&h2a[o & 255u], w);
      if (two_hb && (o >> 8) == bb) atomicAdd(&h2b[o & 255u], w);
    }
  }
  __syncthreads();
  SEL_STAMP(5);
  spec_locate2(h2a, sel[2], two_hb ? h2b : h2a, sel[3], scan, &sel[4], &sel[6]);
  SEL_STAMP(6);
  if (t == 0) {
    const float flo = key_f32(lo + ((ba << 8) | sel[4])), fhi = key_f32(lo + ((bb << 8) | sel[5]));
    const float med = even0 ? 0.5f * (flo + fhi) : flo;
    const float bw = sqrtf(med / ln_n);      // abstract_kernel.py:40
    const float h2 = bw * bw;                // squared_exponential_kernel.py:22 squares it again
    st->lo = flo; st->hi = fhi; st->median = med; st->h2 = h2;
    if (h2_out) *h2_out = h2;
    sp->hit = 1u;
    sp->skip_l0 = 1u;
    if (update) spec_update_dev(st, sp);   // fused call: no separate k_spec_update launch
    SEL_STAMP(7);
  }
  return true;
}

// Small symmetric blocks (fused call, n <= SOLO_MAX_N): when the window misses, this one workgroup runs the whole
// 3-level radix select over the upper triangle of D itself (LDS histograms, digits located by a prefix sum over the
// 2048 bins), so the fused call launches no histogram passes at all -- three launches that, on the usual hit, did
// nothing for 4-5 us each.  A miss costs ~30 us here instead of ~15 us; misses are the first two steps and jumps.
constexpr int SOLO_MAX_N = 512;
__device__ __forceinline__ void solo_locate(const u32* h, u32 rank, u32* wsum, u32* out_bin, u32* out_rest) {
  // 1024 threads, two bins each; *out_bin / *out_rest are LDS words written by the one thread that finds the rank
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const u32 c0 = h[2 * t], c1 = h[2 * t + 1];
  u32 incl = c0 + c1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const u32 v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  u32 base = 0u;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  const u32 excl = base + incl - (c0 + c1);
  if (rank >= excl && rank < excl + c0 + c1) {
    const u32 b = rank < excl + c0 ? 2u * t : 2u * t + 1u;
    *out_bin = b;
    *out_rest = rank - ((b & 1u) ? excl + c0 : excl);
  }
  __syncthreads();
}
__device__ __forceinline__ void solo_select(const float* __restrict__ D, long ldD, int n, SelState* st, SpecState* sp,
                                            float ln_n, float* h2_out) {
  __shared__ u32 sh[2 * STEIN_HIST_BINS];
  __shared__ u32 s_wsum[16], s_bin[2], s_rest[2];
  const int t = threadIdx.x;
  const long ntc = ldD >> 5;
  const int ntr = (n + DT_ROWS - 1) / DT_ROWS, nct = (n + DT_COLS - 1) / DT_COLS;
  const u32 total = (u32)n * (u32)n;
  u32 prefix[2] = {0u, 0u};
  u32 rank[2] = {(total & 1u) ? total / 2 : total / 2 - 1, total / 2};
  bool two = false;
  for (int level = 0; level < STEIN_HIST_LEVELS; ++level) {
    const int bits = level == 2 ? 10 : 11, shift = level == 0 ? 21 : (level == 1 ? 10 : 0);
    for (int b = t; b < 2 * STEIN_HIST_BINS; b += 1024) sh[b] = 0u;
    __syncthreads();
    for (int ti = 0; ti < ntr; ++ti)
      for (int tj = 0; tj < nct; ++tj) {
        if (tj * DT_COLS + DT_COLS <= ti * DT_ROWS) continue;   // wholly below the diagonal
        const float4 v4 = *reinterpret_cast<const float4*>(D + ((size_t)ti * ntc + tj) * DT_ELEMS + t * 4);
        const int row = ti * DT_ROWS + (t >> 3), c0 = tj * DT_COLS + (t & 7) * 4;
        const float x[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int col = c0 + e;
          if (row < n && col < n && col >= row) {
            const u32 key = f32_key(x[e]), w = col != row ? 2u : 1u;
            const u32 digit = (key >> shift) & ((1u << bits) - 1u);
            const u32 hi = level == 0 ? 0u : key >> (shift + bits);
            if (level == 0 || hi == prefix[0]) atomicAdd(&sh[digit], w);
            if (two && hi == prefix[1]) atomicAdd(&sh[STEIN_HIST_BINS + digit], w);
          }
        }
      }
    __syncthreads();
    solo_locate(sh, rank[0], s_wsum, &s_bin[0], &s_rest[0]);
    solo_locate(sh + (two ? STEIN_HIST_BINS : 0), rank[1], s_wsum, &s_bin[1], &s_rest[1]);
    prefix[0] = (prefix[0] << bits) | s_bin[0];
    prefix[1] = (prefix[1] << bits) | s_bin[1];
    rank[0] = s_rest[0];
    rank[1] = s_rest[1];
    two = prefix[0] != prefix[1];
    __syncthreads();   // s_bin / s_rest are rewritten by the next level
  }
  if (t == 0) {
    st->prefix[0] = prefix[0]; st->prefix[1] = prefix[1];
    st->rank[0] = rank[0]; st->rank[1] = rank[1];
    st->diverged = two ? 1u : 0u;
    const float lo = key_f32(prefix[0]), hi = key_f32(prefix[1]);
    const float med = st->even ? 0.5f * (lo + hi) : lo;
    const float bw = sqrtf(med / ln_n);      // abstract_kernel.py:40
    const float h2 = bw * bw;                // squared_exponential_kernel.py:22 squares it again
    st->lo = lo; st->hi = hi; st->median = med; st->h2 = h2;
    if (h2_out) *h2_out = h2;
    spec_update_dev(st, sp);
  }
}

// D != NULL ("solo", fused call on a small symmetric block): a miss is resolved here by solo_select
__global__ __launch_bounds__(1024) void k_spec_select(SelState* st, SpecState* sp, const u64* __restrict__ slots,
                                                      float ln_n, float* h2_out, int update,
                                                      const float* __restrict__ D, long ldD, int n) {
  const bool hit = spec_select_body(st, sp, slots, ln_n, h2_out, update);
  if (!hit && D) {
    __syncthreads();
    solo_select(D, ldD, n, st, sp, ln_n, h2_out);
  }
}

// ---- the window across several ranks: tally -> all-reduce(sum) -> pick ---------------------------------------------
// this rank's window entries -> one counter per key (table[SPEC_TABLE_HDR + key - lo_key]); workgroup 0 also fills the header
__global__ __launch_bounds__(256) void k_spec_tally(const SpecState* __restrict__ sp, const u64* __restrict__ slots,
                                                    u64* __restrict__ table) {
  const u64* __restrict__ buf = slots + SPEC_SLOTS * 8;
  const u32 cnt = sp->count, lo = sp->lo_key;
  const bool bad = sp->width == 0u || sp->overflow || cnt > SPEC_CAP;
  if (blockIdx.x == 0) {
    if (threadIdx.x < (int)SPEC_SLOTS && slots[threadIdx.x * 8])
      atomicAdd(reinterpret_cast<unsigned long long*>(&table[0]), (unsigned long long)slots[threadIdx.x * 8]);
    if (threadIdx.x == 0) {
      if (bad) atomicAdd(reinterpret_cast<unsigned long long*>(&table[1]), 1ull);
      atomicAdd(reinterpret_cast<unsigned long long*>(&table[2]), (unsigned long long)cnt);
    }
  }
  if (bad) return;
  for (u32 i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256) {
    const u64 e = buf[i];
    atomicAdd(reinterpret_cast<unsigned long long*>(&table[SPEC_TABLE_HDR + ((u32)(e >> 2) - lo)]),
              (unsigned long long)(e & 3ull));
  }
}

// one workgroup: the two median targets from the rank-summed table (identical on every rank)
__global__ __launch_bounds__(1024) void k_spec_pick(SelState* st, SpecState* sp, const u64* __restrict__ table,
                                                    float ln_n, float* h2_out, float* median_out) {
  __shared__ u64 part[1024];
  __shared__ u32 found[2];
  const int t = threadIdx.x;
  const u32 width = sp->width, lo = sp->lo_key;
  // k_spec_update sizes the next window from `count`: make it the GLOBAL number of entries so that every rank's
  // predictor stays identical (the local counts differ from rank to rank)
  if (t == 0) sp->count = table[2] > 0xffffffffull ? 0xffffffffu : (u32)table[2];
  if (width == 0u || table[1] != 0ull) return;   // miss on every rank alike
  const u64 total = sp->total, below = table[0];
  const u64 r0 = (total & 1ull) ? total / 2 : total / 2 - 1, r1 = total / 2;
  if (r0 < below) return;
  constexpr int PER = 64;   // 1024 threads x 64 keys >= 65536
  u64 mine = 0ull;
  for (int k = 0; k < PER; ++k) {
    const u32 key = (u32)t * PER + k;
    if (key <= width) mine += table[SPEC_TABLE_HDR + key];
  }
  part[t] = mine;
  if (t < 2) found[t] = 0xffffffffu;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {   // inclusive scan
    u64 v = 0ull;
    if (t >= o) v = part[t - o];
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  const u64 excl = part[t] - mine;
  for (int tg = 0; tg < 2; ++tg) {
    const u64 rank = (tg ? r1 : r0) - below;
    if (excl <= rank && rank < excl + mine) {
      u64 cum = excl;
      for (int k = 0; k < PER; ++k) {
        const u64 c = table[SPEC_TABLE_HDR + (u32)t * PER + k];
        if (rank < cum + c) { found[tg] = (u32)t * PER + k; break; }
        cum += c;
      }
    }
  }
  __syncthreads();
  if (t == 0 && found[0] != 0xffffffffu && found[1] != 0xffffffffu) {
    const float flo = key_f32(lo + found[0]), fhi = key_f32(lo + found[1]);
    const float med = st->even ? 0.5f * (flo + fhi) : flo;
    const float bw = sqrtf(med / ln_n);      // abstract_kernel.py:40
    const float h2 = bw * bw;                // squared_exponential_kernel.py:22 squares it again
    st->lo = flo; st->hi = fhi; st->median = med; st->h2 = h2;
    if (h2_out) *h2_out = h2;
    if (median_out) *median_out = med;
    sp->hit = 1u;
    sp->skip_l0 = 1u;
  }
}

// after the median is final (window or radix passes): predict the next one and size its window (one thread)
__device__ __attribute__((noinline)) void spec_update_dev(const SelState* st, SpecState* sp) {
  const u32 key = f32_key(st->lo);
  const bool had_window = sp->width != 0u;
  u32 hw = 4096u, next = key, earned = 0u;
  if (sp->magic == SPEC_MAGIC1 || sp->magic == SPEC_MAGIC2) {
    // linear extrapolation of the VALUE (key space bends at every power of two)
    const float pred = 2.f * st->lo - key_f32(sp->last_key);
    long c = (long)f32_key(pred == pred ? pred : st->lo);
    c = c < 65536l ? 65536l : (c > 0xfffe0000l ? 0xfffe0000l : c);
    next = (u32)c;
    if (sp->magic == SPEC_MAGIC2 && had_window) {   // the window of this step was centred on a real prediction
      const u32 err = key > sp->center ? key - sp->center : sp->center - key;
      hw = err > SPEC_HW_MAX / 4u ? SPEC_HW_MAX : 4u * err + 48u;
      // ... and never below three quarters of the previous one: one lucky prediction (an error of a few keys) used to shrink
      // the window to ~60 keys and the next ordinary error missed it -- 10 % misses under noisy scores and at n = 4096 in
      // bf16; with the floor 1 %, for a window a third wider on average (scratch/window_policy.py replays the rules on
      // recorded runs: profiles/r04_window_policy.txt).  A miss costs a radix select, a wider window a few more entries.
      // (the floor follows EARNED widths only: the 4096-key window of a predictor without a velocity is not one)
      const u32 floor_hw = sp->earned_hw - sp->earned_hw / 4u;
      if (hw < floor_hw) hw = floor_hw;
      earned = hw;
      if (sp->hit && sp->count > SPEC_CAP / 2 && hw > sp->halfwidth / 2u) hw = sp->halfwidth / 2u + 1u;   // keep the buffer small
    }
    sp->magic = SPEC_MAGIC2;
    sp->n_steps += 1u;
    sp->n_hits += sp->hit ? 1u : 0u;
  } else {
    sp->magic = SPEC_MAGIC1;
    sp->n_steps = 1u;
    sp->n_hits = 0u;
  }
  sp->last_key = key;
  sp->center = next;
  sp->halfwidth = hw > SPEC_HW_MAX ? SPEC_HW_MAX : hw;
  sp->earned_hw = earned > SPEC_HW_MAX ? SPEC_HW_MAX : earned;
}
__global__ void k_spec_update(const SelState* st, SpecState* sp) {
  if (threadIdx.x || blockIdx.x) return;
  spec_update_dev(st, sp);
}

// end of the chained radix select (fused call): all three resolves, the median / bandwidth, the predictor update;
// one workgroup of 256 threads
// (not inlined, like chain_resolve and hs_steal: inlined into k_hist_all their constants and addresses were hoisted in front
// of the level loop -- 114 registers, four workgroups per CU instead of eight, for code that one workgroup runs once)
__device__ __attribute__((noinline)) void resolve_all_body(u64* hist_all, SelState* st, SpecState* sp, float ln_n, float* h2_out) {
  const ChainState cs = chain_resolve<true>(hist_all, STEIN_HIST_LEVELS, st);
  if (threadIdx.x == 0) {
    st->prefix[0] = cs.prefix[0]; st->prefix[1] = cs.prefix[1];
    st->rank[0] = cs.rank[0]; st->rank[1] = cs.rank[1];
    st->diverged = cs.two ? 1u : 0u;
    const float lo = key_f32(cs.prefix[0]), hi = key_f32(cs.prefix[1]);
    const float med = st->even ? 0.5f * (lo + hi) : lo;
    const float bw = sqrtf(med / ln_n);      // abstract_kernel.py:40
    const float h2 = bw * bw;                // squared_exponential_kernel.py:22 squares it again
    st->lo = lo; st->hi = hi; st->median = med; st->h2 = h2;
    if (h2_out) *h2_out = h2;
    spec_update_dev(st, sp);
  }
}

// ------------------------------------------------------------------------------------------------
// k_kernel_matrix: optional K output, K = exp(-D / h2 / 2)
// ------------------------------------------------------------------------------------------------
// upper: D holds only the 128 x 128 tiles on and above the diagonal (the split path's symmetric distance pass); an entry
// of a tile below it is read from its mirror image
__global__ __launch_bounds__(256) void k_kernel_matrix(const float* __restrict__ D, long ldD, int n_local, int n,
                                                       const float* __restrict__ h2p, float* __restrict__ K, long ldK,
                                                       int upper) {
  const float h2 = *h2p;
  const long total = (long)n_local * n;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const long row = e / n, col = e - row * n;
    const bool swap = upper && (col >> 7) < (row >> 7);
    K[row * ldK + col] = expf(-D[d_index(swap ? col : row, swap ? row : col, ldD >> 5)] / h2 / 2.f);
  }
}

// ------------------------------------------------------------------------------------------------
// k_phi_partial: O[z] = P[:, jrange(z)] . V[jrange(z), cblock],  P = exp2(c D) built tile by tile.
//   A operand = P tile [128 rows][32 j], written to LDS as [row][j] right after the exp, read back with
//   ds_read_b128 (same k permutation as k_distance).  B operand = V tile [32 j][128 c], row-major in LDS;
//   a lane reads V[j = 8kk + 4h + s][c = lane & 31] with ds_read_b32 (32 consecutive floats per half wave).
//   The thread that stages P(row, 4 j) keeps the running rowsum for that row.
// ------------------------------------------------------------------------------------------------
template <bool VEC>
__device__ __forceinline__ void load_v_tile(const float* __restrict__ V, int n, int d, int j0, int jend, int c0,
                                            int vr, int vc, float4 (&v)[4]) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int j = j0 + vr + 8 * p;
    const int c = c0 + vc;
    const bool jok = j < jend;
    if (VEC) {
      v[p] = ld4_or_zero(V + (size_t)j * d + c, jok && c < d);
    } else {
      const float* src = V + (size_t)j * d + c;
      v[p].x = (jok && c + 0 < d) ? src[0] : 0.f;
      v[p].y = (jok && c + 1 < d) ? src[1] : 0.f;
      v[p].z = (jok && c + 2 < d) ? src[2] : 0.f;
      v[p].w = (jok && c + 3 < d) ? src[3] : 0.f;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(NTHREADS) void k_phi_partial(const float* __restrict__ D, long ldD,
                                                          const float* __restrict__ G, const float* __restrict__ T,
                                                          const float* __restrict__ h2p, float* __restrict__ OG,
                                                          float* __restrict__ OT, float* __restrict__ RS, int n, int d,
                                                          int n_local, int tiles_m, int cblocks, int split, int jchunk) {
  __shared__ __attribute__((aligned(16))) float smem[BM * LDK + BK * BN];
  float* As = smem;
  float* Bs = smem + BM * LDK;

  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ncb = 2 * cblocks;
  const int cb = logical % ncb;
  const int tile_m = (logical / ncb) % tiles_m;
  const int z = logical / (ncb * tiles_m);
  const bool isT = cb >= cblocks;
  const float* __restrict__ V = isT ? T : G;
  float* __restrict__ O = isT ? OT : OG;
  const int c0 = (isT ? cb - cblocks : cb) * BN;

  const int jbeg = z * jchunk;
  const int jend = min(n, jbeg + jchunk);

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wy = wid >> 1, wx = wid & 1;
  const int lr = t >> 3, lc = (t & 7) * 4;   // P staging: rows lr + 32p, 4 consecutive j
  const int vr = t >> 5, vc = (t & 31) * 4;  // V staging: j rows vr + 8p, 4 consecutive c
  const int i0 = tile_m * BM;

  const float cexp = -1.44269504088896341f / (2.f * *h2p);  // exp(-D/(2 h2)) = exp2(cexp * D)

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  float rs[4] = {0.f, 0.f, 0.f, 0.f};

  float4 rd[4], rv[4];
  // the D tile (tile_m, j0 / 32) is one contiguous [128][32] block (rows are padded to 128 in memory)
  const float* __restrict__ drow = D + (size_t)tile_m * (ldD >> 5) * DT_ELEMS;
  auto load_d = [&](int j0) {
    const float* tile = drow + (size_t)(j0 >> 5) * DT_ELEMS;
#pragma unroll
    for (int p = 0; p < 4; ++p) rd[p] = *reinterpret_cast<const float4*>(tile + (lr + 32 * p) * DT_COLS + lc);
  };
  if (jbeg < jend) {
    load_d(jbeg);
    load_v_tile<VEC>(V, n, d, jbeg, jend, c0, vr, vc, rv);
  }

  const int l31 = lane & 31, h4 = (lane >> 5) * 4;
  for (int j0 = jbeg; j0 < jend; j0 += BK) {
    // P = exp2(cexp * D) for in-range j, 0 outside; stage to LDS, keep the rowsum
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int j = j0 + lc;
      float4 pv;
      pv.x = (j + 0 < jend) ? __builtin_amdgcn_exp2f(cexp * rd[p].x) : 0.f;
      pv.y = (j + 1 < jend) ? __builtin_amdgcn_exp2f(cexp * rd[p].y) : 0.f;
      pv.z = (j + 2 < jend) ? __builtin_amdgcn_exp2f(cexp * rd[p].z) : 0.f;
      pv.w = (j + 3 < jend) ? __builtin_amdgcn_exp2f(cexp * rd[p].w) : 0.f;
      if (i0 + lr + 32 * p >= n_local) pv = make_float4(0.f, 0.f, 0.f, 0.f);
      rs[p] += (pv.x + pv.y) + (pv.z + pv.w);
      *reinterpret_cast<float4*>(As + (lr + 32 * p) * LDK + lc) = pv;
      *reinterpret_cast<float4*>(Bs + (vr + 8 * p) * BN + vc) = rv[p];
    }
    __syncthreads();
    if (j0 + BK < jend) {
      load_d(j0 + BK);
      load_v_tile<VEC>(V, n, d, j0 + BK, jend, c0, vr, vc, rv);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 a[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        a[i] = *reinterpret_cast<const float4*>(As + (wy * 64 + i * 32 + l31) * LDK + kk * 8 + h4);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float b[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = Bs[(kk * 8 + h4 + s) * BN + wx * 64 + j * 32 + l31];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(comp(a[i], s), b[j], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  phi_epilogue(acc, rs, O + (size_t)z * n_local * d, RS + (size_t)z * n_local, d, n_local, i0, c0, cb == 0);
}

// ------------------------------------------------------------------------------------------------
// k_phi_finish: sum the split partials, form phi, per-block partial |phi|^2 in fp64
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 theta4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 theta4(const unsigned short* p) {   // four bf16 values (8-byte aligned)
  const uint2 w = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                     __uint_as_float(w.y & 0xffff0000u));
}
template <typename TIN>
__global__ __launch_bounds__(256) void k_phi_finish(const float* __restrict__ OG, const float* __restrict__ OT,
                                                    const float* __restrict__ RS, const TIN* __restrict__ T,
                                                    const float* __restrict__ h2p, float* __restrict__ phi,
                                                    float* __restrict__ dK, double* __restrict__ sqpart, int n, int d,
                                                    int row0, int n_local, int split, int vec, HistSync* done,
                                                    double* __restrict__ sq_out) {
  // done != NULL (fused call; its completion counters are zero at launch): the last workgroup out also sums the partials --
  // in the order k_sum_partials uses, so the result is the same to the last bit -- which saves that launch
  __shared__ double red[4];
  const float h2 = *h2p;
  const float fn = (float)n;
  const long total = (long)n_local * d;
  const size_t zs = (size_t)n_local * d;
  double sq = 0.0;
  if (vec) {   // host: d % 4 == 0 and every pointer aligned for four columns at a time
    // four consecutive columns of one row per step, 16-byte loads and stores (one entry at a time with an integer
    // division each, the kernel moved its 68 MB at 2.8 TB/s; bf16 inputs took that path until round 4: 19 us at C2)
    const long total4 = total >> 2;
    const int d4 = d >> 2;
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < total4; q += (long)gridDim.x * 256) {
      const int i = (int)(q / d4);
      const long e = q << 2;
      float4 og = make_float4(0.f, 0.f, 0.f, 0.f), ot = og;
      float rs = 0.f;
#pragma unroll 8
      for (int z = 0; z < split; ++z) {   // unrolled: the loads of eight slices in flight, the sums in the same order
        const float4 a = *reinterpret_cast<const float4*>(OG + z * zs + e), b = *reinterpret_cast<const float4*>(OT + z * zs + e);
        og.x += a.x; og.y += a.y; og.z += a.z; og.w += a.w;
        ot.x += b.x; ot.y += b.y; ot.z += b.z; ot.w += b.w;
        rs += RS[(size_t)z * n_local + i];
      }
      const float4 th = theta4(T + (size_t)row0 * d + e);
      float4 dk, ph;
      dk.x = (rs * th.x - ot.x) / h2; dk.y = (rs * th.y - ot.y) / h2; dk.z = (rs * th.z - ot.z) / h2; dk.w = (rs * th.w - ot.w) / h2;
      ph.x = (og.x + dk.x) / fn; ph.y = (og.y + dk.y) / fn; ph.z = (og.z + dk.z) / fn; ph.w = (og.w + dk.w) / fn;
      *reinterpret_cast<float4*>(phi + e) = ph;
      if (dK) *reinterpret_cast<float4*>(dK + e) = dk;
      sq += ((double)ph.x * (double)ph.x + (double)ph.y * (double)ph.y) + ((double)ph.z * (double)ph.z + (double)ph.w * (double)ph.w);
    }
  } else {
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
      const int i = (int)(e / d);
      float og = 0.f, ot = 0.f, rs = 0.f;
      for (int z = 0; z < split; ++z) {
        og += OG[z * zs + e];
        ot += OT[z * zs + e];
        rs += RS[(size_t)z * n_local + i];
      }
      const float th = elem_f32(T + (size_t)row0 * d + e);
      const float dk = (rs * th - ot) / h2;
      const float ph = (og + dk) / fn;
      phi[e] = ph;
      if (dK) dK[e] = dk;
      sq += (double)ph * (double)ph;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
  __syncthreads();
  const double part = (red[0] + red[1]) + (red[2] + red[3]);
  if (!done) {
    if (threadIdx.x == 0) sqpart[blockIdx.x] = part;
    return;
  }
  // the partials cross workgroups: written with device-scope atomics, read with load_fresh (last_workgroup_out)
  if (threadIdx.x == 0)
    __hip_atomic_store(reinterpret_cast<u64*>(sqpart) + blockIdx.x, (u64)__double_as_longlong(part), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  {
    __shared__ u32 s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's partial has been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) s_last = tree_report_done(done->fin_leaf, &done->fin_top, blockIdx.x, gridDim.x) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
  }
  double s2 = 0.0;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) s2 += __longlong_as_double((long long)load_fresh(reinterpret_cast<const u64*>(sqpart) + i));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s2;
  __syncthreads();
  if (threadIdx.x == 0) *sq_out = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_sum_partials(const double* __restrict__ part, int count, double* out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < count; i += 256) s += part[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------
// optimizer apply (clip + map + theta += step); arithmetic in fp64, storage S = float or double
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double clip_scale_of(const double* sq, double host_scale, double thr) {
  if (!sq) return host_scale;
  const double nrm = sqrt(*sq);
  return thr / (nrm > thr ? nrm : thr);
}

// one element of the Adagrad map (adagrad_gradient_descent.py:37-44), the same instruction sequence on every path
// (no compiler-chosen fma contraction: the 16-byte and the scalar loops must agree to the last bit)
template <typename C>
__device__ __forceinline__ C adagrad_elem(C p, C h, int first, C a, C na, C ep, C l, C* hs_out) {
#pragma clang fp contract(off)
  const C pp = p * p;
  const C hs = first ? pp : a * h + na * pp;
  *hs_out = hs;
  return p / (ep + sqrt(hs)) * l;
}

template <typename S, typename P>
__global__ __launch_bounds__(256) void k_apply_adagrad(S* __restrict__ theta, const P* __restrict__ phi,
                                                       S* __restrict__ hist, long count, const double* sq,
                                                       double host_scale, double thr, double lr, double alpha,
                                                       double eps, int first, S* __restrict__ step_out, int vec) {
  // arithmetic in the storage type: fp64 state -> fp64 (the reference's NumPy arithmetic), fp32 state -> fp32 (the
  // results are rounded to fp32 anyway, and the fp64 square root and division made the kernel compute-bound: 25 us
  // for 80 MB at C3)
  typedef typename std::conditional<sizeof(S) == 4, float, double>::type C;
  const C scale = (C)clip_scale_of(sq, host_scale, thr);
  const C a = (C)alpha, na = (C)(1.0 - alpha), ep = (C)eps, l = (C)lr;
  if (sizeof(S) == 4 && sizeof(P) == 4 && vec) {   // host: count % 4 == 0, theta present, every pointer 16-byte aligned, no step_out
    float4* th4 = reinterpret_cast<float4*>(theta);
    float4* hi4 = reinterpret_cast<float4*>(hist);
    const float4* ph4 = reinterpret_cast<const float4*>(phi);
    for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < (count >> 2); q += (long)gridDim.x * 256) {
      const float4 pv = ph4[q], hv = hi4[q];
      float4 tv = th4[q], ho;
      const float pp[4] = {pv.x, pv.y, pv.z, pv.w}, hh[4] = {hv.x, hv.y, hv.z, hv.w};
      float tt[4] = {tv.x, tv.y, tv.z, tv.w}, oo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float hs;
        const float step = adagrad_elem<float>(pp[k] * (float)scale, hh[k], first, (float)a, (float)na, (float)ep, (float)l, &hs);
        oo[k] = hs;
        tt[k] = tt[k] + step;
      }
      ho = make_float4(oo[0], oo[1], oo[2], oo[3]);
      tv = make_float4(tt[0], tt[1], tt[2], tt[3]);
      hi4[q] = ho;
      th4[q] = tv;
    }
    return;
  }
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long)gridDim.x * 256) {
    C hs;
    const C step = adagrad_elem<C>((C)phi[e] * scale, (C)hist[e], first, a, na, ep, l, &hs);
    hist[e] = (S)hs;
    if (step_out) step_out[e] = (S)step;
    if (theta) theta[e] = (S)((C)theta[e] + step);
  }
}

template <typename S, typename P>
__global__ __launch_bounds__(256) void k_apply_adam(S* __restrict__ theta, const P* __restrict__ phi,
                                                    S* __restrict__ mu, S* __restrict__ nu, long count,
                                                    const double* sq, double host_scale, double thr, double lr,
                                                    double b1, double b2, double eps, int first, double corr1,
                                                    double corr2, S* __restrict__ step_out) {
  typedef typename std::conditional<sizeof(S) == 4, float, double>::type C;   // as in k_apply_adagrad
  const C scale = (C)clip_scale_of(sq, host_scale, thr);
  const C c1 = (C)b1, n1 = (C)(1.0 - b1), c2 = (C)b2, n2 = (C)(1.0 - b2), ep = (C)eps, l = (C)lr;
  const C k1 = (C)corr1, k2 = (C)corr2;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long)gridDim.x * 256) {
    const C p = (C)phi[e] * scale;
    const C m = first ? p : c1 * (C)mu[e] + n1 * p;
    const C v = first ? p * p : c2 * (C)nu[e] + n2 * p * p;
    mu[e] = (S)m;
    nu[e] = (S)v;
    const C step = (m / k1) / (ep + sqrt(v / k2)) * l;
    if (step_out) step_out[e] = (S)step;
    if (theta) theta[e] = (S)((C)theta[e] + step);
  }
}

__global__ __launch_bounds__(256) void k_cast_f64_f32(const double* __restrict__ s, float* __restrict__ o, long count) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long)gridDim.x * 256) o[e] = (float)s[e];
}
__global__ __launch_bounds__(256) void k_cast_f32_bf16(const float* __restrict__ s, __hip_bfloat16* __restrict__ o,
                                                       long count) {
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < count; e += (long)gridDim.x * 256)
    o[e] = __float2bfloat16(s[e]);
}

// ================================================================================================
// host side
// ================================================================================================
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

int stein_make_layout(int64_t n_local, int64_t n, int64_t d, int dtype, int flags, SteinLayout* L) {
  if (n < 2) return fail(STEIN_E_BADARG, "n = %lld: the bandwidth divides by ln(n), need n >= 2", (long long)n);
  if (d < 1 || n_local < 1 || n_local > n) return fail(STEIN_E_SHAPE, "bad shape n_local=%lld n=%lld d=%lld", (long long)n_local, (long long)n, (long long)d);
  if (n > (1ll << 30) || d > (1ll << 24) || n * d > (1ll << 40)) return fail(STEIN_E_SHAPE, "shape too large");
  if (dtype != STEIN_F32 && dtype != STEIN_BF16) return fail(STEIN_E_UNSUPPORTED, "dtype %d", dtype);
  if (flags & ~(STEIN_FLAG_X3 | STEIN_FLAG_TIMING | STEIN_FLAG_TILED | STEIN_FLAG_NO_WINDOW | STEIN_FLAG_RANK_WINDOW | STEIN_FLAG_TILE_DISTANCE | STEIN_FLAG_TIMING_CONTRACT)) return fail(STEIN_E_BADARG, "unknown flags 0x%x", flags);
  L->ld_dist = (int64_t)align_up((size_t)n, 64);
  L->tiles_m = (n_local + BM - 1) / BM;
  L->cblocks = (d + BN - 1) / BN;
  // workgroups per unit of split and resident workgroups per round: the fp32 kernel tiles [G | theta] in 128-column
  // blocks at 3 workgroups per CU; the split-precision kernel pairs the blocks up (256 columns) at 1 workgroup per CU
  const bool x3 = (flags & STEIN_FLAG_X3) != 0;
  // the split kernel's 64-row x 512-column form (k_phi_x3fs<NP, 4>: P built once per 512 columns, every D tile read by one
  // workgroup; needs an even number of 128-column blocks per matrix; same workgroup count and partial layout).  Measured
  // equal to the 128 x 256 form within +-1 % at C3 and C4 (scratch/README.md), so the shipped library keeps one form.
  L->phi_wide = 0;
  const int64_t base = x3 ? L->tiles_m * L->cblocks : L->tiles_m * 2 * L->cblocks;
  const int64_t jt = (n + BK - 1) / BK;  // j tiles
  // k_phi_partial runs 3 workgroups per CU (156 registers): 768 resident blocks.  Every block does the same
  // work, so the launch takes ceil(blocks / 768) rounds; pick the j-split that wastes least of the last round
  // (ties -> fewer splits, i.e. less partial traffic), keeping at least 8 j-tiles (256 columns) per split.
  const double resident = x3 ? 256.0 : 768.0;
  int64_t max_split = jt / 8 > 0 ? jt / 8 : 1;
  if (max_split > 16) max_split = 16;
  int64_t split = 1;
  double best = -1.0;
  for (int64_t s = 1; s <= max_split; ++s) {
    const double rounds = (double)(base * s) / resident;
    const double eff = rounds / ceil(rounds) - 0.004 * (double)(s - 1);
    if (eff > best + 1e-9) { best = eff; split = s; }
  }
#ifdef STEIN_FORCE_SPLIT   // (diagnostic builds: scratch/build_variant.py splitN -DSTEIN_FORCE_SPLIT=N)
  split = STEIN_FORCE_SPLIT <= max_split ? STEIN_FORCE_SPLIT : max_split;
#endif
  int64_t tiles_per = (jt + split - 1) / split;
  if (x3) tiles_per = (tiles_per + 3) / 4 * 4;   // a j range of the split kernel starts on a multiple of 128 columns (its
                                                 // pipeline stages then never straddle a row tile's diagonal block)
  L->jchunk = tiles_per * BK;
  split = (jt + tiles_per - 1) / tiles_per;  // drop empty tails
  L->split = split;
  const int64_t elems = n_local * d;
  int64_t sqb = (elems + 1023) / 1024;
  if (sqb > 1024) sqb = 1024;
  L->sq_blocks = sqb;

  size_t at = 0;
  auto put = [&](int sec, size_t bytes) { L->off[sec] = at; at = align_up(at + bytes, 256); };
  put(STEIN_WS_ROWNORM, (size_t)n * 4);
  put(STEIN_WS_DIST, align_up((size_t)n_local, DT_ROWS) * L->ld_dist * 4);   // tile-major, rows padded to 128
  put(STEIN_WS_HIST, (size_t)STEIN_HIST_LEVELS * 2 * STEIN_HIST_BINS * 8);
  put(STEIN_WS_SELECT, sizeof(SelState) + sizeof(SpecState) + sizeof(FuseState));
  put(STEIN_WS_PART_G, (size_t)split * n_local * d * 4);
  put(STEIN_WS_PART_T, (size_t)split * n_local * d * 4);
  put(STEIN_WS_PART_RS, (size_t)split * n_local * 4);
  {  // k_phi_finish writes sq_blocks partials, the one-kernel small path one per 32 parameter columns
    const int64_t small = (d + 31) / 32;
    put(STEIN_WS_SQPART, (size_t)(sqb > small ? sqb : small) * 8);
  }
  // slots | entries | rank-summed table.  Empty when the fused call will take the one-kernel path (stein_small.hip never
  // touches it): the reference's own example sizes then carry a few hundred KB of workspace instead of 16.5 MB
  const bool small_path = !(flags & STEIN_FLAG_TILED) && n_local == n && stein_small_ok(n, d, dtype);
  put(STEIN_WS_SPEC, small_path ? 0 : ((size_t)SPEC_CAP + SPEC_SLOTS * 8 + SPEC_TABLE) * 8);
  // split operand planes: always LAST so the offsets above do not depend on the flag
  L->x3_rows = (int64_t)align_up((size_t)n, 128) + 128;   // a rank's last row tile may start past roundup(n, 128) - 128
  L->x3_dk = (int64_t)align_up((size_t)d, 32);
  L->x3_dc = (int64_t)align_up((size_t)d, 128);
  L->x3_nk = (int64_t)align_up((size_t)n, 32);
  const size_t t3 = align_up((size_t)3 * L->x3_rows * L->x3_dk * 2, 256);
  const size_t tt3 = align_up((size_t)3 * L->x3_dc * L->x3_nk * 2, 256);
  L->x3_t3 = 0;
  L->x3_tt3 = t3;
  L->x3_gt3 = t3 + tt3;
  L->x3_sc = t3 + 2 * tt3;   // scales area: float[4 dc + 4] + u32[2 dc]
  const size_t scb = align_up((size_t)(6 * L->x3_dc + 4) * 4, 256);
  put(STEIN_WS_PLANES, (flags & STEIN_FLAG_X3) ? t3 + 2 * tt3 + scb : 0);
  L->total = at;
  return STEIN_OK;
}

static inline int grid_for(long count, int cap) {
  long b = (count + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" int stein_workspace_bytes(int64_t n_local, int64_t n, int64_t d, int dtype, int flags, size_t* out) {
  if (!out) return fail(STEIN_E_BADARG, "out is NULL");
  SteinLayout L;
  int rc = stein_make_layout(n_local, n, d, dtype, flags, &L);
  if (rc) return rc;
  *out = L.total;
  return STEIN_OK;
}

extern "C" int stein_workspace_layout(int64_t n_local, int64_t n, int64_t d, int dtype, int flags, size_t* offsets,
                                      int64_t* extra) {
  if (!offsets || !extra) return fail(STEIN_E_BADARG, "NULL output");
  SteinLayout L;
  int rc = stein_make_layout(n_local, n, d, dtype, flags, &L);
  if (rc) return rc;
  for (int i = 0; i < STEIN_WS_NSECTIONS; ++i) offsets[i] = L.off[i];
  extra[STEIN_WSX_LD_DIST] = L.ld_dist;
  extra[STEIN_WSX_SPLIT] = L.split;
  extra[STEIN_WSX_SQ_BLOCKS] = L.sq_blocks;
  extra[STEIN_WSX_HIST_BINS] = STEIN_HIST_BINS;
  return STEIN_OK;
}

extern "C" int stein_x3_prepare(const void* theta_all, const void* score_all, int64_t n, int64_t d, int dtype,
                                void* x3_planes, size_t planes_bytes, void* stream) {
  if ((!theta_all && !score_all) || !x3_planes) return fail(STEIN_E_BADARG, "NULL pointer");
  SteinLayout L;
  int rc = stein_make_layout(n, n, d, dtype, STEIN_FLAG_X3, &L);
  if (rc) return rc;
  if (planes_bytes < L.total - L.off[STEIN_WS_PLANES])
    return fail(STEIN_E_WORKSPACE, "planes buffer %zu < %zu bytes", planes_bytes, L.total - L.off[STEIN_WS_PLANES]);
  return stein_x3_split(theta_all, score_all, dtype, n, d, L, (char*)x3_planes, (hipStream_t)stream);
}

extern "C" int stein_rownorms(const void* theta_all, int64_t n, int64_t d, int dtype, float* r_out, void* stream) {
  if (!theta_all || !r_out) return fail(STEIN_E_BADARG, "NULL pointer");
  if (n < 1 || d < 1) return fail(STEIN_E_SHAPE, "bad shape");
  if (dtype != STEIN_F32 && dtype != STEIN_BF16) return fail(STEIN_E_UNSUPPORTED, "rownorms: dtype %d", dtype);
  const int blocks = (int)((n + 3) / 4);
  if (dtype == STEIN_BF16)
    hipLaunchKernelGGL(k_rownorms<unsigned short>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)theta_all, (int)n, (int)d, r_out);
  else
    hipLaunchKernelGGL(k_rownorms<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)theta_all,
                       (int)n, (int)d, r_out);
  LAUNCH_CHECK("k_rownorms");
  return STEIN_OK;
}

template <bool VEC, bool SYM>
static void launch_distance(long nblk, hipStream_t s, const float* T, const float* r, float* D, int n, int d, int row0,
                            int n_local, long ld, int tiles_m, int tiles_n, u64* hist0, SpecState* spec, u64* spec_buf) {
  hipLaunchKernelGGL((k_distance<VEC, SYM>), dim3((unsigned)nblk), dim3(NTHREADS), 0, s, T, r, D, n, d, row0, n_local,
                     ld, tiles_m, tiles_n, hist0, spec, spec_buf);
}

// spec != NULL (single-rank fused call only, needs hist_level0): also feed the speculative median window
static int distance_block_impl(const void* theta_all, const float* r_all, int64_t n, int64_t d, int64_t row0,
                               int64_t n_local, int dtype, float* dist_out, int64_t ld_dist, void* hist_level0,
                               const void* x3_planes, int flags, void* stream, SpecState* spec, u64* spec_buf) {
  if ((!theta_all && !x3_planes) || !r_all || !dist_out) return fail(STEIN_E_BADARG, "NULL pointer");
  if (n < 1 || d < 1 || n_local < 1 || row0 < 0 || row0 + n_local > n) return fail(STEIN_E_SHAPE, "bad row block");
  if (ld_dist < n || (ld_dist & 63)) return fail(STEIN_E_SHAPE, "ld_dist must be >= n and a multiple of 64");
  if (dtype != STEIN_F32 && !(dtype == STEIN_BF16 && x3_planes))
    return fail(STEIN_E_UNSUPPORTED, "distance: dtype %d (bf16 inputs need the operand planes)", dtype);
  const bool sym = (flags & STEIN_STAGE_SYMMETRIC) != 0;
  if (sym && (row0 != 0 || n_local != n || (ld_dist & 63)))
    return fail(STEIN_E_BADARG, "STEIN_STAGE_SYMMETRIC needs the whole matrix (row0 = 0, n_local = n) and ld_dist %% 64 == 0");
  const int tiles_m = (int)((n_local + BM - 1) / BM), tiles_n = (int)((n + BN - 1) / BN);
  const long nblk = distance_grid(sym, tiles_m, tiles_n);
  if (nblk > 0x7fffffffl) return fail(STEIN_E_SHAPE, "too many tiles");
  hipStream_t s = (hipStream_t)stream;
  u64* h0 = (u64*)hist_level0;
  if (x3_planes) {
    SteinLayout L;
    int rc = stein_make_layout(n_local, n, d, dtype, STEIN_FLAG_X3, &L);
    if (rc) return rc;
    return stein_x3_distance((const char*)x3_planes, L, dtype, r_all, dist_out, n, d, row0, n_local, ld_dist, h0, sym, s,
                             spec, spec_buf, (flags & STEIN_STAGE_TILES) ? -1 : ((flags & STEIN_STAGE_PANEL) ? 1 : 0));
  }
  const bool vec = (d % 4 == 0) && (((uintptr_t)theta_all & 15) == 0);
  const float* T = (const float*)theta_all;
  if (vec && sym) launch_distance<true, true>(nblk, s, T, r_all, dist_out, (int)n, (int)d, (int)row0, (int)n_local, (long)ld_dist, tiles_m, tiles_n, h0, spec, spec_buf);
  else if (vec) launch_distance<true, false>(nblk, s, T, r_all, dist_out, (int)n, (int)d, (int)row0, (int)n_local, (long)ld_dist, tiles_m, tiles_n, h0, spec, spec_buf);
  else if (sym) launch_distance<false, true>(nblk, s, T, r_all, dist_out, (int)n, (int)d, (int)row0, (int)n_local, (long)ld_dist, tiles_m, tiles_n, h0, spec, spec_buf);
  else launch_distance<false, false>(nblk, s, T, r_all, dist_out, (int)n, (int)d, (int)row0, (int)n_local, (long)ld_dist, tiles_m, tiles_n, h0, spec, spec_buf);
  LAUNCH_CHECK("k_distance");
  return STEIN_OK;
}

extern "C" int stein_distance_block(const void* theta_all, const float* r_all, int64_t n, int64_t d, int64_t row0,
                                    int64_t n_local, int dtype, float* dist_out, int64_t ld_dist, void* hist_level0,
                                    const void* x3_planes, int flags, void* stream) {
  return distance_block_impl(theta_all, r_all, n, d, row0, n_local, dtype, dist_out, ld_dist, hist_level0, x3_planes,
                             flags, stream, nullptr, nullptr);
}

extern "C" int stein_median_begin(void* hist, void* select_state, int64_t total, void* stream) {
  if (!hist || !select_state) return fail(STEIN_E_BADARG, "NULL pointer");
  if (total < 1) return fail(STEIN_E_SHAPE, "total < 1");
  HIP_TRY(hipMemsetAsync(hist, 0, (size_t)STEIN_HIST_LEVELS * 2 * STEIN_HIST_BINS * 8, (hipStream_t)stream));
  hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(64), 0, (hipStream_t)stream, (SelState*)select_state,
                     (u64)total);
  LAUNCH_CHECK("k_sel_init");
  return STEIN_OK;
}

template <int LEVEL>
static void launch_hist(bool sym, int blocks, hipStream_t s, const float* dist, long ld, int n_local, int n,
                        const SelState* st, u64* h, const u32* skip) {
  if (sym)
    hipLaunchKernelGGL((k_hist<LEVEL, true>), dim3(blocks), dim3(256), 0, s, dist, ld, n_local, n, st, h, skip);
  else
    hipLaunchKernelGGL((k_hist<LEVEL, false>), dim3(blocks), dim3(256), 0, s, dist, ld, n_local, n, st, h, skip);
}

static int hist_pass_impl(const float* dist, int64_t ld_dist, int64_t n_local, int64_t n, int level,
                          const void* select_state, void* hist, int flags, void* stream, const u32* skip) {
  if (!dist || !select_state || !hist) return fail(STEIN_E_BADARG, "NULL pointer");
  if (level < 0 || level >= STEIN_HIST_LEVELS) return fail(STEIN_E_BADARG, "level %d", level);
  if (ld_dist < n || (ld_dist & 31) || n_local < 1) return fail(STEIN_E_SHAPE, "bad distance block shape (ld_dist must be a multiple of 32)");
  const bool sym = (flags & STEIN_STAGE_SYMMETRIC) != 0;
  if (sym && n_local != n) return fail(STEIN_E_BADARG, "STEIN_STAGE_SYMMETRIC needs a square block");
  const long units = ((n_local + DT_ROWS - 1) / DT_ROWS) * ((n + DT_COLS - 1) / DT_COLS);   // [128][32] tiles
  const int blocks = (int)(units < HIST_BLOCKS ? units : HIST_BLOCKS);
  u64* h = (u64*)hist + (size_t)level * 2 * STEIN_HIST_BINS;
  const SelState* st = (const SelState*)select_state;
  hipStream_t s = (hipStream_t)stream;
  if (level == 0) launch_hist<0>(sym, blocks, s, dist, (long)ld_dist, (int)n_local, (int)n, st, h, skip);
  else if (level == 1) launch_hist<1>(sym, blocks, s, dist, (long)ld_dist, (int)n_local, (int)n, st, h, skip);
  else launch_hist<2>(sym, blocks, s, dist, (long)ld_dist, (int)n_local, (int)n, st, h, skip);
  LAUNCH_CHECK("k_hist");
  return STEIN_OK;
}

extern "C" int stein_median_hist_pass(const float* dist, int64_t ld_dist, int64_t n_local, int64_t n, int level,
                                      const void* select_state, void* hist, int flags, void* stream) {
  return hist_pass_impl(dist, ld_dist, n_local, n, level, select_state, hist, flags, stream, nullptr);
}

static int resolve_impl(const void* hist, int level, int64_t n, void* select_state, float* h2_out, float* median_out,
                        void* stream, const u32* skip) {
  if (!hist || !select_state) return fail(STEIN_E_BADARG, "NULL pointer");
  if (level < 0 || level >= STEIN_HIST_LEVELS) return fail(STEIN_E_BADARG, "level %d", level);
  if (n < 2) return fail(STEIN_E_BADARG, "n = %lld: need n >= 2", (long long)n);
  const u64* h = (const u64*)hist + (size_t)level * 2 * STEIN_HIST_BINS;
  const float ln_n = (float)log((double)n);  // np.log(n) in fp64, cast to fp32 by the tf.float32 graph
  hipLaunchKernelGGL(k_resolve, dim3(1), dim3(64), 0, (hipStream_t)stream, h, level, (SelState*)select_state, ln_n,
                     h2_out, median_out, skip);
  LAUNCH_CHECK("k_resolve");
  return STEIN_OK;
}

extern "C" int stein_median_resolve(const void* hist, int level, int64_t n, void* select_state, float* h2_out,
                                    float* median_out, void* stream) {
  return resolve_impl(hist, level, n, select_state, h2_out, median_out, stream, nullptr);
}

// ---- speculative window, staged form (several ranks; include/steinhip.h) -------------------------------------------
static inline SpecState* spec_of(void* select_state) { return (SpecState*)((char*)select_state + sizeof(SelState)); }
static inline u64* spec_table_of(void* spec_buf) { return (u64*)spec_buf + SPEC_SLOTS * 8 + SPEC_CAP; }

extern "C" int stein_spec_begin(void* hist, void* select_state, void* spec_buf, int64_t total, void* stream) {
  if (!hist || !select_state || !spec_buf) return fail(STEIN_E_BADARG, "NULL pointer");
  if (total < 1) return fail(STEIN_E_SHAPE, "total < 1");
  hipLaunchKernelGGL(k_median_init, dim3(16), dim3(256), 0, (hipStream_t)stream, (SelState*)select_state,
                     spec_of(select_state), (u64)total, (u64*)hist, (u64*)spec_buf);
  LAUNCH_CHECK("k_median_init");
  return STEIN_OK;
}

extern "C" int stein_distance_block_spec(const void* theta_all, const float* r_all, int64_t n, int64_t d, int64_t row0,
                                         int64_t n_local, int dtype, float* dist_out, int64_t ld_dist, void* hist_level0,
                                         const void* x3_planes, int flags, void* select_state, void* spec_buf,
                                         void* stream) {
  if (!hist_level0 || !select_state || !spec_buf) return fail(STEIN_E_BADARG, "NULL pointer");
  return distance_block_impl(theta_all, r_all, n, d, row0, n_local, dtype, dist_out, ld_dist, hist_level0, x3_planes,
                             flags, stream, spec_of(select_state), (u64*)spec_buf);
}

extern "C" int stein_spec_tally(void* select_state, void* spec_buf, void* stream) {
  if (!select_state || !spec_buf) return fail(STEIN_E_BADARG, "NULL pointer");
  u64* table = spec_table_of(spec_buf);
  HIP_TRY(hipMemsetAsync(table, 0, (size_t)SPEC_TABLE * 8, (hipStream_t)stream));
  hipLaunchKernelGGL(k_spec_tally, dim3(256), dim3(256), 0, (hipStream_t)stream, spec_of(select_state),
                     (const u64*)spec_buf, table);
  LAUNCH_CHECK("k_spec_tally");
  return STEIN_OK;
}

extern "C" int stein_spec_pick(void* select_state, void* spec_buf, int64_t n, float* h2_out, float* median_out,
                               void* stream) {
  if (!select_state || !spec_buf) return fail(STEIN_E_BADARG, "NULL pointer");
  if (n < 2) return fail(STEIN_E_BADARG, "n = %lld: need n >= 2", (long long)n);
  hipLaunchKernelGGL(k_spec_pick, dim3(1), dim3(1024), 0, (hipStream_t)stream, (SelState*)select_state,
                     spec_of(select_state), (const u64*)spec_table_of(spec_buf), (float)log((double)n), h2_out,
                     median_out);
  LAUNCH_CHECK("k_spec_pick");
  return STEIN_OK;
}

extern "C" int stein_spec_update(void* select_state, void* stream) {
  if (!select_state) return fail(STEIN_E_BADARG, "NULL pointer");
  hipLaunchKernelGGL(k_spec_update, dim3(1), dim3(64), 0, (hipStream_t)stream, (const SelState*)select_state,
                     spec_of(select_state));
  LAUNCH_CHECK("k_spec_update");
  return STEIN_OK;
}

extern "C" int stein_kernel_matrix(const float* dist, int64_t ld_dist, int64_t n_local, int64_t n,
                                   const float* h2_dev, float* K_out, int64_t ld_K, int dist_flags, void* stream) {
  if (!dist || !h2_dev || !K_out) return fail(STEIN_E_BADARG, "NULL pointer");
  if (ld_dist < n || ld_K < n || n_local < 1) return fail(STEIN_E_SHAPE, "bad shape");
  if (dist_flags & ~(STEIN_STAGE_SYMMETRIC | STEIN_STAGE_UPPER)) return fail(STEIN_E_BADARG, "unknown distance flags 0x%x", dist_flags);
  if ((dist_flags & STEIN_STAGE_UPPER) && n_local != n) return fail(STEIN_E_BADARG, "STEIN_STAGE_UPPER needs the whole matrix");
  hipLaunchKernelGGL(k_kernel_matrix, dim3(grid_for((long)n_local * n, 4096)), dim3(256), 0, (hipStream_t)stream,
                     dist, (long)ld_dist, (int)n_local, (int)n, h2_dev, K_out, (long)ld_K,
                     (dist_flags & STEIN_STAGE_UPPER) ? 1 : 0);
  LAUNCH_CHECK("k_kernel_matrix");
  return STEIN_OK;
}

extern "C" int stein_contract_partial(const float* dist, int64_t ld_dist, const void* theta_all,
                                      const void* score_all, int64_t n, int64_t d, int64_t row0, int64_t n_local,
                                      int dtype, const float* h2_dev, const void* x3_planes, void* workspace,
                                      size_t ws_bytes, int dist_flags, void* stream) {
  if (dist_flags & ~(STEIN_STAGE_SYMMETRIC | STEIN_STAGE_UPPER)) return fail(STEIN_E_BADARG, "unknown distance flags 0x%x", dist_flags);
  const bool upper = (dist_flags & STEIN_STAGE_UPPER) != 0;
  if (upper && (!x3_planes || row0 != 0 || n_local != n))
    return fail(STEIN_E_BADARG, "STEIN_STAGE_UPPER: only the split path's symmetric distance pass stores the upper triangle alone");
  if (!dist || (!x3_planes && (!theta_all || !score_all)) || !h2_dev || !workspace)
    return fail(STEIN_E_BADARG, "NULL pointer");
  if (dtype != STEIN_F32 && !(dtype == STEIN_BF16 && x3_planes))
    return fail(STEIN_E_UNSUPPORTED, "contract: dtype %d (bf16 inputs need the operand planes)", dtype);
  if (row0 < 0 || row0 + n_local > n) return fail(STEIN_E_SHAPE, "bad row block");
  SteinLayout L;
  int rc = stein_make_layout(n_local, n, d, dtype, x3_planes ? STEIN_FLAG_X3 : 0, &L);
  if (rc) return rc;
  if (ws_bytes < L.off[STEIN_WS_PLANES]) return fail(STEIN_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.off[STEIN_WS_PLANES]);
  if (ld_dist != L.ld_dist) return fail(STEIN_E_SHAPE, "ld_dist %lld != %lld", (long long)ld_dist, (long long)L.ld_dist);
  char* ws = (char*)workspace;
  float* OG = (float*)(ws + L.off[STEIN_WS_PART_G]);
  float* OT = (float*)(ws + L.off[STEIN_WS_PART_T]);
  float* RS = (float*)(ws + L.off[STEIN_WS_PART_RS]);
  const float* T = (const float*)theta_all;
  const float* G = (const float*)score_all;
  hipStream_t s = (hipStream_t)stream;
  const long nblk = (long)L.tiles_m * 2 * L.cblocks * L.split;
  if (nblk > 0x7fffffffl) return fail(STEIN_E_SHAPE, "too many tiles");
  if (x3_planes)
    return stein_x3_contract_partial(dist, ld_dist, (const char*)x3_planes, L, dtype, h2_dev, OG, OT, RS, n, d, n_local, s, upper);
  const bool vec = (d % 4 == 0) && (((uintptr_t)T & 15) == 0) && (((uintptr_t)G & 15) == 0);
  if (vec)
    hipLaunchKernelGGL(k_phi_partial<true>, dim3((unsigned)nblk), dim3(NTHREADS), 0, s, dist, (long)ld_dist, G, T,
                       h2_dev, OG, OT, RS, (int)n, (int)d, (int)n_local, (int)L.tiles_m, (int)L.cblocks, (int)L.split,
                       (int)L.jchunk);
  else
    hipLaunchKernelGGL(k_phi_partial<false>, dim3((unsigned)nblk), dim3(NTHREADS), 0, s, dist, (long)ld_dist, G, T,
                       h2_dev, OG, OT, RS, (int)n, (int)d, (int)n_local, (int)L.tiles_m, (int)L.cblocks, (int)L.split,
                       (int)L.jchunk);
  LAUNCH_CHECK("k_phi_partial");
  return STEIN_OK;
}

static int contract_finish_impl(const void* theta_all, int64_t n, int64_t d, int64_t row0, int64_t n_local,
                                int dtype, const float* h2_dev, float* phi_local, double* sqnorm_out,
                                float* dK_out, void* workspace, size_t ws_bytes, int flags, void* stream, HistSync* fuse_done) {
  if (!theta_all || !h2_dev || !phi_local || !sqnorm_out || !workspace) return fail(STEIN_E_BADARG, "NULL pointer");
  if (dtype != STEIN_F32 && dtype != STEIN_BF16) return fail(STEIN_E_UNSUPPORTED, "contract: dtype %d", dtype);
  if (row0 < 0 || row0 + n_local > n) return fail(STEIN_E_SHAPE, "bad row block");
  SteinLayout L;
  int rc = stein_make_layout(n_local, n, d, dtype, flags, &L);
  if (rc) return rc;
  if (ws_bytes < L.off[STEIN_WS_PLANES]) return fail(STEIN_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.off[STEIN_WS_PLANES]);
  char* ws = (char*)workspace;
  const float* OG = (const float*)(ws + L.off[STEIN_WS_PART_G]);
  const float* OT = (const float*)(ws + L.off[STEIN_WS_PART_T]);
  const float* RS = (const float*)(ws + L.off[STEIN_WS_PART_RS]);
  double* SQ = (double*)(ws + L.off[STEIN_WS_SQPART]);
  hipStream_t s = (hipStream_t)stream;
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
  const size_t tsz = dtype == STEIN_BF16 ? 2 : 4;
  const int vec = (d % 4 == 0) && al16(OG) && al16(OT) && al16(phi_local) && al16(dK_out) &&
                  (((uintptr_t)theta_all + (size_t)row0 * d * tsz) & (4 * tsz - 1)) == 0;
  if (dtype == STEIN_BF16)
    hipLaunchKernelGGL(k_phi_finish<unsigned short>, dim3((unsigned)L.sq_blocks), dim3(256), 0, s, OG, OT, RS,
                       (const unsigned short*)theta_all, h2_dev, phi_local, dK_out, SQ, (int)n, (int)d, (int)row0,
                       (int)n_local, (int)L.split, vec, fuse_done, sqnorm_out);
  else
    hipLaunchKernelGGL(k_phi_finish<float>, dim3((unsigned)L.sq_blocks), dim3(256), 0, s, OG, OT, RS,
                       (const float*)theta_all, h2_dev, phi_local, dK_out, SQ, (int)n, (int)d, (int)row0, (int)n_local,
                       (int)L.split, vec, fuse_done, sqnorm_out);
  LAUNCH_CHECK("k_phi_finish");
  if (!fuse_done) {
    hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, SQ, (int)L.sq_blocks, sqnorm_out);
    LAUNCH_CHECK("k_sum_partials");
  }
  return STEIN_OK;
}

extern "C" int stein_contract_finish(const void* theta_all, int64_t n, int64_t d, int64_t row0, int64_t n_local,
                                     int dtype, const float* h2_dev, float* phi_local, double* sqnorm_out,
                                     float* dK_out, void* workspace, size_t ws_bytes, int flags, void* stream) {
  return contract_finish_impl(theta_all, n, d, row0, n_local, dtype, h2_dev, phi_local, sqnorm_out, dK_out, workspace,
                              ws_bytes, flags, stream, nullptr);
}

extern "C" int stein_kernel_contract(const float* dist, int64_t ld_dist, const void* theta_all, const void* score_all,
                                     int64_t n, int64_t d, int64_t row0, int64_t n_local, int dtype,
                                     const float* h2_dev, float* phi_local, double* sqnorm_out, float* dK_out,
                                     const void* x3_planes, void* workspace, size_t ws_bytes, int dist_flags,
                                     void* stream) {
  int rc = stein_contract_partial(dist, ld_dist, theta_all, score_all, n, d, row0, n_local, dtype, h2_dev, x3_planes,
                                  workspace, ws_bytes, dist_flags, stream);
  if (rc) return rc;
  return stein_contract_finish(theta_all, n, d, row0, n_local, dtype, h2_dev, phi_local, sqnorm_out, dK_out, workspace,
                               ws_bytes, x3_planes ? STEIN_FLAG_X3 : 0, stream);
}

// ------------------------------------------------------------------------------------------------
// rank-step segments: what one rank of a row-sharded run does between two collectives, as ONE call each (the staged calls
// above, chained on the stream).  The host layer issues: all-gather(theta), all-gather(score) | stein_rank_begin |
// all-reduce(window table or level-0 histogram) | stein_rank_pick or stein_rank_radix x3 (an all-reduce before each) |
// stein_rank_finish | all-reduce(|phi|^2).
// ------------------------------------------------------------------------------------------------
static thread_local std::vector<hipEvent_t> g_tevents;   // (STEIN_T_NSTAGES + 1) events per reserved call
static thread_local int g_tcalls_reserved = 0, g_tcalls_used = 0;
static thread_local std::vector<unsigned char> g_tmode;   // per reserved call: 1 = only the contraction was bracketed

struct RankViews {
  SteinLayout L;
  float* r; float* D; u64* hist; char* sel; u64* spec_buf; char* planes;
};
static int rank_views(int64_t n, int64_t d, int64_t row0, int64_t n_local, int dtype, void* workspace, size_t ws_bytes,
                      int flags, RankViews* v) {
  if (!workspace) return fail(STEIN_E_BADARG, "NULL pointer");
  if (row0 < 0 || n_local < 1 || row0 + n_local > n) return fail(STEIN_E_SHAPE, "bad row block");
  if (flags & ~(STEIN_FLAG_X3 | STEIN_FLAG_TILED | STEIN_FLAG_RANK_WINDOW | STEIN_FLAG_TIMING)) return fail(STEIN_E_BADARG, "unknown flags 0x%x", flags);
  int rc = stein_make_layout(n_local, n, d, dtype, (flags & (STEIN_FLAG_X3 | STEIN_FLAG_TILED)) | STEIN_FLAG_TILED, &v->L);
  if (rc) return rc;
  if (dtype == STEIN_BF16 && !(flags & STEIN_FLAG_X3)) return fail(STEIN_E_UNSUPPORTED, "bf16 inputs need STEIN_FLAG_X3");
  if (ws_bytes < v->L.total) return fail(STEIN_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, v->L.total);
  char* ws = (char*)workspace;
  v->r = (float*)(ws + v->L.off[STEIN_WS_ROWNORM]);
  v->D = (float*)(ws + v->L.off[STEIN_WS_DIST]);
  v->hist = (u64*)(ws + v->L.off[STEIN_WS_HIST]);
  v->sel = ws + v->L.off[STEIN_WS_SELECT];
  v->spec_buf = (u64*)(ws + v->L.off[STEIN_WS_SPEC]);
  v->planes = (flags & STEIN_FLAG_X3) ? ws + v->L.off[STEIN_WS_PLANES] : nullptr;
  return STEIN_OK;
}

extern "C" int stein_rank_begin(const void* theta_all, int64_t n, int64_t d, int64_t row0, int64_t n_local, int dtype,
                                void* workspace, size_t ws_bytes, int flags, void* stream) {
  if (!theta_all) return fail(STEIN_E_BADARG, "NULL pointer");
  RankViews v;
  int rc = rank_views(n, d, row0, n_local, dtype, workspace, ws_bytes, flags, &v);
  if (rc) return rc;
  const bool window = (flags & STEIN_FLAG_RANK_WINDOW) != 0;
  if ((rc = stein_rownorms(theta_all, n, d, dtype, v.r, stream))) return rc;
  if (v.planes && (rc = stein_x3_split(theta_all, nullptr, dtype, n, d, v.L, v.planes, (hipStream_t)stream))) return rc;
  if (window) rc = stein_spec_begin(v.hist, v.sel, v.spec_buf, n * n, stream);
  else rc = stein_median_begin(v.hist, v.sel, n * n, stream);
  if (rc) return rc;
  if ((rc = distance_block_impl(theta_all, v.r, n, d, row0, n_local, dtype, v.D, v.L.ld_dist, v.hist, v.planes, 0, stream,
                                window ? spec_of(v.sel) : nullptr, window ? v.spec_buf : nullptr)))
    return rc;
  if (window) rc = stein_spec_tally(v.sel, v.spec_buf, stream);
  return rc;
}

extern "C" int stein_rank_pick(int64_t n, int64_t d, int64_t row0, int64_t n_local, int dtype, void* workspace,
                               size_t ws_bytes, int flags, float* h2_out, float* median_out, void* flags_host,
                               void* stream) {
  if (!h2_out || !flags_host) return fail(STEIN_E_BADARG, "NULL pointer");
  RankViews v;
  int rc = rank_views(n, d, row0, n_local, dtype, workspace, ws_bytes, flags, &v);
  if (rc) return rc;
  if ((rc = stein_spec_pick(v.sel, v.spec_buf, n, h2_out, median_out, stream))) return rc;
  // `hit` (SpecState + 28) .. `skip_l0` (+ 52): 28 bytes, to page-locked host memory the caller polls behind an event
  HIP_TRY(hipMemcpyAsync(flags_host, v.sel + sizeof(SelState) + 28, 28, hipMemcpyDeviceToHost, (hipStream_t)stream));
  return STEIN_OK;
}

extern "C" int stein_rank_radix(int level, int need_pass, int64_t n, int64_t d, int64_t row0, int64_t n_local, int dtype,
                                void* workspace, size_t ws_bytes, int flags, float* h2_out, float* median_out,
                                void* stream) {
  // need_pass: first take this level's histogram of the local block (level 0 after a window miss that skipped it);
  // then (the caller has summed hist[level] over the ranks unless need_pass) ... see include/steinhip.h
  RankViews v;
  int rc = rank_views(n, d, row0, n_local, dtype, workspace, ws_bytes, flags, &v);
  if (rc) return rc;
  if (level < 0 || level >= STEIN_HIST_LEVELS) return fail(STEIN_E_BADARG, "level %d", level);
  if (need_pass) return hist_pass_impl(v.D, v.L.ld_dist, n_local, n, level, v.sel, v.hist, 0, stream, nullptr);
  if ((rc = resolve_impl(v.hist, level, n, v.sel, h2_out, median_out, stream, nullptr))) return rc;
  if (level + 1 < STEIN_HIST_LEVELS)
    rc = hist_pass_impl(v.D, v.L.ld_dist, n_local, n, level + 1, v.sel, v.hist, 0, stream, nullptr);
  return rc;
}

extern "C" int stein_rank_finish(const void* theta_all, const void* score_all, int64_t n, int64_t d, int64_t row0,
                                 int64_t n_local, int dtype, const float* h2_dev, float* phi_local, double* sqnorm_out,
                                 float* dK_out, void* workspace, size_t ws_bytes, int flags, void* stream) {
  if (!theta_all || !score_all || !h2_dev || !phi_local || !sqnorm_out) return fail(STEIN_E_BADARG, "NULL pointer");
  RankViews v;
  int rc = rank_views(n, d, row0, n_local, dtype, workspace, ws_bytes, flags, &v);
  if (rc) return rc;
  if ((flags & STEIN_FLAG_RANK_WINDOW) && (rc = stein_spec_update(v.sel, stream))) return rc;
  // STEIN_FLAG_TIMING: the contraction and the finish pass are bracketed by HIP events on the stream (the earlier
  // stages of the slot read as zero length); read them back with stein_timing_read
  hipEvent_t* tev = nullptr;
  if ((flags & STEIN_FLAG_TIMING) && g_tcalls_used < g_tcalls_reserved) {
    g_tmode[(size_t)g_tcalls_used] = 0;
    tev = &g_tevents[(size_t)(g_tcalls_used++) * (STEIN_T_NSTAGES + 1)];
  }
  if (tev)
    for (int k = 0; k <= STEIN_T_CONTRACT; ++k) HIP_TRY(hipEventRecord(tev[k], (hipStream_t)stream));
  if ((rc = stein_contract_partial(v.D, v.L.ld_dist, theta_all, score_all, n, d, row0, n_local, dtype, h2_dev, v.planes,
                                   workspace, ws_bytes, 0, stream)))
    return rc;
  if (tev) HIP_TRY(hipEventRecord(tev[STEIN_T_FINISH], (hipStream_t)stream));
  rc = stein_contract_finish(theta_all, n, d, row0, n_local, dtype, h2_dev, phi_local, sqnorm_out, dK_out, workspace,
                             ws_bytes, v.planes ? STEIN_FLAG_X3 : 0, stream);
  if (tev && !rc) HIP_TRY(hipEventRecord(tev[STEIN_T_NSTAGES], (hipStream_t)stream));
  return rc;
}

// ------------------------------------------------------------------------------------------------
// device error word: one u32 per device in page-locked host memory that a kernel raises when it had to give up (today
// only k_hist_all's bounded wait).  The host reads it without touching the stream at the start of the next fused call or
// optimizer apply on that device and turns it into STEIN_E_HIP; the step that raised it has already written NaN into its
// bandwidth, so nothing wrong was consumed silently in between.
// ------------------------------------------------------------------------------------------------
constexpr int MAX_DEVICES = 64;
static u32* g_errword[MAX_DEVICES];
static int device_error_word(u32** out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= MAX_DEVICES) { *out = nullptr; return STEIN_OK; }
  u32* w = __atomic_load_n(&g_errword[dev], __ATOMIC_ACQUIRE);
  if (!w) {
    void* p = nullptr;
    HIP_TRY(hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable));
    *(volatile u32*)p = 0u;
    u32* expect = nullptr;
    if (!__atomic_compare_exchange_n(&g_errword[dev], &expect, (u32*)p, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
      (void)hipHostFree(p);   // another thread was first
      w = expect;
    } else {
      w = (u32*)p;
    }
  }
  *out = w;
  return STEIN_OK;
}
// STEIN_E_HIP if a kernel of an EARLIER call on the current device raised the error word (and lowers it again)
int stein_take_device_error(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) return STEIN_OK;
  u32* w = __atomic_load_n(&g_errword[dev], __ATOMIC_ACQUIRE);
  if (!w || !*(volatile u32*)w) return STEIN_OK;
  *(volatile u32*)w = 0u;
  return fail(STEIN_E_HIP, "an earlier step on device %d gave up inside k_hist_all (bounded wait exhausted): its bandwidth and "
                           "everything computed from it are NaN", dev);
}
// test hooks (per calling thread; tests/test_gpu_spec.py): launch k_hist_all with this many workgroups instead of one per
// virtual workgroup (0 = default); raise the current device's error word as a kernel would
static thread_local int g_hist_all_grid = 0, g_hist_all_nvb = 0;
extern "C" int stein_debug_hist_all_grid(int blocks) {
  if (blocks < 0 || blocks > 65535) return fail(STEIN_E_BADARG, "blocks %d", blocks);
  g_hist_all_grid = blocks;
  return STEIN_OK;
}
extern "C" int stein_debug_hist_all_vblocks(int nvb) {   // (tuning aid: virtual workgroups per level, 0 = default)
  if (nvb < 0 || nvb > 65535) return fail(STEIN_E_BADARG, "nvb %d", nvb);
  g_hist_all_nvb = nvb;
  return STEIN_OK;
}
extern "C" int stein_debug_raise_device_error(void) {
  u32* w = nullptr;
  int rc = device_error_word(&w);
  if (rc) return rc;
  if (w) *(volatile u32*)w = 1u;
  return STEIN_OK;
}

// ------------------------------------------------------------------------------------------------
// stage timing of the fused call (profiling aid; per calling thread, like the last-error string)
// ------------------------------------------------------------------------------------------------

extern "C" int stein_timing_reserve(int calls) {
  if (calls < 0) return fail(STEIN_E_BADARG, "calls < 0");
  const size_t need = (size_t)calls * (STEIN_T_NSTAGES + 1);
  while (g_tevents.size() < need) {
    hipEvent_t e;
    HIP_TRY(hipEventCreate(&e));
    g_tevents.push_back(e);
  }
  g_tmode.assign((size_t)calls, 0);
  g_tcalls_reserved = calls;
  g_tcalls_used = 0;
  return STEIN_OK;
}

extern "C" int stein_timing_read(float* ms_out, int max_calls, int* calls_out) {
  if (!ms_out || !calls_out) return fail(STEIN_E_BADARG, "NULL pointer");
  const int calls = g_tcalls_used < max_calls ? g_tcalls_used : max_calls;
  for (int c = 0; c < calls; ++c) {
    hipEvent_t* ev = &g_tevents[(size_t)c * (STEIN_T_NSTAGES + 1)];
    if (g_tmode[(size_t)c]) {   // STEIN_FLAG_TIMING_CONTRACT: the other stages were not bracketed
      HIP_TRY(hipEventSynchronize(ev[STEIN_T_FINISH]));
      for (int k = 0; k < STEIN_T_NSTAGES; ++k) ms_out[c * STEIN_T_NSTAGES + k] = -1.f;
      HIP_TRY(hipEventElapsedTime(&ms_out[c * STEIN_T_NSTAGES + STEIN_T_CONTRACT], ev[STEIN_T_CONTRACT], ev[STEIN_T_FINISH]));
      continue;
    }
    HIP_TRY(hipEventSynchronize(ev[STEIN_T_NSTAGES]));
    for (int k = 0; k < STEIN_T_NSTAGES; ++k) HIP_TRY(hipEventElapsedTime(&ms_out[c * STEIN_T_NSTAGES + k], ev[k], ev[k + 1]));
  }
  *calls_out = calls;
  return STEIN_OK;
}

extern "C" int stein_svgd_phi(const void* theta_all, const void* score_all, int64_t n, int64_t d, int64_t row0,
                              int64_t n_local, int dtype, float* phi_local, float* h2_out, double* sqnorm_out,
                              float* K_out, float* dK_out, void* workspace, size_t ws_bytes, int flags, void* stream) {
  if (!theta_all || !score_all || !phi_local || !h2_out || !sqnorm_out || !workspace)
    return fail(STEIN_E_BADARG, "NULL pointer");
  if (row0 != 0 || n_local != n)
    return fail(STEIN_E_BADARG, "stein_svgd_phi is the single-rank path (row0 = 0, n_local = n); use the staged calls");
  SteinLayout L;
  int rc = stein_make_layout(n_local, n, d, dtype, flags, &L);
  if (rc) return rc;
  if ((rc = stein_take_device_error())) return rc;   // a kernel of an earlier call on this device gave up: say so now
  if (dtype == STEIN_BF16 && !(flags & STEIN_FLAG_X3))
    return fail(STEIN_E_UNSUPPORTED, "bf16 inputs run on the bf16-MFMA kernels: pass STEIN_FLAG_X3");
  if (ws_bytes < L.total) return fail(STEIN_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, L.total);
  char* ws = (char*)workspace;
  float* r = (float*)(ws + L.off[STEIN_WS_ROWNORM]);
  float* D = (float*)(ws + L.off[STEIN_WS_DIST]);
  void* hist = ws + L.off[STEIN_WS_HIST];
  void* sel = ws + L.off[STEIN_WS_SELECT];
  void* planes = (flags & STEIN_FLAG_X3) ? ws + L.off[STEIN_WS_PLANES] : nullptr;
  // single rank: the block is the whole symmetric matrix -> upper-triangle distance pass with mirrored stores,
  // level-0 histogram taken in its epilogue, levels 1-2 read the upper triangle only
  const int sf = STEIN_STAGE_SYMMETRIC | ((flags & STEIN_FLAG_TILE_DISTANCE) ? STEIN_STAGE_TILES : 0);
  SpecState* spec = (SpecState*)((char*)sel + sizeof(SelState));
  u64* spec_buf = (u64*)(ws + L.off[STEIN_WS_SPEC]);
  hipStream_t s = (hipStream_t)stream;
  hipEvent_t* tev = nullptr;   // STEIN_FLAG_TIMING: one event per stage boundary, while reserved slots last
  // (STEIN_FLAG_TIMING_CONTRACT: only the two events around the contraction -- an event between two kernels costs the step
  // ~3 us of GPU time, scratch/event_cost.py: six of them are 2 % of a C3 step and a quarter of a C2 step)
  const bool tonly = (flags & STEIN_FLAG_TIMING_CONTRACT) != 0;
  if ((flags & STEIN_FLAG_TIMING) && g_tcalls_used < g_tcalls_reserved) {
    g_tmode[(size_t)g_tcalls_used] = tonly ? 1 : 0;
    tev = &g_tevents[(size_t)(g_tcalls_used++) * (STEIN_T_NSTAGES + 1)];
  }
#define STEIN_TSTAMP(k) do { if (tev && (!tonly || (k) == STEIN_T_CONTRACT || (k) == STEIN_T_FINISH)) HIP_TRY(hipEventRecord(tev[k], s)); } while (0)
  STEIN_TSTAMP(STEIN_T_PREPARE);
  if (!(flags & STEIN_FLAG_TILED) && stein_small_ok(n, d, dtype)) {   // the reference's own example sizes: one kernel does it all (stein_small.hip)
    STEIN_TSTAMP(STEIN_T_DISTANCE);
    STEIN_TSTAMP(STEIN_T_MEDIAN);
    STEIN_TSTAMP(STEIN_T_CONTRACT);
    int nparts = 0;
    double* sqp = (double*)(ws + L.off[STEIN_WS_SQPART]);
    if ((rc = stein_small_phi((const float*)theta_all, (const float*)score_all, n, d, phi_local, h2_out, sqp, K_out,
                              dK_out, &nparts, sqnorm_out, s)))
      return rc;
    STEIN_TSTAMP(STEIN_T_FINISH);
    if (nparts) {   // d > 32: several workgroups, their partials are summed here
      hipLaunchKernelGGL(k_sum_partials, dim3(1), dim3(256), 0, s, (const double*)sqp, nparts, sqnorm_out);
      LAUNCH_CHECK("k_sum_partials");
    }
    STEIN_TSTAMP(STEIN_T_NSTAGES);
    return STEIN_OK;
  }
  // The prologue carries the row norms and all set-up; kernels let their last workgroup do what a one-workgroup follow-up
  // launch would (FuseState tickets: scales, |phi|^2 sum); the chained radix select is ONE launch (k_hist_all) and none
  // for n <= SOLO_MAX_N (k_spec_select covers it); bf16 inputs need no scales, so their prologue rides in the split's
  // launch.  fp32 inputs: k_prologue, k_colmax, k_split, distance, k_spec_select, k_hist_all, contraction, k_phi_finish:
  // eight launches whatever n (the last workgroups of k_colmax and k_phi_finish are found with two-level completion counts,
  // HistSync, so no grid is too large for them); bf16 inputs: six.
  FuseState* fuse = (FuseState*)((char*)sel + sizeof(SelState) + sizeof(SpecState));
  {
    u32* cmax = nullptr;
    int ncmax = 0;
    if (planes) {   // the column maxima behind the scales (stein_x3.hip)
      cmax = (u32*)((float*)((char*)planes + L.x3_sc) + 4 * L.x3_dc + 4);
      ncmax = (int)(2 * L.x3_dc);
    }
    PrologueArgs pa;
    pa.n = (int)n; pa.d = (int)d; pa.r = r; pa.st = (SelState*)sel; pa.sp = spec; pa.fs = fuse; pa.total = (u64)(n * n);
    pa.hist = (u64*)hist; pa.slots = spec_buf; pa.cmax = cmax; pa.ncmax = ncmax;
    pa.allow_window = (flags & STEIN_FLAG_NO_WINDOW) ? 0 : 1;
    pa.hsync = (u32*)spec_table_of(spec_buf); pa.hsync_words = (int)(sizeof(HistSync) / 4);
    pa.neutral_sc = (dtype == STEIN_BF16 && planes) ? (float*)((char*)planes + L.x3_sc) : (float*)nullptr;
    pa.dc = (int)L.x3_dc;
    if (dtype == STEIN_BF16 && planes) {
      // bf16 inputs need no scales, so the split does not depend on the prologue: both ride in ONE launch (k_split's grid
      // gets a third slice that does the prologue's work) -- one launch less on the latency-bound sizes this dtype is for
      if ((rc = stein_x3_split(theta_all, score_all, dtype, n, d, L, (char*)planes, s, (HistSync*)spec_table_of(spec_buf), true, &pa))) return rc;
    } else {
      const int row_blocks = (int)((n + 3) / 4);
      const dim3 grid((unsigned)(row_blocks + PRO_INIT_BLOCKS));
      if (dtype == STEIN_BF16)
        hipLaunchKernelGGL(k_prologue<unsigned short>, grid, dim3(256), 0, s, (const unsigned short*)theta_all, pa);
      else
        hipLaunchKernelGGL(k_prologue<float>, grid, dim3(256), 0, s, (const float*)theta_all, pa);
      LAUNCH_CHECK("k_prologue");
      if (planes && (rc = stein_x3_split(theta_all, score_all, dtype, n, d, L, (char*)planes, s, (HistSync*)spec_table_of(spec_buf), false, nullptr)))
        return rc;
    }
  }
  STEIN_TSTAMP(STEIN_T_DISTANCE);
  if ((rc = distance_block_impl(theta_all, r, n, d, row0, n_local, dtype, D, L.ld_dist, hist, planes, sf, stream, spec,
                                spec_buf)))
    return rc;
  STEIN_TSTAMP(STEIN_T_MEDIAN);
  // the window either yields the median now (spec->hit) or the radix-select passes below run; each of them
  // checks the flag on the device, so nothing here waits for the host
  const bool solo = n <= SOLO_MAX_N;   // small block: a miss is resolved inside k_spec_select, no histogram launches
  hipLaunchKernelGGL(k_spec_select, dim3(1), dim3(1024), 0, s, (SelState*)sel, spec, spec_buf,
                     (float)log((double)n), h2_out, 1, solo ? (const float*)D : (const float*)nullptr, (long)L.ld_dist,
                     (int)n);
  LAUNCH_CHECK("k_spec_select");
  // chained radix select, ONE launch whatever n (k_hist_all: in-launch level barriers that need no co-residency; it returns
  // at once when the window hit).  Level 0 comes from the distance epilogue unless this step had a window (then only a
  // miss needs it, and the launch takes it itself: SpecState::skip_l0).
  if (!solo) {
    const HistFinal fin{(SelState*)sel, spec, h2_out, (float)log((double)n)};
    const long units = ((n + DT_ROWS - 1) / DT_ROWS) * ((n + DT_COLS - 1) / DT_COLS);
    // large blocks: as many virtual workgroups as the chip holds at once (the passes like many loads in flight, and a grid
    // beyond residency would leave the surplus to the thieves); small ones: 512 (measured at C2)
    long want = HIST_ALL_VBLOCKS;
    if (n > HIST_ALL_SMALL_N) {
      static int resident = 0;   // (a benign race: every thread computes the same value)
      if (!resident) {
        int per_cu = 0, dev = 0, ncu = 0;
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_hist_all, 256, 0));
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        const long r = (long)(per_cu > 0 ? per_cu : 4) * (ncu > 0 ? ncu : 256);
        resident = (int)(r > HIST_BLOCKS ? HIST_BLOCKS : r);
      }
      want = resident;
    }
    if (g_hist_all_nvb > 0) want = g_hist_all_nvb;
    if (want > HS_NV) want = HS_NV;   // (HistSync::claim holds one flag per virtual workgroup)
    const int nvb = (int)(units < want ? units : want);
    const int blocks = g_hist_all_grid > 0 ? g_hist_all_grid : nvb;   // (test hook: any grid >= 1 must give the same median)
    u32* errword = nullptr;
    if ((rc = device_error_word(&errword))) return rc;
    hipLaunchKernelGGL(k_hist_all, dim3(blocks), dim3(256), 0, s, (const float*)D, (long)L.ld_dist, (int)n,
                       (const SelState*)sel, (u64*)hist, (const u32*)&spec->hit, (const u32*)&spec->skip_l0, fin, fuse,
                       (HistSync*)spec_table_of(spec_buf), (u32)nvb, errword);
    LAUNCH_CHECK("k_hist_all");
  }
  // the split path's symmetric distance pass stores only the tiles on and above the diagonal
  const int df = planes ? (STEIN_STAGE_SYMMETRIC | STEIN_STAGE_UPPER) : STEIN_STAGE_SYMMETRIC;
  if (K_out && (rc = stein_kernel_matrix(D, L.ld_dist, n_local, n, h2_out, K_out, n, df, stream))) return rc;
  STEIN_TSTAMP(STEIN_T_CONTRACT);
  if ((rc = stein_contract_partial(D, L.ld_dist, theta_all, score_all, n, d, row0, n_local, dtype, h2_out, planes,
                                   workspace, ws_bytes, df, stream)))
    return rc;
  STEIN_TSTAMP(STEIN_T_FINISH);
  if ((rc = contract_finish_impl(theta_all, n, d, row0, n_local, dtype, h2_out, phi_local, sqnorm_out, dK_out, workspace,
                                ws_bytes, planes ? STEIN_FLAG_X3 : 0, stream, (HistSync*)spec_table_of(spec_buf))))
    return rc;
  STEIN_TSTAMP(STEIN_T_NSTAGES);
#undef STEIN_TSTAMP
  return STEIN_OK;
}

template <typename S, typename P>
static int apply_adagrad_t(void* theta, const void* phi, void* hist, int64_t count, const double* sq, double hs,
                           double thr, double lr, double alpha, double eps, int first, void* step_out, void* stream) {
  auto al16 = [](const void* p) { return ((uintptr_t)p & 15u) == 0; };
  const int vec = sizeof(S) == 4 && sizeof(P) == 4 && count % 4 == 0 && theta && !step_out && al16(theta) && al16(phi) && al16(hist);
  hipLaunchKernelGGL((k_apply_adagrad<S, P>), dim3(grid_for(vec ? count / 4 : count, 2048)), dim3(256), 0, (hipStream_t)stream,
                     (S*)theta, (const P*)phi, (S*)hist, (long)count, sq, hs, thr, lr, alpha, eps, first, (S*)step_out, vec);
  LAUNCH_CHECK("k_apply_adagrad");
  return STEIN_OK;
}

// phi_dtype STEIN_F64 needs fp64 state: the reference's pure-fp64 `gd.update(phi)` (no rounding of phi to fp32)
static int check_apply_dtypes(int state_dtype, int phi_dtype) {
  if (state_dtype != STEIN_F32 && state_dtype != STEIN_F64) return fail(STEIN_E_UNSUPPORTED, "state dtype %d", state_dtype);
  if (phi_dtype != STEIN_F32 && phi_dtype != STEIN_F64) return fail(STEIN_E_UNSUPPORTED, "phi dtype %d", phi_dtype);
  if (phi_dtype == STEIN_F64 && state_dtype != STEIN_F64) return fail(STEIN_E_UNSUPPORTED, "fp64 phi needs fp64 optimizer state");
  return STEIN_OK;
}

extern "C" int stein_apply_adagrad(void* theta, const void* phi, int phi_dtype, void* hist, int64_t count, int state_dtype,
                                   const double* sqnorm_dev, double clip_scale_host, double clip_threshold, double lr,
                                   double alpha, double eps, int first_step, void* step_out, void* stream) {
  if (!phi || !hist) return fail(STEIN_E_BADARG, "NULL pointer");
  if (count < 1) return fail(STEIN_E_SHAPE, "count < 1");
  if (int rc = check_apply_dtypes(state_dtype, phi_dtype)) return rc;
  if (int rc = stein_take_device_error()) return rc;
  if (state_dtype == STEIN_F32)
    return apply_adagrad_t<float, float>(theta, phi, hist, count, sqnorm_dev, clip_scale_host, clip_threshold, lr, alpha, eps,
                                         first_step, step_out, stream);
  if (phi_dtype == STEIN_F32)
    return apply_adagrad_t<double, float>(theta, phi, hist, count, sqnorm_dev, clip_scale_host, clip_threshold, lr, alpha,
                                          eps, first_step, step_out, stream);
  return apply_adagrad_t<double, double>(theta, phi, hist, count, sqnorm_dev, clip_scale_host, clip_threshold, lr, alpha,
                                         eps, first_step, step_out, stream);
}

template <typename S, typename P>
static int apply_adam_t(void* theta, const void* phi, void* mu, void* nu, int64_t count, const double* sq, double hs,
                        double thr, double lr, double b1, double b2, double eps, int64_t t, void* step_out,
                        void* stream) {
  const double corr1 = 1.0 - pow(b1, (double)t), corr2 = 1.0 - pow(b2, (double)t);
  hipLaunchKernelGGL((k_apply_adam<S, P>), dim3(grid_for(count, 2048)), dim3(256), 0, (hipStream_t)stream, (S*)theta,
                     (const P*)phi, (S*)mu, (S*)nu, (long)count, sq, hs, thr, lr, b1, b2, eps, t == 1 ? 1 : 0, corr1, corr2,
                     (S*)step_out);
  LAUNCH_CHECK("k_apply_adam");
  return STEIN_OK;
}

extern "C" int stein_apply_adam(void* theta, const void* phi, int phi_dtype, void* mu, void* nu, int64_t count,
                                int state_dtype, const double* sqnorm_dev, double clip_scale_host, double clip_threshold,
                                double lr, double beta1, double beta2, double eps, int64_t t, void* step_out, void* stream) {
  if (!phi || !mu || !nu) return fail(STEIN_E_BADARG, "NULL pointer");
  if (count < 1 || t < 1) return fail(STEIN_E_SHAPE, "count < 1 or t < 1");
  if (int rc = check_apply_dtypes(state_dtype, phi_dtype)) return rc;
  if (int rc = stein_take_device_error()) return rc;
  if (state_dtype == STEIN_F32)
    return apply_adam_t<float, float>(theta, phi, mu, nu, count, sqnorm_dev, clip_scale_host, clip_threshold, lr, beta1,
                                      beta2, eps, t, step_out, stream);
  if (phi_dtype == STEIN_F32)
    return apply_adam_t<double, float>(theta, phi, mu, nu, count, sqnorm_dev, clip_scale_host, clip_threshold, lr, beta1,
                                       beta2, eps, t, step_out, stream);
  return apply_adam_t<double, double>(theta, phi, mu, nu, count, sqnorm_dev, clip_scale_host, clip_threshold, lr, beta1,
                                      beta2, eps, t, step_out, stream);
}

extern "C" int stein_cast_f64_to_f32(const double* src, float* dst, int64_t count, void* stream) {
  if (!src || !dst) return fail(STEIN_E_BADARG, "NULL pointer");
  if (count < 1) return fail(STEIN_E_SHAPE, "count < 1");
  hipLaunchKernelGGL(k_cast_f64_f32, dim3(grid_for(count, 2048)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     (long)count);
  LAUNCH_CHECK("k_cast_f64_f32");
  return STEIN_OK;
}

extern "C" int stein_cast_f32_to_bf16(const float* src, void* dst, int64_t count, void* stream) {
  if (!src || !dst) return fail(STEIN_E_BADARG, "NULL pointer");
  if (count < 1) return fail(STEIN_E_SHAPE, "count < 1");
  hipLaunchKernelGGL(k_cast_f32_bf16, dim3(grid_for(count, 2048)), dim3(256), 0, (hipStream_t)stream, src,
                     (__hip_bfloat16*)dst, (long)count);
  LAUNCH_CHECK("k_cast_f32_bf16");
  return STEIN_OK;
}
