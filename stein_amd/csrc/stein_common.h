// stein_common.h -- device helpers shared by the fp32-MFMA kernels (steinhip.hip) and the split-precision
// kernels (stein_x3.hip): tile geometry, tile-id mapping, radix-select keys/state, and the two epilogues.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "steinhip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;
typedef unsigned int u32;

// ---- error plumbing (defined in steinhip.hip) ---------------------------------------------------
int stein_fail(int code, const char* fmt, ...);

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return stein_fail(STEIN_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

#define LAUNCH_CHECK(name)                                                                           \
  do {                                                                                               \
    hipError_t e_ = hipGetLastError();                                                               \
    if (e_ != hipSuccess) return stein_fail(STEIN_E_HIP, "launch %s: %s", name, hipGetErrorString(e_)); \
  } while (0)

// ---- tiling constants shared by the MFMA kernels ---------------------------------------------------
constexpr int BM = 128;        // rows of the output tile per workgroup
constexpr int BN = 128;        // columns of the output tile per workgroup
constexpr int BK = 32;         // contraction depth staged per iteration
constexpr int NTHREADS = 256;  // 4 waves, arranged 2x2, each owning a 64x64 sub-tile = 2x2 MFMA 32x32 tiles

// Blocks b and b+8 share an XCD (round-robin dispatch); give every XCD one contiguous range of
// logical tile ids so neighbouring tiles (which share operand panels) hit the same L2.  Speed only.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// ---- layout of the distance block -------------------------------------------------------------------
// D is stored tile-major: tile (I, J) = rows [128 I, 128 I + 128) x columns [32 J, 32 J + 32) of the block is one
// contiguous [128][32] fp32 array (16 KB) at ((I * ntc + J) * 4096) floats, ntc = ld / 32 column tiles per row block
// (ld = the column count padded to 64, the `ld_dist` of the C ABI; rows are padded to a multiple of 128).
// Every reader of D (histogram passes, the contraction's producers) works tile by tile, so each of their wave-wide
// loads is one 1 KB burst; with row-major rows 4 ld bytes apart a 128 x 32 tile was 128 separate cache-line visits
// and its loads cost three times as much issue time per byte as the operand-plane loads.
constexpr int DT_ROWS = 128, DT_COLS = 32, DT_ELEMS = DT_ROWS * DT_COLS;
__host__ __device__ __forceinline__ size_t d_index(long row, long col, long ntc) {
  return ((size_t)(row >> 7) * ntc + (col >> 5)) * DT_ELEMS + (row & 127) * DT_COLS + (col & 31);
}

__device__ __forceinline__ float4 ld4_or_zero(const float* p, bool ok) {
  return ok ? *reinterpret_cast<const float4*>(p) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// logical workgroup id -> (tile_m, tile_n) of the distance pass; false = this id has no tile (it exits at once).
// Tiles are walked in super-blocks of 8 x 8: the 64 tiles of a super-block are consecutive ids, so they run together
// on one XCD (xcd_remap hands every XCD a contiguous id range) and share 8 A and 8 B operand tiles through its L2.
// Walked row by row, every tile pulled its B operand from beyond the L2 (rocprofv3: 1 GB of fetches for a 17 MB operand).
//   SYM: the super-blocks on or above the diagonal, row by row (row sm holds (sm, sm..S-1)); inside a diagonal
//        super-block only tiles with tile_n >= tile_m exist
//   else: all super-blocks, row by row
constexpr int DSB = 8;
__host__ __device__ __forceinline__ long distance_grid(bool sym, int tiles_m, int tiles_n) {
  const long sm = (tiles_m + DSB - 1) / DSB, sn = (tiles_n + DSB - 1) / DSB;
  return (sym ? sn * (sn + 1) / 2 : sm * sn) * (DSB * DSB);
}
template <bool SYM>
__device__ __forceinline__ bool distance_tile(int logical, int tiles_m, int tiles_n, int& tile_m, int& tile_n) {
  const int sb = logical / (DSB * DSB), in = logical % (DSB * DSB);
  int sbm, sbn;
  if (SYM) {
    const int t_ = (tiles_n + DSB - 1) / DSB;
    int tm = (int)(((2.0 * t_ + 1.0) - sqrt((2.0 * t_ + 1.0) * (2.0 * t_ + 1.0) - 8.0 * (double)sb)) * 0.5);
    tm = max(0, min(tm, t_ - 1));
    while (tm > 0 && (long)tm * t_ - (long)tm * (tm - 1) / 2 > sb) --tm;
    while ((long)(tm + 1) * t_ - (long)(tm + 1) * tm / 2 <= sb) ++tm;
    sbm = tm;
    sbn = tm + (sb - (int)((long)tm * t_ - (long)tm * (tm - 1) / 2));
  } else {
    const int sn = (tiles_n + DSB - 1) / DSB;
    sbm = sb / sn;
    sbn = sb % sn;
  }
  tile_m = sbm * DSB + in / DSB;
  tile_n = sbn * DSB + in % DSB;
  return tile_m < tiles_m && tile_n < tiles_n && (!SYM || tile_n >= tile_m);
}

// ---- radix select: keys and state -----------------------------------------------------------------
struct SelState {
  u64 rank[2];     // remaining 0-based ascending rank of each target inside its current prefix
  u32 prefix[2];   // key bits fixed so far
  u32 diverged;    // the two targets no longer share a prefix
  u32 even;        // n*n even -> median is the mean of the two targets
  float median;
  float h2;
  float lo, hi;    // the two order statistics
  u32 pad[4];
};
static_assert(sizeof(SelState) == 64, "SelState must stay 64 bytes");

// Speculative median window (the fused single-rank call; the staged stein_spec_* calls for several ranks).  SVGD
// moves the particles a little per step, so the median of the n^2 distances drifts smoothly.  The distance
// epilogue counts the entries below a narrow key window around the median extrapolated from the two previous steps
// and appends the entries inside it to a buffer; if both median targets turn out to lie inside the window, an exact
// weighted selection over that buffer (k_spec_select; k_spec_tally + k_spec_pick across ranks) replaces the
// radix-select passes over D.  Otherwise (first steps, jump, overflow) the passes run as before: exact either way.
// Lives right after SelState in the SELECT section and persists across steps (garbage until `magic` is set).
struct SpecState {
  u32 magic;      // SPEC_MAGIC1 once one median has been recorded, SPEC_MAGIC2 once two have (velocity known)
  u32 center;     // predicted key of this step's lower median target: the VALUE is extrapolated linearly from the two
                  // previous medians (key space bends at every power of two), then mapped to its key
  u32 halfwidth;  // half-width of the next window, in key units: follows the last prediction error
  u32 lo_key;     // this step's window [lo_key, lo_key + width]; lo_key = 0xffffffff, width = 0: no window
  u32 width;
  u32 count;      // entries appended this step (may exceed the buffer capacity -> overflow -> miss); k_spec_pick
                  // replaces it by the sum over the ranks
  u32 overflow;   // a workgroup's LDS queue overflowed
  u32 hit;        // this step's median came from the window; the radix-select passes skip themselves
  u32 earned_hw;  // the last half-width that came from a measured prediction error (0: none yet): the floor of the next one
  u32 reserved;   // (the weight below the window is summed from the slots at the head of the buffer)
  u64 total;      // n * n
  u32 last_key;   // key of the previous step's lower median target
  u32 skip_l0;    // the level-0 histogram pass over D is not needed (taken in the distance epilogue, or the window hit)
  u32 n_steps;    // medians recorded since the predictor started (spec_update_dev) ...
  u32 n_hits;     // ... and how many of them came from the window: the hit rate a harness reports (bench.py)
};
static_assert(sizeof(SpecState) == 64, "SpecState must stay 64 bytes");
// Third 64-byte block of the SELECT section (fused call only): "last workgroup out" tickets that let a kernel's last
// workgroup do what used to be a one-workgroup follow-up launch (scales after the column maxima, the final resolve after
// the level-2 histogram, the |phi|^2 sum after k_phi_finish's partials when that kernel has at most 512 workgroups -- with
// 1024 of them waiting on their stores the ticket cost more than the launch it saved).  Zeroed by the fused call's first
// kernel; each ticket also resets itself.
struct FuseState {
  u32 done_colmax, done_finish;
  u32 gave_up;     // k_hist_all: a bounded wait ran out (cannot happen by construction; the bandwidth becomes NaN and the host is told)
  u32 pad[13];
};
static_assert(sizeof(FuseState) == 64, "FuseState must stay 64 bytes");
// Synchronisation state of k_hist_all (the fused call's chained radix select in one launch, steinhip.hip): lives in the
// rank-summed window table of the SPEC section, which a single-rank fused call has no other use for; zeroed by the fused
// call's first kernel.  Laid out so that NO address is hit by more than a few dozen workgroups: a device-scope atomic that
// returns its value costs ~20 ns when two thousand workgroups aim it at one address (they are performed one after the
// other at the memory side), which made round 4's first form -- one draw counter, one arrival counter, one flag -- spend
// as long on its barriers as on its passes at C3 (470 us for levels 1 + 2 against 240 us as two launches).
constexpr int HS_NV = 2048;       // virtual workgroups per level, at most
constexpr int HS_CLASSES = 64;    // virtual workgroup v reports to / listens on class v % 64
struct HistSync {
  u32 claim[STEIN_HIST_LEVELS][HS_NV];     // virtual workgroup v of a level has been taken (by its owner, or by a thief)
  struct Leaf { u32 done; u32 pad[15]; } leaf[STEIN_HIST_LEVELS][HS_CLASSES];   // finished virtual workgroups of a class
  struct Top { u32 classes; u32 pad[15]; } top[STEIN_HIST_LEVELS];               // classes that are complete
  struct Line { u32 gen; u32 pub[6]; u32 pad[9]; } line[HS_CLASSES];   // per class: levels published so far + the select
                                                                       // state the resolver published (prefix[2], rank[2] as halves)
  // k_phi_finish (fused call): the same two-level completion count lets its last workgroup sum the |phi|^2 partials
  // whatever the grid (round 3's single ticket was only worth it up to 512 workgroups)
  Leaf fin_leaf[HS_CLASSES];
  Top fin_top;
  // k_colmax (fused call, fp32 inputs): its last workgroup turns the column maxima into the scales
  Leaf cm_leaf[HS_CLASSES];
  Top cm_top;
};
// thread 0 of a workgroup whose results have been acknowledged: is this the last of `nblocks` workgroups to report?  A
// two-level tree (64 leaves, one top): no counter sees more than nblocks / 64 returning atomics.  Counters zero at launch.
__device__ __forceinline__ bool tree_report_done(HistSync::Leaf* leaf, HistSync::Top* top, u32 id, u32 nblocks) {
  const u32 c = id % HS_CLASSES;
  const u32 quota = (nblocks - c + HS_CLASSES - 1) / HS_CLASSES;
  if (__hip_atomic_fetch_add(&leaf[c].done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u != quota) return false;
  const u32 nclasses = nblocks < (u32)HS_CLASSES ? nblocks : (u32)HS_CLASSES;
  return __hip_atomic_fetch_add(&top->classes, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u == nclasses;
}
static_assert(sizeof(HistSync) % 8 == 0, "HistSync is carved out of a u64 table");
constexpr u32 SPEC_MAGIC1 = 0x5EED0001u, SPEC_MAGIC2 = 0x5EED0002u;
constexpr u32 SPEC_QCAP = 1016;          // per-workgroup LDS queue (entries of 8 bytes; shares its LDS with the histogram)
constexpr u32 SPEC_CAP = (1u << 21) - 2048u;   // global buffer capacity (entries); the 16 MB section starts with
constexpr u32 SPEC_SLOTS = 256;                 // 256 "below" counters, one per 64-byte line (8 u64 apart): every
                                                // workgroup adding to ONE address serialised at the L2 (+165 us)
constexpr u32 SPEC_HW_MAX = 32767;       // window <= 65535 keys: two 8-bit selection passes
// Several ranks (staged calls): every rank tallies its window entries into one counter per key of the window,
// the tables are summed over the ranks and every rank picks the targets from the sum (one all-reduce of
// SPEC_TABLE u64).  Behind the entry buffer: [0] weight below the window, [1] invalid (no window / overflow),
// [2] entries, [8 + k] weight of key lo_key + k.
constexpr u32 SPEC_TABLE_HDR = 8;
constexpr u32 SPEC_TABLE = SPEC_TABLE_HDR + 2 * SPEC_HW_MAX + 2;   // 65544 u64
static_assert(SPEC_SLOTS * 8 + SPEC_CAP == (1u << 21), "the table starts 2^21 words into the SPEC section (stein_amd/_lib.py)");

__device__ __forceinline__ u32 f32_key(float x) {  // monotone: a < b  <=>  key(a) < key(b)
  const u32 u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_f32(u32 k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// "Last workgroup out".  Call once per workgroup, by every thread, after the workgroup's results have been issued: true
// in exactly one workgroup, the last to arrive.  The 8 XCDs have private L2s, so a device-wide fence per workgroup would
// write back and invalidate a whole L2 every time (measured: 3x the kernel time).  Instead the results that cross
// workgroups must be WRITTEN with device-scope atomics (atomicAdd / atomicMax) and READ with load_fresh --
// those go to the coherent level themselves -- and only the order matters here: every wave waits until its outstanding
// memory operations have been acknowledged (s_waitcnt vmcnt(0); a workgroup barrier alone does not) before thread 0
// draws the ticket.  The counter is left at zero.
__device__ __forceinline__ bool last_workgroup_out(u32* counter, u32 nblocks) {
  __shared__ u32 s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    const u32 last = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1u ? 1u : 0u;
    if (last) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = last;
  }
  __syncthreads();
  return s_last != 0u;
}
__device__ __forceinline__ u64 load_fresh(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 load_fresh(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Add `valid` lanes' digits to an LDS histogram.  Distances cluster (a handful of top-level bins hold
// everything), so the wave first merges lanes that share the leader's digit into one atomic, twice,
// and only the stragglers fall back to per-lane atomics.
// `w` is the (wave-uniform) weight of every valid lane: 2 when an upper-triangle entry also stands for its mirror.
__device__ __forceinline__ void hist_add(u32* h, u32 digit, bool valid, int lane, u32 w = 1u) {
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const u64 act = __ballot(valid);
    if (act == 0ull) return;
    const int leader = __ffsll((long long)act) - 1;
    const u32 ld = (u32)__builtin_amdgcn_readlane((int)digit, leader);
    const bool same = valid && digit == ld;
    const u64 m = __ballot(same);
    if (lane == leader) atomicAdd(&h[ld], (u32)__popcll(m) * w);
    valid = valid && !same;
  }
  if (valid) atomicAdd(&h[digit], w);
}

// ---- epilogue of the distance pass ---------------------------------------------------------------------
// acc[i][j] are the four 32x32 MFMA accumulators of this wave's 64x64 sub-tile of S = T T^T
// (C/D map: col = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)).  Writes D = r_i + r_j - 2 S, the
// mirrored copy of an off-diagonal tile when SYM, and (hist0 != NULL) the level-0 radix-select histogram.
//
// SYM: entries strictly below the diagonal are never taken from the accumulators -- every entry above it is stored
// twice (itself and its mirror image) and counted twice, so D is exactly symmetric whatever the product order
// inside the MFMA chain was.  Off-diagonal tiles mirror with 16-byte stores, diagonal tiles element by element.
//
// Level-0 counting: the 64 values of a thread fall into a few neighbouring digits, so each thread keeps eight
// 8-bit counters (one 64-bit register, <= 128 per slot) for the window [base, base+7] around its wave's first digit
// and touches the LDS histogram once per window slot at the end; values outside the window go straight to LDS.
//
// The body is instantiated three times -- interior tile (no predicates), edge tile, diagonal tile -- and the
// workgroup picks one with a single uniform branch.
// LDS map of the distance epilogue (bytes from the start of the kernel's shared array)
constexpr int EPI_STAGE_LD = 132;                             // floats per staged row: 128 columns + 4 pad
constexpr int EPI_STAGE_BYTES = 64 * EPI_STAGE_LD * 4;        // 33792: half tile [64 rows][128 columns]
constexpr int EPI_LDS_BYTES = EPI_STAGE_BYTES + 8192 + 64 + 1024;   // + histogram OR window queue, counters, norms: 43072

// 16-byte store of four consecutive distances.  (Streaming `nt` stores were measured: on the mirror stores, whose 32-byte
// pieces rely on the L2 to merge into lines, the kernel took twice as long; on the row stores alone -1 %, within noise.)
__device__ __forceinline__ void store_d4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

// the staged half tile (phase p: rows wy*64 + p*32 + 0..31 of the tile, staged as row wy*32 + r) -> D, 16 bytes per lane
__device__ __forceinline__ void distance_store_rows(const float* __restrict__ stage, float* __restrict__ D, long ntc,
                                                    int tile_m, int brow0, int p) {
  const int t = threadIdx.x, f4 = t & 31;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int sr = (t >> 5) + 8 * k;
    const int row = tile_m * BM + (sr >> 5) * 64 + p * 32 + (sr & 31);
    const float4 v4 = *reinterpret_cast<const float4*>(stage + sr * EPI_STAGE_LD + 4 * f4);
    store_d4(D + d_index(row, brow0 + 4 * f4, ntc), v4.x, v4.y, v4.z, v4.w);
  }
}

struct SpecCtx {   // per-thread view of the speculative window (SPEC epilogues only)
  u32 lo, width;   // window: raw bit patterns [lo, lo + width] of positive floats (key = bits | 0x80000000)
  u32* qcnt;       // LDS: entries pushed by this workgroup
  u64* q;          // LDS queue, SPEC_QCAP entries of (key << 2 | weight)
  u32 below;       // this thread's weight of entries below the window
};

// MIRROR: an entry above the diagonal also stands for its mirror image (weight 2 in every count); STOREM: the mirror
// image is also STORED (the fp32-MFMA path: D is then a full symmetric image).  The split path stores only the tiles on
// and above the diagonal (STOREM = false): its contraction reads the others transposed (stein_x3.hip), which saves half
// of the distance pass's 4 n^2 bytes of stores -- the pass is bound by them.
template <bool MIRROR, bool PRED, bool DIAG, bool HIST, bool SPEC, bool STOREM>
__device__ __forceinline__ void distance_epilogue_body(const f32x16 (&acc)[2][2], u32* hl, float* __restrict__ D, int n,
                                                       int n_local, long ldD, int tile_m, int brow0, u32 base,
                                                       u64& packed, float two_s,
                                                       SpecCtx& sx, const float* rr, const float* rc, int i,
                                                       float* __restrict__ stage) {
  // One call handles acc[i][*]: rows wy*64 + i*32 .. +31 of the tile, the wave's 64 columns.
  // rr / rc: LDS copies of the tile's 128 row norms and 128 column norms (0 outside the matrix).  Read from global
  // memory here, the 66 dependent loads per thread were most of the epilogue's 35 k cycles.
  // Interior tiles (no predicates) do not store D from here: every lane holds 4 consecutive ROWS of one column, so
  // the tile would go out as 64 dword stores per thread, and store issue bounds the kernel (a non-symmetric row
  // block wrote D at 1.6 TB/s).  They stage the half tile in LDS ([64 rows][EPI_STAGE_LD]) and
  // distance_store_rows writes it with 16-byte stores along the rows.
  constexpr bool STAGED = !PRED && !DIAG;
  const long ntc = ldD >> 5;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wy = wid >> 1, wx = wid & 1;
  const int l31 = lane & 31, h4 = (lane >> 5) * 4;
  u32 below_cnt = 0u;   // SPEC, uniform weight: entries below the window, weighted once at the end
  {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = brow0 + wx * 64 + j * 32 + l31;
      const bool cok = !PRED || col < n;
      const float rj = rc[wx * 64 + j * 32 + l31];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int lrow4 = tile_m * BM + wy * 64 + i * 32 + 8 * g + h4;  // first of 4 consecutive rows (e & 3)
        const float4 ri4 = *reinterpret_cast<const float4*>(rr + wy * 64 + i * 32 + 8 * g + h4);
        const float ris[4] = {ri4.x, ri4.y, ri4.z, ri4.w};
        // Four entries at a time, counted without a branch per entry: a workgroup is one wave per SIMD, so the chain
        // value -> key -> compare -> branch of every single entry ran unhidden (150 cycles per entry, twice the rest of
        // the epilogue).  The rare cases (a key inside the window, a digit outside the packed counters) are collected
        // into one flag per group and handled behind one branch.
        float v[4];
        u32 wq[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int lrow = lrow4 + q;
          const bool rok = !PRED || lrow < n_local;
          v[q] = (ris[q] + rj) - two_s * acc[i][j][4 * g + q];
          u32 w = MIRROR ? 2u : 1u;
          if (DIAG) w = lrow < col ? 2u : (lrow == col ? 1u : 0u);
          if (PRED && !(cok && rok)) w = 0u;
          wq[q] = w;
          if ((!PRED && !DIAG) || w) {
            if (STAGED) stage[(wy * 32 + 8 * g + h4 + q) * EPI_STAGE_LD + wx * 64 + j * 32 + l31] = v[q];
            else D[d_index(lrow, col, ntc)] = v[q];
          }
        }
        if (DIAG) {
          // mirror images: a group whose four rows all lie above the diagonal goes out as one 16-byte store (4
          // consecutive columns of row `col`, as in the off-diagonal tiles); only the groups that straddle the
          // diagonal fall back to single entries.  (One scattered dword store per entry made the diagonal tiles 2.5x
          // as slow as the others, and they are the critical path of every launch that fits the chip in one round.)
          if (wq[0] == 2u && wq[3] == 2u) {
            *reinterpret_cast<float4*>(D + d_index(col, lrow4, ntc)) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (wq[q] == 2u) D[d_index(col, lrow4 + q, ntc)] = v[q];
          }
        }
        if (HIST) {
          u32 dg[4];
          bool far = false;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            dg[q] = f32_key(v[q]) >> 21;
            const u32 off = dg[q] - base;
            const bool near = off < 8u;
            packed += near ? (u64)wq[q] << (8u * (off & 7u)) : 0ull;
            far |= !near && wq[q] != 0u;
          }
          if (__builtin_expect(far, 0)) {   // cold: laid out behind the hot path, which then falls through its branch
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (dg[q] - base >= 8u && wq[q]) atomicAdd(&hl[dg[q]], wq[q]);
          }
        }
        if (SPEC) {
          // The window lies among the keys of positive floats (median_init_body grants no other), where the key is
          // the bit pattern with the sign bit set: "below the window" is a signed compare of the raw bits (negative
          // values included) and "inside" an unsigned range test on them -- no key is formed unless an entry is caught.
          u32 off[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const u32 raw = __float_as_uint(v[q]);
            const bool lower = (int)raw < (int)sx.lo;
            if (!PRED && !DIAG) below_cnt += lower ? 1u : 0u;
            else sx.below += lower ? wq[q] : 0u;
            off[q] = raw - sx.lo;
            if ((PRED || DIAG) && wq[q] == 0u) off[q] = 0xffffffffu;
          }
          if (__builtin_expect(min(min(off[0], off[1]), min(off[2], off[3])) <= sx.width, 0)) {   // rare: a fraction of a percent of the entries
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (off[q] <= sx.width) {
                const u32 slot = atomicAdd(sx.qcnt, 1u);
                if (slot < SPEC_QCAP) sx.q[slot] = ((u64)(__float_as_uint(v[q]) | 0x80000000u) << 2) | wq[q];
              }
          }
        }
        // transposed copy: this lane's 4 rows are 4 consecutive columns of row `col` (lrow4 % 4 == 0: they stay inside
        // one 32-column tile, 16-byte aligned).  Entries past n land in padding, which no later stage reads.
        if (MIRROR && STOREM && cok && (!PRED || lrow4 < n_local))
          store_d4(D + d_index(col, lrow4, ntc), v[0], v[1], v[2], v[3]);
      }
    }
  }
  if (SPEC && !PRED && !DIAG) sx.below += below_cnt * (MIRROR ? 2u : 1u);
}

// What the epilogue needs from global memory, fetched BEFORE the main loop so that the latency (two dependent trips:
// about 2 us per workgroup, 0.04 ms of the C3 launch) hides under it: this thread's entry of the tile's row / column
// norms and this step's window.
struct EpiPrefetch {
  float norm;          // thread t < 128: r of tile row t, else r of tile column t - 128 (0 outside the matrix)
  u32 lo_key, width;   // SpecState window (width 0: none)
};
__device__ __forceinline__ EpiPrefetch distance_epilogue_prefetch(const float* __restrict__ r, int n, int row0,
                                                                  int n_local, int tile_m, int tile_n,
                                                                  const SpecState* __restrict__ spec) {
  EpiPrefetch pf;
  const int t = threadIdx.x;
  const int idx = t < BM ? tile_m * BM + t : tile_n * BN + (t - BM);
  const bool ok = t < BM ? idx < n_local : idx < n;
  pf.norm = ok ? r[(t < BM ? row0 : 0) + idx] : 0.f;
  pf.lo_key = spec ? spec->lo_key : 0xffffffffu;
  pf.width = spec ? spec->width : 0u;
  return pf;
}

// `lds` is the kernel's shared array, EPI_LDS_BYTES at least, which the caller no longer needs (all waves must be past
// their last LDS read): half-tile staging, 8 KB histogram, counters, window queue, row / column norms.
// spec != NULL needs hist0 != NULL.
template <bool SYM, bool STORE_MIRROR = SYM>
__device__ __forceinline__ void distance_epilogue(const f32x16 (&acc)[2][2], u32* lds, float* __restrict__ D, int n,
                                                  int n_local, long ldD, int tile_m, int tile_n,
                                                  u64* __restrict__ hist0,
                                                  const EpiPrefetch& pf, float two_s = 2.f,
                                                  SpecState* __restrict__ spec = nullptr,
                                                  u64* __restrict__ spec_buf = nullptr) {
  // two_s: D = r_i + r_j - two_s * acc.  2 for S = T T^T accumulated at full scale; the split kernels accumulate the
  // product of operands pre-scaled by a power of two and pass 2 / scale^2 (exact either way)
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wy = wid >> 1, wx = wid & 1;
  const int brow0 = tile_n * BN;
  float* stage = reinterpret_cast<float*>(lds);
  u32* hl = lds + EPI_STAGE_BYTES / 4;
  u64 packed = 0ull;
  u32 base = 0u;
  SpecCtx sx;
  sx.lo = 0xffffffffu; sx.width = 0u; sx.below = 0u;
  sx.qcnt = hl + STEIN_HIST_BINS;                               // 16 u32 reserved
  sx.q = reinterpret_cast<u64*>(hl);                            // a step has a window (queue) or a histogram, never both
  static_assert(8 * SPEC_QCAP <= 4 * STEIN_HIST_BINS, "the window queue lives in the histogram's LDS");
  float* rr = reinterpret_cast<float*>(hl + STEIN_HIST_BINS + 16);   // [128] row norms, [128] column norms
  float* rc = rr + BM;
  rr[t] = pf.norm;
  // A step with a window (spec->width != 0) skips the level-0 histogram: it is only needed when the window misses,
  // and then a k_hist<0> pass over D supplies it (SpecState::skip_l0).
  const bool window = spec && pf.width != 0u;
  const bool hist = hist0 && !window;
  if (window) {
    if (t == 0) { sx.qcnt[0] = 0u; sx.qcnt[2] = 0u; }
    sx.lo = pf.lo_key & 0x7fffffffu;   // raw bits of the window's first value (a positive float)
    sx.width = pf.width;
  }
  if (hist) {
    for (int b = t; b < STEIN_HIST_BINS; b += NTHREADS) hl[b] = 0u;
    __syncthreads();
    const float v0 = (rr[wy * 64] + rc[wx * 64]) - two_s * acc[0][0][0];   // any value will do: it only centres the packed counters
    base = (u32)__builtin_amdgcn_readfirstlane((int)(f32_key(v0) >> 21));
    base = base < 3u ? 0u : base - 3u;  // window start
  } else {
    __syncthreads();
  }
  const bool diag = SYM && tile_m == tile_n;
  const bool edge = brow0 + BN > n || tile_m * BM + BM > n_local;
#define STEIN_EPI_I(MIRROR, PRED, DIAG, I)                                                                             \
  do {                                                                                                                 \
    if (window)                                                                                                        \
      distance_epilogue_body<MIRROR, PRED, DIAG, false, true, STORE_MIRROR>(acc, hl, D, n, n_local, ldD, tile_m, brow0,             \
                                                              base, packed, two_s, sx, rr, rc, I, stage);             \
    else if (hist)                                                                                                     \
      distance_epilogue_body<MIRROR, PRED, DIAG, true, false, STORE_MIRROR>(acc, hl, D, n, n_local, ldD, tile_m, brow0,             \
                                                              base, packed, two_s, sx, rr, rc, I, stage);             \
    else                                                                                                               \
      distance_epilogue_body<MIRROR, PRED, DIAG, false, false, STORE_MIRROR>(acc, hl, D, n, n_local, ldD, tile_m, brow0,            \
                                                               base, packed, two_s, sx, rr, rc, I, stage);            \
  } while (0)
  const long ntc = ldD >> 5;
  if (diag) {
    STEIN_EPI_I(false, true, true, 0);
    STEIN_EPI_I(false, true, true, 1);
  } else if (edge) {
    STEIN_EPI_I(SYM, true, false, 0);
    STEIN_EPI_I(SYM, true, false, 1);
  } else {   // interior: the two row halves go through the LDS staging
    STEIN_EPI_I(SYM, false, false, 0);
    __syncthreads();
    distance_store_rows(stage, D, ntc, tile_m, brow0, 0);
    __syncthreads();
    STEIN_EPI_I(SYM, false, false, 1);
    __syncthreads();
    distance_store_rows(stage, D, ntc, tile_m, brow0, 1);
  }
#undef STEIN_EPI_I
  if (hist) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      u32 c = (u32)(packed >> (8 * k)) & 255u;  // wave-sum of slot k, then one LDS atomic per wave
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
      if (lane == 0 && c) atomicAdd(&hl[base + k], c);
    }
    __syncthreads();
    for (int b = t; b < STEIN_HIST_BINS; b += NTHREADS)
      if (hl[b]) atomicAdd(&hist0[b], (u64)hl[b]);
  }
  if (window) {
    // weight below the window: wave sum -> LDS -> one global atomic per workgroup, spread over slots
    u32 b = sx.below;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b += __shfl_xor(b, o);
    if (lane == 0 && b) atomicAdd(&sx.qcnt[2], b);
    __syncthreads();
    if (t == 0 && sx.qcnt[2])
      atomicAdd(reinterpret_cast<unsigned long long*>(spec_buf + (blockIdx.x % SPEC_SLOTS) * 8),
                (unsigned long long)sx.qcnt[2]);
    // flush the workgroup's queue to the global buffer
    const u32 pushed = sx.qcnt[0];
    const u32 nq = min(pushed, SPEC_QCAP);
    if (nq) {
      __syncthreads();
      if (t == 0) {
        if (pushed > SPEC_QCAP) spec->overflow = 1u;
        sx.qcnt[1] = atomicAdd(&spec->count, nq);
      }
      __syncthreads();
      const u32 gbase = sx.qcnt[1];
      for (u32 i = t; i < nq; i += NTHREADS)
        if (gbase + i < SPEC_CAP) spec_buf[SPEC_SLOTS * 8 + gbase + i] = sx.q[i];
    }
  }
}

// ---- epilogue of the contraction: partial O tile + partial rowsum ----------------------------------
// rs[p] is this thread's running sum of P over the rows lr + 32p it staged (8 consecutive lanes share a row).
__device__ __forceinline__ void phi_epilogue(const f32x16 (&acc)[2][2], const float (&rs)[4], float* __restrict__ Oz,
                                             float* __restrict__ RSz, int d, int n_local, int i0, int c0,
                                             bool write_rowsum) {
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wy = wid >> 1, wx = wid & 1;
  const int l31 = lane & 31, h4 = (lane >> 5) * 4;
  const int lr = t >> 3;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = c0 + wx * 64 + j * 32 + l31;
      if (col >= d) continue;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = i0 + wy * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + h4;
        if (row < n_local) Oz[(size_t)row * d + col] = acc[i][j][e];
      }
    }
  }
  if (write_rowsum) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float s = rs[p];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      s += __shfl_xor(s, 4);
      const int row = i0 + lr + 32 * p;
      if ((t & 7) == 0 && row < n_local) RSz[row] = s;
    }
  }
}

// ---- host-side layout (shared by both translation units) -----------------------------------------------
struct SteinLayout {
  size_t off[STEIN_WS_NSECTIONS];
  size_t total;
  int64_t ld_dist, split, sq_blocks, jchunk, tiles_m, cblocks;
  int64_t phi_wide;   // split-precision contraction: 1 = 64-row x 512-column workgroups (k_phi_x3fs<NP, 4>), 0 = 128 x 256
  // split-precision planes (flags & STEIN_FLAG_X3): 16-bit [3 slots][rows][k] each, tile-major
  int64_t x3_rows, x3_dk;   // row-major theta planes: x3_rows x x3_dk  (distance operands)
  int64_t x3_dc, x3_nk;     // transposed planes of theta and of the score: x3_dc x x3_nk  (contraction B operand)
  size_t x3_t3, x3_tt3, x3_gt3;  // byte offsets inside the PLANES section
  size_t x3_sc;                  // power-of-two operand scales (stein_x3.hip: "scales area"), inside the PLANES section
};
int stein_make_layout(int64_t n_local, int64_t n, int64_t d, int dtype, int flags, SteinLayout* L);

// stein_small.hip: the whole phi computation in one kernel for n <= 160 (the reference's own example sizes)
bool stein_small_ok(int64_t n, int64_t d, int dtype);
int stein_small_phi(const float* theta, const float* score, int64_t n, int64_t d, float* phi, float* h2_out,
                    double* sqpart /* one partial |phi|^2 per workgroup, *nparts of them (<= ceil(d / 32)) */,
                    float* K_out, float* dK_out, int* nparts /* 0: a single workgroup wrote *sqnorm_out itself */,
                    double* sqnorm_out, hipStream_t stream);

// ------------------------------------------------------------------------------------------------
// fused-call prologue (k_prologue, steinhip.hip; for bf16 inputs a slice of k_split's grid, stein_x3.hip): k_sel_init + the window set-up + zeroing of the histograms and of the "below" slots
// (gt of gn threads share the zeroing; thread 0 sets the states up)
__device__ __forceinline__ void median_init_body(int gt, int gn, SelState* st, SpecState* sp, u64 total,
                                                 u64* __restrict__ hist, u64* __restrict__ slots, int allow_window = 1) {
  for (int i = gt; i < STEIN_HIST_LEVELS * 2 * STEIN_HIST_BINS; i += gn) hist[i] = 0ull;
  for (int i = gt; i < (int)SPEC_SLOTS * 8; i += gn) slots[i] = 0ull;
  if (gt) return;
  {
    const u32 even = (total & 1ull) ? 0u : 1u;
    st->rank[0] = even ? total / 2 - 1 : total / 2;
    st->rank[1] = total / 2;
    st->prefix[0] = st->prefix[1] = 0u;
    st->diverged = 0u;
    st->even = even;
    st->median = st->h2 = st->lo = st->hi = 0.f;
  }
  // a window is granted only among the keys of positive finite floats, [0x80000000, 0xff000000): the distance epilogue
  // tests raw bit patterns (a median <= 0 means coincident particles: h2 = 0, nothing to speed up)
  if (allow_window && (sp->magic == SPEC_MAGIC1 || sp->magic == SPEC_MAGIC2) && sp->halfwidth <= SPEC_HW_MAX &&
      sp->center >= 0x80000000u + sp->halfwidth && sp->center < 0xff000000u - sp->halfwidth) {
    sp->lo_key = sp->center - sp->halfwidth;
    sp->width = 2u * sp->halfwidth;
  } else {
    sp->lo_key = 0xffffffffu;   // no window: nothing is inside, everything is "below" (and ignored)
    sp->width = 0u;
  }
  sp->count = 0u; sp->overflow = 0u; sp->hit = 0u;
  sp->skip_l0 = sp->width == 0u ? 1u : 0u;   // no window: the distance epilogue takes the level-0 histogram itself
  sp->reserved = 0ull; sp->total = total;
}


// What the fused call's first launch sets up besides the split planes: the row norms (one wave per row, the rows dealt
// round-robin over the waves of the launch's `nb` workgroups) and, shared by all of its threads, everything the later kernels
// expect to find zeroed or set up: the median state (median_init_body), the column maxima of the scales, the FuseState
// tickets and -- bf16 inputs -- the neutral operand scales.
struct PrologueArgs {
  int n, d;
  float* r;
  SelState* st; SpecState* sp; FuseState* fs;
  u64 total;
  u64* hist; u64* slots;
  u32* cmax; int ncmax;
  int allow_window;
  float* neutral_sc; int dc;
  u32* hsync; int hsync_words;   // HistSync of k_hist_all (zeroed here)
};
constexpr int PRO_INIT_BLOCKS = 16;   // (k_prologue's grid: one workgroup per four rows + these; any grid works)
__device__ __forceinline__ float prologue_elem(const float* p) { return *p; }
__device__ __forceinline__ float prologue_elem(const unsigned short* p) { return __uint_as_float((u32)*p << 16); }   // bf16 bits
template <typename TIN>
__device__ __forceinline__ void prologue_body(const TIN* __restrict__ T, const PrologueArgs& a, int b, int nb) {
  const int gt = b * 256 + (int)threadIdx.x, gn = nb * 256;
  for (int i = gt; i < a.ncmax; i += gn) a.cmax[i] = 0u;
  if (a.neutral_sc) {   // bf16 inputs: every operand scale is 1 (what k_make_scales(enable = 0) writes), no launch for it
    for (int i = gt; i < 4 * a.dc; i += gn) a.neutral_sc[i] = 1.f;
    if (gt == 0) { a.neutral_sc[4 * a.dc] = 1.f; a.neutral_sc[4 * a.dc + 1] = 2.f; a.neutral_sc[4 * a.dc + 2] = 1.f; }
  }
  if (gt < 16) reinterpret_cast<u32*>(a.fs)[gt] = 0u;
  for (int i = gt; i < a.hsync_words; i += gn) a.hsync[i] = 0u;
  median_init_body(gt, gn, a.st, a.sp, a.total, a.hist, a.slots, a.allow_window);
  const int lane = threadIdx.x & 63;
  for (int row = gt >> 6; row < a.n; row += gn >> 6) {
    const TIN* rp = T + (size_t)row * a.d;
    float s = 0.f;
    for (int k = lane; k < a.d; k += 64) {
      const float x = prologue_elem(rp + k);
      s = fmaf(x, x, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) a.r[row] = s;
  }
}
