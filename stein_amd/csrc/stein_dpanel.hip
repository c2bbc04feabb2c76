// stein_dpanel.hip -- the distance pass D = r_i + r_j - two_s (T T^T) for operand panels that fit in LDS
// (stein/kernels/abstract_kernel.py:33-35), split-precision operands (stein_x3.hip: T3 planes, fragment order).
//
// Why a second form.  k_distance_x3 gives every 128 x 128 tile to a 256-thread workgroup, three per CU.  Its phases
// (operand staging, MFMAs, epilogue) add up instead of overlapping: the co-resident workgroups fall into step
// (profiles/r02_distance_stamps.txt), and a tile with K = 256 is short -- the 64 KB a tile writes cost as much as its
// MFMAs.  Here ONE 512-thread workgroup per CU keeps the 128 rows x K operand panel of a row tile in LDS (<= 128 KB,
// loaded once per run of column tiles) and its eight waves never synchronise with each other in steady state:
//   * a wave owns a "strip" = 128 rows x 32 columns = exactly one [128][32] tile of the tile-major D image (16 KB);
//   * the strip's own operand (32 particles x K) never touches LDS: its fragments are streamed from L2 straight into
//     registers, DP_RING k tiles ahead, with hand-counted waits (asm loads, as in k_phi_x3fs);
//   * the panel is the OTHER operand of v_mfma_f32_16x16x32: lane = panel row, registers = 4 consecutive columns of the
//     strip, so an accumulator is already a 16-byte piece of a D row.  The wave turns 16 rows at a time into whole 128-byte
//     lines through a private 2.3 KB LDS buffer (no barrier: one wave's LDS operations execute in order) and writes
//     its strip as sixteen 1 KB stores;
//   * the two waves of a SIMD drift into opposite phases -- one wave's epilogue (VALU, LDS, stores) runs under the other
//     wave's MFMAs -- because nothing ties them together; loads and stores of a wave share one in-order counter on gfx950,
//     so the waits below count the stores that sit between a load and its use.
//
// Restrictions (the caller falls back to k_distance_x3): n, n_local and row0 multiples of 128; no level-0 histogram in the
// epilogue -- a step without a median window gets it from a k_hist<0> pass instead (the kernel clears SpecState::skip_l0
// itself).  NP * K / 32 <= 16 (the panel fits 128 KB: d <= 256 for fp32 inputs) for k_distance_panel; a longer K goes to
// k_distance_panel_deep (below), which keeps a chunk of the panel in LDS at a time.
//
// SYM (single rank): only strips on and above the diagonal exist; off-diagonal tiles count twice (weight 2) and are
// stored once (the contraction reads the others transposed).  The four strips of a diagonal tile store entry by entry:
// (i, j) for i <= j and, for i < j, the same VALUE at (j, i) -- D is exactly symmetric whatever the product order was.
#include <type_traits>
#include "stein_x3.h"
#include "stein_x3_dev.h"
#define STEIN_ABLATE_DPANEL
#include "stein_ablate.h"   // DP_STAMP* / DP_STORE16: hooks of the diagnostic builds (nothing in the shipped library)

#ifdef STEIN_DP_NOPRIO   // (A/B hook: the kernels without their issue priorities)
#define DP_SETPRIO(p) do {} while (0)
#else
#define DP_SETPRIO(p) __builtin_amdgcn_s_setprio(p)
#endif
constexpr int DP_THREADS = 512;        // 8 waves: two per SIMD
constexpr int DP_RING = 4;             // k tiles of the strip operand in flight per wave
constexpr int DP_PITCH = 144;          // bytes per staged row: 32 floats + 16 (conflict-free 16-byte writes down a column)
constexpr int DP_STAGE = 16 * DP_PITCH;
constexpr int DP_QCAP = 124;           // window entries a wave buffers before it flushes them (8 bytes each)
constexpr int DP_QBYTES = 1024;
constexpr int DP_PANEL = 16 * XPLANE;  // 128 KB: NP * ntk <= 16 planes of [128 rows][32 k]
constexpr int DP_LDS = DP_PANEL + 8 * DP_STAGE + 8 * DP_QBYTES;   // 157,696 of the 163,840 bytes

// Who works on what.  A segment = one row tile's run of column strips = one operand panel.
//   row block (not SYM): segment g = row tile g, strips [0, 4 tiles_n): all equally long.
//   SYM: the rows of the triangle are folded into "virtual rows" of equal length: virtual row v = row v (strips [4 v, 4 N)),
//        then row N - 1 - v (strips [4 (N - 1 - v), 4 N)): 4 N + 4 strips whatever v (the middle row of an odd N stands alone).
//        Segment 2 v is the long part, 2 v + 1 the short one.
// The (virtual) rows are dealt to the workgroups whole: with fewer rows than workgroups, G / rows workgroups ("members")
// share a row and take its strips INTERLEAVED (member m: strips m, m + M, m + 2 M, ...); with more rows than workgroups a
// workgroup takes rows p, p + G, ... one after the other.  Workgroups that share an XCD hold neighbouring rows (xcd_remap),
// so at any time all of an XCD's workgroups -- members of one row and of its neighbours alike -- sweep the same few column
// strips: one wave pulls a strip's operand into the XCD's L2 and up to 31 others find it there.  The kernel moves ~1 GB of
// operand fragments per launch at C3; with every workgroup on a contiguous QUARTER of its row (round 3's first form) 262 MB
// of that came from beyond the L2, interleaved 102 MB (rocprofv3 FETCH_SIZE) -- at the same launch time: the launch is
// bound by the clock the chip holds while it writes D, not by where the operand comes from.
// Inside a workgroup the strips are dealt to the waves one by one by a counter in LDS.
// (Tried and dropped, round 3: strips drawn from per-row counters in global memory by teams of four workgroups, with
// work stealing once a team had run dry: 0.206-0.219 ms at C3 against 0.192-0.204 for this form on like boxes.)
template <bool SYM>
__device__ __forceinline__ int dp_segments(int tiles_m, int tiles_n) { return SYM ? 2 * ((tiles_n + 1) / 2) : tiles_m; }
template <bool SYM>
__device__ __forceinline__ void dp_segment(int g, int tiles_n, int& row, int& len) {
  if (SYM) {
    const int v = g >> 1;
    row = (g & 1) ? tiles_n - 1 - v : v;
    len = ((g & 1) && row == v) ? 0 : 4 * (tiles_n - row);   // (odd N: the middle row has no partner)
  } else {
    row = g;
    len = 4 * tiles_n;
  }
}
__device__ __forceinline__ const void* dp_uniform(const void* p) {   // make a wave-uniform pointer provably so
  const unsigned long long v = (unsigned long long)p;
  return reinterpret_cast<const void*>(
      ((unsigned long long)(u32)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
      (unsigned long long)(u32)__builtin_amdgcn_readfirstlane((int)v));
}

// one k tile: 8 panel blocks (16 rows each, fragments from LDS) x the strip's two 16-column blocks
template <int NP>
__device__ __forceinline__ void dp_step(const unsigned char* pk /* panel k tile + 16 lane */, const u32x4 (&a)[2][3],
                                        f32x4 (&acc)[8][2]) {
  u32x4 b[2][3];
#pragma unroll
  for (int s = 0; s < NP; ++s) b[0][s] = *reinterpret_cast<const u32x4*>(pk + s * XPLANE);
#pragma unroll
  for (int ib = 0; ib < 8; ++ib) {
#ifdef STEIN_DP_ABL_NOPANEL   // (timing-only ablation: one LDS fragment read per k tile instead of eight)
    b[(ib + 1) & 1][0] = b[ib & 1][0]; b[(ib + 1) & 1][1] = b[ib & 1][1];
#else
    if (ib + 1 < 8) {
#pragma unroll
      for (int s = 0; s < NP; ++s) b[(ib + 1) & 1][s] = *reinterpret_cast<const u32x4*>(pk + s * XPLANE + (ib + 1) * 1024);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) acc[ib][jb] = x3_products16<NP>(a[jb], b[ib & 1], acc[ib][jb]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

struct DpWin {     // the speculative median window as a wave sees it
  u32 lo, width;   // raw bit patterns [lo, lo + width] of positive floats (width 0: no window)
  u32 below;       // weight of the entries below the window, this wave's strips so far (wave-uniform)
  u32 below_lane;  // ... and, PER LANE, the number of such entries in the regular strips a windowed step counts with dp_count8
                   // (each weighs `wt`; summed over the wave once, at the end of the kernel)
  u32 qn;          // entries in the wave's LDS queue (wave-uniform)
  u64* q;          // LDS, DP_QCAP entries of (key << 2 | weight)
  u32 over;        // the queue overflowed (-> SpecState::overflow: the window misses, the radix select runs)
};

// four values of one lane: count those below the window, queue those inside it (weights w[e], 0 = not an entry)
template <bool UNIT>   // UNIT: every weight is `wt` (wave-uniform)
__device__ __forceinline__ void dp_count4(DpWin& sx, const float (&v)[4], const u32 (&w)[4], u32 wt, int lane) {
  u32 off[4];
  u32 mn = 0xffffffffu;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const u32 raw = __float_as_uint(v[e]);
    const bool lower = (int)raw < (int)sx.lo;   // signed compare of the raw bits: negative values are below too
    if (UNIT) {
      sx.below += (u32)__popcll(__ballot(lower)) * wt;
    } else {
      sx.below += (u32)__popcll(__ballot(lower && w[e] == 2u)) * 2u + (u32)__popcll(__ballot(lower && w[e] == 1u));
    }
    off[e] = raw - sx.lo;
    if (!UNIT && w[e] == 0u) off[e] = 0xffffffffu;
    mn = min(mn, off[e]);
  }
  if (__builtin_expect(__ballot(mn <= sx.width) != 0ull, 0)) {   // rare: a fraction of a percent of the groups
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bool in = off[e] <= sx.width;
      const u64 m = __ballot(in);
      if (m) {
        const u32 idx = sx.qn + (u32)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
        if (in) {
          if (idx < (u32)DP_QCAP) sx.q[idx] = ((u64)(__float_as_uint(v[e]) | 0x80000000u) << 2) | (UNIT ? wt : w[e]);
        }
        sx.qn += (u32)__popcll(m);
      }
    }
    if (sx.qn > (u32)DP_QCAP) { sx.over = 1u; sx.qn = DP_QCAP; }
  }
}

// the eight values a lane holds of one 16-row block (two column blocks x four columns), every weight `wt`: the same
// counting with one min-tree and one branch for all eight, and the entries below the window counted PER LANE: a compare into
// vcc and an add-with-carry (two VALU instructions per entry, no scalar ones).  The ballot form -- s_bcnt1 of eight 64-bit
// masks per block -- left the compiler sixteen live SGPR pairs per block, which it parked in VGPR lanes: two v_writelane and
// two v_readlane per ENTRY, ~800 VALU instructions per strip where ~300 do (round 4, found in the assembly; the epilogue of
// one wave competes with the other wave's MFMAs for the SIMD's issue slots).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void dp_count8(DpWin& sx, const float (&v)[2][4], u32 wt, int lane) {
  u32 mn = 0xffffffffu;
  // (one statement: nothing for the compiler to pad or to reorder between a compare and its add)
  asm("v_cmp_gt_i32 vcc, %9, %1\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
      "v_cmp_gt_i32 vcc, %9, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
      "v_cmp_gt_i32 vcc, %9, %3\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
      "v_cmp_gt_i32 vcc, %9, %4\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
      "v_cmp_gt_i32 vcc, %9, %5\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
      "v_cmp_gt_i32 vcc, %9, %6\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
      "v_cmp_gt_i32 vcc, %9, %7\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
      "v_cmp_gt_i32 vcc, %9, %8\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc"   // signed compare of the raw bits: negative values are below too
      : "+v"(sx.below_lane)
      : "v"(v[0][0]), "v"(v[0][1]), "v"(v[0][2]), "v"(v[0][3]), "v"(v[1][0]), "v"(v[1][1]), "v"(v[1][2]), "v"(v[1][3]), "s"(sx.lo)
      : "vcc");
#pragma unroll
  for (int k = 0; k < 8; ++k) mn = min(mn, __float_as_uint(v[k >> 2][k & 3]) - sx.lo);
  if (__builtin_expect(__ballot(mn <= sx.width) != 0ull, 0)) {   // rare: a fraction of a percent of the blocks
#pragma unroll
    for (int k = 0; k < 8; ++k) {   // (the offsets are formed again here: eight registers less across the hot path)
      const u32 raw = __float_as_uint(v[k >> 2][k & 3]);
      const bool in = raw - sx.lo <= sx.width;
      const u64 m = __ballot(in);
      if (m) {
        const u32 idx = sx.qn + (u32)__builtin_amdgcn_mbcnt_hi((u32)(m >> 32), __builtin_amdgcn_mbcnt_lo((u32)m, 0u));
        if (in && idx < (u32)DP_QCAP) sx.q[idx] = ((u64)(raw | 0x80000000u) << 2) | wt;
        sx.qn += (u32)__popcll(m);
      }
    }
    if (sx.qn > (u32)DP_QCAP) { sx.over = 1u; sx.qn = DP_QCAP; }
  }
}
// what the wave counted below the window: the wave-uniform part plus `wt` times the sum of the lanes' counters
__device__ __forceinline__ u32 dp_below_total(const DpWin& sx, u32 wt) {
  u32 c = sx.below_lane;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  return sx.below + c * wt;
}

// the wave's queue -> the global entry buffer (irregular: the caller must not assume a store count afterwards)
__device__ __forceinline__ void dp_flush(DpWin& sx, SpecState* __restrict__ spec, u64* __restrict__ spec_buf, int lane) {
  if (sx.qn == 0u) return;
  u32 base = 0u;
  if (lane == 0) base = atomicAdd(&spec->count, sx.qn);
  base = (u32)__builtin_amdgcn_readfirstlane((int)base);
  for (u32 i = lane; i < sx.qn; i += 64u)
    if (base + i < SPEC_CAP) spec_buf[SPEC_SLOTS * 8 + base + i] = sx.q[i];
  sx.qn = 0u;
}

// a step without a window takes the level-0 histogram here (the LDS histogram occupies the window queues' 8 KB; the first
// __syncthreads of the segment loop -- or the one in front of the final flush -- orders the zeroing before the counting)
#define DP_HIST_SETUP                                                                                                   \
  const bool hist = hist0 != nullptr && !window;                                                                        \
  DpHist hx;                                                                                                            \
  hx.hl = reinterpret_cast<u32*>(smem + DP_PANEL + 8 * DP_STAGE);                                                       \
  hx.packed = 0ull; hx.base = 0u; hx.wt = SYM ? 2u : 1u; hx.nacc = 0u; hx.have = false;                                 \
  if (hist)                                                                                                             \
    for (int b = t; b < STEIN_HIST_BINS; b += DP_THREADS) hx.hl[b] = 0u
// the workgroup's histogram -> the global level-0 histogram (every thread of the workgroup gets here)
#define DP_HIST_FINISH                                                                                                  \
  if (hist) {                                                                                                           \
    dp_hist_flush(hx, lane);                                                                                            \
    __syncthreads();                                                                                                    \
    for (int b = t; b < STEIN_HIST_BINS; b += DP_THREADS)                                                               \
      if (hx.hl[b]) atomicAdd(reinterpret_cast<unsigned long long*>(&hist0[b]), (unsigned long long)hx.hl[b]);          \
  }
static_assert(8 * DP_QBYTES == 4 * STEIN_HIST_BINS, "the level-0 histogram lives in the window queues' LDS");

// ---- level-0 radix-select histogram from the epilogue (a step WITHOUT a median window: the first two steps of a run, the
// staged calls of a sharded run's radix form, STEIN_FLAG_NO_WINDOW) ------------------------------------------------------
// Round 3 left this to a k_hist<0> pass over D (0.15 ms at C3, a third of what a miss costs).  The workgroup's histogram
// lives in the 8 KB the window queues would occupy (a step has a window or a histogram, never both).  The distances of a
// strip fall into a few neighbouring level-0 digits (a digit is a quarter octave), so every lane counts into eight 8-bit
// slots of one 64-bit register for the digits [base, base + 8) around the first value its wave saw (nine VALU
// instructions per entry, no LDS traffic); the wave sums the slots every three strips (<= 192 per slot) and adds them to
// the LDS histogram with eight atomics.  A value outside the slots -- and every entry of a diagonal strip, whose weights
// differ per entry -- goes to the LDS histogram directly.  The workgroup adds its histogram to the global one at the end.
struct DpHist {
  u32* hl;       // LDS: STEIN_HIST_BINS counters of the workgroup
  u64 packed;    // this lane's eight 8-bit slots
  u32 base;      // first RAW digit (bits >> 21 of a positive float, 0 .. 1016) of the slots (wave-uniform)
  u32 wt;        // weight of an entry of a regular strip (2: it also stands for its mirror image)
  u32 nacc;      // regular strips counted into `packed` since the last flush (wave-uniform)
  bool have;     // base has been chosen
};
__device__ __forceinline__ u32 dp_key_digit(u32 rawdigit) {   // bits >> 21 -> level-0 digit of the order-preserving key
  return rawdigit >= 1024u ? 2047u - rawdigit : rawdigit + 1024u;
}
__device__ __forceinline__ void dp_hist4(DpHist& hx, const float (&v)[4]) {
  bool far = false;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const u32 off = (__float_as_uint(v[e]) >> 21) - hx.base;   // a negative value has rawdigit >= 1024 > base + 7: "far"
    const bool near = off < 8u;
    hx.packed += near ? 1ull << (8u * (off & 7u)) : 0ull;
    far |= !near;
  }
  if (__builtin_expect(far, 0)) {   // (the digits are formed again here: four registers less across the hot path)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32 dg = __float_as_uint(v[e]) >> 21;
      if (dg - hx.base >= 8u) atomicAdd(&hx.hl[dp_key_digit(dg)], hx.wt);
    }
  }
}
__device__ __forceinline__ void dp_hist_flush(DpHist& hx, int lane) {
  if (hx.nacc == 0u) return;
  u64 x0 = hx.packed & 0x00ff00ff00ff00ffull, x1 = (hx.packed >> 8) & 0x00ff00ff00ff00ffull;   // slots 0,2,4,6 / 1,3,5,7 as 16-bit fields
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { x0 += __shfl_xor(x0, o); x1 += __shfl_xor(x1, o); }        // <= 64 lanes x 192: fits 16 bits
  if (lane < 8) {
    const u32 c = (u32)(((lane & 1) ? x1 : x0) >> (16 * (lane >> 1))) & 0xffffu;
    if (c) atomicAdd(&hx.hl[1024u + hx.base + (u32)lane], c * hx.wt);
  }
  hx.packed = 0ull;
  hx.nacc = 0u;
}

template <bool SYM, int NP>
__global__ __launch_bounds__(DP_THREADS, 2) void k_distance_panel(const u16* __restrict__ T3, int ntk,
                                                                  const float* __restrict__ r, float* __restrict__ D,
                                                                  int row0, int tiles_m, int tiles_n, long ldD,
                                                                  const float* __restrict__ two_s,
                                                                  SpecState* __restrict__ spec, u64* __restrict__ spec_buf,
                                                                  u64* __restrict__ hist0) {
  constexpr int LPS = 2 * NP;                  // streamed loads per k tile: 2 column blocks x NP planes
  constexpr int DP_HIST_EVERY = 3;             // strips between two flushes of the level-0 slots
  constexpr int W_LATE = 3 * LPS;              // the loads of the three k tiles behind the one waited for stay in flight
  constexpr int W_EARLY = 3 * LPS + 16 + 2;    // ... and the 16 D stores and 2 norm loads issued between its request and its use
  __shared__ __attribute__((aligned(16))) unsigned char smem[DP_LDS + 16];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  unsigned char* const panel = smem;
  unsigned char* const stg = smem + DP_PANEL + w * DP_STAGE;
  const long ntc = ldD >> 5;
  const int groups = (ntk + DP_RING - 1) / DP_RING;
  const float nts = -*two_s;

  DpWin sx;
  sx.lo = 0xffffffffu; sx.width = 0u; sx.below = 0u; sx.below_lane = 0u; sx.qn = 0u; sx.over = 0u;
  sx.q = reinterpret_cast<u64*>(smem + DP_PANEL + 8 * DP_STAGE + w * DP_QBYTES);
  bool window = false;
  if (spec) {
    const u32 lk = spec->lo_key, wd = spec->width;
    window = wd != 0u;
    if (window) { sx.lo = lk & 0x7fffffffu; sx.width = wd; }
    // no window this step and nowhere to put the level-0 histogram: the k_hist<0> pass over D must run
    else if (!hist0 && blockIdx.x == 0 && t == 0) spec->skip_l0 = 0u;
  }
  DP_HIST_SETUP;

  // ---- this workgroup's share of the strip order -----------------------------------------------------------------------
  const int G = gridDim.x, p = xcd_remap(blockIdx.x, G);
  const int U = SYM ? (tiles_n + 1) / 2 : tiles_m;          // (virtual) rows
  int unit, member, members, unit_step;
  if (U >= G) { unit = p; member = 0; members = 1; unit_step = G; }
  else {   // M or M + 1 members per row, the rows with M + 1 first
    const int M = G / U, big = G - U * M;
    if (p < big * (M + 1)) { unit = p / (M + 1); member = p % (M + 1); members = M + 1; }
    else { const int q = p - big * (M + 1); unit = big + q / M; member = q % M; members = M; }
    unit_step = U;                                           // (one row per workgroup)
  }
  u32* const dealer = reinterpret_cast<u32*>(smem + DP_LDS);   // the next strip of the piece nobody has taken yet
  // per-lane constants
  const u32 aoff = (u32)lane * 16u;                 // fragment byte of this lane
  const u32 aoff1 = aoff + (u32)XPLANE;              // ... in the second plane
  const u32 roff = (u32)(lane >> 4) * 16u;          // norms of the 4 columns this lane holds in a 16-column block
  const int li = lane & 15, lq = lane >> 4;
  u32x4 ring[DP_RING][2][3];
  f32x4g rj[2];
  float ri[8];
#ifdef STEIN_DP_ABL_NOSTREAM
  int abl_requests = 0;
#endif
  DP_STAMP_DECL;

  for (; unit < U; unit += unit_step)
  for (int half = 0; half < (SYM ? 2 : 1); ++half) {
    int I, len;
    dp_segment<SYM>(SYM ? 2 * unit + half : unit, tiles_n, I, len);
    const int sb = 0, se = len > member ? (len - member + members - 1) / members : 0;   // this workgroup's strips: member + members k
    if (se <= sb) continue;
    const int cfirst = (SYM ? 4 * I : 0) + member; // column strip of the workgroup's strip k: cfirst + members k
    // ---- panel of row tile I -> LDS: wave w copies fragment w (1 KB) of every (k tile, plane) -------------------------
    __syncthreads();                               // everybody is done with the previous panel
    {
      // (all of a wave's pieces are requested before the first is written: one memory latency per panel, not sixteen)
      const u16* src = T3 + ((size_t)(row0 / 128 + I) * ntk * 3) * XTILE_E + w * 512 + lane * 8;
      // (no branches: planes past NP * ntk repeat the last one -- nobody reads them)
      u32x4 x[16];
      const int qlast = ntk * NP - 1;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int qq = q < qlast ? q : qlast;
        x[q] = *reinterpret_cast<const u32x4*>(src + ((size_t)(qq / NP) * 3 + qq % NP) * XTILE_E);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) *reinterpret_cast<u32x4*>(panel + q * XPLANE + w * 1024 + aoff) = x[q];
#pragma unroll
      for (int ib = 0; ib < 8; ++ib) ri[ib] = r[row0 + 128 * I + 16 * ib + li];
      // the compiler must wait for these loads HERE: left to their first use, its wait (which counts only the loads it
      // knows about) would sit in every strip's epilogue and drain the streamed loads in flight there
#pragma unroll
      for (int ib = 0; ib < 8; ++ib) asm volatile("" : "+v"(ri[ib]));
      if (t == 0) *dealer = (u32)(sb + 16);
    }
    __syncthreads();
    DP_STAMP(4);

    // ---- the wave's strips: sb + w and sb + 8 + w first, then whichever strip of the piece is next (a counter in LDS deals
    // them: at equal priority the older wave of a SIMD wins every issue arbitration, and dealt statically the younger one
    // was left to finish a quarter of its strips alone, with nothing to overlap its epilogues with).  A wave holds the strip
    // it works on and the next one, whose operand it prefetches.
    int s = sb + w, s1 = sb + 8 + w;
    if (s < se) {
      auto strip_base = [&](int strip) {           // operand fragments of column strip `strip` (k tile 0, plane 0, block 0)
        const long j0 = 32l * (cfirst + members * strip);
        return reinterpret_cast<const u16*>(dp_uniform(T3 + ((size_t)(j0 >> 7) * ntk * 3) * XTILE_E + ((j0 & 127) >> 4) * 512));
      };
      // The streamed memory operations are inline asm (neither counted nor waited for by the compiler, stein_x3_dev.h).  Each
      // statement opens with s_nop 4: the compiler may have produced the statement's scalar base address with a VALU
      // instruction a moment earlier (v_readfirstlane, or v_readlane when it reloads a spilled SGPR), and a vector-memory
      // instruction that reads such an SGPR within 5 wait states sees its OLD value -- which the compiler pads for its own
      // instructions but cannot for the inside of an asm statement (cdna_hip_programming.md 5.7 item 2; found the hard way:
      // a diagnostic build with more SGPR pressure sent its atomics to address 0).
      auto request = [&](const u16* sbase, int kt, u32x4 (&slot)[2][3]) {
        const u16* src = sbase + (size_t)(kt < ntk ? kt : ntk - 1) * 3 * XTILE_E;   // past the end: a harmless re-read keeps the counts
#ifdef STEIN_DP_ABL_NOSTREAM   // (timing-only ablation: the strip operand is fetched for the wave's first eight k tiles only;
        // after that the ring keeps those -- real -- values: MFMA power depends on the data, zeros would flatter it)
        if (abl_requests >= 8) {
          asm volatile("" : "+v"(slot[0][0]), "+v"(slot[0][1]), "+v"(slot[1][0]), "+v"(slot[1][1]) : "s"(src));
          return;
        }
        ++abl_requests;
#endif
        if constexpr (NP == 2) {
          asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %6\n\tglobal_load_dwordx4 %1, %5, %6\n\t"
                       "global_load_dwordx4 %2, %4, %6 offset:1024\n\tglobal_load_dwordx4 %3, %5, %6 offset:1024"
                       : "=&v"(slot[0][0]), "=&v"(slot[0][1]), "=&v"(slot[1][0]), "=&v"(slot[1][1])
                       : "v"(aoff), "v"(aoff1), "s"(src));
        } else {
          asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                       : "=&v"(slot[0][0]), "=&v"(slot[1][0]) : "v"(aoff), "s"(src));
        }
      };
      auto request_norms = [&](int strip) {
        const float* rb = reinterpret_cast<const float*>(dp_uniform(r + 32l * (cfirst + members * strip)));
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:64"
                     : "=&v"(rj[0]), "=&v"(rj[1]) : "v"(roff), "s"(rb) : "memory");
      };
      const u16* cur = strip_base(s);
      request_norms(s);
#pragma unroll
      for (int u = 0; u < DP_RING; ++u) request(cur, u, ring[u]);
      bool regular = false;                        // the previous epilogue issued exactly 16 stores + 2 norm loads
      for (;;) {
        DP_STRIP_BEGIN;
        u32 drawn = 0u;                            // the strip after s1
        if (lane == 0) drawn = __hip_atomic_fetch_add(dealer, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const int s2 = __builtin_amdgcn_readfirstlane((int)drawn);
        const bool more = s1 < se;
        const int snext = more ? s1 : s;           // (no next strip: the trailing requests re-read this one)
        const u16* nxt = strip_base(snext);
        f32x4 acc[8][2];
#pragma unroll
        for (int ib = 0; ib < 8; ++ib)
#pragma unroll
          for (int jb = 0; jb < 2; ++jb) acc[ib][jb] = f32x4{0.f, 0.f, 0.f, 0.f};
        // In its k loop a wave outranks the wave of its SIMD that is in an epilogue.  The idea -- an older wave's stream of
        // epilogue VALU instructions keeps the younger one's MFMAs waiting at the shared issue port -- did not survive the
        // measurements: another wave's plain VALU, scalar and LDS instructions cost the matrix pipe nothing
        // (scratch/coissue_probe.hip), and with / without these two s_setprio the pass takes 0.2034 / 0.2027 ms unsettled and
        // 0.1877 / 0.1891 ms settled (scratch/ab_fused.py): inside the noise, kept because it never loses.
        DP_SETPRIO(2);
        for (int g = 0; g < groups; ++g) {
          const bool last = g + 1 >= groups;
#pragma unroll
          for (int u = 0; u < DP_RING; ++u) {
            const int kt = g * DP_RING + u;
            DP_STAMP(2);
            // (the looser wait first, unconditionally: every path from a request to its use then passes a wait, which is
            // what stein_amd/csrc/isa_check.py verifies on the assembly)
            stream_wait<W_EARLY>();
            if (!(g == 0 && regular)) stream_wait<W_LATE>();
            DP_STAMP(0);
            if (kt < ntk) dp_step<NP>(panel + kt * NP * XPLANE + aoff, ring[u], acc);
            DP_STAMP(1);
            request(last ? nxt : cur, last ? u : kt + DP_RING, ring[u]);
          }
        }
        // the strip's column norms were requested a whole k loop ago: everything but the last four requests of this loop is
        // younger than they are (with more than one group of k tiles the loop's own waits have covered them already and
        // this one costs nothing)
        DP_STAMP(2);
        DP_SETPRIO(0);
        stream_wait<DP_RING * LPS>();
        // ---- epilogue of the strip ----------------------------------------------------------------------------------------
#ifdef STEIN_DP_ABL_NOEPI   // (timing-only ablation: no epilogue at all; the accumulators are kept alive)
        const bool diag = false;
#pragma unroll
        for (int ib = 0; ib < 8; ++ib)
#pragma unroll
          for (int jb = 0; jb < 2; ++jb) asm volatile("" :: "v"(acc[ib][jb]));
#else
#define DP_RI(ib) ri[ib]
#define DP_EPI_FAST
#include "stein_dpanel_epilogue.inc"
#undef DP_EPI_FAST
#undef DP_RI
#endif
        regular = !diag;
        if (window && sx.qn >= (u32)(DP_QCAP / 2)) { dp_flush(sx, spec, spec_buf, lane); regular = false; }
        request_norms(snext);                      // ("memory": the strip's stores are issued before this point)
        cur = nxt;
        if (diag) { DP_STAMP(6); DP_STAMP_COUNT(7); } else DP_STAMP(3);
        DP_STAMP_COUNT(5);
        DP_STRIP_END(2 * unit + half, s);
        if (!more) break;
        s = s1;
        s1 = s2;
      }
      stream_wait<0>();                            // the trailing re-reads land before their registers move on
    }
    DP_STAMP(4);
  }
  DP_STAMP_FLUSH(lane);
  DP_STAMP_WG(p, w, lane);
  DP_SLOW_FLUSH(p, w, lane);
  DP_HIST_FINISH
  if (window) {
    dp_flush(sx, spec, spec_buf, lane);
    const u32 below = dp_below_total(sx, SYM ? 2u : 1u);
    if (lane == 0) {
      if (below) atomicAdd(reinterpret_cast<unsigned long long*>(spec_buf + ((blockIdx.x * 8 + w) % SPEC_SLOTS) * 8),
                           (unsigned long long)below);
      if (sx.over) spec->overflow = 1u;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_distance_panel_deep: the same pass for a K that does not fit the LDS panel (NP * K / 32 > 16: d > 256 for fp32 inputs,
// BASELINE config 4's d = 2001).  The panel holds KC = 16 / NP k tiles at a time.  An accumulator tile has to live
// through all of K, so the eight waves take one strip each (a "round", dealt statically: the rounds are barrier-bound anyway),
// and walk K chunk by chunk in step: barrier, all waves copy the panel's next chunk to LDS, barrier, every wave multiplies
// its eight (or sixteen) k tiles -- the strip operand streams through the same register ring as in k_distance_panel, across
// the chunk boundaries -- and after the last chunk runs the same epilogue.  The panel copy is exposed once per chunk
// (128 KB from L2 against 8 x 384 MFMAs per wave); the compiler's own waits for those copies drain the ring at every
// boundary, so the hand-counted waits behind a boundary only ever see fewer operations in flight than they allow.
// ------------------------------------------------------------------------------------------------
template <bool SYM, int NP>
__global__ __launch_bounds__(DP_THREADS, 2) void k_distance_panel_deep(const u16* __restrict__ T3, int ntk,
                                                                       const float* __restrict__ r, float* __restrict__ D,
                                                                       int row0, int tiles_m, int tiles_n, long ldD,
                                                                       const float* __restrict__ two_s,
                                                                       SpecState* __restrict__ spec, u64* __restrict__ spec_buf,
                                                                       u64* __restrict__ hist0) {
  constexpr int LPS = 2 * NP;
  constexpr int DP_HIST_EVERY = 1;
  constexpr int KC = 16 / NP;                  // k tiles of the panel in LDS at a time
  constexpr int W_LATE = 3 * LPS;              // the loads of the three k tiles behind the one waited for stay in flight
  __shared__ __attribute__((aligned(16))) unsigned char smem[DP_LDS + 512];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  unsigned char* const panel = smem;
  unsigned char* const stg = smem + DP_PANEL + w * DP_STAGE;
  float* const rn = reinterpret_cast<float*>(smem + DP_LDS);   // norms of the segment's 128 rows
  const long ntc = ldD >> 5;
  const int groups = (ntk + DP_RING - 1) / DP_RING, nchunks = (ntk + KC - 1) / KC;
  const float nts = -*two_s;

  DpWin sx;
  sx.lo = 0xffffffffu; sx.width = 0u; sx.below = 0u; sx.below_lane = 0u; sx.qn = 0u; sx.over = 0u;
  sx.q = reinterpret_cast<u64*>(smem + DP_PANEL + 8 * DP_STAGE + w * DP_QBYTES);
  bool window = false;
  if (spec) {
    const u32 lk = spec->lo_key, wd = spec->width;
    window = wd != 0u;
    if (window) { sx.lo = lk & 0x7fffffffu; sx.width = wd; }
    // no window this step and nowhere to put the level-0 histogram: the k_hist<0> pass over D must run
    else if (!hist0 && blockIdx.x == 0 && t == 0) spec->skip_l0 = 0u;
  }
  DP_HIST_SETUP;

  // the workgroup's share of the strip order: as in k_distance_panel
  const int G = gridDim.x, p = xcd_remap(blockIdx.x, G);
  const int U = SYM ? (tiles_n + 1) / 2 : tiles_m;
  int unit, member, members, unit_step;
  if (U >= G) { unit = p; member = 0; members = 1; unit_step = G; }
  else {
    const int M = G / U, big = G - U * M;
    if (p < big * (M + 1)) { unit = p / (M + 1); member = p % (M + 1); members = M + 1; }
    else { const int q = p - big * (M + 1); unit = big + q / M; member = q % M; members = M; }
    unit_step = U;
  }
  const u32 aoff = (u32)lane * 16u, aoff1 = aoff + (u32)XPLANE;
  const int li = lane & 15, lq = lane >> 4;
  u32x4 ring[DP_RING][2][3];

  for (; unit < U; unit += unit_step)
  for (int half = 0; half < (SYM ? 2 : 1); ++half) {
    int I, len;
    dp_segment<SYM>(SYM ? 2 * unit + half : unit, tiles_n, I, len);
    const int se = len > member ? (len - member + members - 1) / members : 0;   // this workgroup's strips: member + members k
    if (se <= 0) continue;
    const int cfirst = (SYM ? 4 * I : 0) + member;
    auto strip_base = [&](int strip) {
      const long j0 = 32l * (cfirst + members * strip);
      return reinterpret_cast<const u16*>(dp_uniform(T3 + ((size_t)(j0 >> 7) * ntk * 3) * XTILE_E + ((j0 & 127) >> 4) * 512));
    };
    // (asm statements with vector-memory instructions open with s_nop 4: see k_distance_panel)
    auto request = [&](const u16* sbase, int kt, u32x4 (&slot)[2][3]) {
      const u16* src = sbase + (size_t)(kt < ntk ? kt : ntk - 1) * 3 * XTILE_E;
      if constexpr (NP == 2) {
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %6\n\tglobal_load_dwordx4 %1, %5, %6\n\t"
                     "global_load_dwordx4 %2, %4, %6 offset:1024\n\tglobal_load_dwordx4 %3, %5, %6 offset:1024"
                     : "=&v"(slot[0][0]), "=&v"(slot[0][1]), "=&v"(slot[1][0]), "=&v"(slot[1][1])
                     : "v"(aoff), "v"(aoff1), "s"(src));
      } else {
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:1024"
                     : "=&v"(slot[0][0]), "=&v"(slot[1][0]) : "v"(aoff), "s"(src));
      }
    };
    __syncthreads();                             // nobody still reads the previous segment's row norms
    if (t < 128) rn[t] = r[row0 + 128 * I + t];  // (visible behind the first chunk's barriers)
    const int rounds = (se + 7) / 8;
    int s = w;                                   // round k: strip 8 k + w
    const u16* cur = strip_base(s < se ? s : 0);
    if (s < se) {
#pragma unroll
      for (int u = 0; u < DP_RING; ++u) request(cur, u, ring[u]);
    }
    for (int round = 0; round < rounds; ++round, s += 8) {
      const bool active = s < se;                // (wave-uniform; once false it stays false)
      const int snext = s + 8 < se ? s + 8 : (active ? s : 0);   // (no next strip: the trailing requests re-read this one)
      const u16* nxt = strip_base(snext);
      f32x4 acc[8][2];
#pragma unroll
      for (int ib = 0; ib < 8; ++ib)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) acc[ib][jb] = f32x4{0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < nchunks; ++c) {
        __syncthreads();                         // everybody is done with the chunk in LDS
        {
          // LDS-direct loads (global_load_lds_dwordx4: a wave's 64 x 16 bytes land at a wave-uniform LDS base + 16 lane, which
          // is the panel's layout): the accumulators and the ring are live here and there are no registers for the pieces.
          // Scalar base + one 32-bit lane offset, so that the sixteen addresses cost no vector registers either.
          typedef const __attribute__((address_space(1))) void* gptr_t;
          typedef __attribute__((address_space(3))) void* lptr_t;
          const u16* rowtile = T3 + ((size_t)(row0 / 128 + I) * ntk * 3) * XTILE_E + w * 512;
#pragma unroll
          for (int q = 0; q < 16; ++q) {         // (k tiles past the end repeat the last one: nobody reads them)
            const int ktq = c * KC + q / NP;
            const char* pl = reinterpret_cast<const char*>(dp_uniform(rowtile + ((size_t)(ktq < ntk ? ktq : ntk - 1) * 3 + q % NP) * XTILE_E));
            __builtin_amdgcn_global_load_lds((gptr_t)(pl + aoff), (lptr_t)(panel + q * XPLANE + w * 1024), 16, 0, 0);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (active) {
          for (int gl = 0; gl < KC / DP_RING; ++gl) {
            const int g = c * (KC / DP_RING) + gl;
            if (g >= groups) break;
            const bool last = g + 1 >= groups;
#pragma unroll
            for (int u = 0; u < DP_RING; ++u) {
              const int kt = g * DP_RING + u;
              stream_wait<W_LATE>();
              if (kt < ntk) dp_step<NP>(panel + (kt - c * KC) * NP * XPLANE + aoff, ring[u], acc);
              request(last ? nxt : cur, last ? u : kt + DP_RING, ring[u]);
            }
          }
        }
      }
      if (active) {
        // the norms of the strip's columns: plain loads, here (the compiler's wait for them also lands the next strip's first
        // requests -- the chunk boundary behind the epilogue would have done that anyway); the row norms come from the
        // segment's LDS copy one at a time (eight registers less than holding them all: this kernel has none to spare)
        f32x4g rj[2];
        {
          const float* rb = r + 32l * (cfirst + members * s) + 4 * lq;
          rj[0] = *reinterpret_cast<const f32x4g*>(rb);
          rj[1] = *reinterpret_cast<const f32x4g*>(rb + 16);
        }
#define DP_RI(ib) rn[16 * (ib) + li]
#include "stein_dpanel_epilogue.inc"
#undef DP_RI
        (void)diag;
        if (window && sx.qn >= (u32)(DP_QCAP / 2)) dp_flush(sx, spec, spec_buf, lane);
        cur = nxt;
      }
    }
    stream_wait<0>();                            // the trailing re-reads land before their registers move on
  }
  DP_HIST_FINISH
  if (window) {
    dp_flush(sx, spec, spec_buf, lane);
    const u32 below = dp_below_total(sx, SYM ? 2u : 1u);
    if (lane == 0) {
      if (below) atomicAdd(reinterpret_cast<unsigned long long*>(spec_buf + ((blockIdx.x * 8 + w) % SPEC_SLOTS) * 8),
                           (unsigned long long)below);
      if (sx.over) spec->overflow = 1u;
    }
  }
}

// ================================================================================================
// host side
// ================================================================================================
bool stein_dpanel_ok(const SteinLayout& L, int dtype, int64_t n, int64_t row0, int64_t n_local, bool level0_only,
                     bool any_size) {
  const int np = stein_x3_kind(dtype);
  const int64_t ntk = L.x3_dk / 32;
  (void)level0_only;                                               // (round 4: the epilogue takes the level-0 histogram too)
  if ((n & 127) || (n_local & 127) || (row0 & 127)) return false;
  // (np * ntk > 16: the panel does not fit 128 KB of LDS -> k_distance_panel_deep walks K in chunks)
  // Small blocks: the launch has a floor of ~20 us (one 157 KB workgroup per CU, a panel load, a barrier) where the per-tile
  // kernel needs 14.  Measured on symmetric blocks (scratch/dist_small_ab.py; tiles vs panel, us): n = 8192: d = 40 39 / 36,
  // d = 128 54 / 42, d = 256 88 / 58; n = 4096: d = 128 24 / 24 (bf16 17 / 21), d = 256 36 / 29, d = 1000 100 / 67;
  // n = 2048, d = 2001: 86 / 85.  So: 16384 strips (eight per wave) whatever K, a quarter of that when K is a full panel.
  const int64_t strips = (n_local / 128) * (n / 32);
  return any_size || strips >= 256 * 8 * 8 || (strips >= 256 * 8 * 2 && np * ntk >= 16);
}

int stein_dpanel_distance(const char* planes, const SteinLayout& L, int dtype, const float* r_all, float* dist_out,
                          int64_t n, int64_t row0, int64_t n_local, int64_t ld_dist, bool symmetric, hipStream_t stream,
                          SpecState* spec, u64* spec_buf, u64* hist0) {
  const u16* T3 = reinterpret_cast<const u16*>(planes + L.x3_t3);
  const float* two_s = reinterpret_cast<const float*>(planes + L.x3_sc) + 4 * L.x3_dc + 1;
  const int ntk = (int)(L.x3_dk / 32);
  const int tiles_m = (int)(n_local / 128), tiles_n = (int)(n / 128);
  static int ncu = 0;
  if (!ncu) {
    int dev = 0, v = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
    ncu = v > 0 ? v : 256;
  }
  const int np = stein_x3_kind(dtype);
#define DP_LAUNCH(KERNEL, SYM, NP) hipLaunchKernelGGL((KERNEL<SYM, NP>), dim3((unsigned)ncu), dim3(DP_THREADS), 0, stream, T3, ntk, r_all, dist_out, (int)row0, tiles_m, tiles_n, (long)ld_dist, two_s, spec, spec_buf, hist0)
  if (np * ntk <= 16) {
    if (np == 1) { if (symmetric) DP_LAUNCH(k_distance_panel, true, 1); else DP_LAUNCH(k_distance_panel, false, 1); }
    else { if (symmetric) DP_LAUNCH(k_distance_panel, true, 2); else DP_LAUNCH(k_distance_panel, false, 2); }
  } else {   // K in chunks of 16 / np k tiles
    if (np == 1) { if (symmetric) DP_LAUNCH(k_distance_panel_deep, true, 1); else DP_LAUNCH(k_distance_panel_deep, false, 1); }
    else { if (symmetric) DP_LAUNCH(k_distance_panel_deep, true, 2); else DP_LAUNCH(k_distance_panel_deep, false, 2); }
  }
#undef DP_LAUNCH
  LAUNCH_CHECK("k_distance_panel");
  return STEIN_OK;
}
