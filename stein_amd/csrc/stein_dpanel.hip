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
// A workgroup's strips are a contiguous range of the row-major strip order (cost-balanced, a panel switch is priced in),
// and workgroups that share an XCD get neighbouring ranges: the same column strips, a row tile apart, meet in its L2.
//
// Restrictions (the caller falls back to k_distance_x3): n, n_local and row0 multiples of 128; NP * K / 32 <= 16 (the panel
// fits 128 KB: d <= 256 for fp32 inputs); no level-0 histogram in the epilogue -- a step without a median window gets it from
// a k_hist<0> pass instead (the kernel clears SpecState::skip_l0 itself).
//
// SYM (single rank): only strips on and above the diagonal exist; off-diagonal tiles count twice (weight 2) and are
// stored once (the contraction reads the others transposed).  The four strips of a diagonal tile store entry by entry:
// (i, j) for i <= j and, for i < j, the same VALUE at (j, i) -- D is exactly symmetric whatever the product order was.
#include "stein_x3.h"
#include "stein_x3_dev.h"
#define STEIN_ABLATE_DPANEL
#include "stein_ablate.h"   // DP_STAMP* / DP_STORE16: hooks of the diagnostic builds (nothing in the shipped library)

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
// strips: one wave pulls a strip's operand into the XCD's L2 and up to 31 others find it there.  That matters more than
// anything else about this kernel: it moves ~1 GB of operand fragments per launch at C3, and with every workgroup on a
// contiguous QUARTER of its row (round 3's first form) half of that still came from beyond the L2 (rocprofv3: 0.53 GB of
// fetches + 0.54 GB of D stores in 0.21 ms = the fabric's whole bandwidth, and the D stores of some XCDs then stalled for
// 100-200 us at a time).  Inside a workgroup the strips are dealt to the waves one by one by a counter in LDS.
// (Also tried and dropped, round 3: strips drawn from per-row counters in global memory by teams of four workgroups, with
// work stealing once a team had run dry: 0.206-0.219 ms at C3 against 0.192-0.204 for contiguous pieces.)

// Who works on what.  A segment = one row tile's run of column strips = one operand panel.
//   row block (not SYM): segment g = row tile g, strips [0, 4 tiles_n): all equally long.
//   SYM: the rows of the triangle are folded into "virtual rows" of equal length: virtual row v = row v (strips [4 v, 4 N)),
//        then row N - 1 - v (strips [4 (N - 1 - v), 4 N)): 4 N + 4 strips whatever v (the middle row of an odd N stands alone).
//        Segment 2 v is the long part, 2 v + 1 the short one.
// A workgroup takes a contiguous, equally priced piece of this order (a panel switch is priced in), and workgroups that
// share an XCD take neighbouring pieces (xcd_remap).  With equally long (virtual) rows, workgroup p and workgroup
// p + (workgroups per row) sit at the same place of neighbouring rows and stream the same column strips at the same time:
// one of them pulls a strip's operand into the XCD's L2, the others find it there.  (Dealt in plain row order the
// triangle's rows shrink, the workgroups of an XCD drift apart along the columns, every one of them streams alone from
// beyond the L2, and the XCDs holding the middle rows took 1.5 times as long as the others.)  Inside a workgroup the strips
// of a piece are dealt to the waves one by one by a counter in LDS.
// (Tried and dropped, round 3: teams of four workgroups per virtual row drawing strips from a counter in global memory,
// with work stealing once a team had run dry -- 0.206-0.219 ms at C3 against 0.192 for this form on like boxes; what holds
// a launch up is not the deal but its D stores, see DESIGN.md.)
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
    if (ib + 1 < 8) {
#pragma unroll
      for (int s = 0; s < NP; ++s) b[(ib + 1) & 1][s] = *reinterpret_cast<const u32x4*>(pk + s * XPLANE + (ib + 1) * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) acc[ib][jb] = x3_products16<NP>(a[jb], b[ib & 1], acc[ib][jb]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

struct DpWin {     // the speculative median window as a wave sees it
  u32 lo, width;   // raw bit patterns [lo, lo + width] of positive floats (width 0: no window)
  u32 below;       // weight of the entries below the window, this wave's strips so far (wave-uniform)
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

template <bool SYM, int NP>
__global__ __launch_bounds__(DP_THREADS, 2) void k_distance_panel(const u16* __restrict__ T3, int ntk,
                                                                  const float* __restrict__ r, float* __restrict__ D,
                                                                  int row0, int tiles_m, int tiles_n, long ldD,
                                                                  const float* __restrict__ two_s,
                                                                  SpecState* __restrict__ spec, u64* __restrict__ spec_buf) {
  constexpr int LPS = 2 * NP;                  // streamed loads per k tile: 2 column blocks x NP planes
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
  sx.lo = 0xffffffffu; sx.width = 0u; sx.below = 0u; sx.qn = 0u; sx.over = 0u;
  sx.q = reinterpret_cast<u64*>(smem + DP_PANEL + 8 * DP_STAGE + w * DP_QBYTES);
  bool window = false;
  if (spec) {
    const u32 lk = spec->lo_key, wd = spec->width;
    window = wd != 0u;
    if (window) { sx.lo = lk & 0x7fffffffu; sx.width = wd; }
    // no window this step: this kernel takes no level-0 histogram, so the k_hist<0> pass over D must run
    else if (blockIdx.x == 0 && t == 0) spec->skip_l0 = 0u;
  }

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
        stream_wait<DP_RING * LPS>();
        // ---- epilogue of the strip ----------------------------------------------------------------------------------------
        const int c32 = cfirst + members * s;
        float* __restrict__ dt = D + ((size_t)I * ntc + c32) * DT_ELEMS;
        const bool diag = SYM && (c32 >> 2) == I;
        const u32 wt = SYM ? 2u : 1u;
        // a strip of the diagonal tile stores entry by entry behind the same staging: (i, j) for i <= j and, for i < j, the
        // same value at its mirror place (j, i); the entries below the diagonal belong to the strip that holds their mirror
        const int d4 = c32 & 3;
        float* __restrict__ dmir = D + ((size_t)I * ntc + 4 * I) * DT_ELEMS;      // the diagonal tile's first 32 columns
        const int sr = lane >> 3, sc4 = (lane & 7) * 4;                          // staged row / first column of this lane
        const int lane_dir = sr * 32 + sc4, lane_dij = sr - sc4;                  // (staged-row layout: direct stores)
        const int lane_mir = lq * 128 + li, lane_dji = 4 * lq - li;               // (accumulator layout: mirror stores)
#pragma unroll
        for (int ib = 0; ib < 8; ++ib) {
#pragma unroll
          for (int jb = 0; jb < 2; ++jb) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(acc[ib][jb][e], nts, ri[ib] + rj[jb][e]);
            if (window && !diag) { const u32 w4[4] = {wt, wt, wt, wt}; dp_count4<true>(sx, v, w4, wt, lane); }
            *reinterpret_cast<float4*>(stg + li * DP_PITCH + jb * 64 + lq * 16) = make_float4(v[0], v[1], v[2], v[3]);
            if (diag && ib < 2 * d4 + 2) {
              // mirror places, straight from the accumulator layout: lane = row i, so the 16 lanes of a quarter wave write 64
              // contiguous bytes of mirror row j.  (From the staged rows, where neighbouring lanes hold neighbouring COLUMNS,
              // every lane's 4 bytes were a memory transaction of their own: ~8000 per strip, ~150 us, during which the
              // other waves of the CU could not issue their loads either.)
              const int dji = 32 * d4 + 16 * jb - 16 * ib + lane_dji;               // j - i at e = 0
              float* __restrict__ pm = dmir + (size_t)(ib >> 1) * DT_ELEMS + (d4 * 32 + jb * 16) * 32 + ((16 * ib) & 31);
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (dji + e > 0) pm[lane_mir + e * 32] = v[e];
            }
          }
          const float4 x0 = *reinterpret_cast<const float4*>(stg + sr * DP_PITCH + sc4 * 4);
          const float4 x1 = *reinterpret_cast<const float4*>(stg + (8 + sr) * DP_PITCH + sc4 * 4);
          if (!diag) {
            DP_STORE16(dt + ib * 512 + lane * 4, x0);          // rows 16 ib .. + 7: 1 KB contiguous
            DP_STORE16(dt + ib * 512 + 256 + lane * 4, x1);    // rows 16 ib + 8 .. + 15
          } else if (ib < 2 * d4 + 2) {                                          // (blocks below the diagonal block: nothing)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float xs[4] = {h ? x1.x : x0.x, h ? x1.y : x0.y, h ? x1.z : x0.z, h ? x1.w : x0.w};
              // row i = i0 + sr and column j = 32 d4 + sc4 + e inside the tile; everything but lane_dir / dij is
              // wave-uniform (addresses written out this way keep the diagonal path to a handful of registers)
              const int i0 = 16 * ib + 8 * h;
              float* __restrict__ pd = dt + i0 * 32;
              const int dij = i0 - 32 * d4 + lane_dij;                           // i - j at e = 0
              u32 w4[4];
              // the lane's four columns lie all above the diagonal (one 16-byte store, as everywhere else), all below it
              // (nothing), or straddle it (entry by entry; a handful of lanes).  Sixty-four single-entry stores 16 bytes apart
              // per instruction made a diagonal strip take ~100 us: the waves that drew them held up their whole team.
              if (dij < 0) *reinterpret_cast<float4*>(pd + lane_dir) = make_float4(xs[0], xs[1], xs[2], xs[3]);
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                w4[e] = dij < e ? 2u : (dij == e ? 1u : 0u);
                if (w4[e] && dij >= 0) pd[lane_dir + e] = xs[e];
              }
              if (window) dp_count4<false>(sx, xs, w4, 0u, lane);
            }
          }
        }
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
  if (window) {
    dp_flush(sx, spec, spec_buf, lane);
    if (lane == 0) {
      if (sx.below) atomicAdd(reinterpret_cast<unsigned long long*>(spec_buf + ((blockIdx.x * 8 + w) % SPEC_SLOTS) * 8),
                              (unsigned long long)sx.below);
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
  if (level0_only) return false;                                   // the caller wants the level-0 histogram from the epilogue
  if ((n & 127) || (n_local & 127) || (row0 & 127)) return false;
  if (np * ntk > 16) return false;                                 // the panel must fit 128 KB of LDS
  // below ~16 strips per wave the panel loads and the ragged tail outweigh the overlap (and the launch fits one
  // round of the tile kernel anyway)
  return any_size || (n_local / 128) * (n / 32) >= 256 * 8 * 8;
}

int stein_dpanel_distance(const char* planes, const SteinLayout& L, int dtype, const float* r_all, float* dist_out,
                          int64_t n, int64_t row0, int64_t n_local, int64_t ld_dist, bool symmetric, hipStream_t stream,
                          SpecState* spec, u64* spec_buf) {
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
#define DP_LAUNCH(SYM, NP) hipLaunchKernelGGL((k_distance_panel<SYM, NP>), dim3((unsigned)ncu), dim3(DP_THREADS), 0, stream, T3, ntk, r_all, dist_out, (int)row0, tiles_m, tiles_n, (long)ld_dist, two_s, spec, spec_buf)
  if (stein_x3_kind(dtype) == 1) { if (symmetric) DP_LAUNCH(true, 1); else DP_LAUNCH(false, 1); }
  else { if (symmetric) DP_LAUNCH(true, 2); else DP_LAUNCH(false, 2); }
#undef DP_LAUNCH
  LAUNCH_CHECK("k_distance_panel");
  return STEIN_OK;
}
