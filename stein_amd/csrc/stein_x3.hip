// stein_x3.hip -- both GEMMs of the SVGD step on the 16-bit matrix cores with fp32-level accuracy.
//
// gfx950 runs the 16-bit-input MFMAs at 16x the rate of the fp32-input MFMA.  Every fp32 operand x is split into a
// short sum of 16-bit terms whose pairwise products are exact in fp32 (the MFMA accumulates in fp32), and a product
// a*b is formed from the term pairs that matter.  KIND selects the split:
//   KIND 2 (fp32 inputs)           x * 2^s = hi + lo, two fp16 terms (hi = fp16(x'), lo = fp16(x' - hi), the subtraction
//          is exact: 22 significant bits); products lo*hi, hi*lo, hi*hi (the dropped lo*lo is 2^-22 relative).  fp16 has
//          a 5-bit exponent, so operands are pre-scaled by powers of two (exact, undone exactly in the epilogues):
//          theta^T and score^T per column to a column maximum in [2^13, 2^14), theta for the distance GEMM by one factor
//          for the whole matrix, and P = exp2(c D + 14) in (0, 2^14].  An entry far below its column's maximum keeps
//          an ABSOLUTE error of 2^-25 of the scaled unit, i.e. 2^-38 of the column maximum -- the same norm-wise
//          guarantee an fp32 GEMM gives.
//          (A three-term bf16 split, six products and no scaling, was measured in round 1: same accuracy, twice the
//          matrix-core time; it is no longer built.)
//   KIND 1 (bf16 inputs)           the value itself; one product.
//
//   k_colmax, k_make_scales   column maxima -> power-of-two scales (KIND 2; all ones otherwise)
//   k_split         theta, score -> 16-bit operand tiles ("planes", layout below)
//   k_distance_x3   S = T T^T from the planes; shares the fp32 kernel's epilogue (D, level-0 histogram or window counting).
//                   Single rank: only the 128 x 128 tiles on and above the diagonal are computed AND stored
//   k_phi_x3fs      warp-specialised contraction: producer waves build P = exp2(c D) (split on the fly) in LDS -- a k tile
//                   left of the row tile's diagonal block from its mirror image D[j][i] --, consumer waves stream the V
//                   fragments from L2 and issue the MFMAs
//
// Both GEMMs are "row x row" products (C[i][c] = sum_k A[i][k] B[c][k]) with k contiguous for both operands.
//
// Operand tile = 128 rows x 32 k of one plane = [128][32] x 16 bit = 8 KB; a tile has three plane slots (24 KB, KIND of
// them used; the third is a leftover of the three-term split) and tiles are stored tile-major:  tile(rb, kt) at ((rb * ntk + kt) * 3 + plane) * 4096 elements.
// One wave-wide 16-byte-per-lane load therefore covers 1 KB of consecutive memory (row-major planes made every
// 64-byte row piece its own cache-line visit: the producers spent 2900 cycles per k tile issuing loads).
//   T3   rows = particles, k = parameters   (distance operands)
//   Vt3  rows = parameters, k = particles   (theta^T and score^T: the contraction's B operand; never staged in LDS)
//   Each plane of a tile is stored in MFMA fragment order [row / 16][chunk][row % 16][8] (vfrag_offset), so the operand
//   fragment of a 16-row block is ONE coalesced 1 KB load, lane l reading bytes 16 l .. 16 l + 15.  k_distance_x3 stages
//   T3 through LDS chunk by chunk (its tile's rows may straddle two row blocks when a rank's row0 is not a multiple of
//   128); k_distance_panel (stein_dpanel.hip) copies fragments into LDS and streams them into registers as they are.
//
// LDS image of a plane: [128 rows][64 B], chunk c of row r at 16 * (c ^ ((r >> 2) & 3)).  Conflict-free for
//   ds_read_b128 fragments (16-lane groups {0-3,12-15,20-27}..., 64 banks): (4 row + chunk') mod 16 distinct in a group
//   ds_write_b128 staging  (8 consecutive lanes = 2 rows x 4 chunks, 32 banks): even row -> bytes 0..63, odd -> 64..127
//   ds_write_b64 of P      (16 consecutive lanes = 2 rows x 8 half-chunks): same split
// (an 80-byte padded row made every write 2-way conflicted.)  Lane l of a 32x32x16 MFMA reads the 8 consecutive k
// of row (l & 31) at k offset 8 (l >> 5).
//
// Scales area (floats at planes + L.x3_sc; dc = roundup(d, 128)):
//   [0, dc)       in-scale of the score columns       Gt3 = split(G * in)
//   [dc, 2dc)     in-scale of the theta columns       Tt3 = split(theta * in)
//   [2dc, 3dc)    out-scale of the K.G columns        = 1 / (in * 2^PEXP)
//   [3dc, 4dc)    out-scale of the K.theta columns
//   [4dc + 0]     in-scale of theta for T3 (one factor: S = T T^T mixes the columns)
//   [4dc + 1]     two_s = 2 / in^2:  D = r_i + r_j - two_s * S'
//   [4dc + 2]     2^-PEXP, the rowsum(K) unscale
//   then u32 [2][dc]: bit patterns of the column maxima of |score|, |theta| (scratch of k_colmax)

#include "stein_x3.h"

#include <stdlib.h>

#include "stein_x3_dev.h"
constexpr int PEXP_H2 = 14;   // KIND 2: P = exp2(c D + 14), in (0, 2^14] (fp16 normal range down to P = 2^-28)
template <int KIND> struct SplitTraits { static constexpr int pexp = KIND == 2 ? PEXP_H2 : 0; };

#include <type_traits>
#define STEIN_ABLATE_X3
#include "stein_ablate.h"   // STAMP / X3_STAMP_*: phase-stamp hooks of the diagnostic builds (nothing in the shipped library)

__device__ __forceinline__ int xswz(int row, int chunk) { return (chunk ^ ((row >> 2) & 3)) * 16; }
// LDS image of a P plane in the contraction: [128 rows][64 B], chunk c of row r at 16 * (c ^ g((r >> 2) & 3)) with
// g = {0, 2, 3, 1}.  The 16x16x32 A fragment (lane l: row l & 15, chunk l >> 4) is read by ds_read_b128 in the lane
// groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: each holds the 16 rows once, with chunk c for rows 0-3 / 12-15
// and c + 1 for rows 4-11 (or the reverse); this g makes the four 16-byte slots of every row-mod-4 class distinct in
// all four groups.  Writes (16 lanes = 2 whole rows) are conflict-free under any per-row permutation.
__device__ __forceinline__ int pswz(int row, int chunk) { return (chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3)) * 16; }

// ------------------------------------------------------------------------------------------------
// splitting
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 cvt_pk_bf16(float lo, float hi) {   // round-to-nearest-even, lo -> bits 15:0
  u32 r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

__device__ __forceinline__ u32 cvt_pk_f16(float lo, float hi) {    // round-to-nearest-even, lo -> bits 15:0
  u32 r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// x - (fp16 in the low / high half of h), one instruction, exact
__device__ __forceinline__ float f16_resid_lo(u32 h, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}
__device__ __forceinline__ float f16_resid_hi(u32 h, float x) {
  float r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(x));
  return r;
}

// two fp32 values -> KIND packed 16-bit pairs w[0..KIND) (x in the low half), most significant term first
template <int KIND>
__device__ __forceinline__ void split_pair(float x, float y, u32 (&w)[3]) {
  if (KIND == 2) {
    w[0] = cvt_pk_f16(x, y);
    w[1] = cvt_pk_f16(f16_resid_lo(w[0], x), f16_resid_hi(w[0], y));
  } else {
    w[0] = cvt_pk_bf16(x, y);   // bf16 inputs: exact
  }
}

// ------------------------------------------------------------------------------------------------
// scales (KIND 2)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ u32 abs_bits(u16 v) { return ((u32)v << 16) & 0x7fffffffu; }
__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((u32)(e + 127) << 23); }   // -126 <= e <= 127
// exponent s with  max * 2^s in [2^13, 2^14); 0 for an all-zero / subnormal / non-finite column
__device__ __forceinline__ int scale_exp(u32 maxbits, int lim) {
  const int E = (int)((maxbits >> 23) & 0xffu);
  if (E == 0 || E == 255) return 0;
  const int sft = 140 - E;
  return sft < -lim ? -lim : (sft > lim ? lim : sft);
}

// maxima -> the scales area (see the header), by one workgroup of 256 threads.  enable = 0 writes the neutral scales.
__device__ __forceinline__ void make_scales_body(const u32* cmax, int dc, float* __restrict__ sc, int pexp, int enable,
                                                 u32* red) {
  const int t = threadIdx.x;
  u32 m = 0u;
  for (int c = t; c < dc; c += 256) {
    const u32 mg = load_fresh(cmax + c), mt = load_fresh(cmax + dc + c);
    const int sg = enable ? scale_exp(mg, 100) : 0, st = enable ? scale_exp(mt, 100) : 0;
    sc[c] = pow2i(sg);
    sc[dc + c] = pow2i(st);
    sc[2 * dc + c] = pow2i(-sg - pexp);
    sc[3 * dc + c] = pow2i(-st - pexp);
    m = max(m, mt);
  }
  red[t] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (t < o) red[t] = max(red[t], red[t + o]);
    __syncthreads();
  }
  if (t == 0) {
    const int sa = enable ? scale_exp(red[0], 60) : 0;
    sc[4 * dc + 0] = pow2i(sa);
    sc[4 * dc + 1] = pow2i(1 - 2 * sa);
    sc[4 * dc + 2] = pow2i(-pexp);
  }
}

// column maxima of |X| [n][d] as bit patterns (non-negative floats order like unsigned integers)
// blockIdx.z = 0: X0 -> cmax[0, dc), 1: X1 -> cmax[dc, 2 dc)
// done != NULL (fused call: both matrices given, completion counters zeroed by the prologue): the last workgroup to finish
// also turns the maxima into the scales, which saves the k_make_scales launch (round 4: a two-level count, so whatever the grid).
// V4 (fp32, d % 4 == 0, 16-byte aligned rows; 1024 threads): a lane reads four columns at once, so a wave's load is one full
// KB of a row instead of 256 bytes, sixteen waves walk sixteen rows at a time, and the workgroup's maxima are added by 256
// threads, one column each (coalesced atomics, and a quarter as many per column as with 256-thread workgroups: the
// same-address atomic chains were the larger part of the 4-byte form's 14-16 us at C3).
template <typename TIN, bool V4>
__global__ __launch_bounds__(V4 ? 1024 : 256) void k_colmax(const TIN* __restrict__ X0, const TIN* __restrict__ X1, int n, int d,
                                                            u32* __restrict__ cmax0, int dc, int zbase, float* __restrict__ sc,
                                                            int pexp, HistSync* done) {
  const int z = blockIdx.z + zbase;
  const TIN* __restrict__ X = z ? X1 : X0;
  u32* __restrict__ cmax = cmax0 + (z ? dc : 0);
  constexpr int CW = V4 ? 4 : 1;                 // columns per lane
  constexpr int NW = V4 ? 16 : 4;                // waves = rows in flight per workgroup
  __shared__ u32 red[NW][64 * CW];
  const int t = threadIdx.x, cx = t & 63, ry = t >> 6;
  const int col = (blockIdx.x * 64 + cx) * CW;
  u32 m[CW];
#pragma unroll
  for (int e = 0; e < CW; ++e) m[e] = 0u;
  if (col < d) {
    // four independent loads in flight per lane (one dependent load at a time read the matrices at 2 TB/s)
    const long step = (long)gridDim.y * NW;
    long r = (long)blockIdx.y * NW + ry;
    if constexpr (V4) {
      if constexpr (std::is_same<TIN, float>::value) {
        u32 m1[4] = {0u, 0u, 0u, 0u}, m2[4] = {0u, 0u, 0u, 0u}, m3[4] = {0u, 0u, 0u, 0u};
        auto ld = [&](long row) { return *reinterpret_cast<const float4*>(X + (size_t)row * d + col); };
        auto up = [&](u32 (&acc)[4], const float4& v) {
          acc[0] = max(acc[0], abs_bits(v.x)); acc[1] = max(acc[1], abs_bits(v.y));
          acc[2] = max(acc[2], abs_bits(v.z)); acc[3] = max(acc[3], abs_bits(v.w));
        };
        for (; r + 3 * step < n; r += 4 * step) {
          const float4 a0 = ld(r), a1 = ld(r + step), a2 = ld(r + 2 * step), a3 = ld(r + 3 * step);
          up(m, a0); up(m1, a1); up(m2, a2); up(m3, a3);
        }
        for (; r < n; r += step) { const float4 a0 = ld(r); up(m, a0); }
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = max(max(m[e], m1[e]), max(m2[e], m3[e]));
      }
    } else {
      u32 m1 = 0u, m2 = 0u, m3 = 0u;
      for (; r + 3 * step < n; r += 4 * step) {
        const u32 a0 = abs_bits(X[(size_t)r * d + col]), a1 = abs_bits(X[(size_t)(r + step) * d + col]);
        const u32 a2 = abs_bits(X[(size_t)(r + 2 * step) * d + col]), a3 = abs_bits(X[(size_t)(r + 3 * step) * d + col]);
        m[0] = max(m[0], a0); m1 = max(m1, a1); m2 = max(m2, a2); m3 = max(m3, a3);
      }
      for (; r < n; r += step) m[0] = max(m[0], abs_bits(X[(size_t)r * d + col]));
      m[0] = max(max(m[0], m1), max(m2, m3));
    }
  }
#pragma unroll
  for (int e = 0; e < CW; ++e) red[ry][cx * CW + e] = m[e];
  __syncthreads();
  if (t < 64 * CW) {                             // one thread per column of the workgroup's span
    const int c = blockIdx.x * 64 * CW + t;
    if (c < d) {
      u32 v = 0u;
#pragma unroll
      for (int w = 0; w < NW; ++w) v = max(v, red[w][t]);
      atomicMax(&cmax[c], v);
    }
  }
  if (done) {
    __shared__ u32 s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's maxima have been acknowledged
    __syncthreads();
    if (t == 0) {
      const u32 id = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
      s_last = tree_report_done(done->cm_leaf, &done->cm_top, id, gridDim.x * gridDim.y * gridDim.z) ? 1u : 0u;
    }
    __syncthreads();
    if (s_last) make_scales_body(cmax0, dc, sc, pexp, 1, &red[0][0]);   // (written for 256 threads; more of them repeat columns: same values)
  }
}

// one workgroup: maxima -> the scales area
__global__ __launch_bounds__(256) void k_make_scales(const u32* __restrict__ cmax, int dc, float* __restrict__ sc, int pexp,
                                                     int enable) {
  __shared__ u32 red[256];
  make_scales_body(cmax, dc, sc, pexp, enable, red);
}

// One 64x64 tile of X [n][d] per workgroup.
//   R  != NULL: tile-major image with rows = X rows (dk / 32 k tiles per row block), fragment order, rows < r_rows, k < dk
//   Tt != NULL: tile-major image with rows = X columns (nk / 32 k tiles per row block), pre-swizzled, rows < dc, k < nk
// The grid covers the padded extents; out-of-range source entries are written as zero.
__device__ __forceinline__ float load_as_f32(const float* p) { return *p; }
__device__ __forceinline__ float load_as_f32(const u16* p) { return __uint_as_float((u32)*p << 16); }   // bf16 bits

// TIN = float (KIND 2 or 3) or u16 = bf16 bits (KIND 1).  sc_all scales the row-major image, sc_col[c] column c of the
// transposed one (both powers of two; 1 unless KIND 2).
// blockIdx.z = 0: theta (row-major image R and transposed image Tt0), 1: score (transposed image Tt1 only; the grid is
// sized for theta's padded extents, blocks outside the score's exit at once)
// PRO (bf16 inputs in the fused call): the grid has one more z slice, whose workgroups do the fused call's prologue
// (prologue_body, stein_common.h: row norms, median state, tickets, neutral scales) -- bf16 planes need no scales, so
// nothing here waits for it, and the step is one launch shorter.  The split slices then must not READ the scales (the
// prologue slice writes them in the same launch): they are 1 by definition for KIND 1.
template <typename TIN, int KIND, bool PRO = false>
__global__ __launch_bounds__(256) void k_split(const TIN* __restrict__ X0, const TIN* __restrict__ X1, int n, int d,
                                               u16* __restrict__ R0, long r_rows, int dk, u16* __restrict__ Tt0,
                                               u16* __restrict__ Tt1, int dc, long nk, const float* __restrict__ sc,
                                               int zbase, PrologueArgs pro) {
  if (PRO && blockIdx.z == gridDim.z - 1) {
    prologue_body<TIN>(X0, pro, (int)(blockIdx.y * gridDim.x + blockIdx.x), (int)(gridDim.x * gridDim.y));
    return;
  }
  const bool score = blockIdx.z + zbase != 0;
  if (score && ((int)blockIdx.x * 64 >= dc || (long)blockIdx.y * 64 >= nk)) return;
  const TIN* __restrict__ X = score ? X1 : X0;
  u16* __restrict__ R = score ? nullptr : R0;
  u16* __restrict__ Tt = score ? Tt1 : Tt0;
  const float* __restrict__ sc_all = sc + 4 * dc;                 // scale of the row-major theta image
  const float* __restrict__ sc_col = sc + (score ? 0 : dc);       // per-column scales of this matrix
  __shared__ u16 tile[KIND][64][66];
  const int t = threadIdx.x;
  const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
  const int lr = t >> 4, lc = (t & 15) * 4;
  const long ntk_r = dk >> 5, ntk_t = nk >> 5;
  const float sa = (R && !PRO) ? *sc_all : 1.f;
  float scq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) scq[q] = (!PRO && Tt && col0 + lc + q < dc) ? sc_col[col0 + lc + q] : 1.f;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = row0 + lr + 16 * p, col = col0 + lc;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (row < n && col + q < d) ? load_as_f32(X + (size_t)row * d + col + q) : 0.f;
    if (R && row < r_rows && col < dk) {   // col % 4 == 0: the 4 entries stay inside one 32-wide k tile
      u32 wa[3], wb[3];
      split_pair<KIND>(v[0] * sa, v[1] * sa, wa);
      split_pair<KIND>(v[2] * sa, v[3] * sa, wb);
      u16* dst = R + (((size_t)(row >> 7) * ntk_r + (col >> 5)) * 3) * XTILE_E + vfrag_offset(row & 127, (col & 31) >> 3) + (col & 7);
#pragma unroll
      for (int s = 0; s < KIND; ++s) *reinterpret_cast<uint2*>(dst + s * XTILE_E) = make_uint2(wa[s], wb[s]);
    }
    if (Tt) {
      u32 wa[3], wb[3];
      split_pair<KIND>(v[0] * scq[0], v[1] * scq[1], wa);
      split_pair<KIND>(v[2] * scq[2], v[3] * scq[3], wb);
      const int rr = lr + 16 * p;
#pragma unroll
      for (int s = 0; s < KIND; ++s) {
        tile[s][lc + 0][rr] = (u16)wa[s]; tile[s][lc + 1][rr] = (u16)(wa[s] >> 16);
        tile[s][lc + 2][rr] = (u16)wb[s]; tile[s][lc + 3][rr] = (u16)(wb[s] >> 16);
      }
    }
  }
  if (!Tt) return;
  __syncthreads();
  // transposed store: thread -> (parameter c = t >> 2, the 16 particles starting at j = row0 + 16 (t & 3)) = two
  // 8-element chunks of the k tile, written in MFMA fragment order (vfrag_offset)
  const int c = col0 + (t >> 2), j = row0 + (t & 3) * 16;
  if (c < dc && j < nk) {   // nk % 32 == 0 and j % 16 == 0: both fragments stay inside one k tile
    const int rowc = c & 127, ch0 = (j & 31) >> 3;
#pragma unroll
    for (int s = 0; s < KIND; ++s) {
      u32 w[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        w[q] = (u32)tile[s][t >> 2][(t & 3) * 16 + 2 * q] | ((u32)tile[s][t >> 2][(t & 3) * 16 + 2 * q + 1] << 16);
      u16* base = Tt + (((size_t)(c >> 7) * ntk_t + (j >> 5)) * 3 + s) * XTILE_E;
      *reinterpret_cast<uint4*>(base + vfrag_offset(rowc, ch0)) = make_uint4(w[0], w[1], w[2], w[3]);
      *reinterpret_cast<uint4*>(base + vfrag_offset(rowc, ch0 + 1)) = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// k_distance_x3
//   Thread t stages chunk t and chunk t + 256 (row += 64) of every plane.  The tile's rows may straddle two row
//   blocks of T3, so each thread keeps its two source pointers; k tile / plane steps are uniform immediates.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ const u16* t3_chunk_ptr(const u16* __restrict__ T3, long ntk, long grow, int c16) {
  return T3 + ((size_t)(grow >> 7) * ntk * 3) * XTILE_E + vfrag_offset((int)(grow & 127), c16);
}

template <int NP>
__device__ __forceinline__ void t3_load(const u16* __restrict__ p0, const u16* __restrict__ p1, int kt, u32x4 (&reg)[6]) {
#pragma unroll
  for (int s = 0; s < NP; ++s) {
    reg[s * 2 + 0] = *reinterpret_cast<const u32x4*>(p0 + ((size_t)kt * 3 + s) * XTILE_E);
    reg[s * 2 + 1] = *reinterpret_cast<const u32x4*>(p1 + ((size_t)kt * 3 + s) * XTILE_E);
  }
}

template <int NP>
__device__ __forceinline__ void x3_store_swz(unsigned char* oper, int t, const u32x4 (&reg)[6]) {
#pragma unroll
  for (int s = 0; s < NP; ++s)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int chunk = t + 256 * q;
      *reinterpret_cast<u32x4*>(oper + s * XPLANE + (chunk >> 2) * XROW + xswz(chunk >> 2, chunk & 3)) = reg[s * 2 + q];
    }
}

// The products of one fragment pair on the 32x32x16 shape, smallest first (plane 0 = most significant term; NP planes:
// 2 -> three fp16 products, 1 -> one bf16 product).  X3_BF / X3_HF and the 16x16x32 form: stein_x3_dev.h
template <int NP>
__device__ __forceinline__ f32x16 x3_products(const u32x4 (&a)[3], const u32x4 (&b)[3], f32x16 c) {
  if (NP == 2) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(X3_HF(a[1]), X3_HF(b[0]), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(X3_HF(a[0]), X3_HF(b[1]), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(X3_HF(a[0]), X3_HF(b[0]), c, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(X3_BF(a[0]), X3_BF(b[0]), c, 0, 0, 0);
}


// one 32-deep k tile already in LDS: 2 k16 steps x (2x2 tiles) x NP-dependent products
template <int NP>
__device__ __forceinline__ void x3_mma_tile(const unsigned char* As, const unsigned char* Bs, int wy, int wx, int lane,
                                            f32x16 (&acc)[2][2]) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    u32x4 a[2][3], b[2][3];
    const int co = xswz(l31, 2 * ks + h);   // (row >> 2) & 3 only depends on row mod 16 = l31 mod 16
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < NP; ++s) {
        a[i][s] = *reinterpret_cast<const u32x4*>(As + s * XPLANE + (wy * 64 + i * 32 + l31) * XROW + co);
        b[i][s] = *reinterpret_cast<const u32x4*>(Bs + s * XPLANE + (wx * 64 + i * 32 + l31) * XROW + co);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = x3_products<NP>(a[i], b[j], acc[i][j]);
  }
}

template <bool SYM, int NP>
__global__ __launch_bounds__(NTHREADS, 3) void k_distance_x3(const u16* __restrict__ T3, int ntk,
                                                             const float* __restrict__ r, float* __restrict__ D, int n,
                                                             int row0, int n_local, long ldD, int tiles_m, int tiles_n,
                                                             u64* __restrict__ hist0, const float* __restrict__ two_s,
                                                             SpecState* __restrict__ spec, u64* __restrict__ spec_buf) {
  // two operand tiles in the main loop; the epilogue reuses the array (EPI_LDS_BYTES)
  constexpr int OP = NP * XPLANE;   // one operand tile: NP planes
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * OP > EPI_LDS_BYTES ? 2 * OP : EPI_LDS_BYTES];
  unsigned char* As = smem;
  unsigned char* Bs = smem + OP;
  int tile_m, tile_n;
  if (!distance_tile<SYM>(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tile_m, tile_n)) return;
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wy = wid >> 1, wx = wid & 1;
  const long arow0 = row0 + (long)tile_m * BM, brow0 = (long)tile_n * BN;
  const EpiPrefetch pf = distance_epilogue_prefetch(r, n, row0, n_local, tile_m, tile_n, spec);
  const float two_s_v = *two_s;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const u16* __restrict__ pa0 = t3_chunk_ptr(T3, ntk, arow0 + (t >> 2), t & 3);
  const u16* __restrict__ pa1 = t3_chunk_ptr(T3, ntk, arow0 + (t >> 2) + 64, t & 3);
  const u16* __restrict__ pb0 = t3_chunk_ptr(T3, ntk, brow0 + (t >> 2), t & 3);
  const u16* __restrict__ pb1 = t3_chunk_ptr(T3, ntk, brow0 + (t >> 2) + 64, t & 3);
  // the next k tile's operands are prefetched into registers under the MFMAs (a second register set, two tiles
  // ahead, bought nothing and costs the third workgroup per CU)
  u32x4 ra[6], rb[6];
  X3_STAMP_DECL;
  t3_load<NP>(pa0, pa1, 0, ra);
  t3_load<NP>(pb0, pb1, 0, rb);
  for (int kt = 0; kt < ntk; ++kt) {
    x3_store_swz<NP>(As, t, ra);
    x3_store_swz<NP>(Bs, t, rb);
    STAMP(0);   // waiting for the tile's loads + LDS stores
    __syncthreads();
    STAMP(1);   // barrier
    if (kt + 1 < ntk) {
      t3_load<NP>(pa0, pa1, kt + 1, ra);
      t3_load<NP>(pb0, pb1, kt + 1, rb);
    }
    x3_mma_tile<NP>(As, Bs, wy, wx, lane, acc);
    STAMP(2);   // load issue + fragment reads + MFMAs
    __syncthreads();
    STAMP(1);
  }
  // SYM: only the tiles on and above the diagonal are stored (the contraction reads the others transposed)
  distance_epilogue<SYM, false>(acc, reinterpret_cast<u32*>(smem), D, n, n_local, ldD, tile_m, tile_n, hist0, pf,
                                two_s_v, spec, spec_buf);
  X3_STAMP_FLUSH_DISTANCE;
}

// ------------------------------------------------------------------------------------------------
// k_phi_x3fs: the contraction, warp-specialised, V fragments streamed straight from L2.
//   History (phase stamps, STEIN_STAMPS build): a monolithic kernel (every wave stages, then every wave multiplies)
//   fell into lockstep phases; a producer/consumer split that staged P AND the V tiles through LDS was bound by its
//   producers (64 KB of loads per k tile accepted at the vector-L1 rate while they also ran the exp/split VALU work).
//   Here a 768-thread workgroup owns a 128 x 256 tile of [K.G | K.theta]:
//     waves 0-3   PRODUCERS  load the D tile, P = exp2(c D), split into 16-bit planes, fill LDS stage (it+1) & 1
//     waves 4-11  CONSUMERS  each owns all 128 rows x 32 columns: A fragments (P) by ds_read from LDS stage it & 1,
//                            B fragments (V) by one coalesced 1 KB global load each, issued one k tile ahead right
//                            after the registers' last use; 48 MFMAs per k tile and wave
//   Every SIMD holds one producer and two consumers.  V never touches LDS and is fetched exactly once per workgroup
//   (no two waves share a B fragment).  One barrier per pipeline stage (four k tiles)
//   hands the P stages over.
//   Column space: the 128-column blocks of [G | theta] (each matrix padded to dc = roundup(d, 128)) are paired up,
//   block cb covers pair (2cb, 2cb+1); consumer wave cw takes half cw >> 2, 32-column block cw & 3.
// ------------------------------------------------------------------------------------------------
constexpr int FS_THREADS = 768;

// ---- streamed loads with hand-counted waits ---------------------------------------------------------------------
// The contraction's two roles each keep global loads in flight ACROSS loop iterations (D tiles four k tiles ahead, V
// fragments one k tile ahead).  hipcc counts its own loads' s_waitcnt conservatively across a loop back-edge: in front
// of the first use of a tile it emitted vmcnt(3) ... vmcnt(0), which also waits for the loads issued a moment earlier
// for LATER tiles -- every k tile then paid a full memory latency and the "prefetch" was one tile deep at best (round-2
// finding: the producers needed 1900 cycles per k tile for 500 cycles of work, the matrix waves idled 40 % of the time).
// So these loads are inline asm, which the compiler neither counts nor waits for (cdna_hip_programming.md 5.7), and the
// waits are written by hand: stream_wait<N> lets the N youngest loads of the wave stay in flight; the scheduling fence
// behind it keeps every use of the loaded registers below the wait (an MFMA is not a memory operation: "memory" alone
// does not hold it, cdna_hip_programming.md 5.4 rule 18).  The registers are deliberately NOT operands of the wait: tied
// "+v" operands made the compiler copy the (not yet landed) registers in front of the wait.  Loads complete in issue
// order per wave.  After every change here: check in the .s that no v_mov / spill touches a destination register between
// its load and its wait.
// (stream_load16 / stream_wait: stein_x3_dev.h)
// k tiles per pipeline stage (even: tile parity picks the register set) and the LDS of one k tile (NP planes, packed)
// RB = 16-row blocks of the workgroup's tile: 8 -> 128 rows x 256 columns of [G | theta]; 4 -> 64 rows x 512 columns: the
// P tile, whose exp / split work shares the SIMDs' issue slots with the MFMAs, is then built once per 512 columns, and
// each D tile is read by one workgroup only.  The MFMA count per k tile and wave is the same (RB x CJ x products = 48).
template <int NP, int RB> struct FsGeom {
  static constexpr int KT = 4;                        // 2 stages x KT x NP x 8 KB: 128 KB (NP 2), 64 KB (NP 1); half for RB 4
  static constexpr int ROWS = RB * 16;
  static constexpr int PLN = ROWS * XROW;             // one plane of one k tile: [ROWS][64 B]
  static constexpr int KTB = NP * PLN;                // one k tile in LDS: NP planes
  static constexpr int STAGE = KT * KTB;
  static constexpr int CJ = 16 / RB;                  // 16-column blocks per matrix wave (8 waves: 256 or 512 columns)
  static constexpr int PR = ROWS / 32;                // producer: rows lr + 32 p per thread and tile (4 columns each)
};

template <int NP, int RB>
__global__ __launch_bounds__(FS_THREADS) void k_phi_x3fs(const float* __restrict__ D, long ldD,
                                                         const u16* __restrict__ Gt3, const u16* __restrict__ Tt3,
                                                         long ntj, const float* __restrict__ h2p,
                                                         float* __restrict__ OG, float* __restrict__ OT,
                                                         float* __restrict__ RS, int n, int d, int n_local,
                                                         int tiles_m, int cblocks, int gblocks, int split,
                                                         int jchunk, const float* __restrict__ sc, int dc,
                                                         int upper) {
  // cblocks: workgroups per row tile (grid); gblocks: 128-column blocks per matrix (G and theta each)
  using Geo = FsGeom<NP, RB>;
  constexpr int FS_KT = Geo::KT, FS_KTB = Geo::KTB, FS_STAGE = Geo::STAGE, PLN = Geo::PLN, CJ = Geo::CJ, PR = Geo::PR;
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * FS_STAGE];

  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  // Which (row tile, column block) a workgroup takes.  An XCD runs 32 consecutive logical ids at a time (one workgroup per
  // CU) and its 4 MB L2 is what they share: the workgroups of one row tile read the same D tiles, the workgroups of one
  // column block the same V fragments.  With R row tiles x C column blocks resident, a launch pulls D (16 / C) times and V
  // (tiles_m / R) times from beyond the L2.  Up to 4 column blocks (d <= 256) all of a row tile's workgroups are neighbours
  // anyway; a wide [G | theta] (C4: 16 column blocks) dealt that way made an XCD 2 row tiles x 16 blocks -- V, 134 MB, came
  // in 32 times (4.8 GB of HBM traffic per launch by the counters); 8 row tiles x 4 blocks asks for both four to eight times.
  int cb, tile_m;
  const int plane = cblocks * tiles_m, l2 = logical % plane;
  const int z = logical / plane;
#ifndef STEIN_PHI_ROWMAJOR_MAP   // (diagnostic builds: the plain order everywhere, for same-box A/B runs)
  if (cblocks > 4 && (cblocks & 3) == 0 && (tiles_m & 7) == 0) {
    const int sb = l2 >> 5, in = l2 & 31, cgroups = cblocks >> 2;
    cb = (sb % cgroups) * 4 + (in & 3);
    tile_m = (sb / cgroups) * 8 + (in >> 2);
  } else
#endif
  {
    cb = l2 % cblocks;
    tile_m = l2 / cblocks;
  }
  const int i0 = tile_m * Geo::ROWS;
  const int jbeg = z * jchunk;
  const int jend = min(n, jbeg + jchunk);
  const int ntile = jend > jbeg ? (jend - jbeg + BK - 1) / BK : 0;
  // Visit order of the k tiles inside a pipeline stage.  The `cblocks` workgroups of a row tile all stream the same D row
  // block.  Walking it in step, each of them waits out the full HBM latency of every tile (a request that arrives while
  // another workgroup's fill of the same line is in flight waits for that fill), and a CU holds only ~25 KB of misses in
  // flight.  So workgroup cb walks the FS_KT tiles of a full stage rotated by rot = cb * FS_KT / min(cblocks, FS_KT): at any
  // time the workgroups of a row tile request DIFFERENT tiles, each tile is pulled from HBM by one of them and found in
  // the XCD's L2 a tile or two later by the others (a lag of whole stages, 64 KB per workgroup, did not survive in the
  // 4 MB L2 that 32 CUs share: measured slower).  Slot u of the LDS stage holds tile stage * FS_KT + ((u + rot) % FS_KT);
  // the partial last stage keeps its order.  (The order of the k tiles inside the fp32 accumulation changes with it --
  // deterministically.)
  const int nstage = (ntile + FS_KT - 1) / FS_KT;
  const int rot = (cb % FS_KT) * (cblocks >= FS_KT ? 1 : FS_KT / cblocks) % FS_KT;
  // Visit order of the STAGES (upper, one workgroup per row tile and column block over the whole j range, a power-of-two number
  // of row tiles): stage v of the row tiles that share an XCD (`group` consecutive ones: 32 logical ids / cblocks) is the
  // 128-column block v ^ (tile_m & ~(group - 1)).  Every stored tile D[I][J] is read twice, by row tile I as itself and by
  // row tile J as its mirror image; in the plain order 0, 1, 2 ... the two reads lie |I - J| stages apart (43 on average,
  // 16 MB of D traffic per stage: the second read comes from HBM again), in this order |I % group - J % group| < group
  // stages apart (5 on average), where the memory-side cache still holds the tile.  The row tiles of an XCD keep walking the
  // SAME block at the same time, so V is shared in their L2 as before.
  const int group = cblocks <= 32 ? 32 / cblocks : 1;
  const bool permute = RB == 8 && upper != 0 && jbeg == 0 && jend == n && (cblocks & (cblocks - 1)) == 0 && cblocks <= 4 &&
                       nstage * FS_KT == ntile && (nstage & (nstage - 1)) == 0 && nstage == tiles_m && nstage >= 2 * group;
  const int xmask = permute ? (tile_m & ~(group - 1)) : 0;
  auto stage_of = [&](int v) { return v ^ xmask; };   // (v = nstage, "the stage after the last", stays out of range)
  auto tile_at = [&](int stage, int u) {     // u may run past the stage: u >= FS_KT continues in the next stage
    const int st2 = stage_of(stage + u / FS_KT), u2 = u % FS_KT;
    return st2 * FS_KT + ((st2 + 1) * FS_KT <= ntile ? ((u2 + rot) & (FS_KT - 1)) : u2);
  };

  // upper (single rank, D holds only the 128 x 128 tiles on and above the diagonal): the k tiles left of this row tile's
  // diagonal block are read from their mirror image D[j][i].  The producers gather such a tile with a row-per-lane-group
  // map (below) that leaves every thread holding 4 consecutive j of 4 rows i, so it is written into the SAME natural LDS
  // image with the same number of 8-byte stores: the matrix waves do not know the difference.  The choice is per pipeline
  // stage (FS_KT k tiles = 128 columns; the host keeps jbeg a multiple of 128, so a stage never straddles the diagonal
  // block): the first ntr stages of the j range are mirrored ones.
  const bool up = RB == 8 && upper != 0;
  const int ntr = up ? max(0, min(nstage, (i0 - jbeg) / (FS_KT * BK))) : 0;

  const int t = threadIdx.x;
  // The two roles run separate loops (so neither carries the other's registers) with the same number of
  // barriers: one after the prologue, one per k tile.  The role test is wave-uniform (waves 0-3 / 4-11).
  if (t < 256) {
    // ================================ PRODUCER ================================
    const int pt = t;
    const int lr = pt >> 3, lc = (pt & 7) * 4;   // P staging: rows lr + 32p, 4 consecutive j
    float rs[PR];
#pragma unroll
    for (int p = 0; p < PR; ++p) rs[p] = 0.f;
    // PD register sets (tile index mod PD picks the set): the loads of tile t + PD are issued as soon as tile t has been
    // turned into LDS data.  D streams from HBM (never re-used), so the loads need several tiles of lead
    constexpr int PD = FS_KT;   // 4; FS_KT % PD == 0 keeps the set index static
    f32x4g rd[PD][PR];
    u32 doff[PR];
    const float cexp = -1.44269504088896341f / (2.f * *h2p);   // exp(-D/(2 h2)) = exp2(cexp * D)
    constexpr float pofs = (float)SplitTraits<NP>::pexp;        // P carries 2^pexp (undone by the out-scales)
    // the D tile (tile_m, j0 / 32) is one contiguous [128][32] block of the tile-major distance image (rows past
    // n_local exist as padding and only feed accumulator rows that are never stored)
#pragma unroll
    for (int p = 0; p < PR; ++p) doff[p] = (u32)(((i0 & (DT_ROWS - 1)) + lr + 32 * p) * DT_COLS + lc);
    const float* __restrict__ drow = D + (size_t)(i0 / DT_ROWS) * (ldD >> 5) * DT_ELEMS;
    // mirrored source (upper): wave w gathers from mirror tile w (columns i0 + 32 w .. + 31); thread (c = pt & 7,
    // jq = (pt >> 3) & 7) loads D[j0 + 4 jq + u][i0 + 4 ig .. + 3], ig = 8 w + c, u = 0..3: a quarter wave reads two full
    // 128-byte lines, like the natural map (lanes along j first made every quarter wave touch eight lines: the producers
    // took 2500 instead of 1500 cycles per k tile).  Afterwards the thread holds P(i = 4 ig + e, j = 4 jq + u): four
    // consecutive j of four rows = one 8-byte LDS store per row and plane at row 4 ig + e, byte 8 jq of the natural image
    const int jq = (pt >> 3) & 7, ig = (pt >> 6) * 8 + (pt & 7);
    u32 doff_tr[PR];
#pragma unroll
    for (int u = 0; u < PR; ++u) doff_tr[u] = (u32)((ig >> 3) * DT_ELEMS + (4 * jq + u) * DT_COLS + 4 * (ig & 7)) * 4u;
    const long ntc_d = ldD >> 5;
    float rst[4] = {0.f, 0.f, 0.f, 0.f};   // row sums over the mirrored tiles: rows i0 + 4 ig + e
    // one call site for both sources (base and offsets selected first): a load instruction in each arm of a branch would
    // let the compiler reconcile the two destination registers with copies of registers that are still in flight
    auto request = [&](int j0, bool tr, f32x4g (&rd)[PR]) {
      const float* tile = tr ? D + ((size_t)(j0 >> 7) * ntc_d + 4 * (size_t)(i0 / DT_ROWS)) * DT_ELEMS + (j0 & 127) * DT_COLS
                             : drow + (size_t)(j0 >> 5) * DT_ELEMS;
#pragma unroll
      for (int p = 0; p < PR; ++p) stream_load16(rd[p], tile, tr ? doff_tr[p] : doff[p] * 4u);
    };
    // before tile number v (in visit order) of the ntile is turned into LDS data: its PR loads must have landed, the loads
    // of the up to three later tiles already requested (PR each) stay in flight
    auto wait_loads = [&](int v, f32x4g (&rd)[PR]) {
      const int later = ntile - v - 1;
      (void)rd;
      // (nested, the loosest wait first and unconditional: every path from a request to its use passes a wait, which is
      // what stein_amd/csrc/isa_check.py verifies on the assembly)
      stream_wait<3 * PR>();
      if (later < 3) {
        stream_wait<2 * PR>();
        if (later < 2) {
          stream_wait<PR>();
          if (later < 1) stream_wait<0>();
        }
      }
    };
    // registers of tile j0 -> LDS stage `buf`
    // One k tile: 16 entries per thread, done phase by phase over all 16 (fma, exp, row sums, hi terms, residuals, lo
    // terms, stores) with scheduling fences between the phases.  Entry by entry the stream is a chain of dependent
    // instructions (cvt -> fma_mix -> cvt -> store), and next to two matrix waves on the SIMD a dependent instruction waits
    // for the next free issue window: measured ~16 cycles per instruction.  Sixteen independent instructions per phase
    // fill the windows.
    auto produce = [&](int j0, unsigned char* buf, const f32x4g (&rd)[PR]) {
      const bool full = j0 + BK <= jend;
      float q[PR][4];
#pragma unroll
      for (int p = 0; p < PR; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) q[p][e] = __builtin_fmaf(cexp, rd[p][e], pofs);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < PR; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) q[p][e] = __builtin_amdgcn_exp2f(q[p][e]);
      __builtin_amdgcn_sched_barrier(0);
      if (!full) {   // columns past jend hold whatever the padding holds: force P = 0 there
        const int j = j0 + lc;
#pragma unroll
        for (int p = 0; p < PR; ++p)
#pragma unroll
          for (int e = 0; e < 4; ++e) q[p][e] = (j + e < jend) ? q[p][e] : 0.f;
      }
      if (NP >= 2) {
        u32 hi[PR][2], lo[PR][2];
        float r[PR][4];
#pragma unroll
        for (int p = 0; p < PR; ++p) rs[p] += (q[p][0] + q[p][1]) + (q[p][2] + q[p][3]);
#pragma unroll
        for (int p = 0; p < PR; ++p) { hi[p][0] = cvt_pk_f16(q[p][0], q[p][1]); hi[p][1] = cvt_pk_f16(q[p][2], q[p][3]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < PR; ++p) {
          r[p][0] = f16_resid_lo(hi[p][0], q[p][0]); r[p][1] = f16_resid_hi(hi[p][0], q[p][1]);
          r[p][2] = f16_resid_lo(hi[p][1], q[p][2]); r[p][3] = f16_resid_hi(hi[p][1], q[p][3]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < PR; ++p) { lo[p][0] = cvt_pk_f16(r[p][0], r[p][1]); lo[p][1] = cvt_pk_f16(r[p][2], r[p][3]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < PR; ++p) {
          unsigned char* dst = buf + (lr + 32 * p) * XROW + pswz(lr, lc >> 3) + (lc & 4) * 2;   // (lr+32p)>>2&3 == lr>>2&3
          *reinterpret_cast<uint2*>(dst) = make_uint2(hi[p][0], hi[p][1]);
          *reinterpret_cast<uint2*>(dst + PLN) = make_uint2(lo[p][0], lo[p][1]);
        }
      } else {
#pragma unroll
        for (int p = 0; p < PR; ++p) {
          // bf16 operands: K is rounded to bf16 once and BOTH uses of it (K.theta in the MFMA and rowsum(K) here) see
          // the rounded value, so the repulsion term sum_j K_ij (theta_i - theta_j) stays consistent
          unsigned char* dst = buf + (lr + 32 * p) * XROW + pswz(lr, lc >> 3) + (lc & 4) * 2;
          const u32 h0 = cvt_pk_bf16(q[p][0], q[p][1]), h1 = cvt_pk_bf16(q[p][2], q[p][3]);
          rs[p] += (__uint_as_float(h0 << 16) + __uint_as_float(h0 & 0xffff0000u)) +
                   (__uint_as_float(h1 << 16) + __uint_as_float(h1 & 0xffff0000u));
          *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        }
      }
    };
    // the same for a tile read from its mirror image (always a full tile): rd[u][e] = D[j = 4 jq + u][i = 4 ig + e].
    // (The row sums below run along e, i.e. over adjacent registers: under plain -O3 the SLP vectoriser turned them into
    // v_pk_add_f32, and eight packed adds per k tile beside the MFMAs cost the launch 0.07 ms -- the library is built with
    // -fno-slp-vectorize, __graft_entry__.py.)
    auto produce_tr = [&](unsigned char* buf, const f32x4g (&rd)[PR]) {
      if constexpr (RB == 8) {
        float q[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) q[u][e] = __builtin_fmaf(cexp, rd[u][e], pofs);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 4; ++e) q[u][e] = __builtin_amdgcn_exp2f(q[u][e]);
        __builtin_amdgcn_sched_barrier(0);
        u32 hi[4][2], lo[4][2];   // [row e][j pair]
        if (NP >= 2) {
          float r[4][4];
#pragma unroll
          for (int e = 0; e < 4; ++e) rst[e] += (q[0][e] + q[1][e]) + (q[2][e] + q[3][e]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { hi[e][0] = cvt_pk_f16(q[0][e], q[1][e]); hi[e][1] = cvt_pk_f16(q[2][e], q[3][e]); }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            r[e][0] = f16_resid_lo(hi[e][0], q[0][e]); r[e][1] = f16_resid_hi(hi[e][0], q[1][e]);
            r[e][2] = f16_resid_lo(hi[e][1], q[2][e]); r[e][3] = f16_resid_hi(hi[e][1], q[3][e]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int e = 0; e < 4; ++e) { lo[e][0] = cvt_pk_f16(r[e][0], r[e][1]); lo[e][1] = cvt_pk_f16(r[e][2], r[e][3]); }
          __builtin_amdgcn_sched_barrier(0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {   // bf16: the row sums see the ROUNDED values, as in the natural tiles
            hi[e][0] = cvt_pk_bf16(q[0][e], q[1][e]); hi[e][1] = cvt_pk_bf16(q[2][e], q[3][e]);
            lo[e][0] = lo[e][1] = 0u;
            rst[e] += (__uint_as_float(hi[e][0] << 16) + __uint_as_float(hi[e][0] & 0xffff0000u)) +
                      (__uint_as_float(hi[e][1] << 16) + __uint_as_float(hi[e][1] & 0xffff0000u));
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = 4 * ig + e;   // (row >> 2) & 3 == ig & 3
          unsigned char* dst = buf + row * XROW + pswz(row, jq >> 1) + (jq & 1) * 8;
          *reinterpret_cast<uint2*>(dst) = make_uint2(hi[e][0], hi[e][1]);
          if (NP >= 2) *reinterpret_cast<uint2*>(dst + PLN) = make_uint2(lo[e][0], lo[e][1]);
        }
      }
    };
    auto jt = [&](int tile) { return jbeg + tile * BK; };   // tile index -> first column (jbeg % 32 == 0)
    X3_STAMP_DECL;
    // A pipeline stage holds FS_KT consecutive k tiles, so the workgroup synchronises once per FS_KT tiles.  Tile
    // parity picks the register set (X even, Y odd); a tile's loads are issued two tiles ahead, right after the set is free.
#pragma unroll
    for (int u = 0; u < PD; ++u)
      if (tile_at(0, u) < ntile) request(jt(tile_at(0, u)), stage_of(0) < ntr, rd[u]);
    // stage st: turn the registers of its tiles into LDS data (slot u <- tile_at(st, u)), then request the next stage's
    auto produce_stage = [&](int st, unsigned char* buf) {
      const bool tr_now = stage_of(st) < ntr, tr_next = stage_of(st + 1) < ntr;   // workgroup-uniform
#pragma unroll
      for (int u = 0; u < FS_KT; ++u) {
        const int tile_u = tile_at(st, u), next_u = tile_at(st + 1, u);
        if (tile_u < ntile) {
          STAMP(0);
          wait_loads(st * FS_KT + u, rd[u % PD]);   // (tiles are requested in visit order: position = st * FS_KT + u)
          STAMP(1);   // diagnostic builds: how long the producer waited for this tile's D loads
          if (tr_now) produce_tr(buf + u * FS_KTB, rd[u % PD]);
          else produce(jt(tile_u), buf + u * FS_KTB, rd[u % PD]);
        }
        if (next_u < ntile) request(jt(next_u), tr_next, rd[u % PD]);
      }
    };
    produce_stage(0, smem);
    __syncthreads();
    // iteration st (consumers are on stage st): fill stage st+1 into the other buffer
    for (int st = 0; st < nstage; ++st) {
      STAMP(5);
      if (st + 1 < nstage) produce_stage(st + 1, smem + ((st + 1) & 1) * FS_STAGE);
      STAMP(0);   // produce (includes waiting for the tiles' loads)
      __syncthreads();
      STAMP(2);   // barrier
    }
    X3_STAMP_FLUSH_PRODUCER;
    if (up) {
      // natural tiles: rows lr + 32 p, the 8 threads of a row are 8 consecutive lanes; mirrored tiles: rows 4 ig + e, the 8
      // threads of a row are the lanes jq = 0..7 (lane bits 3..5).  The two row sets meet in LDS (the stage buffers are
      // dead: every wave is past the loop's last barrier; the matrix waves take the same extra barrier) and are added in a
      // fixed order.
      float* red = reinterpret_cast<float*>(smem);   // [128] natural sums | [128] mirrored sums
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float a = rs[p < PR ? p : 0], b = rst[p];
        a += __shfl_xor(a, 1); b += __shfl_xor(b, 8);
        a += __shfl_xor(a, 2); b += __shfl_xor(b, 16);
        a += __shfl_xor(a, 4); b += __shfl_xor(b, 32);
        if ((pt & 7) == 0) red[lr + 32 * p] = a;
        if (jq == 0) red[128 + 4 * ig + p] = b;
      }
      __syncthreads();
      const int row = i0 + pt;
      if (cb == 0 && pt < 128 && row < n_local) RS[(size_t)z * n_local + row] = (red[pt] + red[128 + pt]) * sc[4 * dc + 2];
    } else if (cb == 0) {   // rowsum: the 8 threads of a row are 8 consecutive lanes
#pragma unroll
      for (int p = 0; p < PR; ++p) {
        float sum = rs[p];
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        sum += __shfl_xor(sum, 4);
        const int row = i0 + lr + 32 * p;
        if ((pt & 7) == 0 && row < n_local) RS[(size_t)z * n_local + row] = sum * sc[4 * dc + 2];
      }
    }
  } else {
    // ================================ CONSUMER ================================
    // v_mfma_f32_16x16x32_bf16: one MFMA spans the whole 32-deep k tile.  The chip holds a higher clock on this shape
    // than on 32x32x16 at the same cycles per flop (MI355X_MICROARCH.md, DVFS give-back item 7).
    const int ct = t - 256, lane = ct & 63, cw = ct >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    X3_STAMP_WG_ENTRY;
    // this wave's 128-column block of [G | theta] and its CJ 16-column blocks inside it
    const int g = RB == 8 ? 2 * cb + (cw >> 2) : 4 * cb + (cw >> 1);
    const int wcol = RB == 8 ? (cw & 3) * 32 : (cw & 1) * 64;   // first column inside the block (16 CJ columns)
    // B fragment of (k tile kt, plane s, 16-column block jb): vb + ((kt * 3 + s) * 4096 + jb * 512) elements
    // (wave-uniform part, made provably so for the "s" operand of the streamed loads; per-lane byte offsets boff below)
    const u16* vb_wave = (g < gblocks ? Gt3 + (size_t)g * ntj * 3 * XTILE_E
                                      : Tt3 + (size_t)(g - gblocks) * ntj * 3 * XTILE_E) +
                         (size_t)(jbeg >> 5) * 3 * XTILE_E + wcol * 32;
    const u16* __restrict__ vb = reinterpret_cast<const u16*>(
        ((unsigned long long)(u32)__builtin_amdgcn_readfirstlane((int)((unsigned long long)vb_wave >> 32)) << 32) |
        (unsigned long long)(u32)__builtin_amdgcn_readfirstlane((int)(unsigned long long)vb_wave));
    u32 boff[CJ][3];
#pragma unroll
    for (int j = 0; j < CJ; ++j)   // j = 16-column block
#pragma unroll
      for (int s = 0; s < 3; ++s) boff[j][s] = (u32)(lane * 8 + s * XTILE_E + j * 512) * 2u;
    const int aoff = l15 * XROW + pswz(l15, lq);   // A fragment of 16-row block ib, plane s: + ib * 1024 + s * PLN
    f32x4 acc[RB][CJ];
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // two B register sets: the fragments of tile it+1 are requested at the top of tile it.  Held as 32-bit vectors
    // (loop-carried bf16 vectors get scalarised into 16-bit pieces by the compiler) and bit-cast at the MFMA.
    u32x4 bX[CJ][3], bY[CJ][3];
    auto load_b = [&](int tile, u32x4 (&b)[CJ][3]) {
      const u16* __restrict__ src = vb + (size_t)tile * 3 * XTILE_E;
#pragma unroll
      for (int j = 0; j < CJ; ++j)
#pragma unroll
        for (int s = 0; s < NP; ++s) stream_load16(b[j][s], src, boff[j][s]);
    };
    // before the MFMAs of a tile: its CJ NP fragment loads must have landed; `younger`: the next tile's CJ NP loads have
    // been requested already and stay in flight
    auto wait_b = [&](bool younger, u32x4 (&b)[CJ][3]) {
      (void)b;
      stream_wait<CJ * NP>();
      if (!younger) stream_wait<0>();
    };
    // A fragments are read one 16-row block ahead of their MFMAs; the scheduling fences keep the compiler from hoisting
    // all 24 reads (96 registers) to the top of the tile
    auto read_a = [&](const unsigned char* As, int i, u32x4 (&a)[3]) {
#pragma unroll
      for (int s = 0; s < NP; ++s) a[s] = *reinterpret_cast<const u32x4*>(As + aoff + i * 16 * XROW + s * PLN);
    };
    auto mma_tile = [&](const unsigned char* As, const u32x4 (&b)[CJ][3]) {
      u32x4 a[2][3];
      read_a(As, 0, a[0]);
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        if (i + 1 < RB) read_a(As, i + 1, a[(i + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < CJ; ++j) acc[i][j] = x3_products16<NP>(a[i & 1], b[j], acc[i][j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (ntile > 0) load_b(tile_at(0, 0), bX);
    __syncthreads();
    X3_STAMP_DECL_CONSUMER;
    for (int st = 0; st < nstage; ++st) {
      STAMP(5);
      const unsigned char* As = smem + (st & 1) * FS_STAGE;
      auto after = [&](int u) { return tile_at(st, u); };   // the tile in slot u of this stage (u >= FS_KT: of the next stage)
      constexpr bool kLoadV = true;
#pragma unroll
      for (int u = 0; u < FS_KT; u += 2) {
        if (after(u) >= ntile) break;
        const bool n1 = after(u + 1) < ntile;
        if (kLoadV && n1) load_b(after(u + 1), bY);
        STAMP(3);
        wait_b(kLoadV && n1, bX);
        STAMP(0);   // diagnostic builds: waiting for this tile's V fragments
        // The two matrix waves of a SIMD (cw and cw + 4) take turns at the higher issue priority, tile by tile.  At equal
        // priority the older wave wins every arbitration: it ran ahead (1460 vs 2070 cycles per k tile, per-wave stamps)
        // and idled at the stage barrier while the younger one finished alone.  Measured -2 % on the launch.
        if (cw >> 2) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(2);
        mma_tile(As + u * FS_KTB, bX);
        if (n1) {
          const bool n2 = after(u + 2) < ntile;
          if (kLoadV && n2) load_b(after(u + 2), bX);
          STAMP(3);
          wait_b(kLoadV && n2, bY);
          STAMP(0);
          if (cw >> 2) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
          mma_tile(As + (u + 1) * FS_KTB, bY);
        }
      }
      STAMP(3);   // consumer: fragment reads + MFMAs
      __syncthreads();
      STAMP(4);   // consumer: barrier
    }
    X3_STAMP_FLUSH_CONSUMER;
    if (up) __syncthreads();   // the producers' row-sum exchange (same barrier count in both roles)
    if (g >= 2 * gblocks) return;   // (an odd block count leaves the last workgroup's upper waves without columns)
    float* __restrict__ Oz = (g < gblocks ? OG : OT) + (size_t)z * n_local * d;
    const int cbase = (g < gblocks ? g : g - gblocks) * BN + wcol + l15;
    const float* __restrict__ osc = sc + (g < gblocks ? 2 : 3) * dc;   // out-scales of this wave's matrix
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
      const int col = cbase + j * 16;
      if (col >= d) continue;
      const float os = osc[col];
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int row = i0 + i * 16 + 4 * lq + e;
          if (row < n_local) Oz[(size_t)row * d + col] = acc[i][j][e] * os;
        }
    }
  }
}

// ================================================================================================
// host side
// ================================================================================================
// split KIND of a call: bf16 inputs -> 1 (the values themselves); fp32 inputs -> 2 (two fp16 terms of the scaled value)
static int split_kind(int dtype) { return dtype == STEIN_BF16 ? 1 : 2; }
int stein_x3_kind(int dtype) { return split_kind(dtype); }

template <typename TIN, int KIND>
static void launch_split(hipStream_t stream, const TIN* theta, const TIN* score, int64_t n, int64_t d,
                         const SteinLayout& L, u16* T3, u16* Tt3, u16* Gt3, const float* sc, const PrologueArgs* pro = nullptr) {
  // grid.z walks [theta, score]; a NULL matrix is left out (its planes keep their contents)
  const unsigned nz = (theta ? 1u : 0u) + (score ? 1u : 0u);
  const int zbase = theta ? 0 : 1;
  const int64_t rows = L.x3_rows > L.x3_nk ? L.x3_rows : L.x3_nk;   // particle extent to cover (both multiples of 32)
  const int64_t cols = L.x3_dk > L.x3_dc ? L.x3_dk : L.x3_dc;      // parameter extent
  if (KIND == 1 && pro && theta) {   // + the prologue slice (fused call, bf16)
    const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64), nz + 1u);
    hipLaunchKernelGGL((k_split<TIN, KIND, true>), grid, dim3(256), 0, stream, theta, score, (int)n, (int)d, T3,
                       (long)L.x3_rows, (int)L.x3_dk, Tt3, Gt3, (int)L.x3_dc, (long)L.x3_nk, sc, zbase, *pro);
    return;
  }
  const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64), nz);
  hipLaunchKernelGGL((k_split<TIN, KIND, false>), grid, dim3(256), 0, stream, theta, score, (int)n, (int)d, T3,
                     (long)L.x3_rows, (int)L.x3_dk, Tt3, Gt3, (int)L.x3_dc, (long)L.x3_nk, sc, zbase, PrologueArgs{});
}

// fuse_done != NULL (fused call): the caller has zeroed the column maxima and the ticket, both matrices are given, and
// the column-maxima kernel's last workgroup writes the scales itself
int stein_x3_split(const void* theta_all, const void* score_all, int dtype, int64_t n, int64_t d,
                   const SteinLayout& L, char* planes, hipStream_t stream, HistSync* fuse_done, bool scales_written,
                   const PrologueArgs* prologue) {
  u16* T3 = reinterpret_cast<u16*>(planes + L.x3_t3);
  u16* Tt3 = reinterpret_cast<u16*>(planes + L.x3_tt3);
  u16* Gt3 = reinterpret_cast<u16*>(planes + L.x3_gt3);
  float* sc = reinterpret_cast<float*>(planes + L.x3_sc);
  const int dc = (int)L.x3_dc;
  u32* cmax = reinterpret_cast<u32*>(sc + 4 * dc + 4);
  const int kind = split_kind(dtype);
  if (fuse_done && !(theta_all && score_all)) fuse_done = nullptr;
  if (kind == 2) {   // column maxima of the matrices given (cmax = [score | theta]); the other half keeps its values
    const int zbase = score_all ? 0 : 1;
    const unsigned nz = (score_all ? 1u : 0u) + (theta_all ? 1u : 0u);
    if (!fuse_done) HIP_TRY(hipMemsetAsync(cmax + (size_t)zbase * dc, 0, (size_t)nz * dc * sizeof(u32), stream));
    int gy = (int)((n + 63) / 64);
    if (gy > 256) gy = 256;
    const bool v4 = d % 4 == 0 && (!score_all || ((uintptr_t)score_all & 15) == 0) && (!theta_all || ((uintptr_t)theta_all & 15) == 0);
    if (v4) {
      const int gy4 = gy > 64 ? 64 : gy;
      const dim3 grid((unsigned)((d + 255) / 256), (unsigned)gy4, nz);
      hipLaunchKernelGGL((k_colmax<float, true>), grid, dim3(1024), 0, stream, (const float*)score_all, (const float*)theta_all,
                         (int)n, (int)d, cmax, dc, zbase, sc, PEXP_H2, fuse_done);
    } else {
      const dim3 grid((unsigned)((d + 63) / 64), (unsigned)gy, nz);
      hipLaunchKernelGGL((k_colmax<float, false>), grid, dim3(256), 0, stream, (const float*)score_all, (const float*)theta_all,
                         (int)n, (int)d, cmax, dc, zbase, sc, PEXP_H2, fuse_done);
    }
    LAUNCH_CHECK("k_colmax");
  }
  if ((kind != 2 && !scales_written) || (kind == 2 && !fuse_done)) {
    hipLaunchKernelGGL(k_make_scales, dim3(1), dim3(256), 0, stream, cmax, dc, sc, kind == 2 ? PEXP_H2 : 0,
                       kind == 2 ? 1 : 0);
    LAUNCH_CHECK("k_make_scales");
  }
  if (kind == 1) launch_split<u16, 1>(stream, (const u16*)theta_all, (const u16*)score_all, n, d, L, T3, Tt3, Gt3, sc, prologue);
  else launch_split<float, 2>(stream, (const float*)theta_all, (const float*)score_all, n, d, L, T3, Tt3, Gt3, sc);
  LAUNCH_CHECK("k_split");
  return STEIN_OK;
}

template <bool SYM, int NP>
static void launch_distance_x3(long nblk, hipStream_t stream, const u16* T3, int ntk, const float* r, float* D, int n,
                               int row0, int n_local, long ld, int tiles_m, int tiles_n, u64* hist0, const float* two_s,
                               SpecState* spec, u64* spec_buf) {
  hipLaunchKernelGGL((k_distance_x3<SYM, NP>), dim3((unsigned)nblk), dim3(NTHREADS), 0, stream, T3, ntk, r, D, n, row0,
                     n_local, ld, tiles_m, tiles_n, hist0, two_s, spec, spec_buf);
}

int stein_x3_distance(const char* planes, const SteinLayout& L, int dtype, const float* r_all, float* dist_out,
                      int64_t n, int64_t d, int64_t row0, int64_t n_local, int64_t ld_dist, u64* hist0, bool symmetric,
                      hipStream_t stream, SpecState* spec, u64* spec_buf, int panel) {
  (void)d;
  if (panel >= 0 && stein_dpanel_ok(L, dtype, n, row0, n_local, hist0 != nullptr && spec == nullptr, panel > 0))
    return stein_dpanel_distance(planes, L, dtype, r_all, dist_out, n, row0, n_local, ld_dist, symmetric, stream, spec,
                                 spec_buf, hist0);
  const u16* T3 = reinterpret_cast<const u16*>(planes + L.x3_t3);
  const float* two_s = reinterpret_cast<const float*>(planes + L.x3_sc) + 4 * L.x3_dc + 1;
  const int ntk = (int)(L.x3_dk / 32);
  const int tiles_m = (int)((n_local + BM - 1) / BM), tiles_n = (int)((n + BN - 1) / BN);
  if (row0 + (int64_t)tiles_m * BM > L.x3_rows) return stein_fail(STEIN_E_SHAPE, "row block exceeds the padded planes");
  const long nblk = distance_grid(symmetric, tiles_m, tiles_n);
#define X3_DIST(SYM, NP) launch_distance_x3<SYM, NP>(nblk, stream, T3, ntk, r_all, dist_out, (int)n, (int)row0, (int)n_local, (long)ld_dist, tiles_m, tiles_n, hist0, two_s, spec, spec_buf)
  switch (split_kind(dtype)) {
    case 1: if (symmetric) X3_DIST(true, 1); else X3_DIST(false, 1); break;
    default: if (symmetric) X3_DIST(true, 2); else X3_DIST(false, 2); break;
  }
#undef X3_DIST
  LAUNCH_CHECK("k_distance_x3");
  return STEIN_OK;
}

int stein_x3_contract_partial(const float* dist, int64_t ld_dist, const char* planes, const SteinLayout& L, int dtype,
                              const float* h2_dev, float* OG, float* OT, float* RS, int64_t n, int64_t d,
                              int64_t n_local, hipStream_t stream, bool upper) {
  const u16* Tt3 = reinterpret_cast<const u16*>(planes + L.x3_tt3);
  const u16* Gt3 = reinterpret_cast<const u16*>(planes + L.x3_gt3);
  const float* sc = reinterpret_cast<const float*>(planes + L.x3_sc);
  // 64-row x 512-column workgroups when the 128-column blocks of [G | theta] fill them (an even block count per matrix)
  const bool wide = L.phi_wide != 0 && !upper;   // the 64-row form has no mirrored-tile path
  const long tm = wide ? (n_local + 63) / 64 : L.tiles_m, cbk = wide ? L.cblocks / 2 : L.cblocks;
  const long nblk = tm * cbk * L.split;
#define X3_PHI(NP, RB) hipLaunchKernelGGL((k_phi_x3fs<NP, RB>), dim3((unsigned)nblk), dim3(FS_THREADS), 0, stream, dist, (long)ld_dist, Gt3, Tt3, (long)(L.x3_nk / 32), h2_dev, OG, OT, RS, (int)n, (int)d, (int)n_local, (int)tm, (int)cbk, (int)L.cblocks, (int)L.split, (int)L.jchunk, sc, (int)L.x3_dc, upper ? 1 : 0)
  switch (split_kind(dtype) * 2 + (wide ? 1 : 0)) {
    case 2: X3_PHI(1, 8); break;
    case 3: X3_PHI(1, 4); break;
    case 5: X3_PHI(2, 4); break;
    default: X3_PHI(2, 8); break;
  }
#undef X3_PHI
  LAUNCH_CHECK("k_phi_x3fs");
  return STEIN_OK;
}
