// stein_x3.hip -- both GEMMs of the SVGD step on the bf16 matrix cores with fp32-level accuracy.
//
// gfx950 runs v_mfma_f32_32x32x16_bf16 at 16x the rate of the fp32-input MFMA.  Every fp32 operand x is split
// into three bf16 terms  x = hi + mid + lo  (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); the two
// subtractions are exact in fp32, so the three terms carry 24 significant bits), and a product a*b is formed from
// the six term pairs whose weight is >= 2^-16:  lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi  -- the dropped
// pairs are below 2^-24 relative.  bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32, so
// the result has fp32-MFMA-level error at 6/16 of its matrix-core time.
//
//   k_split3        theta, score -> bf16 planes: theta row-major [3][rows][dk] (distance operands) and
//                   theta / score transposed [3][dc][nk] (the contraction's B operand, k = particle index)
//   k_distance_x3   S = T T^T from the planes; shares the fp32 kernel's epilogue (D, mirror, level-0 histogram)
//   k_phi_x3        P = exp2(c D) is split on the fly into three bf16 planes in LDS; O += P . V^T-planes
//
// Both GEMMs are "row x row" products (C[i][c] = sum_k A[i][k] B[c][k]) with k contiguous in memory for both
// operands, so one LDS image [row][k] (80-byte row stride: conflict-free ds_read_b128) serves A and B fragments:
// lane l of a 32x32x16 MFMA reads 8 consecutive k at row (l & 31), k offset 8 (l >> 5).

#include "stein_x3.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // staging registers (native vector: stays in VGPRs)

// LDS image of one operand plane: [128 rows][32 bf16] = 64-byte rows with the four 16-byte chunks of a row XOR-swizzled
// by (row >> 2) & 3.  Conflict-free for all three access shapes (banking rules of MI355X_MICROARCH.md, LDS):
//   ds_read_b128 fragments (16-lane groups {0-3,12-15,20-27}..., 64 banks): (4 row + chunk') mod 16 distinct in a group
//   ds_write_b128 staging  (8 consecutive lanes = 2 rows x 4 chunks, 32 banks): even row -> bytes 0..63, odd -> 64..127
//   ds_write_b64 of P      (16 consecutive lanes = 2 rows x 8 half-chunks): same split
// (an 80-byte padded row made every write 2-way conflicted: SQ_LDS_BANK_CONFLICT = 1/3 of the LDS cycles.)
#ifdef STEIN_STAMPS   // diagnostic build only (never shipped): per-phase cycle sums of wave 0 of every workgroup
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
extern "C" int stein_debug_stamps(u64* host_out, int reset) {
  if (host_out && hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(u64) * 8) != hipSuccess) return -1;
  if (reset) { u64 z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#else
#define STAMP(k) do {} while (0)
#endif

constexpr int XROW = 64;                 // bytes per LDS row: 32 bf16
constexpr int XPLANE = 128 * XROW;       // one 128-row plane of a tile: 8192 B
constexpr int XOPER = 3 * XPLANE;        // three planes of one operand: 24576 B
__device__ __forceinline__ int xswz(int row, int chunk) { return (chunk ^ ((row >> 2) & 3)) * 16; }

// ------------------------------------------------------------------------------------------------
// splitting
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(u32 b) { return __uint_as_float(b << 16); }

__device__ __forceinline__ u32 cvt_pk_bf16(float lo, float hi) {   // round-to-nearest-even, lo -> bits 15:0
  u32 r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}

// two fp32 values -> three packed bf16 pairs (x in the low half)
__device__ __forceinline__ void split3_pair(float x, float y, u32& hi, u32& mid, u32& lo) {
  hi = cvt_pk_bf16(x, y);
  const float rx = x - __uint_as_float(hi << 16), ry = y - __uint_as_float(hi & 0xffff0000u);   // exact
  mid = cvt_pk_bf16(rx, ry);
  const float sx = rx - __uint_as_float(mid << 16), sy = ry - __uint_as_float(mid & 0xffff0000u);  // exact
  lo = cvt_pk_bf16(sx, sy);
}

// One 64x64 tile of X per workgroup.  R (row-major planes, [3][r_rows][dk]) and/or Tt (transposed planes,
// [3][dc][nk]) may be NULL.  The grid covers the padded extents; out-of-range source entries are zero.
__global__ __launch_bounds__(256) void k_split3(const float* __restrict__ X, int n, int d, u16* __restrict__ R,
                                                long r_rows, int dk, u16* __restrict__ Tt, int dc, long nk) {
  __shared__ u16 tile[3][64][66];
  const int t = threadIdx.x;
  const int row0 = blockIdx.y * 64, col0 = blockIdx.x * 64;
  const int lr = t >> 4, lc = (t & 15) * 4;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int row = row0 + lr + 16 * p, col = col0 + lc;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (row < n && col + q < d) ? X[(size_t)row * d + col + q] : 0.f;
    u32 h0, m0, l0, h1, m1, l1;
    split3_pair(v[0], v[1], h0, m0, l0);
    split3_pair(v[2], v[3], h1, m1, l1);
    if (R && row < r_rows && col < dk) {   // dk % 32 == 0 and col % 4 == 0: the 4 entries stay inside the row
      const size_t o = (size_t)row * dk + col;
      *reinterpret_cast<uint2*>(R + o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(R + (size_t)r_rows * dk + o) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(R + 2 * (size_t)r_rows * dk + o) = make_uint2(l0, l1);
    }
    if (Tt) {
      const int rr = lr + 16 * p;
      tile[0][lc + 0][rr] = (u16)h0; tile[0][lc + 1][rr] = (u16)(h0 >> 16);
      tile[0][lc + 2][rr] = (u16)h1; tile[0][lc + 3][rr] = (u16)(h1 >> 16);
      tile[1][lc + 0][rr] = (u16)m0; tile[1][lc + 1][rr] = (u16)(m0 >> 16);
      tile[1][lc + 2][rr] = (u16)m1; tile[1][lc + 3][rr] = (u16)(m1 >> 16);
      tile[2][lc + 0][rr] = (u16)l0; tile[2][lc + 1][rr] = (u16)(l0 >> 16);
      tile[2][lc + 2][rr] = (u16)l1; tile[2][lc + 3][rr] = (u16)(l1 >> 16);
    }
  }
  if (!Tt) return;
  __syncthreads();
  // transposed store: thread -> (parameter c = t >> 2, 16 particles starting at 16 (t & 3))
  const int c = col0 + (t >> 2), j = row0 + (t & 3) * 16;
  if (c < dc && j < nk) {   // nk % 32 == 0 and j % 16 == 0: 16 entries stay inside the row
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      u32 w[8];
#pragma unroll
      for (int q = 0; q < 8; ++q)
        w[q] = (u32)tile[s][t >> 2][(t & 3) * 16 + 2 * q] | ((u32)tile[s][t >> 2][(t & 3) * 16 + 2 * q + 1] << 16);
      u16* dst = Tt + (size_t)s * dc * nk + (size_t)c * nk + j;
      *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
      *reinterpret_cast<uint4*>(dst + 8) = make_uint4(w[4], w[5], w[6], w[7]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// shared pieces of the two MFMA kernels
// ------------------------------------------------------------------------------------------------
// global -> registers: one operand tile = 3 planes x 128 rows x 32 k (64 B per row) = 3 x 512 16-byte chunks;
// thread t owns chunks t and t + 256 (64 rows further down) of every plane.  Rows and k are padded in memory: no
// bounds checks.  `tile` is wave-uniform (scalar address arithmetic); `toff` = (t >> 2) * ld + (t & 3) * 8 is the
// only per-lane part and is loop invariant.
__device__ __forceinline__ void x3_load_tile(const u16* __restrict__ tile, size_t plane_stride, long ld, u32 toff,
                                             u32x4 (&reg)[6]) {
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int q = 0; q < 2; ++q)
      reg[s * 2 + q] = *reinterpret_cast<const u32x4*>(tile + s * plane_stride + (size_t)q * 64 * ld + toff);
}

__device__ __forceinline__ void x3_store_tile(unsigned char* oper, int t, const u32x4 (&reg)[6]) {
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int chunk = t + 256 * q;
      *reinterpret_cast<u32x4*>(oper + s * XPLANE + (chunk >> 2) * XROW + xswz(chunk >> 2, chunk & 3)) = reg[s * 2 + q];
    }
}

// one 32-deep k tile already in LDS: 2 k16 steps x (2x2 tiles) x 6 products
__device__ __forceinline__ void x3_mma_tile(const unsigned char* As, const unsigned char* Bs, int wy, int wx, int lane,
                                            f32x16 (&acc)[2][2]) {
  const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 a[2][3], b[2][3];
    const int co = xswz(l31, 2 * ks + h);   // (row >> 2) & 3 only depends on row mod 16 = l31 mod 16
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        a[i][s] = *reinterpret_cast<const bf16x8*>(As + s * XPLANE + (wy * 64 + i * 32 + l31) * XROW + co);
        b[i][s] = *reinterpret_cast<const bf16x8*>(Bs + s * XPLANE + (wx * 64 + i * 32 + l31) * XROW + co);
      }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        f32x16 c = acc[i][j];
        // smallest terms first (planes: 0 = hi, 1 = mid, 2 = lo)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
        acc[i][j] = c;
      }
  }
}

// ------------------------------------------------------------------------------------------------
// k_distance_x3
// ------------------------------------------------------------------------------------------------
template <bool SYM>
__global__ __launch_bounds__(NTHREADS, 2) void k_distance_x3(const u16* __restrict__ T3, size_t plane_stride, int dk,
                                                             const float* __restrict__ r, float* __restrict__ D, int n,
                                                             int row0, int n_local, long ldD, int tiles_m, int tiles_n,
                                                             u64* __restrict__ hist0) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * XOPER];
  unsigned char* As = smem;
  unsigned char* Bs = smem + XOPER;
  int tile_m, tile_n;
  distance_tile<SYM>(xcd_remap(blockIdx.x, gridDim.x), tiles_m, tiles_n, tile_m, tile_n);
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wy = wid >> 1, wx = wid & 1;
  const long arow0 = row0 + (long)tile_m * BM, brow0 = (long)tile_n * BN;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  u32x4 ra[6], rb[6];
  const u32 toff = (u32)(t >> 2) * (u32)dk + (u32)(t & 3) * 8u;
  const u16* __restrict__ pa = T3 + (size_t)arow0 * dk;   // wave-uniform tile origins
  const u16* __restrict__ pb = T3 + (size_t)brow0 * dk;
  x3_load_tile(pa, plane_stride, dk, toff, ra);
  x3_load_tile(pb, plane_stride, dk, toff, rb);
  for (int k0 = 0; k0 < dk; k0 += BK) {
    x3_store_tile(As, t, ra);
    x3_store_tile(Bs, t, rb);
    __syncthreads();
    if (k0 + BK < dk) {
      x3_load_tile(pa + k0 + BK, plane_stride, dk, toff, ra);
      x3_load_tile(pb + k0 + BK, plane_stride, dk, toff, rb);
    }
    x3_mma_tile(As, Bs, wy, wx, lane, acc);
    __syncthreads();
  }
  distance_epilogue<SYM>(acc, reinterpret_cast<u32*>(smem), r, D, n, row0, n_local, ldD, tile_m, tile_n, hist0);
}

// ------------------------------------------------------------------------------------------------
// k_phi_x3
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS, 2) void k_phi_x3(const float* __restrict__ D, long ldD,
                                                        const u16* __restrict__ Gt3, const u16* __restrict__ Tt3,
                                                        size_t plane_stride, long nk, const float* __restrict__ h2p,
                                                        float* __restrict__ OG, float* __restrict__ OT,
                                                        float* __restrict__ RS, int n, int d, int n_local, int tiles_m,
                                                        int cblocks, int split, int jchunk) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * XOPER];
  unsigned char* As = smem;
  unsigned char* Bs = smem + XOPER;

  const int logical = xcd_remap(blockIdx.x, gridDim.x);
  const int ncb = 2 * cblocks;
  const int cb = logical % ncb;
  const int tile_m = (logical / ncb) % tiles_m;
  const int z = logical / (ncb * tiles_m);
  const bool isT = cb >= cblocks;
  const u16* __restrict__ V3 = isT ? Tt3 : Gt3;
  float* __restrict__ O = isT ? OT : OG;
  const int c0 = (isT ? cb - cblocks : cb) * BN;
  const int jbeg = z * jchunk;
  const int jend = min(n, jbeg + jchunk);

  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int wy = wid >> 1, wx = wid & 1;
  const int lr = t >> 3, lc = (t & 7) * 4;   // P staging: rows lr + 32p, 4 consecutive j
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

  float4 rd[4];
  u32x4 rv[6];
  const u32 voff = (u32)(t >> 2) * (u32)nk + (u32)(t & 3) * 8u;
  const u16* __restrict__ pv0 = V3 + (size_t)c0 * nk;   // wave-uniform origin of this column block's rows
  // D rows are clamped to the block (rows past n_local only feed accumulator rows that are never stored), so the
  // loads need no predicate; ldD >= roundup(n, 64) keeps j0 + 31 inside the row.
  u32 doff[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) doff[p] = (u32)min(i0 + lr + 32 * p, n_local - 1) * (u32)ldD + (u32)lc;
  auto load_d = [&](int j0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) rd[p] = *reinterpret_cast<const float4*>(D + j0 + doff[p]);
  };
  if (jbeg < jend) {
    load_d(jbeg);
    x3_load_tile(pv0 + jbeg, plane_stride, nk, voff, rv);
  }
#ifdef STEIN_STAMPS
  u64 st_acc[6] = {0, 0, 0, 0, 0, 0};
  u64 st_last = __builtin_amdgcn_s_memtime();
#endif
  for (int j0 = jbeg; j0 < jend; j0 += BK) {
    const bool full = j0 + BK <= jend;   // wave-uniform: only the last tile of the last split can be ragged
#ifdef STEIN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMP(0);   // waiting for the prefetched tile
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      float4 pv;
      pv.x = __builtin_amdgcn_exp2f(cexp * rd[p].x);
      pv.y = __builtin_amdgcn_exp2f(cexp * rd[p].y);
      pv.z = __builtin_amdgcn_exp2f(cexp * rd[p].z);
      pv.w = __builtin_amdgcn_exp2f(cexp * rd[p].w);
      if (!full) {   // columns past jend hold whatever the padding holds: force P = 0 there
        const int j = j0 + lc;
        pv.x = (j + 0 < jend) ? pv.x : 0.f;
        pv.y = (j + 1 < jend) ? pv.y : 0.f;
        pv.z = (j + 2 < jend) ? pv.z : 0.f;
        pv.w = (j + 3 < jend) ? pv.w : 0.f;
      }
      rs[p] += (pv.x + pv.y) + (pv.z + pv.w);
      u32 h0, m0, l0, h1, m1, l1;
      split3_pair(pv.x, pv.y, h0, m0, l0);
      split3_pair(pv.z, pv.w, h1, m1, l1);
      unsigned char* dst = As + (lr + 32 * p) * XROW + xswz(lr, lc >> 3) + (lc & 4) * 2;   // (lr + 32p) >> 2 & 3 == lr >> 2 & 3
      *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(dst + XPLANE) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(dst + 2 * XPLANE) = make_uint2(l0, l1);
    }
    x3_store_tile(Bs, t, rv);
    STAMP(1);   // exp + split + LDS writes
    __syncthreads();
    STAMP(2);   // barrier 1
    if (j0 + BK < jend) {
      load_d(j0 + BK);
      x3_load_tile(pv0 + j0 + BK, plane_stride, nk, voff, rv);
    }
    STAMP(3);   // prefetch issue
    x3_mma_tile(As, Bs, wy, wx, lane, acc);
    STAMP(4);   // LDS fragment reads + MFMAs
    __syncthreads();
    STAMP(5);   // barrier 2
  }
#ifdef STEIN_STAMPS
  if (t == 0)
    for (int k = 0; k < 6; ++k) atomicAdd(&g_stamps[k], st_acc[k]);
  if (t == 0) atomicAdd(&g_stamps[7], 1ull);
#endif
  phi_epilogue(acc, rs, O + (size_t)z * n_local * d, RS + (size_t)z * n_local, d, n_local, i0, c0, cb == 0);
}

// ================================================================================================
// host side
// ================================================================================================
int stein_x3_split(const float* theta_all, const float* score_all, int64_t n, int64_t d, const SteinLayout& L,
                   char* planes, hipStream_t stream) {
  u16* T3 = reinterpret_cast<u16*>(planes + L.x3_t3);
  u16* Tt3 = reinterpret_cast<u16*>(planes + L.x3_tt3);
  u16* Gt3 = reinterpret_cast<u16*>(planes + L.x3_gt3);
  const int64_t rows = L.x3_rows > L.x3_nk ? L.x3_rows : L.x3_nk;   // particle extent to cover (both multiples of 32)
  const int64_t cols = L.x3_dk > L.x3_dc ? L.x3_dk : L.x3_dc;      // parameter extent
  const dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + 63) / 64));
  hipLaunchKernelGGL(k_split3, grid, dim3(256), 0, stream, theta_all, (int)n, (int)d, T3, (long)L.x3_rows,
                     (int)L.x3_dk, Tt3, (int)L.x3_dc, (long)L.x3_nk);
  LAUNCH_CHECK("k_split3(theta)");
  const dim3 grid_g((unsigned)((L.x3_dc + 63) / 64), (unsigned)((L.x3_nk + 63) / 64));
  hipLaunchKernelGGL(k_split3, grid_g, dim3(256), 0, stream, score_all, (int)n, (int)d, (u16*)nullptr, 0l, 0, Gt3,
                     (int)L.x3_dc, (long)L.x3_nk);
  LAUNCH_CHECK("k_split3(score)");
  return STEIN_OK;
}

int stein_x3_distance(const char* planes, const SteinLayout& L, const float* r_all, float* dist_out, int64_t n,
                      int64_t d, int64_t row0, int64_t n_local, int64_t ld_dist, u64* hist0, bool symmetric,
                      hipStream_t stream) {
  (void)d;
  const u16* T3 = reinterpret_cast<const u16*>(planes + L.x3_t3);
  const size_t plane_stride = (size_t)L.x3_rows * L.x3_dk;
  const int tiles_m = (int)((n_local + BM - 1) / BM), tiles_n = (int)((n + BN - 1) / BN);
  if (row0 + (int64_t)tiles_m * BM > L.x3_rows) return stein_fail(STEIN_E_SHAPE, "row block exceeds the padded planes");
  const long nblk = symmetric ? (long)tiles_n * (tiles_n + 1) / 2 : (long)tiles_m * tiles_n;
  if (symmetric)
    hipLaunchKernelGGL((k_distance_x3<true>), dim3((unsigned)nblk), dim3(NTHREADS), 0, stream, T3, plane_stride,
                       (int)L.x3_dk, r_all, dist_out, (int)n, (int)row0, (int)n_local, (long)ld_dist, tiles_m, tiles_n,
                       hist0);
  else
    hipLaunchKernelGGL((k_distance_x3<false>), dim3((unsigned)nblk), dim3(NTHREADS), 0, stream, T3, plane_stride,
                       (int)L.x3_dk, r_all, dist_out, (int)n, (int)row0, (int)n_local, (long)ld_dist, tiles_m, tiles_n,
                       hist0);
  LAUNCH_CHECK("k_distance_x3");
  return STEIN_OK;
}

int stein_x3_contract_partial(const float* dist, int64_t ld_dist, const char* planes, const SteinLayout& L,
                              const float* h2_dev, float* OG, float* OT, float* RS, int64_t n, int64_t d,
                              int64_t n_local, hipStream_t stream) {
  const u16* Tt3 = reinterpret_cast<const u16*>(planes + L.x3_tt3);
  const u16* Gt3 = reinterpret_cast<const u16*>(planes + L.x3_gt3);
  const size_t plane_stride = (size_t)L.x3_dc * L.x3_nk;
  const long nblk = (long)L.tiles_m * 2 * L.cblocks * L.split;
  hipLaunchKernelGGL(k_phi_x3, dim3((unsigned)nblk), dim3(NTHREADS), 0, stream, dist, (long)ld_dist, Gt3, Tt3,
                     plane_stride, (long)L.x3_nk, h2_dev, OG, OT, RS, (int)n, (int)d, (int)n_local, (int)L.tiles_m,
                     (int)L.cblocks, (int)L.split, (int)L.jchunk);
  LAUNCH_CHECK("k_phi_x3");
  return STEIN_OK;
}
