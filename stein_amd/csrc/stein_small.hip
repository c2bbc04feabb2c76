// stein_small.hip -- the whole phi computation in ONE kernel for the particle counts the reference's own examples
// use (n = 20 ... 100): at that size the tiled pipeline is about ten launches of a few microseconds each and nothing
// else.  Every workgroup redundantly computes the row norms, the n x n distance matrix (kept in LDS), its exact
// median (the same 3-level radix select on the order-preserving key, here on LDS histograms) and K, then forms phi
// for its own block of SM_COLS parameter columns; one small reduction kernel sums the per-workgroup |phi|^2.
//
//   D_ij = r_i + r_j - 2 theta_i . theta_j  (fp32)          stein/kernels/abstract_kernel.py:33-35
//   med  = median of all n^2 entries, even count -> mean     stein/utilities/compute_median.py:4-16
//   h2   = sqrt(med / ln n)^2                                abstract_kernel.py:40, squared_exponential_kernel.py:22
//   K    = exp(-D / h2 / 2);  dK = (rowsum(K) theta - K theta) / h2     squared_exponential_kernel.py:22-35
//   phi  = (K G + dK) / n                                    stein/samplers/abstract_stein_sampler.py:100-105
#include "stein_common.h"

#include <type_traits>

constexpr int SM_MAXN = 160;      // particles (the distance matrix lives in LDS: 160 x 160 floats = 100 KB of the CU's 160)
constexpr int SM_STG = (SM_MAXN * 32 + 1023) / 1024;   // staged entries per thread (5)
constexpr int SM_THREADS = 1024;
constexpr int SM_COLS = 32;       // parameter columns per workgroup (their theta / score columns are staged in LDS)
constexpr int SM_CK = 32;         // granule of the theta chunk staged per pass of the distance loop: a chunk is `ck` columns,
                                  // the multiple of 32 (<= 256) for which n * ck entries still fit SM_STG registers per thread
static_assert(SM_COLS == SM_CK, "the theta chunk buffer doubles as the phi stage's theta block");
static inline int small_chunk(int64_t n, int64_t d) {
  int64_t ck = (int64_t)SM_STG * SM_THREADS / n / SM_CK * SM_CK;
  ck = ck < SM_CK ? SM_CK : (ck > 256 ? 256 : ck);
  const int64_t dr = (d + SM_CK - 1) / SM_CK * SM_CK;   // no wider than the matrix
  return (int)(ck < dr ? ck : dr);
}

#ifdef STEIN_STAMPS   // diagnostic build only (scratch/small_stamps.py): cycles per phase of workgroup 0
__device__ unsigned long long g_small_stamps[16];
extern "C" int stein_debug_small(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_small_stamps), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#define SST(k)                                                                     \
  do {                                                                             \
    __syncthreads();                                                               \
    if (threadIdx.x == 0 && blockIdx.x == 0) {                                     \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                \
      g_small_stamps[k] = now_ - sst_last;                                         \
      sst_last = now_;                                                             \
    }                                                                              \
  } while (0)
#else
#define SST(k) do {} while (0)
#endif

__global__ __launch_bounds__(SM_THREADS) void k_svgd_small(const float* __restrict__ T, const float* __restrict__ G, int n,
                                                           int d, float ln_n, float* __restrict__ phi,
                                                           float* __restrict__ h2_out, double* __restrict__ sqpart,
                                                           float* __restrict__ K_out, float* __restrict__ dK_out,
                                                           double* __restrict__ sq_total, int ck) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int ldn = (n + 3) & ~3;   // rows start 16-byte aligned (the phi stage reads K four columns at a time)
  float* Dm = sm;                                   // [n][ldn]  distances, then K
  float* tc = Dm + (size_t)n * ldn;                 // [n][ck + 1] theta chunk
  float* rn = tc + (size_t)n * (ck + 1);            // [n] rowsum(K)
  u32* hist = reinterpret_cast<u32*>(rn + n);       // [2][2048]
  __shared__ u32 s_prefix[2], s_rank[2], s_div, s_wsum[SM_THREADS / 64];
  __shared__ float s_h2;
  __shared__ double s_red[SM_THREADS / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
#ifdef STEIN_STAMPS
  unsigned long long sst_last = __builtin_amdgcn_s_memtime();
#endif

  // ---- S = T T^T, theta staged SM_CK columns at a time.  Thread (bi, bj) = (t / 32, t % 32) owns the entries
  //      (bi + 32 r, bj + 32 s), r, s < R = ceil(n / 32): 2 R LDS reads feed R^2 FMAs per column, the row reads are
  //      broadcasts and the column reads of 32 consecutive lanes hit 32 different banks (row stride 33).  (One thread
  //      per (i <= j) pair read two words per FMA: 23 MB of LDS reads, 90 us, for n = 20, d = 303.) ----
  for (int b = t; b < STEIN_HIST_BINS; b += SM_THREADS) hist[b] = 0u;   // level-0 histogram, filled by the distance stage
  const int bi = t >> 5, bj = t & 31;
  const int R = (n + 31) >> 5;
  const bool mine = bi < n && bj < n;
  // RR = R as a compile-time constant: the inner loop is 2 RR LDS reads (fixed row offsets) and RR^2 FMAs per column
  auto distances = [&](auto rr_tag) {
    constexpr int RR = decltype(rr_tag)::value;
    float acc[RR][RR], na[RR], nb[RR];   // na / nb: squared norms of this thread's rows / columns (abstract_kernel.py:34),
#pragma unroll                          // summed in column order by every thread that needs them -- identical everywhere
    for (int r = 0; r < RR; ++r) {
      na[r] = nb[r] = 0.f;
#pragma unroll
      for (int q = 0; q < RR; ++q) acc[r][q] = 0.f;
    }
    const float* pa[RR];
    const float* pb[RR];
#pragma unroll
    for (int r = 0; r < RR; ++r) {   // rows past n read row n - 1 (harmless, never stored)
      pa[r] = tc + min(bi + 32 * r, n - 1) * (ck + 1);
      pb[r] = tc + min(bj + 32 * r, n - 1) * (ck + 1);
    }
    // the next chunk's theta entries travel from global memory into registers while this chunk is multiplied
    // (n * ck <= SM_STG * SM_THREADS entries: at most SM_STG per thread -- small_chunk())
    static_assert(SM_MAXN * SM_CK <= SM_STG * SM_THREADS, "SM_STG staged entries per thread");
    float pre[SM_STG];
    auto fetch = [&](int c0) {
#pragma unroll
      for (int k = 0; k < SM_STG; ++k) {
        const int e = t + k * SM_THREADS, i = e / ck, c = e - i * ck;
        pre[k] = (e < n * ck && c0 + c < d) ? T[(size_t)i * d + c0 + c] : 0.f;   // zero-filled past d
      }
    };
    fetch(0);
    for (int c0 = 0; c0 < d; c0 += ck) {
      __syncthreads();
#pragma unroll
      for (int k = 0; k < SM_STG; ++k) {
        const int e = t + k * SM_THREADS;
        if (e < n * ck) tc[e + e / ck] = pre[k];   // row e / ck, column e % ck, row stride ck + 1
      }
      if (c0 + ck < d) fetch(c0 + ck);
      __syncthreads();
      if (mine) {
        const int cend = min(ck, d - c0);
#pragma unroll 8
        for (int c = 0; c < cend; ++c) {
          float a[RR], b[RR];
#pragma unroll
          for (int r = 0; r < RR; ++r) {
            a[r] = pa[r][c]; b[r] = pb[r][c];
            na[r] = fmaf(a[r], a[r], na[r]);
            nb[r] = fmaf(b[r], b[r], nb[r]);
          }
#pragma unroll
          for (int r = 0; r < RR; ++r)
#pragma unroll
            for (int q = 0; q < RR; ++q) acc[r][q] = fmaf(a[r], b[q], acc[r][q]);
        }
      }
    }
    if (mine) {
#pragma unroll
      for (int r = 0; r < RR; ++r)
#pragma unroll
        for (int q = 0; q < RR; ++q) {
          const int i = bi + 32 * r, j = bj + 32 * q;
          if (i < n && j < n) {
            const float v = (na[r] + nb[q]) - 2.f * acc[r][q];      // abstract_kernel.py:35
            Dm[i * ldn + j] = v;
            atomicAdd(&hist[f32_key(v) >> 21], 1u);                  // level 0 of the radix select, while the value is at hand
          }
        }
    }
  };
  switch (R) {   // workgroup-uniform
    case 1: distances(std::integral_constant<int, 1>()); break;
    case 2: distances(std::integral_constant<int, 2>()); break;
    case 3: distances(std::integral_constant<int, 3>()); break;
    case 4: distances(std::integral_constant<int, 4>()); break;
    default: distances(std::integral_constant<int, 5>()); break;
  }
  SST(0);   // distances (+ row norms, level-0 histogram)
  // this workgroup's theta / score columns for the phi stage: requested now, so the loads fly during the median
  const int cw0 = blockIdx.x * SM_COLS;
  const int ncols = min(SM_COLS, d - cw0);
  float preT[SM_STG], preG[SM_STG];
#pragma unroll
  for (int k = 0; k < SM_STG; ++k) {
    const int e = t + k * SM_THREADS, j = e / SM_COLS, cl = e % SM_COLS;
    const bool ok = e < n * SM_COLS && cl < ncols;
    preT[k] = ok ? T[(size_t)j * d + cw0 + cl] : 0.f;
    preG[k] = ok ? G[(size_t)j * d + cw0 + cl] : 0.f;
  }
  // ---- exact median of the n^2 entries: 3-level radix select (11 / 11 / 10 bits), two targets for an even count ----
  const u32 total = (u32)n * (u32)n;
  if (t == 0) {
    s_rank[0] = (total & 1u) ? total / 2 : total / 2 - 1;
    s_rank[1] = total / 2;
    s_prefix[0] = s_prefix[1] = 0u;
    s_div = 0u;
  }
  __syncthreads();
  for (int level = 0; level < 3; ++level) {
    const int shift = level == 0 ? 21 : (level == 1 ? 10 : 0), bits = level == 2 ? 10 : 11;
    const u32 pa = s_prefix[0], pb = s_prefix[1];
    const bool two = s_div != 0u;
    if (level > 0) {
      for (int b = t; b < (two ? 2 : 1) * STEIN_HIST_BINS; b += SM_THREADS) hist[b] = 0u;
      __syncthreads();
    }
    for (int r = 0; r < (level > 0 ? R : 0); ++r)   // the distance stage's entry map: (bi + 32 r, bj + 32 q), no divisions
      for (int q = 0; q < R; ++q) {
        const int i = bi + 32 * r, j = bj + 32 * q;
        if (i < n && j < n) {
          const u32 key = f32_key(Dm[i * ldn + j]);
          const u32 digit = (key >> shift) & ((1u << bits) - 1u);
          const u32 hi = level == 0 ? 0u : key >> (shift + bits);
          if (level == 0 || hi == pa) atomicAdd(&hist[digit], 1u);
          if (two && hi == pb) atomicAdd(&hist[STEIN_HIST_BINS + digit], 1u);
        }
      }
    __syncthreads();
    // locate each target's digit with a workgroup-wide prefix sum over the 2048 bins (two bins per thread); a
    // single thread walking the bins took 65 us per level.  While the targets share a prefix (almost always until the
    // last level) one pass over the shared histogram serves both.
    for (int pass = 0; pass < (two ? 2 : 1); ++pass) {
      const u32* h = hist + pass * STEIN_HIST_BINS;
      const u32 c0 = h[2 * t], c1 = h[2 * t + 1];
      u32 incl = c0 + c1;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const u32 v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      if (lane == 63) s_wsum[wave] = incl;
      __syncthreads();
      u32 base = 0u;
      for (int w = 0; w < wave; ++w) base += s_wsum[w];
      const u32 excl = base + incl - (c0 + c1);
      const u32 rank0 = s_rank[0], rank1 = s_rank[1];
      __syncthreads();   // everybody has read s_rank / s_wsum before they change
#pragma unroll
      for (int tg = 0; tg < 2; ++tg) {
        if (two && tg != pass) continue;
        const u32 rank = tg ? rank1 : rank0;
        if (rank >= excl && rank < excl + c0 + c1) {   // exactly one thread (the counts sum to more than rank)
          const u32 b = rank < excl + c0 ? 2u * t : 2u * t + 1u;
          s_prefix[tg] = (s_prefix[tg] << bits) | b;
          s_rank[tg] = rank - (b & 1u ? excl + c0 : excl);
        }
      }
      __syncthreads();
    }
    if (t == 0) s_div = s_prefix[0] != s_prefix[1] ? 1u : 0u;
    __syncthreads();
    SST(1 + level);   // median levels (level 0: locate only)
  }
  if (t == 0) {
    const float lo = key_f32(s_prefix[0]), hi = key_f32(s_prefix[1]);
    const float med = (total & 1u) ? lo : 0.5f * (lo + hi);        // compute_median.py:12-15
    const float bw = sqrtf(med / ln_n);                             // abstract_kernel.py:40
    s_h2 = bw * bw;                                                 // squared_exponential_kernel.py:22
    if (blockIdx.x == 0) *h2_out = s_h2;
  }
  __syncthreads();
  const float h2 = s_h2;
  SST(4);   // bandwidth
  // ---- K in place (exp(-D / h2 / 2) = exp2(kc D), as the tiled kernels form it), rowsum(K): the 32 lanes that share
  //      bi hold 32 columns of a row, so a row's sum is this thread's R entries plus one shuffle reduction (a fixed order) ----
  const float kc = -1.44269504088896341f / (2.f * h2);
  for (int r = 0; r < R; ++r) {
    const int i = bi + 32 * r;
    float s = 0.f;
    for (int q = 0; q < R; ++q) {
      const int j = bj + 32 * q;
      float k = 0.f;
      if (i < n && j < n) {
        k = __builtin_amdgcn_exp2f(kc * Dm[i * ldn + j]);
        Dm[i * ldn + j] = k;
        if (K_out && blockIdx.x == 0) K_out[(size_t)i * n + j] = k;
      }
      s += k;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (bj == 0 && i < n) rn[i] = s;
  }
  __syncthreads();
  SST(5);   // K + rowsum
  // ---- phi for this workgroup's SM_COLS columns.  Their theta / score columns are staged in LDS first (the chunk
  //      buffer and the histograms are free now): read from global memory inside the j loop, every iteration paid
  //      the L2 latency (30+ us at n = 100).  thread -> (row i, column c), lanes along c ----
  float* ts = tc;                                    // [n][SM_COLS + 1]
  float* gs = reinterpret_cast<float*>(hist);        // [n][SM_COLS]   (the region is sized for whichever is larger)
#pragma unroll
  for (int k = 0; k < SM_STG; ++k) {   // fetched before the median (below the distance stage), stored now
    const int e = t + k * SM_THREADS;
    if (e < n * SM_COLS) {
      ts[(e / SM_COLS) * (SM_COLS + 1) + e % SM_COLS] = preT[k];
      gs[e] = preG[k];
    }
  }
  __syncthreads();
  // A wave owns the rows wave + 16 r (r < RB = ceil(n / 16)).  All RB rows advance together: per four j one 16-byte broadcast read of K per row and the staged score /
  // theta entries once, for 8 RB FMAs -- one row at a time read three words per two FMAs and the stage was bound by LDS
  // bandwidth (61 k of the kernel's 144 k cycles at n = 128).
  double sq = 0.0;
  const float fn = (float)n;
  // Lanes: CLW columns x 64 / CLW slices of the j range (slice starts are multiples of 4).  CLW = 32 unless the
  // workgroup has few columns (d = 1 ... 16: the reference's regression examples), where 32 column lanes would spend
  // the stage multiplying padding.
  auto phi_rows = [&](auto rb_tag, auto clw_tag) {
    constexpr int RB = decltype(rb_tag)::value, CLW = decltype(clw_tag)::value, JS = 64 / CLW;
    const int cl = lane & (CLW - 1), jh = lane / CLW;
    const int jq = (((n + JS - 1) / JS) + 3) & ~3;
    const int j0 = min(n, jh * jq), j1 = min(n, j0 + jq);
    float kg[RB], kt[RB];
    const float* krow[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      kg[r] = kt[r] = 0.f;
      krow[r] = Dm + min(wave + (SM_THREADS / 64) * r, n - 1) * ldn;   // rows past n repeat row n - 1 (never stored)
    }
    int j = j0;
    for (; j + 3 < j1; j += 4) {
      float4 k4[RB];
#pragma unroll
      for (int r = 0; r < RB; ++r) k4[r] = *reinterpret_cast<const float4*>(krow[r] + j);
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float g = gs[(j + jj) * SM_COLS + cl], tt = ts[(j + jj) * (SM_COLS + 1) + cl];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
          const float k = jj == 0 ? k4[r].x : (jj == 1 ? k4[r].y : (jj == 2 ? k4[r].z : k4[r].w));
          kg[r] = fmaf(k, g, kg[r]);
          kt[r] = fmaf(k, tt, kt[r]);
        }
      }
    }
    for (; j < j1; ++j) {
      const float g = gs[j * SM_COLS + cl], tt = ts[j * (SM_COLS + 1) + cl];
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const float k = krow[r][j];
        kg[r] = fmaf(k, g, kg[r]);
        kt[r] = fmaf(k, tt, kt[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      float sg = kg[r], st = kt[r];
#pragma unroll
      for (int o = CLW; o < 64; o <<= 1) { sg += __shfl_xor(sg, o); st += __shfl_xor(st, o); }
      const int i = wave + (SM_THREADS / 64) * r;
      if (jh == 0 && cl < ncols && i < n) {
        const int c = cw0 + cl;
        const float dk = (rn[i] * ts[i * (SM_COLS + 1) + cl] - st) / h2;   // squared_exponential_kernel.py:28-35
        const float ph = (sg + dk) / fn;                                    // abstract_stein_sampler.py:105
        phi[(size_t)i * d + c] = ph;
        if (dK_out) dK_out[(size_t)i * d + c] = dk;
        sq += (double)ph * (double)ph;
      }
    }
  };
  auto phi_cols = [&](auto rb_tag) {   // workgroup-uniform choices
    if (ncols <= 8) phi_rows(rb_tag, std::integral_constant<int, 8>());
    else if (ncols <= 16) phi_rows(rb_tag, std::integral_constant<int, 16>());
    else phi_rows(rb_tag, std::integral_constant<int, 32>());
  };
  if (n <= 2 * (SM_THREADS / 64)) phi_cols(std::integral_constant<int, 2>());
  else if (n <= 4 * (SM_THREADS / 64)) phi_cols(std::integral_constant<int, 4>());
  else if (n <= 8 * (SM_THREADS / 64)) phi_cols(std::integral_constant<int, 8>());
  else phi_cols(std::integral_constant<int, 10>());
  static_assert(SM_MAXN <= 10 * (SM_THREADS / 64), "rows per wave");
  SST(6);   // stage + phi
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
  if (lane == 0) s_red[wave] = sq;
  __syncthreads();
  if (t == 0) {
    double s = 0.0;
    for (int w = 0; w < SM_THREADS / 64; ++w) s += s_red[w];
    sqpart[blockIdx.x] = s;
    if (gridDim.x == 1 && sq_total) *sq_total = s;   // d <= 32: this workgroup's partial is the whole |phi|^2
  }
}

// the caller (stein_svgd_phi) checks applicability with stein_small_ok first
bool stein_small_ok(int64_t n, int64_t d, int dtype) {
  return dtype == STEIN_F32 && n >= 2 && n <= SM_MAXN && n * n * d <= 2200000ll /* measured crossover with the tiled kernels: scratch/small_vs_tiled2.py */ && (d + SM_COLS - 1) / SM_COLS <= 1024;
}

int stein_small_phi(const float* theta, const float* score, int64_t n, int64_t d, float* phi, float* h2_out,
                    double* sqpart, float* K_out, float* dK_out, int* nparts, double* sqnorm_out, hipStream_t stream) {
  const int blocks = (int)((d + SM_COLS - 1) / SM_COLS);
  // distances | theta chunk | rowsum | histograms, later the score block [n][32] (whichever is larger)
  const size_t hist_b = 2 * STEIN_HIST_BINS * sizeof(u32), gs_b = (size_t)n * SM_COLS * sizeof(float);
  const size_t lds = ((size_t)n * ((n + 3) & ~(int64_t)3) + (size_t)n * (small_chunk(n, d) + 1) + n) * sizeof(float) + (hist_b > gs_b ? hist_b : gs_b);
  // more than the default 64 KB of dynamic LDS: the attribute belongs to the function ON ONE DEVICE, so it is set once per
  // device the library launches on (a flag per device id; racing threads at worst set it twice)
  static bool attr_set[64] = {false};
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_svgd_small), hipFuncAttributeMaxDynamicSharedMemorySize,
                                150 * 1024));
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  hipLaunchKernelGGL(k_svgd_small, dim3((unsigned)blocks), dim3(SM_THREADS), lds, stream, theta, score, (int)n, (int)d,
                     (float)log((double)n), phi, h2_out, sqpart, K_out, dK_out, sqnorm_out, small_chunk(n, d));
  LAUNCH_CHECK("k_svgd_small");
  *nparts = blocks == 1 ? 0 : blocks;   // one workgroup: it has written *sqnorm_out itself
  return STEIN_OK;
}
