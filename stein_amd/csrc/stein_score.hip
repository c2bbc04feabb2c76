// stein_score.hip -- score producers on the device: d log p / d theta for every particle, for the generalised
// linear models of the reference's examples.  This is the step immediately BEFORE the hot path: the reference
// obtains the score matrix with n sequential sess.run(grad_log_p) calls (stein/samplers/stein_sampler.py:59-68)
// on a TensorFlow graph of the model; here one launch writes the [n][d] score matrix that stein_svgd_phi consumes.
//
// Models (theta packs the weights w [F] and, for the hierarchical prior, log alpha):
//   linear    examples/linear_regression/main.py:18-31
//               log p = -1/2 sum_b (x_b.w - y_b)^2 + sum_c log N(w_c; 0, 1)
//               d/dw_c = scale * sum_b (y_b - x_b.w) x_bc - prec * w_c              (scale = 1, prec = 1 there)
//   logistic  examples/logistic_regression/main.py:23-49
//               log p = scale * sum_b [y_b z_b - softplus(z_b)],  z = X w,  scale = n_train / n_batch
//                       + sum_c log N(w_c; 0, alpha^-1/2) + log Gamma(alpha; 1, rate)   (density at alpha, no Jacobian)
//               d/dw_c       = scale * sum_b (y_b - sigmoid(z_b)) x_bc - alpha w_c
//               d/dlog alpha = F/2 - alpha (1/2 sum_c w_c^2 + rate)
//
// LP lanes per particle (64 / LP particles per wave), lanes over the features (w and the gradient stay in registers),
// the batch X staged in LDS and read by every particle group of a wave at the same address (broadcast).
#include "stein_common.h"

constexpr int SC_MAXLDS = 15 * 1024;   // floats of X per LDS chunk (60 KB) + its y values

template <int LP>
__device__ __forceinline__ float group_sum(float v) {   // sum over the LP consecutive lanes of a particle
#pragma unroll
  for (int o = LP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// KF = features per lane, LP = lanes per particle (F <= LP * KF)
template <int KF, int LP>
__global__ __launch_bounds__(256) void k_score_glm(const float* __restrict__ theta, int n, int d, int kind, int w_col,
                                                   int F, int alpha_col, const float* __restrict__ X,
                                                   const float* __restrict__ y, int B, int chunk_rows, float scale,
                                                   float prior_precision, float gamma_rate, float* __restrict__ score) {
  extern __shared__ float lds[];   // [chunk_rows][F] X, then [chunk_rows] y
  float* xs = lds;
  float* ys = lds + (size_t)chunk_rows * F;
  constexpr int PPB = 4 * (64 / LP);   // particles per workgroup pass
  const int t = threadIdx.x, lane = t % LP, sub = t / LP;
  const int nchunks = (B + chunk_rows - 1) / chunk_rows;
  auto stage = [&](int c0) {
    const int rows = min(chunk_rows, B - c0);
    for (int i = t; i < rows * F; i += 256) xs[i] = X[(size_t)c0 * F + i];
    for (int i = t; i < rows; i += 256) ys[i] = y[c0 + i];
  };
  if (nchunks == 1) {   // the whole batch fits: stage it once for every particle this workgroup handles
    stage(0);
    __syncthreads();
  }
  for (long p0 = (long)blockIdx.x * PPB; p0 < n; p0 += (long)gridDim.x * PPB) {
    const long p = p0 + sub;
    const bool live = p < n;
    const float* th = theta + (size_t)(live ? p : 0) * d;
    float w[KF], g[KF];
#pragma unroll
    for (int k = 0; k < KF; ++k) {
      const int c = lane + LP * k;
      w[k] = (live && c < F) ? th[w_col + c] : 0.f;
      g[k] = 0.f;
    }
    for (int ch = 0; ch < nchunks; ++ch) {
      if (nchunks > 1) {
        __syncthreads();
        stage(ch * chunk_rows);
        __syncthreads();
      }
      const int rows = min(chunk_rows, B - ch * chunk_rows);
      for (int b = 0; b < rows; ++b) {
        const float* xb = xs + (size_t)b * F;
        float xv[KF], part = 0.f;
#pragma unroll
        for (int k = 0; k < KF; ++k) {
          const int c = lane + LP * k;
          xv[k] = c < F ? xb[c] : 0.f;
          part = fmaf(w[k], xv[k], part);
        }
        const float z = group_sum<LP>(part);
        const float r = kind == STEIN_GLM_LOGISTIC ? ys[b] - 1.f / (1.f + expf(-z)) : ys[b] - z;
#pragma unroll
        for (int k = 0; k < KF; ++k) g[k] = fmaf(r, xv[k], g[k]);
      }
    }
    float prec = prior_precision, sw2 = 0.f;
    if (alpha_col >= 0) {
      prec = expf(live ? th[alpha_col] : 0.f);
#pragma unroll
      for (int k = 0; k < KF; ++k) sw2 = fmaf(w[k], w[k], sw2);
      sw2 = group_sum<LP>(sw2);
    }
    if (live) {
      float* out = score + (size_t)p * d;
      for (int c = lane; c < d; c += LP) {   // columns that are neither weights nor log alpha carry no gradient
        if (c < w_col || c >= w_col + F) out[c] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < KF; ++k) {
        const int c = lane + LP * k;
        if (c < F) out[w_col + c] = scale * g[k] - prec * w[k];
      }
      if (alpha_col >= 0 && lane == 0) out[alpha_col] = 0.5f * (float)F - prec * (0.5f * sw2 + gamma_rate);
    }
  }
}

// Few features (F <= SC_FEW): one wave per particle with the lanes over the BATCH instead -- every lane holds all F
// weights, walks b = lane, lane + 64, ... and keeps F partial gradient sums; one wave reduction per feature at the
// end.  (With the lanes over the features, the reference's linear example -- 1 feature, 1000 points -- ran a
// 1000-iteration serial loop with a shuffle reduction in every iteration on one active lane per particle.)
constexpr int SC_FEW = 8;

__global__ __launch_bounds__(256) void k_score_glm_few(const float* __restrict__ theta, int n, int d, int kind, int w_col,
                                                       int F, int alpha_col, const float* __restrict__ X,
                                                       const float* __restrict__ y, int B, float scale,
                                                       float prior_precision, float gamma_rate, float* __restrict__ score) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long p = (long)blockIdx.x * 4 + wave; p < n; p += (long)gridDim.x * 4) {
    const float* th = theta + (size_t)p * d;
    float w[SC_FEW], g[SC_FEW];
#pragma unroll
    for (int f = 0; f < SC_FEW; ++f) { w[f] = f < F ? th[w_col + f] : 0.f; g[f] = 0.f; }
    for (int b = lane; b < B; b += 64) {
      float xv[SC_FEW], z = 0.f;
#pragma unroll
      for (int f = 0; f < SC_FEW; ++f) { xv[f] = f < F ? X[(size_t)b * F + f] : 0.f; z = fmaf(w[f], xv[f], z); }
      const float r = kind == STEIN_GLM_LOGISTIC ? y[b] - 1.f / (1.f + expf(-z)) : y[b] - z;
#pragma unroll
      for (int f = 0; f < SC_FEW; ++f) g[f] = fmaf(r, xv[f], g[f]);
    }
    float prec = prior_precision, sw2 = 0.f;
#pragma unroll
    for (int f = 0; f < SC_FEW; ++f) { g[f] = group_sum<64>(g[f]); sw2 = fmaf(w[f], w[f], sw2); }
    if (alpha_col >= 0) prec = expf(th[alpha_col]);
    float* out = score + (size_t)p * d;
    for (int c = lane; c < d; c += 64) {
      float v = 0.f;   // columns that are neither weights nor log alpha carry no gradient
#pragma unroll
      for (int f = 0; f < SC_FEW; ++f)
        if (c == w_col + f && f < F) v = scale * g[f] - prec * w[f];
      if (c == alpha_col) v = 0.5f * (float)F - prec * (0.5f * sw2 + gamma_rate);
      out[c] = v;
    }
  }
}

extern "C" int stein_score_glm(const float* theta, int64_t n, int64_t d, int kind, int64_t w_col, int64_t n_feats,
                               int64_t alpha_col, const float* X, const float* y, int64_t batch, double scale,
                               double prior_precision, double gamma_rate, float* score, void* stream) {
  if (!theta || !X || !y || !score) return stein_fail(STEIN_E_BADARG, "NULL pointer");
  if (kind != STEIN_GLM_LINEAR && kind != STEIN_GLM_LOGISTIC) return stein_fail(STEIN_E_BADARG, "kind %d", kind);
  if (n < 1 || d < 1 || batch < 1 || n_feats < 1 || n > 0x7fffffffl || d > 0x7fffffffl || batch > 0x7fffffffl)
    return stein_fail(STEIN_E_SHAPE, "bad shape n=%lld d=%lld batch=%lld F=%lld", (long long)n, (long long)d,
                      (long long)batch, (long long)n_feats);
  if (w_col < 0 || w_col + n_feats > d || alpha_col >= d || (alpha_col >= w_col && alpha_col < w_col + n_feats))
    return stein_fail(STEIN_E_SHAPE, "weights [%lld, %lld) / log-alpha column %lld do not fit d=%lld", (long long)w_col,
                      (long long)(w_col + n_feats), (long long)alpha_col, (long long)d);
  if (n_feats > 1024) return stein_fail(STEIN_E_UNSUPPORTED, "more than 1024 features per particle");
  if (n_feats <= SC_FEW) {
    long blocks = (long)((n + 3) / 4);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_score_glm_few, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, theta, (int)n, (int)d,
                       kind, (int)w_col, (int)n_feats, (int)alpha_col, X, y, (int)batch, (float)scale,
                       (float)prior_precision, (float)gamma_rate, score);
    LAUNCH_CHECK("k_score_glm_few");
    return STEIN_OK;
  }
  int64_t chunk_rows = SC_MAXLDS / (n_feats + 1);
  if (chunk_rows > batch) chunk_rows = batch;
  const size_t lds_bytes = (size_t)chunk_rows * (n_feats + 1) * sizeof(float);
  hipStream_t s = (hipStream_t)stream;
  // 16 lanes per particle (4 particles per wave: a quarter of the reduction steps per particle) up to 256 features
  const int lp = n_feats <= 256 ? 16 : 64;
  const int ppb = 4 * (64 / lp);
  long blocks = (long)((n + ppb - 1) / ppb);
  if (blocks > 2048) blocks = 2048;
  const int kf = (int)((n_feats + lp - 1) / lp);
#define SC_LAUNCH(KF, LP)                                                                                            \
  hipLaunchKernelGGL((k_score_glm<KF, LP>), dim3((unsigned)blocks), dim3(256), lds_bytes, s, theta, (int)n, (int)d,   \
                     kind, (int)w_col, (int)n_feats, (int)alpha_col, X, y, (int)batch, (int)chunk_rows, (float)scale, \
                     (float)prior_precision, (float)gamma_rate, score)
  if (lp == 16) {
    if (kf <= 1) SC_LAUNCH(1, 16);
    else if (kf <= 2) SC_LAUNCH(2, 16);
    else if (kf <= 4) SC_LAUNCH(4, 16);
    else if (kf <= 8) SC_LAUNCH(8, 16);
    else SC_LAUNCH(16, 16);
  } else {
    if (kf <= 8) SC_LAUNCH(8, 64);
    else SC_LAUNCH(16, 64);
  }
#undef SC_LAUNCH
  LAUNCH_CHECK("k_score_glm");
  return STEIN_OK;
}

// ------------------------------------------------------------------------------------------------
// Bayesian neural-network regression, one hidden ReLU layer (examples/regression_neural_network/main.py:29-85):
//   pred_b = sum_h relu(sum_f x_bf w1_fh + b1_h) w2_h + b2
//   log p = [ s * sum_b log N(y_b; pred_b, 1/gamma) + log Gamma(lambda; a, b) + log Gamma(gamma; a, b)
//             + sum over w1, b1, w2, b2 of log N(.; 0, 1/lambda) ] / n_train,     s = n_train / n_batch,
//   lambda = exp(log_lambda), gamma = exp(log_gamma) (densities at lambda / gamma, no Jacobian)
// With e_b = y_b - pred_b, a_bh = relu(z_bh), m_bh = [z_bh > 0]:
//   d/dw2_h  = (s gamma sum_b e_b a_bh        - lambda w2_h ) / n_train
//   d/db2    = (s gamma sum_b e_b             - lambda b2   ) / n_train
//   d/db1_h  = (s gamma sum_b e_b w2_h m_bh   - lambda b1_h ) / n_train
//   d/dw1_fh = (s gamma sum_b e_b w2_h m_bh x_bf - lambda w1_fh) / n_train
//   d/dlog gamma  = (s (B/2 - gamma/2 sum_b e_b^2) + (a - 1) - b gamma) / n_train
//   d/dlog lambda = (P/2 - lambda/2 sum w^2 + (a - 1) - b lambda) / n_train,   P = F H + 2 H + 1 weights
// LP lanes per particle over the hidden units (KH per lane); the weights and their gradients stay in registers.
// ------------------------------------------------------------------------------------------------
constexpr int BNN_FMAX = 4;   // input features held in registers per hidden unit

struct BnnCols { int w1, b1, w2, b2, loglam, loggam; };

template <int KH, int LP>
__global__ __launch_bounds__(256) void k_score_bnn(const float* __restrict__ theta, int n, int d, int F, int H, BnnCols c,
                                                   const float* __restrict__ X, const float* __restrict__ y, int B,
                                                   int chunk_rows, float n_train, float ga, float gb,
                                                   float* __restrict__ score) {
  extern __shared__ float lds[];   // [chunk_rows][F] X, then [chunk_rows] y
  float* xs = lds;
  float* ys = lds + (size_t)chunk_rows * F;
  constexpr int PPB = 4 * (64 / LP);
  const int t = threadIdx.x, lane = t % LP, sub = t / LP;
  const int nchunks = (B + chunk_rows - 1) / chunk_rows;
  auto stage = [&](int c0) {
    const int rows = min(chunk_rows, B - c0);
    for (int i = t; i < rows * F; i += 256) xs[i] = X[(size_t)c0 * F + i];
    for (int i = t; i < rows; i += 256) ys[i] = y[c0 + i];
  };
  if (nchunks == 1) {
    stage(0);
    __syncthreads();
  }
  for (long p0 = (long)blockIdx.x * PPB; p0 < n; p0 += (long)gridDim.x * PPB) {
    const long p = p0 + sub;
    const bool live = p < n;
    const float* th = theta + (size_t)(live ? p : 0) * d;
    float w1[BNN_FMAX][KH], b1[KH], w2[KH], gw1[BNN_FMAX][KH], gb1[KH], gw2[KH];
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      const int h = lane + LP * k;
      const bool ok = live && h < H;
      b1[k] = ok ? th[c.b1 + h] : 0.f;
      w2[k] = ok ? th[c.w2 + h] : 0.f;
      gb1[k] = 0.f; gw2[k] = 0.f;
#pragma unroll
      for (int f = 0; f < BNN_FMAX; ++f) {
        w1[f][k] = (ok && f < F) ? th[c.w1 + f * H + h] : 0.f;
        gw1[f][k] = 0.f;
      }
    }
    const float b2 = live ? th[c.b2] : 0.f;
    float se = 0.f, se2 = 0.f;
    for (int ch = 0; ch < nchunks; ++ch) {
      if (nchunks > 1) {
        __syncthreads();
        stage(ch * chunk_rows);
        __syncthreads();
      }
      const int rows = min(chunk_rows, B - ch * chunk_rows);
      for (int b = 0; b < rows; ++b) {
        float xb[BNN_FMAX];
#pragma unroll
        for (int f = 0; f < BNN_FMAX; ++f) xb[f] = f < F ? xs[(size_t)b * F + f] : 0.f;
        float z[KH], part = 0.f;
#pragma unroll
        for (int k = 0; k < KH; ++k) {
          float v = b1[k];
#pragma unroll
          for (int f = 0; f < BNN_FMAX; ++f) v = fmaf(xb[f], w1[f][k], v);
          z[k] = v;
          part = fmaf(fmaxf(v, 0.f), w2[k], part);
        }
        const float e = ys[b] - (group_sum<LP>(part) + b2);
        se += e;
        se2 = fmaf(e, e, se2);
#pragma unroll
        for (int k = 0; k < KH; ++k) {
          gw2[k] = fmaf(e, fmaxf(z[k], 0.f), gw2[k]);
          const float tk = z[k] > 0.f ? e * w2[k] : 0.f;
          gb1[k] += tk;
#pragma unroll
          for (int f = 0; f < BNN_FMAX; ++f) gw1[f][k] = fmaf(tk, xb[f], gw1[f][k]);
        }
      }
    }
    float sw2 = 0.f;   // sum of squares of every weight under the N(0, 1/lambda) prior
#pragma unroll
    for (int k = 0; k < KH; ++k) {
      sw2 = fmaf(b1[k], b1[k], fmaf(w2[k], w2[k], sw2));
#pragma unroll
      for (int f = 0; f < BNN_FMAX; ++f) sw2 = fmaf(w1[f][k], w1[f][k], sw2);
    }
    sw2 = group_sum<LP>(sw2) + b2 * b2;
    if (live) {
      const float lam = expf(th[c.loglam]), gam = expf(th[c.loggam]);
      const float s = n_train / (float)B, cg = s * gam, inv = 1.f / n_train;
      float* out = score + (size_t)p * d;
      for (int j = lane; j < d; j += LP) {   // columns outside the model carry no gradient
        const bool in_model = (j >= c.w1 && j < c.w1 + F * H) || (j >= c.b1 && j < c.b1 + H) || (j >= c.w2 && j < c.w2 + H) ||
                              j == c.b2 || j == c.loglam || j == c.loggam;
        if (!in_model) out[j] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < KH; ++k) {
        const int h = lane + LP * k;
        if (h < H) {
          out[c.b1 + h] = (cg * gb1[k] - lam * b1[k]) * inv;
          out[c.w2 + h] = (cg * gw2[k] - lam * w2[k]) * inv;
#pragma unroll
          for (int f = 0; f < BNN_FMAX; ++f)
            if (f < F) out[c.w1 + f * H + h] = (cg * gw1[f][k] - lam * w1[f][k]) * inv;
        }
      }
      if (lane == 0) {
        const float P = (float)(F * H + 2 * H + 1);
        out[c.b2] = (cg * se - lam * b2) * inv;
        out[c.loggam] = (s * (0.5f * (float)B - 0.5f * gam * se2) + (ga - 1.f) - gb * gam) * inv;
        out[c.loglam] = (0.5f * P - 0.5f * lam * sw2 + (ga - 1.f) - gb * lam) * inv;
      }
    }
  }
}

extern "C" int stein_score_bnn(const float* theta, int64_t n, int64_t d, int64_t n_in, int64_t n_hidden,
                               const int64_t* cols /*[6]: w1, b1, w2, b2, log_lambda, log_gamma*/, const float* X,
                               const float* y, int64_t batch, double n_train, double gamma_a, double gamma_b,
                               float* score, void* stream) {
  if (!theta || !cols || !X || !y || !score) return stein_fail(STEIN_E_BADARG, "NULL pointer");
  if (n < 1 || d < 1 || batch < 1 || n_in < 1 || n_hidden < 1 || n > 0x7fffffffl || d > 0x7fffffffl || batch > 0x7fffffffl)
    return stein_fail(STEIN_E_SHAPE, "bad shape");
  if (n_in > BNN_FMAX) return stein_fail(STEIN_E_UNSUPPORTED, "more than %d input features", BNN_FMAX);
  if (n_hidden > 1024) return stein_fail(STEIN_E_UNSUPPORTED, "more than 1024 hidden units");
  const int64_t len[6] = {n_in * n_hidden, n_hidden, n_hidden, 1, 1, 1};
  for (int i = 0; i < 6; ++i) {
    if (cols[i] < 0 || cols[i] + len[i] > d) return stein_fail(STEIN_E_SHAPE, "block %d does not fit d=%lld", i, (long long)d);
    for (int j = 0; j < i; ++j)
      if (cols[i] < cols[j] + len[j] && cols[j] < cols[i] + len[i]) return stein_fail(STEIN_E_SHAPE, "blocks %d and %d overlap", j, i);
  }
  if (!(n_train > 0.0)) return stein_fail(STEIN_E_BADARG, "n_train must be positive");
  BnnCols c = {(int)cols[0], (int)cols[1], (int)cols[2], (int)cols[3], (int)cols[4], (int)cols[5]};
  int64_t chunk_rows = SC_MAXLDS / (n_in + 1);
  if (chunk_rows > batch) chunk_rows = batch;
  const size_t lds_bytes = (size_t)chunk_rows * (n_in + 1) * sizeof(float);
  const int lp = n_hidden <= 128 ? 16 : 64;
  const int ppb = 4 * (64 / lp);
  long blocks = (long)((n + ppb - 1) / ppb);
  if (blocks > 4096) blocks = 4096;
  const int kh = (int)((n_hidden + lp - 1) / lp);
  hipStream_t s = (hipStream_t)stream;
#define BNN_LAUNCH(KH, LP)                                                                                             \
  hipLaunchKernelGGL((k_score_bnn<KH, LP>), dim3((unsigned)blocks), dim3(256), lds_bytes, s, theta, (int)n, (int)d,     \
                     (int)n_in, (int)n_hidden, c, X, y, (int)batch, (int)chunk_rows, (float)n_train, (float)gamma_a,    \
                     (float)gamma_b, score)
  if (lp == 16) {
    if (kh <= 1) BNN_LAUNCH(1, 16);
    else if (kh <= 2) BNN_LAUNCH(2, 16);
    else if (kh <= 4) BNN_LAUNCH(4, 16);
    else BNN_LAUNCH(8, 16);
  } else {
    if (kh <= 4) BNN_LAUNCH(4, 64);
    else if (kh <= 8) BNN_LAUNCH(8, 64);
    else BNN_LAUNCH(16, 64);
  }
#undef BNN_LAUNCH
  LAUNCH_CHECK("k_score_bnn");
  return STEIN_OK;
}
