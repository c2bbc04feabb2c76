/* steinhip.h -- C ABI of libsteinhip.so, the MI355X (gfx950) SVGD particle-update engine.
 *
 * The reference (JamesBrofos/Stein, pure Python) has no FFI layer; its seams on this path are
 * Python duck-typed calls.  Every entry point below names the reference call it replaces
 * (file:line relative to the reference tree).  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (row-major, contiguous) unless the name ends in _host;
 *     the library never allocates or frees caller memory.  Scratch comes from a caller-owned
 *     workspace sized by stein_workspace_bytes() and described by stein_workspace_layout().
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*); none of them
 *     synchronises with the host.
 *   - return value: 0 on success, a negative STEIN_E_* code otherwise; stein_last_error()
 *     returns a per-thread message for the last failure.
 *   - n = total number of particles, d = parameters per particle.  A rank owns rows
 *     [row0, row0 + n_local) of theta / score / phi / optimizer state; single GPU: row0 = 0,
 *     n_local = n.
 */
#ifndef STEINHIP_H
#define STEINHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STEIN_VERSION 100 /* 0.1.0 */

enum {
  STEIN_OK = 0,
  STEIN_E_BADARG = -1,
  STEIN_E_SHAPE = -2,
  STEIN_E_WORKSPACE = -3,
  STEIN_E_HIP = -4,
  STEIN_E_RCCL = -5, /* an RCCL call of the library's own communicator failed (stein_comm_*, stein_rank_step) */
  STEIN_E_UNSUPPORTED = -6
};

/* element types of caller buffers */
enum { STEIN_F32 = 0, STEIN_BF16 = 1, STEIN_F64 = 2 };

/* flags for stein_svgd_phi / stein_workspace_bytes */
enum {
  STEIN_FLAG_NONE = 0,
  STEIN_FLAG_X3 = 1, /* run both GEMMs on the 16-bit matrix cores at fp32-level accuracy: every fp32 operand is scaled
                        by a power of two and split into two fp16 terms, three products per pair; see
                        stein_amd/csrc/stein_x3.hip.  bf16 inputs use one bf16 product.  Adds the PLANES section to
                        the workspace. */
  STEIN_FLAG_TIMING = 4, /* stein_svgd_phi only: record a HIP event at every stage boundary (see stein_timing_reserve) */
  STEIN_FLAG_TILED = 8,  /* stein_svgd_phi only: never take the one-kernel path for n <= 160 (stein_small.hip); the
                            tiled kernels then also leave D, the histograms and the planes in the workspace */
  STEIN_FLAG_NO_WINDOW = 16, /* stein_svgd_phi only: never grant the speculative median window, i.e. run the radix-select
                                passes over D on every call (same result; what a window miss costs, for measurements) */
  STEIN_FLAG_RANK_WINDOW = 32, /* stein_rank_* only: this step uses the cross-rank speculative window (tally / pick) instead
                                  of the radix-select histograms */
  STEIN_FLAG_TILE_DISTANCE = 64, /* stein_svgd_phi only: run the distance pass with the per-tile kernel even where the
                                    panel-resident form (STEIN_STAGE_TILES below) would be taken; for A/B measurements */
  STEIN_FLAG_TIMING_CONTRACT = 128 /* stein_svgd_phi, with STEIN_FLAG_TIMING: record only the two events that bracket the
                                      contraction (an event between two kernels costs the step ~3 us of GPU time: six of
                                      them perturb what they time); stein_timing_read reports -1 for the other stages */
};
/* flags for the staged distance / histogram calls */
enum {
  STEIN_STAGE_SYMMETRIC = 1, /* the block is the whole n x n matrix (row0 = 0, n_local = n): compute / count only the
                                upper triangle; the histograms weigh off-diagonal entries by 2.  On the fp32-MFMA path
                                (x3_planes = NULL) the distance pass stores each off-diagonal tile twice (mirrored): a
                                full image.  On the split path it stores ONLY the 128 x 128 tiles on and above the
                                diagonal; pass STEIN_STAGE_UPPER with that image to its readers.  Results are identical. */
  STEIN_STAGE_UPPER = 2,     /* readers of a distance image (stein_contract_partial, stein_kernel_contract,
                                stein_kernel_matrix): the image holds only the tiles on and above the diagonal of the
                                whole symmetric matrix, as stein_distance_block(STEIN_STAGE_SYMMETRIC) with x3_planes
                                leaves it; entries of the other tiles are read from their mirror images */
  STEIN_STAGE_TILES = 4,     /* stein_distance_block*: always the per-tile kernel (one 128 x 128 tile per workgroup).  Without
                                it, large blocks on the split path whose n, n_local and row0 are multiples of 128 take the
                                panel-resident kernels (stein_amd/csrc/stein_dpanel.hip: the operand panel of a row tile in
                                LDS, whole for d <= 256 with fp32 inputs / 512 with bf16, a chunk of K at a time beyond that):
                                the same formula, entries may differ from the per-tile kernel's in the last bit */
  STEIN_STAGE_PANEL = 8      /* stein_distance_block*: take the panel-resident kernel whenever its restrictions hold, however
                                small the block (tests) */
};

/* Workspace sections reported by stein_workspace_layout (byte offsets into the workspace). */
enum {
  STEIN_WS_ROWNORM = 0,  /* float  [n]                      r_i = |theta_i|^2                     */
  STEIN_WS_DIST = 1,     /* float  [n_local][ld_dist]       squared distances, row block          */
  STEIN_WS_HIST = 2,     /* int64  [3 levels][2][2048]      radix-select histograms               */
  STEIN_WS_SELECT = 3,   /* 192 B  select state (ranks, prefixes, median, h2) + the speculative-window state,
                            which must persist from one stein_svgd_phi call to the next, + 64 B of
                            launch tickets private to stein_svgd_phi                                  */
  STEIN_WS_PART_G = 4,   /* float  [split][n_local][d]      partial K.G                           */
  STEIN_WS_PART_T = 5,   /* float  [split][n_local][d]      partial K.theta                       */
  STEIN_WS_PART_RS = 6,  /* float  [split][n_local]         partial rowsum(K)                     */
  STEIN_WS_SQPART = 7,   /* double [sq_blocks]              per-block partial |phi|^2             */
  STEIN_WS_SPEC = 8,     /* 16 MB  entries caught by the speculative median window (stein_svgd_phi only)  */
  STEIN_WS_PLANES = 9,   /* split-precision operand planes + scales (STEIN_FLAG_X3 only; empty otherwise), always last */
  STEIN_WS_NSECTIONS = 10
};
/* extra[] entries reported by stein_workspace_layout */
enum { STEIN_WSX_LD_DIST = 0, STEIN_WSX_SPLIT = 1, STEIN_WSX_SQ_BLOCKS = 2, STEIN_WSX_HIST_BINS = 3, STEIN_WSX_N = 4 };

#define STEIN_HIST_BINS 2048
#define STEIN_HIST_LEVELS 3

int stein_version(void);
const char* stein_last_error(void);

/* Workspace sizing. */
int stein_workspace_bytes(int64_t n_local, int64_t n, int64_t d, int dtype, int flags, size_t* out_bytes);
int stein_workspace_layout(int64_t n_local, int64_t n, int64_t d, int dtype, int flags,
                           size_t* offsets /*[STEIN_WS_NSECTIONS]*/, int64_t* extra /*[STEIN_WSX_N]*/);

/* ---- fused single-rank path ------------------------------------------------------------------
 * Replaces AbstractSteinSampler.compute_phi (stein/samplers/abstract_stein_sampler.py:100-105)
 * together with everything it calls: SquaredExponentialKernel.kernel_and_grad
 * (stein/kernels/squared_exponential_kernel.py:25-35), the distance / bandwidth graph
 * (stein/kernels/abstract_kernel.py:30-40) and compute_median (stein/utilities/compute_median.py:4-16).
 *   theta_all, score_all : [n][d] of `dtype` (STEIN_F32 or STEIN_BF16)
 *   phi_local            : [n_local][d] float, rows row0..row0+n_local      (unclipped phi)
 *   h2_out               : float[1], receives bandwidth^2
 *   sqnorm_out           : double[1], receives sum(phi_local^2) (the rank-local part of |phi|_F^2)
 *   K_out / dK_out       : optional (may be NULL): [n_local][n] float / [n_local][d] float
 * row0 / n_local: this is the SINGLE-RANK entry -- row0 must be 0 and n_local must equal n (one rank sees every row, so
 * the median is global), anything else returns STEIN_E_BADARG.  The two arguments stay in the signature because SURVEY.md
 * section 8(b) fixes it (a binding written against that table keeps working); a rank of a sharded run calls
 * stein_rank_step, or the stein_rank_* segments with its own collectives between them (below).
 * The workspace carries state from one call to the next: the SELECT section keeps the median of the two previous
 * calls and, from the third call on, the distance pass counts the entries below a narrow window around the
 * extrapolated median and collects the entries inside it; when both median ranks fall inside the window an exact
 * selection among the collected entries replaces the two radix-select passes over D, otherwise those run.  The
 * result is identical either way (and for any workspace contents), only the time differs: keep ONE workspace per
 * particle set.
 */
int stein_svgd_phi(const void* theta_all, const void* score_all, int64_t n, int64_t d,
                   int64_t row0, int64_t n_local, int dtype,
                   float* phi_local, float* h2_out, double* sqnorm_out,
                   float* K_out, float* dK_out,
                   void* workspace, size_t ws_bytes, int flags, void* stream);

/* ---- staged path (tests, multi-rank: the host puts collectives between the stages) ------------ */

/* r_i = sum_k theta_ik^2            abstract_kernel.py:34 */
int stein_rownorms(const void* theta_all, int64_t n, int64_t d, int dtype, float* r_out, void* stream);

/* D[i][j] = r_i + r_j - 2 <theta_i, theta_j>, rows row0..row0+n_local, all n columns.
 * abstract_kernel.py:35.  dist_out is the tile-major distance image of leading dimension
 * ld_dist (>= n, multiple of 64; take it from stein_workspace_layout), rows padded to a multiple of 128.
 * hist_level0: optional (NULL to skip) pointer to the level-0 histogram int64[2][2048] (zeroed by
 * stein_median_begin): the kernel adds the level-0 counts of the entries it produces, which replaces
 * stein_median_hist_pass(level 0).  flags: 0 or STEIN_STAGE_SYMMETRIC. */
int stein_distance_block(const void* theta_all, const float* r_all, int64_t n, int64_t d,
                         int64_t row0, int64_t n_local, int dtype,
                         float* dist_out, int64_t ld_dist, void* hist_level0, const void* x3_planes, int flags,
                         void* stream);

/* Speculative median window, staged form for several ranks (stein_svgd_phi does the same inside one call; the state
 * and the idea are described there).  Every rank keeps its own SELECT and SPEC sections; they stay identical because
 * every rank sees the same medians.  Per step, on every rank:
 *   stein_spec_begin            instead of stein_median_begin: zeroes the histograms, sets the ranks and this step's window
 *   stein_distance_block_spec   stein_distance_block that also counts the weight below the window and collects the
 *                               entries inside it (and then does NOT fill hist_level0: state word `skip_l0` = 0)
 *   stein_spec_tally            entries -> table of SPEC table words (uint64): [0] weight below, [1] != 0: invalid,
 *                               [8 + k] weight of the k-th key of the window; the table is 65544 uint64 starting
 *                               2^21 uint64 into the SPEC section
 *   all-reduce(sum) of the table over the ranks (host)
 *   stein_spec_pick             both median targets from the summed table: sets h2 / median and the state word `hit`;
 *                               when `hit` stays 0 (read it back: uint32 at select_state + 64 + 28, `skip_l0` at + 52)
 *                               run the radix-select calls: stein_median_hist_pass(level 0) if skip_l0 == 0, then
 *                               all-reduce / stein_median_resolve / levels 1, 2 as usual
 *   stein_spec_update           after the median is final either way: predict the next step's window */
int stein_spec_begin(void* hist, void* select_state, void* spec_buf, int64_t total, void* stream);
int stein_distance_block_spec(const void* theta_all, const float* r_all, int64_t n, int64_t d,
                              int64_t row0, int64_t n_local, int dtype,
                              float* dist_out, int64_t ld_dist, void* hist_level0, const void* x3_planes, int flags,
                              void* select_state, void* spec_buf, void* stream);
int stein_spec_tally(void* select_state, void* spec_buf, void* stream);
int stein_spec_pick(void* select_state, void* spec_buf, int64_t n, float* h2_out, float* median_out, void* stream);
int stein_spec_update(void* select_state, void* stream);

/* Split-precision mode (STEIN_FLAG_X3), staged form: fills the PLANES section (x3_planes = workspace +
 * offsets[STEIN_WS_PLANES], planes_bytes = total - offsets[STEIN_WS_PLANES]) with the power-of-two scales and the
 * 16-bit terms (see STEIN_FLAG_X3) of every entry of theta (row-major and transposed) and of the score (transposed).
 * Passing the same pointer as `x3_planes` to stein_distance_block / stein_contract_partial / stein_kernel_contract
 * selects the 16-bit-MFMA kernels there; NULL selects the fp32-MFMA kernels.
 * dtype = STEIN_BF16 (BASELINE config 2): theta_all / score_all hold bf16 values; they are the single operand plane,
 * K is rounded to bf16 once (its rowsum uses the rounded values), every product is one bf16 MFMA with fp32
 * accumulation.  bf16 inputs always need STEIN_FLAG_X3 / the planes.
 * Either matrix may be NULL: then only the other one's scales and planes are rebuilt (theta before the distance pass,
 * the score any time before the contraction -- e.g. while its all-gather overlaps the distance pass). */
int stein_x3_prepare(const void* theta_all, const void* score_all, int64_t n, int64_t d, int dtype, void* x3_planes,
                     size_t planes_bytes, void* stream);

/* Exact median of all n*n distances by 3-level radix select on the fp32 bit pattern.
 * compute_median.py:4-16 (tf.nn.top_k of n^2//2+1 values; even count -> mean of the two middle).
 *   begin   : zero the histograms, set the two target ranks for `total` values in all
 *             (the path uses total = n*n; even -> ranks total/2-1 and total/2)
 *   hist    : add this rank's row block to level `level`'s histogram
 *   resolve : pick the digit holding each target rank (run after the histograms of all ranks
 *             have been summed); at the last level writes median and h2 = sqrt(med/ln n)^2
 *             (abstract_kernel.py:40, squared_exponential_kernel.py:22) into the select state
 *             and to h2_out.
 */
int stein_median_begin(void* hist, void* select_state, int64_t total, void* stream);
int stein_median_hist_pass(const float* dist, int64_t ld_dist, int64_t n_local, int64_t n, int level,
                           const void* select_state, void* hist, int flags, void* stream);
int stein_median_resolve(const void* hist, int level, int64_t n, void* select_state,
                         float* h2_out, float* median_out, void* stream);

/* K = exp(-D / h2 / 2)               squared_exponential_kernel.py:22 (optional output) */
int stein_kernel_matrix(const float* dist, int64_t ld_dist, int64_t n_local, int64_t n,
                        const float* h2_dev, float* K_out, int64_t ld_K, int dist_flags /* 0 or STEIN_STAGE_UPPER */,
                        void* stream);

/* phi rows of this rank:  (K.G + (rowsum(K) theta - K.theta)/h2) / n
 * squared_exponential_kernel.py:23,32 (dK) and abstract_stein_sampler.py:105 (phi).
 * Fused exp + fp32-MFMA contraction over the materialised distance block, then a finish pass. */
int stein_kernel_contract(const float* dist, int64_t ld_dist, const void* theta_all, const void* score_all,
                          int64_t n, int64_t d, int64_t row0, int64_t n_local, int dtype,
                          const float* h2_dev, float* phi_local, double* sqnorm_out, float* dK_out,
                          const void* x3_planes, void* workspace, size_t ws_bytes, int dist_flags /* 0 or STEIN_STAGE_UPPER */,
                          void* stream);

/* The two halves of stein_kernel_contract, exposed so a harness can time the MFMA kernel on its own:
 *   partial : k_phi_partial only (fills the PART_G / PART_T / PART_RS workspace sections)
 *   finish  : sums the split partials, forms phi (and dK), reduces |phi|^2 */
int stein_contract_partial(const float* dist, int64_t ld_dist, const void* theta_all, const void* score_all,
                           int64_t n, int64_t d, int64_t row0, int64_t n_local, int dtype,
                           const float* h2_dev, const void* x3_planes, void* workspace, size_t ws_bytes,
                           int dist_flags /* 0 or STEIN_STAGE_UPPER */, void* stream);
int stein_contract_finish(const void* theta_all, int64_t n, int64_t d, int64_t row0, int64_t n_local, int dtype,
                          const float* h2_dev, float* phi_local, double* sqnorm_out, float* dK_out,
                          void* workspace, size_t ws_bytes, int flags /* same STEIN_FLAG_* as the partial */,
                          void* stream);

/* ---- rank-step segments (row-sharded runs) ------------------------------------------------------------------------
 * What one rank does between two collectives, one call each: the staged calls above chained on `stream`, so that the
 * host layer issues per step only  all-gather(theta), all-gather(score) | stein_rank_begin | all-reduce | stein_rank_pick
 * (window form) or stein_rank_radix x3 with an all-reduce before each (radix form) | stein_rank_finish |
 * all-reduce(|phi|^2).  `workspace` is sized by stein_workspace_bytes(n_local, n, d, dtype, flags | STEIN_FLAG_TILED);
 * flags: STEIN_FLAG_X3 and, for the window form, STEIN_FLAG_RANK_WINDOW.  They replace, for rank p's rows, the same
 * reference lines as stein_svgd_phi.
 *   stein_rank_begin   row norms, theta's operand planes, median set-up, the [n_local, n] distance block (level-0
 *                      histogram or window counting in its epilogue) and, window form, the tally.  Then all-reduce(sum)
 *                      the window table (65544 uint64 at 2^21 uint64 into the SPEC section) or histogram level 0.
 *   stein_rank_pick    window form: both median targets from the summed table -> h2 / median; copies the 28 bytes of
 *                      window state from `hit` to `skip_l0` (uint32 [0] = hit, [6] = skip_l0) to flags_host -- page-locked
 *                      host memory -- on the stream: wait for it (an event) and, if hit == 0, run the radix form:
 *                      stein_rank_radix(0, need_pass = !skip_l0) then all-reduce level 0, stein_rank_radix(0, 0) ...
 *   stein_rank_radix   need_pass = 1: add the local block's histogram of `level`.  need_pass = 0 (hist[level] has been
 *                      summed over the ranks): resolve `level` and, below the last level, take the local histogram of
 *                      level + 1 (all-reduce it, call again).  After level 2: h2 / median are final.
 *   stein_rank_finish  window form: predictor update; then the contraction on the local rows -> phi_local, the local part
 *                      of |phi|^2 (all-reduce it), optional dK.  The score's operand planes must have been built
 *                      (stein_x3_prepare(NULL, score_all, ...), any time after the score rows have arrived). */
int stein_rank_begin(const void* theta_all, int64_t n, int64_t d, int64_t row0, int64_t n_local, int dtype,
                     void* workspace, size_t ws_bytes, int flags, void* stream);
int stein_rank_pick(int64_t n, int64_t d, int64_t row0, int64_t n_local, int dtype, void* workspace, size_t ws_bytes,
                    int flags, float* h2_out, float* median_out, void* flags_host, void* stream);
int stein_rank_radix(int level, int need_pass, int64_t n, int64_t d, int64_t row0, int64_t n_local, int dtype,
                     void* workspace, size_t ws_bytes, int flags, float* h2_out, float* median_out, void* stream);
int stein_rank_finish(const void* theta_all, const void* score_all, int64_t n, int64_t d, int64_t row0, int64_t n_local,
                      int dtype, const float* h2_dev, float* phi_local, double* sqnorm_out, float* dK_out,
                      void* workspace, size_t ws_bytes, int flags, void* stream);

/* ---- the sharded step with the library's own RCCL communicator ("one communicator per process-rank") ----
 * The reference has no counterpart (stein_sampler.py:11-14 "does not exploit parallelism"); per rank the step replaces
 * the same reference lines as stein_svgd_phi.  RCCL is looked up at run time (the copy already loaded in the process,
 * e.g. PyTorch-ROCm's, else librccl.so.1); without it these calls return STEIN_E_RCCL.
 *   stein_comm_unique_id  one rank makes the 128-byte id; the host layer hands it to every rank (any channel).
 *   stein_comm_init       collective over the nranks processes, on the CURRENT HIP device; blocks until all have joined.
 *   stein_rank_step       one whole step for rank p's rows [p n / nranks, (p + 1) n / nranks), queued on `stream`:
 *                         all-gather(theta rows, score rows) as one group -> stein_rank_begin -> all-reduce of the window
 *                         table -> stein_rank_pick (window form, STEIN_FLAG_RANK_WINDOW; the call waits for the 4-byte hit
 *                         flag and on a miss runs the radix form) or three histogram all-reduces with stein_rank_radix
 *                         (radix form: nothing waits) -> stein_rank_finish -> all-reduce(|phi|^2).
 *                         theta_all / score_all: [n, d] gather targets; h2_out, median_out, sqnorm_out: device scalars,
 *                         identical on every rank afterwards; window_hit_out (may be NULL): 1 / 0, -1 in the radix form.
 *                         flags: STEIN_FLAG_X3, STEIN_FLAG_RANK_WINDOW, STEIN_FLAG_TIMING; workspace sized by
 *                         stein_workspace_bytes(n / nranks, n, d, dtype, flags | STEIN_FLAG_TILED). */
#define STEIN_COMM_ID_BYTES 128
int stein_comm_unique_id(void* id_out, size_t id_bytes);
int stein_comm_init(const void* id, size_t id_bytes, int nranks, int rank, void** comm_out);
int stein_comm_info(void* comm, int* nranks_out, int* rank_out);
int stein_comm_destroy(void* comm);
int stein_rank_step(void* comm, const void* theta_local, const void* score_local, void* theta_all, void* score_all,
                    int64_t n, int64_t d, int dtype, float* phi_local, float* h2_out, float* median_out,
                    double* sqnorm_out, float* dK_out, void* workspace, size_t ws_bytes, int flags,
                    int* window_hit_out, void* stream);

/* Stage timing of the fused call (profiling aid; the events belong to the calling thread, like the last-error string:
 * reserve, call and read from one thread).  stein_timing_reserve(calls)
 * creates the HIP events for `calls` fused calls and rewinds the cursor; each stein_svgd_phi call made with
 * STEIN_FLAG_TIMING then records an event at every stage boundary on its stream while reserved slots last.
 * stein_timing_read waits for the recorded events and returns the stage durations in milliseconds,
 * ms_out[call * STEIN_T_NSTAGES + stage]; *calls_out = calls reported (<= max_calls). */
enum {
  STEIN_T_PREPARE = 0,  /* row norms, operand scales and planes, median set-up */
  STEIN_T_DISTANCE = 1, /* distance pass (+ level-0 histogram, speculative window) */
  STEIN_T_MEDIAN = 2,   /* window selection or radix-select passes */
  STEIN_T_CONTRACT = 3, /* K.[G|theta] partial contraction */
  STEIN_T_FINISH = 4,   /* phi, |phi|^2 */
  STEIN_T_NSTAGES = 5
};
int stein_timing_reserve(int calls);
int stein_timing_read(float* ms_out, int max_calls, int* calls_out);

/* ---- optimizer apply ---------------------------------------------------------------------------
 * Fuses the norm clip  phi *= 10 / max(10, |phi|_F)  (abstract_stein_sampler.py:125), the optimizer
 * map and  theta += step  (abstract_stein_sampler.py:126).
 *   sqnorm_dev  : device double[1] holding the GLOBAL |phi|_F^2, or NULL to use clip_scale_host
 *   theta       : [count] of state_dtype, updated in place; may be NULL (state + step_out only:
 *                 this is `gd.update(phi)` on its own)
 *   step_out    : optional [count] of state_dtype, receives the step
 *   state_dtype : STEIN_F32 or STEIN_F64 for theta / optimizer state / step_out; the map is evaluated in that type
 *                 (STEIN_F64 = the reference's NumPy float64 arithmetic)
 *   phi_dtype   : STEIN_F32 (what stein_svgd_phi produces) or, with STEIN_F64 state only, STEIN_F64: `gd.update(phi)`
 *                 on a float64 array then is the reference's pure-fp64 map, phi is not rounded to fp32 first
 * Adagrad: stein/optimizers/adagrad_gradient_descent.py:37-44 (first_step -> hist = phi^2).
 * Adam   : stein/optimizers/adam_gradient_descent.py:45-58 (t = n_iters AFTER increment;
 *          t == 1 -> mu = phi, nu = phi^2); the caller multiplies lr by decay afterwards.
 */
int stein_apply_adagrad(void* theta, const void* phi, int phi_dtype, void* hist, int64_t count, int state_dtype,
                        const double* sqnorm_dev, double clip_scale_host, double clip_threshold,
                        double lr, double alpha, double eps, int first_step,
                        void* step_out, void* stream);
int stein_apply_adam(void* theta, const void* phi, int phi_dtype, void* mu, void* nu, int64_t count, int state_dtype,
                     const double* sqnorm_dev, double clip_scale_host, double clip_threshold,
                     double lr, double beta1, double beta2, double eps, int64_t t,
                     void* step_out, void* stream);

/* ---- score producers (the step before the hot path) ------------------------------
 * d log p / d theta of every particle for the generalised linear models of the reference's examples, in one launch;
 * replaces the n sequential sess.run(grad_log_p) calls of SteinSampler.train_on_batch
 * (stein/samplers/stein_sampler.py:59-68) for these models.
 *   theta [n][d] float   particle i holds the weights w at columns w_col .. w_col + n_feats and, when alpha_col >= 0,
 *                        log(alpha) at column alpha_col; any other column gets score 0
 *   X [batch][n_feats], y [batch] float (device)
 *   STEIN_GLM_LINEAR    examples/linear_regression/main.py:18-31:  log p = -1/2 sum_b (x_b.w - y_b)^2 + log prior
 *   STEIN_GLM_LOGISTIC  examples/logistic_regression/main.py:23-49: log p = scale * sum_b [y_b z_b - softplus(z_b)] + log prior,
 *                       z = X w, scale = n_train / n_batch
 *   prior: alpha_col < 0: w ~ N(0, 1 / prior_precision);  alpha_col >= 0: w ~ N(0, 1 / alpha), alpha ~ Gamma(1, gamma_rate)
 *          evaluated at alpha = exp(theta[alpha_col]) without a Jacobian term, as the reference does
 *   score [n][d] float   d/dw_c = scale * sum_b resid_b x_bc - precision * w_c;  d/dlog alpha = F/2 - alpha (sum w^2 / 2 + rate) */
enum { STEIN_GLM_LINEAR = 0, STEIN_GLM_LOGISTIC = 1 };
int stein_score_glm(const float* theta, int64_t n, int64_t d, int kind, int64_t w_col, int64_t n_feats, int64_t alpha_col,
                    const float* X, const float* y, int64_t batch, double scale, double prior_precision,
                    double gamma_rate, float* score, void* stream);

/* Bayesian neural-network regression with one hidden ReLU layer, examples/regression_neural_network/main.py:29-85:
 *   pred = relu(X w1 + b1) w2 + b2;  log p = [ (n_train / batch) sum_b log N(y_b; pred_b, 1/gamma) + log Gamma(lambda; a, b)
 *   + log Gamma(gamma; a, b) + sum over all weights of log N(.; 0, 1/lambda) ] / n_train, lambda = exp(log_lambda),
 *   gamma = exp(log_gamma), densities evaluated without a Jacobian term, as the reference does.
 *   cols[6]: first column of w1 ([n_in][n_hidden] row-major), b1 [n_hidden], w2 [n_hidden], b2, log_lambda, log_gamma in
 *   the packed particle; any other column gets score 0.  n_in <= 4, n_hidden <= 1024. */
int stein_score_bnn(const float* theta, int64_t n, int64_t d, int64_t n_in, int64_t n_hidden, const int64_t* cols,
                    const float* X, const float* y, int64_t batch, double n_train, double gamma_a, double gamma_b,
                    float* score, void* stream);

/* small helpers used by the host layer */
int stein_cast_f64_to_f32(const double* src, float* dst, int64_t count, void* stream);
int stein_cast_f32_to_bf16(const float* src, void* dst, int64_t count, void* stream);

/* ---- device error word -----------------------------------------------------------------------------
 * The entry points are asynchronous, so a kernel that has to give up cannot return a code.  It writes NaN into what it
 * was computing (the step's bandwidth: the reference's own failure value, compute_median.py:4-16 of identical particles)
 * and raises a per-device word in page-locked host memory; stein_svgd_phi and stein_apply_* look at that word -- a plain
 * host read, no synchronisation -- before they queue anything and return STEIN_E_HIP once, naming the kernel.  Today one
 * kernel can raise it: the one-launch radix select of the fused call (512 < n <= 4096), whose level barriers are
 * bounded although they cannot deadlock by construction.
 * stein_take_device_error: the same check on demand (0, or STEIN_E_HIP once).
 * Test hooks (per calling thread, no effect on results): stein_debug_hist_all_grid(blocks) launches that kernel with
 * `blocks` workgroups instead of 512 (0 = default; any grid >= 1 gives the same median: tests/test_gpu_spec.py);
 * stein_debug_raise_device_error raises the current device's word as a kernel would. */
int stein_take_device_error(void);
int stein_debug_hist_all_grid(int blocks);
int stein_debug_hist_all_vblocks(int nvb);   /* tuning aid: virtual workgroups per level of that kernel (0 = default) */
int stein_debug_raise_device_error(void);

#ifdef __cplusplus
}
#endif
#endif /* STEINHIP_H */
