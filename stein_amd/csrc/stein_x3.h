// stein_x3.h -- host entry points of the split-precision ("x3") kernels in stein_x3.hip.
#pragma once
#include "stein_common.h"

// dtype = STEIN_F32: fp32 inputs, two fp16 planes, three products; STEIN_BF16: bf16 inputs, one plane, one product
int stein_x3_kind(int dtype);   // 1 or 2
int stein_x3_split(const void* theta_all, const void* score_all, int dtype, int64_t n, int64_t d, const SteinLayout& L,
                   char* planes, hipStream_t stream, HistSync* fuse_done = nullptr, bool scales_written = false,
                   const PrologueArgs* prologue = nullptr /* fused call, bf16: the launch also does the prologue's work */);
int stein_x3_distance(const char* planes, const SteinLayout& L, int dtype, const float* r_all, float* dist_out,
                      int64_t n, int64_t d, int64_t row0, int64_t n_local, int64_t ld_dist, u64* hist0, bool symmetric,
                      hipStream_t stream, SpecState* spec = nullptr, u64* spec_buf = nullptr,
                      int panel = 0 /* -1: never the panel-resident kernel, 1: whenever it can run, 0: where it pays */);
// stein_dpanel.hip: the panel-resident distance kernel and the test that picks it (stein_x3_distance applies it)
bool stein_dpanel_ok(const SteinLayout& L, int dtype, int64_t n, int64_t row0, int64_t n_local, bool level0_only,
                     bool any_size);
int stein_dpanel_distance(const char* planes, const SteinLayout& L, int dtype, const float* r_all, float* dist_out,
                          int64_t n, int64_t row0, int64_t n_local, int64_t ld_dist, bool symmetric, hipStream_t stream,
                          SpecState* spec, u64* spec_buf, u64* hist0);
int stein_x3_contract_partial(const float* dist, int64_t ld_dist, const char* planes, const SteinLayout& L, int dtype,
                              const float* h2_dev, float* OG, float* OT, float* RS, int64_t n, int64_t d,
                              int64_t n_local, hipStream_t stream, bool upper /* dist holds only the tiles on and above
                              the diagonal of a symmetric block (what stein_x3_distance(symmetric) stores) */);
