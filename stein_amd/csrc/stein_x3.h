// stein_x3.h -- host entry points of the split-precision ("x3") kernels in stein_x3.hip.
#pragma once
#include "stein_common.h"

// dtype = STEIN_F32: fp32 inputs, two fp16 planes, three products; STEIN_BF16: bf16 inputs, one plane, one product
int stein_x3_kind(int dtype);   // 1 or 2
int stein_x3_split(const void* theta_all, const void* score_all, int dtype, int64_t n, int64_t d, const SteinLayout& L,
                   char* planes, hipStream_t stream, u32* fuse_done = nullptr);
int stein_x3_distance(const char* planes, const SteinLayout& L, int dtype, const float* r_all, float* dist_out,
                      int64_t n, int64_t d, int64_t row0, int64_t n_local, int64_t ld_dist, u64* hist0, bool symmetric,
                      hipStream_t stream, SpecState* spec = nullptr, u64* spec_buf = nullptr);
int stein_x3_contract_partial(const float* dist, int64_t ld_dist, const char* planes, const SteinLayout& L, int dtype,
                              const float* h2_dev, float* OG, float* OT, float* RS, int64_t n, int64_t d,
                              int64_t n_local, hipStream_t stream, bool upper /* dist holds only the tiles on and above
                              the diagonal of a symmetric block (what stein_x3_distance(symmetric) stores) */);
