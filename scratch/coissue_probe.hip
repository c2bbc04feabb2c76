// coissue_probe.hip -- what does an instruction issued by ANOTHER wave of the same SIMD cost the matrix pipe?
// 512-thread workgroups, one per CU: waves 0-3 (one per SIMD) issue v_mfma_f32_16x16x32_f16 back to back into 16 independent
// accumulators and time themselves with s_memtime; waves 4-7 (the second wave of each SIMD) run a filler loop of one kind
// until the MFMA waves are done (a flag in LDS).  Prints shader cycles per MFMA (16 = the pipe saturated by one wave) and the
// filler instructions the second wave got through per MFMA.  The distance kernel's question: its epilogue (VALU, LDS staging,
// stores) runs in one wave of a SIMD while the other one is in its MFMA loop.
// build: hipcc --offload-arch=gfx950 -O3 scratch/coissue_probe.hip -o /tmp/coissue ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { F_NONE, F_IDLE_SLEEP, F_VALU_INDEP, F_VALU_CHAIN, F_PK_FMA, F_CMP_ADDC, F_SALU, F_LDS_RW, F_STORE, F_MIX, F_MFMA, F_FMA_MIX, F_EXP, F_CVT_PK, F_FMAAK, F_CVT_SDWA, F_DOT2, F_LOAD, F_LOAD_STORE, F_COUNT };
static const char* const NAMES[] = {"no second wave", "second wave sleeps", "v_fma_f32, 8 independent", "v_fma_f32, one chain",
                                    "v_pk_fma_f32, 8 independent", "v_cmp + v_addc chain", "s_add loop", "ds_write_b128 + ds_read_b128",
                                    "global_store_dwordx4 (1 KB)", "epilogue-like mix", "second wave issues MFMAs too", "v_fma_mix_f32, 8 independent",
                                    "v_exp_f32, 8 independent", "v_cvt_pk_f16_f32, 8 independent", "v_fmaak_f32, 8 independent",
                                    "v_cvt_f32_f16 (sdwa word 1) + v_sub", "v_dot2_f32_f16, 8 independent",
                                    "global_load_dwordx4 (1 KB, L2 hits)", "2 loads + 1 store of 1 KB"};

template <int FILL, int PRIO, int AIDLE>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float* sink, int iters) {
  __shared__ int done;
  __shared__ __attribute__((aligned(16))) float stage[8][16 * 36];
  __shared__ char pad[120 * 1024];   // one workgroup per CU
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) { done = 0; pad[0] = 0; }
  __syncthreads();
  if (w < 4) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i * 0.37f - 1.f); b[i] = (_Float16)(1.0f / (1 + i + lane) - 0.2f); }
    f32x4 acc[16];
    for (int c = 0; c < 16; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (PRIO) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (AIDLE) {   // the first wave only waits, for as long as its MFMAs would take when the pipe is saturated
      while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)iters * 16ull * 17ull) __builtin_amdgcn_s_sleep(20);
    } else
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < 16; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) { out[(blockIdx.x * 8 + w) * 2] = t1 - t0; atomicAdd(&done, 1); }
    if (s == 12345.678f) sink[threadIdx.x] = s;
  } else if (FILL != F_NONE) {
    unsigned long long n = 0;
    float x[8], y = 1.0001f, z = 0.5f;
    for (int i = 0; i < 8; ++i) x[i] = lane * 0.001f + i;
    unsigned cnt = 0;
    float* gp = sink + 4096 + (size_t)(blockIdx.x * 8 + w) * 4096 + lane * 4;
    volatile int* vd = &done;
    while (*vd < 4) {
      if (FILL == F_IDLE_SLEEP) { __builtin_amdgcn_s_sleep(100); n += 1; continue; }
#pragma unroll
      for (int rep = 0; rep < 8; ++rep) {
        if (FILL == F_VALU_INDEP) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(y), "v"(z));
        } else if (FILL == F_VALU_CHAIN) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(y), "v"(z));
        } else if (FILL == F_PK_FMA) {
          f32x2* p = reinterpret_cast<f32x2*>(x);
          const f32x2 yy = {y, y}, zz = {z, z};
#pragma unroll
          for (int i = 0; i < 4; ++i) { asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(yy), "v"(zz)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(yy), "v"(zz)); }
        } else if (FILL == F_CMP_ADDC) {
          asm volatile("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_cmp_gt_i32 vcc, %1, %3\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
                       "v_cmp_gt_i32 vcc, %1, %4\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_cmp_gt_i32 vcc, %1, %5\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc"
                       : "+v"(cnt) : "s"(0x40000000), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]) : "vcc");
        } else if (FILL == F_SALU) {
          int sreg = __builtin_amdgcn_readfirstlane((int)n);
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
          cnt += sreg;
        } else if (FILL == F_LDS_RW) {
          // (8 LDS instructions: 4 x (write 1 KB, read 1 KB), the distance epilogue's staging pattern)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<float4*>(&stage[w][(lane & 15) * 36 + (lane >> 4) * 4]) = make_float4(x[0], x[1], x[2], x[3]);
            const float4 r = *reinterpret_cast<const float4*>(&stage[w][(lane >> 3) * 36 + (lane & 7) * 4]);
            x[0] += r.x; asm volatile("" ::: "memory");
          }
        } else if (FILL == F_STORE) {
#pragma unroll
          for (int i = 0; i < 8; ++i) { *reinterpret_cast<float4*>(gp + i * 256) = make_float4(x[0], x[1], x[2], x[3]); asm volatile("" ::: "memory"); }
        } else if (FILL == F_MIX) {
          // one 16-row block of the distance epilogue, roughly: 8 packed value instructions, 16 compare/add, 12 window, 2 + 2 LDS, 2 stores
          f32x2* p = reinterpret_cast<f32x2*>(x);
          const f32x2 yy = {y, y}, zz = {z, z};
#pragma unroll
          for (int i = 0; i < 4; ++i) { asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(zz)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(yy), "v"(zz)); }
          asm volatile("v_cmp_gt_i32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_cmp_gt_i32 vcc, %1, %3\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
                       "v_cmp_gt_i32 vcc, %1, %4\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_cmp_gt_i32 vcc, %1, %5\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
                       "v_cmp_gt_i32 vcc, %1, %6\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_cmp_gt_i32 vcc, %1, %7\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\t"
                       "v_cmp_gt_i32 vcc, %1, %8\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc\n\tv_cmp_gt_i32 vcc, %1, %9\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc"
                       : "+v"(cnt) : "s"(0x40000000), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]) : "vcc");
          unsigned mn = 0xffffffffu;
#pragma unroll
          for (int i = 0; i < 8; ++i) mn = min(mn, __float_as_uint(x[i]) - 0x3f000000u);
          if (__ballot(mn <= 2u)) cnt += 1000;
          *reinterpret_cast<float4*>(&stage[w][(lane & 15) * 36 + (lane >> 4) * 4]) = make_float4(x[0], x[1], x[2], x[3]);
          *reinterpret_cast<float4*>(&stage[w][(lane & 15) * 36 + 16 + (lane >> 4) * 4]) = make_float4(x[4], x[5], x[6], x[7]);
          const float4 r0 = *reinterpret_cast<const float4*>(&stage[w][(lane >> 3) * 36 + (lane & 7) * 4]);
          const float4 r1 = *reinterpret_cast<const float4*>(&stage[w][(8 + (lane >> 3)) * 36 + (lane & 7) * 4]);
          *reinterpret_cast<float4*>(gp + (rep & 7) * 512) = r0;
          *reinterpret_cast<float4*>(gp + (rep & 7) * 512 + 256) = r1;
          asm volatile("" ::: "memory");
        } else if (FILL == F_FMA_MIX) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(y), "v"(z));
        } else if (FILL == F_EXP) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        } else if (FILL == F_CVT_PK) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
        } else if (FILL == F_FMAAK) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x41600000" : "+v"(x[i]) : "v"(y));
        } else if (FILL == F_CVT_SDWA) {
#pragma unroll
          for (int i = 0; i < 4; ++i) { float tmp; asm volatile("v_cvt_f32_f16_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(tmp) : "v"(y)); asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(tmp)); }
        } else if (FILL == F_DOT2) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(y), "v"(z));
        } else if (FILL == F_LOAD) {
          // (eight 1 KB loads from a 64 KB region per wave that stays in the L2; waited for once per loop body)
          f32x4 r[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[i]) : "v"(gp + ((rep * 8 + i) & 15) * 256));
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < 8; ++i) x[i] += r[i][0] * 1e-30f;
        } else if (FILL == F_LOAD_STORE) {
          f32x4 r[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r[i]) : "v"(gp + ((rep * 4 + i) & 7) * 256));
          *reinterpret_cast<float4*>(gp + 2048 + (rep & 3) * 256) = make_float4(x[0], x[1], x[2], x[3]);
          *reinterpret_cast<float4*>(gp + 3072 + (rep & 3) * 256) = make_float4(x[4], x[5], x[6], x[7]);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int i = 0; i < 4; ++i) x[i] += r[i][0] * 1e-30f;
        } else if (FILL == F_MFMA) {
          f16x8 a, b;
          for (int i = 0; i < 8; ++i) { a[i] = (_Float16)x[i]; b[i] = (_Float16)(x[i] * 0.5f); }
          f32x4 c4 = {x[0], x[1], x[2], x[3]};
#pragma unroll
          for (int i = 0; i < 8; ++i) c4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4, 0, 0, 0);
          x[0] = c4[0];
        }
      }
      n += 8;
    }
    float s = cnt;
    for (int i = 0; i < 8; ++i) s += x[i];
    if (lane == 0) out[(blockIdx.x * 8 + w) * 2 + 1] = n;
    if (s == 12345.678f) sink[threadIdx.x] = s;
  }
}

template <int FILL, int PRIO, int AIDLE = 0>
static void run(unsigned long long* out, float* sink) {
  const int iters = 20000, grid = 256;
  static unsigned long long host[256 * 16];
  hipMemset(out, 0, sizeof(host));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<FILL, PRIO, AIDLE>), dim3(grid), dim3(512), 0, 0, out, sink, 200);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<FILL, PRIO, AIDLE>), dim3(grid), dim3(512), 0, 0, out, sink, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(host, out, sizeof(host), hipMemcpyDeviceToHost);
  double cyc = 0, fill = 0;
  for (int g = 0; g < grid; ++g) for (int w = 0; w < 4; ++w) { cyc += (double)host[(g * 8 + w) * 2]; fill += (double)host[(g * 8 + 4 + w) * 2 + 1]; }
  const double mf = 16.0 * iters;
  printf("%-30s %s prio %d: %7.3f ms  %6.2f shader cycles per MFMA (16 = saturated)  -> %5.3f busy at %.2f GHz | filler loop bodies per MFMA %.4f\n",
         NAMES[FILL], AIDLE ? "(first wave idle)" : "                 ", PRIO, ms, cyc / (1024.0 * mf), 16.0 * 1024.0 * mf / cyc, cyc / 1024.0 / (ms * 1e6), fill / (1024.0 * mf));
}

int main() {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 256 * 16 * 8);
  hipMalloc(&sink, (4096 + 2048 * 4096 + 8192) * 4);
  run<F_NONE, 0>(out, sink); run<F_IDLE_SLEEP, 0>(out, sink);
  run<F_VALU_INDEP, 0>(out, sink); run<F_VALU_INDEP, 1>(out, sink);
  run<F_VALU_CHAIN, 0>(out, sink); run<F_VALU_CHAIN, 1>(out, sink);
  run<F_PK_FMA, 0>(out, sink); run<F_PK_FMA, 1>(out, sink);
  run<F_CMP_ADDC, 0>(out, sink); run<F_CMP_ADDC, 1>(out, sink);
  run<F_SALU, 0>(out, sink);
  run<F_LDS_RW, 0>(out, sink); run<F_LDS_RW, 1>(out, sink);
  run<F_STORE, 0>(out, sink); run<F_STORE, 1>(out, sink);
  run<F_MIX, 0>(out, sink); run<F_MIX, 1>(out, sink);
  run<F_MFMA, 0>(out, sink);
  run<F_FMA_MIX, 0>(out, sink); run<F_EXP, 0>(out, sink); run<F_CVT_PK, 0>(out, sink); run<F_FMAAK, 0>(out, sink); run<F_CVT_SDWA, 0>(out, sink); run<F_DOT2, 0>(out, sink);
  run<F_LOAD, 0>(out, sink); run<F_LOAD, 0, 1>(out, sink); run<F_LOAD_STORE, 0>(out, sink); run<F_LOAD_STORE, 0, 1>(out, sink);
  run<F_FMA_MIX, 0, 1>(out, sink); run<F_EXP, 0, 1>(out, sink); run<F_CVT_PK, 0, 1>(out, sink); run<F_FMAAK, 0, 1>(out, sink); run<F_CVT_SDWA, 0, 1>(out, sink); run<F_DOT2, 0, 1>(out, sink);
  run<F_VALU_INDEP, 0, 1>(out, sink); run<F_VALU_CHAIN, 0, 1>(out, sink); run<F_PK_FMA, 0, 1>(out, sink); run<F_CMP_ADDC, 0, 1>(out, sink);
  run<F_LDS_RW, 0, 1>(out, sink); run<F_STORE, 0, 1>(out, sink); run<F_MIX, 0, 1>(out, sink); run<F_MFMA, 0, 1>(out, sink);
  return 0;
}
