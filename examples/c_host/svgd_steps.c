/* svgd_steps.c -- a host WITHOUT Python or PyTorch driving libsteinhip.so through its C ABI (include/steinhip.h):
 * plain hipMalloc'ed buffers, a caller-owned workspace, a caller-created stream.
 *
 *   svgd_steps <in.bin> <out.bin> <n> <d> <steps>
 *     in.bin : float32 theta[n*d], float32 score[n*d]                       (row-major)
 *     out.bin: float32 phi[n*d] of the LAST step (unclipped), float32 theta[n*d] after `steps` steps,
 *              float32 h2[steps], float64 sqnorm[steps]
 *
 * Each step is what AbstractSteinSampler.update_particles does (stein/samplers/abstract_stein_sampler.py:107-127) with a
 * fixed score: phi = compute_phi(theta, score); phi *= 10 / max(10, |phi|); theta += AdagradGradientDescent(1e-3).update(phi)
 * (stein/optimizers/adagrad_gradient_descent.py:37-44).  Built and run by tests/test_gpu_c_host.py, which compares the
 * output with the oracle; the build alone (no GPU needed) is checked by tests/test_abi.py.
 *
 * Build: gcc -std=c11 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_host/svgd_steps.c -Lstein_amd -lsteinhip \
 *        -L/opt/rocm/lib -lamdhip64 -o svgd_steps      (plain C11: the HIP runtime API header and steinhip.h are C headers)
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "steinhip.h"

#define HIP_OK(x)                                                                       \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
  } while (0)
#define STEIN_CALL(x)                                                                   \
  do {                                                                                  \
    int rc_ = (x);                                                                      \
    if (rc_ != STEIN_OK) { fprintf(stderr, "%s: %d %s\n", #x, rc_, stein_last_error()); return 3; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 6) { fprintf(stderr, "usage: svgd_steps in.bin out.bin n d steps\n"); return 1; }
  const int64_t n = atoll(argv[3]), d = atoll(argv[4]);
  const int steps = atoi(argv[5]);
  const size_t count = (size_t)n * (size_t)d;
  float* host = (float*)malloc(2 * count * sizeof(float));
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(host, sizeof(float), 2 * count, f) != 2 * count) { fprintf(stderr, "cannot read %s\n", argv[1]); return 1; }
  fclose(f);

  const int flags = STEIN_FLAG_X3;   /* split-precision GEMMs on the 16-bit matrix cores (the default of the Python layer) */
  size_t ws_bytes = 0;
  STEIN_CALL(stein_workspace_bytes(n, n, d, STEIN_F32, flags, &ws_bytes));

  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  float *theta, *score, *phi, *hist, *h2;
  double* sqnorm;
  void* ws;
  HIP_OK(hipMalloc((void**)&theta, count * sizeof(float)));
  HIP_OK(hipMalloc((void**)&score, count * sizeof(float)));
  HIP_OK(hipMalloc((void**)&phi, count * sizeof(float)));
  HIP_OK(hipMalloc((void**)&hist, count * sizeof(float)));
  HIP_OK(hipMalloc((void**)&h2, sizeof(float)));
  HIP_OK(hipMalloc((void**)&sqnorm, sizeof(double)));
  HIP_OK(hipMalloc(&ws, ws_bytes));
  /* the workspace's select section carries the median predictor from step to step: start it clean once */
  HIP_OK(hipMemsetAsync(ws, 0, ws_bytes < ((size_t)1 << 20) ? ws_bytes : ((size_t)1 << 20), stream));
  {
    size_t off[STEIN_WS_NSECTIONS];
    int64_t extra[STEIN_WSX_N];
    STEIN_CALL(stein_workspace_layout(n, n, d, STEIN_F32, flags, off, extra));
    HIP_OK(hipMemsetAsync((char*)ws + off[STEIN_WS_SELECT], 0, 192, stream));
  }
  HIP_OK(hipMemcpyAsync(theta, host, count * sizeof(float), hipMemcpyHostToDevice, stream));
  HIP_OK(hipMemcpyAsync(score, host + count, count * sizeof(float), hipMemcpyHostToDevice, stream));

  float* h2_log = (float*)malloc(steps * sizeof(float));
  double* sq_log = (double*)malloc(steps * sizeof(double));
  for (int s = 0; s < steps; ++s) {
    STEIN_CALL(stein_svgd_phi(theta, score, n, d, 0, n, STEIN_F32, phi, h2, sqnorm, NULL, NULL, ws, ws_bytes, flags, stream));
    /* the scalars are only copied out for the report: the apply kernel reads |phi|^2 through the device pointer */
    HIP_OK(hipMemcpyAsync(&h2_log[s], h2, sizeof(float), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(&sq_log[s], sqnorm, sizeof(double), hipMemcpyDeviceToHost, stream));
    STEIN_CALL(stein_apply_adagrad(theta, phi, STEIN_F32, hist, (int64_t)count, STEIN_F32, sqnorm, 1.0, 10.0, 1e-3, 0.9, 1e-6,
                                   s == 0, NULL, stream));
  }
  float* out = (float*)malloc(2 * count * sizeof(float));
  HIP_OK(hipMemcpyAsync(out, phi, count * sizeof(float), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipMemcpyAsync(out + count, theta, count * sizeof(float), hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));

  f = fopen(argv[2], "wb");
  if (!f) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
  fwrite(out, sizeof(float), 2 * count, f);
  fwrite(h2_log, sizeof(float), steps, f);
  fwrite(sq_log, sizeof(double), steps, f);
  fclose(f);
  printf("n=%lld d=%lld steps=%d workspace=%zu bytes h2[last]=%g |phi|^2[last]=%g version=%d\n", (long long)n, (long long)d,
         steps, ws_bytes, (double)h2_log[steps - 1], sq_log[steps - 1], stein_version());
  return 0;
}
