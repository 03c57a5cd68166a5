// Where a bi-GRU forward step spends its cycles: includes the library's bigru.hip with the issue-time stamps compiled in
// and launches the forward recurrence on random data (B=32, T=360).  Prints wave 0's average cycles per step per segment.
// build (repo root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -Itacotron_amd/csrc -Iinclude -o gpurun_out/gru_trace tools/micro/gru_trace.hip
#ifndef NO_STAMPS
#define TACO_GRU_TRACE 1
#endif
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../tacotron_amd/csrc/bigru.hip"

void taco_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int taco_prof_begin(int, hipStream_t) { return -1; }
void taco_prof_end(int, int, hipStream_t, double) {}
void taco_prof_label(int, int, const char*, ...) {}

static float* dev_rand(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX - 0.5f);
  float* d;
  hipMalloc(&d, n * sizeof(float));
  hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return d;
}

int main() {
  const int B = 32, T = 360;
  BiGruWeights w;
  for (int d = 0; d < 2; ++d) {
    w.wg[d] = dev_rand(256 * 256, 0.2f); w.bg[d] = dev_rand(256, 0.1f);
    w.wc[d] = dev_rand(256 * 128, 0.2f); w.bc[d] = dev_rand(128, 0.1f);
  }
  float* xg = dev_rand((size_t)B * T * 768, 1.f);
  float* out = dev_rand((size_t)B * T * 256, 0.f);
  float* ruc = dev_rand((size_t)B * T * 768, 0.f);
  long long* tr;
  hipMalloc(&tr, 8 * sizeof(long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 3; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(bigru_fwd_kernel, dim3(B, 2), dim3(NTG), 0, 0, xg, w, (const float*)nullptr, out, ruc, B, T, tr);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h[8];
    hipMemcpy(h, tr, sizeof(h), hipMemcpyDeviceToHost);
    double tot = 0; for (int i = 0; i < 7; ++i) tot += (double)h[i] / T;
    printf("launch %.1f us = %.3f us/step; wave 0 cycles per step:", ms * 1e3, ms * 1e3 / T);
    for (int i = 0; i < 7; ++i) printf(" [%d] %.0f", i, (double)h[i] / T);
    printf("  sum %.0f\n", tot);
  }
  return 0;
}
