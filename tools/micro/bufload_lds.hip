// Semantics check of buffer_load_dwordx4 ... offen lds on gfx950: out-of-range lanes (voffset >= num_records) must deposit ZEROS in
// LDS; the SGPR offset must not take part in the range check; M0 carries the wave's LDS base.
// build: hipcc --offload-arch=gfx950 -O3 -o build/bufload_lds tools/micro/bufload_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const float* p, float* out, int soff_floats, unsigned records) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < 512; i += 64) smem[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, records, 0x00020000);
  int voff = threadIdx.x * 16;
  if (threadIdx.x % 3 == 1) voff = 0x80000000;
  const int soff = __builtin_amdgcn_readfirstlane(soff_floats) * 4;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 256), 16, voff, soff, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = smem[i];
}
int main() {
  const int n = 1 << 20;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 512 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (int pass = 0; pass < 2; ++pass) {
    const int soff = pass ? 100000 : 8;
    const unsigned rec = pass ? 4096u : 0x7fffffffu;   // pass 1: soffset far beyond num_records, voffsets inside
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 2048, 0, d, o, soff, rec);
    std::vector<float> r(512);
    hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
    int bad = 0, untouched = 0;
    for (int i = 0; i < 256; ++i) untouched += r[i] == -1.f;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const float want = (l % 3 == 1) ? 0.f : (float)(soff + 4 * l + j);
        if (r[256 + 4 * l + j] != want) { if (bad < 5) printf("  lane %d.%d got %g want %g\n", l, j, r[256 + 4 * l + j], want); ++bad; }
      }
    printf("pass %d (soffset %d floats, num_records %u): %d mismatches, %d/256 floats below the destination untouched\n", pass, soff, rec, bad, untouched);
  }
  return 0;
}
