// How fast can a wave split fp32 values into bf16 planes (h + m + l)?  The weight-gradient kernels spend 88 VALU operations per six
// MFMAs on it (bf16x3.h split2: truncation planes through v_perm / v_and / v_sub).  Candidates:
//   0  split2 as in bf16x3.h (11 operations per value pair)
//   1  round-to-nearest planes: v_cvt_pk_bf16_f32 + v_dot2_f32_bf16 against a (-1, 0) / (0, -1) selector (7 per pair)
//   2  round-to-nearest planes: v_cvt_pk_bf16_f32 + shift / and + v_sub (11 per pair)
// Reports cycles per value pair (one wave per SIMD and two), and for the round-to-nearest forms the worst |x - (h + m + l)| / |x|.
// build: hipcc --offload-arch=gfx950 -O3 -o build/split_probe tools/micro/split_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
__device__ __forceinline__ void split_trunc(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
  h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
  const float r0 = x0 - __uint_as_float(u0 & 0xffff0000u), r1 = x1 - __uint_as_float(u1 & 0xffff0000u);
  const unsigned v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
  m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
  const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
  l = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
}
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
// (builtins, not asm: a DOT result consumed by another VALU class needs wait states that the compiler only inserts for its own
//  instructions -- the asm form of this function returned garbage m planes; the selectors come through opaque registers because
//  the compiler folds a constant (-1, 0) pair into the inline constant -1.0, which the instruction applies to BOTH halves)
__device__ __forceinline__ void split_dot2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  unsigned klo = 0x0000bf80u, khi = 0xbf800000u;   // bf16(-1) in the low / high half
  asm volatile("" : "+s"(klo), "+s"(khi));
  const bf2 sel0 = __builtin_bit_cast(bf2, klo), sel1 = __builtin_bit_cast(bf2, khi);
  const bf2 hb = __builtin_convertvector(f2{x0, x1}, bf2);
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(hb, sel0, x0, false), r1 = __builtin_amdgcn_fdot2_f32_bf16(hb, sel1, x1, false);
  const bf2 mb = __builtin_convertvector(f2{r0, r1}, bf2);
  const float s0 = __builtin_amdgcn_fdot2_f32_bf16(mb, sel0, r0, false), s1 = __builtin_amdgcn_fdot2_f32_bf16(mb, sel1, r1, false);
  const bf2 lb = __builtin_convertvector(f2{s0, s1}, bf2);
  h = __builtin_bit_cast(unsigned, hb); m = __builtin_bit_cast(unsigned, mb); l = __builtin_bit_cast(unsigned, lb);
}
__device__ __forceinline__ void split_rne(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h) : "v"(x0), "v"(x1));
  const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xffff0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(m) : "v"(r0), "v"(r1));
  const float s0 = r0 - __uint_as_float(m << 16), s1 = r1 - __uint_as_float(m & 0xffff0000u);
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l) : "v"(s0), "v"(s1));
}
template <int V>
__device__ __forceinline__ void split(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
  if constexpr (V == 0) split_trunc(x0, x1, h, m, l);
  else if constexpr (V == 1) split_dot2(x0, x1, h, m, l);
  else split_rne(x0, x1, h, m, l);
}
template <int V>
__global__ void timing(const float* x, unsigned* o, long long* cyc, int iters) {
  float v[16];
  for (int i = 0; i < 16; ++i) v[i] = x[threadIdx.x * 16 + i];
  unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // eight INDEPENDENT pairs per iteration (the kernels split 8 + 8 values per MFMA group)
      unsigned h, m, l;
      split<V>(v[2 * i], v[2 * i + 1], h, m, l);
      acc[i] ^= h ^ m ^ l;           // 2 operations of glue per pair (v_xor3, v_xor) + 1 below
      v[2 * i] = __uint_as_float(__float_as_uint(v[2 * i]) ^ (acc[i] & 0x10u));
    }
  }
  const long long t1 = clock64();
  unsigned a = 0;
  for (int i = 0; i < 8; ++i) a ^= acc[i];
  o[blockIdx.x * blockDim.x + threadIdx.x] = a;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int V>
__global__ void planes(const float* x, unsigned* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i * 2 + 1 < n) {
    unsigned h, m, l;
    split<V>(x[2 * i], x[2 * i + 1], h, m, l);
    o[3 * i] = h; o[3 * i + 1] = m; o[3 * i + 2] = l;
  }
}
static double bf(unsigned half) { unsigned u = half << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  const int n = 1 << 20;
  std::vector<float> x(n);
  unsigned s = 12345;
  for (int i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    unsigned u = s;
    if ((i & 7) == 0) u = (u & 0x807fffffu) | ((100 + (s >> 24) % 60) << 23);   // wide exponent range
    else u = (u & 0x807fffffu) | ((120 + (i % 16)) << 23);
    if ((i & 1023) == 5) u |= 0x007fffffu;                                        // all-ones mantissas
    if ((i & 1023) == 6) u = (u & 0xff800000u) | 0x00008000u;                     // ties of the first rounding
    if ((i & 1023) == 7) u = (u & 0xff800000u) | 0x00018000u;
    if ((i & 1023) == 8) u = 0;
    memcpy(&x[i], &u, 4);
  }
  float* dx; unsigned* dout; long long* dc;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, (size_t)n / 2 * 3 * 4 + 4096 * 4); hipMalloc(&dc, 8);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  std::vector<unsigned> o((size_t)n / 2 * 3);
  for (int V = 0; V < 3; ++V) {
    if (V == 0) hipLaunchKernelGGL(planes<0>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dout, n);
    if (V == 1) hipLaunchKernelGGL(planes<1>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dout, n);
    if (V == 2) hipLaunchKernelGGL(planes<2>, dim3(n / 2 / 256), dim3(256), 0, 0, dx, dout, n);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst2 = 0; int inexact = 0;
    for (int i = 0; i < n; ++i) {
      const unsigned* p = &o[(size_t)(i / 2) * 3];
      const int sh = (i & 1) * 16;
      const double h = bf((p[0] >> sh) & 0xffff), m = bf((p[1] >> sh) & 0xffff), l = bf((p[2] >> sh) & 0xffff);
      const double e = fabs((double)x[i] - (h + m + l)), e2 = fabs((double)x[i] - (h + m));
      if (x[i] != 0.f) {
        worst = fmax(worst, e / fabs((double)x[i]));
        worst2 = fmax(worst2, e2 / fabs((double)x[i]));
      }
      if (e != 0 && ++inexact <= 4 && V == 1) printf("   x = %.9g (%08x): h %g m %g l %g\n", x[i], *(unsigned*)&x[i], h, m, l);
    }
    printf("variant %d: worst |x - (h+m+l)| / |x| = %.3g (%d of %d values inexact), worst |x - (h+m)| / |x| = %.3g\n", V, worst, inexact, n, worst2);
  }
  for (int waves = 4; waves <= 8; waves += 4)
    for (int V = 0; V < 3; ++V) {
      long long c = 0;
      for (int rep = 0; rep < 2; ++rep) {
        if (V == 0) hipLaunchKernelGGL(timing<0>, dim3(256), dim3(64 * waves), 0, 0, dx, dout, dc, 4096);
        if (V == 1) hipLaunchKernelGGL(timing<1>, dim3(256), dim3(64 * waves), 0, 0, dx, dout, dc, 4096);
        if (V == 2) hipLaunchKernelGGL(timing<2>, dim3(256), dim3(64 * waves), 0, 0, dx, dout, dc, 4096);
        hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
      }
      printf("variant %d, %d waves per SIMD: %.1f cycles per value pair per wave (incl. 4 operations of loop glue)\n", V, waves / 4, (double)c / (4096.0 * 8));
    }
  return 0;
}
