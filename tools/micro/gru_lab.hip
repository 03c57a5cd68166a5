// Ablation lab for the bi-GRU forward recurrence: a copy of bigru.hip's forward kernel with pieces of the step switched off
// one at a time (results are WRONG under every flag; only the time per step is of interest).  B=32, T=360.
// build (repo root): hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -o build/gru_lab tools/micro/gru_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <stdint.h>

constexpr int H = 128, NTG = 512, CH = 8, HP = H + 16;
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ float hsum(f2 v) { return v.x + v.y; }
template <int CTRL> __device__ __forceinline__ float dpp_move(float old, float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float quad_sum(float v) { v += dpp_move<0xb1>(0.f, v); v += dpp_move<0x4e>(0.f, v); return v; }
__device__ __forceinline__ void dma64(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 4, 0, 0);
}
__device__ __forceinline__ void wait_vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ int pad32(int c) { return c + 4 * (c >> 5); }

enum { UNPACKED = 1024, ACC4 = 512, CHEAPDMA = 256, NOSTORE = 1, NOTRANS = 2, FAKEREAD = 4, NOBAR1 = 8, NOBAR2 = 16, NODMA = 32, NOFMA = 64, NOXROW = 128 };

template <int F> __device__ __forceinline__ float sig(float x) { return (F & NOTRANS) ? x * 0.25f + 0.5f : __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
template <int F> __device__ __forceinline__ float tnh(float x) { return (F & NOTRANS) ? x * 0.5f : 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

template <int F, int NF> __device__ __forceinline__ float dot32(const float4* v, const f2* w) {
  if (F & ACC4) {
    f2 a0 = {0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
#pragma unroll
    for (int k4 = 0; k4 < NF; k4 += 2) {
      a0 = pk_fma(f2{v[k4].x, v[k4].y}, w[2 * k4], a0);
      a1 = pk_fma(f2{v[k4].z, v[k4].w}, w[2 * k4 + 1], a1);
      if (k4 + 1 < NF) {
        a2 = pk_fma(f2{v[k4 + 1].x, v[k4 + 1].y}, w[2 * k4 + 2], a2);
        a3 = pk_fma(f2{v[k4 + 1].z, v[k4 + 1].w}, w[2 * k4 + 3], a3);
      }
    }
    return hsum((a0 + a1) + (a2 + a3));
  } else if (F & UNPACKED) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < NF; ++k4) {
      a0 = __builtin_fmaf(v[k4].x, w[2 * k4].x, a0);
      a1 = __builtin_fmaf(v[k4].y, w[2 * k4].y, a1);
      a2 = __builtin_fmaf(v[k4].z, w[2 * k4 + 1].x, a2);
      a3 = __builtin_fmaf(v[k4].w, w[2 * k4 + 1].y, a3);
    }
    return (a0 + a1) + (a2 + a3);
  } else {
    f2 a0 = {0.f, 0.f}, a1 = a0;
#pragma unroll
    for (int k4 = 0; k4 < NF; ++k4) {
      a0 = pk_fma(f2{v[k4].x, v[k4].y}, w[2 * k4], a0);
      a1 = pk_fma(f2{v[k4].z, v[k4].w}, w[2 * k4 + 1], a1);
    }
    return hsum(a0 + a1);
  }
}

struct W { const float* wg[2]; const float* wc[2]; };

template <int F>
__global__ __launch_bounds__(NTG, 1) void fwd(const float* __restrict__ xg, W w, float* __restrict__ out, float* __restrict__ ruc, int B, int T) {
  const int b = blockIdx.x, d = blockIdx.y, t_ = threadIdx.x;
  const int cp = t_ >> 2, kq = t_ & 3, lane = t_ & 63, wv = t_ >> 6;
  __shared__ __attribute__((aligned(16))) float hs[HP];
  __shared__ __attribute__((aligned(16))) float rhs[HP];
  __shared__ __attribute__((aligned(16))) float xgs[2][CH][3 * H];
  f2 wr[H / 8], wu[H / 8], wcand[H / 8];
  {
    const float* wg = w.wg[d] + (int64_t)(H + kq * (H / 4)) * (2 * H) + cp;
    const float* wc = w.wc[d] + (int64_t)(H + kq * (H / 4)) * H + cp;
#pragma unroll
    for (int k = 0; k < H / 8; ++k) {
      wr[k] = f2{wg[(int64_t)(2 * k) * (2 * H)], wg[(int64_t)(2 * k + 1) * (2 * H)]};
      wu[k] = f2{wg[(int64_t)(2 * k) * (2 * H) + H], wg[(int64_t)(2 * k + 1) * (2 * H) + H]};
      wcand[k] = f2{wc[(int64_t)(2 * k) * H], wc[(int64_t)(2 * k + 1) * H]};
    }
  }
  if (t_ < H) hs[pad32(t_)] = 0.f;
  const int64_t row0 = (int64_t)b * T;
  const int tstart = d == 0 ? 0 : T - 1, tstep = d == 0 ? 1 : -1;
  auto dma_chunk = [&](int c) {
    const int s = c * CH + wv;
    if (s < T) {
      const float* src = xg + (row0 + tstart + s * tstep) * (6 * H) + d * 3 * H + lane;
      float* dst = &xgs[c & 1][wv][0];
#pragma unroll
      for (int q = 0; q < 6; ++q) dma64(src + 64 * q, dst + 64 * q);
    }
  };
  dma_chunk(0);
  wait_vm0();
  lds_barrier();
  // CHEAPDMA: wave-uniform step index in an SGPR, the source pointer carried from chunk to chunk
  const int wvs = __builtin_amdgcn_readfirstlane(wv);
  const float* srcp = xg + (row0 + tstart + (int64_t)(CH + wvs) * tstep) * (6 * H) + d * 3 * H + lane;
  const int64_t src_adv = (int64_t)CH * tstep * (6 * H);
  constexpr int RM = (F & FAKEREAD) ? 1 : 7;
  for (int s = 0, t = tstart; s < T; ++s, t += tstep) {
    const int c = s / CH, i = s - c * CH;
    if (F & CHEAPDMA) {
      if (i == 0) {
        if ((c + 1) * CH + wvs < T) {
          float* dst = &xgs[(c + 1) & 1][wvs][0];
#pragma unroll
          for (int q = 0; q < 6; ++q) dma64(srcp + 64 * q, dst + 64 * q);
        }
        srcp += src_adv;
      }
    } else if (!(F & NODMA) && i == 0 && (c + 1) * CH < T) dma_chunk(c + 1);
    const float* xrow = (F & NOXROW) ? &xgs[0][0][0] : &xgs[c & 1][i][0];
    float4 hv[H / 16];
#pragma unroll
    for (int k4 = 0; k4 < H / 16; ++k4) hv[k4] = reinterpret_cast<const float4*>(hs + kq * (H / 4 + 4))[k4 & RM];
    constexpr int NF = (F & NOFMA) ? 1 : H / 16;
    const float rsum = dot32<F, NF>(hv, wr);
    const float rg = sig<F>(quad_sum(rsum) + xrow[cp]);
    const float hprev = hs[pad32(cp)];
    if (kq == 0) rhs[pad32(cp)] = rg * hprev;
    if (!(F & NOBAR1)) lds_barrier();
    float4 rv[H / 16];
#pragma unroll
    for (int k4 = 0; k4 < H / 16; ++k4) rv[k4] = reinterpret_cast<const float4*>(rhs + kq * (H / 4 + 4))[k4 & RM];
    const float usum = dot32<F, NF>(hv, wu);
    const float ug = sig<F>(quad_sum(usum) + xrow[H + cp]);
    const float csum = dot32<F, NF>(rv, wcand);
    const float cpre = quad_sum(csum) + xrow[2 * H + cp];
    if (!(F & NODMA) && i == CH - 1) wait_vm0();
    if (kq == 0) {
      const float cc = tnh<F>(cpre);
      const float hn = ug * hprev + (1.f - ug) * cc;
      hs[pad32(cp)] = hn;
      if (!(F & NOSTORE) || s == T - 1) {
        out[(row0 + t) * (2 * H) + d * H + cp] = hn;
        float* rp = ruc + (row0 + t) * (6 * H) + d * 3 * H;
        rp[cp] = rg; rp[H + cp] = ug; rp[2 * H + cp] = cc;
      }
    }
    if (!(F & NOBAR2)) lds_barrier();
  }
}

static float* dev_rand(size_t n, float scale) {
  std::vector<float> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = scale * ((float)rand() / RAND_MAX - 0.5f);
  float* d; hipMalloc(&d, n * sizeof(float)); hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
  return d;
}

template <int F> void run(const char* name, const float* xg, W w, float* out, float* ruc, int B, int T) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int it = 0; it < 4; ++it) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(fwd<F>, dim3(B, 2), dim3(NTG), 0, 0, xg, w, out, ruc, B, T);
    hipEventRecord(e1, 0); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (it && ms < best) best = ms;
  }
  printf("%-44s %.3f us/step (%.0f cycles at 2.4 GHz)\n", name, best * 1e3 / T, best * 1e3 / T * 2400);
}

int main() {
  const int B = 32, T = 360;
  W w;
  for (int d = 0; d < 2; ++d) { w.wg[d] = dev_rand(256 * 256, 0.2f); w.wc[d] = dev_rand(256 * 128, 0.2f); }
  float* xg = dev_rand((size_t)B * T * 768, 1.f);
  float* out = dev_rand((size_t)B * T * 256, 0.f);
  float* ruc = dev_rand((size_t)B * T * 768, 0.f);
  run<0>("full step", xg, w, out, ruc, B, T);
  run<NOSTORE>("no global stores", xg, w, out, ruc, B, T);
  run<NOTRANS>("no exp/rcp", xg, w, out, ruc, B, T);
  run<FAKEREAD>("2 of 8 LDS vector reads", xg, w, out, ruc, B, T);
  run<NODMA>("no input DMA", xg, w, out, ruc, B, T);
  run<CHEAPDMA>("scalar-addressed DMA issue", xg, w, out, ruc, B, T);
  run<ACC4>("4 accumulator chains per product", xg, w, out, ruc, B, T);
  run<UNPACKED>("unpacked v_fma_f32", xg, w, out, ruc, B, T);
  run<ACC4 | CHEAPDMA>("4 chains + scalar DMA", xg, w, out, ruc, B, T);
  run<NOXROW>("fixed xrow address", xg, w, out, ruc, B, T);
  run<NOBAR1>("no barrier 1", xg, w, out, ruc, B, T);
  run<NOBAR2>("no barrier 2", xg, w, out, ruc, B, T);
  run<NOBAR1 | NOBAR2>("no barriers", xg, w, out, ruc, B, T);
  run<NOFMA>("1/8 of the products", xg, w, out, ruc, B, T);
  run<NOFMA | FAKEREAD>("1/8 products, 2 of 8 reads", xg, w, out, ruc, B, T);
  run<NOFMA | FAKEREAD | NOSTORE | NOTRANS | NODMA>("1/8 products, 2/8 reads, no stores/trans/dma", xg, w, out, ruc, B, T);
  run<NOFMA | FAKEREAD | NOSTORE | NOTRANS | NODMA | NOBAR1 | NOBAR2>("... and no barriers", xg, w, out, ruc, B, T);
  return 0;
}
