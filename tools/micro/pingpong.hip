// Ping-pong latency between two workgroups through 8-byte {epoch,value} granules, for the decoder exchange design:
//   mode 0: agent-scope atomics (sc1; what decoder.hip uses; works across XCDs)
//   mode 1: workgroup-scope atomics (sc0: bypass L1, served by the XCD's L2; only coherent if both blocks share an XCD)
// Blocks are launched in a grid of G; block `a` and block `b` play, the rest idle.  XCC id of each block is recorded.
// build: hipcc --offload-arch=gfx950 -O3 -o pingpong pingpong.hip ; run: ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

template <int SCOPE>
__device__ __forceinline__ void put(u64* p, unsigned e, unsigned v) {
  __hip_atomic_store((gu64*)p, ((u64)e << 32) | v, __ATOMIC_RELAXED, SCOPE);
}
template <int SCOPE>
__device__ __forceinline__ bool get(u64* p, unsigned e) {
  for (unsigned spin = 0; spin < (1u << 22); ++spin) {
    const u64 x = __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, SCOPE);
    if ((unsigned)(x >> 32) == e) return true;
  }
  return false;
}

template <int SCOPE, int SSCOPE = SCOPE>
__global__ void pingpong(u64* buf, int a, int b, int iters, long long* out, int* xcc) {
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = id & 0xf;
  }
  if ((int)blockIdx.x != a && (int)blockIdx.x != b) return;
  if (threadIdx.x != 0) return;
  const bool first = (int)blockIdx.x == a;
  u64* mine = buf + (first ? 0 : 64);    // separate 512-byte regions
  u64* theirs = buf + (first ? 64 : 0);
  bool ok = true;
  const long long t0 = wall_clock64();
  for (int i = 1; i <= iters && ok; ++i) {
    if (first) {
      put<SSCOPE>(mine, i, i);
      ok = get<SCOPE>(theirs, i);
    } else {
      ok = get<SCOPE>(theirs, i);
      put<SSCOPE>(mine, i, i);
    }
  }
  const long long t1 = wall_clock64();
  if (first) { out[0] = t1 - t0; out[1] = ok ? 1 : 0; }
}

int main() {
  const int G = 64, iters = 2000;
  u64* buf; long long* out; int* xcc;
  hipMalloc(&buf, 4096); hipMalloc(&out, 64); hipMalloc(&xcc, G * sizeof(int));
  int hx[G]; long long ho[2];
  for (int mode = 0; mode < 3; ++mode) {
    for (int b : {8, 1, 16, 4}) {   // partner of block 0: +8 / +16 = same XCD if round-robin over 8 XCDs; 1 / 4 = other XCD
      hipMemset(buf, 0, 4096); hipMemset(out, 0, 64);
      if (mode == 0) hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_AGENT>, dim3(G), dim3(64), 0, 0, buf, 0, b, iters, out, xcc);
      else if (mode == 1) hipLaunchKernelGGL(pingpong<__HIP_MEMORY_SCOPE_WORKGROUP>, dim3(G), dim3(64), 0, 0, buf, 0, b, iters, out, xcc);
      // mode 2: stores at workgroup scope (the line STAYS in the XCD's L2), loads at agent scope (bypass L1, served by that L2):
      // coherent only when both blocks share an XCD
      else hipLaunchKernelGGL((pingpong<__HIP_MEMORY_SCOPE_AGENT, __HIP_MEMORY_SCOPE_WORKGROUP>), dim3(G), dim3(64), 0, 0, buf, 0, b, iters, out, xcc);
      hipDeviceSynchronize();
      hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost); hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost);
      printf("mode %d (%s) blocks 0<->%d  xcc %d/%d  ok=%lld  round trip %.3f us\n", mode, mode == 0 ? "agent scope" : mode == 1 ? "workgroup scope" : "wg-scope store + agent-scope load", b,
             hx[0], hx[b], ho[1], ho[0] / 100.0 / iters);
    }
  }
  printf("xcc of blocks 0..15:");
  for (int i = 0; i < 16; ++i) printf(" %d", hx[i]);
  printf("\n");
  return 0;
}
