// event_gap.hip -- what a cross-stream dependency costs the PRODUCING stream on gfx950 / ROCm 7.2.
// A chain of N short kernels on stream A; between consecutive kernels one of:
//   plain      nothing
//   record     hipEventRecord(ev, A)                       (hipEventDisableTiming)              -- what side_fork / tn_route do
//   nofence    the same, event created with hipEventDisableSystemFence
//   todevice   the same, event created with hipEventReleaseToDevice
//   extstop    no marker: the event rides on the kernel's own packet (hipExtLaunchKernelGGL stopEvent)
//   *+wait     ... and stream B waits for the event and runs a short kernel (the realistic fork)
//   joinold    A waits for an event recorded on B long ago (the realistic late join)
// Prints the chain's wall time per kernel minus the kernel's own spin time.
// build: hipcc --offload-arch=gfx950 -O3 tools/micro/event_gap.hip -o build/event_gap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#include <chrono>

__global__ void spin_kernel(long long ticks, float* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (threadIdx.x == 0) out[blockIdx.x] = (float)t0;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void stamp_kernel(long long ticks, long long* out) {   // out[0] = start, out[1] = end (100 MHz wall clock)
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) { }
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t0; out[1] = wall_clock64(); }
}

// Does a stream that waits for a stop event (never passed to hipEventRecord) really wait?  A (2 ms) on s1 carries e; s2 waits
// for e and runs B.  Then the event is re-used by a second launch A2 (4 ms) on s1 BEFORE B can have started: B must still start
// behind A (the wait captured A's command) -- behind A2 as well would be correct but later than necessary.
static int dependency_check(hipStream_t s1, hipStream_t s2) {
  long long* st;
  if (hipMalloc(&st, 6 * sizeof(long long)) != hipSuccess) return 1;
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return 1;
  int bad = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(st, 0, 6 * sizeof(long long));
    hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s1, nullptr, e, 0, 200000LL, st);        // A: 2 ms
    if (hipStreamWaitEvent(s2, e, 0) != hipSuccess) return 1;
    hipExtLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s1, nullptr, e, 0, 400000LL, st + 2);    // A2: 4 ms, same event
    hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s2, 10LL, st + 4);                           // B
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    long long h[6];
    (void)hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    const double a_end = h[1] / 100.0, a2_end = h[3] / 100.0, b_start = h[4] / 100.0, a_start = h[0] / 100.0;
    printf("dependency check %d: A ends at %.1f us, B starts at %.1f us, A2 ends at %.1f us (since A's start) -> %s\n", rep, a_end - a_start,
           b_start - a_start, a2_end - a_start,
           b_start >= a_end ? (b_start < a2_end ? "B waited for A (not for A2): OK" : "B waited for A2 as well: correct, late") : "B DID NOT WAIT");
    if (b_start < a_end) bad = 1;
  }
  return bad;
}

// Is what the producing launch wrote VISIBLE to a launch on another stream that only waited for its stop event (no marker, no
// system-scope fence)?  The producer fills 64 MB with a per-round pattern from workgroups on every XCD; the consumer -- other
// workgroup-to-XCD assignment (reversed block order) -- reads it back and counts mismatches; the consumer's own result buffer is
// then overwritten by the next round's producer on the first stream, ordered the same way in the other direction.
__global__ void fill_kernel(unsigned* buf, size_t n, unsigned pat) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = pat ^ (unsigned)i;
}
__global__ void verify_kernel(const unsigned* buf, size_t n, unsigned pat, unsigned* bad) {
  unsigned miss = 0;
  const size_t nb = gridDim.x;
  for (size_t i = (size_t)(nb - 1 - blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += nb * blockDim.x) miss += buf[i] != (pat ^ (unsigned)i);
  if (miss) atomicAdd(bad, miss);
}
static int visibility_check(hipStream_t s1, hipStream_t s2) {
  const size_t n = (size_t)16 << 20;   // 64 MB: 16 x the L2 of one XCD
  unsigned *buf, *bad;
  if (hipMalloc(&buf, n * 4) != hipSuccess || hipMalloc(&bad, 4) != hipSuccess) return 1;
  (void)hipMemset(bad, 0, 4);
  hipEvent_t e1, e2;
  if (hipEventCreateWithFlags(&e1, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&e2, hipEventDisableTiming) != hipSuccess) return 1;
  (void)hipDeviceSynchronize();
  const int rounds = 200;
  for (int r = 0; r < rounds; ++r) {
    const unsigned pat = 0x9e3779b9u * (unsigned)(r + 1);
    if (r) (void)hipStreamWaitEvent(s1, e2, 0);   // the producer overwrites only after the previous round's consumer has read
    hipExtLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, s1, nullptr, e1, 0, buf, n, pat);
    (void)hipStreamWaitEvent(s2, e1, 0);
    hipExtLaunchKernelGGL(verify_kernel, dim3(2048), dim3(256), 0, s2, nullptr, e2, 0, (const unsigned*)buf, n, pat, bad);
  }
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  unsigned h = 1;
  (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("visibility check: %d rounds of 64 MB written on one stream, verified on another behind its stop event: %u stale words\n", rounds, h);
  return h != 0;
}

int main() {
  {
    hipStream_t q1, q2;
    CK(hipStreamCreateWithFlags(&q1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&q2, hipStreamNonBlocking));
    if (dependency_check(q1, q2)) { printf("DEPENDENCY CHECK FAILED\n"); return 2; }
    if (visibility_check(q1, q2)) { printf("VISIBILITY CHECK FAILED\n"); return 3; }
  }
  const int N = 40, REP = 5;
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  float* out;
  CK(hipMalloc(&out, 4096 * 4));
  const long long ticks = 2000;   // wall_clock64 runs at 100 MHz: 20 us
  const dim3 grid(256), block(256);
  struct Case { const char* name; unsigned flags; int mode; bool wait; };   // mode 0 plain, 1 record, 2 ext stop event, 3 join old
  const Case cases[] = {
      {"plain", 0, 0, false},
      {"record", hipEventDisableTiming, 1, false},
      {"record+wait", hipEventDisableTiming, 1, true},
      {"nofence", hipEventDisableTiming | hipEventDisableSystemFence, 1, false},
      {"nofence+wait", hipEventDisableTiming | hipEventDisableSystemFence, 1, true},
      {"todevice", hipEventDisableTiming | hipEventReleaseToDevice, 1, false},
      {"todevice+wait", hipEventDisableTiming | hipEventReleaseToDevice, 1, true},
      {"extstop", hipEventDisableTiming, 2, false},
      {"extstop+wait", hipEventDisableTiming, 2, true},
      {"extstop-nofence+wait", hipEventDisableTiming | hipEventDisableSystemFence, 2, true},
      {"joinold", hipEventDisableTiming, 3, false},
      {"joinold-nofence", hipEventDisableTiming | hipEventDisableSystemFence, 3, false},
  };
  for (const Case& c : cases) {
    std::vector<hipEvent_t> ev(N);
    for (auto& e : ev) CK(hipEventCreateWithFlags(&e, c.flags ? c.flags : hipEventDisableTiming));
    double best = 1e30;
    for (int rep = 0; rep < REP; ++rep) {
      if (c.mode == 3) {   // events recorded on B up front, long done when A reaches its waits
        for (int i = 0; i < N; ++i) {
          hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, B, 10LL, out + 2048);
          CK(hipEventRecord(ev[i], B));
        }
        CK(hipStreamSynchronize(B));
      }
      CK(hipDeviceSynchronize());
      const auto t0 = std::chrono::steady_clock::now();
      for (int i = 0; i < N; ++i) {
        if (c.mode == 3) CK(hipStreamWaitEvent(A, ev[i], 0));
        if (c.mode == 2) hipExtLaunchKernelGGL(spin_kernel, grid, block, 0, A, nullptr, ev[i], 0, ticks, out);
        else hipLaunchKernelGGL(spin_kernel, grid, block, 0, A, ticks, out);
        if (c.mode == 1) CK(hipEventRecord(ev[i], A));
        if (c.wait) {
          CK(hipStreamWaitEvent(B, ev[i], 0));
          hipLaunchKernelGGL(spin_kernel, dim3(8), dim3(64), 0, B, 200LL, out + 1024);
        }
      }
      CK(hipStreamSynchronize(A));
      const auto t1 = std::chrono::steady_clock::now();
      CK(hipDeviceSynchronize());
      const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
      if (us < best) best = us;
    }
    printf("%-24s %7.2f us per kernel (spin 20.00)  -> overhead %6.2f us\n", c.name, best / N, best / N - 20.0);
    for (auto& e : ev) (void)hipEventDestroy(e);
  }
  return 0;
}
