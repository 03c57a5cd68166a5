// common.h -- shared host/device helpers for libtaco_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/taco_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- host-side error plumbing -----------------------------------------------------------------------------
void taco_set_error(const char* fmt, ...);

#define TACO_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      taco_set_error(__VA_ARGS__);     \
      return TACO_EINVAL;              \
    }                                  \
  } while (0)

#define TACO_LAUNCH_CHECK(what)                                             \
  do {                                                                      \
    hipError_t e__ = hipGetLastError();                                     \
    if (e__ != hipSuccess) {                                                \
      taco_set_error("%s: %s", what, hipGetErrorString(e__));               \
      return TACO_ELAUNCH;                                                  \
    }                                                                       \
  } while (0)

#define TACO_TRY(expr)          \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != TACO_OK) return rc__; \
  } while (0)

// Model constants (reference Config, tacotron.py:12-33; CBHG widths ops.py:48-49, tacotron.py:131,147).
constexpr int kEmbed = 256;
constexpr int kPre1 = 256;
constexpr int kPre2 = 128;
constexpr int kCb = 128;    // CBHG channel width / GRU units
constexpr int kAtt = 256;   // attention_units
constexpr int kDec = 256;   // decoder_units
constexpr int kMel = 80;
constexpr int kFft = 1025;
constexpr float kBnEps = 1e-3f;

// ---- device math ------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanh_f(float x) {
  // 2/(1+e^-2x) - 1 : absolute error ~1e-7, saturates cleanly (e^-2x -> inf gives -1, -> 0 gives 1).
  return 2.0f / (1.0f + expf(-2.0f * x)) - 1.0f;
}

// Attention inner loops only: tanh on the hardware exp / rcp units (v_exp_f32, v_rcp_f32; ~1 ulp each), absolute error
// ~2e-7 -- 51 K tanh per row-step make the IEEE-division version the largest VALU cost of the attention phases.
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}

// Wave64 reductions on the DPP path (gfx9 row_ror / row_bcast: plain VALU moves, no LDS-crossbar ds_bpermute as __shfl_xor
// uses): quad_perm swaps + row_ror 4/8 leave every lane of a 16-lane row with the row total; row_bcast:15 / :31 then fold the
// four rows so lanes 48..63 hold the wave total, which v_readlane broadcasts.  ~6 VALU ops instead of 6 dependent shuffles.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_move(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_move<0xb1>(0.f, v);          // quad_perm [1,0,3,2]
  v += dpp_move<0x4e>(0.f, v);          // quad_perm [2,3,0,1]
  v += dpp_move<0x124>(0.f, v);         // row_ror:4
  v += dpp_move<0x128>(0.f, v);         // row_ror:8
  v += dpp_move<0x142, 0xa>(0.f, v);    // row_bcast:15 -> rows 1,3
  v += dpp_move<0x143, 0xc>(0.f, v);    // row_bcast:31 -> rows 2,3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_move<0xb1>(v, v));
  v = fmaxf(v, dpp_move<0x4e>(v, v));
  v = fmaxf(v, dpp_move<0x124>(v, v));
  v = fmaxf(v, dpp_move<0x128>(v, v));
  v = fmaxf(v, dpp_move<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_move<0x143, 0xc>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Workgroup barrier for LDS-only communication: waits for this wave's LDS traffic (lgkmcnt) but NOT for its outstanding
// global loads/stores (vmcnt), unlike __syncthreads(), so prefetch loads and stash stores stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ float apply_act(float x, int act) {
  switch (act) {
    case TACO_ACT_RELU: return fmaxf(x, 0.0f);
    case TACO_ACT_SIGMOID: return sigmoid_f(x);
    case TACO_ACT_TANH: return tanh_f(x);
    default: return x;
  }
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of the (function, DEVICE) pair: cache the "already raised" fact per
// device, not per process (a second GPU used by the same process would otherwise launch without it and fail).  `done` is the
// call site's own static table.
struct DynSmemOnce {
  bool ok[32] = {};
};
static inline bool ensure_dyn_smem(DynSmemOnce& done, const void* func, size_t bytes) {
  if (bytes <= 64 * 1024) return true;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (dev >= 0 && dev < 32 && done.ok[dev]) return true;
  if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  if (dev >= 0 && dev < 32) done.ok[dev] = true;
  return true;
}

// ---- tail events (layout.hip): cross-stream dependencies without marker packets --------------------------------------
// A fork / join between two streams used to be hipEventRecord on the producing stream + hipStreamWaitEvent on the consuming one.
// The record is a marker packet with a system-scope release BETWEEN two kernels of the producing stream: 7-12 us of bubble per
// fork in the step's timeline (~17 of them on the critical path), 4.6 us with empty kernels in tools/micro/event_gap.hip
// (profiles/r06_event_gap.txt: record + wait 6.5 us per kernel vs 1.9 plain; hipExtLaunchKernelGGL's stop event + wait 3.0).
// While a TailScope is open (taco_forward / taco_backward / taco_infer; not while the stream is being captured), every launch
// of the library carries a stop event of a per-stream ring on its OWN dispatch packet; "everything enqueued on s so far" is then
// the event of the last kernel launched on s (streams are in order), and a fork / join waits for that -- no marker.  Whatever
// else is enqueued on a stream (memset, event wait) "touches" it: its tail event no longer covers the stream and the next
// fork falls back to a recorded event.  TACO_TAIL_EVENTS=0: recorded events everywhere (A/B runs).
hipEvent_t taco_tail_take(hipStream_t s, hipEvent_t* start);   // the stop event the NEXT launch on s carries (nullptr: none) and, for a launch bracketed by the profiling ring, its start event
hipEvent_t taco_tail_event(hipStream_t s, hipStream_t for_stream);   // the event of the LAST launch on s if it covers everything `for_stream` has to wait for, else nullptr
hipEvent_t taco_tail_steal(hipStream_t s, hipEvent_t give, bool* owned);   // tail event for a longer-lived use: taken OUT of the ring (*owned; `give` refills the slot) or an alias of an event the ring does not own; nullptr: none
void taco_tail_touch(hipStream_t s);           // something that is not a library launch was enqueued on s
void taco_tail_open(uint64_t key);              // opens the scope of one C-ABI call; key = kind + shape of the call (its launch plan, layout.hip)
void taco_tail_close();
bool taco_tail_wait(hipStream_t waiter, hipStream_t producer);   // waiter waits for producer's tail event; false: caller records an event
// profiling ring (model.hip): inside a scope a bracket's start / stop events ride on the bracketed launch itself instead of two markers
bool taco_tail_arm_timing(hipStream_t s, hipEvent_t start, hipEvent_t stop);   // false: no scope, the caller records `start`
int taco_tail_disarm_timing(hipStream_t s);    // launches on s since the arm (1: the pair rode on that launch; else the caller records what is missing)

#define TACO_KLAUNCH(kernel, grid, block, smem, stream, ...)                                                        \
  do {                                                                                                              \
    hipEvent_t tst__ = nullptr;                                                                                     \
    hipEvent_t tev__ = taco_tail_take(stream, &tst__);                                                              \
    if (tev__) hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, tst__, tev__, 0, __VA_ARGS__);              \
    else hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                                        \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// TACO_DETERMINISTIC=1: every reduction that is otherwise combined with fp32 atomics (split-M weight gradients, the K-way conv
// bank input gradient, BN / bias column sums, the embedding scatter) runs in a fixed order instead -- slower, but two runs of
// the same step give bit-identical gradients (needed to bisect a training divergence).  Read on every call.
static inline bool taco_deterministic() {
  const char* e = getenv("TACO_DETERMINISTIC");
  return e && atoi(e) != 0;
}
