// decoder.hip -- persistent attention-decoder kernels (tacotron.py:46-105 create_decoder + :134-138 dynamic_decode,
// with the TF-r1.2 AttentionWrapper / BahdanauAttention / GRUCell / projection-wrapper / helper semantics restated
// in SURVEY.md §8a rows a11-a15).
//
// ONE launch replaces the reference's 180-iteration tf.while_loop (~13k op launches).  A CLUSTER of P workgroups
// (512 threads each; P = 8, or 16 at inference when they fit) owns ONE batch row for ALL Td steps; workgroup `peer` of the
// cluster computes the column slice [peer*N/P, (peer+1)*N/P) of every mat-vec.  Block b -> (row = b / P, peer = b % P);
// the dispatcher places block b on XCD b % 8 (measured, tools/micro/pingpong.hip; a speed assumption only), so every XCD's
// L2 holds just its 1/8 slice of the decoder weight set, shared by all rows, instead of thrashing on the whole set
// (round-1 v0 measured 10.5 GB/launch of L2 misses).  All recurrent state is replicated in each peer's LDS; after every
// round the peers all-gather the round's output vectors through 8-byte {epoch tag, value} granules written with ONE
// agent-scope store and polled with agent-scope loads (MI355X guide, Guideline 16 recipe R2: the data is the flag, no
// fence, placement independent; one hop measures 0.3-0.5 us).  Every spin is bounded; a timeout raises `err` and the
// launch drains.
//
// A step is 8 exchange rounds forward and 9 backward: wherever two linear maps of the reference's cell follow each other
// without a nonlinearity (attention layer -> input projection -> GRU-1 gates, output projection -> query layer / next
// pre_net), the host forms their product once per call (DecComposite, model.hip) and the kernel runs them as one round;
// the context (linear in the alignments, consumed only by the next step's input projection) is folded the same way against the
// row's own attention memory (VWx / VWg), which removes its mat-vec and all-gather from the step altogether.
// The next round's first weight rows are prefetched into registers while the all-gather is in flight.
//
// The backward kernel walks the steps in reverse with the same structure on pre-transposed weights and emits the
// per-step pre-activation gradients ("gstash"); all weight gradients are then dense MFMA GEMMs over B*Td rows.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int NT = 512;
constexpr int kPartFloats = NT * 8;   // two partial-sum regions of NT*4 floats (merged rounds run two mat-vecs per barrier)
constexpr int kPartRegion = NT * 4;
constexpr int kAR = 4;   // attention memory rows kept register-resident per wave
constexpr int kKR = 13;  // rows per thread kept resident by the unit-split energy backward (NSG = 16: Tt <= 208)

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) int gi32;

// TIMING PROBES (garbage results, real timing; tools/dec_probe.py) exist only in the probe build (-DTACO_DEC_PROBES ->
// libtaco_probe.so): compiled into the production kernels their branches cost 1.4 us per step (instruction-cache footprint).
#ifdef TACO_DEC_PROBES
constexpr bool kProbes = true;
#else
constexpr bool kProbes = false;
#endif

struct Xchg {
  u64* base;        // this row's granule area
  unsigned epoch;   // step tag, never 0
  int P, peer;
  int lgP;          // log2(P)
  int* err;         // global error word
  int* dead;        // LDS: set once this workgroup has given up polling
  long long* trace; // optional: 4 wall_clock64 stamps per phase (enabled for one workgroup at one step), else null
  int tslot;
  int fake;         // probe build only.  1 (TACO_DEC_FAKEW): every weight row aliases row 0 (L1 hits); 2 (TACO_DEC_FAKEX): no polling;
                    // 8 (TACO_DEC_NOPF): prefetched rows cost no memory traffic; 16 (TACO_DEC_NOLIVE): nor do the live rows
};

__device__ __forceinline__ bool probe(const Xchg& X, int bit) { return kProbes && (X.fake & bit) != 0; }

// stamp k (0 entry, 1 partials done, 2 own slice published, 3 gather done) of the current phase
__device__ __forceinline__ void tstamp(const Xchg& X, int k) {
  if (X.trace && threadIdx.x == 0) X.trace[X.tslot * 4 + k] = wall_clock64();
}

__device__ __forceinline__ void xput(const Xchg& X, int idx, float v) {
  __hip_atomic_store((gu64*)(X.base + idx), ((u64)X.epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float xget(const Xchg& X, int idx) {
  if (*X.dead) return 0.f;
  if (probe(X, 2)) return 0.f;   // timing probe: no polling at all
  gu64* g = (gu64*)(X.base + idx);
  for (unsigned spin = 0;; ++spin) {
    const u64 x = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((unsigned)(x >> 32) == X.epoch) return __uint_as_float((unsigned)x);
    if ((spin & 1023u) == 1023u) {
      if (spin > (1u << 23) || __hip_atomic_load((gi32*)X.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store((gi32*)X.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *X.dead = 1;
        return 0.f;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// free-form trace mark (slot 40+id) for sections that are not phase() calls
__device__ __forceinline__ void tmark(const Xchg& X, int id) {
  if (X.trace && threadIdx.x == 0) {
    X.trace[160 + id] = wall_clock64();
    if (id == 0 || id == 5) X.trace[190 + (id != 0)] = clock64();   // shader-cycle counter: effective clock = dcycles / dwall
  }
}

// Two granules polled concurrently (a merged round gathers two vectors: polling them one after the other would cost two
// dependent fabric round trips).  needA / needB: whether this thread has a granule to fetch at all.
__device__ __forceinline__ void xget2(const Xchg& X, int idxA, bool needA, int idxB, bool needB, float& vA, float& vB) {
  vA = vB = 0.f;
  if (*X.dead) return;
  if (probe(X, 2)) return;
  gu64* gA = (gu64*)(X.base + idxA);
  gu64* gB = (gu64*)(X.base + idxB);
  for (unsigned spin = 0; needA || needB; ++spin) {
    u64 xa = 0, xb = 0;
    if (needA) xa = __hip_atomic_load(gA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (needB) xb = __hip_atomic_load(gB, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (needA && (unsigned)(xa >> 32) == X.epoch) { vA = __uint_as_float((unsigned)xa); needA = false; }
    if (needB && (unsigned)(xb >> 32) == X.epoch) { vB = __uint_as_float((unsigned)xb); needB = false; }
    if (!(needA || needB)) break;
    if ((spin & 1023u) == 1023u) {
      if (spin > (1u << 23) || __hip_atomic_load((gi32*)X.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        __hip_atomic_store((gi32*)X.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *X.dead = 1;
        return;
      }
    }
    __builtin_amdgcn_s_sleep(1);
  }
}

// One mat-vec phase of the cluster:  y[n] = sum_k x[k] * W[k*ldw + n]  for this peer's column slice, then
//   v = epi(n, y) (owner only: activation, stash writes), put(n, v) on EVERY peer (LDS state update) after the all-gather.
// x lives in LDS and must be readable up to K rounded up to 4.  N % 4 == 0 and N/4 >= P.  Contains ONE lds_barrier();
// the caller must __syncthreads() afterwards before `part`/x/the put() targets are reused.
// ---- one mat-vec phase of the cluster, in three parts so that independent mat-vecs can share an exchange round ----
//   y[n] = sum_k x[k] * W[k*ldw + n]  for this peer's column slice;  v = epi(n, y) on the owner (activation, stash writes);
//   put(n, v) on EVERY peer (LDS state update) once the value is known locally or gathered.
// x lives in LDS and must be readable up to K rounded up to 4.  N % 4 == 0 and N/4 >= P.
// threadIdx.x behind an opaque move: everything a phase derives from it (slice coordinates, weight pointers) is then
// recomputed per phase instead of being hoisted out of the step loop into ~100 long-lived registers
__device__ __forceinline__ int opaque_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
struct Slice {
  int g0, n4, nloc, nbeg, rows;
  int lg4;    // log2(n4)
  int lgKG;   // log2(k-groups)
  bool shfl;
};
// Every mat-vec width N in these kernels is a power of two >= 4*P and P is a power of two (pick_cluster), so slices are
// shifts: no integer division anywhere on the step path.
__device__ __forceinline__ Slice slice_of(const Xchg& X, int N) {
  Slice s;
  s.lg4 = (31 - __builtin_clz(N >> 2)) - X.lgP;
  s.n4 = 1 << s.lg4;
  s.g0 = X.peer << s.lg4;
  s.nloc = s.n4 * 4;
  s.nbeg = s.g0 * 4;
  // k-groups: the NT / n4 threads with equal c4 split K.  n4 <= 64: the lanes of a wave that share c4 are reduced with
  // __shfl_xor and only 8 per-wave partials reach LDS; wider slices (small clusters) keep NT / n4 <= 4 partial rows.
  s.shfl = s.lg4 <= 6;
  s.lgKG = 9 - s.lg4;   // NT = 512
  s.rows = s.shfl ? NT / 64 : (1 << s.lgKG);
  return s;
}

// Cross-round weight prefetch: the first kPF rows (float4 each) of this thread's k-chunk of the NEXT mat-vec, fetched
// into registers right after the current round's slice has been published, i.e. while the all-gather is in flight.  The
// weights are data independent, so the next mat-vec then starts as pure FMAs instead of an exposed L2 round trip.
constexpr int kPF = 8;
struct Pref {
  float4 w[kPF];
  const float* W = nullptr;   // which matrix the registers hold (tag checked by phase_mv)
};
__device__ __forceinline__ void prefetch_w(Pref& pf, const float* __restrict__ W, int ldw, int K, int N, const Xchg& X) {
  const int tid = opaque_tid();
  const Slice S = slice_of(X, N);
  const int kg = tid >> S.lg4, c4 = tid & (S.n4 - 1);
  pf.W = W;
  if (probe(X, 8)) return;   // timing probe: the prefetched rows cost no memory traffic at all (registers keep stale values)
  {
    const int Kc = (((K + (1 << S.lgKG) - 1) >> S.lgKG) + 3) & ~3;
    const int k0 = kg * Kc;
    const int k1 = min(K, k0 + Kc);
    const float* wp = W + (S.g0 + c4) * 4;
#pragma unroll
    for (int i = 0; i < kPF; ++i) {
      const bool ok = k0 + i < k1;
      pf.w[i] = *reinterpret_cast<const float4*>(wp + (int64_t)(ok && !(probe(X, 1)) ? k0 + i : 0) * ldw);   // clamped address, value unused if !ok
    }
  }
}

// Mat-vec of one (W, x) segment accumulated into `acc` (the thread's four columns over its k-chunk of THIS segment); several
// segments with different matrices may feed one output vector (round G0: shared weights for [p2 ; out ; h1], the row's own
// VW for the alignments).  mv_store then reduces over the k-groups of the wave and leaves the per-wave partials in `part`.
// lw / lrows: launch-resident copy (LDS, one private float4 slot per thread and row: lw[i * NT + tid]) of rows kPF .. kPF+lrows-1
// of this thread's k-chunk -- they follow the prefetched rows without touching the vector-memory pipe (lres_fill).
template <bool PF>
__device__ __forceinline__ float4 mv_accum(const float* __restrict__ W, int ldw, int K, int N, const float* x, const Xchg& X,
                                           const Pref& pf, float4 acc, const float4* lw = nullptr, int lrows = 0) {
  if (probe(X, 1)) ldw = 0;
  const int tid = opaque_tid();
  const Slice S = slice_of(X, N);
  const int n4 = S.n4;
  const int kg = tid >> S.lg4, c4 = tid & (n4 - 1);
  {
    const int Kc = (((K + (1 << S.lgKG) - 1) >> S.lgKG) + 3) & ~3;
    const int k0 = kg * Kc;
    const int k1 = min(K, k0 + Kc);
    const float* wp = W + (int64_t)k0 * ldw + (S.g0 + c4) * 4;
    int k = k0;
    if (PF && pf.W == W) {   // rows k0 .. k0+kPF-1 are already in registers (as far as whole groups of 4 exist)
#pragma unroll
      for (int g = 0; g < kPF / 4; ++g) {
        if (k + 3 < k1) {
          const float4 xv = *reinterpret_cast<const float4*>(x + k);
          const float4 w0 = pf.w[4 * g + 0], w1 = pf.w[4 * g + 1], w2 = pf.w[4 * g + 2], w3 = pf.w[4 * g + 3];
          acc.x = fmaf(xv.x, w0.x, acc.x); acc.y = fmaf(xv.x, w0.y, acc.y); acc.z = fmaf(xv.x, w0.z, acc.z); acc.w = fmaf(xv.x, w0.w, acc.w);
          acc.x = fmaf(xv.y, w1.x, acc.x); acc.y = fmaf(xv.y, w1.y, acc.y); acc.z = fmaf(xv.y, w1.z, acc.z); acc.w = fmaf(xv.y, w1.w, acc.w);
          acc.x = fmaf(xv.z, w2.x, acc.x); acc.y = fmaf(xv.z, w2.y, acc.y); acc.z = fmaf(xv.z, w2.z, acc.z); acc.w = fmaf(xv.z, w2.w, acc.w);
          acc.x = fmaf(xv.w, w3.x, acc.x); acc.y = fmaf(xv.w, w3.y, acc.y); acc.z = fmaf(xv.w, w3.z, acc.z); acc.w = fmaf(xv.w, w3.w, acc.w);
          k += 4;
          wp += 4 * (int64_t)ldw;
        }
      }
      if (lw) {
        const int tl = threadIdx.x;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          if (4 * g < lrows && k + 3 < k1) {
            const float4 xv = *reinterpret_cast<const float4*>(x + k);
            const float4 w0 = lw[(4 * g + 0) * NT + tl], w1 = lw[(4 * g + 1) * NT + tl], w2 = lw[(4 * g + 2) * NT + tl],
                         w3 = lw[(4 * g + 3) * NT + tl];
            acc.x = fmaf(xv.x, w0.x, acc.x); acc.y = fmaf(xv.x, w0.y, acc.y); acc.z = fmaf(xv.x, w0.z, acc.z); acc.w = fmaf(xv.x, w0.w, acc.w);
            acc.x = fmaf(xv.y, w1.x, acc.x); acc.y = fmaf(xv.y, w1.y, acc.y); acc.z = fmaf(xv.y, w1.z, acc.z); acc.w = fmaf(xv.y, w1.w, acc.w);
            acc.x = fmaf(xv.z, w2.x, acc.x); acc.y = fmaf(xv.z, w2.y, acc.y); acc.z = fmaf(xv.z, w2.z, acc.z); acc.w = fmaf(xv.z, w2.w, acc.w);
            acc.x = fmaf(xv.w, w3.x, acc.x); acc.y = fmaf(xv.w, w3.y, acc.y); acc.z = fmaf(xv.w, w3.z, acc.z); acc.w = fmaf(xv.w, w3.w, acc.w);
            k += 4;
            wp += 4 * (int64_t)ldw;
          }
        }
      }
    }
    if (probe(X, 16)) {   // timing probe: the live rows cost no memory traffic (same FMAs and LDS reads on a constant)
      const float4 wc = make_float4(1.f, 2.f, 3.f, 4.f);
      for (; k + 3 < k1; k += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(x + k);
        acc.x = fmaf(xv.x, wc.x, acc.x); acc.y = fmaf(xv.x, wc.y, acc.y); acc.z = fmaf(xv.x, wc.z, acc.z); acc.w = fmaf(xv.x, wc.w, acc.w);
        acc.x = fmaf(xv.y, wc.x, acc.x); acc.y = fmaf(xv.y, wc.y, acc.y); acc.z = fmaf(xv.y, wc.z, acc.z); acc.w = fmaf(xv.y, wc.w, acc.w);
        acc.x = fmaf(xv.z, wc.x, acc.x); acc.y = fmaf(xv.z, wc.y, acc.y); acc.z = fmaf(xv.z, wc.z, acc.z); acc.w = fmaf(xv.z, wc.w, acc.w);
        acc.x = fmaf(xv.w, wc.x, acc.x); acc.y = fmaf(xv.w, wc.y, acc.y); acc.z = fmaf(xv.w, wc.z, acc.z); acc.w = fmaf(xv.w, wc.w, acc.w);
      }
    }
#pragma unroll 2
    for (; k + 3 < k1; k += 4) {
      const float4 xv = *reinterpret_cast<const float4*>(x + k);
      const float4 w0 = *reinterpret_cast<const float4*>(wp);
      const float4 w1 = *reinterpret_cast<const float4*>(wp + ldw);
      const float4 w2 = *reinterpret_cast<const float4*>(wp + 2 * (int64_t)ldw);
      const float4 w3 = *reinterpret_cast<const float4*>(wp + 3 * (int64_t)ldw);
      wp += 4 * (int64_t)ldw;
      acc.x = fmaf(xv.x, w0.x, acc.x); acc.y = fmaf(xv.x, w0.y, acc.y); acc.z = fmaf(xv.x, w0.z, acc.z); acc.w = fmaf(xv.x, w0.w, acc.w);
      acc.x = fmaf(xv.y, w1.x, acc.x); acc.y = fmaf(xv.y, w1.y, acc.y); acc.z = fmaf(xv.y, w1.z, acc.z); acc.w = fmaf(xv.y, w1.w, acc.w);
      acc.x = fmaf(xv.z, w2.x, acc.x); acc.y = fmaf(xv.z, w2.y, acc.y); acc.z = fmaf(xv.z, w2.z, acc.z); acc.w = fmaf(xv.z, w2.w, acc.w);
      acc.x = fmaf(xv.w, w3.x, acc.x); acc.y = fmaf(xv.w, w3.y, acc.y); acc.z = fmaf(xv.w, w3.z, acc.z); acc.w = fmaf(xv.w, w3.w, acc.w);
    }
    for (; k < k1; ++k) {
      const float xs = x[k];
      const float4 w0 = *reinterpret_cast<const float4*>(wp);
      wp += ldw;
      acc.x = fmaf(xs, w0.x, acc.x); acc.y = fmaf(xs, w0.y, acc.y); acc.z = fmaf(xs, w0.z, acc.z); acc.w = fmaf(xs, w0.w, acc.w);
    }
  }
  return acc;
}
// Copies rows kPF .. kPF+lrows-1 of every thread's k-chunk of mat-vec (W, K, N) into its private LDS slots (once per launch).
__device__ __forceinline__ void lres_fill(float4* lw, int lrows, const float* __restrict__ W, int ldw, int K, int N, const Xchg& X) {
  const int tid = threadIdx.x;
  const Slice S = slice_of(X, N);
  const int kg = tid >> S.lg4, c4 = tid & (S.n4 - 1);
  const int Kc = (((K + (1 << S.lgKG) - 1) >> S.lgKG) + 3) & ~3;
  const int k0 = kg * Kc + kPF, k1 = min(K, kg * Kc + Kc);
  const float* wp = W + (S.g0 + c4) * 4;
  for (int i = 0; i < lrows; ++i) lw[i * NT + tid] = *reinterpret_cast<const float4*>(wp + (int64_t)(k0 + i < k1 ? k0 + i : 0) * ldw);
}

// Two (W, x) segments with the same pitch treated as ONE K range [0, K1 + K2) (K1 % 4 == 0): the k-groups split the
// concatenation, so a thread's chunk is ceil((K1+K2)/groups) rows instead of ceil(K1/groups) + ceil(K2/groups), each rounded
// up to 4 -- round G0 (shared weights for [p2 ; out ; h1], the row's own VW for the alignments) is the longest mat-vec of a
// step and every row it does not load is off the critical path.  Groups of four rows never straddle the boundary.
struct Seg2 {
  const float* W1; const float* x1; int K1;
  const float* W2; const float* x2; int K2;
};
__device__ __forceinline__ void prefetch_w2(Pref& pf, const Seg2& g, int ldw, int N, const Xchg& X) {
  const int tid = opaque_tid();
  const Slice S = slice_of(X, N);
  const int kg = tid >> S.lg4, c4 = tid & (S.n4 - 1);
  const int K = g.K1 + g.K2;
  pf.W = g.W1;
  if (probe(X, 8)) return;
  const int Kc = (((K + (1 << S.lgKG) - 1) >> S.lgKG) + 3) & ~3;
  const int k0 = kg * Kc;
  const int k1 = min(K, k0 + Kc);
  const int col = (S.g0 + c4) * 4;
#pragma unroll
  for (int i = 0; i < kPF; ++i) {
    const int k = k0 + i;
    const bool ok = k < k1 && !(probe(X, 1));
    const float* row = (k < g.K1 ? g.W1 + (int64_t)(ok ? k : 0) * ldw : g.W2 + (int64_t)(ok ? k - g.K1 : 0) * ldw);
    pf.w[i] = *reinterpret_cast<const float4*>(row + col);   // clamped address, value unused if !ok
  }
}
template <bool PF>
__device__ __forceinline__ float4 mv_accum2(const Seg2& g, int ldw, int N, const Xchg& X, const Pref& pf, float4 acc) {
  if (probe(X, 1)) ldw = 0;
  const int tid = opaque_tid();
  const Slice S = slice_of(X, N);
  const int kg = tid >> S.lg4, c4 = tid & (S.n4 - 1);
  const int K = g.K1 + g.K2;
  const int Kc = (((K + (1 << S.lgKG) - 1) >> S.lgKG) + 3) & ~3;
  const int k0 = kg * Kc;
  const int k1 = min(K, k0 + Kc);
  const int col = (S.g0 + c4) * 4;
  int k = k0;
  if (PF && pf.W == g.W1) {
#pragma unroll
    for (int q = 0; q < kPF / 4; ++q) {
      if (k + 3 < k1) {
        const float4 xv = *reinterpret_cast<const float4*>(k < g.K1 ? g.x1 + k : g.x2 + (k - g.K1));
        const float4 w0 = pf.w[4 * q + 0], w1 = pf.w[4 * q + 1], w2 = pf.w[4 * q + 2], w3 = pf.w[4 * q + 3];
        acc.x = fmaf(xv.x, w0.x, acc.x); acc.y = fmaf(xv.x, w0.y, acc.y); acc.z = fmaf(xv.x, w0.z, acc.z); acc.w = fmaf(xv.x, w0.w, acc.w);
        acc.x = fmaf(xv.y, w1.x, acc.x); acc.y = fmaf(xv.y, w1.y, acc.y); acc.z = fmaf(xv.y, w1.z, acc.z); acc.w = fmaf(xv.y, w1.w, acc.w);
        acc.x = fmaf(xv.z, w2.x, acc.x); acc.y = fmaf(xv.z, w2.y, acc.y); acc.z = fmaf(xv.z, w2.z, acc.z); acc.w = fmaf(xv.z, w2.w, acc.w);
        acc.x = fmaf(xv.w, w3.x, acc.x); acc.y = fmaf(xv.w, w3.y, acc.y); acc.z = fmaf(xv.w, w3.z, acc.z); acc.w = fmaf(xv.w, w3.w, acc.w);
        k += 4;
      }
    }
  }
  if (probe(X, 16)) {
    const float4 wc = make_float4(1.f, 2.f, 3.f, 4.f);
    for (; k + 3 < k1; k += 4) {
      const float4 xv = *reinterpret_cast<const float4*>(k < g.K1 ? g.x1 + k : g.x2 + (k - g.K1));
      acc.x = fmaf(xv.x, wc.x, acc.x); acc.y = fmaf(xv.x, wc.y, acc.y); acc.z = fmaf(xv.x, wc.z, acc.z); acc.w = fmaf(xv.x, wc.w, acc.w);
      acc.x = fmaf(xv.y, wc.x, acc.x); acc.y = fmaf(xv.y, wc.y, acc.y); acc.z = fmaf(xv.y, wc.z, acc.z); acc.w = fmaf(xv.y, wc.w, acc.w);
      acc.x = fmaf(xv.z, wc.x, acc.x); acc.y = fmaf(xv.z, wc.y, acc.y); acc.z = fmaf(xv.z, wc.z, acc.z); acc.w = fmaf(xv.z, wc.w, acc.w);
      acc.x = fmaf(xv.w, wc.x, acc.x); acc.y = fmaf(xv.w, wc.y, acc.y); acc.z = fmaf(xv.w, wc.z, acc.z); acc.w = fmaf(xv.w, wc.w, acc.w);
    }
  }
#pragma unroll 2
  for (; k + 3 < k1; k += 4) {
    const bool first = k < g.K1;
    const float4 xv = *reinterpret_cast<const float4*>(first ? g.x1 + k : g.x2 + (k - g.K1));
    const float* wp = (first ? g.W1 + (int64_t)k * ldw : g.W2 + (int64_t)(k - g.K1) * ldw) + col;
    const float4 w0 = *reinterpret_cast<const float4*>(wp);
    const float4 w1 = *reinterpret_cast<const float4*>(wp + ldw);
    const float4 w2 = *reinterpret_cast<const float4*>(wp + 2 * (int64_t)ldw);
    const float4 w3 = *reinterpret_cast<const float4*>(wp + 3 * (int64_t)ldw);
    acc.x = fmaf(xv.x, w0.x, acc.x); acc.y = fmaf(xv.x, w0.y, acc.y); acc.z = fmaf(xv.x, w0.z, acc.z); acc.w = fmaf(xv.x, w0.w, acc.w);
    acc.x = fmaf(xv.y, w1.x, acc.x); acc.y = fmaf(xv.y, w1.y, acc.y); acc.z = fmaf(xv.y, w1.z, acc.z); acc.w = fmaf(xv.y, w1.w, acc.w);
    acc.x = fmaf(xv.z, w2.x, acc.x); acc.y = fmaf(xv.z, w2.y, acc.y); acc.z = fmaf(xv.z, w2.z, acc.z); acc.w = fmaf(xv.z, w2.w, acc.w);
    acc.x = fmaf(xv.w, w3.x, acc.x); acc.y = fmaf(xv.w, w3.y, acc.y); acc.z = fmaf(xv.w, w3.z, acc.z); acc.w = fmaf(xv.w, w3.w, acc.w);
  }
  for (; k < k1; ++k) {   // (only the tail of the second segment: K1 % 4 == 0)
    const bool first = k < g.K1;
    const float xs = first ? g.x1[k] : g.x2[k - g.K1];
    const float4 w0 = *reinterpret_cast<const float4*>((first ? g.W1 + (int64_t)k * ldw : g.W2 + (int64_t)(k - g.K1) * ldw) + col);
    acc.x = fmaf(xs, w0.x, acc.x); acc.y = fmaf(xs, w0.y, acc.y); acc.z = fmaf(xs, w0.z, acc.z); acc.w = fmaf(xs, w0.w, acc.w);
  }
  return acc;
}
// part: a kPartRegion-float LDS region private to this mat-vec until its phase_fin has run
__device__ __forceinline__ void mv_store(int N, const Xchg& X, float4 acc, float* part) {
  const int tid = opaque_tid();
  const Slice S = slice_of(X, N);
  const int n4 = S.n4, nloc = S.nloc;
  const int kg = tid >> S.lg4, c4 = tid & (n4 - 1);
  if (S.shfl) {
    for (int off = n4; off < 64; off <<= 1) {
      acc.x += __shfl_xor(acc.x, off, 64);
      acc.y += __shfl_xor(acc.y, off, 64);
      acc.z += __shfl_xor(acc.z, off, 64);
      acc.w += __shfl_xor(acc.w, off, 64);
    }
    if ((tid & 63) < n4) *reinterpret_cast<float4*>(part + (tid >> 6) * nloc + c4 * 4) = acc;
  } else {
    *reinterpret_cast<float4*>(part + kg * nloc + c4 * 4) = acc;
  }
}
template <bool PF>
__device__ __forceinline__ void phase_mv_impl(const float* __restrict__ W, int ldw, int K, int N, const float* x,
                                              float* part, const Xchg& X, const Pref& pf, const float4* lw = nullptr, int lrows = 0) {
  mv_store(N, X, mv_accum<PF>(W, ldw, K, N, x, X, pf, make_float4(0.f, 0.f, 0.f, 0.f), lw, lrows), part);
}

// after a workgroup barrier: reduce the partial rows, run the owner epilogue, update local state, publish the slice
__device__ __forceinline__ void phase_mv(const float* __restrict__ W, int ldw, int K, int N, const float* x, float* part,
                                         const Xchg& X) {
  Pref none;
  phase_mv_impl<false>(W, ldw, K, N, x, part, X, none);
}
__device__ __forceinline__ void phase_mv(const float* __restrict__ W, int ldw, int K, int N, const float* x, float* part,
                                         const Xchg& X, const Pref& pf, const float4* lw = nullptr, int lrows = 0) {
  phase_mv_impl<true>(W, ldw, K, N, x, part, X, pf, lw, lrows);
}

template <class Epi, class Put>
__device__ __forceinline__ void phase_fin(int N, const float* part, const Xchg& X, int reg, Epi epi, Put put) {
  const Slice S = slice_of(X, N);
  for (int i = threadIdx.x; i < S.nloc; i += NT) {
    float y0 = 0.f, y1 = 0.f, y2 = 0.f, y3 = 0.f;
    int g = 0;
#pragma unroll 2
    for (; g + 3 < S.rows; g += 4) {
      y0 += part[g * S.nloc + i];
      y1 += part[(g + 1) * S.nloc + i];
      y2 += part[(g + 2) * S.nloc + i];
      y3 += part[(g + 3) * S.nloc + i];
    }
    for (; g < S.rows; ++g) y0 += part[g * S.nloc + i];
    const int n = S.nbeg + i;
    const float v = epi(n, (y0 + y1) + (y2 + y3));
    put(n, v);
    if (X.P > 1) xput(X, reg + n, v);
  }
}

// all-gather of the other peers' slices
template <class Put>
__device__ __forceinline__ void phase_gather(int N, const Xchg& X, int reg, Put put) {
  if (X.P > 1) {
    const Slice S = slice_of(X, N);
    for (int n = threadIdx.x; n < N; n += NT) {
      if (n >= S.nbeg && n < S.nbeg + S.nloc) continue;
      put(n, xget(X, reg + n));
    }
  }
}

// all-gather of two vectors published in the same round (NB <= NT); the two granules of a thread are polled concurrently
template <class PutA, class PutB>
__device__ __forceinline__ void phase_gather2(int NA, int regA, PutA putA, int NB, int regB, PutB putB, const Xchg& X) {
  if (X.P > 1) {
    const Slice SA = slice_of(X, NA), SB = slice_of(X, NB);
    const int n = threadIdx.x;
    const bool needA = n < NA && !(n >= SA.nbeg && n < SA.nbeg + SA.nloc);
    const bool needB = n < NB && !(n >= SB.nbeg && n < SB.nbeg + SB.nloc);
    float vA, vB;
    xget2(X, regA + n, needA, regB + n, needB, vA, vB);
    if (needA) putA(n, vA);
    if (needB) putB(n, vB);
    for (int m = n + NT; m < NA; m += NT) {
      if (m >= SA.nbeg && m < SA.nbeg + SA.nloc) continue;
      putA(m, xget(X, regA + m));
    }
  }
}

// A single mat-vec as its own exchange round.  Contains ONE workgroup barrier; the caller must barrier afterwards before
// `part`/x/the put() targets are reused.
// next mat-vec to prefetch (weights only), or W == nullptr
struct NextMv {
  const float* W = nullptr;
  int ldw = 0, K = 0, N = 0;
};
template <class Epi, class Put>
__device__ __forceinline__ void phase(const float* __restrict__ W, int ldw, int K, int N, const float* x, float* part,
                                      Xchg& X, int reg, Epi epi, Put put, Pref& pf, NextMv nx, const float4* lw = nullptr,
                                      int lrows = 0) {
  tstamp(X, 0);
  phase_mv(W, ldw, K, N, x, part, X, pf, lw, lrows);
  tstamp(X, 1);
  lds_barrier();
  phase_fin(N, part, X, reg, epi, put);
  tstamp(X, 2);
  if (nx.W) prefetch_w(pf, nx.W, nx.ldw, nx.K, nx.N, X);
  phase_gather(N, X, reg, put);
  tstamp(X, 3);
  X.tslot++;
}

template <class Epi, class Put>
__device__ __forceinline__ void phase(const float* __restrict__ W, int ldw, int K, int N, const float* x, float* part,
                                      Xchg& X, int reg, Epi epi, Put put) {
  tstamp(X, 0);
  phase_mv(W, ldw, K, N, x, part, X);
  tstamp(X, 1);
  lds_barrier();
  phase_fin(N, part, X, reg, epi, put);
  tstamp(X, 2);
  phase_gather(N, X, reg, put);
  tstamp(X, 3);
  X.tslot++;
}

// ---- forward exchange regions (granule indices within a row's area) ----
constexpr int XF_P1 = 0, XF_P2 = 256, XF_X = 384, XF_G = 640 /* +l*768 */, XF_C = 1152 /* +l*768 */, XF_O = 2944 /* NO <= 1024 */,
              XF_CTX = 3968, XF_E = 4224;
// ---- backward exchange regions ----
constexpr int XB_FA = 0 /* NO <= 1024 */, XB_DP2 = 1024, XB_P2 = 1152, XB_DQP = 1408, XB_OUT = 1664, XB_C = 1920 /* +l*1024 */,
              XB_G = 2432 /* +l*1024 */, XB_DAL = 7200;
constexpr int kXchgFixed = 7200;

// Forward step, 8 exchange rounds (DecComposite folds the purely linear links of the reference's cell, tacotron.py:54-60,73-76):
//   G0   x = [p2 ; out'] Wx_po + al' VWx + bi   and   gates_1 = sigmoid([p2 ; out' ; h1] Wg0' + al' VWg + bg0')   (' = previous step)
//   C0   candidate_1 / h1            G1 C1 G2 C2  likewise for GRU 2, 3 (plain weights)
//   OUT  [q | cell_output] = (x + h3) [Wo Wq | Wo] + ...    and   pre_net layer 1 of step t+1
//   E    energies (all-gather) + softmax   and   pre_net layer 2 of step t+1
// The AttentionWrapper's attention vector [cell_output ; context] Wa is never formed: it only feeds the next step's input
// projection, and Wa Wi_a is part of Wx.  Neither is the context: context = alignments . values is linear in the alignments and
// only ever enters the next step through Wx_c (and Wx_c Wg0_x), so the host forms VWx = values Wx_c and VWg = values Wx_c Wg0_x
// once per call ((B,Tt,256) and (B,Tt,512), model.hip) and round G0 takes the ALIGNMENTS as an input segment -- which every
// peer already holds after the softmax.  That removes the context mat-vec and its all-gather round from every step.
struct DecSmem {
  float* part;   // kPartFloats
  float* fr;     // 80   pre-net input frame
  float* p1;     // 256
  float* u0;     // 128+80r+256  [p2 ; cell_output ; h1]: shared-weight input segment of round G0
  float* xs;     // 256  in-proj output (residual)
  float* cat;    // 3*512 [layer input ; h_l] (l = 1, 2; slot 0 unused)
  float* catc;   // 512  [layer input ; r*h_l]
  float* us;     // 256
  float* ys;     // 256
  float* qs;     // 256
  float* es;     // TtP energies
  float* als;    // TtP alignments
  float* bias;   // kBiasFloats: every decoder bias, staged once
  float* km1;    // 256 pre-net dropout multipliers (2 / 0, or 1 without dropout) of the current step
  float* km2;    // 128
  int* dead;
};
// bias staging offsets
constexpr int BO_P1 = 0, BO_P2 = 256, BO_IN = 384, BO_G = 640 /* +l*768 */, BO_C = 1152 /* +l*768 */, BO_P1O = 2944, BO_O = 3200;
constexpr int kBiasFloats = 3200 + 1024;
constexpr int kU0Max = kPre2 + kMel * 5 + kAtt + kDec;

__device__ __forceinline__ DecSmem carve(float* base, int TtP) {
  DecSmem s;
  float* p = base;
  s.part = p; p += kPartFloats;
  s.fr = p; p += 80;
  s.p1 = p; p += 256;
  s.u0 = p; p += kU0Max;
  s.xs = p; p += 256;
  s.cat = p; p += 3 * 512;
  s.catc = p; p += 512;
  s.us = p; p += 256;
  s.ys = p; p += 256;
  s.qs = p; p += 256;
  s.es = p; p += TtP;
  s.als = p; p += TtP;
  s.bias = p; p += kBiasFloats;
  s.km1 = p; p += 256;
  s.km2 = p; p += 128;
  s.dead = reinterpret_cast<int*>(p); p += 4;
  return s;
}
constexpr int kFwdSmemFixed =
    kPartFloats + 80 + 256 + kU0Max + 256 + 3 * 512 + 512 + 256 + 256 + 256 + kBiasFloats + 256 + 128 + 4;

// TR: per-phase time stamps compiled in (TACO_DEC_TRACE=1 launches); the production instantiation carries none of it
template <bool TR>
__global__ __launch_bounds__(NT) void decoder_fwd_kernel(DecFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = a.P;
  const int b = blockIdx.x >> (31 - __builtin_clz(P));
  const int B = a.B, Tt = a.Tt, Td = a.Td, r = a.r;
  const int R80 = kMel * r;
  const int KA = kPre2 + R80;          // u0 = [p2 (0) ; out (128) ; h1 (KA)]; rows [0,KA) of Wx, rows [0,KA+256) of Wg0'  
  const int NO = a.c.NO;
  const int TtP = (Tt + 3) & ~3;
  DecSmem S = carve(smem, TtP);
  const DecWeights& w = a.w;
  const DecComposite& cw = a.c;
  Xchg X;
  X.P = P;
  X.lgP = 31 - __builtin_clz(P);
  X.peer = blockIdx.x - b * P;
  X.base = reinterpret_cast<u64*>(a.xchg) + (int64_t)b * (kXchgFixed + TtP);
  X.err = a.err;
  X.dead = S.dead;
  X.epoch = 0;
  X.trace = nullptr;
  X.tslot = 0;
  X.fake = a.fakew;
  const bool lead = X.peer == 0;

  int len = a.text_length[b];
  len = len < 1 ? 1 : (len > Tt ? Tt : len);
  const float* keys = a.keys + (int64_t)b * Tt * kAtt;
  const float* vwx = a.vwx + (int64_t)b * Tt * kDec;        // this row's values . Wx_c        (Tt, 256)
  const float* vwg = a.vwg + (int64_t)b * Tt * 2 * kDec;    // this row's values . Wx_c Wg0_x  (Tt, 512)
  float* const u_out = S.u0 + kPre2;
  float* const h1 = S.u0 + KA;

  // zero state (AttentionWrapper.zero_state, tacotron.py:94): h = 0, attention = 0 (<=> previous output and context 0)
  for (int i = tid; i < 3 * 512; i += NT) S.cat[i] = 0.f;
  for (int i = tid; i < KA + kDec; i += NT) S.u0[i] = 0.f;
  for (int i = tid; i < TtP; i += NT) { S.als[i] = 0.f; S.es[i] = 0.f; }
  if (tid < kMel) S.fr[tid] = a.mel ? a.mel[((int64_t)b * Td) * R80 + kMel * (r - 1) + tid] : 0.f;
  if (tid == 0) {
    *S.dead = 0;
    // placement census (diagnostic, one atomic per workgroup per launch): histogram of (blockIdx - XCC id) mod 8 in the
    // integer words 4..11 of the error region.  The cluster layout is FAST when every workgroup falls in one bin (peer p of
    // every row on the same XCD => each XCD's L2 holds one weight slice); any other placement is still correct, only slower.
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    atomicAdd(a.err + 4 + ((blockIdx.x - (xcc & 7u)) & 7u), 1);
  }
  for (int i = tid; i < kPre1; i += NT) S.bias[BO_P1 + i] = w.pre_b1[i];
  for (int i = tid; i < kPre1; i += NT) S.bias[BO_P1O + i] = cw.bp1o[i];
  for (int i = tid; i < kPre2; i += NT) S.bias[BO_P2 + i] = w.pre_b2[i];
  for (int i = tid; i < kDec; i += NT) S.bias[BO_IN + i] = w.in_b[i];
  for (int l = 0; l < 3; ++l) {
    for (int i = tid; i < 2 * kDec; i += NT) S.bias[BO_G + l * 768 + i] = l == 0 ? cw.bg0[i] : w.gb[l][i];
    for (int i = tid; i < kDec; i += NT) S.bias[BO_C + l * 768 + i] = w.cb[l][i];
  }
  for (int i = tid; i < NO; i += NT) S.bias[BO_O + i] = cw.bo[i];
  if (tid < kPre1) S.km1[tid] = a.keep1 ? (a.keep1[((int64_t)b * Td) * kPre1 + tid] ? 2.f : 0.f) : 1.f;
  if (tid < kPre2) S.km2[tid] = a.keep2 ? (a.keep2[((int64_t)b * Td) * kPre2 + tid] ? 2.f : 0.f) : 1.f;
  const float4 v4 = reinterpret_cast<const float4*>(w.att_v)[lane];
  // This wave always scores the same memory rows s = peer + P*wave + i*P*8: keep the first kAR of them in registers
  // for the whole launch (Tt <= kAR*8*P rows are fully resident); further rows are streamed.
  const int s_first = X.peer + P * wave, s_stride = P * (NT / 64);
  float4 kres[kAR];
#pragma unroll
  for (int i = 0; i < kAR; ++i) {
    const int s = s_first + i * s_stride;
    kres[i] = s < len ? reinterpret_cast<const float4*>(keys + (int64_t)s * kAtt)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // launch-resident second halves of the GRU-2 / GRU-3 gate chunks (free LDS behind the state; sizes chosen by the launcher)
  float4* const lw1 = reinterpret_cast<float4*>(smem + kFwdSmemFixed + 2 * TtP);
  float4* const lw2 = lw1 + a.lres0 * NT;
  lres_fill(lw1, a.lres0, w.gw[1], 2 * kDec, 2 * kDec, 2 * kDec, X);
  lres_fill(lw2, a.lres1, w.gw[2], 2 * kDec, 2 * kDec, 2 * kDec, X);
  lds_barrier();

  // pre_net (tacotron.py:38-44, 64-71) of step tt: two layers, each an exchange round.  Step 0 runs them standalone here;
  // step t+1's layers ride along with step t's OUT / E rounds, which removes two dependent rounds from every step.
  // Layer 1 of a step fed by the previous cell_output (sampled / inference) is computed from (x + h3) with Wo[:, frame] W1.
  auto p1_epi = [&](int64_t btt, int bofs) {
    return [&, btt, bofs](int n, float y) {
      y = fmaxf(y + S.bias[bofs + n], 0.f) * S.km1[n];
      if (a.stash) a.stash[btt * kStRec + kStP1 + n] = y;
      return y;
    };
  };
  auto p2_epi = [&](int64_t btt) {
    return [&, btt](int n, float y) {
      y = fmaxf(y + S.bias[BO_P2 + n], 0.f) * S.km2[n];
      if (a.stash) a.stash[btt * kStRec + kStP2 + n] = y;
      return y;
    };
  };
  auto p1_put = [&](int n, float v) { S.p1[n] = v; };
  auto p2_put = [&](int n, float v) { S.u0[n] = v; };
  Pref pf;   // next round's first weight rows, fetched during the current round's all-gather
  // (a second register set holding the first rows of G0's gate mat-vec, fetched once round E's all-gather has landed so that
  //  the loads fly during the softmax, was measured: 196 VGPRs, 26.9 -> 29.1 us per step)
  const NextMv nx_o{cw.wo, NO, kDec, NO};
  const Seg2 sx{cw.wx, S.u0, KA, vwx, S.als, len};             // x     = [p2 ; out] Wx_po + al VWx
  const Seg2 sg{cw.wg0, S.u0, KA + kDec, vwg, S.als, len};     // gates = [p2 ; out ; h1] Wg0' + al VWg
  {
    const int64_t bt0 = (int64_t)b * Td;
    X.epoch = 0x7fffffffu;   // prologue tag, distinct from every step tag
    if (a.prein && lead && tid < kMel) a.prein[bt0 * kMel + tid] = S.fr[tid];
    phase(w.pre_w1, kPre1, kMel, kPre1, S.fr, S.part, X, XF_P1, p1_epi(bt0, BO_P1), p1_put);
    lds_barrier();
    phase(w.pre_w2, kPre2, kPre1, kPre2, S.p1, S.part, X, XF_P2, p2_epi(bt0), p2_put, pf, NextMv());
    prefetch_w2(pf, sx, kDec, kDec, X);
    lds_barrier();
  }
  // parked one step ahead in registers: dropout multipliers and the teacher frame of step t+1
  float km1n = 1.f, km2n = 1.f, frn = 0.f, p2n = 0.f;
  // teacher-forced steps: the pre-net output p2 is known before the launch (a.pre2, formed by two GEMMs over all B*Td frames,
  // model.hip); the kernel then only loads it, and runs the pre-net riders of the OUT / E rounds on steps fed by the previous output
  const bool hoisted = a.pre2 != nullptr;
  auto park_next = [&](int tn) {   // tn = step whose pre-net inputs are fetched
    km1n = km2n = 1.f;
    frn = 0.f;
    if (tn < Td) {
      if (hoisted && tid < kPre2) p2n = a.pre2[((int64_t)b * Td + tn) * a.ldpre2 + tid];
      if (a.keep1 && tid < kPre1) km1n = a.keep1[((int64_t)b * Td + tn) * kPre1 + tid] ? 2.f : 0.f;
      if (a.keep2 && tid < kPre2) km2n = a.keep2[((int64_t)b * Td + tn) * kPre2 + tid] ? 2.f : 0.f;
      if (a.mel && tid < kMel) frn = a.mel[((int64_t)b * Td + tn) * R80 + kMel * (r - 1) + tid];
    }
  };
  park_next(1);

  for (int t = 0; t < Td; ++t) {
    X.epoch = (unsigned)(t + 1);
    X.trace = (TR && a.trace && blockIdx.x == 0 && t == Td / 2) ? a.trace : nullptr;
    X.tslot = 0;
    const int64_t bt = (int64_t)b * Td + t;
    float* st = a.stash ? a.stash + bt * kStRec : nullptr;
    const bool has_next = t + 1 < Td;
    // helper.next_inputs (TrainingHelper / ScheduledOutputTrainingHelper / InferenceHelper): step t+1 is fed cell_output[t]
    // at inference or when sampled, else mel[t+1]
    const bool from_out = (a.mel == nullptr) || (a.sample && a.sample[(int64_t)t * B + b]);
    // land the parked step-(t+1) values (step t's copies were consumed during step t-1), fetch those of step t+2
    if (tid < kPre1) S.km1[tid] = km1n;
    if (tid < kPre2) S.km2[tid] = km2n;
    if (tid < kMel) S.fr[tid] = frn;
    if (hoisted && tid < kPre2) S.p1[tid] = p2n;   // parked p2 of step t+1 (S.p1 is free until this step's OUT round)
    park_next(t + 2);
    const bool rider = has_next && (from_out || !hoisted);   // pre-net of step t+1 computed in this step's OUT / E rounds

    // ---- round G0: InputProjectionWrapper x = [pre_net ; attention] Wi + bi (tacotron.py:56-59) with the attention layer
    //      and the context folded in (alignments of step t-1 against this row's VWx / VWg), and GRU-1's gates straight from
    //      x's inputs ----
    {
      auto x_epi = [&](int n, float y) {
        y += S.bias[BO_IN + n];
        if (st) st[kStX + n] = y;
        return y;
      };
      auto x_put = [&](int n, float v) {
        S.xs[n] = v;
        S.catc[n] = v;
      };
      auto g_epi = [&](int n, float y) {
        const float g = sigmoid_fast(y + S.bias[BO_G + n]);
        if (n < kDec) {
          const float rh = g * h1[n];
          if (st) { st[kStR + n] = g; st[kStRH + n] = rh; }
          return rh;
        }
        if (st) st[kStU + n - kDec] = g;
        return g;
      };
      auto g_put = [&](int n, float v) {
        if (n < kDec) S.catc[kDec + n] = v;   // r * h
        else S.us[n - kDec] = v;              // u
      };
      tstamp(X, 0);
      {
        // (step 0: the alignments are still the zero state, the VW rows multiply zeros)
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        mv_store(kDec, X, mv_accum2<true>(sx, kDec, kDec, X, pf, z4), S.part);
        mv_store(2 * kDec, X, mv_accum2<false>(sg, 2 * kDec, 2 * kDec, X, pf, z4), S.part + kPartRegion);
      }
      tstamp(X, 1);
      lds_barrier();
      phase_fin(kDec, S.part, X, XF_X, x_epi, x_put);
      phase_fin(2 * kDec, S.part + kPartRegion, X, XF_G, g_epi, g_put);
      tstamp(X, 2);
      prefetch_w(pf, w.cw[0], kDec, 2 * kDec, kDec, X);
      phase_gather2(2 * kDec, XF_G, g_put, kDec, XF_X, x_put, X);
      tstamp(X, 3);
      X.tslot++;
    }
    lds_barrier();
    // ---- MultiRNNCell[GRUCell(256) x3] inside ONE ResidualWrapper (tacotron.py:54-58) ----
    for (int l = 0; l < 3; ++l) {
      float* cl = S.cat + l * 512;                 // [layer input ; h_l] (l >= 1)
      float* hl = l == 0 ? h1 : cl + kDec;
      if (l > 0) {
        phase(w.gw[l], 2 * kDec, 2 * kDec, 2 * kDec, cl, S.part, X, XF_G + l * 768,
              [&](int n, float y) {
                const float g = sigmoid_fast(y + S.bias[BO_G + l * 768 + n]);
                if (n < kDec) {
                  const float rh = g * hl[n];
                  if (st) { st[kStR + l * kDec + n] = g; st[kStRH + l * kDec + n] = rh; }
                  return rh;
                }
                if (st) st[kStU + l * kDec + n - kDec] = g;
                return g;
              },
              [&](int n, float v) {
                if (n < kDec) S.catc[kDec + n] = v;   // r * h
                else S.us[n - kDec] = v;              // u
              },
              pf, NextMv{w.cw[l], kDec, 2 * kDec, kDec}, l == 1 ? lw1 : lw2, l == 1 ? a.lres0 : a.lres1);
        lds_barrier();
      }
      phase(w.cw[l], kDec, 2 * kDec, kDec, S.catc, S.part, X, XF_C + l * 768,
            [&](int n, float y) {
              const float c = tanh_fast(y + S.bias[BO_C + l * 768 + n]);
              const float u = S.us[n];
              const float hn = u * hl[n] + (1.f - u) * c;
              if (st) {
                st[kStC + l * kDec + n] = c;
                st[kStH + l * kDec + n] = hn;
                if (l == 2) st[kStY + n] = S.xs[n] + hn;
              }
              return hn;
            },
            [&](int n, float hn) {
              hl[n] = hn;
              if (l < 2) {
                S.cat[(l + 1) * 512 + n] = hn;
                S.catc[n] = hn;
              } else {
                S.ys[n] = S.xs[n] + hn;
              }
            },
            pf, l < 2 ? NextMv{w.gw[l + 1], 2 * kDec, 2 * kDec, 2 * kDec} : nx_o);
      lds_barrier();
    }
    // ---- round OUT: OutputProjectionWrapper cell_output = (x + h3) Wo + bo (tacotron.py:54-60), the BahdanauAttention
    //      query layer q = cell_output Wq (no bias of its own) folded to (x + h3) Wo Wq + bo Wq, and pre_net layer 1 of
    //      step t+1 ----
    {
      auto o_epi = [&](int n, float y) {
        y += S.bias[BO_O + n];
        if (n < kAtt) {
          if (st) st[kStQ + n] = y;
        } else if (n < kAtt + R80) {
          a.out[bt * R80 + n - kAtt] = y;
        }
        return y;
      };
      auto o_put = [&](int n, float v) {
        if (n < kAtt) {
          S.qs[n] = v;
        } else if (n < kAtt + R80) {
          const int c = n - kAtt;
          u_out[c] = v;
          if (from_out && c >= kMel * (r - 1)) S.fr[c - kMel * (r - 1)] = v;   // next pre-net input = last frame of the group
        }
      };
      const int p1b = from_out ? BO_P1O : BO_P1;
      tstamp(X, 0);
      phase_mv(cw.wo, NO, kDec, NO, S.ys, S.part, X, pf);
      if (rider) {
        if (from_out) phase_mv(cw.wp1o, kPre1, kDec, kPre1, S.ys, S.part + kPartRegion, X);
        else phase_mv(w.pre_w1, kPre1, kMel, kPre1, S.fr, S.part + kPartRegion, X);
      }
      tstamp(X, 1);
      lds_barrier();
      phase_fin(NO, S.part, X, XF_O, o_epi, o_put);
      if (rider) phase_fin(kPre1, S.part + kPartRegion, X, XF_P1, p1_epi(bt + 1, p1b), p1_put);
      tstamp(X, 2);
      if (rider) prefetch_w(pf, w.pre_w2, kPre2, kPre1, kPre2, X);
      else if (has_next) prefetch_w2(pf, sx, kDec, kDec, X);
      phase_gather2(NO, XF_O, o_put, rider ? kPre1 : 0, XF_P1, p1_put, X);
      tstamp(X, 3);
      X.tslot++;
    }
    lds_barrier();
    if (a.prein && lead && has_next && tid < kMel) a.prein[(bt + 1) * kMel + tid] = S.fr[tid];
    // ---- round E: energies e[s] = sum_u v_u tanh(keys[s,u] + q_u): one wave per memory row, rows dealt round-robin to
    //      peers;  same round: pre_net layer 2 of step t+1 ----
    {
      tstamp(X, 0);
      tmark(X, 0);
      const float4 q4 = reinterpret_cast<const float4*>(S.qs)[lane];
      auto score = [&](int s, float4 k4) {
        float e = v4.x * tanh_fast(k4.x + q4.x) + v4.y * tanh_fast(k4.y + q4.y) + v4.z * tanh_fast(k4.z + q4.z) +
                  v4.w * tanh_fast(k4.w + q4.w);
        e = wave_sum(e);
        if (lane == 0) {
          S.es[s] = e;
          if (P > 1) xput(X, XF_E + s, e);
        }
      };
#pragma unroll
      for (int i = 0; i < kAR; ++i) {
        const int s = s_first + i * s_stride;
        if (s < len) score(s, kres[i]);
      }
      for (int s = s_first + kAR * s_stride; s < len; s += s_stride)
        score(s, reinterpret_cast<const float4*>(keys + (int64_t)s * kAtt)[lane]);
      if (rider) {
        phase_mv(w.pre_w2, kPre2, kPre1, kPre2, S.p1, S.part, X, pf);
        lds_barrier();
        phase_fin(kPre2, S.part, X, XF_P2, p2_epi(bt + 1), p2_put);
        prefetch_w2(pf, sx, kDec, kDec, X);
      } else if (has_next) {
        if (tid < kPre2) S.u0[tid] = S.p1[tid];   // the hoisted pre-net output of step t+1 (G0's p2 segment; barrier below)
      }
      tstamp(X, 2);
      tmark(X, 1);
      if (P > 1) {   // energies of the other peers' rows and their slices of p2, polled concurrently
        const Slice SB = slice_of(X, kPre2);
        const bool needA = tid < len && (tid & (P - 1)) != X.peer;
        const bool needB = rider && tid < kPre2 && !(tid >= SB.nbeg && tid < SB.nbeg + SB.nloc);
        float vA, vB;
        xget2(X, XF_E + tid, needA, XF_P2 + tid, needB, vA, vB);
        if (needA) S.es[tid] = vA;
        if (needB) p2_put(tid, vB);
        for (int s = tid + NT; s < len; s += NT)
          if ((s & (P - 1)) != X.peer) S.es[s] = xget(X, XF_E + s);
      }
      tstamp(X, 3);
      tmark(X, 2);
      X.tslot++;
    }
    lds_barrier();
    tmark(X, 3);
    // ---- masked softmax over s < len (score_mask_value = -inf => alignment 0 past text_length) ----
    {
      float m = -INFINITY;
      for (int s = lane; s < len; s += 64) m = fmaxf(m, S.es[s]);
      m = wave_max(m);
      float z = 0.f;
      for (int s = lane; s < len; s += 64) z += __expf(S.es[s] - m);
      z = wave_sum(z);
      const float inv = 1.0f / z;
      for (int s = tid; s < Tt; s += NT) {
        const float al = s < len ? __expf(S.es[s] - m) * inv : 0.f;
        S.als[s] = al;
        if (lead) a.align[bt * Tt + s] = al;
      }
    }
    tmark(X, 4);
    lds_barrier();   // the alignments are the last product of a step: they are an input segment of the next step's round G0
    tmark(X, 5);
  }
}

// ------------------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------------------
// Backward step, 9 exchange rounds, mirroring the folded forward step:
//   FAN   d alignments_t[s] = VWx[s] . dx_{t+1}  (memory rows dealt to peers)   and
//         d p2_{t+1} = dx_{t+1} Wi_p^T  (pre_net of step t+1, one step late)
//   DQ    softmax / energy backward (unit split), dq all-gather   and   d p1_{t+1} = d p2pre_{t+1} W2^T
//   OUT   d(x + h3) = [d out (direct) ; dq ; d p1pre_{t+1} (if step t+1 was fed out_t)] [Wo^T ; (Wo Wq)^T ; (Wo_f W1)^T]
//                     + dx_{t+1} (Wx_o^T Wo^T)      (cell_output_t -> x_{t+1} -> back: a second input segment, a.wdx; the total
//                     d cell_output is only needed for the output projection's weight gradient and is formed by a GEMM afterwards)
//   C2 G2 C1 G1 C0 G0   GRU layers top down; G0 leaves dx_t
// The gradients the folded links skip (d attention, d context, total d cell_output) are only needed for WEIGHT / memory
// gradients and are recovered there from small products (model.hip).
struct DecBwdSmem {
  float* part;    // kPartFloats
  float* dh;      // 3*256 carried dL/dh_l
  float* dx;      // 256  dL/dx of the step processed before (t+1) until round G0 overwrites it with this step's
  float* vo;      // 80r+512  [d cell_output (direct) ; dq ; d p1pre_{t+1}]: input of round OUT
  float* dctx;    // 256
  float* dy;      // 256 d(x + h3)
  float* dht;     // 256 total dL/dh_l at this step
  float* dcp;     // 256
  float* dgp;     // 512
  float* dinp;    // 256 gradient into the layer input
  float* dp2;     // 128
  float* red;     // 8*256 cross-wave dq reduction
  float* rec;     // kRecFloats: this step's forward stash pieces (prefetched one step ahead)
  float* p1prev;  // 256: pre-net layer-1 activations of the step processed before (its backward runs one step late)
  float* p2prev;  // 128: likewise layer 2
  float* als;     // TtP
  float* des;     // TtP
  int* dead;
};
// LDS copy of the forward stash record of step t (+ h of step t-1):
constexpr int RL_P1 = 0, RL_P2 = 256, RL_R = 384, RL_U = 1152, RL_C = 1920, RL_Q = 2688, RL_HP = 2944, kRecFloats = 3712;
constexpr int kRecRegs = (kRecFloats + NT - 1) / NT;   // 8 registers per thread hold the next record in flight
constexpr int kVoMax = kMel * 5 + 512;
constexpr int kBwdSmemFixed = kPartFloats + 768 + 256 + kVoMax + 256 + 256 + 256 + 256 + 512 + 256 + 128 + 8 * 256 + kRecFloats +
                              256 + 128 + 4;

__device__ __forceinline__ DecBwdSmem carve_bwd(float* base, int TtP) {
  DecBwdSmem s;
  float* p = base;
  s.part = p; p += kPartFloats;
  s.dh = p; p += 768;
  s.dx = p; p += 256;
  s.vo = p; p += kVoMax;
  s.dctx = p; p += 256;
  s.dy = p; p += 256;
  s.dht = p; p += 256;
  s.dcp = p; p += 256;
  s.dgp = p; p += 512;
  s.dinp = p; p += 256;
  s.dp2 = p; p += 128;
  s.red = p; p += 8 * 256;
  s.rec = p; p += kRecFloats;
  s.p1prev = p; p += 256;
  s.p2prev = p; p += 128;
  s.als = p; p += TtP;
  s.des = p; p += TtP;
  s.dead = reinterpret_cast<int*>(p); p += 4;
  return s;
}

// Source address of element j of the LDS record for step (b,t): four contiguous runs of the forward stash.
__device__ __forceinline__ float rec_load(const float* stash, int64_t bt, int t, int j) {
  if (j >= kRecFloats) return 0.f;
  const float* st = stash + bt * kStRec;
  if (j < 384) return st[j];                                   // P1 | P2
  if (j < 2688) return st[kStR + (j - 384)];                   // R | U | C  (contiguous in the stash)
  if (j < 2944) return st[kStQ + (j - 2688)];                  // Q
  return t > 0 ? (st - kStRec)[kStH + (j - 2944)] : 0.f;       // H of the previous step
}

template <bool TR>
__global__ __launch_bounds__(NT) void decoder_bwd_kernel(DecBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = a.P;
  const int b = blockIdx.x >> (31 - __builtin_clz(P));
  const int B = a.B, Tt = a.Tt, Td = a.Td, r = a.r;
  const int R80 = kMel * r;
  const int TtP = (Tt + 3) & ~3;
  DecBwdSmem S = carve_bwd(smem, TtP);
  const DecWeights& w = a.wT;
  Xchg X;
  X.P = P;
  X.lgP = 31 - __builtin_clz(P);
  X.peer = blockIdx.x - b * P;
  X.base = reinterpret_cast<u64*>(a.xchg) + (int64_t)b * (kXchgFixed + TtP);
  X.err = a.err;
  X.dead = S.dead;
  X.epoch = 0;
  X.trace = nullptr;
  X.tslot = 0;
  X.fake = a.fakew;
  const bool lead = X.peer == 0;
  float* const dov = S.vo;               // d cell_output (direct part)
  float* const dq = S.vo + R80;
  float* const dp1 = S.vo + R80 + kAtt;

  int len = a.text_length[b];
  len = len < 1 ? 1 : (len > Tt ? Tt : len);
  const float* keys = a.keys + (int64_t)b * Tt * kAtt;
  const float* vwx = a.vwx + (int64_t)b * Tt * kDec;   // this row's values . Wx_c
  float* dkeys = a.dkeys + (int64_t)b * Tt * a.ldk;
  const int ldk = a.ldk;

  for (int i = tid; i < 768; i += NT) S.dh[i] = 0.f;
  if (tid < 256) S.dx[tid] = 0.f;
  for (int i = tid; i < TtP; i += NT) { S.als[i] = 0.f; S.des[i] = 0.f; }
  if (tid == 0) *S.dead = 0;
  // Register-resident attention memory.  d alignments is split by memory ROW: this wave's rows of VWx (dealt as the
  // forward kernel deals the keys).  The energy backward (3c) is split by attention UNIT: this peer owns units [ub, ub+un) of all rows, so
  // dq needs no cross-peer partial sums; thread (ul = tid % un, sg = tid / un) walks rows s = sg, sg+NSG, ... and keeps
  // keys[s][u] and the dkeys[s][u] accumulators of its first kKR rows in registers for the whole launch.
  const int s_first = X.peer + P * wave, s_stride = P * (NT / 64);
  float4 vres[kAR];
#pragma unroll
  for (int i = 0; i < kAR; ++i) {
    const int s = s_first + i * s_stride;
    vres[i] = s < len ? reinterpret_cast<const float4*>(vwx + (int64_t)s * kDec)[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const Slice US = slice_of(X, kAtt);
  const int ub = US.nbeg, un = US.nloc;
  const int NSG = NT >> (US.lg4 + 2);            // row groups (un = 256 / P divides NT)
  const int ul = tid & (un - 1), sg = tid >> (US.lg4 + 2);
  const bool uact = true;
  const float vu = a.att_v[ub + ul];
  float kr[kKR], dkr[kKR];
#pragma unroll
  for (int i = 0; i < kKR; ++i) {
    const int s = sg + i * NSG;
    kr[i] = (uact && s < len) ? keys[(int64_t)s * kAtt + ub + ul] : 0.f;
    dkr[i] = 0.f;
  }
  float dvu = 0.f;                               // d attention_v[ub + ul] partial of this thread
  // Prefetch registers: the record / output gradient / alignment row of the step about to be processed.
  float pre[kRecRegs];
  float pre_dout = 0.f, pre_al = 0.f;
  {
    const int t = Td - 1;
    const int64_t bt = (int64_t)b * Td + t;
#pragma unroll
    for (int i = 0; i < kRecRegs; ++i) pre[i] = rec_load(a.stash, bt, t, tid + i * NT);
    if (tid < R80) pre_dout = a.dout[bt * R80 + tid];
    if (tid < Tt) pre_al = a.align[bt * Tt + tid];
  }
  lds_barrier();

  // launch-resident second halves of the GRU-3 / GRU-2 gate chunks (transposed weights; free LDS behind the state)
  float4* const lw2 = reinterpret_cast<float4*>(smem + kBwdSmemFixed + 2 * TtP);
  float4* const lw1 = lw2 + a.lres0 * NT;
  lres_fill(lw2, a.lres0, w.gw[2], 2 * kDec, 2 * kDec, 2 * kDec, X);
  lres_fill(lw1, a.lres1, w.gw[1], 2 * kDec, 2 * kDec, 2 * kDec, X);
  lds_barrier();

  // The pre-net backward of a step is off the critical path and runs one step late: its layer-2 input gradient comes out
  // of the next processed step's FAN round (it needs the same dx), layer 1 rides in that step's DAL round.
  // `pend` = such a deferred pre-net (of step t+1) exists.
  bool pend = false;
  float* gs_pend = nullptr;
  const float km1c = a.keep1 ? 2.f : 1.f, km2c = a.keep2 ? 2.f : 1.f;
  auto dp2_epi = [&](int n, float y) {
    const float g = S.p2prev[n] > 0.f ? km2c * y : 0.f;
    gs_pend[kGsP2 + n] = g;
    return g;
  };
  auto dp2_put = [&](int n, float v) { S.dp2[n] = v; };
  auto p2T_epi = [&](int n, float y) {
    const float g = S.p1prev[n] > 0.f ? km1c * y : 0.f;
    gs_pend[kGsP1 + n] = g;
    return g;
  };
  auto p2T_put = [&](int n, float v) { dp1[n] = v; };
  Pref pf;   // next round's first weight rows, fetched during the current round's all-gather
  const NextMv nx_p2T{w.pre_w2, kPre1, kPre2, kPre1}, nx_dp2{w.in_w, kPre2 + kAtt, kDec, kPre2};

  for (int t = Td - 1; t >= 0; --t) {
    X.epoch = (unsigned)(Td - t);
    X.trace = (TR && a.trace && blockIdx.x == 0 && t == Td / 2) ? a.trace : nullptr;
    X.tslot = 0;
    const int64_t bt = (int64_t)b * Td + t;
    float* gs = a.gstash + bt * kGsRec;
    const bool next_from_out = (t + 1 < Td) && a.sample && a.sample[(int64_t)t * B + b];
    // the pre-net backward of step t+1 rides in this step's FAN / DQ rounds only where the recurrence needs it (step t+1 was
    // fed by this step's output); with a.hoisted the teacher-forced steps' pre-net gradients are GEMMs after the launch (model.hip)
    const bool rider = pend && (next_from_out || !a.hoisted);

    // 0. land the prefetched record in LDS, start fetching the one for step t-1
#pragma unroll
    for (int i = 0; i < kRecRegs; ++i)
      if (tid + i * NT < kRecFloats) S.rec[tid + i * NT] = pre[i];
    if (tid < R80) dov[tid] = pre_dout;   // direct part: loss + post-net
    if (tid < Tt) S.als[tid] = pre_al;
    for (int s = tid + NT; s < Tt; s += NT) S.als[s] = a.align[bt * Tt + s];
    if (t > 0) {
#pragma unroll
      for (int i = 0; i < kRecRegs; ++i) pre[i] = rec_load(a.stash, bt - 1, t - 1, tid + i * NT);
      if (tid < R80) pre_dout = a.dout[(bt - 1) * R80 + tid];
      if (tid < Tt) pre_al = a.align[(bt - 1) * Tt + tid];
    }
    lds_barrier();
    // 1. round FAN: dx_{t+1} against this wave's rows of VWx -> d alignments_t (the context of step t only ever fed x_{t+1}, so
    //    d alignments = VWx . dx_{t+1}: no d context, no round of its own), and through Wi_p^T (wT.in_w columns [0,128)) ->
    //    d p2_{t+1}
    {
      tstamp(X, 0);
      tmark(X, 10);
      {
        const float4 c4 = reinterpret_cast<const float4*>(S.dx)[lane];
        auto dal = [&](int s, float4 x4) {
          float dd = x4.x * c4.x + x4.y * c4.y + x4.z * c4.z + x4.w * c4.w;
          dd = wave_sum(dd);
          if (lane == 0) {
            S.des[s] = dd;
            if (P > 1) xput(X, XB_DAL + s, dd);
          }
        };
#pragma unroll
        for (int i = 0; i < kAR; ++i) {
          const int s = s_first + i * s_stride;
          if (s < len) dal(s, vres[i]);
        }
        for (int s = s_first + kAR * s_stride; s < len; s += s_stride)
          dal(s, reinterpret_cast<const float4*>(vwx + (int64_t)s * kDec)[lane]);
      }
      tmark(X, 11);
      if (rider) {
        phase_mv(w.in_w, kPre2 + kAtt, kDec, kPre2, S.dx, S.part, X, pf);
        tstamp(X, 1);
        lds_barrier();
        phase_fin(kPre2, S.part, X, XB_DP2, dp2_epi, dp2_put);
        tstamp(X, 2);
        prefetch_w(pf, nx_p2T.W, nx_p2T.ldw, nx_p2T.K, nx_p2T.N, X);
      }
      if (P > 1) {   // d p2 slices and the other peers' d alignments rows, polled concurrently
        const Slice SP = slice_of(X, kPre2);
        const bool needA = rider && tid < kPre2 && !(tid >= SP.nbeg && tid < SP.nbeg + SP.nloc);
        const bool needB = tid < len && (tid & (P - 1)) != X.peer;
        float vA, vB;
        xget2(X, XB_DP2 + tid, needA, XB_DAL + tid, needB, vA, vB);
        if (needA) dp2_put(tid, vA);
        if (needB) S.des[tid] = vB;
        for (int s = tid + NT; s < len; s += NT)
          if ((s & (P - 1)) != X.peer) S.des[s] = xget(X, XB_DAL + s);
      }
      tstamp(X, 3);
      tmark(X, 12);
      X.tslot++;
    }
    lds_barrier();
    tmark(X, 13);
    // d p1pre of step t+1 reaches this step's cell_output only if that step was fed by it (sampled rows)
    const bool use_p1 = rider && next_from_out;
    // 3b. softmax backward: de = al * (dal - sum al*dal)   (every wave computes the dot redundantly)
    {
      float dot = 0.f;
      for (int s = lane; s < len; s += 64) dot += S.als[s] * S.des[s];
      dot = wave_sum(dot);
      lds_barrier();
      for (int s = tid; s < len; s += NT) S.des[s] = S.als[s] * (S.des[s] - dot);
    }
    lds_barrier();
    tmark(X, 14);
    // 3c. energy backward on this peer's UNITS over all rows: th = tanh(keys+q); dpre = de*v*(1-th^2);
    //     dq[u] = sum_s dpre; dkeys[s,u] += dpre; dv[u] += de*th
    {
      float dqa = 0.f;
      if (uact) {
        const float qu = S.rec[RL_Q + ub + ul];
#pragma unroll
        for (int i = 0; i < kKR; ++i) {
          const int s = sg + i * NSG;
          if (s < len) {
            const float de = S.des[s];
            const float th = tanh_fast(kr[i] + qu);
            const float pre = de * vu * (1.f - th * th);
            dqa += pre;
            dkr[i] += pre;
            dvu += de * th;
          }
        }
        for (int s = sg + kKR * NSG; s < len; s += NSG) {   // rows beyond the resident set: read-modify-write in memory
          const float de = S.des[s];
          const float th = tanh_fast(keys[(int64_t)s * kAtt + ub + ul] + qu);
          const float pre = de * vu * (1.f - th * th);
          dqa += pre;
          dkeys[(int64_t)s * ldk + ub + ul] += pre;
          dvu += de * th;
        }
        S.red[sg * un + ul] = dqa;
      }
      // same round: deferred pre-net layer 2 of step t+1: d p1 = d p2pre . W2^T (wT.pre_w2 is (128, 256))
      if (rider) phase_mv(w.pre_w2, kPre1, kPre2, kPre1, S.dp2, S.part, X, pf);
      tmark(X, 15);
    }
    lds_barrier();
    tmark(X, 16);
    {
      auto dq_put = [&](int n, float v) { dq[n] = v; };
      if (tid < un) {
        float dsum = 0.f;
        for (int g = 0; g < NSG; ++g) dsum += S.red[g * un + tid];
        const int n = ub + tid;
        dq[n] = dsum;
        gs[kGsQ + n] = dsum;
        if (P > 1) xput(X, XB_DQP + n, dsum);
      }
      if (rider) phase_fin(kPre1, S.part, X, XB_P2, p2T_epi, p2T_put);
      prefetch_w2(pf, Seg2{a.wot, S.vo, R80 + kAtt + (use_p1 ? kPre1 : 0), a.wdx, S.dx, kDec}, kDec, kDec, X);
      if (P > 1) {   // dq slices and the other peers' slices of d p1, polled concurrently
        const Slice SB = slice_of(X, kPre1);
        const bool needA = tid < kAtt && !(tid >= ub && tid < ub + un);
        const bool needB = rider && tid < kPre1 && !(tid >= SB.nbeg && tid < SB.nbeg + SB.nloc);
        float vA, vB;
        xget2(X, XB_DQP + tid, needA, XB_P2 + tid, needB, vA, vB);
        if (needA) dq_put(tid, vA);
        if (needB) p2T_put(tid, vB);
      }
    }
    tmark(X, 17);
    lds_barrier();
    tmark(X, 18);
    // record d p1pre_{t+1} (or 0) for the output projection's weight gradient
    if (lead && tid < kPre1) gs[kGsP1S + tid] = use_p1 ? dp1[tid] : 0.f;
    // 4. round OUT: dy = [d cell_output (direct) ; dq ; d p1pre_{t+1}] . [Wo^T ; (Wo Wq)^T ; (Wo_f W1)^T]   (a.wot, (80r+512, 256))
    //               + dx_{t+1} . (Wx_o^T Wo^T)   (a.wdx, (256, 256))
    {
      const Seg2 so{a.wot, S.vo, R80 + kAtt + (use_p1 ? kPre1 : 0), a.wdx, S.dx, kDec};
      auto o_put = [&](int n, float v) {
        S.dy[n] = v;
        S.dht[n] = S.dh[2 * kDec + n] + v;   // dL/dh3' = carried + residual path
      };
      tstamp(X, 0);
      mv_store(kDec, X, mv_accum2<true>(so, kDec, kDec, X, pf, make_float4(0.f, 0.f, 0.f, 0.f)), S.part);
      tstamp(X, 1);
      lds_barrier();
      phase_fin(kDec, S.part, X, XB_OUT, [&](int n, float y) { return y; }, o_put);
      tstamp(X, 2);
      prefetch_w(pf, w.cw[2], 2 * kDec, kDec, 2 * kDec, X);
      phase_gather(kDec, X, XB_OUT, o_put);
      tstamp(X, 3);
      X.tslot++;
    }
    lds_barrier();
    // 6. GRU layers, top down
    for (int l = 2; l >= 0; --l) {
      const float* Rl = S.rec + RL_R + l * kDec;
      const float* Ul = S.rec + RL_U + l * kDec;
      const float* Cl = S.rec + RL_C + l * kDec;
      const float* HPl = S.rec + RL_HP + l * kDec;
      if (tid < kDec) {
        const int n = tid;
        const float u = Ul[n], c = Cl[n], hp = HPl[n];
        const float dht = S.dht[n];
        const float du = dht * (hp - c);
        const float dc = dht * (1.f - u);
        const float dcp = dc * (1.f - c * c);
        const float dup = du * u * (1.f - u);
        S.dcp[n] = dcp;
        S.dgp[kDec + n] = dup;
        if (lead) {
          gs[kGsC + l * kDec + n] = dcp;
          gs[kGsG + l * 512 + kDec + n] = dup;
        }
      }
      lds_barrier();
      // [d inp ; d(r*h)] = dcp . Wc^T     (wT.cw[l] is (256, 512))
      phase(w.cw[l], 2 * kDec, kDec, 2 * kDec, S.dcp, S.part, X, XB_C + l * 1024,
            [&](int n, float y) {
              if (n >= kDec) {
                const int i = n - kDec;
                const float rr = Rl[i];
                gs[kGsG + l * 512 + i] = y * HPl[i] * rr * (1.f - rr);
              }
              return y;
            },
            [&](int n, float y) {
              if (n < kDec) {
                S.dinp[n] = y;
              } else {
                const int i = n - kDec;
                const float rr = Rl[i];
                S.dgp[i] = y * HPl[i] * rr * (1.f - rr);
                S.dh[l * kDec + i] = S.dht[i] * Ul[i] + y * rr;   // partial new carry: dht*u + d(rh)*r
              }
            },
            pf, NextMv{w.gw[l], 2 * kDec, 2 * kDec, 2 * kDec});
      lds_barrier();
      // [d inp ; d h] += dgp . Wg^T       (wT.gw[l] is (512, 512))
      phase(w.gw[l], 2 * kDec, 2 * kDec, 2 * kDec, S.dgp, S.part, X, XB_G + l * 1024, [&](int n, float y) { return y; },
            [&](int n, float y) {
              if (n < kDec) {
                const float di = S.dinp[n] + y;
                if (l > 0) S.dht[n] = S.dh[(l - 1) * kDec + n] + di;   // next layer down: carried + input path
                else S.dx[n] = S.dy[n] + di;                            // x feeds GRU1 and the residual
              } else {
                S.dh[l * kDec + n - kDec] += y;
              }
            },
            pf, l > 0 ? NextMv{w.cw[l - 1], 2 * kDec, kDec, 2 * kDec} : nx_dp2, l == 2 ? lw2 : (l == 1 ? lw1 : nullptr),
            l == 2 ? a.lres0 : (l == 1 ? a.lres1 : 0));
      lds_barrier();
    }
    if (lead && tid < kDec) gs[kGsX + tid] = S.dx[tid];
    // the pre-net backward of THIS step is deferred into the next processed step's FAN / DAL rounds
    if (tid < kPre1) S.p1prev[tid] = S.rec[RL_P1 + tid];
    if (tid < kPre2) S.p2prev[tid] = S.rec[RL_P2 + tid];
    pend = true;
    gs_pend = gs;
    lds_barrier();
  }
  // deferred pre-net of step 0 (its layer-1 input gradient is not needed: nothing precedes step 0)
  if (pend && !a.hoisted) {
    X.epoch = (unsigned)(Td + 1);
    phase(w.in_w, kPre2 + kAtt, kDec, kPre2, S.dx, S.part, X, XB_DP2, dp2_epi, dp2_put, pf, nx_p2T);
    lds_barrier();
    phase(w.pre_w2, kPre1, kPre2, kPre1, S.dp2, S.part, X, XB_P2, p2T_epi, p2T_put, pf, NextMv());
    lds_barrier();
  }
  // resident dkeys accumulators -> memory (each (row, unit) is owned by exactly one thread of one peer)
  if (uact) {
#pragma unroll
    for (int i = 0; i < kKR; ++i) {
      const int s = sg + i * NSG;
      if (s < len) dkeys[(int64_t)s * ldk + ub + ul] = dkr[i];
    }
    S.red[sg * un + ul] = dvu;
  }
  // attention_v gradient: reduce the row groups, one atomic per owned unit
  lds_barrier();
  if (tid < un) {
    float dsum = 0.f;
    for (int g = 0; g < NSG; ++g) dsum += S.red[g * un + tid];
    a.datt_v[(int64_t)b * kAtt + ub + tid] = dsum;   // per-row partial (each (row, unit) has one owner); rows are summed in order afterwards
  }
}

// Largest cluster width in {32,...,2,1} whose B*P workgroups are all co-resident (the all-gather needs every peer running).
// The mat-vecs are bound by the per-CU vector-memory path (64 B/clk), so small batches spread a row over more CUs.
int g_last_cluster[2] = {0, 0};   // cluster width of the most recent forward / backward launch (taco_debug_last_cluster)
template <class K>
int pick_cluster(K kernel, size_t smem, int B, int want) {
  int dev = 0, cus = 0, per_cu = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, NT, smem);
  if (e != hipSuccess || per_cu < 1) {
    // a silent fall-back to one workgroup per row is 3x slower: say so
    fprintf(stderr, "taco: decoder occupancy query failed (%s, %d blocks/CU, %zu B LDS): running ONE workgroup per row\n",
            hipGetErrorString(e), per_cu, smem);
    return 1;
  }
  const int64_t cap = (int64_t)cus * per_cu;
  for (int p = 32; p >= 2; p >>= 1)
    if (p <= want && (int64_t)B * p <= cap) return p;
  return 1;
}

// Training uses at most 8 peers so that results are bit-identical for every per-GPU batch <= 32 (the summation order
// depends on the cluster width; data-parallel shards of a batch must reproduce the unsharded gradients exactly).  Inference
// has no such contract and takes 16 peers when they fit (B <= 16): -7 % per decoder step.  TACO_DEC_CLUSTER overrides.
// The LDS the state leaves free holds launch-resident weight rows: 8 or 4 rows x 512 threads x 16 bytes for each of two gate
// mat-vecs (TACO_DEC_NO_LRES=1 disables it, for A/B runs).
void lres_plan(size_t& smem, int& r0, int& r1) {
  r0 = r1 = 0;
  if (getenv("TACO_DEC_NO_LRES")) return;
  for (int l = 0; l < 2; ++l)
    for (int rows = 8; rows >= 4; rows -= 4)
      if (smem + (size_t)rows * NT * 16 <= (size_t)158 * 1024) {
        (l == 0 ? r0 : r1) = rows;
        smem += (size_t)rows * NT * 16;
        break;
      }
}
int probe_bits() {
  if (!kProbes) return 0;
  return (getenv("TACO_DEC_FAKEW") ? 1 : 0) | (getenv("TACO_DEC_FAKEX") ? 2 : 0) | (getenv("TACO_DEC_NOPF") ? 8 : 0) |
         (getenv("TACO_DEC_NOLIVE") ? 16 : 0);
}
int env_cluster(int dflt) {
  const char* e = getenv("TACO_DEC_CLUSTER");
  if (!e) return dflt;
  const int v = atoi(e);
  return (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32) ? v : dflt;
}

}  // namespace

int decoder_last_cluster(int which) { return g_last_cluster[which & 1]; }
void decoder_note_cluster(int which, int P) { g_last_cluster[which & 1] = P; }

int64_t decoder_xchg_bytes(int B, int Tt) {
  const int TtP = (Tt + 3) & ~3;
  return (int64_t)B * (kXchgFixed + TtP) * 8;
}

int launch_decoder_fwd(DecFwdArgs a, hipStream_t s) {
  TACO_REQUIRE(a.B > 0 && a.Tt > 0 && a.Td > 0 && a.r >= 1 && a.r <= 5, "decoder_fwd: bad dims B=%d Tt=%d Td=%d r=%d", a.B,
               a.Tt, a.Td, a.r);
  TACO_REQUIRE(a.xchg && a.err, "decoder_fwd: exchange area missing");
  const int TtP = (a.Tt + 3) & ~3;
  size_t smem = (size_t)(kFwdSmemFixed + 2 * TtP) * sizeof(float);
  TACO_REQUIRE(smem <= 160 * 1024, "decoder_fwd: Tt=%d needs %zu bytes of LDS (> 160 KiB)", a.Tt, smem);
  lres_plan(smem, a.lres0, a.lres1);
  void (*kern)(DecFwdArgs) = a.trace ? decoder_fwd_kernel<true> : decoder_fwd_kernel<false>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) {
      taco_set_error("decoder_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return TACO_ELAUNCH;
    }
  }
  a.P = pick_cluster(kern, smem, a.B, env_cluster(a.mel ? 8 : 16));
  g_last_cluster[0] = a.P;
  a.fakew = probe_bits();
  if (a.P > 1) {
    hipError_t e = hipMemsetAsync(a.xchg, 0, (size_t)decoder_xchg_bytes(a.B, a.Tt), s);
    taco_tail_touch(s);
    if (e != hipSuccess) {
      taco_set_error("decoder_fwd: memset: %s", hipGetErrorString(e));
      return TACO_ELAUNCH;
    }
  }
  TACO_KLAUNCH(kern, dim3(a.B * a.P), dim3(NT), smem, s, a);
  TACO_LAUNCH_CHECK("decoder_fwd");
  return TACO_OK;
}

int launch_decoder_bwd(DecBwdArgs a, hipStream_t s) {
  TACO_REQUIRE(a.B > 0 && a.Tt > 0 && a.Td > 0 && a.r >= 1 && a.r <= 5, "decoder_bwd: bad dims");
  TACO_REQUIRE(a.xchg && a.err, "decoder_bwd: exchange area missing");
  const int TtP = (a.Tt + 3) & ~3;
  size_t smem = (size_t)(kBwdSmemFixed + 2 * TtP) * sizeof(float);
  TACO_REQUIRE(smem <= 160 * 1024, "decoder_bwd: Tt=%d needs %zu bytes of LDS (> 160 KiB)", a.Tt, smem);
  lres_plan(smem, a.lres0, a.lres1);
  void (*kern)(DecBwdArgs) = a.trace ? decoder_bwd_kernel<true> : decoder_bwd_kernel<false>;
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) {
      taco_set_error("decoder_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return TACO_ELAUNCH;
    }
  }
  a.P = pick_cluster(kern, smem, a.B, env_cluster(8));
  g_last_cluster[1] = a.P;
  a.fakew = probe_bits();
  if (a.P > 1 && !a.xchg_zeroed) {
    hipError_t e = hipMemsetAsync(a.xchg, 0, (size_t)decoder_xchg_bytes(a.B, a.Tt), s);
    taco_tail_touch(s);
    if (e != hipSuccess) {
      taco_set_error("decoder_bwd: memset: %s", hipGetErrorString(e));
      return TACO_ELAUNCH;
    }
  }
  TACO_KLAUNCH(kern, dim3(a.B * a.P), dim3(NT), smem, s, a);
  TACO_LAUNCH_CHECK("decoder_bwd");
  return TACO_OK;
}
