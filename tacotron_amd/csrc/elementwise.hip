// elementwise.hip -- HBM-bound glue kernels of the Tacotron hot path: embedding gather/scatter (tacotron.py:111-114),
// BN-affine + max-pool (ops.py:64-71), highway blend (ops.py:46), activation/dropout derivatives, column reductions
// for bias / BN gradients, L1 loss + sign gradient (tacotron.py:158-160), weight transposes for the backward GEMMs,
// global-norm clip + TF-form Adam (tacotron.py:167-185) and the Bernoulli mask generator.
// All are float4-vectorised where the layout allows, grid-stride, one pass over their tensors.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int kThreads = 256;
inline int grid_for(int64_t n_items, int per_block = kThreads, int cap = 4096) {
  int64_t g = (n_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

__device__ __forceinline__ float block_sum(float v, float* red /* >= 4 floats LDS */) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

__global__ void embedding_kernel(const float* __restrict__ table, const int32_t* __restrict__ ids,
                                 float* __restrict__ out, int64_t rows, int V, int width) {
  const int w4 = width / 4;
  const int64_t total = rows * w4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / w4;
    const int c4 = (int)(i % w4);
    int id = ids[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);
    reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(table + (int64_t)id * width)[c4];
  }
}

// dtable[v] += sum of the dout rows whose id is v.  Workgroup (v, segment) handles the rows of its segment: its threads first
// list the matching rows (each thread scans a contiguous run, a block-wide prefix sum keeps the list sorted), then four row
// lanes x 64 column quads add the listed rows (lane l takes list entries l, l+4, ...; the lanes are summed in a fixed order).
// One segment (deterministic mode): plain read-modify-write, reproducible.  Several: one atomicAdd per element and segment.
// (Padded text makes id 0 by far the most frequent row -- a single workgroup walking all of its rows was 0.5 ms.)
constexpr int kEmbSeg = 4096;   // rows listed per pass (LDS: 16 KB)
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const float* __restrict__ dout, const int32_t* __restrict__ ids,
                                                            float* __restrict__ dtable, int64_t rows, int V, int width,
                                                            int64_t rows_per_seg) {
  __shared__ int list[kEmbSeg];
  __shared__ int cnt[256];
  __shared__ float4 red[3][64];
  const int v = blockIdx.x, tid = threadIdx.x, q0 = tid & 63, rl = tid >> 6;
  const int W4 = width >> 2;
  const int64_t seg0 = (int64_t)blockIdx.y * rows_per_seg;
  const int64_t seg1 = min(rows, seg0 + rows_per_seg);
  float4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = make_float4(0.f, 0.f, 0.f, 0.f);   // column quads q0, q0+64, ... (width <= 1024)
  for (int64_t base = seg0; base < seg1; base += kEmbSeg) {
    const int seg = (int)min((int64_t)kEmbSeg, seg1 - base);
    const int per = (seg + 255) >> 8;
    const int r0 = min(seg, tid * per), r1 = min(seg, r0 + per);
    int c = 0;
    for (int r = r0; r < r1; ++r) {
      int id = ids[base + r];
      id = id < 0 ? 0 : (id >= V ? V - 1 : id);
      c += id == v;
    }
    cnt[tid] = c;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {   // inclusive scan
      const int x = tid >= off ? cnt[tid - off] : 0;
      __syncthreads();
      cnt[tid] += x;
      __syncthreads();
    }
    int o = cnt[tid] - c;
    const int total = cnt[255];
    for (int r = r0; r < r1; ++r) {
      int id = ids[base + r];
      id = id < 0 ? 0 : (id >= V ? V - 1 : id);
      if (id == v) list[o++] = r;
    }
    __syncthreads();
#pragma unroll 4
    for (int j = rl; j < total; j += 4) {
      const float4* src = reinterpret_cast<const float4*>(dout + (base + list[j]) * width);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (q0 + i * 64 < W4) {
          const float4 d = src[q0 + i * 64];
          acc[i].x += d.x; acc[i].y += d.y; acc[i].z += d.z; acc[i].w += d.w;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i * 64 >= W4) break;
    if (rl > 0) red[rl - 1][q0] = acc[i];
    __syncthreads();
    if (rl == 0 && q0 + i * 64 < W4) {
      float4 t = acc[i];
      for (int l = 0; l < 3; ++l) { t.x += red[l][q0].x; t.y += red[l][q0].y; t.z += red[l][q0].z; t.w += red[l][q0].w; }
      float* dst = dtable + (int64_t)v * width + (q0 + i * 64) * 4;
      if (gridDim.y == 1) {
        dst[0] += t.x; dst[1] += t.y; dst[2] += t.z; dst[3] += t.w;
      } else {
        atomicAdd(dst + 0, t.x); atomicAdd(dst + 1, t.y); atomicAdd(dst + 2, t.z); atomicAdd(dst + 3, t.w);
      }
    }
    __syncthreads();
  }
}

__global__ void bn_maxpool_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float* __restrict__ y, int B, int T, int C) {
  const int C4 = C / 4;
  const int64_t total = (int64_t)B * T * C4;
  const float rs = 1.0f / sqrtf(1.0f + kBnEps);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c4 = (int)(i % C4);
    const int64_t row = i / C4;
    const int t = (int)(row % T);
    const float4 g = reinterpret_cast<const float4*>(gamma)[c4];
    const float4 b = reinterpret_cast<const float4*>(beta)[c4];
    const float4 a = reinterpret_cast<const float4*>(x)[i];
    float4 z = make_float4(a.x * (g.x * rs) + b.x, a.y * (g.y * rs) + b.y, a.z * (g.z * rs) + b.z, a.w * (g.w * rs) + b.w);
    if (t + 1 < T) {
      const float4 a2 = reinterpret_cast<const float4*>(x)[i + C4];
      z.x = fmaxf(z.x, a2.x * (g.x * rs) + b.x);
      z.y = fmaxf(z.y, a2.y * (g.y * rs) + b.y);
      z.z = fmaxf(z.z, a2.z * (g.z * rs) + b.z);
      z.w = fmaxf(z.w, a2.w * (g.w * rs) + b.w);
    }
    reinterpret_cast<float4*>(y)[i] = z;
  }
}

// blockDim = (64 channel quads, 4 row lanes); grid = (C/256 ceil, row chunks).  A row lane walks a CONTIGUOUS run of rows, so
// the three pooled neighbours z[row-1], z[row], z[row+1] and dy[row-1] roll through registers: x and dy are read once.
// x is the ReLU output the affine was applied to (ops.py:64-66: relu -> batch_norm -> max_pool): the ReLU backward (x > 0)
// is applied to the stored gradient in the same pass.
__global__ void bn_maxpool_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                      const float* __restrict__ beta, const float* __restrict__ dy,
                                      float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                      int B, int T, int C, int rows_per_block) {
  const int C4 = C >> 2;
  const int c4 = blockIdx.x * 64 + threadIdx.x;
  const int64_t M = (int64_t)B * T;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  const int seg = (rows_per_block + 3) >> 2;
  const int64_t ra = r0 + (int64_t)threadIdx.y * seg;
  const int64_t rb = ra + seg < r1 ? ra + seg : r1;
  const float rs = 1.0f / sqrtf(1.0f + kBnEps);
  float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
  if (c4 < C4 && ra < rb) {
    const float4 g4 = reinterpret_cast<const float4*>(gamma)[c4];
    const float4 be4 = reinterpret_cast<const float4*>(beta)[c4];
    const float sc[4] = {g4.x * rs, g4.y * rs, g4.z * rs, g4.w * rs};
    const float be[4] = {be4.x, be4.y, be4.z, be4.w};
    const float4* X = reinterpret_cast<const float4*>(x) + c4;
    const float4* DY = reinterpret_cast<const float4*>(dy) + c4;
    float4* DX = reinterpret_cast<float4*>(dx) + c4;
    int t = (int)(ra % T);
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 xp = t > 0 ? X[(ra - 1) * C4] : zero4, dyp = t > 0 ? DY[(ra - 1) * C4] : zero4;
    float4 xc = X[ra * C4];
    for (int64_t row = ra; row < rb; ++row) {
      const bool last = t == T - 1, first = t == 0;
      const float4 xn = last ? zero4 : X[(row + 1) * C4];
      const float4 dyc = DY[row * C4];
      const float xpv[4] = {xp.x, xp.y, xp.z, xp.w}, xcv[4] = {xc.x, xc.y, xc.z, xc.w}, xnv[4] = {xn.x, xn.y, xn.z, xn.w};
      const float dpv[4] = {dyp.x, dyp.y, dyp.z, dyp.w}, dcv[4] = {dyc.x, dyc.y, dyc.z, dyc.w};
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float z = xcv[q] * sc[q] + be[q];
        float dz = 0.f;
        if (last || z >= xnv[q] * sc[q] + be[q]) dz = dcv[q];          // y[row] = max(z[row], z[row+1]) takes z[row] on ties
        if (!first && z > xpv[q] * sc[q] + be[q]) dz += dpv[q];        // y[row-1] takes z[row] only when strictly larger
        o[q] = xcv[q] > 0.f ? dz * sc[q] : 0.f;
        ag[q] += dz * xcv[q] * rs;
        ab[q] += dz;
      }
      DX[row * C4] = make_float4(o[0], o[1], o[2], o[3]);
      xp = xc; xc = xn; dyp = dyc;
      t = last ? 0 : t + 1;
      if (last && row + 1 < rb) xc = X[(row + 1) * C4];   // next sequence starts: its first row was not fetched as a neighbour
    }
  }
  __shared__ float red[2][4][256];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    red[0][threadIdx.y][threadIdx.x * 4 + q] = ag[q];
    red[1][threadIdx.y][threadIdx.x * 4 + q] = ab[q];
  }
  __syncthreads();
  const int tid = threadIdx.y * 64 + threadIdx.x;   // 256 threads <-> 256 channels of the block
  const int c = blockIdx.x * 256 + tid;
  if (c < C) {
    float g = 0.f, b2 = 0.f;
    for (int i = 0; i < 4; ++i) {
      g += red[0][i][tid];
      b2 += red[1][i][tid];
    }
    atomicAdd(&dgamma[c], g);
    atomicAdd(&dbeta[c], b2);
  }
}

__global__ void highway_combine_kernel(const float* __restrict__ th, const float* __restrict__ x,
                                       float* __restrict__ y, int64_t M) {
  const int64_t total = M * (kCb / 4);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / (kCb / 4);
    const int c4 = (int)(i % (kCb / 4));
    const float4 t = reinterpret_cast<const float4*>(th + row * 2 * kCb)[c4];
    const float4 h = reinterpret_cast<const float4*>(th + row * 2 * kCb + kCb)[c4];
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    float4 o;
    o.x = h.x * t.x + xv.x * (1.f - t.x);
    o.y = h.y * t.y + xv.y * (1.f - t.y);
    o.z = h.z * t.z + xv.z * (1.f - t.z);
    o.w = h.w * t.w + xv.w * (1.f - t.w);
    reinterpret_cast<float4*>(y)[i] = o;
  }
}

__global__ void highway_combine_bwd_kernel(const float* __restrict__ th, const float* __restrict__ x,
                                           const float* __restrict__ dy, float* __restrict__ dth,
                                           float* __restrict__ dx, int64_t M) {
  const int64_t total = M * kCb;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / kCb;
    const int c = (int)(i % kCb);
    const float t = th[row * 2 * kCb + c];
    const float h = th[row * 2 * kCb + kCb + c];
    const float g = dy[i];
    dth[row * 2 * kCb + c] = g * (h - x[i]) * t * (1.f - t);
    dth[row * 2 * kCb + kCb + c] = h > 0.f ? g * t : 0.f;
    dx[i] = g * (1.f - t);
  }
}

__global__ void act_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                               const uint8_t* __restrict__ keep, float* __restrict__ dz, int64_t n, int act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float yv = y[i];
    float g = dy[i];
    if (act == TACO_ACT_RELU) g = yv > 0.f ? g : 0.f;
    else if (act == TACO_ACT_SIGMOID) g = g * yv * (1.f - yv);
    else if (act == TACO_ACT_TANH) g = g * (1.f - yv * yv);
    if (keep) g = keep[i] ? g * 2.0f : 0.f;
    dz[i] = g;
  }
}

// blockDim = (64 cols, 4 row lanes); grid = (N/64 ceil, row chunks)
__global__ void affine_act_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ gamma,
                                      const float* __restrict__ dy, float* __restrict__ dz, float* __restrict__ dgamma,
                                      float* __restrict__ dbeta, int64_t M, int N, int act, int rows_per_block) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
  const float rs = 1.0f / sqrtf(1.0f + kBnEps);
  float ag = 0.f, ab = 0.f;
  if (c < N) {
    const float s = gamma[c] * rs;
#pragma unroll 4
    for (int64_t row = r0 + threadIdx.y; row < r1; row += 4) {
      const float g = dy[row * N + c];
      const float p = pre[row * N + c];
      ag += g * p * rs;
      ab += g;
      float d = g * s;
      if (act == TACO_ACT_RELU && !(p > 0.f)) d = 0.f;
      dz[row * N + c] = d;
    }
  }
  __shared__ float red[2][4][64];
  red[0][threadIdx.y][threadIdx.x] = ag;
  red[1][threadIdx.y][threadIdx.x] = ab;
  __syncthreads();
  if (threadIdx.y == 0 && c < N) {
    float g = 0.f, b = 0.f;
    for (int i = 0; i < 4; ++i) {
      g += red[0][i][threadIdx.x];
      b += red[1][i][threadIdx.x];
    }
    atomicAdd(&dgamma[c], g);
    atomicAdd(&dbeta[c], b);
  }
}


// grid = (N/64 ceil, B); block = (64, 4)
__global__ void colsum_batched_kernel(const float* __restrict__ x, int ld, float* __restrict__ out, int T, int N) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int b = blockIdx.y;
  float a = 0.f;
  if (c < N)
    for (int t = threadIdx.y; t < T; t += 4) a += x[((int64_t)b * T + t) * ld + c];
  __shared__ float red[4][64];
  red[threadIdx.y][threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.y == 0 && c < N)
    out[(int64_t)b * N + c] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void mask_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ len, float* __restrict__ y,
                                 int B, int T, int C) {
  const int C4 = C / 4;
  const int64_t total = (int64_t)B * T * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / C4;
    const int b = (int)(row / T), t = (int)(row % T);
    float4 v = reinterpret_cast<const float4*>(x)[i];
    if (t >= len[b]) v = make_float4(0.f, 0.f, 0.f, 0.f);
    reinterpret_cast<float4*>(y)[i] = v;
  }
}

// out[b,s,:] = x[b,s,:] - mean over s' < len[b] of x[b,s',:]   (rows s >= len[b]: 0).  One thread per column walks the rows in
// order (reproducible).  Used for the attention memory of the decoder backward: d alignments only matter up to a per-step
// constant (the softmax backward removes the alignment-weighted mean), and removing the rows' common component BEFORE the
// fold with Wx_c keeps the per-row rounding errors relative to what the softmax backward actually sees.
__global__ __launch_bounds__(256) void center_rows_kernel(const float* __restrict__ x, const int32_t* __restrict__ len,
                                                          float* __restrict__ out, int T, int C) {
  // block = 4 row lanes x 64 column quads (256 columns); grid = (C/256 ceil, B, 4 row quarters).  Every block forms the column
  // means of its sequence itself (lane l sums rows l, l+4, ...; the four lanes are added in a fixed order: reproducible) and
  // writes one quarter of the rows.
  __shared__ float4 red[3][64];
  const int b = blockIdx.y;
  const int q = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c4 = blockIdx.x * 64 + q, C4 = C >> 2;
  const bool act = c4 < C4;
  int n = len[b];
  n = n < 1 ? 1 : (n > T ? T : n);
  const float4* xb = reinterpret_cast<const float4*>(x + (int64_t)b * T * C) + (act ? c4 : 0);
  float4* ob = reinterpret_cast<float4*>(out + (int64_t)b * T * C) + (act ? c4 : 0);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
  for (int s = rl; s < n; s += 4) {
    const float4 v = xb[(int64_t)s * C4];
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (rl > 0) red[rl - 1][q] = acc;
  __syncthreads();
  if (rl == 0) {
    for (int l = 0; l < 3; ++l) { acc.x += red[l][q].x; acc.y += red[l][q].y; acc.z += red[l][q].z; acc.w += red[l][q].w; }
    const float inv = 1.0f / (float)n;
    red[0][q] = make_float4(acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv);
  }
  __syncthreads();
  const float4 mean = red[0][q];
  if (!act) return;
  const int per = (T + gridDim.z - 1) / gridDim.z;
  const int s0 = blockIdx.z * per, s1 = min(T, s0 + per);
#pragma unroll 4
  for (int s = s0 + rl; s < s1; s += 4) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < n) {
      v = xb[(int64_t)s * C4];
      v.x -= mean.x; v.y -= mean.y; v.z -= mean.z; v.w -= mean.w;
    }
    ob[(int64_t)s * C4] = v;
  }
}

// g[r][c] = y[r][c] > 0 ? scale * g[r][c] : 0  (ReLU + dropout backward against the stored, already masked activation; both
// tensors are column blocks of wider records: row pitches ldg / ldy)
__global__ void mask_pos_kernel(float* __restrict__ g, int ldg, const float* __restrict__ y, int ldy, int64_t rows, int cols,
                                float scale) {
  const int64_t total = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    const float v = g[r * ldg + c];
    g[r * ldg + c] = y[r * ldy + c] > 0.f ? scale * v : 0.f;
  }
}

__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    y[i] = a[i] + b[i];
}

__global__ __launch_bounds__(1024) void l1_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ grad, int ldg,
                          float* __restrict__ loss_slot, int64_t M, int N) {
  // one wave per row at a time: the row/column split of the padded gradient costs no division, and a wave's 64 lanes walk a
  // row in coalesced 256-byte pieces (N = 1025 rows are not 16-byte aligned, so the accesses stay scalar)
  __shared__ float red[16];
  float acc = 0.f;
  const int lane = threadIdx.x & 63;
  const int64_t wave0 = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  for (int64_t m = wave0; m < M; m += nwaves) {
    const float* ar = a + m * N;
    const float* br = b + m * N;
    float* gr = grad ? grad + m * ldg : nullptr;
    // four 64-wide pieces at a time: all eight loads (clamped addresses instead of a branch around them), then the arithmetic,
    // then the stores -- a load issued behind a store would have to wait for that store's acknowledgement before its use
    for (int n0 = 0; n0 < ldg; n0 += 256) {
      float av[4], bv[4], g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = n0 + 64 * u + lane;
        const int nn = n < N ? n : N - 1;
        av[u] = ar[nn];
        bv[u] = br[nn];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int n = n0 + 64 * u + lane;
        const float d = av[u] - bv[u];
        const bool in = n < N;
        acc += in ? fabsf(d) : 0.f;
        g[u] = in ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 0.f;
      }
      if (gr) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int n = n0 + 64 * u + lane;
          if (n < ldg) gr[n] = g[u];
        }
      }
    }
  }
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0) loss_slot[blockIdx.x] = t;   // one partial per block; finish_loss adds them in block order
}

// loss[1], loss[2] = ordered sums of the two terms' block partials (kLossParts each, zero-padded by the l1 launches);
// loss[0] = their sum (tacotron.py:160).  `out` (nullable) receives a copy of the three values.  One block, no atomics:
// the loss is reproducible bit for bit.
__global__ void finish_loss_kernel(float* __restrict__ loss, const float* __restrict__ parts, float* __restrict__ out) {
  __shared__ float red[8];
  float a = 0.f, b = 0.f;
  for (int i = threadIdx.x; i < kLossParts; i += blockDim.x) {
    a += parts[i];
    b += parts[kLossParts + i];
  }
  a = block_sum(a, red);
  b = block_sum(b, red);
  if (threadIdx.x == 0) {
    loss[1] = a; loss[2] = b; loss[0] = a + b;
    if (out) { out[0] = a + b; out[1] = a; out[2] = b; }
  }
}

__global__ void transpose_batch_kernel(TransposeBatch b) {
  __shared__ float tile[32][33];
  const int t = blockIdx.x;
  int ji = 0;
  for (int i = 1; i < b.n; ++i)
    if (t >= b.j[i].tile0) ji = i;
  const TransposeJob& J = b.j[ji];
  int rel = t - J.tile0;
  const int tn = (J.N + 31) / 32, tk = (J.K + 31) / 32;
  const int tp = rel / (tn * tk);
  rel -= tp * tn * tk;
  const int by = rel / tn, bx = rel - by * tn;
  const float* src = J.in + (int64_t)(J.taps - 1 - tp) * J.K * J.N;
  float* dst = J.out + (int64_t)tp * J.K * J.N;
  const int n0 = bx * 32, k0 = by * 32;
  const int ldi = J.ldi ? J.ldi : J.N, ldo = J.ldo ? J.ldo : J.K;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int k = k0 + j, n = n0 + threadIdx.x;
    tile[j][threadIdx.x] = (k < J.K && n < J.N) ? src[(int64_t)k * ldi + n] : 0.f;
  }
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int n = n0 + j, k = k0 + threadIdx.x;
    if (n < J.N && k < J.K) dst[(int64_t)n * ldo + k] = tile[threadIdx.x][j];
  }
}

__global__ void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
  __shared__ float red[8];
  float acc = 0.f;
  const int64_t n4 = n / 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  float a1 = 0.f, a2 = 0.f, a3 = 0.f;
  for (; i + 3 * stride < n4; i += 4 * stride) {   // four independent 16-byte loads in flight per thread
    const float4 v0 = reinterpret_cast<const float4*>(x)[i];
    const float4 v1 = reinterpret_cast<const float4*>(x)[i + stride];
    const float4 v2 = reinterpret_cast<const float4*>(x)[i + 2 * stride];
    const float4 v3 = reinterpret_cast<const float4*>(x)[i + 3 * stride];
    acc += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
    a1 += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
    a2 += v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w;
    a3 += v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w;
  }
  for (; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  acc += (a1 + a2) + a3;
  if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) {
    const float v = x[n4 * 4 + threadIdx.x];
    acc += v * v;
  }
  const float t = block_sum(acc, red);
  if (threadIdx.x == 0) out[blockIdx.x] = t;   // one partial per block (kSumsqParts blocks): no atomics, no pre-zeroing
}

// Every block first adds the kSumsqParts partial sums of squares in the same fixed order (=> the same global norm in every
// block and on every run), then updates its slice.  `err` (nullable): the decoder kernels' error words; when either is set
// the gradients are garbage (a cluster exchange timed out), so the update is SKIPPED and gnorm_out reads -1.
__global__ void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, int64_t n, float lr_t, float cap, const float* __restrict__ sumsq,
                                 float* __restrict__ gnorm_out, const int32_t* __restrict__ err) {
  __shared__ float red[8];
  if (err && (err[0] | err[1])) {
    if (gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) gnorm_out[0] = -1.f;
    return;
  }
  float part = 0.f;
  for (int i = threadIdx.x; i < kSumsqParts; i += blockDim.x) part += sumsq[i];
  const float gn = sqrtf(block_sum(part, red));
  const float scale = cap > 0.f ? cap / fmaxf(gn, cap) : 1.0f;
  if (gnorm_out && blockIdx.x == 0 && threadIdx.x == 0) gnorm_out[0] = gn;
  const float b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float gi = g[i] * scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// Inverse r-frame layout + de-normalisation (audio.reshape_frames(forward=False), audio.py:29-35; test.py:64
// `out * stft_std + stft_mean`).  out (B, Td, r*C): row 4c + j, column i*C + ch holds frame 4rc + 4i + j, feature ch.
// spec (B, F, C) (nullable) = chronological log-magnitude frames, F = (Td / 4) * 4 * r (only whole chunks, like the
// reference); mag_t (B, C, F) (nullable) = exp(spec) transposed = the (1 + n_fft/2, frames) magnitude matrix
// audio.invert_spectrogram hands to Griffin-Lim.  32 x 32 tiles through LDS so both images are written coalesced.
__global__ void denorm_unframe_kernel(const float* __restrict__ out, const float* __restrict__ mean,
                                      const float* __restrict__ stdv, float* __restrict__ spec, float* __restrict__ mag_t,
                                      int Td, int r, int C, int F) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, c0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  const int RC = r * C;
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int f = f0 + j, ch = c0 + threadIdx.x;
    float v = 0.f;
    if (f < F && ch < C) {
      const int chunk = f / (4 * r), rem = f - chunk * 4 * r;
      const int i = rem >> 2, jj = rem & 3;
      const int col = i * C + ch;
      v = out[((int64_t)b * Td + chunk * 4 + jj) * RC + col] * stdv[col] + mean[col];
      if (spec) spec[((int64_t)b * F + f) * C + ch] = v;
    }
    tile[j][threadIdx.x] = v;
  }
  if (!mag_t) return;
  __syncthreads();
  for (int j = threadIdx.y; j < 32; j += 8) {
    const int ch = c0 + j, f = f0 + threadIdx.x;
    if (ch < C && f < F) mag_t[((int64_t)b * C + ch) * F + f] = expf(tile[threadIdx.x][j]);
  }
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

__global__ void bernoulli_kernel(uint8_t* __restrict__ out, int64_t n, uint32_t thresh, uint64_t seed) {
  // each thread produces 8 bytes from one 64-bit hash: 8-bit resolution per draw would be too coarse, so draw
  // one hash per byte pair (2 x 32-bit lanes).
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i * 2 < n; i += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t h = splitmix64(seed * 0xD1342543DE82EF95ull + (uint64_t)i);
    const uint32_t lo = (uint32_t)h, hi = (uint32_t)(h >> 32);
    out[i * 2] = lo < thresh ? 1 : 0;
    if (i * 2 + 1 < n) out[i * 2 + 1] = hi < thresh ? 1 : 0;
  }
}

}  // namespace

#define EW_LAUNCH(kernel, n_items, stream, ...)                                                        \
  do {                                                                                                 \
    TACO_KLAUNCH(kernel, dim3(grid_for(n_items)), dim3(kThreads), 0, stream, __VA_ARGS__);      \
    TACO_LAUNCH_CHECK(#kernel);                                                                        \
  } while (0)

int launch_embedding(const float* table, const int32_t* ids, float* out, int64_t rows, int V, hipStream_t s, int width) {
  TACO_REQUIRE(width % 4 == 0, "embedding: width %% 4 != 0");
  EW_LAUNCH(embedding_kernel, rows * (width / 4), s, table, ids, out, rows, V, width);
  return TACO_OK;
}
int launch_embedding_bwd(const float* dout, const int32_t* ids, float* dtable, int64_t rows, int V, hipStream_t s, int width) {
  TACO_REQUIRE(width <= 1024 && width % 4 == 0, "embedding_bwd: width %d must be a multiple of 4, <= 1024", width);
  int segs = taco_deterministic() ? 1 : (int)std::min<int64_t>(32, (rows + 255) / 256);
  if (segs < 1) segs = 1;
  const int64_t rps = (rows + segs - 1) / segs;
  TACO_KLAUNCH(embedding_bwd_kernel, dim3(V, segs), dim3(256), 0, s, dout, ids, dtable, rows, V, width, rps);
  TACO_LAUNCH_CHECK("embedding_bwd");
  return TACO_OK;
}
int launch_center_rows(const float* x, const int32_t* len, float* out, int B, int T, int C, hipStream_t s) {
  TACO_REQUIRE(C % 4 == 0, "center_rows: C %% 4 != 0");
  TACO_KLAUNCH(center_rows_kernel, dim3((C + 255) / 256, B, 4), dim3(256), 0, s, x, len, out, T, C);
  TACO_LAUNCH_CHECK("center_rows");
  return TACO_OK;
}
int launch_mask_pos(float* g, int ldg, const float* y, int ldy, int64_t rows, int cols, float scale, hipStream_t s) {
  EW_LAUNCH(mask_pos_kernel, rows * cols, s, g, ldg, y, ldy, rows, cols, scale);
  return TACO_OK;
}
int launch_colsum_batched(const float* x, int ld, float* out, int B, int T, int N, hipStream_t s) {
  TACO_KLAUNCH(colsum_batched_kernel, dim3((N + 63) / 64, B), dim3(64, 4), 0, s, x, ld, out, T, N);
  TACO_LAUNCH_CHECK("colsum_batched");
  return TACO_OK;
}
int launch_bn_maxpool(const float* x, const float* gamma, const float* beta, float* y, int B, int T, int C, hipStream_t s) {
  TACO_REQUIRE(C % 4 == 0, "bn_maxpool: C %% 4 != 0");
  EW_LAUNCH(bn_maxpool_kernel, (int64_t)B * T * (C / 4), s, x, gamma, beta, y, B, T, C);
  return TACO_OK;
}
static inline void col_grid(int64_t M, int N, dim3& grid, int& rpb) {
  int chunks = (int)((M + 31) / 32);   // 8 rows per thread: enough blocks to cover 256 CUs even for 128-column tensors
  if (chunks > 2048) chunks = 2048;
  if (taco_deterministic()) chunks = 1;   // one workgroup per column range sums all rows in a fixed order: a single atomicAdd each
  if (chunks < 1) chunks = 1;
  rpb = (int)((M + chunks - 1) / chunks);
  grid = dim3((N + 63) / 64, (unsigned)((M + rpb - 1) / rpb));
}
int launch_bn_maxpool_bwd(const float* x, const float* gamma, const float* beta, const float* dy, float* dx,
                          float* dgamma, float* dbeta, int B, int T, int C, hipStream_t s) {
  dim3 grid;
  int rpb;
  TACO_REQUIRE(C % 4 == 0, "bn_maxpool_bwd: C %% 4 != 0");
  col_grid((int64_t)B * T, C, grid, rpb);
  grid.x = (C / 4 + 63) / 64;
  TACO_KLAUNCH(bn_maxpool_bwd_kernel, grid, dim3(64, 4), 0, s, x, gamma, beta, dy, dx, dgamma, dbeta, B, T, C, rpb);
  TACO_LAUNCH_CHECK("bn_maxpool_bwd");
  return TACO_OK;
}
int launch_highway_combine(const float* th, const float* x, float* y, int64_t M, hipStream_t s) {
  EW_LAUNCH(highway_combine_kernel, M * (kCb / 4), s, th, x, y, M);
  return TACO_OK;
}
int launch_highway_combine_bwd(const float* th, const float* x, const float* dy, float* dth, float* dx, int64_t M,
                               hipStream_t s) {
  EW_LAUNCH(highway_combine_bwd_kernel, M * kCb, s, th, x, dy, dth, dx, M);
  return TACO_OK;
}
int launch_act_bwd(const float* y, const float* dy, const uint8_t* keep, float* dz, int64_t n, int act, hipStream_t s) {
  EW_LAUNCH(act_bwd_kernel, n, s, y, dy, keep, dz, n, act);
  return TACO_OK;
}
int launch_affine_act_bwd(const float* pre, const float* gamma, const float* dy, float* dz, float* dgamma, float* dbeta,
                          int64_t M, int N, int act, hipStream_t s) {
  dim3 grid;
  int rpb;
  col_grid(M, N, grid, rpb);
  TACO_KLAUNCH(affine_act_bwd_kernel, grid, dim3(64, 4), 0, s, pre, gamma, dy, dz, dgamma, dbeta, M, N, act, rpb);
  TACO_LAUNCH_CHECK("affine_act_bwd");
  return TACO_OK;
}
int launch_mask_rows(const float* x, const int32_t* len, float* y, int B, int T, int C, hipStream_t s) {
  TACO_REQUIRE(C % 4 == 0, "mask_rows: C %% 4 != 0");
  EW_LAUNCH(mask_rows_kernel, (int64_t)B * T * (C / 4), s, x, len, y, B, T, C);
  return TACO_OK;
}
int launch_add(const float* a, const float* b, float* y, int64_t n, hipStream_t s) {
  EW_LAUNCH(add_kernel, n, s, a, b, y, n);
  return TACO_OK;
}
int launch_l1(const float* a, const float* b, float* grad, int ldg, float* loss_parts, int64_t M, int N, hipStream_t s) {
  TACO_REQUIRE(ldg >= N, "l1: ldg < N");
  // exactly kLossParts blocks: block i leaves its partial in loss_parts[i] (blocks without work write 0)
  TACO_KLAUNCH(l1_kernel, dim3(kLossParts), dim3(1024), 0, s,   // 16 waves per workgroup: the kernel is latency bound, the partial count is fixed
                     a, b, grad, ldg, loss_parts, M, N);
  TACO_LAUNCH_CHECK("l1");
  return TACO_OK;
}
int launch_finish_loss(float* loss, const float* parts, float* out, hipStream_t s) {
  TACO_KLAUNCH(finish_loss_kernel, dim3(1), dim3(kThreads), 0, s, loss, parts, out);
  TACO_LAUNCH_CHECK("finish_loss");
  return TACO_OK;
}
int launch_transpose_batch(TransposeBatch& b, hipStream_t s) {
  TACO_REQUIRE(b.n >= 1 && b.n <= kMaxTransposeBatch, "transpose_batch: %d jobs out of range", b.n);
  int tiles = 0;
  for (int i = 0; i < b.n; ++i) {
    b.j[i].tile0 = tiles;
    tiles += b.j[i].taps * ((b.j[i].N + 31) / 32) * ((b.j[i].K + 31) / 32);
  }
  TACO_KLAUNCH(transpose_batch_kernel, dim3(tiles), dim3(32, 8), 0, s, b);
  TACO_LAUNCH_CHECK("transpose_batch");
  return TACO_OK;
}
int launch_denorm_unframe(const float* out, const float* mean, const float* stdv, float* spec, float* mag_t, int B, int Td,
                          int r, int C, hipStream_t s) {
  const int F = (Td / 4) * 4 * r;
  TACO_REQUIRE(F > 0, "denorm_unframe: Td=%d holds no whole chunk of 4 steps", Td);
  TACO_KLAUNCH(denorm_unframe_kernel, dim3((C + 31) / 32, (F + 31) / 32, B), dim3(32, 8), 0, s, out, mean, stdv, spec,
                     mag_t, Td, r, C, F);
  TACO_LAUNCH_CHECK("denorm_unframe");
  return TACO_OK;
}
int launch_sumsq(const float* x, int64_t n, float* out, hipStream_t s) {
  TACO_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "sumsq: x must be 16-byte aligned");
  // one block per CU, one partial each
  TACO_KLAUNCH(sumsq_kernel, dim3(kSumsqParts), dim3(kThreads), 0, s, x, n, out);
  TACO_LAUNCH_CHECK("sumsq");
  return TACO_OK;
}
int launch_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float cap, int64_t step,
                     const float* sumsq, float* gnorm_out, const int32_t* err, hipStream_t s) {
  const double b1 = 0.9, b2 = 0.999;
  const double lr_t = (double)lr * sqrt(1.0 - pow(b2, (double)step)) / (1.0 - pow(b1, (double)step));
  TACO_KLAUNCH(clip_adam_kernel, dim3(grid_for(n, kThreads, 2048)), dim3(kThreads), 0, s, p, g, m, v, n,
                     (float)lr_t, cap, sumsq, gnorm_out, err);
  TACO_LAUNCH_CHECK("clip_adam");
  return TACO_OK;
}
int launch_bernoulli(uint8_t* out, int64_t n, float p_one, uint64_t seed, hipStream_t s) {
  double t = (double)p_one * 4294967296.0;
  uint32_t thresh = t >= 4294967295.0 ? 0xFFFFFFFFu : (t <= 0 ? 0u : (uint32_t)t);
  TACO_KLAUNCH(bernoulli_kernel, dim3(grid_for((n + 1) / 2)), dim3(kThreads), 0, s, out, n, thresh, seed);
  TACO_LAUNCH_CHECK("bernoulli");
  return TACO_OK;
}

// ---- communication-kernel stand-in (taco_debug_spin): `blocks` workgroups of `threads` threads holding `lds_bytes` of LDS
//      each, spinning for `usec` microseconds of wall time.  The footprint of a collective kernel for co-residency tests of
//      the persistent decoder launches (tests/test_gpu_dist.py); does no work.
__global__ void spin_kernel(long long ticks, int lds_words) {
  extern __shared__ __attribute__((aligned(16))) int spin_lds[];
  for (int i = threadIdx.x; i < lds_words; i += blockDim.x) spin_lds[i] = i;   // touch the allocation
  __syncthreads();
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  if (lds_words > 0 && spin_lds[(threadIdx.x * 7) % lds_words] == -12345) spin_lds[0] = 1;   // keeps the touch alive
}
int launch_spin(int blocks, int threads, int lds_bytes, int usec, hipStream_t s) {
  TACO_REQUIRE(blocks > 0 && threads > 0 && threads <= 1024 && lds_bytes >= 0 && lds_bytes <= 160 * 1024 && usec >= 0,
               "debug_spin: bad arguments");
  if (lds_bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) {
      taco_set_error("debug_spin: hipFuncSetAttribute: %s", hipGetErrorString(e));
      return TACO_ELAUNCH;
    }
  }
  int rate_khz = 100000;   // wall_clock64 ticks at the constant s_memrealtime rate (100 MHz on gfx950); ask the runtime
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, dev);
  if (rate_khz <= 0) rate_khz = 100000;
  TACO_KLAUNCH(spin_kernel, dim3(blocks), dim3(threads), (size_t)lds_bytes, s, (long long)usec * rate_khz / 1000, lds_bytes / 4);
  TACO_LAUNCH_CHECK("debug_spin");
  return TACO_OK;
}

// ---- shader-clock probe (taco_debug_clock_probe): every CU runs one 512-thread workgroup whose waves each walk a chain of
//      `iters` dependent v_fma_f32 -- the occupancy and issue pattern of the persistent decoder kernels, without their memory
//      traffic -- and workgroup 0 reports the elapsed shader cycles (s_memtime) and constant-rate ticks (s_memrealtime, 100 MHz).
//      cycles / time = the clock this chip sustains under a chip-wide LATENCY-BOUND load, which is what the decoder and bi-GRU
//      kernels scale with; boxes of one pool differ (round 4: the same build ran 8.86 and 9.30 ms per step).
__global__ __launch_bounds__(512) void clock_probe_kernel(long long* out, int iters, float a, float b) {
  float x = (float)threadIdx.x;
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; i += 16) {
#pragma unroll
    for (int j = 0; j < 16; ++j) x = fmaf(x, a, b);
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = w1 - w0;
  }
  if (x == 12345.678f) out[2] = 1;   // keeps the chain alive
}
int launch_clock_probe(long long* out, int iters, hipStream_t s) {
  TACO_REQUIRE(out && iters > 0, "clock_probe: bad arguments");
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  TACO_KLAUNCH(clock_probe_kernel, dim3(cus > 0 ? cus : 256), dim3(512), 0, s, out, iters, 0.999f, 0.001f);
  TACO_LAUNCH_CHECK("clock_probe");
  return TACO_OK;
}

// ---- fabric probe (taco_debug_fabric_probe): what the latency-bound kernels of this library actually wait on, measured on THIS
//      box (the shader clock does not tell a 12.4 us BPTT box from a 14.0 us one).  One launch, 64 workgroups of 64
//      threads (workgroup b runs on XCD b % 8 with the dispatcher's round-robin; the XCC ids are recorded, not assumed):
//        blocks 0 <-> 8   ping-pong through an 8-byte {epoch, value} granule in decoder3's fast form: workgroup-scope store
//                         (stays in the XCD's L2) + agent-scope load (bypasses L1)                     -> out[0] ticks / hop pair
//        blocks 1 <-> 9   the same with agent-scope stores (written through; decoder3's fallback form) -> out[1]
//        blocks 2 <-> 3   agent scope, two DIFFERENT XCDs (decoder.hip's exchange)                     -> out[2]
//        block  20        a chain of dependent loads over a warm 64 KB region (L2 hits)                 -> out[3]
//        block  21        a chain of dependent AGENT-SCOPE loads over the whole scratch buffer (cold lines: the memory-side
//                         round trip an L2-bypassing load pays when no peer has just written the line)          -> out[4]
//        block  22        64 lanes streaming 8 MB of the scratch buffer with 16-byte loads (one CU's stream bandwidth) -> out[5]
//      Ticks are the constant 100 MHz counter; out[8 + b] = XCC id of block b (b < 24); out[6] = iterations, out[7] = ok flags.
typedef unsigned long long fp_u64;
typedef __attribute__((address_space(1))) fp_u64 fp_gu64;
template <int LSCOPE, int SSCOPE>
__device__ __forceinline__ long long fabric_pingpong(fp_u64* mine, fp_u64* theirs, bool first, int iters, bool* ok_out) {
  bool ok = true;
  const long long t0 = wall_clock64();
  for (int i = 1; i <= iters && ok; ++i) {
    if (first) __hip_atomic_store((fp_gu64*)mine, ((fp_u64)i << 32) | (fp_u64)i, __ATOMIC_RELAXED, SSCOPE);
    ok = false;
    for (unsigned spin = 0; spin < (1u << 20); ++spin) {
      const fp_u64 x = __hip_atomic_load((fp_gu64*)theirs, __ATOMIC_RELAXED, LSCOPE);
      if ((unsigned)(x >> 32) == (unsigned)i) { ok = true; break; }
    }
    if (!first) __hip_atomic_store((fp_gu64*)mine, ((fp_u64)i << 32) | (fp_u64)i, __ATOMIC_RELAXED, SSCOPE);
  }
  *ok_out = ok;
  return wall_clock64() - t0;
}
__global__ __launch_bounds__(64) void fabric_probe_kernel(long long* out, fp_u64* gran, const unsigned* scratch, long long scratch_words, int iters) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0 && b < 24) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    out[8 + b] = id & 0xf;
  }
  if (b == 22) {   // one CU streaming: 8 MB, 16 bytes per lane and load, 8 loads in flight per lane
    typedef unsigned fp_u32x4 __attribute__((ext_vector_type(4)));
    const fp_u32x4* src = reinterpret_cast<const fp_u32x4*>(scratch);
    const long long n16 = (8ll << 20) / 16;
    unsigned acc = 0;
    const long long t0 = wall_clock64();
    for (long long i = threadIdx.x; i + 7 * 64 < n16; i += 8 * 64) {
      fp_u32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(src + i + j * 64);
#pragma unroll
      for (int j = 0; j < 8; ++j) acc += v[j][0] ^ v[j][3];
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[5] = t1 - t0;
    if (acc == 0x12345678u) out[7] |= 1ll << 40;
    return;
  }
  if (threadIdx.x != 0) return;
  bool ok = true;
  if (b == 0 || b == 8) {
    const long long t = fabric_pingpong<__HIP_MEMORY_SCOPE_AGENT, __HIP_MEMORY_SCOPE_WORKGROUP>(gran + (b == 0 ? 0 : 64), gran + (b == 0 ? 64 : 0), b == 0, iters, &ok);
    if (b == 0) { out[0] = t; if (ok) atomicOr((unsigned long long*)&out[7], 1ull); }
  } else if (b == 1 || b == 9) {
    const long long t = fabric_pingpong<__HIP_MEMORY_SCOPE_AGENT, __HIP_MEMORY_SCOPE_AGENT>(gran + (b == 1 ? 128 : 192), gran + (b == 1 ? 192 : 128), b == 1, iters, &ok);
    if (b == 1) { out[1] = t; if (ok) atomicOr((unsigned long long*)&out[7], 2ull); }
  } else if (b == 2 || b == 3) {
    const long long t = fabric_pingpong<__HIP_MEMORY_SCOPE_AGENT, __HIP_MEMORY_SCOPE_AGENT>(gran + (b == 2 ? 256 : 320), gran + (b == 2 ? 320 : 256), b == 2, iters, &ok);
    if (b == 2) { out[2] = t; if (ok) atomicOr((unsigned long long*)&out[7], 4ull); }
  } else if (b == 20 || b == 21) {
    // dependent loads: the next index is a multiplicative walk over the region PLUS the loaded word (the buffer holds zeros, which
    // the compiler cannot know), 32 words = one 128-byte line apart.  The L2-hit walk covers 64 KB and touches every line once
    // before the clock starts (64 KB exceeds the CU's L1: the walk is served by the XCD's L2); plain loads -- agent-scope loads
    // were measured to take the same ~320 ns whether the line is in the L2 or nowhere, i.e. they are fabric round trips
    long long lines = (b == 20 ? (64ll << 10) : scratch_words * 4) / 128;
    while (lines & (lines - 1)) lines &= lines - 1;   // largest power of two: the walk is a mask, not a 64-bit software modulo
    const unsigned long long mask = (unsigned long long)lines - 1;
    unsigned long long idx = 1;
    if (b == 20) {
      unsigned w = 0;
      for (long long l = 0; l < lines; ++l)
        w += scratch[(16ll << 20) / 4 + l * 32];
      idx += w;
    }
    const unsigned* base = b == 20 ? scratch + (16ll << 20) / 4 : scratch;   // (the streaming block reads the first 8 MB)
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
      // block 20: a plain load (served by the L2 after the warm-up); block 21: an agent-scope load, which bypasses L1 AND is
      // not served from another XCD's L2 -- the kind of load every exchange poll is, here on lines nobody has written lately
      const unsigned v = b == 20 ? base[idx * 32]
                                 : __hip_atomic_load((const __attribute__((address_space(1))) unsigned*)(base + idx * 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      idx = (idx * 1664525ull + 1013904223ull + v) & mask;   // (full-period LCG modulo 2^k)
    }
    const long long t1 = wall_clock64();
    out[b == 20 ? 3 : 4] = t1 - t0;
    if (idx == 0xfffffffffull) out[7] |= 1ll << 41;
  }
  if (b == 0) out[6] = iters;
}
int launch_fabric_probe(long long* out32, void* gran4k, const void* scratch, int64_t scratch_bytes, int iters, hipStream_t s) {
  TACO_REQUIRE(out32 && gran4k && scratch && scratch_bytes >= (32 << 20) && iters > 0, "fabric_probe: bad arguments");
  TACO_KLAUNCH(fabric_probe_kernel, dim3(64), dim3(64), 0, s, out32, reinterpret_cast<fp_u64*>(gran4k),
                     reinterpret_cast<const unsigned*>(scratch), (long long)(scratch_bytes / 4), iters);
  TACO_LAUNCH_CHECK("fabric_probe");
  return TACO_OK;
}

// ---- batched zero-fill / copy (InitBatch, kernels.h) ----
__global__ __launch_bounds__(256) void init_batch_kernel(InitBatch b) {
  int ji = 0;
#pragma unroll 1
  for (int i = 1; i < b.n; ++i)
    if ((int)blockIdx.x >= b.j[i].blk0) ji = i;
  const InitJob& q = b.j[ji];
  const int nb = (ji + 1 < b.n ? b.j[ji + 1].blk0 : (int)gridDim.x) - q.blk0;
  const int64_t total = (int64_t)q.rows * q.cols;
  const bool vec = (q.cols % 4 == 0) && (q.ldd % 4 == 0) && (q.lds % 4 == 0) && ((reinterpret_cast<uintptr_t>(q.dst) & 15) == 0) &&
                   (q.src == nullptr || (reinterpret_cast<uintptr_t>(q.src) & 15) == 0);
  if (vec) {
    const int64_t c4 = q.cols / 4, n4 = total / 4;
    for (int64_t i = (int64_t)(blockIdx.x - q.blk0) * 256 + threadIdx.x; i < n4; i += (int64_t)nb * 256) {
      const int64_t r = i / c4, c = (i - r * c4) * 4;
      const float4 v = q.src ? *reinterpret_cast<const float4*>(q.src + r * q.lds + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(q.dst + r * q.ldd + c) = v;
    }
  } else {
    for (int64_t i = (int64_t)(blockIdx.x - q.blk0) * 256 + threadIdx.x; i < total; i += (int64_t)nb * 256) {
      const int64_t r = i / q.cols, c = i - r * q.cols;
      q.dst[r * q.ldd + c] = q.src ? q.src[r * q.lds + c] : 0.f;
    }
  }
}
int launch_init_batch(InitBatch& b, hipStream_t s) {
  if (b.n == 0) return TACO_OK;
  int blocks = 0;
  for (int i = 0; i < b.n; ++i) {
    b.j[i].blk0 = blocks;
    const int64_t total = (int64_t)b.j[i].rows * b.j[i].cols;
    int64_t nb = (total + 256 * 16 - 1) / (256 * 16);      // 16 floats (4 float4) per thread at most
    nb = nb < 1 ? 1 : (nb > 2048 ? 2048 : nb);
    blocks += (int)nb;
  }
  TACO_KLAUNCH(init_batch_kernel, dim3(blocks), dim3(256), 0, s, b);
  TACO_LAUNCH_CHECK("init_batch");
  b.n = 0;
  return TACO_OK;
}
