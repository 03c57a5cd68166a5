// vocoder.hip -- Griffin-Lim phase reconstruction on the GPU (audio.griffinlim / invert_spectrogram,
// audio.py:69-97: 50 rounds of librosa.istft -> librosa.stft with n_fft 2048, win_length 1200, hop_length 300, 'hann').
//
// One workgroup (256 threads) owns one frame: a 2048-point complex FFT in LDS (in-place radix-2, bit-reversed load, a
// 1024-entry twiddle table computed once per workgroup with sincospif), no vendor FFT.  One Griffin-Lim round is two launches:
//   gl_synth : frame t: Hermitian-extend mag * e^{i angle} -> inverse FFT -> times the zero-padded periodic Hann window ->
//              the 1200 non-zero samples of the frame's segment to `seg` (B, F, 1200)
//   gl_anal  : frame t: gather its 2048 input samples on the fly = overlap-add of <= 4 segments, divided by the window
//              sum-of-squares, trimmed by n_fft/2 and reflect-padded exactly as librosa's center=True STFT does -> window ->
//              forward FFT -> new unit-modulus angles (B, F, 1025)
// The overlap-add is a GATHER (each output sample sums the segments that cover it, in frame order): no atomics, results are
// reproducible.  gl_wave applies the same gather once more for the final waveform.
#include <algorithm>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int NFFT = 2048, NBIN = 1025, WIN = 1200, HOP = 300, WOFF = (NFFT - WIN) / 2;   // window occupies [424, 1624)
constexpr int FT = 256;

__device__ __forceinline__ float hann(int i) {   // periodic Hann(1200), i in [0, 1200)
  float s, c;
  sincospif(2.0f * (float)i / (float)WIN, &s, &c);
  return 0.5f - 0.5f * c;
}

// in-place 2048-point FFT of (re, im) in LDS; data must already be in bit-reversed order.  sign = -1 forward, +1 inverse
// (unscaled).  tw[k] = e^{-2 pi i k / 2048}, k < 1024.
__device__ __forceinline__ void fft2048(float* re, float* im, const float* twr, const float* twi, float sign) {
#pragma unroll 1
  for (int s = 0; s < 11; ++s) {
    const int half = 1 << s;
    __syncthreads();
    for (int j = threadIdx.x; j < NFFT / 2; j += FT) {
      const int pos = j & (half - 1);
      const int i0 = ((j >> s) << (s + 1)) + pos, i1 = i0 + half;
      const int k = pos << (10 - s);
      const float wr = twr[k], wi = -sign * twi[k];   // twi holds sin(-2 pi k / N): forward uses it as is
      const float xr = re[i1], xi = im[i1];
      const float tr = xr * wr - xi * wi, ti = xr * wi + xi * wr;
      const float ur = re[i0], ui = im[i0];
      re[i0] = ur + tr; im[i0] = ui + ti;
      re[i1] = ur - tr; im[i1] = ui - ti;
    }
  }
  __syncthreads();
}
__device__ __forceinline__ int bitrev11(int x) { return (int)(__brev((unsigned)x) >> 21); }

__device__ __forceinline__ void make_twiddles(float* twr, float* twi) {
  for (int k = threadIdx.x; k < NFFT / 2; k += FT) {
    float s, c;
    sincospif(-2.0f * (float)k / (float)NFFT, &s, &c);
    twr[k] = c;
    twi[k] = s;
  }
}

// window sum-of-squares (librosa.filters.window_sumsquare) over n = NFFT + HOP (F - 1) samples
__global__ void gl_wss_kernel(float* __restrict__ wss, int F) {
  const int n = NFFT + HOP * (F - 1);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float acc = 0.f;
    // frames t with t*HOP + WOFF <= i < t*HOP + WOFF + WIN
    int t_hi = (i - WOFF) / HOP;
    if (i - WOFF < 0) t_hi = -1;
    for (int t = t_hi; t >= 0 && t > t_hi - 4; --t) {
      const int j = i - t * HOP - WOFF;
      if (t < F && j >= 0 && j < WIN) {
        const float w = hann(j);
        acc += w * w;
      }
    }
    wss[i] = acc;
  }
}

// angles (B, F, NBIN, 2) <- unit phasors of the given phase angles (radians)
__global__ void gl_init_kernel(const float* __restrict__ phase, float* __restrict__ ang, int F, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t bt = i / NBIN;
    const int k = (int)(i - bt * NBIN);
    const int64_t b = bt / F;
    const int t = (int)(bt - b * F);
    float s, c;
    sincosf(phase[(b * NBIN + k) * F + t], &s, &c);   // phase is (B, NBIN, F) like the magnitude matrix
    ang[i * 2] = c;
    ang[i * 2 + 1] = s;
  }
}

__global__ __launch_bounds__(FT) void gl_synth_kernel(const float* __restrict__ mag_t, const float* __restrict__ ang,
                                                      float* __restrict__ seg, int F) {
  __shared__ float re[NFFT], im[NFFT], twr[NFFT / 2], twi[NFFT / 2];
  const int t = blockIdx.x, b = blockIdx.y;
  make_twiddles(twr, twi);
  const float* a = ang + ((int64_t)b * F + t) * NBIN * 2;
  const float* m = mag_t + (int64_t)b * NBIN * F + t;
  // X[k] = mag e^{i angle}, Hermitian extension X[N - k] = conj(X[k]); stored bit-reversed for the in-place FFT
  for (int k = threadIdx.x; k < NBIN; k += FT) {
    const float mg = fabsf(m[(int64_t)k * F]);
    const float xr = mg * a[2 * k], xi = mg * a[2 * k + 1];
    const int r0 = bitrev11(k);
    re[r0] = xr; im[r0] = xi;
    if (k > 0 && k < NFFT / 2) {
      const int r1 = bitrev11(NFFT - k);
      re[r1] = xr; im[r1] = -xi;
    }
  }
  fft2048(re, im, twr, twi, +1.0f);
  float* o = seg + ((int64_t)b * F + t) * WIN;
  for (int j = threadIdx.x; j < WIN; j += FT) o[j] = re[WOFF + j] * (1.0f / NFFT) * hann(j);
}

// sample i of the overlap-added, normalised signal of length NFFT + HOP (F - 1) (before the centre trim)
__device__ __forceinline__ float ola_sample(const float* __restrict__ seg_b, const float* __restrict__ wss, int i, int F) {
  float acc = 0.f;
  int t_hi = (i - WOFF) / HOP;
  if (i - WOFF < 0) return 0.f;
  if (t_hi > F - 1) t_hi = F - 1;
  // frames in increasing order (fixed summation order)
  int t_lo = t_hi - 3;
  if (t_lo < 0) t_lo = 0;
  for (int t = t_lo; t <= t_hi; ++t) {
    const int j = i - t * HOP - WOFF;
    if (j >= 0 && j < WIN) acc += seg_b[(int64_t)t * WIN + j];
  }
  const float w = wss[i];
  return w > 1.17549435e-38f ? acc / w : acc;
}

__global__ __launch_bounds__(FT) void gl_anal_kernel(const float* __restrict__ seg, const float* __restrict__ wss,
                                                     float* __restrict__ ang, int F) {
  __shared__ float re[NFFT], im[NFFT], twr[NFFT / 2], twi[NFFT / 2];
  const int t = blockIdx.x, b = blockIdx.y;
  make_twiddles(twr, twi);
  const int L = HOP * (F - 1);   // length of the trimmed signal y
  const float* sb = seg + (int64_t)b * F * WIN;
  for (int j = threadIdx.x; j < NFFT; j += FT) {
    float v = 0.f;
    if (j >= WOFF && j < WOFF + WIN) {
      // padded signal index p = t*HOP + j over yp = reflect_pad(y, NFFT/2): y index q = p - NFFT/2 reflected into [0, L)
      int q = t * HOP + j - NFFT / 2;
      if (q < 0) q = -q;
      if (q >= L) q = 2 * (L - 1) - q;
      v = ola_sample(sb, wss, q + NFFT / 2, F) * hann(j - WOFF);
    }
    const int r0 = bitrev11(j);
    re[r0] = v;
    im[r0] = 0.f;
  }
  fft2048(re, im, twr, twi, -1.0f);
  float* a = ang + ((int64_t)b * F + t) * NBIN * 2;
  for (int k = threadIdx.x; k < NBIN; k += FT) {
    const float xr = re[k], xi = im[k];
    const float n2 = xr * xr + xi * xi;
    float c = 1.f, s = 0.f;   // np.angle(0) = 0
    if (n2 > 0.f) {
      const float inv = rsqrtf(n2);
      c = xr * inv;
      s = xi * inv;
    }
    a[2 * k] = c;
    a[2 * k + 1] = s;
  }
}

__global__ void gl_wave_kernel(const float* __restrict__ seg, const float* __restrict__ wss, float* __restrict__ wave, int F) {
  const int L = HOP * (F - 1);
  const int b = blockIdx.y;
  const float* sb = seg + (int64_t)b * F * WIN;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < L; i += gridDim.x * blockDim.x)
    wave[(int64_t)b * L + i] = ola_sample(sb, wss, i + NFFT / 2, F);
}

}  // namespace

int64_t griffinlim_workspace_floats(int B, int F) {
  return (int64_t)B * F * NBIN * 2 + (int64_t)B * F * WIN + (NFFT + (int64_t)HOP * (F - 1)) + 64;
}

int launch_griffinlim(const float* mag_t, const float* phase0, float* wave, float* work, int B, int F, int n_iter,
                      hipStream_t s) {
  TACO_REQUIRE(mag_t && phase0 && wave && work && B > 0 && n_iter >= 0, "griffinlim: bad arguments");
  TACO_REQUIRE(F >= 5, "griffinlim: F=%d frames < 5 (the reflect padding of n_fft/2 needs more than 1024 samples)", F);
  float* ang = work;
  float* seg = ang + (int64_t)B * F * NBIN * 2;
  float* wss = seg + (int64_t)B * F * WIN;
  const int n = NFFT + HOP * (F - 1);
  TACO_KLAUNCH(gl_wss_kernel, dim3((n + 255) / 256), dim3(256), 0, s, wss, F);
  const int64_t total = (int64_t)B * F * NBIN;
  TACO_KLAUNCH(gl_init_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 4096)), dim3(256), 0, s, phase0, ang, F,
                     total);
  for (int it = 0; it < n_iter; ++it) {
    TACO_KLAUNCH(gl_synth_kernel, dim3(F, B), dim3(FT), 0, s, mag_t, ang, seg, F);
    TACO_KLAUNCH(gl_anal_kernel, dim3(F, B), dim3(FT), 0, s, seg, wss, ang, F);
  }
  TACO_KLAUNCH(gl_synth_kernel, dim3(F, B), dim3(FT), 0, s, mag_t, ang, seg, F);
  TACO_KLAUNCH(gl_wave_kernel, dim3((HOP * (F - 1) + 255) / 256, B), dim3(256), 0, s, seg, wss, wave, F);
  TACO_LAUNCH_CHECK("griffinlim");
  return TACO_OK;
}
