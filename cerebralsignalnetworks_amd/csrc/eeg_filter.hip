// K1+K2: fused Butterworth band-pass (biquad cascade, causal) + per-channel z-score.
// Replaces scipy.signal filtering named by /root/reference/utils/EEGFilters.py:2,26 and
// EEGDataset.normlizeEEG (/root/reference/utils/PerilsEEGDataset.py:454-461).
//
// Data layout in HBM: x[B][C][T] float32 (channel-first, contiguous along time, as the
// reference stores segments) -> y[B][T][C] or y[T][B][C] (channels fastest: what the LSTM's
// input projection reads).  Algorithmic traffic: read 4*C*T + write sizeof(out)*C*T per segment.
//
// v1 kernel ("rows"): one lane per (segment, channel) row, 64 consecutive channels per wave so
// that every output store instruction writes 64 consecutive channels of one time step
// (256 B coalesced).  The IIR recurrence and the statistics are carried in float64 (full-rate
// v_fma_f64 on gfx950): the poles of this band sit at radius 0.9997, where float32 state costs
// ~2e-4 absolute over 500 samples.  Two passes over the row: pass 1
// accumulates sum / sum of squares of the filtered signal, pass 2 re-runs the recurrence and
// writes (y - mean) * rsqrt(var); the second read of the row is served by L2.
#include "csn_common.h"

namespace csn {

struct SosParams {
  double c[8][5];  // b0 b1 b2 a1 a2, normalised by a0
};

template <int NSEC>
__device__ __forceinline__ double biquad_cascade(double v, const SosParams& p, double (&s1)[8], double (&s2)[8]) {
#pragma unroll
  for (int s = 0; s < NSEC; ++s) {
    const double y = fma(p.c[s][0], v, s1[s]);
    s1[s] = fma(p.c[s][1], v, fma(-p.c[s][3], y, s2[s]));
    s2[s] = fma(p.c[s][2], v, -p.c[s][4] * y);
    v = y;
  }
  return v;
}

template <int NSEC, typename OutT>
__global__ void __launch_bounds__(64) eeg_filter_rows_kernel(const float* __restrict__ x, OutT* __restrict__ y,
                                                             int B, int C, int T, SosParams p, int ddof,
                                                             int time_major) {
  const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (row >= (int64_t)B * C) return;
  const int b = (int)(row / C), c = (int)(row % C);
  const float* xr = x + row * (int64_t)T;
  const bool vec = (T % 4 == 0);

  double s1[8], s2[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) { s1[s] = 0.0; s2[s] = 0.0; }
  double sum = 0.0, sumsq = 0.0;
  if (vec) {
    const float4* xv = reinterpret_cast<const float4*>(xr);
#pragma unroll 2
    for (int t4 = 0; t4 < T / 4; ++t4) {
      const float4 q = xv[t4];
      const float in[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double v = biquad_cascade<NSEC>((double)in[j], p, s1, s2);
        sum += v;
        sumsq = fma(v, v, sumsq);
      }
    }
  } else {
    for (int t = 0; t < T; ++t) {
      const double v = biquad_cascade<NSEC>((double)xr[t], p, s1, s2);
      sum += v;
      sumsq = fma(v, v, sumsq);
    }
  }
  const double mean = sum / (double)T;
  const double var = (sumsq - sum * mean) / (double)(T - ddof);
  const double inv = 1.0 / sqrt(var);

#pragma unroll
  for (int s = 0; s < 8; ++s) { s1[s] = 0.0; s2[s] = 0.0; }
  const int64_t t_stride = time_major ? (int64_t)B * C : (int64_t)C;
  OutT* yo = y + (time_major ? (int64_t)b * C + c : ((int64_t)b * T) * C + c);
  if (vec) {
    const float4* xv = reinterpret_cast<const float4*>(xr);
#pragma unroll 2
    for (int t4 = 0; t4 < T / 4; ++t4) {
      const float4 q = xv[t4];
      const float in[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const double v = biquad_cascade<NSEC>((double)in[j], p, s1, s2);
        yo[(int64_t)(t4 * 4 + j) * t_stride] = from_f32<OutT>((float)((v - mean) * inv));
      }
    }
  } else {
    for (int t = 0; t < T; ++t) {
      const double v = biquad_cascade<NSEC>((double)xr[t], p, s1, s2);
      yo[(int64_t)t * t_stride] = from_f32<OutT>((float)((v - mean) * inv));
    }
  }
}

template <int NSEC>
static int launch_rows(const float* x, void* y, int B, int C, int T, const SosParams& p, int ddof, int out_dtype,
                       int time_major, hipStream_t st) {
  const int64_t rows = (int64_t)B * C;
  const unsigned grid = (unsigned)((rows + 63) / 64);
  if (out_dtype == CSN_BF16)
    eeg_filter_rows_kernel<NSEC, bf16_t><<<grid, 64, 0, st>>>(x, (bf16_t*)y, B, C, T, p, ddof, time_major);
  else
    eeg_filter_rows_kernel<NSEC, float><<<grid, 64, 0, st>>>(x, (float*)y, B, C, T, p, ddof, time_major);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

}  // namespace csn

extern "C" int csn_eeg_bandpass_znorm(const float* x, int B, int C, int T, const double* sos, int nsec, int ddof,
                                      void* y, int out_dtype, int time_major, csnStream_t stream) {
  using namespace csn;
  CSN_REQUIRE(x && y, "csn_eeg_bandpass_znorm: null pointer");
  CSN_REQUIRE(B > 0 && C > 0 && T > 1, "csn_eeg_bandpass_znorm: bad shape B=%d C=%d T=%d", B, C, T);
  CSN_REQUIRE(nsec >= 0 && nsec <= 8, "csn_eeg_bandpass_znorm: nsec=%d outside 0..8", nsec);
  CSN_REQUIRE(nsec == 0 || sos, "csn_eeg_bandpass_znorm: sos is null");
  CSN_REQUIRE(ddof == 0 || ddof == 1, "csn_eeg_bandpass_znorm: ddof must be 0 or 1");
  CSN_REQUIRE(out_dtype == CSN_F32 || out_dtype == CSN_BF16, "csn_eeg_bandpass_znorm: bad out_dtype %d", out_dtype);
  CSN_REQUIRE((reinterpret_cast<uintptr_t>(x) & 15) == 0, "csn_eeg_bandpass_znorm: x must be 16-byte aligned");
  SosParams p;
  for (int s = 0; s < 8; ++s)
    for (int k = 0; k < 5; ++k) p.c[s][k] = 0.0;
  for (int s = 0; s < nsec; ++s) {
    const double a0 = sos[s * 6 + 3];
    CSN_REQUIRE(a0 != 0.0, "csn_eeg_bandpass_znorm: section %d has a0 == 0", s);
    p.c[s][0] = sos[s * 6 + 0] / a0;
    p.c[s][1] = sos[s * 6 + 1] / a0;
    p.c[s][2] = sos[s * 6 + 2] / a0;
    p.c[s][3] = sos[s * 6 + 4] / a0;
    p.c[s][4] = sos[s * 6 + 5] / a0;
  }
  hipStream_t st = as_stream(stream);
  switch (nsec) {
    case 0: return launch_rows<0>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
    case 1: return launch_rows<1>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
    case 2: return launch_rows<2>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
    case 3: return launch_rows<3>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
    case 4: return launch_rows<4>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
    case 5: return launch_rows<5>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
    case 6: return launch_rows<6>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
    case 7: return launch_rows<7>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
    default: return launch_rows<8>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
  }
}
