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

// ---------------------------------------------------------------------------------------------
// v2 kernel ("scan"): the time recurrence is parallelised, so the kernel is HBM-bound instead of
// bound by one lane walking 500 dependent steps.
//   A workgroup takes 16 consecutive rows (channels) and cuts each into 16 chunks of 32 samples;
//   thread (row, chunk) filters its chunk from a ZERO state (float64), giving the zero-state
//   response and the chunk's end state.  The cascade is linear, so the true output is
//       y[n] = y_zero_state[n] + sum_i s0_i * phi_i[n],   s0(chunk k+1) = A^32 s0(chunk k) + end_k
//   with phi_i = response to a unit initial state component and A^32 the 32-step state transition --
//   both produced once per call by a tiny basis kernel.  Row statistics come from the corrected
//   chunks; everything stays in LDS between the coalesced load (16 rows are one contiguous 32 KB
//   run of x) and the transposed, channel-fastest store.
// LDS: 16 x 546 floats (chunk stride 33, row stride 546 = 2 mod 32: conflict-free for the
// per-chunk walk AND for the transposed read of the store phase) + end states + statistics.
// ---------------------------------------------------------------------------------------------
static constexpr int kScanRows = 16, kScanChunks = 16, kScanLen = 32, kScanRS = 546;

struct ScanBasis {
  double phi[16][kScanLen];   // [state component][n]
  double apow[16][16];        // [j][i] = component j of the state after kScanLen steps from e_i
};

template <int NSEC>
__global__ void __launch_bounds__(64) eeg_filter_basis_kernel(SosParams p, ScanBasis* out) {
  const int i = threadIdx.x;
  if (i >= 2 * NSEC) return;
  double s1[8], s2[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) { s1[s] = 0.0; s2[s] = 0.0; }
#pragma unroll
  for (int s = 0; s < NSEC; ++s) {
    if (i == 2 * s) s1[s] = 1.0;
    if (i == 2 * s + 1) s2[s] = 1.0;
  }
  for (int n = 0; n < kScanLen; ++n) out->phi[i][n] = biquad_cascade<NSEC>(0.0, p, s1, s2);
#pragma unroll
  for (int s = 0; s < NSEC; ++s) {
    out->apow[2 * s][i] = s1[s];
    out->apow[2 * s + 1][i] = s2[s];
  }
}

template <int NSEC, typename OutT>
__global__ void __launch_bounds__(256)
eeg_filter_scan_kernel(const float* __restrict__ x, OutT* __restrict__ y, int B, int C, int T, SosParams p,
                       const ScanBasis* __restrict__ basis, int ddof, int time_major) {
  constexpr int NS = 2 * NSEC > 0 ? 2 * NSEC : 1;
  __shared__ float xs[kScanRows * kScanRS];
  __shared__ double ez[kScanRows][kScanChunks][NS];
  __shared__ double st[kScanRows][kScanChunks][2];
  __shared__ double phi_s[NS][kScanLen];
  __shared__ double apow_s[NS][NS];
  const int tid = threadIdx.x;
  const int64_t rows_total = (int64_t)B * C;
  const int64_t row0 = (int64_t)blockIdx.x * kScanRows;
  const int nrows = (int)((rows_total - row0 < kScanRows) ? rows_total - row0 : kScanRows);

  // ---- phase 1: coalesced load of nrows x T floats (one contiguous run of x) into the skewed tile
  {
    const float* src = x + row0 * (int64_t)T;
    const int total = nrows * T;
    if ((T & 3) == 0) {
      for (int e = tid * 4; e < total; e += 256 * 4) {
        const float4 v = *reinterpret_cast<const float4*>(src + e);
        const int r = e / T, t = e - r * T;
        float* d = xs + r * kScanRS + (t >> 5) * 33 + (t & 31);
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    } else {
      for (int e = tid; e < total; e += 256) {
        const int r = e / T, t = e - r * T;
        xs[r * kScanRS + (t >> 5) * 33 + (t & 31)] = src[e];
      }
    }
    for (int e = tid; e < NS * kScanLen; e += 256) phi_s[e / kScanLen][e % kScanLen] = basis->phi[e / kScanLen][e % kScanLen];
    for (int e = tid; e < NS * NS; e += 256) apow_s[e / NS][e % NS] = basis->apow[e / NS][e % NS];
  }
  __syncthreads();

  // ---- phase 2: zero-state response of chunk k of row r, in place
  const int r = tid & 15, k = tid >> 4;
  const int t_beg = k * kScanLen;
  const int len = (t_beg >= T) ? 0 : ((T - t_beg < kScanLen) ? T - t_beg : kScanLen);
  float* mine = xs + r * kScanRS + k * 33;
  {
    double s1[8], s2[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) { s1[s] = 0.0; s2[s] = 0.0; }
    if (r < nrows)
      for (int j = 0; j < len; ++j) mine[j] = (float)biquad_cascade<NSEC>((double)mine[j], p, s1, s2);
#pragma unroll
    for (int s = 0; s < NSEC; ++s) {
      ez[r][k][2 * s] = s1[s];
      ez[r][k][2 * s + 1] = s2[s];
    }
  }
  __syncthreads();

  // ---- phase 3: true initial state of this chunk = fold of the earlier chunks' end states
  double s0[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) s0[i] = 0.0;
  for (int kk = 0; kk < k; ++kk) {
    double nx[NS];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      double a = ez[r][kk][j];
#pragma unroll
      for (int i = 0; i < NS; ++i) a = fma(apow_s[j][i], s0[i], a);
      nx[j] = a;
    }
#pragma unroll
    for (int j = 0; j < NS; ++j) s0[j] = nx[j];
  }

  // ---- phase 4: add the homogeneous response, accumulate the row statistics
  double sum = 0.0, sumsq = 0.0;
  if (r < nrows)
    for (int j = 0; j < len; ++j) {
      double v = (double)mine[j];
#pragma unroll
      for (int i = 0; i < (NSEC > 0 ? NS : 0); ++i) v = fma(s0[i], phi_s[i][j], v);
      sum += v;
      sumsq = fma(v, v, sumsq);
      mine[j] = (float)v;
    }
  st[r][k][0] = sum;
  st[r][k][1] = sumsq;
  __syncthreads();

  // ---- phase 5: normalise and store channel-fastest: lane = (channel c, time phase tq)
  const int c = tid & 15, tq = tid >> 4;
  if (c >= nrows) return;
  double tsum = 0.0, tsq = 0.0;
#pragma unroll
  for (int kk = 0; kk < kScanChunks; ++kk) { tsum += st[c][kk][0]; tsq += st[c][kk][1]; }
  const double mean = tsum / (double)T;
  const double inv = 1.0 / sqrt((tsq - tsum * mean) / (double)(T - ddof));
  const float meanf = (float)mean, invf = (float)inv;
  const int64_t row = row0 + c;
  const int b = (int)(row / C), ch = (int)(row % C);
  const int64_t t_stride = time_major ? (int64_t)B * C : (int64_t)C;
  OutT* yo = y + (time_major ? (int64_t)b * C + ch : ((int64_t)b * T) * C + ch);
  const float* src = xs + c * kScanRS;
  for (int t = tq; t < T; t += 16) {
    const double v = ((double)src[(t >> 5) * 33 + (t & 31)] - mean) * inv;
    yo[(int64_t)t * t_stride] = from_f32<OutT>((float)v);
  }
  (void)meanf; (void)invf;
}

static ScanBasis* g_basis[16] = {nullptr};

template <int NSEC>
static int launch_scan(const float* x, void* y, int B, int C, int T, const SosParams& p, int ddof, int out_dtype,
                       int time_major, hipStream_t st) {
  int dev = 0;
  CSN_HIP_CHECK(hipGetDevice(&dev));
  CSN_REQUIRE(dev >= 0 && dev < 16, "device index %d out of range", dev);
  if (g_basis[dev] == nullptr) CSN_HIP_CHECK(hipMalloc((void**)&g_basis[dev], sizeof(ScanBasis)));
  eeg_filter_basis_kernel<NSEC><<<1, 64, 0, st>>>(p, g_basis[dev]);
  CSN_LAUNCH_CHECK();
  const int64_t rows = (int64_t)B * C;
  const unsigned grid = (unsigned)((rows + kScanRows - 1) / kScanRows);
  if (out_dtype == CSN_BF16)
    eeg_filter_scan_kernel<NSEC, bf16_t><<<grid, 256, 0, st>>>(x, (bf16_t*)y, B, C, T, p, g_basis[dev], ddof, time_major);
  else
    eeg_filter_scan_kernel<NSEC, float><<<grid, 256, 0, st>>>(x, (float*)y, B, C, T, p, g_basis[dev], ddof, time_major);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
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

// ---------------------------------------------------------------------------------------------
// zero-phase variant: forward-backward filtering with odd extension and steady-state initial
// conditions -- scipy.signal.filtfilt as applied by Utilities.remove_noise
// (/root/reference/utils/Utilities.py:411-428), evaluated on the biquad cascade in float64.
// Layout [S][T][C] (channels fastest, as remove_noise takes it): lanes = channels, so every load and
// store of a time step is a coalesced run.  The forward pass writes its float64 output to scratch
// [n][row]; the backward pass reads it in reverse, restarts the cascade at steady state for its first
// sample and writes the un-padded samples.
// ---------------------------------------------------------------------------------------------
struct ZiParams {
  double s1[8], s2[8];   // steady-state DF2T state of each section per unit of the CASCADE input
};

template <int NSEC>
__global__ void __launch_bounds__(64)
eeg_filtfilt_kernel(const float* __restrict__ x, float* __restrict__ y, double* __restrict__ scratch, int S, int T,
                    int C, int pad, SosParams p, ZiParams zi) {
  const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;      // (segment, channel)
  const int64_t rows = (int64_t)S * C;
  if (row >= rows) return;
  const int sgm = (int)(row / C), c = (int)(row % C);
  const float* xr = x + (int64_t)sgm * T * C + c;                  // element t at xr[t*C]
  float* yr = y + (int64_t)sgm * T * C + c;
  const int N = T + 2 * pad;
  auto ext = [&](int n) -> double {                                // odd extension about both ends
    if (n < pad) return 2.0 * (double)xr[0] - (double)xr[(int64_t)(pad - n) * C];
    if (n >= pad + T) return 2.0 * (double)xr[(int64_t)(T - 1) * C] - (double)xr[(int64_t)(2 * T + pad - 2 - n) * C];
    return (double)xr[(int64_t)(n - pad) * C];
  };
  double s1[8], s2[8];
  const double x0 = ext(0);
#pragma unroll
  for (int s = 0; s < 8; ++s) { s1[s] = zi.s1[s] * x0; s2[s] = zi.s2[s] * x0; }
  for (int n = 0; n < N; ++n) scratch[(int64_t)n * rows + row] = biquad_cascade<NSEC>(ext(n), p, s1, s2);
  const double y0 = scratch[(int64_t)(N - 1) * rows + row];
#pragma unroll
  for (int s = 0; s < 8; ++s) { s1[s] = zi.s1[s] * y0; s2[s] = zi.s2[s] * y0; }
  for (int n = N - 1; n >= 0; --n) {
    const double v = biquad_cascade<NSEC>(scratch[(int64_t)n * rows + row], p, s1, s2);
    if (n >= pad && n < pad + T) yr[(int64_t)(n - pad) * C] = (float)v;
  }
}

template <int NSEC>
static int launch_filtfilt(const float* x, float* y, double* scratch, int S, int T, int C, int pad, const SosParams& p,
                           const ZiParams& zi, hipStream_t st) {
  const int64_t rows = (int64_t)S * C;
  eeg_filtfilt_kernel<NSEC><<<(unsigned)((rows + 63) / 64), 64, 0, st>>>(x, y, scratch, S, T, C, pad, p, zi);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

static int fill_sos(const double* sos, int nsec, SosParams* p, const char* fn) {
  for (int s = 0; s < 8; ++s)
    for (int k = 0; k < 5; ++k) p->c[s][k] = 0.0;
  for (int s = 0; s < nsec; ++s) {
    const double a0 = sos[s * 6 + 3];
    CSN_REQUIRE(a0 != 0.0, "%s: section %d has a0 == 0", fn, s);
    p->c[s][0] = sos[s * 6 + 0] / a0;
    p->c[s][1] = sos[s * 6 + 1] / a0;
    p->c[s][2] = sos[s * 6 + 2] / a0;
    p->c[s][3] = sos[s * 6 + 4] / a0;
    p->c[s][4] = sos[s * 6 + 5] / a0;
  }
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
  if (T <= kScanChunks * kScanLen && nsec <= 5 && !options_from_env().filter_v1) {
    switch (nsec) {
      case 0: return launch_scan<0>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
      case 1: return launch_scan<1>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
      case 2: return launch_scan<2>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
      case 3: return launch_scan<3>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
      case 4: return launch_scan<4>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
      default: return launch_scan<5>(x, y, B, C, T, p, ddof, out_dtype, time_major, st);
    }
  }
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

extern "C" size_t csn_eeg_filtfilt_scratch_bytes(int S, int T, int C, int nsec) {
  if (S <= 0 || T <= 0 || C <= 0 || nsec <= 0) return 0;
  const int pad = 3 * (2 * nsec + 1);
  return (size_t)(T + 2 * pad) * (size_t)S * (size_t)C * sizeof(double);
}

extern "C" int csn_eeg_filtfilt(const float* x, int S, int T, int C, const double* sos, int nsec, float* y,
                                void* scratch, csnStream_t stream) {
  using namespace csn;
  CSN_REQUIRE(x && y && sos && scratch, "csn_eeg_filtfilt: null pointer");
  CSN_REQUIRE(nsec >= 1 && nsec <= 8, "csn_eeg_filtfilt: nsec=%d outside 1..8", nsec);
  const int pad = 3 * (2 * nsec + 1);        // scipy filtfilt default: 3 * max(len(a), len(b))
  CSN_REQUIRE(S > 0 && C > 0 && T > pad, "csn_eeg_filtfilt: T=%d must exceed padlen=%d", T, pad);
  SosParams p;
  if (int rc = fill_sos(sos, nsec, &p, "csn_eeg_filtfilt")) return rc;
  // steady state of each DF2T section for a constant cascade input of 1 (sosfilt_zi):
  // y = G u, s1 = (G - b0) u, s2 = (b2 - a2 G) u, and the next section sees u' = G u
  ZiParams zi;
  double u = 1.0;
  for (int s = 0; s < 8; ++s) zi.s1[s] = zi.s2[s] = 0.0;
  for (int s = 0; s < nsec; ++s) {
    const double b0 = p.c[s][0], b1 = p.c[s][1], b2 = p.c[s][2], a1 = p.c[s][3], a2 = p.c[s][4];
    const double G = (b0 + b1 + b2) / (1.0 + a1 + a2);
    zi.s1[s] = (G - b0) * u;
    zi.s2[s] = (b2 - a2 * G) * u;
    u *= G;
  }
  hipStream_t st = as_stream(stream);
  double* sc = (double*)scratch;
  switch (nsec) {
    case 1: return launch_filtfilt<1>(x, y, sc, S, T, C, pad, p, zi, st);
    case 2: return launch_filtfilt<2>(x, y, sc, S, T, C, pad, p, zi, st);
    case 3: return launch_filtfilt<3>(x, y, sc, S, T, C, pad, p, zi, st);
    case 4: return launch_filtfilt<4>(x, y, sc, S, T, C, pad, p, zi, st);
    case 5: return launch_filtfilt<5>(x, y, sc, S, T, C, pad, p, zi, st);
    case 6: return launch_filtfilt<6>(x, y, sc, S, T, C, pad, p, zi, st);
    case 7: return launch_filtfilt<7>(x, y, sc, S, T, C, pad, p, zi, st);
    default: return launch_filtfilt<8>(x, y, sc, S, T, C, pad, p, zi, st);
  }
}
