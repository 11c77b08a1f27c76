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
__host__ __device__ __forceinline__ double biquad_cascade(double v, const SosParams& p, double (&s1)[8], double (&s2)[8]) {
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
// v3 kernel ("scan"): the time recurrence is parallelised and everything between the load and the
// transposed store lives in registers, so the kernel runs at the speed of its loads and stores.
//   A workgroup (512 threads) takes 32 consecutive rows (channels); the 16 lanes of a DPP row are the
//   16 chunks of 32 samples of ONE row.  Thread (row, chunk):
//     1. loads its 32 samples (8 x 16 B; the four rows of a wave are one contiguous 8 KB run of x),
//     2. filters them from a ZERO state in float64 (the poles of this band sit at radius 0.9997:
//        float32 state costs 2e-4) -> zero-state response + the chunk's end state e_k,
//     3. the cascade is linear, so the true state at the end of chunk k is S_k = A S_{k-1} + e_k with
//        A the 32-step state transition: a Kogge-Stone scan over the 16 lanes in 4 steps,
//        S_k += A^(2^m) S_{k-2^m}, the exchange by DPP row shifts (no LDS, no barrier),
//     4. adds the homogeneous response  y[n] += sum_i S_{k-1,i} phi_i[n]  (phi_i = response to a unit
//        state component) and accumulates the row statistics (DPP rotations over the 16 lanes),
//     5. normalises and hands the tile to LDS once, for the channel-fastest store: 16 B per lane,
//        8 lanes = the 32 channels of one time step = one full 128-byte line.
//   phi and the four powers of A depend on the coefficients alone: computed on the HOST per call
//   (192 cascade steps) and passed by value with the kernel arguments -- no device buffer, no setup
//   launch, nothing shared between calls (the previous version kept a per-device global and re-ran a
//   setup kernel on the caller's stream every call).
// LDS: tile[t][32 channels] f32, 64 KB, column of (row r, chunk k) rotated by 4 (k & 7): the 32 lanes
// of a store group (2 rows x 16 chunks) then hit every bank at most twice (free for ds_write_b32) and a
// 16-lane group of the 16-byte reads covers all 64 banks once.
// ---------------------------------------------------------------------------------------------
static constexpr int kScanRows = 32, kScanChunks = 16, kScanLen = 32;

template <int NSEC>
struct ScanBasis {
  static constexpr int NS = NSEC > 0 ? 2 * NSEC : 1;
  double phi[NS][kScanLen];    // [state component][n]
  double apow[4][NS][NS];      // apow[m][j][i] = component j of the state 32 * 2^m steps after e_i
};

template <int NSEC>
static void fill_scan_basis(const SosParams& p, ScanBasis<NSEC>* b) {
  constexpr int NS = ScanBasis<NSEC>::NS;
  for (int i = 0; i < NS; ++i) {
    double s1[8] = {0.0}, s2[8] = {0.0};
    for (int s = 0; s < NSEC; ++s) {
      if (i == 2 * s) s1[s] = 1.0;
      if (i == 2 * s + 1) s2[s] = 1.0;
    }
    for (int n = 0; n < kScanLen; ++n) b->phi[i][n] = NSEC > 0 ? biquad_cascade<NSEC>(0.0, p, s1, s2) : 0.0;
    for (int j = 0; j < NS; ++j) b->apow[0][j][i] = 0.0;
    for (int s = 0; s < NSEC; ++s) {
      b->apow[0][2 * s][i] = s1[s];
      b->apow[0][2 * s + 1][i] = s2[s];
    }
  }
  for (int m = 1; m < 4; ++m)
    for (int j = 0; j < NS; ++j)
      for (int i = 0; i < NS; ++i) {
        double a = 0.0;
        for (int l = 0; l < NS; ++l) a += b->apow[m - 1][j][l] * b->apow[m - 1][l][i];
        b->apow[m][j][i] = a;
      }
}

// lane l of a 16-lane DPP row <- lane l - N (zero where there is none) / lane (l - N) mod 16
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
#define CSN_DPP_ROW_SHR(n) (0x110 | (n))
#define CSN_DPP_ROW_ROR(n) (0x120 | (n))

template <int NSEC, int M>
__device__ __forceinline__ void scan_step(double (&sv)[ScanBasis<NSEC>::NS], const ScanBasis<NSEC>& bs) {
  constexpr int NS = ScanBasis<NSEC>::NS;
  double sh[NS];
#pragma unroll
  for (int i = 0; i < NS; ++i) sh[i] = dpp_f64<CSN_DPP_ROW_SHR(1 << M)>(sv[i]);
#pragma unroll
  for (int j = 0; j < NS; ++j)
#pragma unroll
    for (int i = 0; i < NS; ++i) sv[j] = fma(bs.apow[M][j][i], sh[i], sv[j]);
}

__device__ __forceinline__ double row16_sum(double v) {
  v += dpp_f64<CSN_DPP_ROW_ROR(8)>(v);
  v += dpp_f64<CSN_DPP_ROW_ROR(4)>(v);
  v += dpp_f64<CSN_DPP_ROW_ROR(2)>(v);
  v += dpp_f64<CSN_DPP_ROW_ROR(1)>(v);
  return v;
}

template <typename OutT>
__device__ __forceinline__ void store_quad(OutT* dst, const float4& v);
template <>
__device__ __forceinline__ void store_quad<float>(float* dst, const float4& v) { *reinterpret_cast<float4*>(dst) = v; }
template <>
__device__ __forceinline__ void store_quad<bf16_t>(bf16_t* dst, const float4& v) {
  bf16x4 o;
  o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
  *reinterpret_cast<bf16x4*>(dst) = o;
}

template <int NSEC, typename OutT>
__global__ void __launch_bounds__(512, 4)
eeg_filter_scan_kernel(const float* __restrict__ x, OutT* __restrict__ y, int B, int C, int T, const SosParams p,
                       const ScanBasis<NSEC> bs, int ddof, int time_major) {
  constexpr int NS = ScanBasis<NSEC>::NS;
  __shared__ __attribute__((aligned(16))) float tile[kScanChunks * kScanLen * kScanRows];     // [t][channel], 64 KB
  const int tid = threadIdx.x;
  const int k = tid & 15, rl = tid >> 4;
  const int64_t rows_total = (int64_t)B * C;
  const int64_t row0 = (int64_t)blockIdx.x * kScanRows;
  const int64_t row = row0 + rl;
  const bool rok = row < rows_total;
  const int t0 = k * kScanLen;

  // ---- 1: this thread's 32 samples (zeros beyond T / beyond the last row)
  double v[kScanLen];
  {
    const float* src = x + row * (int64_t)T + t0;
    if ((T & 3) == 0) {
#pragma unroll
      for (int i = 0; i < kScanLen / 4; ++i) {
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rok && t0 + 4 * i < T) q = *reinterpret_cast<const float4*>(src + 4 * i);
        v[4 * i + 0] = (double)q.x; v[4 * i + 1] = (double)q.y; v[4 * i + 2] = (double)q.z; v[4 * i + 3] = (double)q.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < kScanLen; ++j) v[j] = (rok && t0 + j < T) ? (double)src[j] : 0.0;
    }
  }

  if constexpr (NSEC > 0) {
    // ---- 2: zero-state response in place, end state
    double sv[NS];
    {
      double s1[8], s2[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) { s1[s] = 0.0; s2[s] = 0.0; }
#pragma unroll
      for (int j = 0; j < kScanLen; ++j) v[j] = biquad_cascade<NSEC>(v[j], p, s1, s2);
#pragma unroll
      for (int s = 0; s < NSEC; ++s) { sv[2 * s] = s1[s]; sv[2 * s + 1] = s2[s]; }
    }
    // ---- 3: inclusive scan of the end states over the row's 16 chunks; this chunk starts from S_{k-1}
    scan_step<NSEC, 0>(sv, bs);
    scan_step<NSEC, 1>(sv, bs);
    scan_step<NSEC, 2>(sv, bs);
    scan_step<NSEC, 3>(sv, bs);
    double s0[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) s0[i] = dpp_f64<CSN_DPP_ROW_SHR(1)>(sv[i]);
    // ---- 4a: homogeneous response
#pragma unroll
    for (int j = 0; j < kScanLen; ++j)
#pragma unroll
      for (int i = 0; i < NS; ++i) v[j] = fma(s0[i], bs.phi[i][j], v[j]);
  }

  // ---- 4b: row statistics (samples beyond T do not count)
  double sum = 0.0, sumsq = 0.0;
#pragma unroll
  for (int j = 0; j < kScanLen; ++j) {
    const double m = (t0 + j < T) ? v[j] : 0.0;
    sum += m;
    sumsq = fma(m, m, sumsq);
  }
  sum = row16_sum(sum);
  sumsq = row16_sum(sumsq);
  const double mean = sum / (double)T;
  const double inv = 1.0 / sqrt((sumsq - sum * mean) / (double)(T - ddof));

  // ---- 5: normalise, transpose through LDS, store channel-fastest
  {
    const int col = (rl + 4 * (k & 7)) & 31;
#pragma unroll
    for (int j = 0; j < kScanLen; ++j) tile[(t0 + j) * kScanRows + col] = (float)((v[j] - mean) * inv);
  }
  __syncthreads();
  const bool quad_ok = (C & 3) == 0;            // the 4 rows of a quad are 4 consecutive channels of one segment
#pragma unroll
  for (int it = 0; it < kScanChunks * kScanLen * 8 / 512; ++it) {
    const int idx = it * 512 + tid;
    const int q = idx & 7, t = idx >> 3;
    if (t >= T) continue;
    const float4 o = *reinterpret_cast<const float4*>(tile + t * kScanRows + ((4 * q + 4 * ((t >> 5) & 7)) & 31));
    const int64_t r4 = row0 + 4 * q;
    if (quad_ok) {
      if (r4 >= rows_total) continue;
      const int b = (int)(r4 / C), ch = (int)(r4 % C);
      OutT* dst = y + (time_major ? ((int64_t)t * B + b) * C + ch : ((int64_t)b * T + t) * C + ch);
      store_quad<OutT>(dst, o);
    } else {
      const float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (r4 + e >= rows_total) continue;
        const int b = (int)((r4 + e) / C), ch = (int)((r4 + e) % C);
        y[time_major ? ((int64_t)t * B + b) * C + ch : ((int64_t)b * T + t) * C + ch] = from_f32<OutT>(ov[e]);
      }
    }
  }
}

template <int NSEC>
static int launch_scan(const float* x, void* y, int B, int C, int T, const SosParams& p, int ddof, int out_dtype,
                       int time_major, hipStream_t st) {
  ScanBasis<NSEC> bs;
  fill_scan_basis<NSEC>(p, &bs);
  const int64_t rows = (int64_t)B * C;
  const unsigned grid = (unsigned)((rows + kScanRows - 1) / kScanRows);
  if (out_dtype == CSN_BF16)
    eeg_filter_scan_kernel<NSEC, bf16_t><<<grid, 512, 0, st>>>(x, (bf16_t*)y, B, C, T, p, bs, ddof, time_major);
  else
    eeg_filter_scan_kernel<NSEC, float><<<grid, 512, 0, st>>>(x, (float*)y, B, C, T, p, bs, ddof, time_major);
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
