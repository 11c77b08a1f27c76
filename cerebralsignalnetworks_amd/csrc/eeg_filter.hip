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

template <int NSEC, typename Params>      // Params: SosParams, in whatever address space it lives (kernel-argument segment in the scan kernel)
__host__ __device__ __forceinline__ double biquad_cascade(double v, const Params& p, double (&s1)[8], double (&s2)[8]) {
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
//     1. gets its 32 samples: the four rows of a wave are one contiguous run of x, loaded 1 KB per instruction and
//        passed through the wave's own corner of LDS (a lane-strided direct load thrashed L1),
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
  double phi[kScanLen][NS];    // [n][state component]: read in this order by the kernel (sequential scalar loads)
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
    for (int n = 0; n < kScanLen; ++n) b->phi[n][i] = NSEC > 0 ? biquad_cascade<NSEC>(0.0, p, s1, s2) : 0.0;
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

template <int NSEC, int M, typename Basis>
__device__ __forceinline__ void scan_step(double (&sv)[ScanBasis<NSEC>::NS], const Basis& bs) {
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

// input staging: [row][chunk] blocks of 32 samples, block pitch 36 dwords -- the 16 lanes of a ds_read_b128 group (16
// different chunk indices, whatever their rows: the row pitch is 0 mod 64 banks) land on 16 different 4-bank slots
static constexpr int kScanPitch = 36;
static constexpr int kScanLdsBytes = kScanRows * kScanChunks * kScanPitch * 4;      // 73 728 B (>= the 64 KB output tile)

// ALL kernel arguments in one struct = the kernarg segment from offset 0: the tile loop re-reads the coefficients through
// a laundered pointer to that segment, so the compiler cannot hoist the ~340 coefficient loads of the basis out of the
// loop (it did: 800 - 1600 SGPRs spilled to VGPR lanes, those spilled to scratch).
template <int NSEC>
struct ScanArgs {
  const float* x;
  void* y;
  int B, C, T, ddof, time_major, pad_;
  SosParams p;
  ScanBasis<NSEC> bs;
};
template <int NSEC>
using ScanArgsK = const __attribute__((address_space(4))) ScanArgs<NSEC>;

template <int NSEC, typename OutT>
__global__ void __launch_bounds__(512, 4) eeg_filter_scan_kernel(const ScanArgs<NSEC> args) {
  constexpr int NS = ScanBasis<NSEC>::NS;
  extern __shared__ __attribute__((aligned(16))) float tile[];     // input staging, then [t][channel] (64 KB) for the store
  const float* __restrict__ x = args.x;
  OutT* __restrict__ y = reinterpret_cast<OutT*>(args.y);
  const int B = args.B, C = args.C, T = args.T, ddof = args.ddof, time_major = args.time_major;
  const int tid = threadIdx.x;
  const int k = tid & 15, rl = tid >> 4;
  const int64_t rows_total = (int64_t)B * C;
  // Local time u = t + pad, pad = 512 - T: the row is RIGHT-aligned in its 16 chunks, the missing samples are leading
  // zeros.  Zero input from zero state gives zero output and zero state, so neither the cascade nor the statistics
  // need a mask (with the row left-aligned, samples past T had to be masked out of both: a compare and two selects per
  // sample), and only the store looks at u >= pad.
  const int u0 = k * kScanLen;
  const int lane = tid & 63, wrow = (tid >> 6) * 4;               // first of this wave's rows inside the workgroup
  const int q4 = T >> 2;                                          // float4 per row
  const int pad = kScanChunks * kScanLen - T, pq = pad >> 2;      // leading zeros per row (floats, float4s)
  const int ntiles = (int)((rows_total + kScanRows - 1) / kScanRows);

  // ---- 1a: a wave's four rows are one contiguous run of 4 T floats (T % 4 == 0): loaded 1 KB per instruction.
  // (Loading a thread's 32 samples straight into its registers -- 64 lanes x 16 B at a 128-byte stride, eight
  // instructions over the same 64 lines -- thrashed L1: 8 x the L2 -> L1 traffic, 56 us per 256 segments.)
  // Where float4 `idx` of the wave's run goes in the staging area (tile-invariant: kept).  The 512 float4 slots of the
  // eight load instructions are exactly the wave's 4 rows x 16 chunks x 8 float4: the 4 q4 slots of the run carry data,
  // the 4 pq slots behind it write the rows' leading zeros -- every position is written for every tile.
  int sa[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int idx = lane + 64 * j;
    const bool data = idx < 4 * q4;
    const int z = data ? idx : idx - 4 * q4;
    const int per = data ? q4 : pq;
    const int r = (z >= per) + (z >= 2 * per) + (z >= 3 * per);
    const int u = 4 * (z - r * per) + (data ? pad : 0);
    sa[j] = ((wrow + r) * kScanChunks + (u >> 5)) * kScanPitch + (u & 31);
  }
  f32x4 g[8];                 // (native vectors: an array of HIP's float4 structs was kept in scratch memory)
  auto request = [&](int tl) {
    // C % 4 == 0 (the launcher sends everything else to the row-walking kernel): a wave's four rows exist or none does;
    // a wave beyond the last row loads rows 0..3 again -- nothing of it is stored
    const int64_t r0 = (int64_t)tl * kScanRows + wrow;
    const f32x4* src = reinterpret_cast<const f32x4*>(x + (r0 < rows_total ? r0 : 0) * (int64_t)T);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int idx = lane + 64 * j;
      const f32x4 got = src[idx < 4 * q4 ? idx : 4 * q4 - 1];
      g[j] = idx < 4 * q4 ? got : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };
  // the next tile's loads stay in flight across a tile only where the 32 registers fit (3 sections: the reference's
  // training filter); the other section counts load at the top of each tile -- no instantiation may spill
  constexpr bool PREFETCH = NSEC >= 1 && NSEC <= 3;
  if constexpr (PREFETCH) request(blockIdx.x);
  // store phase: this thread writes channel quad q of time steps (tid >> 3) + 64 it
  const int sq = tid & 7, st0 = tid >> 3;
  const int64_t tstride = time_major ? (int64_t)B * C : (int64_t)C;

  // The workgroup walks tiles blockIdx.x, + gridDim.x, ...: the NEXT tile's loads are in flight while this one is
  // filtered, and this one's stores drain under the next one's arithmetic.
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int64_t row0 = (int64_t)tl * kScanRows;
    ScanArgsK<NSEC>* ka = (ScanArgsK<NSEC>*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));            // opaque per tile: coefficient loads stay inside the loop, next to their uses
    const auto& p = ka->p;
    const auto& bs = ka->bs;
    // ---- 1b: through the wave's own part of LDS (no workgroup barrier: written and read by this wave only) into this
    // thread's 32 samples (float32 between the phases: 32 registers; every phase computes in float64)
    float v[kScanLen];
    if constexpr (!PREFETCH) request(tl);
#pragma unroll
    for (int j = 0; j < 8; ++j) *reinterpret_cast<f32x4*>(__builtin_assume_aligned(tile + sa[j], 16)) = g[j];
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (PREFETCH) {
      if (tl + (int)gridDim.x < ntiles) request(tl + gridDim.x);
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_wave_barrier();
    {
      const float* mine = reinterpret_cast<const float*>(__builtin_assume_aligned(tile + (rl * kScanChunks + k) * kScanPitch, 16));
#pragma unroll
      for (int i = 0; i < kScanLen / 4; ++i) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(mine + 4 * i);
        v[4 * i + 0] = q[0]; v[4 * i + 1] = q[1]; v[4 * i + 2] = q[2]; v[4 * i + 3] = q[3];
      }
    }

    double sum = 0.0, sumsq = 0.0;
    if constexpr (NSEC > 0) {
      // ---- 2: zero-state response in place, end state
      double sv[NS];
      {
        double s1[8], s2[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) { s1[s] = 0.0; s2[s] = 0.0; }
#pragma unroll
        for (int j = 0; j < kScanLen; ++j) {
          v[j] = (float)biquad_cascade<NSEC>((double)v[j], p, s1, s2);
          if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (bounds what the scheduler keeps in flight: 128 registers)
        }
#pragma unroll
        for (int s = 0; s < NSEC; ++s) { sv[2 * s] = s1[s]; sv[2 * s + 1] = s2[s]; }
      }
      // ---- 3: inclusive scan of the end states over the row's 16 chunks; this chunk starts from S_{k-1}
      // (scheduling fences: left alone, the compiler requests the coefficients of all steps / samples at once and
      // spills 100 - 200 SGPRs to VGPR lanes, each use then a v_readlane)
      __builtin_amdgcn_sched_barrier(0);
      scan_step<NSEC, 0>(sv, bs);
      __builtin_amdgcn_sched_barrier(0);
      scan_step<NSEC, 1>(sv, bs);
      __builtin_amdgcn_sched_barrier(0);
      scan_step<NSEC, 2>(sv, bs);
      __builtin_amdgcn_sched_barrier(0);
      scan_step<NSEC, 3>(sv, bs);
      __builtin_amdgcn_sched_barrier(0);
      double s0[NS];
#pragma unroll
      for (int i = 0; i < NS; ++i) s0[i] = dpp_f64<CSN_DPP_ROW_SHR(1)>(sv[i]);
      // ---- 4: homogeneous response, row statistics (the leading zeros stay zeros: S_{k-1} is zero there)
#pragma unroll
      for (int j = 0; j < kScanLen; ++j) {
        double c = (double)v[j];
#pragma unroll
        for (int i = 0; i < NS; ++i) c = fma(s0[i], bs.phi[j][i], c);
        v[j] = (float)c;
        sum += c;
        sumsq = fma(c, c, sumsq);
        if ((j & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < kScanLen; ++j) {
        const double m = (double)v[j];
        sum += m;
        sumsq = fma(m, m, sumsq);
        if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    }
    sum = row16_sum(sum);
    sumsq = row16_sum(sumsq);
    const double mean = sum / (double)T;
    const double inv = 1.0 / sqrt((sumsq - sum * mean) / (double)(T - ddof));

    // ---- 5: normalise, transpose through LDS, store channel-fastest
    __syncthreads();          // every wave has read its staged input: the tile is reused for the output
    {
      float* dst = tile + u0 * kScanRows + ((rl + 4 * (k & 7)) & 31);
#pragma unroll
      for (int j = 0; j < kScanLen; ++j) {
        dst[j * kScanRows] = (float)(((double)v[j] - mean) * inv);
        if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();
    {
      const int64_t r4 = row0 + 4 * sq;                             // 4 consecutive channels of one segment (C % 4 == 0)
      if (r4 < rows_total) {
        const unsigned bq = (unsigned)(r4 / (unsigned)C);           // (rows_total < 2^32: B, C are ints and the tile fits HBM)
        const unsigned ch = (unsigned)r4 - bq * (unsigned)C;
        OutT* dst = y + (time_major ? (int64_t)bq * C + ch : (int64_t)bq * T * C + ch) + (st0 - pad) * tstride;
        const float* src = tile + st0 * kScanRows;
#pragma unroll
        for (int it = 0; it < kScanChunks * kScanLen / 64; ++it) {
          const int u = st0 + 64 * it;                               // local time; chunk u >> 5 = 2 it + (st0 >> 5)
          if (u >= pad)
            store_quad<OutT>(dst + (int64_t)(64 * it) * tstride,
                             *reinterpret_cast<const float4*>(__builtin_assume_aligned(
                                 src + 64 * it * kScanRows + ((4 * sq + 4 * ((u >> 5) & 7)) & 31), 16)));
        }
      }
    }
    __syncthreads();          // the output tile has been read: the next tile's staging may overwrite it
  }
}

template <int NSEC>
static int launch_scan(const float* x, void* y, int B, int C, int T, const SosParams& p, int ddof, int out_dtype,
                       int time_major, hipStream_t st) {
  ScanArgs<NSEC> ka;
  ka.x = x; ka.y = y; ka.B = B; ka.C = C; ka.T = T; ka.ddof = ddof; ka.time_major = time_major; ka.pad_ = 0;
  ka.p = p;
  fill_scan_basis<NSEC>(p, &ka.bs);
  const int64_t rows = (int64_t)B * C;
  const unsigned ntiles = (unsigned)((rows + kScanRows - 1) / kScanRows);
  // two workgroups per CU are resident (LDS); each walks its share of the tiles with the next one prefetched
  int dev = 0, cus = 0;
  CSN_HIP_CHECK(hipGetDevice(&dev));
  CSN_HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
  const unsigned slots = (unsigned)(2 * (cus > 0 ? cus : 256));
  const unsigned per_wg = (ntiles + slots - 1) / slots;                 // tiles per workgroup
  const unsigned grid = (ntiles + per_wg - 1) / per_wg;                 // evenly loaded workgroups, <= slots
  if (int rc = ensure_dyn_lds<&eeg_filter_scan_kernel<NSEC, bf16_t>>(kScanLdsBytes)) return rc;
  if (int rc = ensure_dyn_lds<&eeg_filter_scan_kernel<NSEC, float>>(kScanLdsBytes)) return rc;
  if (out_dtype == CSN_BF16) eeg_filter_scan_kernel<NSEC, bf16_t><<<grid, 512, kScanLdsBytes, st>>>(ka);
  else eeg_filter_scan_kernel<NSEC, float><<<grid, 512, kScanLdsBytes, st>>>(ka);
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
  if (T <= kScanChunks * kScanLen && (T & 3) == 0 && (C & 3) == 0 && nsec <= 5 && !options_from_env().filter_v1) {
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
