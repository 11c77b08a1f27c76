// K3 fast path (bf16): per-timestep LSTM cell kernels on fragment-major, gate-interleaved
// operands; one launch advances up to 4 independent cell problems (layers of a wavefront).
// Replaces the per-step body of nn.LSTM reached at /root/reference/LSTMDistill.py:118,132.
//
// What bounds a step (measured on MI355X, profiles/ and tools/stamp_cells.hip): the recurrent
// product of one timestep is tiny for the chip (1.2 GFLOP) but forces a launch per step; inside a
// launch the operand loads dominate and run at the per-CU L2->CU rate (~60-70 GB/s per CU), and a
// launch boundary costs ~2 us.  Hence:
//   * operands of the recurrent product (h_{t-1}, W_hh; dgates_{t+1}, W_hh^T) live in the
//     fragment-major layout of lstm_cell_common.h: a fragment load is one contiguous 1 KB read
//     straight to VGPRs (row-major fragments touched 16 half-used lines per instruction and cost
//     5.7 / 11.5 us per launch);
//   * gate rows are interleaved, n' = 4*unit + gate, everywhere inside the fast path, so one
//     16-row MFMA tile is 4 units x (i,f,g,o): tiles of any multiple of 4 units, and all per-cell
//     traffic of the epilogue (xproj, saved gates, dgates) is one contiguous 32/64-byte run;
//   * K is split over the 4 waves of a workgroup: every operand byte is loaded once per
//     workgroup, all loads of a pass are issued back to back (register double-buffering), the
//     partial tiles are summed through LDS and re-read in a (row, 4 units)-per-thread mapping;
//   * the epilogue's own inputs are requested BEFORE the K loop (their HBM latency hides under it);
//   * blockIdx.z selects one of several independent problems (layer l at step t, layer l+1 at
//     step t - lag, ...): a launch boundary and the per-launch ramp are paid once per diagonal,
//     and tiles are sized so that a 2-layer diagonal is 256 workgroups = one per CU
//     (forward 64 rows x 24 units, 246 KB of operands; backward 32 rows x 48 units, 492 KB).
// The epilogue writes h_t / dgates_t twice: row-major (read by the big GEMMs / the caller) and
// fragment-major (read by the next step; 2-slot ping-pong per layer).
#include "csn_common.h"
#include "lstm_cell_common.h"
#include "lstm_cell_blk.h"

#ifdef CSN_STAMPS
// Diagnostic build only (tools/stamp_cells.hip): wall-clock stamps (100 MHz) of thread 0 of every
// workgroup go to a buffer of their own; no output value depends on them.
__device__ unsigned long long* g_stamps = nullptr;
#define CSN_STAMP(i)                                                                      \
  do {                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                    \
    if (threadIdx.x == 0)                                                                 \
      g_stamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); \
    __builtin_amdgcn_sched_barrier(0);                                                    \
  } while (0)
#else
#define CSN_STAMP(i)
#endif

namespace csn {

// ------------------------------------------------------------------------------------------
// layout preparation (weights change every optimiser step, so these run once per forward)
// ------------------------------------------------------------------------------------------
// standard gate-major row g*H + u of a [4H, *] parameter <-> interleaved row n' = 4u + g
__device__ __forceinline__ int64_t std_row(int64_t nprime, int64_t H) { return (nprime & 3) * H + (nprime >> 2); }

// dst = fragment-major bf16 [R, K]; element (r, k) = src[sr(r) * ld_r + sk(k) * ld_k] with optional
// interleave permutation on the row and/or the k index.
__global__ void __launch_bounds__(256)
blockify_cast_kernel(const float* __restrict__ src, int64_t ld_r, int64_t ld_k, int64_t R, int64_t K, int perm_r,
                     int perm_k, int64_t H, bf16_t* __restrict__ dst) {
  const int64_t nchunks = R * K / 8;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ci < nchunks; ci += stride) {
    const int64_t blk = ci >> 6, lane = ci & 63;
    const int64_t kblocks = K >> 5;
    const int64_t r = (blk / kblocks) * 16 + (lane & 15);
    const int64_t k = (blk % kblocks) * 32 + 8 * (lane >> 4);
    const int64_t rs = perm_r ? std_row(r, H) : r;
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t ks = perm_k ? std_row(k + j, H) : (k + j);
      v[j] = (bf16_t)src[rs * ld_r + ks * ld_k];
    }
    *reinterpret_cast<bf16x8*>(dst + ci * 8) = v;
  }
}

// dst[n'][i] = (bf16) src[std_row(n')][i]            (W_ih with interleaved rows, row-major)
__global__ void permute_rows_cast_kernel(const float* __restrict__ src, int64_t H, int64_t I, bf16_t* __restrict__ dst) {
  const int64_t total = 4 * H * I;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t n = i / I, c = i % I;
    dst[i] = (bf16_t)src[std_row(n, H) * I + c];
  }
}

// dst[i][n'] = (bf16) src[std_row(n')][i]            (W_ih^T with interleaved columns)
__global__ void transpose_perm_cast_kernel(const float* __restrict__ src, int64_t H, int64_t I, bf16_t* __restrict__ dst) {
  const int64_t G = 4 * H, total = G * I;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t c = i / G, n = i % G;
    dst[i] = (bf16_t)src[std_row(n, H) * I + c];
  }
}

// dst[n'] = a[std_row(n')] + b[std_row(n')]
__global__ void bias_perm_sum_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t H,
                                     float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 4 * H) dst[i] = a[std_row(i, H)] + b[std_row(i, H)];
}

// out[std_row(n')][c] = sum_s slabs[s][n'][c]        (un-permute weight / bias gradients)
__global__ void reduce_slabs_unperm_kernel(const float* __restrict__ slabs, int64_t slab_stride, int S, int64_t H,
                                           int64_t C, float* __restrict__ out) {
  const int64_t total = 4 * H * C;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t n = i / C, c = i % C;
    float acc = 0.0f;
    for (int s = 0; s < S; ++s) acc += slabs[(int64_t)s * slab_stride + i];
    out[std_row(n, H) * C + c] = acc;
  }
}

// ---- all layout preparation of a forward in ONE launch ---------------------------------------------------------
// A training step re-prepares the operands of every layer (the weights change every step): per layer W_ih with
// interleaved rows, W_hh fragment-major, the summed bias, and for training their transposes; the input in both of its
// layouts.  As separate kernels that was 13 launches of 5-40 us with a dependent-launch gap behind each (0.15 ms of an
// 11 ms step); here every job gets a slice of one grid.
__global__ void __launch_bounds__(256) prep_multi_kernel(PrepArgs A) {
  // (the two TRANSPOSING jobs -- W_ih^T and the fragment-major W_hh^T of the backward -- read down columns of a row-major
  // float32 matrix: as 4-byte gathers they set the length of this launch (58 us); through a 64 x 64 LDS tile both sides
  // of the transpose move whole 64 / 256-byte runs)
  __shared__ __attribute__((aligned(16))) bf16_t tp[64][72];
  int j = 0;
  while (j + 1 < A.njobs && blockIdx.x >= A.job[j + 1].blk_begin) ++j;
  const PrepJob& J = A.job[j];
  const int64_t vb = blockIdx.x - J.blk_begin, vg = J.blk_count;
  const int64_t gid = vb * blockDim.x + threadIdx.x, stride = vg * blockDim.x;
  switch (J.kind) {
    case kPrepBlockify: {                 // fragment-major bf16 image of a (permuted) float32 matrix
      const int64_t R = J.n0, K = J.n1, H = J.H, ld_r = J.s0, ld_k = J.s1, kblocks = K >> 5;
      bf16_t* dst = (bf16_t*)J.dst;
      if (J.perm_k && !J.perm_r && ld_r == 1 && (R & 63) == 0 && (K & 63) == 0 && (ld_k & 3) == 0) {
        // element (u, k') = a[std_row(k') * ld_k + u]: tile = 64 units x 64 k'; source rows are contiguous along u
        const int64_t tiles_u = R >> 6, ntiles = tiles_u * (K >> 6);
        const int t = threadIdx.x, rr = t >> 2, seg = t & 3;
        for (int64_t tile = vb; tile < ntiles; tile += vg) {
          const int64_t tu = tile % tiles_u, tk = tile / tiles_u;
          const float4* sp = reinterpret_cast<const float4*>(J.a + std_row(tk * 64 + rr, H) * ld_k + tu * 64 + seg * 16);
          const float4 q0 = sp[0], q1 = sp[1], q2 = sp[2], q3 = sp[3];
          *reinterpret_cast<bf16x8*>(&tp[rr][seg * 16]) = (bf16x8){(bf16_t)q0.x, (bf16_t)q0.y, (bf16_t)q0.z, (bf16_t)q0.w, (bf16_t)q1.x, (bf16_t)q1.y, (bf16_t)q1.z, (bf16_t)q1.w};
          *reinterpret_cast<bf16x8*>(&tp[rr][seg * 16 + 8]) = (bf16x8){(bf16_t)q2.x, (bf16_t)q2.y, (bf16_t)q2.z, (bf16_t)q2.w, (bf16_t)q3.x, (bf16_t)q3.y, (bf16_t)q3.z, (bf16_t)q3.w};
          __syncthreads();
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c = t + 256 * h, fb = c >> 6, lane = c & 63, rb = fb >> 1, kb = fb & 1;
            const int ul = rb * 16 + (lane & 15), kl = kb * 32 + 8 * (lane >> 4);
            bf16x8 v;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tp[kl + e][ul];
            const int64_t blk = (tu * 4 + rb) * kblocks + tk * 2 + kb;
            *reinterpret_cast<bf16x8*>(dst + (blk * 64 + lane) * 8) = v;
          }
          __syncthreads();
        }
        break;
      }
      for (int64_t ci = gid; ci < R * K / 8; ci += stride) {
        const int64_t blk = ci >> 6, lane = ci & 63;
        const int64_t r = (blk / kblocks) * 16 + (lane & 15), k = (blk % kblocks) * 32 + 8 * (lane >> 4);
        const int64_t rs = J.perm_r ? std_row(r, H) : r;
        bf16x8 v;
        if (!J.perm_k && ld_k == 1 && (ld_r & 3) == 0) {        // the 8 elements are contiguous in the source: two 16-byte loads
          const float4* sp = reinterpret_cast<const float4*>(J.a + rs * ld_r + k);
          const float4 u = sp[0], w = sp[1];
          v = (bf16x8){(bf16_t)u.x, (bf16_t)u.y, (bf16_t)u.z, (bf16_t)u.w, (bf16_t)w.x, (bf16_t)w.y, (bf16_t)w.z, (bf16_t)w.w};
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int64_t ks = J.perm_k ? std_row(k + e, H) : (k + e);
            v[e] = (bf16_t)J.a[rs * ld_r + ks * ld_k];
          }
        }
        *reinterpret_cast<bf16x8*>(dst + ci * 8) = v;
      }
      break;
    }
    case kPrepPermRows: {                 // dst[n'][i] = src[std_row(n')][i]
      const int64_t H = J.H, I = J.n1;
      bf16_t* dst = (bf16_t*)J.dst;
      if ((I & 7) == 0) {                   // 8 elements per thread and trip: two 16-byte loads, one 16-byte store
        for (int64_t c = gid; c < 4 * H * I / 8; c += stride) {
          const int64_t i = c * 8;
          const float4* sp = reinterpret_cast<const float4*>(J.a + std_row(i / I, H) * I + i % I);
          const float4 u = sp[0], v = sp[1];
          *reinterpret_cast<bf16x8*>(dst + i) = (bf16x8){(bf16_t)u.x, (bf16_t)u.y, (bf16_t)u.z, (bf16_t)u.w,
                                                         (bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
        }
      } else {
        for (int64_t i = gid; i < 4 * H * I; i += stride) dst[i] = (bf16_t)J.a[std_row(i / I, H) * I + i % I];
      }
      break;
    }
    case kPrepTransPerm: {                // dst[i][n'] = src[std_row(n')][i]
      const int64_t H = J.H, I = J.n1, G = 4 * H;
      bf16_t* dst = (bf16_t*)J.dst;
      if ((I & 63) == 0 && (G & 63) == 0) {
        const int64_t tiles_n = G >> 6, ntiles = tiles_n * (I >> 6);
        const int t = threadIdx.x, rr = t >> 2, seg = t & 3;
        for (int64_t tile = vb; tile < ntiles; tile += vg) {
          const int64_t tn = tile % tiles_n, ti = tile / tiles_n;
          const float4* sp = reinterpret_cast<const float4*>(J.a + std_row(tn * 64 + rr, H) * I + ti * 64 + seg * 16);
          const float4 q0 = sp[0], q1 = sp[1], q2 = sp[2], q3 = sp[3];
          const float v[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
          for (int e = 0; e < 16; ++e) tp[seg * 16 + e][rr] = (bf16_t)v[e];      // [i][n']
          __syncthreads();
          bf16_t* dp = dst + (ti * 64 + rr) * G + tn * 64 + seg * 16;              // rr = i here: 16 consecutive n'
          *reinterpret_cast<bf16x8*>(dp) = *reinterpret_cast<const bf16x8*>(&tp[rr][seg * 16]);
          *reinterpret_cast<bf16x8*>(dp + 8) = *reinterpret_cast<const bf16x8*>(&tp[rr][seg * 16 + 8]);
          __syncthreads();
        }
        break;
      }
      for (int64_t i = gid; i < G * I; i += stride) dst[i] = (bf16_t)J.a[std_row(i % G, H) * I + i / G];
      break;
    }
    case kPrepBias: {                     // dst[n'] = a[std_row(n')] + b[std_row(n')]
      float* dst = (float*)J.dst;
      for (int64_t i = gid; i < 4 * J.H; i += stride) dst[i] = J.a[std_row(i, J.H)] + J.b[std_row(i, J.H)];
      break;
    }
    case kPrepCastX: {                    // x[b][t][i] (strides s0, s1) -> time-major [T][B][I] bf16
      const int64_t Bn = J.n0, Tn = J.n1, I = J.n2;
      bf16_t* dst = (bf16_t*)J.dst;
      if ((I & 7) == 0 && (J.s0 & 3) == 0 && (J.s1 & 3) == 0 && (reinterpret_cast<uintptr_t>(J.a) & 15) == 0) {
        for (int64_t c = gid; c < Bn * Tn * I / 8; c += stride) {     // 8 elements per thread and trip
          const int64_t i = c * 8, i2 = i % I, r = i / I, b = r % Bn, t = r / Bn;
          const float4* sp = reinterpret_cast<const float4*>(J.a + b * J.s0 + t * J.s1 + i2);
          const float4 u = sp[0], v = sp[1];
          *reinterpret_cast<bf16x8*>(dst + i) = (bf16x8){(bf16_t)u.x, (bf16_t)u.y, (bf16_t)u.z, (bf16_t)u.w,
                                                         (bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
        }
      } else {
        for (int64_t i = gid; i < Bn * Tn * I; i += stride) {
          const int64_t i2 = i % I, r = i / I, b = r % Bn, t = r / Bn;
          dst[i] = (bf16_t)J.a[b * J.s0 + t * J.s1 + i2];
        }
      }
      break;
    }
    case kPrepBlockifyX: {                // x -> [T][Bpad * I] fragment-major bf16 slabs (rows >= B zero)
      const int64_t Bn = J.n0, Tn = J.n1, I = J.n2, Bpad = J.H, per_t = Bpad * I / 8, kblocks = I >> 5;
      bf16_t* dst = (bf16_t*)J.dst;
      for (int64_t ci = gid; ci < per_t * Tn; ci += stride) {
        const int64_t t = ci / per_t, c = ci % per_t, blk = c >> 6, lane = c & 63;
        const int64_t r = (blk / kblocks) * 16 + (lane & 15), k = (blk % kblocks) * 32 + 8 * (lane >> 4);
        bf16x8 v;
        if (r < Bn && ((J.s0 | J.s1) & 3) == 0 && (reinterpret_cast<uintptr_t>(J.a) & 15) == 0) {
          const float4* sp = reinterpret_cast<const float4*>(J.a + r * J.s0 + t * J.s1 + k);
          const float4 u = sp[0], w = sp[1];
          v = (bf16x8){(bf16_t)u.x, (bf16_t)u.y, (bf16_t)u.z, (bf16_t)u.w, (bf16_t)w.x, (bf16_t)w.y, (bf16_t)w.z, (bf16_t)w.w};
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (bf16_t)(r < Bn ? J.a[r * J.s0 + t * J.s1 + k + e] : 0.f);
        }
        *reinterpret_cast<bf16x8*>(dst + ci * 8) = v;
      }
      break;
    }
  }
}

int launch_prep_multi(PrepArgs& A, hipStream_t st) {
  CSN_REQUIRE(A.njobs >= 1 && A.njobs <= kPrepMaxJobs, "launch_prep_multi: %d jobs", A.njobs);
  unsigned total = 0;
  for (int j = 0; j < A.njobs; ++j) {
    int64_t blocks = (A.job[j].work + 255) / 256;      // work = items (threads) of the job
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    A.job[j].blk_begin = total;
    A.job[j].blk_count = (unsigned)blocks;
    total += (unsigned)blocks;
  }
  prep_multi_kernel<<<total, 256, 0, st>>>(A);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

static inline unsigned cap_grid(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// x (b, t, i) float32 strided -> [T][Bpad * I] fragment-major bf16 slabs (rows >= B zero)
__global__ void blockify_x_kernel(const float* __restrict__ x, int64_t xsb, int64_t xst, int B, int Bpad, int T, int I,
                                  bf16_t* __restrict__ dst) {
  const int64_t per_t = (int64_t)Bpad * I / 8, nchunks = per_t * T;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int kblocks = I >> 5;
  for (int64_t ci = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; ci < nchunks; ci += stride) {
    const int64_t t = ci / per_t, c = ci % per_t;
    const int64_t blk = c >> 6, lane = c & 63;
    const int64_t r = (blk / kblocks) * 16 + (lane & 15);
    const int64_t k = (blk % kblocks) * 32 + 8 * (lane >> 4);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)(r < B ? x[r * xsb + t * xst + k + j] : 0.f);
    *reinterpret_cast<bf16x8*>(dst + ci * 8) = v;
  }
}
int launch_blockify_x(const float* x, int64_t xsb, int64_t xst, int B, int T, int I, void* dst, hipStream_t st) {
  const int Bpad = (B + 63) / 64 * 64;
  blockify_x_kernel<<<cap_grid((int64_t)T * Bpad * I / 8), 256, 0, st>>>(x, xsb, xst, B, Bpad, T, I, (bf16_t*)dst);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

int launch_blockify(const float* src, int64_t ld_r, int64_t ld_k, int64_t R, int64_t K, int perm_r, int perm_k,
                    int64_t H, void* dst, hipStream_t st) {
  blockify_cast_kernel<<<cap_grid(R * K / 8), 256, 0, st>>>(src, ld_r, ld_k, R, K, perm_r, perm_k, H, (bf16_t*)dst);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}
int launch_permute_rows_cast(const float* src, int64_t H, int64_t I, void* dst, hipStream_t st) {
  permute_rows_cast_kernel<<<cap_grid(4 * H * I), 256, 0, st>>>(src, H, I, (bf16_t*)dst);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}
int launch_transpose_perm_cast(const float* src, int64_t H, int64_t I, void* dst, hipStream_t st) {
  transpose_perm_cast_kernel<<<cap_grid(4 * H * I), 256, 0, st>>>(src, H, I, (bf16_t*)dst);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}
int launch_bias_perm_sum(const float* a, const float* b, int64_t H, float* dst, hipStream_t st) {
  bias_perm_sum_kernel<<<(unsigned)((4 * H + 255) / 256), 256, 0, st>>>(a, b, H, dst);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}
int launch_reduce_slabs_unperm(const float* slabs, int64_t slab_stride, int S, int64_t H, int64_t C, float* out,
                               hipStream_t st) {
  reduce_slabs_unperm_kernel<<<cap_grid(4 * H * C), 256, 0, st>>>(slabs, slab_stride, S, H, C, out);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

// ------------------------------------------------------------------------------------------
// forward: tile = 64 rows x 4*NQ units; wave w contracts k in [w H/4, (w+1) H/4)
// ------------------------------------------------------------------------------------------
template <int NQ, int NK>
__global__ void __launch_bounds__(256) lstm_cell_fwd_il_kernel(CellFwdArgs a) {
  constexpr int NT = 4 * NQ;                         // accumulator tiles per wave
  constexpr int NPAIR = 64 * NQ;                     // (row, unit-quad) pairs of the tile
  constexpr int NPASS = (NPAIR + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float4 red[];   // [4][NT][65]
  const CellFwdProb& P = a.p[blockIdx.z];
  const int B = a.B, H = a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 4 * NQ, m0 = blockIdx.y * 64;
  const int kblocks = H >> 5;
  CSN_STAMP(0);

  // epilogue operands: pair p -> row m0 + p / NQ, units u0 + 4 (p % NQ) .. + 3.  Requested FIRST (in flight under the
  // whole contraction) where the registers allow it; with 8 unit quads per workgroup (H = 128, 256, 512, 1024) the 40
  // registers are what made the kernel spill 74 -- and this compiler's spill code is not trusted: there
  // they are requested behind the contraction.
  constexpr bool EPI_EARLY = NQ < 8;
  float4 xp[NPASS][4], cp[NPASS];
  auto request_epilogue = [&]() {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const int p = tid + ps * 256;
      const int row = m0 + p / NQ, uq = u0 + 4 * (p % NQ);
      cp[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < NPAIR && row < B) {
        const float4* xr = reinterpret_cast<const float4*>(P.xproj + (int64_t)row * 4 * H + 4 * (int64_t)uq);
#pragma unroll
        for (int q = 0; q < 4; ++q) xp[ps][q] = nt_load(xr + q);
        if (P.c_prev != nullptr) cp[ps] = *reinterpret_cast<const float4*>(P.c_prev + (int64_t)row * H + uq);
      }
    }
  };
  if constexpr (EPI_EARLY) request_epilogue();

  f32x4 acc[4][NQ];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int j = 0; j < NQ; ++j) acc[rg][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (P.h_prev_blk != nullptr) {
    const int ks_beg = wave * (kblocks >> 2), ks_end = ks_beg + (kblocks >> 2);
    const bf16_t* hb[4];
    const bf16_t* wb[NQ];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) hb[rg] = P.h_prev_blk + ((int64_t)((m0 >> 4) + rg) * kblocks) * 512 + lane * 8;
#pragma unroll
    for (int j = 0; j < NQ; ++j) wb[j] = P.w_blk + ((int64_t)((u0 >> 2) + j) * kblocks) * 512 + lane * 8;

    bf16x8 hA[NK][4], wA[NK][NQ], hB[NK][4], wB[NK][NQ];
    auto load_pass = [&](int ks0, bf16x8 (&hf)[NK][4], bf16x8 (&wf)[NK][NQ]) {
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) hf[ks][rg] = *reinterpret_cast<const bf16x8*>(hb[rg] + (int64_t)(ks0 + ks) * 512);
#pragma unroll
        for (int j = 0; j < NQ; ++j) wf[ks][j] = *reinterpret_cast<const bf16x8*>(wb[j] + (int64_t)(ks0 + ks) * 512);
      }
    };
    auto mma_pass = [&](bf16x8 (&hf)[NK][4], bf16x8 (&wf)[NK][NQ]) {
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
          for (int j = 0; j < NQ; ++j)   // D[row = 4*unit_sub + gate][col = batch row]
            acc[rg][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][j], hf[ks][rg], acc[rg][j], 0, 0, 0);
    };
    int ks0 = ks_beg;
    load_pass(ks0, hA, wA);
    while (true) {
      int kn = ks0 + NK;
      if (kn < ks_end) load_pass(kn, hB, wB);
      mma_pass(hA, wA);
      if (kn >= ks_end) break;
      ks0 = kn;
      kn = ks0 + NK;
      if (kn < ks_end) load_pass(kn, hA, wA);
      mma_pass(hB, wB);
      if (kn >= ks_end) break;
      ks0 = kn;
    }
  }
  CSN_STAMP(1);

  // lane holds batch row (lane & 15), unit 4j + (lane >> 4), gates (i,f,g,o) = acc[..][0..3]
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int j = 0; j < NQ; ++j)
      red[(wave * NT + rg * NQ + j) * 65 + lane] = make_float4(acc[rg][j][0], acc[rg][j][1], acc[rg][j][2], acc[rg][j][3]);
  if constexpr (!EPI_EARLY) request_epilogue();
  __syncthreads();
  CSN_STAMP(2);

#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int p = tid + ps * 256;
    const int rl = p / NQ, j = p % NQ;
    const int row = m0 + rl, uq = u0 + 4 * j;
    if (p >= NPAIR || row >= B) continue;
    float gi[4], gf[4], gg[4], go[4], cn[4], hn[4];
    const float cpv[4] = {cp[ps].x, cp[ps].y, cp[ps].z, cp[ps].w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = ((rl >> 4) * NQ + j) * 65 + (rl & 15) + 16 * q;
      float4 s = red[idx];
#pragma unroll
      for (int w2 = 1; w2 < 4; ++w2) {
        const float4 v = red[w2 * NT * 65 + idx];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      gi[q] = fast_sigmoid(s.x + xp[ps][q].x);
      gf[q] = fast_sigmoid(s.y + xp[ps][q].y);
      gg[q] = fast_tanh(s.z + xp[ps][q].z);
      go[q] = fast_sigmoid(s.w + xp[ps][q].w);
      cn[q] = gf[q] * cpv[q] + gi[q] * gg[q];
      hn[q] = go[q] * fast_tanh(cn[q]);
    }
    if (P.gates_out != nullptr) {   // interleaved [row][4u + gate]: 16 consecutive bf16
      bf16x8 lo = {(bf16_t)gi[0], (bf16_t)gf[0], (bf16_t)gg[0], (bf16_t)go[0], (bf16_t)gi[1], (bf16_t)gf[1], (bf16_t)gg[1], (bf16_t)go[1]};
      bf16x8 hi = {(bf16_t)gi[2], (bf16_t)gf[2], (bf16_t)gg[2], (bf16_t)go[2], (bf16_t)gi[3], (bf16_t)gf[3], (bf16_t)gg[3], (bf16_t)go[3]};
      bf16x8* gp = reinterpret_cast<bf16x8*>(P.gates_out + (int64_t)row * 4 * H + 4 * (int64_t)uq);
      nt_store(gp, lo);
      nt_store(gp + 1, hi);
    }
    nt_store(reinterpret_cast<float4*>(P.c_out + (int64_t)row * H + uq), make_float4(cn[0], cn[1], cn[2], cn[3]));
    {
      bf16x4 hv = {(bf16_t)hn[0], (bf16_t)hn[1], (bf16_t)hn[2], (bf16_t)hn[3]};
      nt_store(reinterpret_cast<bf16x4*>(P.h_out + (int64_t)row * H + uq), hv);
    }
    Vec4<bf16_t>::store(P.h_out_blk + blk_offset(row, uq, H), hn);
  }
  CSN_STAMP(3);
}

// ------------------------------------------------------------------------------------------
// backward: tile = 32 rows x 16*NUG units; K' = 4H (interleaved); wave w contracts k' in [w H, (w+1) H)
// ------------------------------------------------------------------------------------------
template <int NUG, int NK>
__global__ void __launch_bounds__(256) lstm_cell_bwd_il_kernel(CellBwdArgs a) {
  constexpr int NT = 2 * NUG;
  constexpr int NPAIR = 32 * 4 * NUG;                // (row, unit-quad) pairs
  constexpr int NPASS = (NPAIR + 255) / 256;
  __shared__ float4 red[4 * NT * 65];
  const CellBwdProb& P = a.p[blockIdx.z];
  const int B = a.B, H = a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 16 * NUG, m0 = blockIdx.y * 32;
  const int K = 4 * H, kblocks = K >> 5;
  CSN_STAMP(0);

  // epilogue operands: pair p -> row m0 + p / (4 NUG), units u0 + 4 (p % (4 NUG)) .. + 3
  bf16x8 gt[NPASS][2];
  float4 cc[NPASS], cpv[NPASS], dcn[NPASS], dyv[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int p = tid + ps * 256;
    const int row = m0 + p / (4 * NUG), uq = u0 + 4 * (p % (4 * NUG));
    cpv[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
    dyv[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p < NPAIR && row < B) {
      const bf16x8* gp = reinterpret_cast<const bf16x8*>(P.gates + (int64_t)row * K + 4 * (int64_t)uq);
      gt[ps][0] = nt_load(gp);
      gt[ps][1] = nt_load(gp + 1);
      cc[ps] = *reinterpret_cast<const float4*>(P.c + (int64_t)row * H + uq);
      if (P.c_prev != nullptr) cpv[ps] = *reinterpret_cast<const float4*>(P.c_prev + (int64_t)row * H + uq);
      dcn[ps] = *reinterpret_cast<const float4*>(P.dc_carry + (int64_t)row * H + uq);
      if (P.dy != nullptr) dyv[ps] = *reinterpret_cast<const float4*>(P.dy + (int64_t)row * P.dy_ld + uq);
    }
  }

  f32x4 acc[2][NUG];
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int ug = 0; ug < NUG; ++ug) acc[rg][ug] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (P.dg_next_blk != nullptr) {
    const int ks_beg = wave * (H >> 5), ks_end = ks_beg + (H >> 5);
    const bf16_t* db[2];
    const bf16_t* wb[NUG];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) db[rg] = P.dg_next_blk + ((int64_t)((m0 >> 4) + rg) * kblocks) * 512 + lane * 8;
#pragma unroll
    for (int ug = 0; ug < NUG; ++ug) wb[ug] = P.wt_blk + ((int64_t)((u0 >> 4) + ug) * kblocks) * 512 + lane * 8;

    bf16x8 dA[NK][2], wA[NK][NUG], dB[NK][2], wB[NK][NUG];
    auto load_pass = [&](int ks0, bf16x8 (&df)[NK][2], bf16x8 (&wf)[NK][NUG]) {
#pragma unroll
      for (int ks = 0; ks < NK; ++ks) {
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) df[ks][rg] = *reinterpret_cast<const bf16x8*>(db[rg] + (int64_t)(ks0 + ks) * 512);
#pragma unroll
        for (int ug = 0; ug < NUG; ++ug) wf[ks][ug] = *reinterpret_cast<const bf16x8*>(wb[ug] + (int64_t)(ks0 + ks) * 512);
      }
    };
    auto mma_pass = [&](bf16x8 (&df)[NK][2], bf16x8 (&wf)[NK][NUG]) {
#pragma unroll
      for (int ks = 0; ks < NK; ++ks)
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
          for (int ug = 0; ug < NUG; ++ug)   // D[row = unit][col = batch row]
            acc[rg][ug] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][ug], df[ks][rg], acc[rg][ug], 0, 0, 0);
    };
    int ks0 = ks_beg;
    load_pass(ks0, dA, wA);
    while (true) {
      int kn = ks0 + NK;
      if (kn < ks_end) load_pass(kn, dB, wB);
      mma_pass(dA, wA);
      if (kn >= ks_end) break;
      ks0 = kn;
      kn = ks0 + NK;
      if (kn < ks_end) load_pass(kn, dA, wA);
      mma_pass(dB, wB);
      if (kn >= ks_end) break;
      ks0 = kn;
    }
  }
  CSN_STAMP(1);

  // lane holds batch row (lane & 15), units 16 ug + (lane >> 4) * 4 + r
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int ug = 0; ug < NUG; ++ug)
      red[(wave * NT + rg * NUG + ug) * 65 + lane] =
          make_float4(acc[rg][ug][0], acc[rg][ug][1], acc[rg][ug][2], acc[rg][ug][3]);
  __syncthreads();
  CSN_STAMP(2);

#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int p = tid + ps * 256;
    const int rl = p / (4 * NUG), jq = p % (4 * NUG);
    const int row = m0 + rl, uq = u0 + 4 * jq;
    if (p >= NPAIR || row >= B) continue;
    const int idx = ((rl >> 4) * NUG + (jq >> 2)) * 65 + (rl & 15) + 16 * (jq & 3);
    float4 s = red[idx];
#pragma unroll
    for (int w2 = 1; w2 < 4; ++w2) {
      const float4 v = red[w2 * NT * 65 + idx];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const float dh[4] = {s.x + dyv[ps].x, s.y + dyv[ps].y, s.z + dyv[ps].z, s.w + dyv[ps].w};
    const float cv[4] = {cc[ps].x, cc[ps].y, cc[ps].z, cc[ps].w};
    const float cpr[4] = {cpv[ps].x, cpv[ps].y, cpv[ps].z, cpv[ps].w};
    const float dcv[4] = {dcn[ps].x, dcn[ps].y, dcn[ps].z, dcn[ps].w};
    float out[16], dcarry[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bf16x8& g8 = gt[ps][q >> 1];
      const float gi = (float)g8[(q & 1) * 4 + 0], gf = (float)g8[(q & 1) * 4 + 1];
      const float gg = (float)g8[(q & 1) * 4 + 2], go = (float)g8[(q & 1) * 4 + 3];
      const float tc = fast_tanh(cv[q]);
      const float d_o = dh[q] * tc;
      const float dc = dh[q] * go * (1.0f - tc * tc) + dcv[q];
      out[4 * q + 0] = dc * gg * gi * (1.0f - gi);
      out[4 * q + 1] = dc * cpr[q] * gf * (1.0f - gf);
      out[4 * q + 2] = dc * gi * (1.0f - gg * gg);
      out[4 * q + 3] = d_o * go * (1.0f - go);
      dcarry[q] = dc * gf;
    }
    bf16x8 lo, hi;
#pragma unroll
    for (int e = 0; e < 8; ++e) { lo[e] = (bf16_t)out[e]; hi[e] = (bf16_t)out[8 + e]; }
    bf16x8* op = reinterpret_cast<bf16x8*>(P.dg_out + (int64_t)row * K + 4 * (int64_t)uq);
    nt_store(op, lo);
    nt_store(op + 1, hi);
    *reinterpret_cast<bf16x8*>(P.dg_out_blk + blk_offset(row, 4 * (int64_t)uq, K)) = lo;
    *reinterpret_cast<bf16x8*>(P.dg_out_blk + blk_offset(row, 4 * (int64_t)uq + 8, K)) = hi;
    Vec4<float>::store(P.dc_carry + (int64_t)row * H + uq, dcarry);
  }
  CSN_STAMP(3);
}

// ------------------------------------------------------------------------------------------
static int pick_nk(int steps, int max_nk) {
  for (int nk = max_nk; nk >= 1; --nk)
    if (steps % nk == 0) return nk;
  return 1;
}

bool cell_blk_supported(int H, int dtype, const Options& opt) {
  return dtype == CSN_BF16 && H % 128 == 0 && !opt.cell_v1;
}

template <int NQ, int NK>
static int launch_fwd_t(const CellFwdArgs& a, int nprob, hipStream_t st) {
  const size_t lds = (size_t)4 * 4 * NQ * 65 * sizeof(float4);
  if (int rc = ensure_dyn_lds<&lstm_cell_fwd_il_kernel<NQ, NK>>((int)lds)) return rc;
  dim3 grid((unsigned)(a.H / (4 * NQ)), (unsigned)((a.B + 63) / 64), (unsigned)nprob);
  lstm_cell_fwd_il_kernel<NQ, NK><<<grid, 256, lds, st>>>(a);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

int launch_cell_fwd_il(const CellFwdArgs& a, int nprob, hipStream_t st, int max_nk) {
  const int H = a.H;
  const int steps = H / 128;                       // k-steps per wave
  const int nk = pick_nk(steps, max_nk);
  const int nq = (H % 24 == 0) ? 6 : ((H % 32 == 0) ? 8 : 4);
#define CSN_CASE(NQ, NK) if (nq == NQ) return launch_fwd_t<NQ, NK>(a, nprob, st)
  // (k-blocks per pass: only 1 is built -- the deeper passes CSN_FWD_NK used to select spilled 32 - 147 registers)
  (void)nk;
  CSN_CASE(6, 1); CSN_CASE(8, 1); CSN_CASE(4, 1);
#undef CSN_CASE
  if (nq == 8) return launch_fwd_t<8, 1>(a, nprob, st);
  return fail(CSN_ERR_UNSUPPORTED, "launch_cell_fwd_il: no kernel for H=%d", H);
}

template <int NUG, int NK>
static int launch_bwd_t(const CellBwdArgs& a, int nprob, hipStream_t st) {
  dim3 grid((unsigned)(a.H / (16 * NUG)), (unsigned)((a.B + 31) / 32), (unsigned)nprob);
  lstm_cell_bwd_il_kernel<NUG, NK><<<grid, 256, 0, st>>>(a);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

int launch_cell_bwd_il(const CellBwdArgs& a, int nprob, hipStream_t st) {
  const int H = a.H;
  const int steps = H / 32;
  const int nug = (H % 48 == 0) ? 3 : ((H % 64 == 0) ? 4 : 2);
  const int nk = pick_nk(steps, 4);
#define CSN_CASE(NUG, NK) if (nug == NUG && nk == NK) return launch_bwd_t<NUG, NK>(a, nprob, st)
  CSN_CASE(3, 4); CSN_CASE(3, 3); CSN_CASE(3, 2); CSN_CASE(3, 1);
  CSN_CASE(4, 4); CSN_CASE(4, 2); CSN_CASE(4, 1);
  CSN_CASE(2, 4); CSN_CASE(2, 2); CSN_CASE(2, 1);
#undef CSN_CASE
  return fail(CSN_ERR_UNSUPPORTED, "launch_cell_bwd_il: no kernel for H=%d", H);
}

}  // namespace csn
