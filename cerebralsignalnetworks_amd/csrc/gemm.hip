// GEMMs of the LSTM path (input projections, input gradients, weight gradients).
// They replace the GEMM calls inside ATen's LSTM that nn.LSTM reaches at
// /root/reference/LSTMDistill.py:118,132.
//
//  gemm_generic<T>   any strides, T in {f32, bf16}; operands are widened to f32 and multiplied
//                    with the exact-f32 MFMA v_mfma_f32_16x16x4_f32 (bitwise an fmaf chain).
//                    This is the CSN_F32 parity path and the fallback for odd shapes.
//  gemm_nt_bf16      C[M,N] = A[M,K] * Bt[N,K]^T, 128x128x64 tiles, 4 waves (2x2) of 64x64,
//                    v_mfma_f32_16x16x32_bf16, double-buffered LDS with an XOR swizzle that makes
//                    the ds_read_b128 fragment reads conflict-free, register-staged prefetch.
//                    MFMA-bound roofline; algorithmic flops 2*M*N*K.
//  gemm_tn_bf16      C[M,N] = A[K,M]^T * B[K,N] (weight gradient: contraction over T*B rows),
//                    operands staged row-major [k][m] / [k][n] in LDS and read column-major with
//                    ds_read_b64_tr_b16; split-K over blockIdx.z into float32 slabs that a
//                    fixed-order reduction combines (bitwise reproducible, no atomics).
#include "csn_common.h"
#include "lstm_cell_common.h"

namespace csn {

// =====================================================================================
// generic strided GEMM on the f32 MFMA
// =====================================================================================
template <typename T>
__global__ void __launch_bounds__(256)
gemm_generic_kernel(const T* __restrict__ A, int64_t sam, int64_t sak, const T* __restrict__ Bm, int64_t sbk,
                    int64_t sbn, const float* __restrict__ bias, void* __restrict__ Cv, int64_t ldc, int64_t M,
                    int64_t N, int64_t K, int out_bf16, int accumulate, int64_t k_per_split, int64_t slab_stride) {
  __shared__ float As[16][64 + 1];
  __shared__ float Bs[16][64 + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * 64, n0 = (int64_t)blockIdx.x * 64;
  const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  const bool a_kfast = (sak == 1), b_kfast = (sbk == 1);

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int64_t k0 = kbeg; k0 < kend; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      int kk, mm;
      if (a_kfast) { kk = idx & 15; mm = idx >> 4; } else { mm = idx & 63; kk = idx >> 6; }
      const int64_t m = m0 + mm, k = k0 + kk;
      As[kk][mm] = (m < M && k < kend) ? to_f32(A[m * sam + k * sak]) : 0.0f;
      int kb, nn;
      if (b_kfast) { kb = idx & 15; nn = idx >> 4; } else { nn = idx & 63; kb = idx >> 6; }
      const int64_t n = n0 + nn, k2 = k0 + kb;
      Bs[kb][nn] = (n < N && k2 < kend) ? to_f32(Bm[k2 * sbk + n * sbn]) : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = As[kk * 4 + (lane >> 4)][wm * 32 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 2; ++j) bv[j] = Bs[kk * 4 + (lane >> 4)][wn * 32 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          // operands swapped: D[row = n][col = m]  -> lane holds m = lane&15, n = (lane>>4)*4 + r
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  float* Cf = (float*)Cv + (int64_t)blockIdx.z * slab_stride;
  bf16_t* Cb = (bf16_t*)Cv;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int64_t m = m0 + wm * 32 + i * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t n = n0 + wn * 32 + j * 16 + (lane >> 4) * 4 + r;
        if (m < M && n < N) {
          float v = acc[i][j][r];
          if (bias) v += bias[n];
          if (out_bf16) Cb[m * ldc + n] = (bf16_t)v;
          else if (accumulate) Cf[m * ldc + n] += v;
          else Cf[m * ldc + n] = v;
        }
      }
    }
}

// =====================================================================================
// float32 GEMM on the exact f32 MFMA, 128 x 128 x 16 tiles (round 3: the exact-f32 training path spent 52 of its 108 ms
// in the 64 x 64 generic kernel above at 61 TFLOP/s).  Both operands contiguous along K (KFAST: C = A[M,K] B[N,K]^T, the
// projections) or along M / N (C = A[K,M]^T B[K,N], the weight gradients); 16-byte global loads, register-staged double
// buffer, 4 waves of 64 x 64 outputs (16 accumulator tiles each), D[row = n][col = m] so that a lane's four results are
// four consecutive columns of one row of C.  Shapes outside (M, N multiples of 128, K slices multiples of 16, 16-byte
// aligned) stay with the generic kernel.
// =====================================================================================
static constexpr int kF32Pitch = 144;          // floats per LDS k-row: 128 + 16, i.e. 16 banks: the two k-rows of a ds_read_b32 half-wave never collide
template <bool KFAST>
__global__ void __launch_bounds__(256)
gemm_f32_128_kernel(const float* __restrict__ A, int64_t lda, const float* __restrict__ Bm, int64_t ldb,
                    const float* __restrict__ bias, void* __restrict__ Cv, int64_t ldc, int64_t M, int64_t N, int64_t K,
                    int out_bf16, int accumulate, int64_t k_per_split, int64_t slab_stride) {
  __shared__ __attribute__((aligned(16))) float As[2][16][kF32Pitch];
  __shared__ __attribute__((aligned(16))) float Bs[2][16][kF32Pitch];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t m0 = (int64_t)blockIdx.y * 128, n0 = (int64_t)blockIdx.x * 128;
  const int64_t kbeg = (int64_t)blockIdx.z * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  const int nk = (int)((kend - kbeg) / 16);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 ra[2], rb[2];
  auto load_tile = [&](int kt) {
    const int64_t k0 = kbeg + (int64_t)kt * 16;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      if constexpr (KFAST) {
        const int row = idx >> 2, k4 = idx & 3;
        ra[i] = *reinterpret_cast<const f32x4*>(A + (m0 + row) * lda + k0 + 4 * k4);
        rb[i] = *reinterpret_cast<const f32x4*>(Bm + (n0 + row) * ldb + k0 + 4 * k4);
      } else {
        const int kr = idx >> 5, c4 = idx & 31;
        ra[i] = *reinterpret_cast<const f32x4*>(A + (k0 + kr) * lda + m0 + 4 * c4);
        rb[i] = *reinterpret_cast<const f32x4*>(Bm + (k0 + kr) * ldb + n0 + 4 * c4);
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int idx = tid + i * 256;
      if constexpr (KFAST) {
        const int row = idx >> 2, k4 = idx & 3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          As[buf][4 * k4 + e][row] = ra[i][e];
          Bs[buf][4 * k4 + e][row] = rb[i][e];
        }
      } else {
        const int kr = idx >> 5, c4 = idx & 31;
        *reinterpret_cast<f32x4*>(&As[buf][kr][4 * c4]) = ra[i];
        *reinterpret_cast<f32x4*>(&Bs[buf][kr][4 * c4]) = rb[i];
      }
    }
  };
  if (nk > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);            // in flight under this tile's MFMAs
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[buf][kk * 4 + (lane >> 4)][wm * 64 + i * 16 + (lane & 15)];
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = Bs[buf][kk * 4 + (lane >> 4)][wn * 64 + j * 16 + (lane & 15)];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);           // the other buffer: every wave left it at the barrier below, one tile ago
    __syncthreads();
  }

  float* Cf = (float*)Cv + (int64_t)blockIdx.z * slab_stride;
  bf16_t* Cb = (bf16_t*)Cv;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      f32x4 v = acc[i][j];
      if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
      if (out_bf16) {
        *reinterpret_cast<bf16x4*>(Cb + m * ldc + n) = (bf16x4){(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      } else {
        f32x4* dst = reinterpret_cast<f32x4*>(Cf + m * ldc + n);
        *dst = accumulate ? *dst + v : v;
      }
    }
  }
}

static int launch_generic(const void* A, int64_t sam, int64_t sak, const void* Bm, int64_t sbk, int64_t sbn,
                          const float* bias, void* C, int64_t ldc, int64_t M, int64_t N, int64_t K, int dtype,
                          int out_dtype, int accumulate, int splits, int64_t slab_stride, hipStream_t st) {
  int64_t kper = (K + splits - 1) / splits;
  kper = (kper + 15) / 16 * 16;
  if (dtype == CSN_F32 && M % 128 == 0 && N % 128 == 0 && K % 16 == 0 && ldc % 4 == 0 &&
      ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bm) | reinterpret_cast<uintptr_t>(C) |
        reinterpret_cast<uintptr_t>(bias)) & 15) == 0 && slab_stride % 4 == 0) {
    const dim3 g128((unsigned)(N / 128), (unsigned)(M / 128), (unsigned)splits);
    if (sak == 1 && sbk == 1 && sam % 4 == 0 && sbn % 4 == 0) {            // both operands contiguous along K
      gemm_f32_128_kernel<true><<<g128, 256, 0, st>>>((const float*)A, sam, (const float*)Bm, sbn, bias, C, ldc, M, N, K,
                                                       out_dtype == CSN_BF16, accumulate, kper, slab_stride);
      CSN_LAUNCH_CHECK();
      return CSN_OK;
    }
    if (sam == 1 && sbn == 1 && sak % 4 == 0 && sbk % 4 == 0) {            // contiguous along M / N (contraction over rows)
      gemm_f32_128_kernel<false><<<g128, 256, 0, st>>>((const float*)A, sak, (const float*)Bm, sbk, bias, C, ldc, M, N, K,
                                                        out_dtype == CSN_BF16, accumulate, kper, slab_stride);
      CSN_LAUNCH_CHECK();
      return CSN_OK;
    }
  }
  dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 63) / 64), (unsigned)splits);
  if (dtype == CSN_BF16)
    gemm_generic_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)A, sam, sak, (const bf16_t*)Bm, sbk, sbn, bias, C,
                                                      ldc, M, N, K, out_dtype == CSN_BF16, accumulate, kper,
                                                      slab_stride);
  else
    gemm_generic_kernel<float><<<grid, 256, 0, st>>>((const float*)A, sam, sak, (const float*)Bm, sbk, sbn, bias, C,
                                                     ldc, M, N, K, out_dtype == CSN_BF16, accumulate, kper,
                                                     slab_stride);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

// XCD-aware block order (cdna_hip_programming.md T1): hardware deals consecutive workgroups round-robin
// over the 8 XCDs (each with a private L2), so tiles that share an operand panel would land on 8
// different L2s and the panel would cross the fabric 8 times (measured with FETCH_SIZE: 12x the
// algorithmic bytes on the NT GEMM, 4.5x on the TN GEMM).  This bijection hands every XCD one
// contiguous range of the logical tile order instead.  Speed only -- any placement is correct.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nwg) {
  const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// =====================================================================================
// bf16 NT GEMM, 128x128x64 tiles
// =====================================================================================
// LDS image of one operand tile: 128 rows x 64 k (bf16) = 128-byte rows, eight 16-byte chunks
// per row.  Chunk ch of row r is stored at chunk (ch ^ ((r >> 1) & 7)): with that XOR the 16
// lanes of every ds_read_b128 lane group ({0-3,12-15,20-27}, ...) hit 16 distinct 16-byte
// slots of the 256-byte bank row when lane l reads row (l & 15), chunk kk*4 + (l >> 4).
__device__ __forceinline__ int nt_lds_off(int row, int chunk) {  // byte offset inside a tile
  return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4);
}

// NBUF = 2: double-buffered LDS (64 KB, one barrier per k-tile).  NBUF = 1: one 32 KB buffer, two
// barriers per k-tile -- small enough to co-reside on a CU with the weight-stationary LSTM kernel
// (100 KB LDS), which is how the input-projection GEMMs overlap the recurrence.
template <typename OutT, int NBUF>
__global__ void __launch_bounds__(256)
gemm_nt_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bt, const float* __restrict__ bias,
                    OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NBUF buffers x (A 16 KB + B 16 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // logical order: N tiles fastest inside an M tile, so one XCD walks the N tiles of "its" M tiles and
  // the A panel of an M tile is fetched into a single L2
  const unsigned ntn = (unsigned)((N + 127) / 128);
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const int64_t m0 = (int64_t)(lid / ntn) * 128, n0 = (int64_t)(lid % ntn) * 128;
  const int nk = (int)((K + 63) / 64);

  // global->register staging: each thread moves 4 chunks of A and 4 of B per k-tile.
  // chunk id q = tid + i*256 (0..1023): row = q >> 3, chunk = q & 7 (8 lanes cover one 128-B row).
  uint4 ra[4], rb[4];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + i * 256;
      const int row = q >> 3, ch = q & 7;
      const int64_t k = (int64_t)kt * 64 + ch * 8;
      const int64_t am = m0 + row, bn = n0 + row;
      ra[i] = (am < M && k < K) ? *reinterpret_cast<const uint4*>(A + am * K + k) : make_uint4(0, 0, 0, 0);
      rb[i] = (bn < N && k < K) ? *reinterpret_cast<const uint4*>(Bt + bn * K + k) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tile = [&](int buf) {
    char* a_s = smem + buf * 32768;
    char* b_s = a_s + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = tid + i * 256;
      const int row = q >> 3, ch = q & 7;
      *reinterpret_cast<uint4*>(a_s + nt_lds_off(row, ch)) = ra[i];
      *reinterpret_cast<uint4*>(b_s + nt_lds_off(row, ch)) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_tile(0);
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = (NBUF == 2) ? (kt & 1) : 0;
    if (kt + 1 < nk) load_tile(kt + 1);
    const char* a_s = smem + buf * 32768;
    const char* b_s = a_s + 16384;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(a_s + nt_lds_off(wm * 64 + i * 16 + (lane & 15), kk * 4 + (lane >> 4)));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(b_s + nt_lds_off(wn * 64 + j * 16 + (lane & 15), kk * 4 + (lane >> 4)));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          // swapped operands: D[row = n][col = m]: lane holds m = lane&15, n = (lane>>4)*4 + r
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (NBUF == 1) __syncthreads();           // everyone has read the tile before it is overwritten
    if (kt + 1 < nk) store_tile(NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (n + 3 < N) {
        f32x4 v = acc[i][j];
        if (bias) {
          const float4 bz = *reinterpret_cast<const float4*>(bias + n);
          v[0] += bz.x; v[1] += bz.y; v[2] += bz.z; v[3] += bz.w;
        }
        if constexpr (sizeof(OutT) == 4) {
          float4* dst = reinterpret_cast<float4*>((float*)C + m * N + n);
          if (accumulate) { const float4 o = *dst; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
          *dst = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          *reinterpret_cast<bf16x4*>((bf16_t*)C + m * N + n) = o;
        }
      } else {
        for (int r = 0; r < 4; ++r)
          if (n + r < N) {
            float v = acc[i][j][r] + (bias ? bias[n + r] : 0.f);
            if constexpr (sizeof(OutT) == 4) {
              float* dst = (float*)C + m * N + n + r;
              *dst = accumulate ? *dst + v : v;
            } else {
              ((bf16_t*)C)[m * N + n + r] = (bf16_t)v;
            }
          }
      }
    }
  }
}

// =====================================================================================
// bf16 TN GEMM (weight gradient), 128x128 output tile, BK = 32 rows of the contraction
// =====================================================================================
// LDS image of one operand tile: 32 k-rows x 128 columns (bf16) = 256-byte rows, eight
// 16-column blocks per row.  Block cb of k-row r is stored at block (cb ^ s(r)),
// s(r) = (r & 3) | (((r >> 3) & 1) << 2), so the 8 k-rows that the lower (or upper) 32 lanes of
// a ds_read_b64_tr_b16 address fall into 8 distinct 32-byte bank windows.
__device__ __forceinline__ int tn_lds_off(int krow, int col) {  // byte offset; col in elements
  const int s = (krow & 3) | (((krow >> 3) & 1) << 2);
  return krow * 256 + ((((col >> 4) ^ s) << 4) + (col & 15)) * 2;
}

// Fragment of a 16(m) x 32(k) operand from the [k][m] LDS image: lane l (g = l>>4, i = l&15)
// ends with elements j = 0..7 = tile[k = 8g + j][m = mbase + i].
// ds_read_b64_tr_b16 (per 16-lane group): lane 4q+p supplies the address of k-row q, columns
// 4p..4p+3 of a 4x16 block; lane i receives column i, k-row q in element q.
template <bool USE_TR>
__device__ __forceinline__ bf16x8 tn_load_frag(const char* tile, int mbase, int lane) {
  const int g = lane >> 4, i = lane & 15;
  bf16x8 out;
  if constexpr (USE_TR) {
    const int q = i >> 2, p = i & 3;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + tn_lds_off(8 * g + q, mbase + 4 * p)));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)(tile + tn_lds_off(8 * g + 4 + q, mbase + 4 * p)));
    union { s16x4 s[2]; bf16x8 b; } u;
    u.s[0] = lo;
    u.s[1] = hi;
    out = u.b;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = *reinterpret_cast<const bf16_t*>(tile + tn_lds_off(8 * g + j, mbase + i));
  }
  return out;
}

template <bool USE_TR>
__global__ void __launch_bounds__(256)
gemm_tn_bf16_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bm, float* __restrict__ slabs,
                    int64_t M, int64_t N, int64_t K, int64_t k_per_split) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 16384];  // 2 buffers x (A 8 KB + B 8 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // logical order [split][m tile][n tile]: an XCD gets whole K splits (or contiguous parts of one), so
  // the tiles that stream the same rows of A and B at the same time share one L2
  const unsigned ntn = (unsigned)((N + 127) / 128), ntm = (unsigned)((M + 127) / 128);
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned zsplit = lid / (ntm * ntn), rem = lid % (ntm * ntn);
  const int64_t m0 = (int64_t)(rem / ntn) * 128, n0 = (int64_t)(rem % ntn) * 128;
  const int64_t kbeg = (int64_t)zsplit * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  const int nk = (int)((kend - kbeg + 31) / 32);

  // staging: tile = 32 k-rows x 16 chunks (16 B = 8 columns); q = tid + i*256, i<2: krow = q>>4, ch = q&15
  uint4 ra[2], rb[2];
  auto load_tile = [&](int kt) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + i * 256;
      const int kr = q >> 4, ch = q & 15;
      const int64_t k = kbeg + (int64_t)kt * 32 + kr;
      const int64_t am = m0 + ch * 8, bn = n0 + ch * 8;
      ra[i] = (k < kend && am < M) ? *reinterpret_cast<const uint4*>(A + k * M + am) : make_uint4(0, 0, 0, 0);
      rb[i] = (k < kend && bn < N) ? *reinterpret_cast<const uint4*>(Bm + k * N + bn) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tile = [&](int buf) {
    char* a_s = smem + buf * 16384;
    char* b_s = a_s + 8192;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int q = tid + i * 256;
      const int kr = q >> 4, ch = q & 15;
      *reinterpret_cast<uint4*>(a_s + tn_lds_off(kr, ch * 8)) = ra[i];
      *reinterpret_cast<uint4*>(b_s + tn_lds_off(kr, ch * 8)) = rb[i];
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(kt + 1);
    const char* a_s = smem + buf * 16384;
    const char* b_s = a_s + 8192;
    bf16x8 af[4], bfr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = tn_load_frag<USE_TR>(a_s, wm * 64 + i * 16, lane);
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = tn_load_frag<USE_TR>(b_s, wn * 64 + j * 16, lane);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  float* C = slabs + (int64_t)zsplit * M * N;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (n + 3 < N && (N & 3) == 0) {
        *reinterpret_cast<float4*>(C + m * N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      } else {
        for (int r = 0; r < 4; ++r)
          if (n + r < N) C[m * N + n + r] = acc[i][j][r];
      }
    }
  }
}

// =====================================================================================
// LDS-DMA variants (global_load_lds, 16 B per lane): no VGPR staging and no ds_write pass -- the tile goes
// HBM/L2 -> LDS directly, two LDS buffers, ONE barrier per 64-deep k-step, the next tile's loads in flight
// under the MFMAs of the current one.  An LDS-DMA wave-instruction writes 1 KB lane-linearly (LDS address =
// wave-uniform base + 16 * lane), so the XOR swizzles of the images above are applied on the SOURCE side:
// lane i fetches the global chunk whose swizzled position is i (the read side is unchanged).
// Preconditions (checked by the launchers, otherwise the register-staged kernels run): K % 64 == 0, so no
// partial k-tile exists (LDS-DMA cannot zero-fill); rows beyond M / N are clamped to the last valid row and
// their results discarded by the epilogue.
// =====================================================================================
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
}

template <typename OutT>
__global__ void __launch_bounds__(256)
gemm_nt_dma_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bt, const float* __restrict__ bias,
                   OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, int accumulate) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // 2 buffers x (A 16 KB + B 16 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned ntn = (unsigned)((N + 127) / 128);
  const unsigned ntiles = ntn * (unsigned)((M + 127) / 128);
  const int nk = (int)(K / 64);
  // a grid smaller than the tile count walks the tiles (the "beside" form: few workgroups, one per CU, next to a
  // persistent LSTM launch); pass `it` gives the workgroups of one XCD a contiguous range of logical tiles
  for (unsigned lid = xcd_remap(blockIdx.x, gridDim.x); lid < ntiles; lid += gridDim.x) {
  const int64_t m0 = (int64_t)(lid / ntn) * 128, n0 = (int64_t)(lid % ntn) * 128;
  __syncthreads();     // every wave has left the previous tile's last LDS buffer

  // staging: instruction i of wave w fills rows [(4 i + w) 8, +8) of a tile (8 rows x 8 chunks = 1 KB);
  // lane -> row r = lane >> 3, LDS chunk position c = lane & 7 <- global chunk c ^ ((row >> 1) & 7)
  const bf16_t* a_src[4];
  const bf16_t* b_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (4 * i + wave) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((row >> 1) & 7);
    int64_t am = m0 + row, bn = n0 + row;
    am = am < M ? am : M - 1;
    bn = bn < N ? bn : N - 1;
    a_src[i] = A + am * K + ch * 8;
    b_src[i] = Bt + bn * K + ch * 8;
  }
  auto issue = [&](int kt, int buf) {
    char* a_s = smem + buf * 32768;
    char* b_s = a_s + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16(a_src[i] + (int64_t)kt * 64, a_s + (4 * i + wave) * 1024);
      glds16(b_src[i] + (int64_t)kt * 64, b_s + (4 * i + wave) * 1024);
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  for (int kt = 0; kt < nk; ++kt) {
    // tile kt has landed (own loads: vmcnt; everyone's: barrier) and every wave is done reading the other buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
    const char* a_s = smem + (kt & 1) * 32768;
    const char* b_s = a_s + 16384;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(a_s + nt_lds_off(wm * 64 + i * 16 + (lane & 15), kk * 4 + (lane >> 4)));
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(b_s + nt_lds_off(wn * 64 + j * 16 + (lane & 15), kk * 4 + (lane >> 4)));
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (n + 3 < N) {
        f32x4 v = acc[i][j];
        if (bias) {
          const float4 bz = *reinterpret_cast<const float4*>(bias + n);
          v[0] += bz.x; v[1] += bz.y; v[2] += bz.z; v[3] += bz.w;
        }
        if constexpr (sizeof(OutT) == 4) {
          float4* dst = reinterpret_cast<float4*>((float*)C + m * N + n);
          if (accumulate) { const float4 o = *dst; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
          *dst = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          *reinterpret_cast<bf16x4*>((bf16_t*)C + m * N + n) = o;
        }
      } else {
        for (int r = 0; r < 4; ++r)
          if (n + r < N) {
            float v = acc[i][j][r] + (bias ? bias[n + r] : 0.f);
            if constexpr (sizeof(OutT) == 4) {
              float* dst = (float*)C + m * N + n + r;
              *dst = accumulate ? *dst + v : v;
            } else {
              ((bf16_t*)C)[m * N + n + r] = (bf16_t)v;
            }
          }
      }
    }
  }
  }   // tile loop
}

// NT, 256 x 128 output tile, 8 waves (4 along M x 2 along N, 64 x 64 outputs per wave), ring of NSTAGE LDS-DMA
// stages of 64 contraction columns (A 32 KB + B 16 KB each, the 128-byte-row image and swizzle of nt_lds_off),
// counted vmcnt + raw barrier, fragment reads in inline asm (see gemm_tn_256_kernel for why).
__device__ __forceinline__ bf16x8 lds_read_b128(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

template <typename OutT, int NSTAGE>
__global__ void __launch_bounds__(512)
gemm_nt_256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bt, const float* __restrict__ bias,
                   OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, int accumulate) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // NSTAGE x (A 32 KB + B 16 KB)
  constexpr unsigned kStage = 49152;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned ntn = (unsigned)((N + 127) / 128);
  const unsigned ntiles = ntn * (unsigned)((M + 255) / 256);
  const int nk = (int)(K / 64);
  // a grid smaller than the tile count walks the tiles (the "beside" form, see gemm_nt_dma_kernel)
  for (unsigned lid = xcd_remap(blockIdx.x, gridDim.x); lid < ntiles; lid += gridDim.x) {
  const int64_t m0 = (int64_t)(lid / ntn) * 256, n0 = (int64_t)(lid % ntn) * 128;
  __syncthreads();     // every wave has left the previous tile's last stage

  // staging: a 1 KB instruction fills 8 rows x 8 chunks; A has 32 of them per stage (4 per wave), B 16 (2 per
  // wave); lane -> row r = lane >> 3, LDS chunk position c = lane & 7 <- global chunk c ^ ((row >> 1) & 7)
  const bf16_t* a_src[4];
  const bf16_t* b_src[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (8 * i + wave) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((row >> 1) & 7);
    int64_t am = m0 + row;
    am = am < M ? am : M - 1;
    a_src[i] = A + am * K + ch * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = (8 * i + wave) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((row >> 1) & 7);
    int64_t bn = n0 + row;
    bn = bn < N ? bn : N - 1;
    b_src[i] = Bt + bn * K + ch * 8;
  }
  auto issue = [&](int kt) {
    char* a_s = smem + (kt % NSTAGE) * kStage;
    char* b_s = a_s + 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + (int64_t)kt * 64, a_s + (8 * i + wave) * 1024);
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(b_src[i] + (int64_t)kt * 64, b_s + (8 * i + wave) * 1024);
  };

  // fragment addresses inside a stage: tile i of A is 2048 bytes further, the second k-half flips chunk bit 2
  const unsigned sw = (unsigned)((lane & 15) >> 1);
  const unsigned a_base = (unsigned)((wm * 64 + (lane & 15)) * 128) + ((((unsigned)lane >> 4) ^ sw) << 4);
  const unsigned b_base = 32768u + (unsigned)((wn * 64 + (lane & 15)) * 128) + ((((unsigned)lane >> 4) ^ sw) << 4);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int h = 0; h < NSTAGE - 1; ++h)
    if (h < nk) issue(h);
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;            // stages issued after kt: min(NSTAGE - 2, ahead) stay in flight
    if (ahead >= NSTAGE - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * (NSTAGE - 2)) : "memory");
    else if (NSTAGE > 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1);
    const unsigned sb = (unsigned)(kt % NSTAGE) * kStage;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const unsigned xo = kk ? 64u : 0u;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = lds_read_b128(((b_base ^ xo) + sb) + j * 2048u);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = lds_read_b128(((a_base ^ xo) + sb) + i * 2048u);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i == 0) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else if (i == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if (i == 2) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue through LDS: in the accumulator layout a wave's store instruction covers 16 rows x 64 bytes (16 half
  // cache lines); staged as a [256][128] f32 image in the ring's memory, every store instruction writes two full
  // 512-byte rows of the tile
  constexpr int LDC = 132;                    // padded row (floats): the 16 rows of a fragment column fall in 16 distinct bank groups
  static_assert(256 * LDC * 4 <= NSTAGE * 49152, "the C image lives in the staging ring");
  float* const cs = reinterpret_cast<float*>(smem);
  __syncthreads();                            // (every DMA has landed: the last k-step waited vmcnt(0)) all waves have left the ring
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<f32x4*>(cs + (wm * 64 + i * 16 + (lane & 15)) * LDC + wn * 64 + j * 16 + (lane >> 4) * 4) = acc[i][j];
  __syncthreads();
  {
    const int c4 = tid & 31, r0 = tid >> 5;
    const int64_t n = n0 + c4 * 4;
    const bool vec = n + 3 < N;
    f32x4 bz = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (n + r < N) bz[r] = bias[n + r];
    }
#pragma unroll 4
    for (int pass = 0; pass < 16; ++pass) {
      const int row = pass * 16 + r0;
      const int64_t m = m0 + row;
      if (m >= M) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(cs + row * LDC + c4 * 4) + bz;
#if defined(CSN_NT_ABL) && CSN_NT_ABL == 1     // (ablation, timing only: no C stores unless a value is NaN)
      if (v[0] == v[0]) continue;
#endif
      if (vec) {
        if constexpr (sizeof(OutT) == 4) {
          float4* dst = reinterpret_cast<float4*>((float*)C + m * N + n);
          if (accumulate) { const float4 o = *dst; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
          *dst = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          *reinterpret_cast<bf16x4*>((bf16_t*)C + m * N + n) = o;
        }
      } else {
        for (int r = 0; r < 4; ++r)
          if (n + r < N) {
            if constexpr (sizeof(OutT) == 4) {
              float* dst = (float*)C + m * N + n + r;
              *dst = accumulate ? *dst + v[r] : v[r];
            } else {
              ((bf16_t*)C)[m * N + n + r] = (bf16_t)v[r];
            }
          }
      }
    }
  }
  }   // tile loop
}

// NT, 256 x BN output tile, BN = 192 or 256 (N % BN == 0): the same kernel with a wider tile -- at the LSTM's chunk shapes
// (8192 x 3072 x 768 with BN = 192, 8192 x 4096 x 1024 with BN = 256) 512 tiles = exactly two per CU instead of three /
// four, 21 % / 33 % fewer bytes through the L2 -> CU fabric, which is what the 256 x 128 form is bound by (DESIGN.md
// section 6: the vendor library's 192 x 256 kernel is faster by exactly that ratio).  8 waves (4 along M x 2 along N,
// 64 x BN/2 outputs per wave).  A stage of 64 contraction columns is 32 KB of A + 24 / 32 KB of B, three of which do not
// fit the 160 KB of LDS: A keeps a ring of THREE stages (two k-steps ahead: the streaming operand), B a ring of TWO (one
// k-step ahead: a few hundred KB per column tile, read by 16 workgroups at a time, L2-resident).  144 / 160 KB.
template <typename OutT, int BN>
__global__ void __launch_bounds__(512)
gemm_nt_wide_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bt, const float* __restrict__ bias,
                   OutT* __restrict__ C, int64_t M, int64_t N, int64_t K, int accumulate) {
  static_assert(BN == 192 || BN == 256, "column tile");
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // A: 3 x 32 KB, then B: 2 x (24 | 32) KB
  constexpr unsigned kBBase = 3 * 32768, kBStage = BN * 128;
  constexpr int NBI = BN / 64;       // B staging instructions per wave and stage
  constexpr int NBF = BN / 32;       // B fragments (16 columns each) per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const unsigned ntn = (unsigned)(N / BN);
  const unsigned ntiles = ntn * (unsigned)((M + 255) / 256);
  const int nk = (int)(K / 64);
#if defined(CSN_NT_STAGGER)      // timing experiment (tools/abl_build.sh): every second first-wave workgroup of an XCD starts late, so that
  if (blockIdx.x < 256u && ((blockIdx.x >> 3) & 1u)) {      // one half of the chip stores its tile while the other half streams operands
    const unsigned long long t0_ = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0_ < (unsigned long long)(CSN_NT_STAGGER)) __builtin_amdgcn_s_sleep(8);
  }
#endif
  for (unsigned lid = xcd_remap(blockIdx.x, gridDim.x); lid < ntiles; lid += gridDim.x) {
  const int64_t m0 = (int64_t)(lid / ntn) * 256, n0 = (int64_t)(lid % ntn) * BN;
  __syncthreads();     // every wave has left the previous tile's last stage

  // staging as in gemm_nt_256_kernel: a 1 KB instruction fills 8 rows x 8 chunks, chunk c of LDS row r <- global chunk
  // c ^ ((r >> 1) & 7); A has 32 instructions per stage (4 per wave), B 24 / 32 (3 / 4 per wave)
  const bf16_t* a_src[4];
  const bf16_t* b_src[NBI];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (8 * i + wave) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((row >> 1) & 7);
    int64_t am = m0 + row;
    am = am < M ? am : M - 1;
    a_src[i] = A + am * K + ch * 8;
  }
#pragma unroll
  for (int i = 0; i < NBI; ++i) {
    const int row = (8 * i + wave) * 8 + (lane >> 3);
    const int ch = (lane & 7) ^ ((row >> 1) & 7);
    b_src[i] = Bt + (n0 + row) * K + ch * 8;
  }
#if defined(CSN_NT_ROT)     // timing experiment: the column tiles that share an A panel (and the row tiles that share a B panel) start
  const int nrot_ = (int)(lid % (unsigned)nk);      // their k walk at different steps, so that they are not all on the same lines at once
#else
  const int nrot_ = 0;
#endif
  auto kpos = [&](int kt) { const int k = kt + nrot_; return (int64_t)(k >= nk ? k - nk : k) * 64; };
  auto issue_a = [&](int kt) {
    char* a_s = smem + (kt % 3) * 32768;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(a_src[i] + kpos(kt), a_s + (8 * i + wave) * 1024);
  };
  auto issue_b = [&](int kt) {
    char* b_s = smem + kBBase + (kt % 2) * kBStage;
#pragma unroll
    for (int i = 0; i < NBI; ++i) glds16(b_src[i] + kpos(kt), b_s + (8 * i + wave) * 1024);
  };

  const unsigned sw = (unsigned)((lane & 15) >> 1);
  const unsigned a_base = (unsigned)((wm * 64 + (lane & 15)) * 128) + ((((unsigned)lane >> 4) ^ sw) << 4);
  const unsigned b_base = kBBase + (unsigned)((wn * (BN / 2) + (lane & 15)) * 128) + ((((unsigned)lane >> 4) ^ sw) << 4);

  f32x4 acc[4][NBF];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < NBF; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // issue order: A(0) B(0) A(1) | step kt: B(kt+1) A(kt+2).  Behind B(kt) the wave has issued only A(kt+1) (4 pieces):
  // vmcnt(4) = "A(kt) and B(kt) have landed" (vector-memory operations retire in order)
  issue_a(0);
  issue_b(0);
  if (nk > 1) issue_a(1);
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // everybody's pieces of step kt are there; everybody has left step kt - 1
    if (kt + 1 < nk) issue_b(kt + 1);  // (slot of B(kt-1))
    if (kt + 2 < nk) issue_a(kt + 2);  // (slot of A(kt-1))
    const unsigned sa = (unsigned)(kt % 3) * 32768u, sb = (unsigned)(kt % 2) * kBStage;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const unsigned xo = kk ? 64u : 0u;
      bf16x8 af[4], bfr[NBF];
#pragma unroll
      for (int j = 0; j < NBF; ++j) bfr[j] = lds_read_b128(((b_base ^ xo) + sb) + j * 2048u);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = lds_read_b128(((a_base ^ xo) + sa) + i * 2048u);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i == 0) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else if (i == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if (i == 2) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NBF; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // epilogue from the accumulators: lane holds row m = .. + (lane & 15), columns n .. n + 3 (a store instruction covers 16
  // rows x 64 bytes)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < NBF; ++j) {
      const int64_t n = n0 + wn * (BN / 2) + j * 16 + (lane >> 4) * 4;
      f32x4 v = acc[i][j];
      if (bias) v += *reinterpret_cast<const f32x4*>(bias + n);
#if defined(CSN_NT_ABL) && CSN_NT_ABL == 1     // (ablation, timing only: no C stores unless a value is NaN)
      if (v[0] == v[0]) continue;
#endif
      if constexpr (sizeof(OutT) == 4) {
        f32x4* dst = reinterpret_cast<f32x4*>((float*)C + m * N + n);
        if (accumulate) v += *dst;
        *dst = v;
      } else {
        *reinterpret_cast<bf16x4*>((bf16_t*)C + m * N + n) = (bf16x4){(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      }
    }
  }
  }   // tile loop
}

// TN, 256 x 256 output tile, 8 waves (2 along M x 4 along N, 128 x 64 outputs per wave), LDS-DMA stages of
// 32 k-rows: per 64 contraction rows a workgroup moves 64 KB for 8.4 MFLOP -- half the L2 bytes per flop of the
// 128 x 128 tiles, whose ceiling is the aggregate L2 bandwidth (~14 TB/s measured => ~0.9 PFLOP/s).
// Stage image: 32 k-rows x 256 columns (512-byte rows, 16 blocks of 16 columns), block cb of k-row r at
// block (cb ^ s(r)) with the s(r) of tn_lds_off, so the 8 k-rows a half-wave of ds_read_b64_tr_b16 touches
// fall into 8 distinct 32-byte bank windows.
__device__ __forceinline__ int tn256_lds_off(int krow, int col) {  // byte offset; col in elements
  const int sw = (krow & 3) | (((krow >> 3) & 1) << 2);
  return krow * 512 + ((((col >> 4) ^ sw) << 4) + (col & 15)) * 2;
}
__device__ __forceinline__ bf16x8 tn256_load_frag(const char* tile, int mbase, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  union { s16x4 s[2]; bf16x8 b; } u;
  u.s[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(tile + tn256_lds_off(8 * g + q, mbase + 4 * p)));
  u.s[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) s16x4*)(tile + tn256_lds_off(8 * g + 4 + q, mbase + 4 * p)));
  return u.b;
}

// Pipeline: a ring of NSTAGE stages of 32 k-rows (A 16 KB + B 16 KB each); before the MFMAs of stage h the
// stages h+1 .. h+NSTAGE-2 are already in flight and stage h+NSTAGE-1 is issued (into the slot of stage h-1,
// which every wave has left: it passed this step's barrier).  A wave waits for its own part of stage h with a
// COUNTED vmcnt (4 LDS-DMA instructions per stage and wave) and the raw barrier then covers everybody's part;
// __syncthreads() would drain the whole ring (it waits vmcnt(0) while LDS-DMA is outstanding).
// both ds_read_b64_tr_b16 of one 16 x 32 fragment (k-rows q and q + 4 of each 8-row group), LDS byte address
// relative to the workgroup's LDS base (the dynamic array is the kernel's only LDS object, at offset 0)
__device__ __forceinline__ bf16x8 tr_read_pair(unsigned addr) {
  union { s16x4 s[2]; bf16x8 b; } u;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(u.s[0]) : "v"(addr) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:2048" : "=v"(u.s[1]) : "v"(addr) : "memory");
  return u.b;
}

template <int NSTAGE, bool COLSUM, bool STAGGER = false>
__global__ void __launch_bounds__(512)
gemm_tn_256_kernel(const bf16_t* __restrict__ A, const bf16_t* __restrict__ Bm, float* __restrict__ slabs,
                   int64_t M, int64_t N, int64_t K, int64_t k_per_split, float* __restrict__ colsum, int a_blocked) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];  // NSTAGE x (A 16 KB + B 16 KB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const unsigned ntn = (unsigned)((N + 255) / 256), ntm = (unsigned)((M + 255) / 256);
  const unsigned lid = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned zsplit = lid / (ntm * ntn), rem = lid % (ntm * ntn);
  const int64_t m0 = (int64_t)(rem / ntn) * 256, n0 = (int64_t)(rem % ntn) * 256;
  const int64_t kbeg = (int64_t)zsplit * k_per_split;
  const int64_t kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;
  const int nh = kend > kbeg ? (int)((kend - kbeg) / 32) : 0;     // stages of 32 k-rows

  // staging: instruction i (0..1) of wave w fills k-rows 2 (8 i + w), +1 of a stage (2 rows x 32 chunks = 1 KB);
  // lane -> k-row r = lane >> 5, LDS chunk position c = lane & 31 <- global columns ((c >> 1) ^ s) * 16 + (c & 1) * 8
  const bf16_t* a_src[2];
  const bf16_t* b_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int kr = 2 * (8 * i + wave) + (lane >> 5);
    const int c = lane & 31;
    const int sw = (kr & 3) | (((kr >> 3) & 1) << 2);
    const int col = (((c >> 1) ^ sw) << 4) + (c & 1) * 8;
    int64_t am = m0 + col, bn = n0 + col;
    am = am + 8 <= M ? am : M - 8;
    bn = bn + 8 <= N ? bn : N - 8;
    // (a_blocked: A[K, M] in the recurrence's fragment-major 16 x 32 blocks, blk_offset -- the same 16-byte pieces at other
    // addresses; 32 more k-rows are two row blocks = 32 M elements further, like in the row-major layout)
    a_src[i] = a_blocked ? A + blk_offset(kbeg + kr, am, M) : A + (kbeg + kr) * M + am;
    b_src[i] = Bm + (kbeg + kr) * N + bn;
  }
#if defined(CSN_TN_ROT)     // timing experiment (tools/abl_build.sh): every tile of a K slab starts its walk a few stages further on
  const int rot_ = nh > CSN_TN_ROT ? (int)(rem % (unsigned)(CSN_TN_ROT)) : 0;
#endif
  auto issue = [&](int h) {
    char* a_s = smem + (h % NSTAGE) * 32768;
    char* b_s = a_s + 16384;
#if defined(CSN_TN_ROT)
    const int hs_ = h + rot_;
    const int64_t hr = hs_ >= nh ? hs_ - nh : hs_;
#else
    const int64_t hr = h;
#endif
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      glds16(a_src[i] + hr * 32 * M, a_s + (8 * i + wave) * 1024);
      glds16(b_src[i] + hr * 32 * N, b_s + (8 * i + wave) * 1024);
    }
  };

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // optional column sums of A (the bias gradient: sum over the contraction rows of dgates): in the workgroups of
  // the first column tile every wave multiplies two of its eight A fragments by a fragment of ones too (the four
  // waves that share the A rows split the eight fragments: + 6 % MFMAs there, nothing elsewhere)
  // (a template switch: the extra accumulators cost the plain kernel 8 % even when unused)
  // (the wave index goes through readfirstlane so that the condition around the extra MFMA is a SCALAR branch:
  // under a mere EXEC mask the MFMA would still execute -- MFMA ignores EXEC -- and add foreign fragments)
  const int wn_s = __builtin_amdgcn_readfirstlane(wn);
  const bool do_colsum = COLSUM && colsum != nullptr && n0 == 0;
  const bf16x8 ones = {(bf16_t)1.0f, (bf16_t)1.0f, (bf16_t)1.0f, (bf16_t)1.0f, (bf16_t)1.0f, (bf16_t)1.0f, (bf16_t)1.0f, (bf16_t)1.0f};
  f32x4 acc_cs[2];
  acc_cs[0] = acc_cs[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // byte address (inside a stage) of this lane's first ds_read_b64_tr_b16 of every fragment; the second one
  // (k-rows + 4) is 2048 bytes further
  unsigned a_addr[8], b_addr[4];
  {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
    for (int i = 0; i < 8; ++i) a_addr[i] = (unsigned)tn256_lds_off(8 * g + q, wm * 128 + i * 16 + 4 * pp);
#pragma unroll
    for (int j = 0; j < 4; ++j) b_addr[j] = (unsigned)tn256_lds_off(8 * g + q, wn * 64 + j * 16 + 4 * pp);
  }

  if constexpr (STAGGER) {
    // Two wave groups (waves 0-3 / 4-7: one wave of each per SIMD) run the same k-steps ONE BARRIER INTERVAL apart: a
    // k-step is  A = {retire stage h+1, request stage h+3, issue the 24 fragment reads of stage h} | barrier |
    // B = {32 MFMAs} | barrier,  so while one group's waves multiply, the other group's waves on the same SIMDs issue
    // their DMA pieces and LDS reads (with one common phase per k-step both waves of a SIMD want the MFMA pipe, then
    // the LDS pipe, at the same time: MFMA busy 48 %).  Intervals I_n between barriers n and n+1: the leading group
    // does A_h in I_2h, B_h in I_2h+1, the lagging group A_h in I_2h+1, B_h in I_2h+2.
    //   read-after-DMA : a wave retires ITS pieces of stage s at the start of its A_(s-1) (counted vmcnt), i.e. before
    //                    barrier 2s-1 (leading) / 2s (lagging); the first read of stage s is in I_2s.
    //   DMA-after-read : the last reads of stage h retire inside the lagging group's B_h (I_2h+2); its slot takes
    //                    stage h+5, requested in A_(h+2) = I_2h+4 / I_2h+5.  Hence 5 stages, prefetch distance 3.
    static_assert(!STAGGER || NSTAGE == 5, "staggered form: ring of 5 stages");
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
#pragma unroll
    for (int h = 0; h < 3; ++h)
      if (h < nh) issue(h);
    if (nh >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (nh == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int h = 0; h < nh; ++h) {
      if (h + 2 < nh) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !defined(CSN_TN_ABL) || CSN_TN_ABL != 1     // (ablation 1, timing only: no DMA after the prologue)
      if (h + 3 < nh) issue(h + 3);
#endif
      const unsigned sb = (unsigned)(h % NSTAGE) * 32768u;
      bf16x8 af[8], bfr[4];
#if defined(CSN_TN_ABL) && CSN_TN_ABL == 2       // (ablation 2, timing only: no LDS reads)
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = __builtin_bit_cast(bf16x8, (f32x4){(float)sb, 1.f, 2.f, (float)j});
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = __builtin_bit_cast(bf16x8, (f32x4){(float)sb, 3.f, 2.f, (float)i});
#else
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = tr_read_pair(b_addr[j] + sb + 16384u);
#pragma unroll
      for (int i = 0; i < 8; ++i) af[i] = tr_read_pair(a_addr[i] + sb);
#endif
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // reads behind the ones MFMA group i needs: the pairs of A fragments i+1 .. 7
        if (i == 0) asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory");
        else if (i == 1) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
        else if (i == 2) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
        else if (i == 3) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        else if (i == 4) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
        else if (i == 5) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if (i == 6) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        if constexpr (COLSUM)
          if (do_colsum && (i >> 1) == wn_s) acc_cs[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], acc_cs[i & 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
  } else {
#pragma unroll
  for (int h = 0; h < NSTAGE - 1; ++h)
    if (h < nh) issue(h);
  for (int h = 0; h < nh; ++h) {
    // stages issued after h and still allowed in flight: min(NSTAGE - 2, nh - 1 - h)
    const int ahead = nh - 1 - h;
    if (ahead >= NSTAGE - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (NSTAGE - 2)) : "memory");
    else if (NSTAGE > 3 && ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (h + NSTAGE - 1 < nh) issue(h + NSTAGE - 1);
    // Fragment reads in inline asm: for an LDS read it can see, the compiler waits vmcnt(0) while ANY LDS-DMA is
    // outstanding (it cannot tell the stages apart), which would drain the ring every step.  LDS operations
    // complete in order, so counted lgkmcnt waits release the MFMAs of A tile i while tiles i+1, i+2 are in flight.
    const unsigned sb = (unsigned)(h % NSTAGE) * 32768u;
    bf16x8 af[8], bfr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bfr[j] = tr_read_pair(b_addr[j] + sb + 16384u);
    af[0] = tr_read_pair(a_addr[0] + sb);
    af[1] = tr_read_pair(a_addr[1] + sb);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i + 2 < 8) {
        af[i + 2] = tr_read_pair(a_addr[i + 2] + sb);
        asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      } else if (i + 1 < 8) {
        asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef CSN_TN_SETPRIO
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
#ifdef CSN_TN_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
      if constexpr (COLSUM)
        if (do_colsum && (i >> 1) == wn_s) acc_cs[i & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[i], acc_cs[i & 1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  }

  if (COLSUM && do_colsum && (lane >> 4) == 0) {       // D[n][m]: every row n holds the same sum; lanes 0..15 write column m
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int64_t m = m0 + wm * 128 + (2 * wn + e) * 16 + (lane & 15);
      if (m < M) colsum[(int64_t)zsplit * M + m] = acc_cs[e][0];
    }
  }

  float* C = slabs + (int64_t)zsplit * M * N;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t m = m0 + wm * 128 + i * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
      if (n + 3 < N && (N & 3) == 0) {
        *reinterpret_cast<float4*>(C + m * N + n) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      } else {
        for (int r = 0; r < 4; ++r)
          if (n + r < N) C[m * N + n + r] = acc[i][j][r];
      }
    }
  }
}

static int tn_splits(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  int64_t s = (1024 + tiles - 1) / tiles;          // aim at >= 4 workgroups per CU
  const int64_t max_by_k = (K + 511) / 512;        // keep >= 512 contraction rows per split
  if (s > max_by_k) s = max_by_k;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}
// the 256 x 256 kernel: one workgroup per CU (128 KB of LDS), so the K splits fill the 256 CUs once
static bool tn_use_256(int64_t M, int64_t N, int64_t K, const Options& opt) {
  // (N >= 128: at N = 128 -- the layer-0 input weight gradient -- half of every tile's columns are clamped duplicates
  // whose results are discarded, and it still beats the 128 x 128 register-staged kernel)
  return M >= 256 && N >= 128 && K % 64 == 0 && K >= 8192 && !opt.gemm_no_dma && !opt.gemm_no_256;
}
static int tn_splits_256(int64_t M, int64_t N, int64_t K) {
  const int64_t tiles = ((M + 255) / 256) * ((N + 255) / 256);
  int64_t s = 256 / tiles;
  const int64_t max_by_k = K / 2048;               // keep >= 32 k-steps per split
  if (s > max_by_k) s = max_by_k;
  if (s < 1) s = 1;
  if (s > 64) s = 64;
  return (int)s;
}

int gemm_nt(const void* A, const void* Bt, const float* bias, void* C, int64_t M, int64_t N, int64_t K, int dtype,
            int out_dtype, int accumulate, hipStream_t st, const Options& opt) {
  CSN_REQUIRE(A && Bt && C, "csn_gemm_nt: null pointer");
  CSN_REQUIRE(M > 0 && N > 0 && K > 0, "csn_gemm_nt: bad shape M=%lld N=%lld K=%lld", (long long)M, (long long)N,
              (long long)K);
  CSN_REQUIRE(dtype == CSN_F32 || dtype == CSN_BF16, "csn_gemm_nt: bad dtype %d", dtype);
  CSN_REQUIRE(out_dtype == CSN_F32 || out_dtype == CSN_BF16, "csn_gemm_nt: bad out_dtype %d", out_dtype);
  CSN_REQUIRE(!(accumulate && out_dtype != CSN_F32), "csn_gemm_nt: accumulate needs a float32 C");
  const bool fast = dtype == CSN_BF16 && (K % 8 == 0) && (N % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(Bt) |
                      reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(bias)) & 15) == 0 &&
                    !opt.gemm_generic;
  if (!fast)
    return launch_generic(A, K, 1, Bt, 1, K, bias, C, N, M, N, K, dtype, out_dtype, accumulate, 1, 0, st);
  // 256 x 128 tiles where the tile count still fills the chip a few times over (measured at the LSTM's chunk
  // shapes: 58 vs 67 us at N = 3072, 56 vs 51 us at N = 768)
  if (K % 64 == 0 && K >= 128 && M >= 256 && !opt.gemm_no_dma && !opt.gemm_no_256 && !opt.gemm_no_192) {
    // 256 x 256 or 256 x 192 tiles where they give every CU whole tiles: 33 % / 21 % fewer operand bytes per flop than
    // 256 x 128 (measured at 8192 x 3072 x 768: 53.2 vs 59.0 us, 39.4 vs 47.2 without the C stores; at 8192 x 768 x 3072
    // -- 128 tiles of 192 -- 66 vs 47: hence "at least one tile per CU, and at most 10 % of the last round empty")
    auto fills = [&](int bn) {
      if (N % bn != 0) return false;
      const int64_t t = (N / bn) * ((M + 255) / 256), rounds = (t + 255) / 256;
      return t >= 256 && t * 10 >= rounds * 256 * 9;
    };
    const int bn = fills(256) ? 256 : (fills(192) ? 192 : 0);
    if (bn != 0) {
      const int lds = 3 * 32768 + 2 * bn * 128;
      dim3 gridw((unsigned)((N / bn) * ((M + 255) / 256)));
      if (bn == 256) {
        if (int rc = ensure_dyn_lds<&gemm_nt_wide_kernel<bf16_t, 256>>(lds)) return rc;
        if (int rc = ensure_dyn_lds<&gemm_nt_wide_kernel<float, 256>>(lds)) return rc;
        if (out_dtype == CSN_BF16)
          gemm_nt_wide_kernel<bf16_t, 256><<<gridw, 512, lds, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (bf16_t*)C, M, N, K, 0);
        else
          gemm_nt_wide_kernel<float, 256><<<gridw, 512, lds, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (float*)C, M, N, K, accumulate);
      } else {
        if (int rc = ensure_dyn_lds<&gemm_nt_wide_kernel<bf16_t, 192>>(lds)) return rc;
        if (int rc = ensure_dyn_lds<&gemm_nt_wide_kernel<float, 192>>(lds)) return rc;
        if (out_dtype == CSN_BF16)
          gemm_nt_wide_kernel<bf16_t, 192><<<gridw, 512, lds, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (bf16_t*)C, M, N, K, 0);
        else
          gemm_nt_wide_kernel<float, 192><<<gridw, 512, lds, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (float*)C, M, N, K, accumulate);
      }
      CSN_LAUNCH_CHECK();
      return CSN_OK;
    }
  }
  if (K % 64 == 0 && K >= 256 && M >= 256 && N >= 1024 && !opt.gemm_no_dma && !opt.gemm_no_256) {
    if (int rc = ensure_dyn_lds<&gemm_nt_256_kernel<bf16_t, 3>>(3 * 49152)) return rc;
    if (int rc = ensure_dyn_lds<&gemm_nt_256_kernel<float, 3>>(3 * 49152)) return rc;
    dim3 grid256((unsigned)(((N + 127) / 128) * ((M + 255) / 256)));
    if (out_dtype == CSN_BF16)
      gemm_nt_256_kernel<bf16_t, 3><<<grid256, 512, 3 * 49152, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (bf16_t*)C, M, N, K, 0);
    else
      gemm_nt_256_kernel<float, 3><<<grid256, 512, 3 * 49152, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (float*)C, M, N, K, accumulate);
    CSN_LAUNCH_CHECK();
    return CSN_OK;
  }
  dim3 grid((unsigned)(((N + 127) / 128) * ((M + 127) / 128)));
  if (K % 64 == 0 && !opt.gemm_no_dma) {
    if (int rc = ensure_dyn_lds<&gemm_nt_dma_kernel<bf16_t>>(98304)) return rc;
    if (int rc = ensure_dyn_lds<&gemm_nt_dma_kernel<float>>(98304)) return rc;
    if (out_dtype == CSN_BF16)
      gemm_nt_dma_kernel<bf16_t><<<grid, 256, 65536, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (bf16_t*)C, M, N, K, 0);
    else
      gemm_nt_dma_kernel<float><<<grid, 256, 65536, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (float*)C, M, N, K, accumulate);
    CSN_LAUNCH_CHECK();
    return CSN_OK;
  }
  const bool two = opt.gemm_lds64;
  if (out_dtype == CSN_BF16) {
    if (two) gemm_nt_bf16_kernel<bf16_t, 2><<<grid, 256, 65536, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (bf16_t*)C, M, N, K, 0);
    else gemm_nt_bf16_kernel<bf16_t, 1><<<grid, 256, 32768, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (bf16_t*)C, M, N, K, 0);
  } else {
    if (two) gemm_nt_bf16_kernel<float, 2><<<grid, 256, 65536, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (float*)C, M, N, K, accumulate);
    else gemm_nt_bf16_kernel<float, 1><<<grid, 256, 32768, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, (float*)C, M, N, K, accumulate);
  }
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

}  // namespace csn

using namespace csn;

extern "C" int csn_gemm_nt(const void* A, const void* Bt, const float* bias, void* C, int64_t M, int64_t N, int64_t K,
                           int dtype, int out_dtype, int accumulate, csnStream_t stream) {
  return gemm_nt(A, Bt, bias, C, M, N, K, dtype, out_dtype, accumulate, as_stream(stream), options_from_env());
}

namespace csn {
// NT GEMM on at most `max_wgs` workgroups, each alone on its CU (96 KB of LDS requested), walking the 128 x 128
// tiles: runs beside a persistent LSTM launch on the CUs that launch leaves idle.  bf16 operands, K % 64 == 0.
int launch_gemm_nt_beside(const void* A, const void* Bt, const float* bias, float* C, int64_t M, int64_t N, int64_t K,
                          int max_wgs, hipStream_t st) {
  CSN_REQUIRE(K % 64 == 0 && N % 4 == 0 && max_wgs >= 1, "launch_gemm_nt_beside: unsupported shape");
  if (int rc = ensure_dyn_lds<&gemm_nt_256_kernel<float, 3>>(3 * 49152)) return rc;
  if (int rc = ensure_dyn_lds<&gemm_nt_dma_kernel<float>>(98304)) return rc;
  if (M >= 256 && N >= 128 && K >= 256) {
    // 256 x 128 tiles, 8 waves, 144 KB of LDS: alone on a CU this kernel runs at twice the rate of the 4-wave one
    const int64_t tiles = ((N + 127) / 128) * ((M + 255) / 256);
    dim3 grid((unsigned)(tiles < max_wgs ? tiles : max_wgs));
    gemm_nt_256_kernel<float, 3><<<grid, 512, 3 * 49152, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, C, M, N, K, 0);
  } else {
    const int64_t tiles = ((N + 127) / 128) * ((M + 127) / 128);
    dim3 grid((unsigned)(tiles < max_wgs ? tiles : max_wgs));
    gemm_nt_dma_kernel<float><<<grid, 256, 98304, st>>>((const bf16_t*)A, (const bf16_t*)Bt, bias, C, M, N, K, 0);
  }
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}
}  // namespace csn

namespace csn {
size_t gemm_tn_scratch_bytes(int64_t M, int64_t N, int64_t K, const Options& opt) {
  if (M <= 0 || N <= 0 || K <= 0) return 0;
  const int a = tn_splits(M, N, K), b = tn_use_256(M, N, K, opt) ? tn_splits_256(M, N, K) : 0;
  return (size_t)(a > b ? a : b) * (size_t)M * (size_t)N * sizeof(float);
}
}  // namespace csn

extern "C" size_t csn_gemm_tn_scratch_bytes(int64_t M, int64_t N, int64_t K) {
  return gemm_tn_scratch_bytes(M, N, K, options_from_env());
}

namespace csn {
// colsum (optional): device buffer of at least 64 * M floats; if the kernel that runs can produce the column sums
// of A on the way (one partial row of M values per K split), *colsum_done is set to 1.
bool gemm_tn_takes_blocked_a(int64_t M, int64_t N, int64_t K, const Options& opt) {
  return (M % 32 == 0) && (N % 8 == 0) && (K % 16 == 0) && tn_use_256(M, N, K, opt);
}
int launch_gemm_tn_slabs(const void* A, const void* B, float* slabs, int64_t M, int64_t N, int64_t K, int dtype,
                         hipStream_t st, int* S_out, float* colsum, int* colsum_done, const Options& opt, int a_blocked) {
  if (colsum_done) *colsum_done = 0;
  CSN_REQUIRE(!a_blocked || (dtype == CSN_BF16 && gemm_tn_takes_blocked_a(M, N, K, opt)),
              "launch_gemm_tn_slabs: a fragment-major A needs the 256 x 256 kernel");
  const bool aligned16 = ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0;
  if (dtype == CSN_BF16 && (M % 8 == 0) && (N % 8 == 0) && aligned16 && tn_use_256(M, N, K, opt)) {
    const int S2 = tn_splits_256(M, N, K);
    int64_t kper2 = (K + S2 - 1) / S2;
    kper2 = (kper2 + 63) / 64 * 64;
    *S_out = S2;
    if (int rc = ensure_dyn_lds<&gemm_tn_256_kernel<5, false>>(5 * 32768)) return rc;
    if (int rc = ensure_dyn_lds<&gemm_tn_256_kernel<4, false>>(4 * 32768)) return rc;
    if (int rc = ensure_dyn_lds<&gemm_tn_256_kernel<3, false>>(3 * 32768)) return rc;
    if (int rc = ensure_dyn_lds<&gemm_tn_256_kernel<4, true>>(4 * 32768)) return rc;
    dim3 grid((unsigned)(((N + 255) / 256) * ((M + 255) / 256) * S2));
    const int nst = opt.tn_stages;
    const bf16_t* Ab = (const bf16_t*)A;
    const bf16_t* Bb = (const bf16_t*)B;
    if (!opt.tn_no_stagger && nst == 4) {      // (the default; CSN_TN_STAGES selects one of the single-phase rings)
      if (int rc = ensure_dyn_lds<&gemm_tn_256_kernel<5, false, true>>(5 * 32768)) return rc;
      if (int rc = ensure_dyn_lds<&gemm_tn_256_kernel<5, true, true>>(5 * 32768)) return rc;
      if (colsum) gemm_tn_256_kernel<5, true, true><<<grid, 512, 5 * 32768, st>>>(Ab, Bb, slabs, M, N, K, kper2, colsum, a_blocked);
      else gemm_tn_256_kernel<5, false, true><<<grid, 512, 5 * 32768, st>>>(Ab, Bb, slabs, M, N, K, kper2, nullptr, a_blocked);
    } else if (colsum) gemm_tn_256_kernel<4, true><<<grid, 512, 4 * 32768, st>>>(Ab, Bb, slabs, M, N, K, kper2, colsum, a_blocked);
    else if (nst == 3) gemm_tn_256_kernel<3, false><<<grid, 512, 3 * 32768, st>>>(Ab, Bb, slabs, M, N, K, kper2, nullptr, a_blocked);
    else if (nst == 5) gemm_tn_256_kernel<5, false><<<grid, 512, 5 * 32768, st>>>(Ab, Bb, slabs, M, N, K, kper2, nullptr, a_blocked);
    else gemm_tn_256_kernel<4, false><<<grid, 512, 4 * 32768, st>>>(Ab, Bb, slabs, M, N, K, kper2, nullptr, a_blocked);
    if (colsum && colsum_done) *colsum_done = 1;
    CSN_LAUNCH_CHECK();
    return CSN_OK;
  }
  const int S = tn_splits(M, N, K);
  int64_t kper = (K + S - 1) / S;
  kper = (kper + 63) / 64 * 64;
  *S_out = S;
  const bool fast = dtype == CSN_BF16 && (M % 8 == 0) && (N % 8 == 0) &&
                    ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0 &&
                    !opt.gemm_generic;
  if (fast) {
    dim3 grid((unsigned)(((N + 127) / 128) * ((M + 127) / 128) * S));
    if (opt.tn_no_tr)
      gemm_tn_bf16_kernel<false><<<grid, 256, 0, st>>>((const bf16_t*)A, (const bf16_t*)B, slabs, M, N, K, kper);
    else
      gemm_tn_bf16_kernel<true><<<grid, 256, 0, st>>>((const bf16_t*)A, (const bf16_t*)B, slabs, M, N, K, kper);
    CSN_LAUNCH_CHECK();
    return CSN_OK;
  }
  // A(m,k) = A[k*M + m], B(k,n) = B[k*N + n]
  return launch_generic(A, 1, M, B, N, 1, nullptr, slabs, N, M, N, K, dtype, CSN_F32, 0, S, M * N, st);
}
}  // namespace csn

extern "C" int csn_gemm_tn(const void* A, const void* B, float* C, int64_t M, int64_t N, int64_t K, int dtype,
                           void* scratch, csnStream_t stream) {
  CSN_REQUIRE(A && B && C && scratch, "csn_gemm_tn: null pointer");
  CSN_REQUIRE(M > 0 && N > 0 && K > 0, "csn_gemm_tn: bad shape");
  CSN_REQUIRE(dtype == CSN_F32 || dtype == CSN_BF16, "csn_gemm_tn: bad dtype %d", dtype);
  hipStream_t st = as_stream(stream);
  int S = 1;
  if (int rc = launch_gemm_tn_slabs(A, B, (float*)scratch, M, N, K, dtype, st, &S, nullptr, nullptr, options_from_env(), 0)) return rc;
  return launch_reduce_slabs((const float*)scratch, M * N, S, C, M * N, 0, st);
}
