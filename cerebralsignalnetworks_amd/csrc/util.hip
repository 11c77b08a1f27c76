// Error plumbing + small memory-bound helper kernels (casts, transposes, slab reduction,
// column sums).  All are HBM-bound streaming kernels: 16-B accesses where alignment allows,
// grids capped at 2048 blocks with grid-stride loops.
#include <stdarg.h>

#include "csn_common.h"

namespace csn {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }

int fail(int status, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return status;
}

static inline unsigned capped_grid(int64_t work_items, int per_block) {
  int64_t g = (work_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return (unsigned)g;
}

// ---- cast (dense) -------------------------------------------------------------------------
template <typename T>
__global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n / 4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    dst[i * 4 + 0] = from_f32<T>(v.x);
    dst[i * 4 + 1] = from_f32<T>(v.y);
    dst[i * 4 + 2] = from_f32<T>(v.z);
    dst[i * 4 + 3] = from_f32<T>(v.w);
  }
  for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = from_f32<T>(src[i]);
}

int launch_cast(const float* src, void* dst, int64_t n, int dtype, hipStream_t st) {
  if (n <= 0) return CSN_OK;
  const unsigned grid = capped_grid(n / 4 + 1, 256);
  if (dtype == CSN_BF16) cast_kernel<bf16_t><<<grid, 256, 0, st>>>(src, (bf16_t*)dst, n);
  else cast_kernel<float><<<grid, 256, 0, st>>>(src, (float*)dst, n);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

template <typename T>
__global__ void upcast_kernel(const T* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = to_f32(src[i]);
}

int launch_upcast(const void* src, int dtype, float* dst, int64_t n, hipStream_t st) {
  if (n <= 0) return CSN_OK;
  const unsigned grid = capped_grid(n, 256);
  if (dtype == CSN_BF16) upcast_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)src, dst, n);
  else upcast_kernel<float><<<grid, 256, 0, st>>>((const float*)src, dst, n);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

// ---- strided cast: dst[(i1*n0 + i0)*n2 + i2] = src[i0*s0 + i1*s1 + i2] ----------------------
// Used for x[b][t][i] (strides s0=b, s1=t) -> time-major [T][B][I] in the compute dtype.
template <typename T>
__global__ void cast_strided_kernel(const float* __restrict__ src, int64_t s0, int64_t s1, int64_t n0, int64_t n1,
                                    int64_t n2, T* __restrict__ dst) {
  const int64_t total = n0 * n1 * n2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t i2 = i % n2;
    const int64_t r = i / n2;       // r = i1*n0 + i0
    const int64_t i0 = r % n0, i1 = r / n0;
    dst[i] = from_f32<T>(src[i0 * s0 + i1 * s1 + i2]);
  }
}

int launch_cast_strided(const float* src, int64_t s0, int64_t s1, int64_t n0, int64_t n1, int64_t n2, void* dst,
                        int dtype, hipStream_t st) {
  const int64_t total = n0 * n1 * n2;
  if (total <= 0) return CSN_OK;
  const unsigned grid = capped_grid(total, 256);
  if (dtype == CSN_BF16)
    cast_strided_kernel<bf16_t><<<grid, 256, 0, st>>>(src, s0, s1, n0, n1, n2, (bf16_t*)dst);
  else
    cast_strided_kernel<float><<<grid, 256, 0, st>>>(src, s0, s1, n0, n1, n2, (float*)dst);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

// ---- transpose + cast: dst[c*R + r] = src[r*C + c] ------------------------------------------
template <typename T>
__global__ void transpose_cast_kernel(const float* __restrict__ src, int64_t R, int64_t C, T* __restrict__ dst) {
  __shared__ float tile[32][33];
  const int64_t r0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int j = ty; j < 32; j += 8) {
    const int64_t r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < R && c < C) ? src[r * C + c] : 0.0f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    const int64_t c = c0 + j, r = r0 + tx;
    if (r < R && c < C) dst[c * R + r] = from_f32<T>(tile[tx][j]);
  }
}

int launch_transpose_cast(const float* src, int64_t R, int64_t C, void* dst, int dtype, hipStream_t st) {
  dim3 grid((unsigned)((C + 31) / 32), (unsigned)((R + 31) / 32));
  if (dtype == CSN_BF16) transpose_cast_kernel<bf16_t><<<grid, 256, 0, st>>>(src, R, C, (bf16_t*)dst);
  else transpose_cast_kernel<float><<<grid, 256, 0, st>>>(src, R, C, (float*)dst);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

__global__ void add_vec_kernel(const float* a, const float* b, float* out, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = a[i] + b[i];
}

int launch_add_vec(const float* a, const float* b, float* out, int64_t n, hipStream_t st) {
  add_vec_kernel<<<capped_grid(n, 256), 256, 0, st>>>(a, b, out, n);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

// ---- slab reduction (split-K combine), fixed summation order => bitwise reproducible ---------
__global__ void reduce_slabs_kernel(const float* __restrict__ slabs, int64_t stride_s, int S, float* __restrict__ out,
                                    int64_t n, int accumulate) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float acc = accumulate ? out[i] : 0.0f;
    for (int s = 0; s < S; ++s) acc += slabs[(int64_t)s * stride_s + i];
    out[i] = acc;
  }
}

int launch_reduce_slabs(const float* slabs, int64_t stride, int S, float* out, int64_t n, int accumulate,
                        hipStream_t st) {
  reduce_slabs_kernel<<<capped_grid(n, 256), 256, 0, st>>>(slabs, stride, S, out, n, accumulate);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

// ---- column sums of X[R,N]: stage 1 = partial sums over row chunks, stage 2 = slab reduce ----
static const int kColsumChunks = 128;
size_t colsum_scratch_bytes(int64_t N) { return (size_t)kColsumChunks * (size_t)N * sizeof(float); }

// 256 threads = 32 column groups (8 columns = one 16-byte load for bf16) x 8 row lanes; a block
// covers 256 columns of one row chunk, streams its rows with 16-byte loads, then folds the 8 row
// lanes through LDS in a fixed order.
template <typename T>
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const T* __restrict__ X, int64_t R, int64_t N, float* __restrict__ partial) {
  __shared__ float sh[8][256 + 1];
  const int cg = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int64_t col0 = (int64_t)blockIdx.x * 256 + cg * 8;
  const int64_t rows_per = (R + gridDim.y - 1) / gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per;
  const int64_t r1 = (r0 + rows_per < R) ? r0 + rows_per : R;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
  if (col0 + 7 < N && (N % 8 == 0)) {
    for (int64_t r = r0 + rl; r < r1; r += 8) {
      if constexpr (sizeof(T) == 2) {
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(X + r * N + col0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
      } else {
        const float4 a = *reinterpret_cast<const float4*>(X + r * N + col0);
        const float4 b = *reinterpret_cast<const float4*>(X + r * N + col0 + 4);
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
      }
    }
  } else {
    for (int64_t r = r0 + rl; r < r1; r += 8)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (col0 + j < N) acc[j] += to_f32(X[r * N + col0 + j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) sh[rl][cg * 8 + j] = acc[j];
  __syncthreads();
  const int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (col < N) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += sh[k][threadIdx.x];
    partial[(int64_t)blockIdx.y * N + col] = t;
  }
}

int colsum_chunks() { return kColsumChunks; }

int launch_colsum_partial(const void* X, int64_t R, int64_t N, int dtype, void* scratch, hipStream_t st) {
  dim3 grid((unsigned)((N + 255) / 256), kColsumChunks);
  float* partial = (float*)scratch;
  if (dtype == CSN_BF16) colsum_partial_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)X, R, N, partial);
  else colsum_partial_kernel<float><<<grid, 256, 0, st>>>((const float*)X, R, N, partial);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

int launch_colsum(const void* X, int64_t R, int64_t N, int dtype, float* out, void* scratch, hipStream_t st) {
  if (int rc = launch_colsum_partial(X, R, N, dtype, scratch, st)) return rc;
  return launch_reduce_slabs((const float*)scratch, N, kColsumChunks, out, N, 0, st);
}

}  // namespace csn

extern "C" int csn_abi_version(void) { return CSN_ABI_VERSION; }
extern "C" const char* csn_last_error(void) { return csn::g_last_error.c_str(); }
extern "C" const char* csn_target_arch(void) { return "gfx950"; }
