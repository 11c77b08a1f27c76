// Shared device/host helpers for libcsn_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/csn_hip.h"

namespace csn {

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;

// ---- host-side error plumbing -------------------------------------------------------
void set_error(const std::string& msg);
int fail(int status, const char* fmt, ...);

#define CSN_HIP_CHECK(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess)                                                                \
      return ::csn::fail(CSN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                            \
  } while (0)

#define CSN_LAUNCH_CHECK()                                                               \
  do {                                                                                   \
    hipError_t _e = hipGetLastError();                                                   \
    if (_e != hipSuccess)                                                                \
      return ::csn::fail(CSN_ERR_HIP, "kernel launch failed: %s (%s:%d)",                \
                         hipGetErrorString(_e), __FILE__, __LINE__);                     \
  } while (0)

#define CSN_REQUIRE(cond, ...)                                                           \
  do {                                                                                   \
    if (!(cond)) return ::csn::fail(CSN_ERR_INVALID_ARGUMENT, __VA_ARGS__);              \
  } while (0)

static inline hipStream_t as_stream(csnStream_t s) { return reinterpret_cast<hipStream_t>(s); }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline size_t dtype_size(int dtype) { return dtype == CSN_BF16 ? 2 : 4; }

// ---- device helpers --------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE, v_cvt_pk_bf16_f32

// Accurate logistic / tanh in f32 (parity with ATen's CPU kernels to ~1e-7).
__device__ __forceinline__ float sigmoid_f32(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanh_f32(float x) { return tanhf(x); }

// ---- internal launchers shared across translation units --------------------------------
// out[r*ldo + c] = (T)in[strided]; generic strided cast used for the time-major input copy.
int launch_cast_strided(const float* src, int64_t s0, int64_t s1, int64_t n0, int64_t n1, int64_t n2,
                        void* dst, int dtype, hipStream_t st);
// dst[c*R + r] = (T)src[r*C + c]
int launch_transpose_cast(const float* src, int64_t R, int64_t C, void* dst, int dtype, hipStream_t st);
// dst = (T)src, n elements
int launch_cast(const float* src, void* dst, int64_t n, int dtype, hipStream_t st);
// dst(f32) = (float)src(T)
int launch_upcast(const void* src, int dtype, float* dst, int64_t n, hipStream_t st);
int launch_add_vec(const float* a, const float* b, float* out, int64_t n, hipStream_t st);
// out[i] = (accumulate ? out[i] : 0) + sum_s slabs[s*stride + i]
int launch_reduce_slabs(const float* slabs, int64_t stride, int S, float* out, int64_t n, int accumulate, hipStream_t st);
// out[N] = column sums of X[R,N] (dtype), deterministic; scratch >= colsum_scratch_bytes(N)
size_t colsum_scratch_bytes(int64_t N);
int launch_colsum(const void* X, int64_t R, int64_t N, int dtype, float* out, void* scratch, hipStream_t st);
int colsum_chunks();
int launch_colsum_partial(const void* X, int64_t R, int64_t N, int dtype, void* scratch, hipStream_t st);
// split-K slabs of C[M,N] = A[K,M]^T B[K,N] into `slabs` ([S][M*N] f32); returns S through S_out
int launch_gemm_nt_beside(const void* A, const void* Bt, const float* bias, float* C, int64_t M, int64_t N, int64_t K,
                          int max_wgs, hipStream_t st);
int launch_gemm_tn_slabs(const void* A, const void* B, float* slabs, int64_t M, int64_t N, int64_t K, int dtype,
                         hipStream_t st, int* S_out, float* colsum, int* colsum_done);

}  // namespace csn
