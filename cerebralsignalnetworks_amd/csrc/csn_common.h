// Shared device/host helpers for libcsn_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <atomic>
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

// ---- diagnostic switches --------------------------------------------
// None is needed in production.  They are read from the environment into this struct -- once per LSTM plan
// (csn_lstm_plan_create) or once per call of a stateless entry point -- and passed down by reference: the
// library keeps no mutable global state, so plans on different streams / threads / devices never share any.
struct Options {
  bool cell_v1, no_persist, no_persist_bwd, persist_streams, no_xcd_local, no_rotate, no_fuse_x, no_beside,
      no_side_stream, gemm_slot, fwd_ksplit, fwd_nsplit, fwd_halves, fwd_ws, fwd_flags, bwd_flags, dpoll_no_hint, fwd_hint, beside_fwd, xproj_bf16, wgrad_overlap, gemm_no_dma, gemm_no_256, gemm_generic, gemm_lds64, tn_no_tr, tn_no_stagger, filter_v1, tags_no_rearm, bwd_single_copy, gemm_no_192;
  int chunk;       // timesteps per weight-stationary launch
  int tn_stages;   // LDS-DMA ring depth of the 256 x 256 weight-gradient kernel
  int fwd_nk;
};
static inline Options options_from_env() {
  auto on = [](const char* name) { return getenv(name) != nullptr; };
  auto num = [](const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
  };
  Options o{};
  o.cell_v1 = on("CSN_CELL_V1");
  o.no_persist = on("CSN_NO_PERSIST");
  o.no_persist_bwd = on("CSN_NO_PERSIST_BWD");
  o.persist_streams = on("CSN_PERSIST_STREAMS");
  o.no_xcd_local = on("CSN_NO_XCD_LOCAL");
  o.no_rotate = on("CSN_NO_ROTATE");
  o.no_fuse_x = on("CSN_NO_FUSE_X");
  o.no_beside = on("CSN_NO_BESIDE");
  o.no_side_stream = on("CSN_NO_SIDE_STREAM");
  o.gemm_slot = on("CSN_GEMM_SLOT");
  o.fwd_ksplit = on("CSN_FWD_KSPLIT");
  o.fwd_nsplit = on("CSN_FWD_NSPLIT");
#ifdef CSN_EXPERIMENTS      // losing variants: only lib/libcsn_hip_experiments.so (`make experiments`) honours these
  o.fwd_halves = on("CSN_FWD_HALVES");
  o.fwd_ws = on("CSN_FWD_WS");
  o.beside_fwd = on("CSN_BESIDE_FWD");
  o.xproj_bf16 = on("CSN_XPROJ_BF16");
  o.wgrad_overlap = on("CSN_WGRAD_OVERLAP");
  o.bwd_single_copy = on("CSN_BWD_SINGLE_COPY");
#endif
  o.fwd_flags = on("CSN_FWD_FLAGS");
  o.bwd_flags = on("CSN_BWD_FLAGS");
  o.dpoll_no_hint = on("CSN_DPOLL_NO_HINT");
  o.fwd_hint = on("CSN_FWD_HINT");
  o.tn_no_stagger = on("CSN_TN_NO_STAGGER");
  o.gemm_no_dma = on("CSN_GEMM_NO_DMA");
  o.gemm_no_256 = on("CSN_GEMM_NO_256");
  o.gemm_no_192 = on("CSN_GEMM_NO_192");
  o.gemm_generic = on("CSN_GEMM_GENERIC");
  o.gemm_lds64 = on("CSN_GEMM_LDS64");
  o.tn_no_tr = on("CSN_TN_NO_TR");
  o.filter_v1 = on("CSN_FILTER_V1");
#ifdef CSN_SLAB_TAGS
  o.tags_no_rearm = on("CSN_TAGS_NO_REARM");     // fault injection of the debug library (lstm_cell_blk.h)
#endif
  o.chunk = num("CSN_LSTM_CHUNK", 32);
  if (o.chunk < 1) o.chunk = 1;
  o.tn_stages = num("CSN_TN_STAGES", 4);
  o.fwd_nk = num("CSN_FWD_NK", 1);
  return o;
}

static inline hipStream_t as_stream(csnStream_t s) { return reinterpret_cast<hipStream_t>(s); }
// A kernel that uses more than 64 KB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize raised once
// per DEVICE.  One bit per device in a per-kernel atomic word: idempotent, so a race between two threads only
// sets the attribute twice.
template <auto Kernel>
static int ensure_dyn_lds(int bytes) {
  static std::atomic<unsigned> done{0u};
  int dev = 0;
  CSN_HIP_CHECK(hipGetDevice(&dev));
  const unsigned bit = 1u << (dev & 31);
  if ((done.load(std::memory_order_acquire) & bit) == 0u) {
    CSN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.fetch_or(bit, std::memory_order_release);
  }
  return CSN_OK;
}

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

// ---- generic / exact-f32 cell kernels (lstm_cell.hip) ------------------------------------
// One launch advances up to 4 independent cell problems (blockIdx.z): the layers of a wavefront diagonal of the generic /
// exact-f32 path (lstm.hip: forward_v1 / backward_v1) -- one launch boundary per timestep instead of one per layer-step.
struct CellFwdOne {
  const void* h_prev; const void* w_hh; const float* xproj; int64_t xproj_ld; const float* c_prev;
  void* gates_out; float* c_out; void* h_out;
};
struct CellFwdBatch { CellFwdOne p[4]; };
struct CellBwdOne {
  const void* dg_next; const void* w_hh_t; const float* dy; int64_t dy_ld; const void* gates; const float* c;
  const float* c_prev; float* dc_carry; void* dg_out;
};
struct CellBwdBatch { CellBwdOne p[4]; };
int launch_cell_fwd_batch(const CellFwdBatch& b, int np, int B, int H, int dtype, hipStream_t st);
int launch_cell_bwd_batch(const CellBwdBatch& b, int np, int B, int H, int dtype, hipStream_t st);

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
                         hipStream_t st, int* S_out, float* colsum, int* colsum_done, const Options& opt, int a_blocked = 0);
// may A[K, M] of that call be handed over in the fragment-major block layout of the recurrence (blk_offset)?
bool gemm_tn_takes_blocked_a(int64_t M, int64_t N, int64_t K, const Options& opt);
// C[M,N] (+)= A[M,K] Bt[N,K]^T (+ bias): the body of csn_gemm_nt with the switches passed in
int gemm_nt(const void* A, const void* Bt, const float* bias, void* C, int64_t M, int64_t N, int64_t K, int dtype,
            int out_dtype, int accumulate, hipStream_t st, const Options& opt);
size_t gemm_tn_scratch_bytes(int64_t M, int64_t N, int64_t K, const Options& opt);

}  // namespace csn
