// Helpers shared by the LSTM cell kernels: 4-wide vector load/store in either storage type.
#pragma once
#include "csn_common.h"

// Streaming (touched-once) epilogue traffic -- xproj, saved gates, c, row-major h / dgates --
// is marked non-temporal so it does not displace the re-read operands (W, h) from L2 and leaves
// fewer dirty lines for the end-of-kernel write-back.
#ifndef CSN_NT
#define CSN_NT 1
#endif
template <typename V> __device__ __forceinline__ V nt_load(const V* p) {
#if CSN_NT
  return __builtin_nontemporal_load(p);
#else
  return *p;
#endif
}
template <typename V> __device__ __forceinline__ void nt_store(V* p, const V& v) {
#if CSN_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
// float4 (a HIP struct type) goes through the equivalent clang vector type
__device__ __forceinline__ float4 nt_load(const float4* p) {
  const csn::f32x4 v = nt_load(reinterpret_cast<const csn::f32x4*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void nt_store(float4* p, const float4& v) {
  nt_store(reinterpret_cast<csn::f32x4*>(p), (csn::f32x4){v.x, v.y, v.z, v.w});
}

namespace csn {

template <typename T> struct Vec4;
template <> struct Vec4<float> {
  static __device__ __forceinline__ void store(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
  static __device__ __forceinline__ void load(const float* p, float (&v)[4]) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  }
};
template <> struct Vec4<bf16_t> {
  static __device__ __forceinline__ void store(bf16_t* p, const float (&v)[4]) {
    bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
    *reinterpret_cast<bf16x4*>(p) = o;
  }
  static __device__ __forceinline__ void load(const bf16_t* p, float (&v)[4]) {
    const bf16x4 q = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)q[0]; v[1] = (float)q[1]; v[2] = (float)q[2]; v[3] = (float)q[3];
  }
};


// Fast activations for the bf16 path: v_exp_f32 + v_rcp_f32 (about 1 ulp each), far inside the
// 8 significant bits the bf16 operands keep.  The f32 parity path uses expf / tanhf instead.
__device__ __forceinline__ float fast_sigmoid(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
  // 1 - 2 / (1 + e^{2x}); saturates correctly for large |x| (exp2 -> inf or 0)
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// "Fragment-major" (blocked) bf16 layout of a [R, K] matrix, K % 32 == 0, rows padded to 16:
// 16-row x 32-k blocks of 1 KB, block (rb, kb) at ((rb * K/32) + kb) * 512 elements; inside a
// block the 16-byte chunk of (row r, k-octet q) sits at lane position r + 16 q -- exactly the
// operand fragment of v_mfma_f32_16x16x32_bf16 -- so one wave-wide 16-B-per-lane load of a
// fragment is ONE contiguous 1 KB read (8 full cache lines instead of 16 half-used ones).
__host__ __device__ __forceinline__ int64_t blk_offset(int64_t r, int64_t k, int64_t K) {
  return (((r >> 4) * (K >> 5)) + (k >> 5)) * 512 + (((r & 15) + 16 * ((k & 31) >> 3)) << 3) + (k & 7);
}

}  // namespace csn
