// Diagnostic (GPU box): issue rate of v_mfma_f32_16x16x32_bf16 for one wave per SIMD with 8 independent accumulators,
// the stationary operand in VGPRs (builtin) or in AGPRs (inline asm), optionally with an s_barrier every 48 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o build/mfma_rate && ./build/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* cyc) {
  bf16x8 w[8], h[4];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) w[i][e] = (__bf16)(0.001f * (threadIdx.x + i + e));
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) h[i][e] = (__bf16)(0.002f * (threadIdx.x + 2 * i + e));
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i], h[i & 3], acc[i], 0, 0, 0);
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "a"(w[i]), "v"(h[i & 3]));
      }
      if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE >= 1) __builtin_amdgcn_s_barrier();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc; unsigned long long h;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
  const int iters = 1000;
  for (int grid : {1, 256}) {
    for (int mode = 0; mode < 3; ++mode) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) k<0><<<grid, 256>>>(out, iters, cyc);
        else if (mode == 1) k<1><<<grid, 256>>>(out, iters, cyc);
        else k<2><<<grid, 256>>>(out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      printf("grid %3d mode %d: %.1f ns per MFMA (event), %.1f shader-clock cycles per MFMA\n", grid, mode, ms * 1e6 / (iters * 48.0), (double)h / (iters * 48.0));
    }
  }
  return 0;
}
