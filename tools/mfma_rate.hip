// Diagnostic (GPU box): issue rate of v_mfma_f32_16x16x32_bf16, one wave per SIMD, 8 independent accumulators, by where
// the operands live (the K-split kernels leave it to the compiler, the N-split family pins the stationary operand in
// AGPRs through inline asm), measured WARM: the kernel is launched back to back for 2 s before the timed launches
// (a cold, otherwise idle chip runs a short kernel at 1.2-1.4 GHz: the first version of this tool reported 14 ns per
// MFMA for that reason), with the in-kernel clock = delta s_memtime / delta s_memrealtime x 100 MHz.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o build/mfma_rate && ./build/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
// MODE 0: builtin (compiler places everything)   1: A agpr, C agpr   2: A agpr, C vgpr   3: A vgpr, C vgpr   4: A vgpr, C agpr
template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, int iters, unsigned long long* cyc) {
  bf16x8 w[8], h[4];
  for (int i = 0; i < 8; ++i) for (int e = 0; e < 8; ++e) w[i][e] = (__bf16)(0.001f * ((threadIdx.x * 7 + i * 3 + e) % 97));
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) h[i][e] = (__bf16)(0.002f * ((threadIdx.x * 5 + 2 * i + e) % 89));
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[i], h[i & 3], acc[i], 0, 0, 0);
        else if (MODE == 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "a"(w[i]), "v"(h[i & 3]));
        else if (MODE == 2) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "a"(w[i]), "v"(h[i & 3]));
        else if (MODE == 3) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w[i]), "v"(h[i & 3]));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(w[i]), "v"(h[i & 3]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = r1 - r0; }
}
template <int MODE>
static void run(int grid, float* out, unsigned long long* cyc, int iters) {
  unsigned long long h[2];
  const auto t_begin = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() < 2.0) {   // warm: 2 s back to back
    for (int i = 0; i < 50; ++i) k<MODE><<<grid, 256>>>(out, iters, cyc);
    hipDeviceSynchronize();
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) k<MODE><<<grid, 256>>>(out, iters, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("grid %3d mode %d: %.2f ns per MFMA (events, 20 launches), %.2f shader cycles per MFMA, in-kernel clock %.0f MHz\n",
         grid, MODE, ms * 1e6 / (20.0 * iters * 48.0), (double)h[0] / (iters * 48.0), 100.0 * (double)h[0] / (double)h[1]);
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 16);
  const int iters = 2000;
  for (int grid : {1, 256}) {
    run<0>(grid, out, cyc, iters); run<1>(grid, out, cyc, iters); run<2>(grid, out, cyc, iters);
    run<3>(grid, out, cyc, iters); run<4>(grid, out, cyc, iters);
  }
  return 0;
}
