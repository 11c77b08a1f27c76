// K5 / K7: small fused reductions of the distillation losses.
//   cosine loss  : /root/reference/LstmDistillFromDinoV2Train.py:36-43
//                  loss = 1 - mean_b cos(s_b, t_b), nn.CosineSimilarity(dim=1, eps=1e-8)
//   Barlow terms : /root/reference/EEG-BarlowNetworks/net.py:6-9,39-40
//                  on = sum_i (c_ii - 1)^2, off = sum_{i != j} c_ij^2
// Both are tiny (B x 384, 384 x 384): one wave per row with shuffle reductions, float64
// accumulation, and a fixed-order final sum so results are bitwise reproducible.
#include "csn_common.h"

namespace csn {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// one wave per row b: cos_b and (optionally) the gradient row
__global__ void __launch_bounds__(256)
cosine_rows_kernel(const float* __restrict__ s, const float* __restrict__ t, int B, int D, double* __restrict__ cos_out,
                   float* __restrict__ ds, float grad_scale) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= B) return;
  const float* sr = s + (int64_t)row * D;
  const float* tr = t + (int64_t)row * D;
  double dot = 0.0, ss = 0.0, tt = 0.0;
  for (int i = lane; i < D; i += 64) {
    const double a = sr[i], b = tr[i];
    dot = fma(a, b, dot);
    ss = fma(a, a, ss);
    tt = fma(b, b, tt);
  }
  dot = wave_sum(dot);
  ss = wave_sum(ss);
  tt = wave_sum(tt);
  const double eps = 1e-8;
  const double sn = fmax(sqrt(ss), eps), tn = fmax(sqrt(tt), eps);
  const double c = dot / (sn * tn);
  if (lane == 0) cos_out[row] = c;
  if (ds != nullptr) {
    // d(1 - mean cos)/ds_b = -(1/B) * ( t/(|s||t|) - cos * s/|s|^2 )
    const double k = -(double)grad_scale / (double)B;
    const double inv_st = 1.0 / (sn * tn), inv_ss = c / (sn * sn);
    for (int i = lane; i < D; i += 64)
      ds[(int64_t)row * D + i] = (float)(k * ((double)tr[i] * inv_st - (double)sr[i] * inv_ss));
  }
}

__global__ void __launch_bounds__(64) cosine_finish_kernel(const double* __restrict__ cos_rows, int B,
                                                          float* __restrict__ loss) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < B; i += 64) acc += cos_rows[i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) loss[0] = (float)(1.0 - acc / (double)B);
}

__global__ void __launch_bounds__(1024) barlow_kernel(const float* __restrict__ c, int D, float* __restrict__ out) {
  __shared__ double sh_on[16], sh_off[16];
  double on = 0.0, off = 0.0;
  const int64_t total = (int64_t)D * D;
  for (int64_t i = threadIdx.x; i < total; i += 1024) {
    const int r = (int)(i / D), col = (int)(i % D);
    const double v = c[i];
    if (r == col) on = fma(v - 1.0, v - 1.0, on);
    else off = fma(v, v, off);
  }
  on = wave_sum(on);
  off = wave_sum(off);
  if ((threadIdx.x & 63) == 0) { sh_on[threadIdx.x >> 6] = on; sh_off[threadIdx.x >> 6] = off; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double a = 0.0, b = 0.0;
    for (int w = 0; w < 16; ++w) { a += sh_on[w]; b += sh_off[w]; }
    out[0] = (float)a;
    out[1] = (float)b;
  }
}

}  // namespace csn

using namespace csn;

// The per-row cosines go through caller-owned scratch: the library keeps no buffer of its own (ABI 3).
extern "C" size_t csn_cosine_loss_scratch_bytes(int B) { return B > 0 ? (size_t)B * sizeof(double) : 0; }

extern "C" int csn_cosine_loss(const float* student, const float* teacher, int B, int D, float* loss, float* dstudent,
                               float grad_scale, void* scratch, csnStream_t stream) {
  CSN_REQUIRE(student && teacher && loss && scratch, "csn_cosine_loss: null pointer");
  CSN_REQUIRE(B > 0 && D > 0, "csn_cosine_loss: bad shape B=%d D=%d", B, D);
  CSN_REQUIRE((reinterpret_cast<uintptr_t>(scratch) & 7) == 0, "csn_cosine_loss: scratch must be 8-byte aligned");
  hipStream_t st = as_stream(stream);
  double* cos_rows = (double*)scratch;
  cosine_rows_kernel<<<(unsigned)((B + 3) / 4), 256, 0, st>>>(student, teacher, B, D, cos_rows, dstudent, grad_scale);
  CSN_LAUNCH_CHECK();
  cosine_finish_kernel<<<1, 64, 0, st>>>(cos_rows, B, loss);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

extern "C" int csn_barlow_offdiag_sqsum(const float* c, int D, float* out, csnStream_t stream) {
  CSN_REQUIRE(c && out, "csn_barlow_offdiag_sqsum: null pointer");
  CSN_REQUIRE(D > 0, "csn_barlow_offdiag_sqsum: bad D=%d", D);
  barlow_kernel<<<1, 1024, 0, as_stream(stream)>>>(c, D, out);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

// ------------------------------------------------------------------------------------------------
// Optimiser step of the hot loop on ONE flat parameter / gradient buffer.
// Replaces: torch.optim.RMSprop(model.parameters(), lr).step() at /root/reference/LstmDistillFromDinoV2Train.py:329,373
// with that call's defaults (alpha 0.99, eps 1e-8, no momentum, not centred, no weight decay):
//     v <- alpha v + (1 - alpha) g^2 ;   p <- p - lr g / (sqrt(v) + eps)
// One pass over 3 x n floats in, 2 x n out (torch's multi-tensor path: five kernels per step).
namespace csn {
__global__ void __launch_bounds__(256) rmsprop_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ v,
                                                          int64_t n, float lr, float alpha, float eps) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    vv.x = alpha * vv.x + (1.0f - alpha) * gv.x * gv.x;
    vv.y = alpha * vv.y + (1.0f - alpha) * gv.y * gv.y;
    vv.z = alpha * vv.z + (1.0f - alpha) * gv.z * gv.z;
    vv.w = alpha * vv.w + (1.0f - alpha) * gv.w * gv.w;
    pv.x -= lr * gv.x / (sqrtf(vv.x) + eps);
    pv.y -= lr * gv.y / (sqrtf(vv.y) + eps);
    pv.z -= lr * gv.z / (sqrtf(vv.z) + eps);
    pv.w -= lr * gv.w / (sqrtf(vv.w) + eps);
    reinterpret_cast<float4*>(v)[i] = vv;
    reinterpret_cast<float4*>(p)[i] = pv;
  }
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float gi = g[i];
    const float vi = alpha * v[i] + (1.0f - alpha) * gi * gi;
    v[i] = vi;
    p[i] -= lr * gi / (sqrtf(vi) + eps);
  }
}
}  // namespace csn

extern "C" int csn_rmsprop_step(float* params, const float* grads, float* square_avg, int64_t n, float lr, float alpha,
                                float eps, csnStream_t stream) {
  CSN_REQUIRE(params && grads && square_avg && n > 0, "csn_rmsprop_step: null pointer or empty buffer");
  CSN_REQUIRE(((reinterpret_cast<uintptr_t>(params) | reinterpret_cast<uintptr_t>(grads) | reinterpret_cast<uintptr_t>(square_avg)) & 15) == 0,
              "csn_rmsprop_step: buffers must be 16-byte aligned");
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  csn::rmsprop_flat_kernel<<<(unsigned)blocks, 256, 0, csn::as_stream(stream)>>>(params, grads, square_avg, n, lr, alpha, eps);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}
