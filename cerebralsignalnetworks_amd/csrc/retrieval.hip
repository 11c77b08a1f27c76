// K8: exact squared-L2 top-k (brute force).  Replaces faiss.IndexFlatL2(d).add/.search(k)
// at /root/reference/utils/Utilities.py:45-55.
//
// Stage 1: D2[q][g] = sum_d (query[q][d] - gallery[g][d])^2, accumulated in float64 in a fixed
//          order (so near-ties resolve exactly like the float64 oracle); a 16x16 (q x g) tile
//          per workgroup, operand rows staged through LDS in 64-wide slices of d.
// Stage 2: per query, k rounds of a workgroup-wide arg-min over (distance, index) with
//          already-selected entries masked; ties go to the lower gallery index.
#include "csn_common.h"

namespace csn {

__global__ void __launch_bounds__(256)
l2_dist_kernel(const float* __restrict__ gallery, const float* __restrict__ query, int64_t Ng, int64_t Nq, int D,
               double* __restrict__ dist) {
  __shared__ float qs[16][64 + 1];
  __shared__ float gs[16][64 + 1];
  const int tq = threadIdx.x >> 4, tg = threadIdx.x & 15;
  const int64_t q0 = (int64_t)blockIdx.y * 16, g0 = (int64_t)blockIdx.x * 16;
  double acc = 0.0;
  for (int d0 = 0; d0 < D; d0 += 64) {
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
      const int r = i >> 6, c = i & 63;
      qs[r][c] = (q0 + r < Nq && d0 + c < D) ? query[(q0 + r) * D + d0 + c] : 0.0f;
      gs[r][c] = (g0 + r < Ng && d0 + c < D) ? gallery[(g0 + r) * D + d0 + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll 8
    for (int c = 0; c < 64; ++c) {
      const double df = (double)qs[tq][c] - (double)gs[tg][c];
      acc = fma(df, df, acc);
    }
    __syncthreads();
  }
  if (q0 + tq < Nq && g0 + tg < Ng) dist[(q0 + tq) * Ng + g0 + tg] = acc;
}

__global__ void __launch_bounds__(256)
topk_select_kernel(double* __restrict__ dist, int64_t Ng, int k, int64_t* __restrict__ out_idx,
                   float* __restrict__ out_dist) {
  __shared__ double sd[256];
  __shared__ int64_t si[256];
  double* row = dist + (int64_t)blockIdx.x * Ng;
  const double INF = __builtin_inf();
  for (int j = 0; j < k; ++j) {
    double best = INF;
    int64_t bi = INT64_MAX;
    for (int64_t g = threadIdx.x; g < Ng; g += 256) {
      const double v = row[g];
      if (v < best || (v == best && g < bi)) { best = v; bi = g; }
    }
    sd[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (threadIdx.x < s) {
        const double v = sd[threadIdx.x + s];
        const int64_t i2 = si[threadIdx.x + s];
        if (v < sd[threadIdx.x] || (v == sd[threadIdx.x] && i2 < si[threadIdx.x])) {
          sd[threadIdx.x] = v;
          si[threadIdx.x] = i2;
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const int64_t w = si[0];
      out_idx[(int64_t)blockIdx.x * k + j] = (w == INT64_MAX) ? -1 : w;
      out_dist[(int64_t)blockIdx.x * k + j] = (float)sd[0];
      if (w != INT64_MAX) row[w] = INF;   // mask for the next round
    }
    __syncthreads();
  }
}

}  // namespace csn

using namespace csn;

extern "C" size_t csn_l2_topk_scratch_bytes(int64_t Ng, int64_t Nq) {
  if (Ng <= 0 || Nq <= 0) return 0;
  return (size_t)Ng * (size_t)Nq * sizeof(double);
}

extern "C" int csn_l2_topk(const float* gallery, const float* query, int64_t Ng, int64_t Nq, int D, int k,
                           int64_t* out_idx, float* out_dist, void* scratch, csnStream_t stream) {
  CSN_REQUIRE(gallery && query && out_idx && out_dist && scratch, "csn_l2_topk: null pointer");
  CSN_REQUIRE(Ng > 0 && Nq > 0 && D > 0, "csn_l2_topk: bad shape");
  CSN_REQUIRE(k > 0 && k <= 64 && k <= Ng, "csn_l2_topk: k=%d must be in 1..min(64, Ng)", k);
  hipStream_t st = as_stream(stream);
  dim3 grid((unsigned)((Ng + 15) / 16), (unsigned)((Nq + 15) / 16));
  l2_dist_kernel<<<grid, 256, 0, st>>>(gallery, query, Ng, Nq, D, (double*)scratch);
  CSN_LAUNCH_CHECK();
  topk_select_kernel<<<(unsigned)Nq, 256, 0, st>>>((double*)scratch, Ng, k, out_idx, out_dist);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}
