// Diagnostic (GPU box only): column sums produced by the 256x256 TN kernel against a host sum.
#include <vector>
#include <cstdio>
#include <cmath>
#include "../cerebralsignalnetworks_amd/csrc/csn_common.h"
using namespace csn;
int main() {
  const int64_t M = 1024, N = 256, K = 8320;
  std::vector<float> a((size_t)K * M), b((size_t)K * N);
  for (size_t i = 0; i < a.size(); ++i) a[i] = (float)((int)((i * 2654435761u) >> 20 & 15) - 8) / 8.0f;
  for (size_t i = 0; i < b.size(); ++i) b[i] = (float)((int)((i * 40503u) >> 7 & 15) - 8) / 8.0f;
  std::vector<bf16_t> ab(a.size()), bb(b.size());
  for (size_t i = 0; i < a.size(); ++i) ab[i] = (bf16_t)a[i];
  for (size_t i = 0; i < b.size(); ++i) bb[i] = (bf16_t)b[i];
  void *dA, *dB; float *slabs, *cs;
  hipMalloc(&dA, ab.size() * 2); hipMalloc(&dB, bb.size() * 2);
  hipMalloc((void**)&slabs, csn_gemm_tn_scratch_bytes(M, N, K)); hipMalloc((void**)&cs, 128 * M * 4);
  hipMemset(cs, 0xff, 128 * M * 4);
  hipMemcpy(dA, ab.data(), ab.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, bb.data(), bb.size() * 2, hipMemcpyHostToDevice);
  int S = 0, done = 0;
  int rc = launch_gemm_tn_slabs(dA, dB, slabs, M, N, K, CSN_BF16, nullptr, &S, cs, &done);
  hipDeviceSynchronize();
  printf("rc %d S %d done %d\n", rc, S, done);
  std::vector<float> h((size_t)S * M);
  hipMemcpy(h.data(), cs, h.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int64_t m = 0; m < M; ++m) {
    double want = 0, got = 0;
    for (int64_t k = 0; k < K; ++k) want += a[k * M + m];
    for (int s = 0; s < S; ++s) got += h[(size_t)s * M + m];
    if (!(std::fabs(got - want) < 1e-2 * (1 + std::fabs(want)))) { if (bad < 12) printf("m %lld got %g want %g\n", (long long)m, got, want); ++bad; }
  }
  printf("bad %d of %lld\n", bad, (long long)M);
  return bad != 0;
}
