// Diagnostic (GPU box only): where does a forward cell launch spend its time?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCSN_STAMPS -I. tools/stamp_cells.hip \
//         cerebralsignalnetworks_amd/csrc/util.hip -o /tmp/stamp_cells && /tmp/stamp_cells
#include <algorithm>
#include <vector>
#include "../cerebralsignalnetworks_amd/csrc/lstm_cell_blk.hip"

using namespace csn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const int B = 256, H = 768, T = 32, G = 4 * H;
  void *hblk[2], *wblk, *gates, *hout; float *xproj, *c;
  CK(hipMalloc(&hblk[0], (size_t)B * H * 2)); CK(hipMalloc(&hblk[1], (size_t)B * H * 2));
  CK(hipMalloc(&wblk, (size_t)G * H * 2));
  CK(hipMalloc(&gates, (size_t)T * B * G * 2)); CK(hipMalloc(&hout, (size_t)T * B * H * 2));
  CK(hipMalloc(&xproj, (size_t)T * B * G * 4)); CK(hipMalloc(&c, (size_t)(T + 1) * B * H * 4));
  CK(hipMemset(hblk[0], 0, (size_t)B * H * 2)); CK(hipMemset(hblk[1], 0, (size_t)B * H * 2));
  CK(hipMemset(wblk, 0, (size_t)G * H * 2)); CK(hipMemset(xproj, 0, (size_t)T * B * G * 4));
  CK(hipMemset(c, 0, (size_t)(T + 1) * B * H * 4));
  const int nwg = (H / 16) * (B / 64);
  unsigned long long* stamps;
  CK(hipMalloc(&stamps, (size_t)nwg * 8 * 8));
  CK(hipMemset(stamps, 0, (size_t)nwg * 64));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int n) {
    for (int i = 0; i < n; ++i) {
      int t = i % T;
      launch_cell_fwd_blk(hblk[t & 1], wblk, xproj + (size_t)t * B * G, G, c + (size_t)t * B * H,
                          (char*)gates + (size_t)t * B * G * 2, c + (size_t)(t + 1) * B * H,
                          (char*)hout + (size_t)t * B * H * 2, hblk[(t + 1) & 1], B, H, st);
    }
  };
  run(50);
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  run(500);
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("fwd blk: %.2f us per launch (back-to-back, C++ loop)\n", ms * 1e3 / 500);
  std::vector<unsigned long long> h(nwg * 8);
  CK(hipMemcpy(h.data(), stamps, nwg * 64, hipMemcpyDeviceToHost));
  unsigned long long t_first = ~0ull, t_last = 0;
  for (int w = 0; w < nwg; ++w) { t_first = std::min(t_first, h[w * 8]); t_last = std::max(t_last, h[w * 8 + 4]); }
  double seg[5] = {0, 0, 0, 0, 0}, start_spread = 0;
  for (int w = 0; w < nwg; ++w) {
    start_spread = std::max(start_spread, (double)(h[w * 8] - t_first));
    for (int i = 1; i <= 4; ++i) seg[i] += (double)(h[w * 8 + i] - h[w * 8 + i - 1]);
  }
  printf("last launch: first WG start -> last WG end %.2f us; start spread %.2f us\n", (t_last - t_first) * 0.01, start_spread * 0.01);
  printf("mean per-WG segments (us): loads+mfma %.2f | lds write+barrier %.2f | reduce+math %.2f | stores issue %.2f\n",
         seg[1] / nwg * 0.01, seg[2] / nwg * 0.01, seg[3] / nwg * 0.01, seg[4] / nwg * 0.01);
  return 0;
}
