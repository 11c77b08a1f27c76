// Diagnostic (GPU box only): where does a cell launch spend its time?  Built with -DCSN_STAMPS.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCSN_STAMPS tools/stamp_cells.hip \
//         cerebralsignalnetworks_amd/csrc/util.hip -o build/stamp_cells && ./build/stamp_cells [nprob]
#include <algorithm>
#include <vector>
#include "../cerebralsignalnetworks_amd/csrc/lstm_cell_blk.hip"

using namespace csn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

static void* dmalloc(size_t n) { void* p = nullptr; if (hipMalloc(&p, n) != hipSuccess) { printf("malloc fail\n"); exit(1);} hipMemset(p, 0, n); return p; }

int main(int argc, char** argv) {
  const int NP = argc > 1 ? atoi(argv[1]) : 2;
  const int B = 256, H = 768, T = 32, G = 4 * H;
  unsigned long long* stamps = (unsigned long long*)dmalloc(4096 * 64);
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &stamps, sizeof(stamps)));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct Bufs { void *hblk[2], *wblk, *wtblk, *gates, *hout, *dg, *dgblk[2]; float *xproj, *c, *dc, *dy; } b[4];
  for (int i = 0; i < NP; ++i) {
    b[i].hblk[0] = dmalloc((size_t)B * H * 2); b[i].hblk[1] = dmalloc((size_t)B * H * 2);
    b[i].wblk = dmalloc((size_t)G * H * 2); b[i].wtblk = dmalloc((size_t)G * H * 2);
    b[i].gates = dmalloc((size_t)T * B * G * 2); b[i].hout = dmalloc((size_t)T * B * H * 2);
    b[i].dg = dmalloc((size_t)T * B * G * 2);
    b[i].dgblk[0] = dmalloc((size_t)B * G * 2); b[i].dgblk[1] = dmalloc((size_t)B * G * 2);
    b[i].xproj = (float*)dmalloc((size_t)T * B * G * 4); b[i].c = (float*)dmalloc((size_t)(T + 1) * B * H * 4);
    b[i].dc = (float*)dmalloc((size_t)B * H * 4); b[i].dy = (float*)dmalloc((size_t)T * B * H * 4);
  }
  auto report = [&](const char* name, int nwg, float us) {
    std::vector<unsigned long long> h(nwg * 8);
    hipMemcpy(h.data(), stamps, nwg * 64, hipMemcpyDeviceToHost);
    unsigned long long t_first = ~0ull, t_last = 0;
    for (int w = 0; w < nwg; ++w) { t_first = std::min(t_first, h[w * 8]); t_last = std::max(t_last, h[w * 8 + 3]); }
    double seg[4] = {0, 0, 0, 0}, spread = 0;
    for (int w = 0; w < nwg; ++w) {
      spread = std::max(spread, (double)(h[w * 8] - t_first));
      for (int i = 1; i <= 3; ++i) seg[i] += (double)(h[w * 8 + i] - h[w * 8 + i - 1]);
    }
    printf("%s x%d: %.2f us/launch back-to-back | in-kernel first start->last end %.2f us, start spread %.2f | mean WG: loads+mfma %.2f, lds+barrier %.2f, epilogue %.2f\n",
           name, NP, us, (t_last - t_first) * 0.01, spread * 0.01, seg[1] / nwg * 0.01, seg[2] / nwg * 0.01, seg[3] / nwg * 0.01);
  };
  auto run_fwd = [&](int n) {
    for (int it = 0; it < n; ++it) {
      int t = it % T;
      CellFwdArgs a{}; a.B = B; a.H = H;
      for (int i = 0; i < NP; ++i) {
        CellFwdProb& P = a.p[i];
        P.h_prev_blk = (bf16_t*)b[i].hblk[t & 1]; P.w_blk = (bf16_t*)b[i].wblk; P.xproj = b[i].xproj + (size_t)t * B * G;
        P.c_prev = b[i].c + (size_t)t * B * H; P.gates_out = (bf16_t*)b[i].gates + (size_t)t * B * G;
        P.c_out = b[i].c + (size_t)(t + 1) * B * H; P.h_out = (bf16_t*)b[i].hout + (size_t)t * B * H;
        P.h_out_blk = (bf16_t*)b[i].hblk[(t + 1) & 1];
      }
      launch_cell_fwd_il(a, NP, st);
    }
  };
  auto run_bwd = [&](int n) {
    for (int it = 0; it < n; ++it) {
      int t = it % T;
      CellBwdArgs a{}; a.B = B; a.H = H;
      for (int i = 0; i < NP; ++i) {
        CellBwdProb& P = a.p[i];
        P.dg_next_blk = (bf16_t*)b[i].dgblk[(t + 1) & 1]; P.wt_blk = (bf16_t*)b[i].wtblk; P.dy = b[i].dy + (size_t)t * B * H; P.dy_ld = H;
        P.gates = (bf16_t*)b[i].gates + (size_t)t * B * G; P.c = b[i].c + (size_t)(t + 1) * B * H; P.c_prev = b[i].c + (size_t)t * B * H;
        P.dc_carry = b[i].dc; P.dg_out = (bf16_t*)b[i].dg + (size_t)t * B * G; P.dg_out_blk = (bf16_t*)b[i].dgblk[t & 1];
      }
      launch_cell_bwd_il(a, NP, st);
    }
  };
  float ms;
  run_fwd(50); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st)); run_fwd(500); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  CK(hipEventElapsedTime(&ms, e0, e1));
  report("fwd_il", (H / 24) * (B / 64) * NP, ms * 1e3 / 500);
  run_bwd(50); CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st)); run_bwd(500); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
  CK(hipEventElapsedTime(&ms, e0, e1));
  report("bwd_il", (H / 48) * (B / 32) * NP, ms * 1e3 / 500);
  return 0;
}
