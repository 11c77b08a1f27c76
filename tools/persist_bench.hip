// Diagnostic (GPU box only): per-step time of the weight-stationary forward kernel, 1 or 2 layers
// side by side.   ./build/persist_bench [nlayers] [nsteps]
#include <vector>
#include "../cerebralsignalnetworks_amd/csrc/lstm_fwd_persist.hip"
using namespace csn;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill_bf16(bf16_t* p, size_t n, float scale, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = (bf16_t)(((int)(x & 0xffff) - 32768) / 32768.0f * scale);
  }
}
__global__ void fill_f32(float* p, size_t n, float scale, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u + seed; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
    p[i] = ((int)(x & 0xffff) - 32768) / 32768.0f * scale;
  }
}
static void* dmalloc(size_t n) { void* p = nullptr; if (hipMalloc(&p, n) != hipSuccess) { printf("malloc fail\n"); exit(1);} hipMemset(p, 0, n); return p; }
int main(int argc, char** argv) {
  const int NL = argc > 1 ? atoi(argv[1]) : 1, NS = argc > 2 ? atoi(argv[2]) : 64;
  const int B = 256, H = 768, T = NS, G = 4 * H, Bpad = 256, MT = 4;
  PersistFwdArgs a[2];
  hipStream_t st[2];
  unsigned* err = (unsigned*)dmalloc(256);
  for (int l = 0; l < NL; ++l) {
    CK(hipStreamCreateWithFlags(&st[l], hipStreamNonBlocking));
    a[l].w_blk = (bf16_t*)dmalloc((size_t)G * H * 2);
    a[l].xproj = (float*)dmalloc((size_t)T * B * G * 4);
    a[l].gates = (bf16_t*)dmalloc((size_t)T * B * G * 2);
    a[l].c_all = (float*)dmalloc((size_t)(T + 1) * B * H * 4);
    a[l].h_all = (bf16_t*)dmalloc((size_t)(T + 1) * B * H * 2);
    a[l].h_blk_all = (bf16_t*)dmalloc((size_t)(T + 1) * Bpad * H * 2);
    a[l].counters = (unsigned*)dmalloc((size_t)(T + 1) * MT * 4);
    a[l].error_flag = err;
    if (argc > 3) {
      fill_bf16<<<1024, 256>>>((bf16_t*)a[l].w_blk, (size_t)G * H, 0.036f, 1 + l);
      fill_f32<<<2048, 256>>>((float*)a[l].xproj, (size_t)T * B * G, 1.0f, 7 + l);
    }
    a[l].B = B; a[l].H = H; a[l].T = T; a[l].t0 = 0; a[l].nsteps = NS; a[l].Bpad = Bpad;
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    for (int l = 0; l < NL; ++l) CK(hipMemsetAsync(a[l].counters, 0, (size_t)(T + 1) * MT * 4, st[l]));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, st[0]));
    for (int l = 0; l < NL; ++l) if (launch_fwd_persist(a[l], st[l])) { printf("launch: %s\n", csn_last_error()); return 1; }
    for (int l = 1; l < NL; ++l) { hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, st[l])); CK(hipStreamWaitEvent(st[0], ev, 0)); }
    CK(hipEventRecord(e1, st[0]));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned flag = 0; CK(hipMemcpy(&flag, err, 4, hipMemcpyDeviceToHost));
    printf("persist fwd: %d layer(s) x %d steps: %.1f us total, %.2f us/step, error_flag=%u\n", NL, NS, ms * 1e3, ms * 1e3 / NS, flag);
#ifdef CSN_PSTAMPS
    unsigned long long ps[8];
    CK(hipMemcpyFromSymbol(ps, HIP_SYMBOL(g_pstamps), sizeof(ps)));
    printf("   per step (us): wait %.2f | h loads+mfma %.2f | lds write+barrier %.2f | epilogue %.2f | drain+barrier %.2f | signal %.2f\n",
           ps[0] * 0.01 / NS / NL, ps[1] * 0.01 / NS / NL, ps[2] * 0.01 / NS / NL, ps[3] * 0.01 / NS / NL, ps[4] * 0.01 / NS / NL, ps[5] * 0.01 / NS / NL);
    unsigned long long z[8] = {0,0,0,0,0,0,0,0};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pstamps), z, sizeof(z)));
#endif
  }
  return 0;
}
