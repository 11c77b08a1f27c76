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
  // ./persist_bench [nlayers] [nsteps] [mode: 0 = one launch per layer on its own stream (placement-independent),
  //                                        1 = grouped launch, placement-independent hand-off, 2 = grouped, L2-local]
  const int NL = argc > 1 ? atoi(argv[1]) : 1, NS = argc > 2 ? atoi(argv[2]) : 64, mode = argc > 3 ? atoi(argv[3]) : 2;
  const int B = 256, H = 768, T = NS, G = 4 * H, Bpad = 256, MT = 4;
  PersistFwdArgs a{};
  hipStream_t st[2];
  unsigned* err = (unsigned*)dmalloc(256);
  unsigned long long* agree = (unsigned long long*)dmalloc(64);
  for (int l = 0; l < NL; ++l) {
    CK(hipStreamCreateWithFlags(&st[l], hipStreamNonBlocking));
    PersistFwdSlot& S = a.slot[l];
    S.w_blk = (bf16_t*)dmalloc((size_t)G * H * 2);
    S.xproj = (float*)dmalloc((size_t)T * B * G * 4);
    S.gates = (bf16_t*)dmalloc((size_t)T * B * G * 2);
    S.c_all = (float*)dmalloc((size_t)(T + 1) * B * H * 4);
    S.h_all = (bf16_t*)dmalloc((size_t)(T + 1) * B * H * 2);
    S.h_blk_all = (bf16_t*)dmalloc((size_t)(T + 1) * Bpad * H * 2);
    S.flags = (unsigned*)dmalloc((size_t)(T + 1) * MT * kPersistFlagLine * 4);
    fill_bf16<<<1024, 256>>>((bf16_t*)S.w_blk, (size_t)G * H, 0.036f, 1 + l);
    fill_f32<<<2048, 256>>>((float*)S.xproj, (size_t)T * B * G, 1.0f, 7 + l);
    S.x_blk = nullptr; S.wih_blk = nullptr; S.bias = nullptr; S.I = 0;
    S.t0 = 0; S.nsteps = NS;
  }
  a.error_flag = err;
  a.B = B; a.H = H; a.T = T; a.Bpad = Bpad; a.MT = MT; a.rotate = getenv("CSN_NO_ROTATE") == nullptr;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    for (int l = 0; l < NL; ++l) CK(hipMemsetAsync(a.slot[l].flags, 0, (size_t)(T + 1) * MT * kPersistFlagLine * 4, st[0]));
    CK(hipMemsetAsync(agree, 0, 64, st[0]));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, st[0]));
    if (mode == 0) {
      for (int l = 0; l < NL; ++l) {
        PersistFwdArgs one = a;
        one.slot[0] = a.slot[l]; one.nslots = 1; one.xcd_groups = 0; one.agree = nullptr;
        if (launch_fwd_persist(one, st[l])) { printf("launch: %s\n", csn_last_error()); return 1; }
      }
      for (int l = 1; l < NL; ++l) { hipEvent_t ev; CK(hipEventCreate(&ev)); CK(hipEventRecord(ev, st[l])); CK(hipStreamWaitEvent(st[0], ev, 0)); }
    } else {
      a.nslots = NL; a.xcd_groups = 1; a.agree = mode == 2 ? agree : nullptr;
      if (launch_fwd_persist(a, st[0])) { printf("launch: %s\n", csn_last_error()); return 1; }
    }
    CK(hipEventRecord(e1, st[0]));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned flag = 0; CK(hipMemcpy(&flag, err, 4, hipMemcpyDeviceToHost));
    unsigned long long ag[8]; CK(hipMemcpy(ag, agree, 64, hipMemcpyDeviceToHost));
    printf("persist fwd mode %d: %d layer(s) x %d steps: %.1f us total, %.2f us/step, error_flag=%u, agree[0]=%llx agree[%d]=%llx\n",
           mode, NL, NS, ms * 1e3, ms * 1e3 / NS, flag, ag[0], NL * MT - 1, ag[NL * MT - 1]);
#ifdef CSN_PSTAMPS
    unsigned long long ps[8];
    CK(hipMemcpyFromSymbol(ps, HIP_SYMBOL(g_pstamps), sizeof(ps)));
    printf("   per step (us): wait %.2f | h loads+mfma %.2f | lds write+barrier %.2f | epilogue %.2f | drain+barrier %.2f | signal %.2f\n",
           ps[0] * 0.01 / NS, ps[1] * 0.01 / NS, ps[2] * 0.01 / NS, ps[3] * 0.01 / NS, ps[4] * 0.01 / NS, ps[5] * 0.01 / NS);
    unsigned long long z[8] = {0,0,0,0,0,0,0,0};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_pstamps), z, sizeof(z)));
#endif
  }
  // checksum of the last h slab per layer: the three modes must agree bit for bit
  for (int l = 0; l < NL; ++l) {
    std::vector<unsigned short> h((size_t)Bpad * H);
    CK(hipMemcpy(h.data(), a.slot[l].h_blk_all + (size_t)T * Bpad * H, h.size() * 2, hipMemcpyDeviceToHost));
    unsigned long long cs = 1469598103934665603ull;
    for (unsigned short v : h) { cs ^= v; cs *= 1099511628211ull; }
    printf("layer %d h_T checksum %016llx\n", l, cs);
  }
  return 0;
}
