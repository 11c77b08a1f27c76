// K3 orchestration: stacked LSTM forward / backward over the whole sequence.
// Replaces nn.LSTM(batch_first=True).forward / autograd backward as called at
// /root/reference/LSTMDistill.py:118,132 and /root/reference/LSTMDistillRetreival.py:91,103.
//
// Two paths share the C entry points:
//
//  * "il" fast path (bf16, H % 128 == 0): lstm_cell_blk.hip kernels.  The layers advance as a
//    WAVEFRONT: one launch per diagonal runs layer 0 at step d, layer 1 at step d - lag, ...
//    (lag = 2 chunks), so a launch boundary (~2 us) and the launch ramp are paid once per
//    diagonal, not once per layer-step, and a 2-layer diagonal is exactly one workgroup per CU.
//    The non-recurrent contractions are big GEMMs on a second, library-owned HIP stream:
//      forward : xproj_{l+1}[chunk] = h_l[chunk] W_ih^T + b   as soon as layer l finished a chunk;
//      backward: dx_l[chunk] = dgates_l[chunk] W_ih (input gradient of layer l = dy of layer l-1),
//                then dW_hh, dW_ih, db of layer l once its recurrence is done
//    and HIP events order the two streams, so the MFMA-bound GEMMs run beside the
//    latency/bandwidth-bound recurrence of the other layer instead of after it.
//  * "v1" path (exact-f32 parity path, or shapes the fast path does not cover): layer after
//    layer, generic cell kernels of lstm_cell.hip.
//
// Layout in HBM (inside the caller's workspace; [T,B,*] time-major so one timestep is one slab):
//   per layer: compute-dtype weight copies (+ transposes / fragment-major forms), bias,
//              xproj[T,B,4H] f32, gates[T,B,4H], c_all[T+1,B,H] f32, h_all[T+1,B,H],
//              dgates[T,B,4H], dx[T,B,I_l] f32; fast path adds the fragment-major ping-pong
//              buffers of h and dgates.  In the fast path every 4H axis is gate-interleaved
//              (n' = 4 unit + gate); parameters and their gradients are (un)permuted at the API.
#include <algorithm>
#include <vector>

#include "csn_common.h"
#include "lstm_cell_blk.h"
#include "lstm_f32_persist.h"

namespace csn {

int launch_cell_fwd(const void* h_prev, const void* w_hh, const float* xproj, int64_t xproj_ld, const float* c_prev,
                    void* gates_out, float* c_out, void* h_out, int B, int H, int dtype, hipStream_t st);
int launch_cell_bwd(const void* dg_next, const void* w_hh_t, const float* dy, int64_t dy_ld, const void* gates,
                    const float* c, const float* c_prev, float* dc_carry, void* dg_out, int B, int H, int dtype,
                    hipStream_t st);

struct LayerWs {
  size_t wih, whh, whht, wiht, whh_blk, whht_blk, bias, xproj, gates, c_all, h_all, dgates, dx, dc_carry, hblk[2],
      dgblk[2], h_blk_all, counters, dg_blk_all, bflags;
};
struct WsLayout {
  LayerWs layer[8];
  size_t x_c, x_blk, wih0_blk, dy_tm, tn_scratch, colsum, tn_scratch2, colsum2, status, agree, agree_b, tile_ctr, zeros_bh, f32_bias_part, total;
  // weight-stationary paths: everything that must be zero at the start of a forward / a backward sits in ONE block each
  size_t zero_fwd, zero_fwd_bytes, zero_bwd, zero_bwd_bytes;
  bool fuse_x;
  bool il, persist, persist_bwd;
  bool fwd_ns;             // forward runs the N-split kernel (lstm_fwd_ns.hip), else the K-split one
  bool f32_persist;        // exact-float32 path: weight-stationary recurrence (lstm_f32_persist.hip), one launch per layer and row block
};

static bool whole_chip() {       // (asked when a layout is made -- plan creation --, not per launch)
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  return hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 256;
}

static WsLayout make_layout(const csnLstmDesc& d, int training, const Options& opt) {
  WsLayout w{};
  w.il = cell_blk_supported(d.H, d.dtype, opt);
  // (the K-split weight-stationary kernels address their per-step hand-off slabs with 32-bit byte offsets from step 0: a
  // sequence whose slabs reach 4 GiB takes the per-diagonal launches; the N-split kernel bases its resources per launch)
  const bool slabs_fit = ((size_t)d.T + 1) * (((size_t)d.B + 63) / 64 * 64) * (size_t)d.H * 8 < ((size_t)1 << 32);
  w.fwd_ns = w.il && fwd_ns_supported(d.B, d.H, d.dtype, opt) && d.L <= 4;
  w.persist = w.fwd_ns || (w.il && slabs_fit && fwd_persist_supported(d.B, d.H, d.dtype, opt) && d.L <= 4);
  w.persist_bwd = w.persist && slabs_fit && training && bwd_persist_supported(d.B, d.H, d.dtype, opt);
  size_t off = 0;
  const size_t es = dtype_size(d.dtype);
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += align_up(bytes, 256);
    return o;
  };
  const size_t TB = (size_t)d.T * d.B, H = d.H, G = 4 * (size_t)d.H;
  const size_t Bpad = ((size_t)d.B + 63) / 64 * 64;
  size_t tn_bytes = 0;
  for (int l = 0; l < d.L; ++l) {
    const size_t I = l == 0 ? d.I : d.H;
    LayerWs& L = w.layer[l];
    L.wih = take(G * I * es);
    L.wiht = take(G * I * es);
    L.bias = take(G * 4);
    if (w.il) {
      L.whh_blk = take(G * H * 2);
      L.whht_blk = take(G * H * 2);
      L.hblk[0] = take(Bpad * H * 2);
      L.hblk[1] = take(Bpad * H * 2);
      if (w.persist) L.h_blk_all = take(((size_t)(d.T < 3 ? 3 : d.T) + 1) * Bpad * H * 2);   // (>= 4: the data-poll hand-off uses the first 4 as a ring)
    } else {
      L.whh = take(G * H * es);
      L.whht = take(G * H * es);
    }
    L.xproj = take(TB * G * 4);
    L.gates = take(TB * G * es);
    L.c_all = take((TB + d.B) * H * 4);
    L.h_all = take((TB + d.B) * H * es);
    if (training) {
      L.dgates = take(TB * G * es);
      L.dx = take(TB * I * 4);
      if (!w.persist_bwd) L.dc_carry = take((size_t)d.B * H * 4);
      if (w.il) {
        L.dgblk[0] = take(Bpad * G * 2);
        L.dgblk[1] = take(Bpad * G * 2);
      }
      if (w.persist_bwd) L.dg_blk_all = take((size_t)(d.T < 4 ? 4 : d.T) * Bpad * G * 2);   // (>= 4: ring of the data-poll hand-off)
    }
    size_t a = gemm_tn_scratch_bytes(G, I, TB, opt), b = gemm_tn_scratch_bytes(G, H, TB, opt);
    if (a > tn_bytes) tn_bytes = a;
    if (b > tn_bytes) tn_bytes = b;
  }
  w.x_c = take(TB * d.I * es);
  w.status = take(256);
  // (H = 512 excluded: its 64 x 32-unit tile leaves no registers for the W_ih fragments.  N-split kernel at H = 1024: W_hh
  // fills the 256 AGPRs, W_ih's 32 registers are VGPR operands of their MFMAs -- round 3's fused instantiation asked for
  // AGPRs there too and the compiler's copies ran into an MFMA read hazard the recogniser cannot see inside inline asm
  // (wrong row group 0; tools/check_asm_hazards.py).  CSN_NO_FUSE_X keeps the projection GEMM.)
  w.fuse_x = w.persist && !opt.no_fuse_x &&
             (w.fwd_ns ? d.I == 128 : (d.I % 32 == 0 && d.I <= 128 && d.H != 512));
  if (w.fuse_x) {
    w.x_blk = take((size_t)d.T * Bpad * d.I * 2);
    w.wih0_blk = take(G * d.I * 2);
  }
  if (w.persist_bwd) w.zeros_bh = take((size_t)d.B * H * 4);
  // exact-float32 path, weight-stationary: needs a whole MI355X like the bf16 kernels (all workgroups of a launch co-resident)
  w.f32_persist = !w.il && d.dtype == CSN_F32 && !opt.no_persist && !opt.cell_v1 && f32_persist_supported(d.B, d.H) && whole_chip();
  if (w.f32_persist) {
    const size_t MTt = Bpad / 64;
    w.zero_fwd = off;
    for (int l = 0; l < d.L; ++l) w.layer[l].counters = take(((size_t)d.T + 1) * MTt * kF32FlagLine * 4);
    w.zero_fwd_bytes = off - w.zero_fwd;
    for (int l = 0; l < d.L; ++l) w.layer[l].h_blk_all = take(((size_t)d.T + 1) * Bpad * H * 4);       // fragment-major hand-off copies
    if (training) {
      for (int l = 0; l < d.L; ++l) w.layer[l].dg_blk_all = take((size_t)d.T * Bpad * G * 4);
      w.zeros_bh = take((size_t)d.B * H * 4);
      w.f32_bias_part = take(MTt * 4 * G * 4);      // per-row-group bias-gradient partial sums of ONE layer (consumed before the next launch)
      w.zero_bwd = off;
      for (int l = 0; l < d.L; ++l) w.layer[l].bflags = take((size_t)d.T * MTt * kF32FlagLine * 4);
      w.zero_bwd_bytes = off - w.zero_bwd;
    }
  }
  if (w.persist) {
    w.zero_fwd = off;
    // (x 4: the wave-specialised forward body keeps one flag line per 16-row chain, the half-pipelined one per 32-row half)
    for (int l = 0; l < d.L; ++l) w.layer[l].counters = take(((size_t)d.T + 1) * (Bpad / 64) * 4 * kPersistFlagLine * 4);
    w.agree = take(((size_t)d.T + 8) * 8 * sizeof(unsigned long long));   // 8 words per launch
    w.tile_ctr = take(((size_t)d.T + 8) * 4 * sizeof(unsigned));          // GEMM tile counters, 4 per launch
    w.zero_fwd_bytes = off - w.zero_fwd;
  }
  if (w.persist_bwd) {
    w.zero_bwd = off;
    for (int l = 0; l < d.L; ++l) {
      w.layer[l].dc_carry = take((size_t)d.B * H * 4);
      w.layer[l].bflags = take((size_t)d.T * (Bpad / 64) * kPersistFlagLine * 4);
    }
    w.agree_b = take(((size_t)d.T + 8) * 8 * sizeof(unsigned long long));
    w.zero_bwd_bytes = off - w.zero_bwd;
  }
  if (training) {
    w.dy_tm = take(TB * H * 4);
    w.tn_scratch = take(tn_bytes);
    w.colsum = take(colsum_scratch_bytes(G));
    if (w.persist_bwd && d.L > 1) {     // weight gradients of the upper layers run beside the last backward launches
      w.tn_scratch2 = take(tn_bytes);
      w.colsum2 = take(colsum_scratch_bytes(G));
    }
  }
  w.total = off;
  return w;
}

static int check_desc(const char* fn, const csnLstmDesc* d) {
  CSN_REQUIRE(d != nullptr, "%s: null descriptor", fn);
  CSN_REQUIRE(d->B > 0 && d->T > 0 && d->I > 0 && d->H > 0, "%s: bad shape B=%d T=%d I=%d H=%d", fn, d->B, d->T,
              d->I, d->H);
  CSN_REQUIRE(d->L >= 1 && d->L <= 8, "%s: L=%d outside 1..8", fn, d->L);
  CSN_REQUIRE(d->H % 32 == 0, "%s: H=%d must be a multiple of 32", fn, d->H);
  CSN_REQUIRE(d->dtype == CSN_F32 || d->dtype == CSN_BF16, "%s: bad dtype %d", fn, d->dtype);
  return CSN_OK;
}

// y_all[b][t][h] (f32, batch-first) <- h_all[t+1][b][h] (dtype, time-major)
template <typename T>
__global__ void gather_y_all_kernel(const T* __restrict__ h_all, float* __restrict__ y, int B, int Tn, int H) {
  const int64_t total = (int64_t)B * Tn * H;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t h = i % H, r = i / H, t = r % Tn, b = r / Tn;
    y[i] = to_f32(h_all[((t + 1) * B + b) * (int64_t)H + h]);
  }
}

// dst[t][b][h] = src[b][t][h]  (f32)
__global__ void bt_to_tb_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int Tn, int H) {
  const int64_t total = (int64_t)B * Tn * H;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t h = i % H, r = i / H, b = r % B, t = r / B;
    dst[i] = src[(b * Tn + t) * (int64_t)H + h];
  }
}

// dst[b][t][i] = src[t][b][i]  (f32)
__global__ void tb_to_bt_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int Tn, int H) {
  const int64_t total = (int64_t)B * Tn * H;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int64_t h = i % H, r = i / H, t = r % Tn, b = r / Tn;
    dst[i] = src[(t * B + b) * (int64_t)H + h];
  }
}

__global__ void add_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

// out[n] = part[0][n] + part[1][n] + ... (fixed order): the row groups' bias-gradient partial sums of the float32 backward
__global__ void sum_rows_kernel(const float* __restrict__ part, int P, int64_t n, float* __restrict__ out, float* __restrict__ out2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int p = 0; p < P; ++p) s += part[(int64_t)p * n + i];
  out[i] = s;
  out2[i] = s;
}

static inline unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

// ---- second stream + event pool: owned by the PLAN (created on first use, destroyed with it) -------
struct SideCtx {
  hipStream_t side = nullptr;           // GEMMs
  hipStream_t layer[8] = {nullptr};     // weight-stationary forward: one stream per layer >= 1
  hipStream_t wgrad = nullptr;          // LOWEST priority: weight-gradient GEMMs of finished layers beside the last backward launches
  std::vector<hipEvent_t> events;
  size_t next = 0;
};

static int side_ctx(SideCtx& c) {
  if (c.side == nullptr) {
    CSN_HIP_CHECK(hipStreamCreateWithFlags(&c.side, hipStreamNonBlocking));
    for (int l = 1; l < 8; ++l) CSN_HIP_CHECK(hipStreamCreateWithFlags(&c.layer[l], hipStreamNonBlocking));
    int lo = 0, hi = 0;                 // (numerically: lo = least urgent)
    CSN_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CSN_HIP_CHECK(hipStreamCreateWithPriority(&c.wgrad, hipStreamNonBlocking, lo));
  }
  c.next = 0;
  return CSN_OK;
}
static int next_event(SideCtx* c, hipEvent_t* ev) {
  if (c->next == c->events.size()) {
    hipEvent_t e;
    CSN_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    c->events.push_back(e);
  }
  *ev = c->events[c->next++];
  return CSN_OK;
}
// record on `from`, make `to` wait
static int hand_off(SideCtx* c, hipStream_t from, hipStream_t to) {
  hipEvent_t ev;
  if (int rc = next_event(c, &ev)) return rc;
  CSN_HIP_CHECK(hipEventRecord(ev, from));
  CSN_HIP_CHECK(hipStreamWaitEvent(to, ev, 0));
  return CSN_OK;
}

// ---- optional event timing of the recurrence window ----------------------------------------
struct Prof {
  bool on = false;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // fwd begin/end, bwd begin/end
  int launches[2] = {0, 0}, cells[2] = {0, 0};
  bool have[2] = {false, false};
  // grouped weight-stationary paths: the recurrence launches alternate with GEMMs on the same stream, so each
  // launch is bracketed by its own event pair and the reported time is the sum over the pairs
  std::vector<hipEvent_t> pair[2];
  size_t pairs_used[2] = {0, 0};
};
static int prof_mark(Prof& g_prof, int which, hipStream_t st) {
  if (!g_prof.on) return CSN_OK;
  if ((which & 1) == 0) g_prof.pairs_used[which >> 1] = 0;
  if (g_prof.ev[which] == nullptr) CSN_HIP_CHECK(hipEventCreate(&g_prof.ev[which]));
  CSN_HIP_CHECK(hipEventRecord(g_prof.ev[which], st));
  return CSN_OK;
}

static int prof_pair(Prof& g_prof, int k, bool end, hipStream_t st) {   // k: 0 forward, 1 backward
  if (!g_prof.on) return CSN_OK;
  const size_t i = 2 * g_prof.pairs_used[k] + (end ? 1 : 0);
  while (g_prof.pair[k].size() <= i) {
    hipEvent_t e;
    CSN_HIP_CHECK(hipEventCreate(&e));
    g_prof.pair[k].push_back(e);
  }
  CSN_HIP_CHECK(hipEventRecord(g_prof.pair[k][i], st));
  if (end) ++g_prof.pairs_used[k];
  return CSN_OK;
}

}  // namespace csn

// The plan: shapes, the switches read once at creation, the workspace layout, and every piece of host-side
// state a forward / backward needs (side streams, event pool, profiling events).  Nothing of it is global, so
// plans are independent: one per (stream, thread, device) as the caller likes.  A plan is not itself
// thread-safe -- do not call the same plan from two threads at once.
struct csnLstmPlan {
  csnLstmDesc d;
  int training;
  int device;
  csn::Options opt;
  csn::WsLayout w;
  csn::SideCtx sc;
  csn::Prof prof;
  int dgates_copies = 0;      // what the last backward wrote per step (csn_lstm_plan_dgates_copies)
  csnGradReadyFn grad_cb = nullptr;      // csn_lstm_plan_set_grad_callback
  void* grad_cb_user = nullptr;
  void grads_ready(int layer) const {
    if (grad_cb != nullptr) grad_cb(grad_cb_user, layer);
  }
};

using namespace csn;
typedef csnLstmPlan Plan;

extern "C" int csn_lstm_plan_create(const csnLstmDesc* d, int training, csnLstmPlan** out) {
  if (int rc = check_desc("csn_lstm_plan_create", d)) return rc;
  CSN_REQUIRE(out != nullptr, "csn_lstm_plan_create: null output pointer");
  Plan* P = new Plan();
  P->d = *d;
  P->training = training != 0;
  if (hipGetDevice(&P->device) != hipSuccess) {
    delete P;
    return fail(CSN_ERR_HIP, "csn_lstm_plan_create: hipGetDevice failed");
  }
  P->opt = options_from_env();
  {
    // the weight-stationary kernels address the steps of ONE launch with 32-bit byte offsets from the launch's base
    // (lstm_fwd_ns.hip; launch_fwd_ns refuses more): a CSN_LSTM_CHUNK that large is clamped here, not left to wrap
    const unsigned long long Bp = ((unsigned long long)d->B + 63) / 64 * 64;
    const unsigned long long per_step = std::max({(unsigned long long)d->B * d->H * 16ull, Bp * (unsigned long long)d->I * 2ull, Bp * (unsigned long long)d->H * 2ull});
    const unsigned long long cmax = ((1ull << 32) - 1ull) / per_step;
    if (cmax >= 3 && (unsigned long long)P->opt.chunk > cmax - 2) P->opt.chunk = (int)(cmax - 2);
  }
  P->w = make_layout(*d, P->training, P->opt);
  *out = P;
  return CSN_OK;
}

extern "C" void csn_lstm_plan_destroy(csnLstmPlan* P) {
  if (P == nullptr) return;
  int cur = 0;
  const bool switched = hipGetDevice(&cur) == hipSuccess && cur != P->device && hipSetDevice(P->device) == hipSuccess;
  for (hipEvent_t e : P->sc.events) (void)hipEventDestroy(e);
  if (P->sc.side) (void)hipStreamDestroy(P->sc.side);
  if (P->sc.wgrad) (void)hipStreamDestroy(P->sc.wgrad);
  for (int l = 1; l < 8; ++l)
    if (P->sc.layer[l]) (void)hipStreamDestroy(P->sc.layer[l]);
  for (int i = 0; i < 4; ++i)
    if (P->prof.ev[i]) (void)hipEventDestroy(P->prof.ev[i]);
  for (int k = 0; k < 2; ++k)
    for (hipEvent_t e : P->prof.pair[k]) (void)hipEventDestroy(e);
  if (switched) (void)hipSetDevice(cur);
  delete P;
}

extern "C" size_t csn_lstm_plan_workspace_bytes(const csnLstmPlan* P) { return P ? P->w.total : 0; }

extern "C" int csn_lstm_plan_path(const csnLstmPlan* P) {
  if (P == nullptr) return -1;
  return P->w.f32_persist ? 4 : (P->w.persist_bwd ? 3 : (P->w.persist ? 2 : (P->w.il ? 1 : 0)));
}

extern "C" int csn_lstm_plan_dgates_copies(const csnLstmPlan* P) { return P == nullptr ? -1 : P->dgates_copies; }

// does the backward of this plan take the grouped weight-stationary form (backward_il's dispatch)?
static bool bwd_grouped(const csnLstmPlan* P) {
  if (!P->w.persist_bwd) return false;
  const int MTg = (P->d.B + 63) / 64, nchg = (P->d.T + P->opt.chunk - 1) / P->opt.chunk;
  const int slots = P->d.L < nchg ? P->d.L : nchg;
  return slots <= 4 && slots * MTg <= 8 && !P->opt.persist_streams;
}

extern "C" const char* csn_lstm_plan_kernel_name(const csnLstmPlan* P, int which) {
  if (P == nullptr) return nullptr;
  const bool ks = (P->d.dtype == CSN_F32 ? P->d.H % 128 == 0 : P->d.H % 256 == 0);      // K-split cell kernels (lstm_cell.hip)
  if (P->w.f32_persist) return which == 0 ? "lstm_fwd_f32_persist_kernel" : (which == 1 ? "lstm_bwd_f32_persist_kernel" : nullptr);
  if (which == 0) {
    if (P->w.fwd_ns) return P->opt.fwd_ws && P->d.H == 768 ? "lstm_fwd_ws_kernel" : "lstm_fwd_ns_kernel";
    if (P->w.persist) return "lstm_fwd_persist_kernel";
    if (P->w.il) return "lstm_cell_fwd_il_kernel";
    return ks ? "lstm_cell_fwd_ks_kernel" : "lstm_cell_fwd_kernel";
  }
  if (which == 1) {
    if (bwd_grouped(P)) return "lstm_bwd_persist_kernel";
    if (P->w.il) return "lstm_cell_bwd_il_kernel";
    return ks ? "lstm_cell_bwd_ks_kernel" : "lstm_cell_bwd_kernel";
  }
  return nullptr;
}

extern "C" int csn_lstm_plan_set_grad_callback(csnLstmPlan* P, csnGradReadyFn fn, void* user) {
  CSN_REQUIRE(P != nullptr, "csn_lstm_plan_set_grad_callback: null plan");
  P->grad_cb = fn;
  P->grad_cb_user = fn ? user : nullptr;
  return CSN_OK;
}

extern "C" int csn_lstm_profile_enable(csnLstmPlan* P, int on) {
  CSN_REQUIRE(P != nullptr, "csn_lstm_profile_enable: null plan");
  P->prof.on = on != 0;
  P->prof.have[0] = P->prof.have[1] = false;
  return CSN_OK;
}

extern "C" int csn_lstm_profile_read(csnLstmPlan* P, double* fwd_ms, int* fwd_launches, int* fwd_cells, double* bwd_ms,
                                     int* bwd_launches, int* bwd_cells) {
  CSN_REQUIRE(P && fwd_ms && fwd_launches && fwd_cells && bwd_ms && bwd_launches && bwd_cells,
              "csn_lstm_profile_read: null pointer");
  Prof& g_prof = P->prof;
  *fwd_ms = *bwd_ms = 0.0;
  *fwd_launches = *fwd_cells = *bwd_launches = *bwd_cells = 0;
  for (int k = 0; k < 2; ++k) {
    if (!g_prof.have[k]) continue;
    float ms = 0.f;
    CSN_HIP_CHECK(hipEventSynchronize(g_prof.ev[2 * k + 1]));
    if (g_prof.pairs_used[k] > 0) {
      for (size_t i = 0; i < g_prof.pairs_used[k]; ++i) {
        float one = 0.f;
        CSN_HIP_CHECK(hipEventElapsedTime(&one, g_prof.pair[k][2 * i], g_prof.pair[k][2 * i + 1]));
        ms += one;
      }
    } else {
      CSN_HIP_CHECK(hipEventElapsedTime(&ms, g_prof.ev[2 * k], g_prof.ev[2 * k + 1]));
    }
    (k == 0 ? *fwd_ms : *bwd_ms) = ms;
    (k == 0 ? *fwd_launches : *bwd_launches) = g_prof.launches[k];
    (k == 0 ? *fwd_cells : *bwd_cells) = g_prof.cells[k];
  }
  return CSN_OK;
}

// Once per workspace, before its first forward: the status word and everything the kernels only ever READ as zero
// (slot 0 of h_all / c_all of every layer = the zero initial state; the zero row the backward reads where a step has
// no incoming gradient)
extern "C" int csn_lstm_workspace_init(const csnLstmPlan* P, void* workspace, csnStream_t stream) {
  CSN_REQUIRE(P && workspace, "csn_lstm_workspace_init: null pointer");
  CSN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "csn_lstm_workspace_init: workspace must be 256-B aligned");
  hipStream_t st = as_stream(stream);
  char* ws = (char*)workspace;
  const WsLayout& w = P->w;
  const size_t es = dtype_size(P->d.dtype);
  CSN_HIP_CHECK(hipMemsetAsync(ws + w.status, 0, 256, st));
  for (int l = 0; l < P->d.L; ++l) {
    CSN_HIP_CHECK(hipMemsetAsync(ws + w.layer[l].h_all, 0, (size_t)P->d.B * P->d.H * es, st));
    CSN_HIP_CHECK(hipMemsetAsync(ws + w.layer[l].c_all, 0, (size_t)P->d.B * P->d.H * 4, st));
  }
  if (w.persist_bwd || (w.f32_persist && P->training))
    CSN_HIP_CHECK(hipMemsetAsync(ws + w.zeros_bh, 0, (size_t)P->d.B * P->d.H * 4, st));
  return CSN_OK;
}

// The workspace's status word (sticky; see include/csn_hip.h)
extern "C" int csn_lstm_status_clear(const csnLstmPlan* P, void* workspace, csnStream_t stream) {
  CSN_REQUIRE(P && workspace, "csn_lstm_status_clear: null pointer");
  CSN_HIP_CHECK(hipMemsetAsync((char*)workspace + P->w.status, 0, 256, as_stream(stream)));
  return CSN_OK;
}
extern "C" int csn_lstm_status_raise(const csnLstmPlan* P, void* workspace, csnStream_t stream) {
  CSN_REQUIRE(P && workspace, "csn_lstm_status_raise: null pointer");
  CSN_HIP_CHECK(hipMemsetAsync((char*)workspace + P->w.status, 1, 1, as_stream(stream)));   // word = 1, as a timed-out wait leaves it
  return CSN_OK;
}
extern "C" int csn_lstm_status_read(const csnLstmPlan* P, const void* workspace, int* status) {
  CSN_REQUIRE(P && workspace && status, "csn_lstm_status_read: null pointer");
  // word 0: a bounded wait timed out; word 1: a non-finite gradient reached the backward; word 2 (debug library built
  // with -DCSN_SLAB_TAGS only): a consumer was served a stale occupant of a hand-off ring slot
  unsigned flag[3] = {0u, 0u, 0u};
  CSN_HIP_CHECK(hipMemcpy(flag, (const char*)workspace + P->w.status, sizeof(flag), hipMemcpyDeviceToHost));
  *status = (flag[0] ? CSN_STATUS_TIMEOUT : 0) | (flag[1] ? CSN_STATUS_NONFINITE : 0) | (flag[2] ? CSN_STATUS_STALE_SLOT : 0);
#ifdef CSN_SLAB_TAGS
  if (flag[2] && getenv("CSN_TAGS_VERBOSE")) {
    unsigned dbg[24];
    CSN_HIP_CHECK(hipMemcpy(dbg, (const char*)workspace + P->w.status + 28, sizeof(dbg), hipMemcpyDeviceToHost));
    if (dbg[0])
      fprintf(stderr, "stale piece (backward): t=%u step-in-launch=%u group=%u slice=%u wave=%u lane=%u kb*4+rg=%u u0=%08x local=%u redone=%u "
              "single=%u phase=%u rot=%u nsteps=%u xcc=%u lanes=%u\n", dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7], dbg[8],
              dbg[9], dbg[10], dbg[11], dbg[12], dbg[13], dbg[14], dbg[15], dbg[16]);
  }
#endif
  return CSN_OK;
}

extern "C" size_t csn_lstm_workspace_bytes(const csnLstmDesc* d, int training) {
  if (check_desc("csn_lstm_workspace_bytes", d) != CSN_OK) return 0;
  return make_layout(*d, training, options_from_env()).total;
}

// C[M,N] = A[K,M]^T B[K,N] through the split-K slabs + their fixed-order reduction (the body of csn_gemm_tn)
static int gemm_tn_full(const void* A, const void* B, float* C, int64_t M, int64_t N, int64_t K, int dtype, void* scratch,
                        hipStream_t st, const Options& opt) {
  int S = 1;
  if (int rc = launch_gemm_tn_slabs(A, B, (float*)scratch, M, N, K, dtype, st, &S, nullptr, nullptr, opt)) return rc;
  return launch_reduce_slabs((const float*)scratch, M * N, S, C, M * N, 0, st);
}

// =============================================================================================
// v1 path
// =============================================================================================
// Wavefront over the layers, as in the fast path: diagonal d runs layer l at step d - l * lag in ONE launch (blockIdx.z =
// layer; round 3 ran layer after layer, one launch per layer-step: 2 T L launches per pass, each bound by its launch
// boundary and its own fill, not by the float32 MFMA rate).  The input projection of layer l + 1 follows layer l chunk by
// chunk (GEMM over `chunk` steps on the same stream), so lag = chunk.
static int forward_v1(Plan& P, char* ws, const float* x, int64_t xsb, int64_t xst,
                      const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                      const float* const* b_hh, int training, csnStream_t stream) {
  const csnLstmDesc* d = &P.d;
  const WsLayout& w = P.w;
  hipStream_t st = as_stream(stream);
  const int B = d->B, T = d->T, H = d->H, dt = d->dtype, NL = d->L;
  const int64_t G = 4 * (int64_t)H, TB = (int64_t)T * B;
  const size_t es = dtype_size(dt);
  const int Cz = P.opt.chunk, lag = Cz;
  int rc;
  if ((rc = launch_cast_strided(x, xsb, xst, B, T, d->I, ws + w.x_c, dt, st))) return rc;
  for (int l = 0; l < NL; ++l) {
    const LayerWs& L = w.layer[l];
    const int64_t I = l == 0 ? d->I : H;
    if ((rc = launch_cast(w_ih[l], ws + L.wih, G * I, dt, st))) return rc;
    if ((rc = launch_cast(w_hh[l], ws + L.whh, G * H, dt, st))) return rc;
    if (training) {
      if ((rc = launch_transpose_cast(w_hh[l], G, H, ws + L.whht, dt, st))) return rc;
      if ((rc = launch_transpose_cast(w_ih[l], G, I, ws + L.wiht, dt, st))) return rc;
    }
    if ((rc = launch_add_vec(b_ih[l], b_hh[l], (float*)(ws + L.bias), G, st))) return rc;
    CSN_HIP_CHECK(hipMemsetAsync(ws + L.h_all, 0, (size_t)B * H * es, st));
    CSN_HIP_CHECK(hipMemsetAsync(ws + L.c_all, 0, (size_t)B * H * 4, st));
  }
  // layer 0: projection of every step in one GEMM
  if ((rc = gemm_nt(ws + w.x_c, ws + w.layer[0].wih, (const float*)(ws + w.layer[0].bias), ws + w.layer[0].xproj, TB, G,
                    d->I, dt, CSN_F32, 0, st, P.opt)))
    return rc;
  const int D = T + lag * (NL - 1);
  for (int dg = 0; dg < D; ++dg) {
    CellFwdBatch b{};
    int np = 0;
    for (int l = 0; l < NL; ++l) {
      const int t = dg - lag * l;
      if (t < 0 || t >= T) continue;
      const LayerWs& L = w.layer[l];
      if (np == 4) {       // (more than 4 layers on one diagonal: a second launch, the problems are independent)
        if ((rc = launch_cell_fwd_batch(b, np, B, H, dt, st))) return rc;
        np = 0;
      }
      b.p[np++] = CellFwdOne{ws + L.h_all + (size_t)t * B * H * es, ws + L.whh, (const float*)(ws + L.xproj) + (size_t)t * B * G, G,
                             (const float*)(ws + L.c_all) + (size_t)t * B * H,
                             training ? ws + L.gates + (size_t)t * B * G * es : nullptr,
                             (float*)(ws + L.c_all) + (size_t)(t + 1) * B * H, ws + L.h_all + (size_t)(t + 1) * B * H * es};
    }
    if (np > 0 && (rc = launch_cell_fwd_batch(b, np, B, H, dt, st))) return rc;
    // a layer that just finished a chunk feeds the next layer's input projection
    for (int l = 0; l + 1 < NL; ++l) {
      const int t = dg - lag * l;
      if (t < 0 || t >= T || ((t + 1) % Cz != 0 && t != T - 1)) continue;
      const int t0 = (t / Cz) * Cz, nsteps = t - t0 + 1;
      const LayerWs& Ln = w.layer[l + 1];
      if ((rc = gemm_nt(ws + w.layer[l].h_all + (size_t)(t0 + 1) * B * H * es, ws + Ln.wih, (const float*)(ws + Ln.bias),
                        (float*)(ws + Ln.xproj) + (size_t)t0 * B * G, (int64_t)nsteps * B, G, H, dt, CSN_F32, 0, st, P.opt)))
        return rc;
    }
  }
  return CSN_OK;
}

static int backward_v1(Plan& P, char* ws, const float* dy_last, const float* dy_tm,
                       float* const* dw_ih, float* const* dw_hh, float* const* db_ih, float* const* db_hh, float* dx,
                       csnStream_t stream) {
  const csnLstmDesc* d = &P.d;
  const WsLayout& w = P.w;
  hipStream_t st = as_stream(stream);
  const int B = d->B, T = d->T, H = d->H, dt = d->dtype, NL = d->L;
  const int64_t G = 4 * (int64_t)H, TB = (int64_t)T * B;
  const size_t es = dtype_size(dt);
  const int Cz = P.opt.chunk, lag = Cz;
  int rc;
  for (int l = 0; l < NL; ++l) CSN_HIP_CHECK(hipMemsetAsync(ws + w.layer[l].dc_carry, 0, (size_t)B * H * 4, st));
  // diagonal d: layer l at reverse step d - lag * (L - 1 - l); the gradient w.r.t. a layer's input (= dy of the layer
  // below) follows chunk by chunk
  const int D = T + lag * (NL - 1);
  for (int dg = 0; dg < D; ++dg) {
    CellBwdBatch b{};
    int np = 0;
    for (int l = NL - 1; l >= 0; --l) {
      const int r = dg - lag * (NL - 1 - l);
      if (r < 0 || r >= T) continue;
      const int t = T - 1 - r;
      const LayerWs& L = w.layer[l];
      const bool top = (l == NL - 1);
      if (np == 4) {
        if ((rc = launch_cell_bwd_batch(b, np, B, H, dt, st))) return rc;
        np = 0;
      }
      const float* dy_t = top ? (dy_tm ? dy_tm + (size_t)t * B * H : (t == T - 1 ? dy_last : nullptr))
                              : (const float*)(ws + w.layer[l + 1].dx) + (size_t)t * B * H;
      b.p[np++] = CellBwdOne{(t == T - 1) ? nullptr : (const void*)(ws + L.dgates + (size_t)(t + 1) * B * G * es), ws + L.whht, dy_t, H,
                             ws + L.gates + (size_t)t * B * G * es, (const float*)(ws + L.c_all) + (size_t)(t + 1) * B * H,
                             (const float*)(ws + L.c_all) + (size_t)t * B * H, (float*)(ws + L.dc_carry),
                             ws + L.dgates + (size_t)t * B * G * es};
    }
    if (np > 0 && (rc = launch_cell_bwd_batch(b, np, B, H, dt, st))) return rc;
    for (int l = NL - 1; l >= 1; --l) {
      const int r = dg - lag * (NL - 1 - l);
      if (r < 0 || r >= T || ((r + 1) % Cz != 0 && r != T - 1)) continue;
      const int t_lo = T - 1 - r, t_hi = T - 1 - (r / Cz) * Cz;       // steps of this reverse chunk
      const LayerWs& L = w.layer[l];
      if ((rc = gemm_nt(ws + L.dgates + (size_t)t_lo * B * G * es, ws + L.wiht, nullptr, (float*)(ws + L.dx) + (size_t)t_lo * B * H,
                        (int64_t)(t_hi - t_lo + 1) * B, H, G, dt, CSN_F32, 0, st, P.opt)))
        return rc;
    }
  }
  for (int l = NL - 1; l >= 0; --l) {
    const LayerWs& L = w.layer[l];
    const int64_t I = l == 0 ? d->I : H;
    const void* inp = l == 0 ? (const void*)(ws + w.x_c)
                             : (const void*)(ws + w.layer[l - 1].h_all + (size_t)B * H * es);
    if ((rc = gemm_tn_full(ws + L.dgates, ws + L.h_all, dw_hh[l], G, H, TB, dt, ws + w.tn_scratch, st, P.opt))) return rc;
    if ((rc = gemm_tn_full(ws + L.dgates, inp, dw_ih[l], G, I, TB, dt, ws + w.tn_scratch, st, P.opt))) return rc;
    if ((rc = launch_colsum(ws + L.dgates, TB, G, dt, db_ih[l], ws + w.colsum, st))) return rc;
    CSN_HIP_CHECK(hipMemcpyAsync(db_hh[l], db_ih[l], (size_t)G * 4, hipMemcpyDeviceToDevice, st));
    P.grads_ready(l);
  }
  if (dx) {
    const LayerWs& L = w.layer[0];
    if ((rc = gemm_nt(ws + L.dgates, ws + L.wiht, nullptr, ws + L.dx, TB, d->I, G, dt, CSN_F32, 0, st, P.opt))) return rc;
    tb_to_bt_kernel<<<grid_for(TB * d->I), 256, 0, st>>>((const float*)(ws + L.dx), dx, B, T, d->I);
    CSN_LAUNCH_CHECK();
  }
  return CSN_OK;
}

// Exact-float32 path, weight-stationary (lstm_f32_persist.hip): layer after layer, ONE recurrence launch per layer and
// block of M-tiles, the non-recurrent contractions as whole-sequence GEMMs between them (the same GEMM kernels, the same
// K order per output element as the chunked calls of forward_v1: row chunking does not enter a row's sum).
static int forward_f32p(Plan& P, char* ws, const float* x, int64_t xsb, int64_t xst,
                        const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                        const float* const* b_hh, int training, csnStream_t stream) {
  const csnLstmDesc* d = &P.d;
  const WsLayout& w = P.w;
  Prof& g_prof = P.prof;
  hipStream_t st = as_stream(stream);
  const int B = d->B, T = d->T, H = d->H, NL = d->L;
  const int64_t G = 4 * (int64_t)H, TB = (int64_t)T * B;
  const int MTt = (B + 63) / 64, per = f32_persist_tiles_per_launch(H);
  int rc;
  if ((rc = launch_cast_strided(x, xsb, xst, B, T, d->I, ws + w.x_c, CSN_F32, st))) return rc;
  for (int l = 0; l < NL; ++l) {
    const LayerWs& L = w.layer[l];
    const int64_t I = l == 0 ? d->I : H;
    if ((rc = launch_cast(w_ih[l], ws + L.wih, G * I, CSN_F32, st))) return rc;
    if ((rc = launch_cast(w_hh[l], ws + L.whh, G * H, CSN_F32, st))) return rc;
    if (training) {
      if ((rc = launch_transpose_cast(w_hh[l], G, H, ws + L.whht, CSN_F32, st))) return rc;
      if ((rc = launch_transpose_cast(w_ih[l], G, I, ws + L.wiht, CSN_F32, st))) return rc;
    }
    if ((rc = launch_add_vec(b_ih[l], b_hh[l], (float*)(ws + L.bias), G, st))) return rc;
    CSN_HIP_CHECK(hipMemsetAsync(ws + L.h_all, 0, (size_t)B * H * 4, st));
    CSN_HIP_CHECK(hipMemsetAsync(ws + L.c_all, 0, (size_t)B * H * 4, st));
  }
  CSN_HIP_CHECK(hipMemsetAsync(ws + w.zero_fwd, 0, w.zero_fwd_bytes, st));      // the flag lines of every layer
  int n_launch = 0;
  if ((rc = prof_mark(g_prof, 0, st))) return rc;
  for (int l = 0; l < NL; ++l) {
    const LayerWs& L = w.layer[l];
    const void* inp = l == 0 ? (const void*)(ws + w.x_c) : (const void*)(ws + w.layer[l - 1].h_all + (size_t)B * H * 4);
    if ((rc = gemm_nt(inp, ws + L.wih, (const float*)(ws + L.bias), ws + L.xproj, TB, G, l == 0 ? d->I : H, CSN_F32, CSN_F32, 0, st, P.opt)))
      return rc;
    F32PersistFwdArgs a{};
    a.w_hh = (const float*)(ws + L.whh);
    a.xproj = (const float*)(ws + L.xproj);
    a.gates = training ? (float*)(ws + L.gates) : nullptr;
    a.c_all = (float*)(ws + L.c_all);
    a.h_all = (float*)(ws + L.h_all);
    a.h_blk = (float*)(ws + L.h_blk_all);
    a.flags = (unsigned*)(ws + L.counters);
    a.error_flag = (unsigned*)(ws + w.status);
    a.B = B; a.T = T; a.MT_total = MTt;
    for (int m = 0; m < MTt; m += per) {
      a.mt0 = m;
      a.MT = MTt - m < per ? MTt - m : per;
      if ((rc = prof_pair(g_prof, 0, false, st))) return rc;
      if ((rc = launch_fwd_f32_persist(a, H, st))) return rc;
      if ((rc = prof_pair(g_prof, 0, true, st))) return rc;
      ++n_launch;
    }
  }
  if ((rc = prof_mark(g_prof, 1, st))) return rc;
  g_prof.launches[0] = n_launch;
  g_prof.cells[0] = T * NL;
  g_prof.have[0] = g_prof.on;
  return CSN_OK;
}

static int backward_f32p(Plan& P, char* ws, const float* dy_last, const float* dy_tm,
                         float* const* dw_ih, float* const* dw_hh, float* const* db_ih, float* const* db_hh, float* dx,
                         csnStream_t stream) {
  const csnLstmDesc* d = &P.d;
  const WsLayout& w = P.w;
  Prof& g_prof = P.prof;
  hipStream_t st = as_stream(stream);
  const int B = d->B, T = d->T, H = d->H, NL = d->L;
  const int64_t G = 4 * (int64_t)H, TB = (int64_t)T * B;
  const int MTt = (B + 63) / 64, per = f32_persist_tiles_per_launch(H);
  int rc;
  CSN_HIP_CHECK(hipMemsetAsync(ws + w.zero_bwd, 0, w.zero_bwd_bytes, st));
  int n_launch = 0;
  if ((rc = prof_mark(g_prof, 2, st))) return rc;
  for (int l = NL - 1; l >= 0; --l) {
    const LayerWs& L = w.layer[l];
    const bool top = (l == NL - 1);
    F32PersistBwdArgs a{};
    a.w_hh_t = (const float*)(ws + L.whht);
    a.gates = (const float*)(ws + L.gates);
    a.c_all = (const float*)(ws + L.c_all);
    a.dy = top ? dy_tm : (const float*)(ws + w.layer[l + 1].dx);
    a.dy_last = (top && dy_tm == nullptr) ? dy_last : nullptr;
    a.zeros = (const float*)(ws + w.zeros_bh);
    a.dgates = (float*)(ws + L.dgates);
    a.dg_blk = (float*)(ws + L.dg_blk_all);
    a.bias_part = (float*)(ws + w.f32_bias_part);
    a.flags = (unsigned*)(ws + L.bflags);
    a.error_flag = (unsigned*)(ws + w.status);
    a.B = B; a.T = T; a.MT_total = MTt;
    for (int m = 0; m < MTt; m += per) {
      a.mt0 = m;
      a.MT = MTt - m < per ? MTt - m : per;
      if ((rc = prof_pair(g_prof, 1, false, st))) return rc;
      if ((rc = launch_bwd_f32_persist(a, H, st))) return rc;
      if ((rc = prof_pair(g_prof, 1, true, st))) return rc;
      ++n_launch;
    }
    // bias gradients: the row groups' partial sums in fixed order (db_ih = db_hh)
    sum_rows_kernel<<<(unsigned)((G + 255) / 256), 256, 0, st>>>((const float*)(ws + w.f32_bias_part), MTt * 4, G, db_ih[l], db_hh[l]);
    CSN_LAUNCH_CHECK();
    // gradient w.r.t. this layer's input = dy of the layer below, whole sequence
    if (l > 0 && (rc = gemm_nt(ws + L.dgates, ws + L.wiht, nullptr, ws + L.dx, TB, H, G, CSN_F32, CSN_F32, 0, st, P.opt))) return rc;
  }
  if ((rc = prof_mark(g_prof, 3, st))) return rc;
  g_prof.launches[1] = n_launch;
  g_prof.cells[1] = T * NL;
  g_prof.have[1] = g_prof.on;
  for (int l = NL - 1; l >= 0; --l) {
    const LayerWs& L = w.layer[l];
    const int64_t I = l == 0 ? d->I : H;
    const void* inp = l == 0 ? (const void*)(ws + w.x_c) : (const void*)(ws + w.layer[l - 1].h_all + (size_t)B * H * 4);
    if ((rc = gemm_tn_full(ws + L.dgates, ws + L.h_all, dw_hh[l], G, H, TB, CSN_F32, ws + w.tn_scratch, st, P.opt))) return rc;
    if ((rc = gemm_tn_full(ws + L.dgates, inp, dw_ih[l], G, I, TB, CSN_F32, ws + w.tn_scratch, st, P.opt))) return rc;
    P.grads_ready(l);
  }
  if (dx) {
    const LayerWs& L = w.layer[0];
    if ((rc = gemm_nt(ws + L.dgates, ws + L.wiht, nullptr, ws + L.dx, TB, d->I, G, CSN_F32, CSN_F32, 0, st, P.opt))) return rc;
    tb_to_bt_kernel<<<grid_for(TB * d->I), 256, 0, st>>>((const float*)(ws + L.dx), dx, B, T, d->I);
    CSN_LAUNCH_CHECK();
  }
  return CSN_OK;
}

// =============================================================================================
// il fast path (wavefront over layers, GEMMs on the side stream)
// =============================================================================================
static int forward_persist(Plan& P, char* ws, int training, hipStream_t st, SideCtx* sc, hipStream_t side);

static int forward_il(Plan& P, char* ws, const float* x, int64_t xsb, int64_t xst,
                      const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                      const float* const* b_hh, int training, csnStream_t stream) {
  const csnLstmDesc* d = &P.d;
  const WsLayout& w = P.w;
  Prof& g_prof = P.prof;
  hipStream_t st = as_stream(stream);
  SideCtx* sc = &P.sc;
  int rc;
  if ((rc = side_ctx(P.sc))) return rc;
  hipStream_t side = P.opt.no_side_stream ? st : sc->side;
  const int B = d->B, T = d->T, H = d->H, NL = d->L;
  const int64_t G = 4 * (int64_t)H, TB = (int64_t)T * B;
  const size_t Bpad = ((size_t)B + 63) / 64 * 64;
  const int Cz = P.opt.chunk, lag = 2 * Cz;

  // all layout preparation in one launch (prep_multi_kernel)
  PrepArgs pa{};
  auto job = [&](int kind, const float* a_, const float* b_, void* dst, int64_t n0, int64_t n1, int64_t n2, int64_t s0,
                 int64_t s1, int64_t Hh, int pr, int pk, int64_t work) {
    PrepJob& J = pa.job[pa.njobs++];
    J = PrepJob{kind, a_, b_, dst, n0, n1, n2, s0, s1, Hh, pr, pk, work, 0u, 0u};
  };
  job(kPrepCastX, x, nullptr, ws + w.x_c, B, T, d->I, xsb, xst, 0, 0, 0, TB * d->I);
  for (int l = 0; l < NL; ++l) {
    const LayerWs& L = w.layer[l];
    const int64_t I = l == 0 ? d->I : H;
    job(kPrepPermRows, w_ih[l], nullptr, ws + L.wih, 0, I, 0, 0, 0, H, 0, 0, G * I);
    job(kPrepBlockify, w_hh[l], nullptr, ws + L.whh_blk, G, H, 0, H, 1, H, 1, 0, G * H / 8);
    job(kPrepBias, b_ih[l], b_hh[l], ws + L.bias, 0, 0, 0, 0, 0, H, 0, 0, G);
    if (training) {
      job(kPrepTransPerm, w_ih[l], nullptr, ws + L.wiht, 0, I, 0, 0, 0, H, 0, 0, G * I);
      // W_hh^T [H rows = unit][k' = 4u'+g]: element (u, k') = W_hh[std_row(k')][u]
      job(kPrepBlockify, w_hh[l], nullptr, ws + L.whht_blk, H, G, 0, 1, H, H, 0, 1, H * G / 8);
    }
    if (!w.persist) {       // ping-pong hand-off buffers of the per-timestep launches
      CSN_HIP_CHECK(hipMemsetAsync(ws + L.hblk[0], 0, Bpad * H * 2, st));
      CSN_HIP_CHECK(hipMemsetAsync(ws + L.hblk[1], 0, Bpad * H * 2, st));
      // (weight-stationary paths: slot 0 is never written, csn_lstm_workspace_init zeroed it)
      CSN_HIP_CHECK(hipMemsetAsync(ws + L.h_all, 0, (size_t)B * H * 2, st));
      CSN_HIP_CHECK(hipMemsetAsync(ws + L.c_all, 0, (size_t)B * H * 4, st));
    }
  }
  if (w.fuse_x) {
    // layer 0 multiplies x_t itself inside the weight-stationary kernel: fragment-major x and W_ih instead of
    // a [T, B, 4H] float32 projection written to and re-read from HBM
    job(kPrepBlockifyX, x, nullptr, ws + w.x_blk, B, T, d->I, xsb, xst, (int64_t)Bpad, 0, 0, (int64_t)T * Bpad * d->I / 8);
    job(kPrepBlockify, w_ih[0], nullptr, ws + w.wih0_blk, G, d->I, 0, d->I, 1, H, 1, 0, G * d->I / 8);
  }
  if ((rc = launch_prep_multi(pa, st))) return rc;
  if (!w.fuse_x) {
    // layer 0 input projection for every step, main stream
    if ((rc = gemm_nt(ws + w.x_c, ws + w.layer[0].wih, (const float*)(ws + w.layer[0].bias), ws + w.layer[0].xproj,
                          TB, G, d->I, CSN_BF16, CSN_F32, 0, st, P.opt)))
      return rc;
  }
  if (w.persist) return forward_persist(P, ws, training, st, sc, side);
  if (NL > 1 && (rc = hand_off(sc, st, side))) return rc;   // side stream sees the prepared weights

  const int nch = (T + Cz - 1) / Cz;
  std::vector<hipEvent_t> xproj_ready((size_t)NL * nch, nullptr);
  const int D = T + lag * (NL - 1);
  int n_launch = 0, n_cells = 0;
  if ((rc = prof_mark(g_prof, 0, st))) return rc;
  for (int dg = 0; dg < D; ++dg) {
    CellFwdArgs a{};
    a.B = B;
    a.H = H;
    int np = 0;
    for (int l = 0; l < NL; ++l) {
      const int t = dg - lag * l;
      if (t < 0 || t >= T) continue;
      const LayerWs& L = w.layer[l];
      if (l > 0 && t % Cz == 0) CSN_HIP_CHECK(hipStreamWaitEvent(st, xproj_ready[(size_t)l * nch + t / Cz], 0));
      CellFwdProb& P = a.p[np++];
      P.h_prev_blk = t == 0 ? nullptr : (const bf16_t*)(ws + L.hblk[t & 1]);
      P.w_blk = (const bf16_t*)(ws + L.whh_blk);
      P.xproj = (const float*)(ws + L.xproj) + (size_t)t * B * G;
      P.c_prev = t == 0 ? nullptr : (const float*)(ws + L.c_all) + (size_t)t * B * H;
      P.gates_out = training ? (bf16_t*)(ws + L.gates) + (size_t)t * B * G : nullptr;
      P.c_out = (float*)(ws + L.c_all) + (size_t)(t + 1) * B * H;
      P.h_out = (bf16_t*)(ws + L.h_all) + (size_t)(t + 1) * B * H;
      P.h_out_blk = (bf16_t*)(ws + L.hblk[(t + 1) & 1]);
    }
    if (np == 0) continue;
    if ((rc = launch_cell_fwd_il(a, np, st, P.opt.fwd_nk))) return rc;
    ++n_launch;
    n_cells += np;
    // a layer that just finished a chunk feeds the next layer's input projection (side stream)
    for (int l = 0; l + 1 < NL; ++l) {
      const int t = dg - lag * l;
      if (t < 0 || t >= T) continue;
      if ((t + 1) % Cz != 0 && t != T - 1) continue;
      const int c = t / Cz, t0 = c * Cz, nsteps = t - t0 + 1;
      if ((rc = hand_off(sc, st, side))) return rc;
      const LayerWs& Ln = w.layer[l + 1];
      rc = gemm_nt((const bf16_t*)(ws + w.layer[l].h_all) + (size_t)(t0 + 1) * B * H, ws + Ln.wih,
                       (const float*)(ws + Ln.bias), (float*)(ws + Ln.xproj) + (size_t)t0 * B * G,
                       (int64_t)nsteps * B, G, H, CSN_BF16, CSN_F32, 0, side, P.opt);
      if (rc) return rc;
      hipEvent_t ev;
      if ((rc = next_event(sc, &ev))) return rc;
      CSN_HIP_CHECK(hipEventRecord(ev, side));
      xproj_ready[(size_t)(l + 1) * nch + c] = ev;
    }
  }
  if ((rc = prof_mark(g_prof, 1, st))) return rc;
  g_prof.launches[0] = n_launch;
  g_prof.cells[0] = n_cells;
  g_prof.have[0] = g_prof.on;
  return CSN_OK;
}

// Weight-stationary forward (lstm_fwd_persist.hip).
//
// Grouped form (the fast one, taken when slots * M-tiles <= 8): ONE launch per chunk diagonal advances layer
// l through chunk (dg - l) for every layer in range -- at cfg2 two layers x four 64-row M-tiles = 8 hand-off
// groups of 32 workgroups, one group per XCD under the round-robin dispatch, so a group's h hand-off stays
// inside one L2 (verified per launch by the kernel; otherwise it uses the placement-independent protocol).
// The input projection of layer l+1 for the chunk layer l just finished is a GEMM between two launches, on
// the same stream: the persistent workgroups own every CU (408 VGPRs, 100 KB LDS), nothing co-resides.
//
// Stream form (any number of layers / M-tiles): layer l runs chunk after chunk on its own stream, the GEMMs
// on the side stream, ordered by events; placement-independent hand-off.
static int forward_persist(Plan& P, char* ws, int training, hipStream_t st, SideCtx* sc, hipStream_t side) {
  const csnLstmDesc* d = &P.d;
  const WsLayout& w = P.w;
  Prof& g_prof = P.prof;
  const int B = d->B, T = d->T, H = d->H, NL = d->L;
  const int64_t G = 4 * (int64_t)H;
  const int Bpad = (B + 63) / 64 * 64, MT = Bpad / 64;
  const int Cz = P.opt.chunk;
  const int nch = (T + Cz - 1) / Cz;
  int rc;
  // one fill: the flag lines of every layer, the XCD agreement words, the GEMM tile counters
  CSN_HIP_CHECK(hipMemsetAsync(ws + w.zero_fwd, 0, w.zero_fwd_bytes, st));
  auto fill_slot = [&](PersistFwdSlot& S, int l, int c) {
    const LayerWs& L = w.layer[l];
    S.w_blk = (const bf16_t*)(ws + L.whh_blk);
    S.xproj = (const float*)(ws + L.xproj);
    S.gates = training ? (bf16_t*)(ws + L.gates) : nullptr;
    S.c_all = (float*)(ws + L.c_all);
    S.h_all = (bf16_t*)(ws + L.h_all);
    S.h_blk_all = (bf16_t*)(ws + L.h_blk_all);
    S.flags = (unsigned*)(ws + L.counters);
    S.x_blk = nullptr;
    S.wih_blk = nullptr;
    S.bias = nullptr;
    S.I = 0;
    S.xproj_bf16 = 0;
    if (l == 0 && w.fuse_x) {
      S.x_blk = (const bf16_t*)(ws + w.x_blk);
      S.wih_blk = (const bf16_t*)(ws + w.wih0_blk);
      S.bias = (const float*)(ws + L.bias);
      S.I = d->I;
    }
    S.t0 = c * Cz;
    S.nsteps = (S.t0 + Cz <= T) ? Cz : T - S.t0;
  };
  auto xproj_gemm = [&](int l, int c, hipStream_t on) {   // layer l finished chunk c -> xproj_{l+1}[chunk c]
    const LayerWs& L = w.layer[l];
    const LayerWs& Ln = w.layer[l + 1];
    const int t0 = c * Cz, nsteps = (t0 + Cz <= T) ? Cz : T - t0;
    return gemm_nt((const bf16_t*)(ws + L.h_all) + (size_t)(t0 + 1) * B * H, ws + Ln.wih,
                       (const float*)(ws + Ln.bias), (float*)(ws + Ln.xproj) + (size_t)t0 * B * G,
                       (int64_t)nsteps * B, G, H, CSN_BF16, CSN_F32, 0, on, P.opt);
  };
  PersistFwdArgs a{};
  a.error_flag = (unsigned*)(ws + w.status);
  a.B = B; a.H = H; a.T = T; a.Bpad = Bpad; a.MT = MT;
  a.rotate = !P.opt.no_rotate;
  // (forward: no hint words by default -- the first k-block's own pieces are what the wave spins on; measured 205 -> 200 us
  // per launch; CSN_FWD_HINT restores them)
  a.data_polls = (!w.fwd_ns && !P.opt.fwd_flags) ? (P.opt.fwd_hint && !P.opt.dpoll_no_hint ? 1 : 2) : 0;
#ifdef CSN_SLAB_TAGS
  if (a.data_polls && P.opt.tags_no_rearm) a.data_polls |= 4;
#endif
  if (a.data_polls)        // the ring of 4 hand-off slabs of every layer starts as sentinel (lstm_fwd_persist.hip)
    for (int l = 0; l < NL; ++l)
      CSN_HIP_CHECK(hipMemsetAsync(ws + w.layer[l].h_blk_all, 0xff, (size_t)4 * Bpad * H * 2, st));
  a.chains = (w.fwd_ns && H == 768 && P.opt.fwd_ws) ? 4 : 1;
  a.half_tiles = (w.fwd_ns && H == 768 && P.opt.fwd_halves && a.chains == 1) ? 1 : 0;
  int n_launch = 0;

  const int max_slots = NL < nch ? NL : nch;
  const int fwd_slices = w.fwd_ns ? fwd_ns_slices(H) : fwd_persist_slices(H);
  auto launch_fwd = [&](const PersistFwdArgs& args, hipStream_t on) {
#ifdef CSN_EXPERIMENTS
    if (args.chains == 4) return launch_fwd_ws(args, on);
#endif
    return w.fwd_ns ? launch_fwd_ns(args, on) : launch_fwd_persist(args, on);
  };
  const bool grouped = max_slots <= 4 && max_slots * MT <= 8 && fwd_slices <= 32 &&
                       !P.opt.persist_streams;
  if (grouped) {
    // CSN_BESIDE_FWD (N-split kernel, H <= 768): 24 of the 32 CUs of an XCD carry a group, the 8 others (and every CU
    // of an XCD without a group) walk the input-projection GEMM xproj_{l+1}[chunk] = h_l[chunk] W_ih^T + b of the chunk
    // layer l finished ONE LAUNCH AGO (gemm_beside.h): both of its dependencies are then kernel boundaries, and the
    // layer above lags two chunks -- the arrangement of the backward launches.  Measured at cfg2 (profiles/r02_c): the
    // GEMM needs 17 k CU-us as a kernel of its own (67 us on 256 CUs) but 26 k beside the recurrence (4-wave tiles,
    // per-CU load bandwidth shared with nothing but itself), and a forward launch leaves only 64 CUs x 215 us = 14 k:
    // the launches stretch to 263 us and the step ends where it started (10.98 vs 10.96 ms).  Not the default.
    const bool beside = w.fwd_ns && NL > 1 && fwd_slices <= 28 && H % 64 == 0 && P.opt.beside_fwd;
    const bool xproj_bf16 = beside && P.opt.xproj_bf16;
    const int lag = beside ? 2 : 1;
    const int ndiag = nch + lag * (NL - 1);
    const bool try_local = !P.opt.no_xcd_local;
    BesideGemm pending[3];
    int npending = 0;
    a.grid_slices = 32;
    if ((rc = prof_mark(g_prof, 0, st))) return rc;
    for (int dg = 0; dg < ndiag; ++dg) {
      int lay[4], chk[4], ns = 0;
      for (int l = 0; l < NL; ++l) {
        const int c = dg - lag * l;
        if (c < 0 || c >= nch) continue;
        lay[ns] = l;
        chk[ns] = c;
        fill_slot(a.slot[ns], l, c);
        a.slot[ns].xproj_bf16 = (l > 0 && xproj_bf16) ? 1 : 0;
        ++ns;
      }
      a.nslots = ns;
      a.xcd_groups = 1;
      a.ngemm = npending;
      for (int i = 0; i < npending; ++i) a.gemm[i] = pending[i];
      npending = 0;
      a.agree = try_local ? (unsigned long long*)(ws + w.agree) + (size_t)dg * 8 : nullptr;
      if (ns == 0 && a.ngemm == 0) continue;
      if ((rc = prof_pair(g_prof, 0, false, st))) return rc;
      if ((rc = launch_fwd(a, st))) return rc;
      if ((rc = prof_pair(g_prof, 0, true, st))) return rc;
      ++n_launch;
      for (int i = 0; i < ns; ++i) {
        const int l = lay[i];
        if (l + 1 >= NL) continue;
        if (!beside) {
          if ((rc = xproj_gemm(l, chk[i], st))) return rc;
          continue;
        }
        const LayerWs& L = w.layer[l];
        const LayerWs& Ln = w.layer[l + 1];
        const int t0 = chk[i] * Cz, nst = (t0 + Cz <= T) ? Cz : T - t0;
        // (xproj in bf16: half the GEMM's C stream and half of what the recurrence reads back per step; the tile
        // counter lets the recurrence workgroups take tiles once their chunk is done)
        float* Cx = xproj_bf16 ? (float*)((bf16_t*)(ws + Ln.xproj) + (size_t)t0 * B * G) : (float*)(ws + Ln.xproj) + (size_t)t0 * B * G;
        pending[npending] = BesideGemm{(const bf16_t*)(ws + L.h_all) + (size_t)(t0 + 1) * B * H, (const bf16_t*)(ws + Ln.wih),
                                       Cx, nst * B, (int)G, H, (const float*)(ws + Ln.bias),
                                       (unsigned*)(ws + w.tile_ctr) + (size_t)dg * 4 + npending, xproj_bf16 ? 1 : 0};
        ++npending;
      }
    }
    if ((rc = prof_mark(g_prof, 1, st))) return rc;
    g_prof.launches[0] = n_launch;
    g_prof.cells[0] = T * NL;
    g_prof.have[0] = g_prof.on;
    return CSN_OK;
  }

  // every launch needs ALL its workgroups resident (one per CU): layers on streams of their own may only run side
  // by side while together they fit the chip; otherwise everything goes down the caller's stream, layer after layer
  if (NL * fwd_slices * MT > 256) side = st;
  hipStream_t ls[8];
  ls[0] = st;
  for (int l = 1; l < NL; ++l) {
    ls[l] = (side == st) ? st : sc->layer[l];
    if (ls[l] != st && (rc = hand_off(sc, st, ls[l]))) return rc;
  }
  if (side != st && (rc = hand_off(sc, st, side))) return rc;
  const bool gemm_slot = P.opt.gemm_slot;
  if ((rc = prof_mark(g_prof, 0, st))) return rc;
  a.nslots = 1;
  a.xcd_groups = 0;
  a.agree = nullptr;
  for (int c = 0; c < nch; ++c) {
    for (int l = 0; l < NL; ++l) {
      fill_slot(a.slot[0], l, c);
      if ((rc = launch_fwd(a, ls[l]))) return rc;
      ++n_launch;
      if (l + 1 < NL) {
        // layer l finished chunk c -> GEMM xproj_{l+1}[chunk] on the side stream -> layer l+1 may start it
        if ((rc = hand_off(sc, ls[l], side))) return rc;
        if ((rc = xproj_gemm(l, c, side))) return rc;
        if (side != ls[l + 1] && (rc = hand_off(sc, side, ls[l + 1]))) return rc;
        // optional: the producing layer's next chunk also waits, so the GEMM runs in a slot of its own
        if (gemm_slot && side != ls[l] && (rc = hand_off(sc, side, ls[l]))) return rc;
      }
    }
  }
  // the caller's stream resumes after every layer stream (and the side stream) has drained
  for (int l = 1; l < NL; ++l)
    if (ls[l] != st && (rc = hand_off(sc, ls[l], st))) return rc;
  if (side != st && (rc = hand_off(sc, side, st))) return rc;
  if ((rc = prof_mark(g_prof, 1, st))) return rc;
  g_prof.launches[0] = n_launch;
  g_prof.cells[0] = T * NL;
  g_prof.have[0] = g_prof.on;
  return CSN_OK;
}

static int backward_persist(Plan& P, char* ws, const float* dy_last,
                            const float* dy_tm, float* const* dw_ih, float* const* dw_hh, float* const* db_ih,
                            float* const* db_hh, float* dx, hipStream_t st);

static int backward_il(Plan& P, char* ws, const float* dy_last, const float* dy_tm,
                       float* const* dw_ih, float* const* dw_hh, float* const* db_ih, float* const* db_hh, float* dx,
                       csnStream_t stream) {
  const csnLstmDesc* d = &P.d;
  const WsLayout& w = P.w;
  Prof& g_prof = P.prof;
  hipStream_t st = as_stream(stream);
  if (bwd_grouped(&P)) return backward_persist(P, ws, dy_last, dy_tm, dw_ih, dw_hh, db_ih, db_hh, dx, st);
  SideCtx* sc = &P.sc;
  int rc;
  if ((rc = side_ctx(P.sc))) return rc;
  hipStream_t side = P.opt.no_side_stream ? st : sc->side;
  const int B = d->B, T = d->T, H = d->H, NL = d->L;
  const int64_t G = 4 * (int64_t)H, TB = (int64_t)T * B;
  const size_t Bpad = ((size_t)B + 63) / 64 * 64;
  const int Cz = P.opt.chunk, lag = 2 * Cz;
  const int nch = (T + Cz - 1) / Cz;

  for (int l = 0; l < NL; ++l) {
    const LayerWs& L = w.layer[l];
    CSN_HIP_CHECK(hipMemsetAsync(ws + L.dc_carry, 0, (size_t)B * H * 4, st));
    CSN_HIP_CHECK(hipMemsetAsync(ws + L.dgblk[0], 0, Bpad * G * 2, st));
    CSN_HIP_CHECK(hipMemsetAsync(ws + L.dgblk[1], 0, Bpad * G * 2, st));
  }
  if ((rc = hand_off(sc, st, side))) return rc;

  // weight / bias gradients of one layer, on the side stream (after its recurrence is complete)
  auto weight_grads = [&](int l) -> int {
    const LayerWs& L = w.layer[l];
    const int64_t I = l == 0 ? d->I : H;
    const void* inp = l == 0 ? (const void*)(ws + w.x_c)
                             : (const void*)((const bf16_t*)(ws + w.layer[l - 1].h_all) + (size_t)B * H);
    float* slabs = (float*)(ws + w.tn_scratch);
    int S = 1, r;
    int cs_done = 0, S_cs = 1;
    if ((r = launch_gemm_tn_slabs(ws + L.dgates, ws + L.h_all, slabs, G, H, TB, CSN_BF16, side, &S, (float*)(ws + w.colsum), &cs_done, P.opt))) return r;
    S_cs = S;
    if ((r = launch_reduce_slabs_unperm(slabs, G * H, S, H, H, dw_hh[l], side))) return r;
    if ((r = launch_gemm_tn_slabs(ws + L.dgates, inp, slabs, G, I, TB, CSN_BF16, side, &S, nullptr, nullptr, P.opt))) return r;
    if ((r = launch_reduce_slabs_unperm(slabs, G * I, S, H, I, dw_ih[l], side))) return r;
    // bias gradient = column sums of dgates: partial sums come out of the dW_hh GEMM when its kernel provides them
    if (!cs_done) {
      if ((r = launch_colsum_partial(ws + L.dgates, TB, G, CSN_BF16, ws + w.colsum, side))) return r;
      S_cs = colsum_chunks();
    }
    if ((r = launch_reduce_slabs_unperm((const float*)(ws + w.colsum), G, S_cs, H, 1, db_ih[l], side)))
      return r;
    CSN_HIP_CHECK(hipMemcpyAsync(db_hh[l], db_ih[l], (size_t)G * 4, hipMemcpyDeviceToDevice, side));
    return CSN_OK;
  };

  std::vector<hipEvent_t> dx_ready((size_t)NL * nch, nullptr);
  const int D = T + lag * (NL - 1);
  int n_launch = 0, n_cells = 0;
  if ((rc = prof_mark(g_prof, 2, st))) return rc;
  for (int dg = 0; dg < D; ++dg) {
    CellBwdArgs a{};
    a.B = B;
    a.H = H;
    int np = 0;
    for (int l = NL - 1; l >= 0; --l) {
      const int r = dg - lag * (NL - 1 - l);      // reverse step index of layer l on this diagonal
      if (r < 0 || r >= T) continue;
      const int t = T - 1 - r;
      const LayerWs& L = w.layer[l];
      const bool top = (l == NL - 1);
      if (!top && r % Cz == 0) CSN_HIP_CHECK(hipStreamWaitEvent(st, dx_ready[(size_t)(l + 1) * nch + r / Cz], 0));
      CellBwdProb& P = a.p[np++];
      P.dg_next_blk = (t == T - 1) ? nullptr : (const bf16_t*)(ws + L.dgblk[(t + 1) & 1]);
      P.wt_blk = (const bf16_t*)(ws + L.whht_blk);
      if (top) {
        P.dy = dy_tm ? dy_tm + (size_t)t * B * H : (t == T - 1 ? dy_last : nullptr);
      } else {
        P.dy = (const float*)(ws + w.layer[l + 1].dx) + (size_t)t * B * H;
      }
      P.dy_ld = H;
      P.gates = (const bf16_t*)(ws + L.gates) + (size_t)t * B * G;
      P.c = (const float*)(ws + L.c_all) + (size_t)(t + 1) * B * H;
      P.c_prev = t == 0 ? nullptr : (const float*)(ws + L.c_all) + (size_t)t * B * H;
      P.dc_carry = (float*)(ws + L.dc_carry);
      P.dg_out = (bf16_t*)(ws + L.dgates) + (size_t)t * B * G;
      P.dg_out_blk = (bf16_t*)(ws + L.dgblk[t & 1]);
    }
    if (np == 0) continue;
    if ((rc = launch_cell_bwd_il(a, np, st))) return rc;
    ++n_launch;
    n_cells += np;
    for (int l = NL - 1; l >= 0; --l) {
      const int r = dg - lag * (NL - 1 - l);
      if (r < 0 || r >= T) continue;
      const bool chunk_end = ((r + 1) % Cz == 0) || r == T - 1;
      if (!chunk_end) continue;
      const LayerWs& L = w.layer[l];
      const int64_t I = l == 0 ? d->I : H;
      const int cr = r / Cz;
      const int t_lo = T - 1 - r, t_hi = T - 1 - cr * Cz;       // steps covered by this (reverse) chunk
      const bool last = (r == T - 1);
      if (l > 0 || last) {
        if ((rc = hand_off(sc, st, side))) return rc;
      }
      if (l > 0) {
        // dx_l[chunk] = dgates_l[chunk] (interleaved K) * W_ih;  Bt = W_ih^T [I, 4H']
        rc = gemm_nt((const bf16_t*)(ws + L.dgates) + (size_t)t_lo * B * G, ws + L.wiht, nullptr,
                         (float*)(ws + L.dx) + (size_t)t_lo * B * I, (int64_t)(t_hi - t_lo + 1) * B, I, G, CSN_BF16,
                         CSN_F32, 0, side, P.opt);
        if (rc) return rc;
        hipEvent_t ev;
        if ((rc = next_event(sc, &ev))) return rc;
        CSN_HIP_CHECK(hipEventRecord(ev, side));
        dx_ready[(size_t)l * nch + cr] = ev;
      } else if (last && dx != nullptr) {
        rc = gemm_nt(ws + L.dgates, ws + L.wiht, nullptr, ws + L.dx, TB, I, G, CSN_BF16, CSN_F32, 0,
                         side, P.opt);
        if (rc) return rc;
        tb_to_bt_kernel<<<grid_for(TB * I), 256, 0, side>>>((const float*)(ws + L.dx), dx, B, T, (int)I);
        CSN_LAUNCH_CHECK();
      }
      if (last && (rc = weight_grads(l))) return rc;
    }
  }
  if ((rc = prof_mark(g_prof, 3, st))) return rc;
  g_prof.launches[1] = n_launch;
  g_prof.cells[1] = n_cells;
  g_prof.have[1] = g_prof.on;
  if ((rc = hand_off(sc, side, st))) return rc;   // the caller's stream resumes after all side-stream work
  // (the weight gradients of this path run on the side stream: only now are they ordered before the caller's stream)
  for (int l = NL - 1; l >= 0; --l) P.grads_ready(l);
  return CSN_OK;
}

// Weight-stationary backward recurrence (lstm_bwd_persist.hip), grouped form: ONE launch per chunk diagonal
// walks every layer in range backwards through one chunk.  The input gradient of a layer's chunk
// (dx_l = dgates_l W_ih, the dy of the layer below) is a GEMM, and it runs INSIDE the next launch: the backward
// kernel's groups occupy 24 of the 32 CUs of their XCD (24 slices of 32 units at H = 768), so the launch is
// widened to 32 workgroups per XCD and the 8 extra ones (plus all workgroups of XCDs without a group) walk
// the GEMM's tiles (gemm_beside.h).  Both of the GEMM's dependencies are then kernel boundaries: its input was
// written by the launch before, its output is read by the launch after -- for which the layer below lags TWO
// chunks: diagonal dg holds layer l at reverse chunk dg - 2 (L-1-l).  (A GEMM kernel on a second stream beside
// the launch was measured too: the event waits between the streams cost ~30 us per launch, most of the gain.)
// CSN_NO_BESIDE: lag one chunk, GEMM between two launches.
// The weight / bias gradients follow once the recurrence is complete.
static int backward_persist(Plan& P, char* ws, const float* dy_last,
                            const float* dy_tm, float* const* dw_ih, float* const* dw_hh, float* const* db_ih,
                            float* const* db_hh, float* dx, hipStream_t st) {
  const csnLstmDesc* d = &P.d;
  const WsLayout& w = P.w;
  Prof& g_prof = P.prof;
  const int B = d->B, T = d->T, H = d->H, NL = d->L;
  const int64_t G = 4 * (int64_t)H, TB = (int64_t)T * B;
  const int Bpad = (B + 63) / 64 * 64, MT = Bpad / 64;
  const int Cz = P.opt.chunk;
  const int nch = (T + Cz - 1) / Cz;
  const bool beside = NL > 1 && NL <= 4 && bwd_persist_slices(H) <= 28 && G % 64 == 0 && !P.opt.no_beside;
  const int lag = beside ? 2 : 1;
  const int ndiag = nch + lag * (NL - 1);
  int rc;
  // one fill: carried dc and flag lines of every layer, the XCD agreement words (zeros_bh: csn_lstm_workspace_init)
  CSN_HIP_CHECK(hipMemsetAsync(ws + w.zero_bwd, 0, w.zero_bwd_bytes, st));
  const bool try_local = !P.opt.no_xcd_local;

  int single_copy = 0;       // (decided below, before the first launch)
  // weight / bias gradients of one layer (its recurrence complete): four launches on stream `on`
  auto weight_grads = [&](int l, hipStream_t on, size_t scratch_off, size_t colsum_off) -> int {
    const LayerWs& L = w.layer[l];
    const int64_t I = l == 0 ? d->I : H;
    const void* inp = l == 0 ? (const void*)(ws + w.x_c)
                             : (const void*)((const bf16_t*)(ws + w.layer[l - 1].h_all) + (size_t)B * H);
    float* slabs = (float*)(ws + scratch_off);
    int S = 1, r;
    int cs_done = 0, S_cs = 1;
    const int blocked = single_copy;
    const char* dgm = ws + (blocked ? L.dg_blk_all : L.dgates);
    if ((r = launch_gemm_tn_slabs(dgm, ws + L.h_all, slabs, G, H, TB, CSN_BF16, on, &S, (float*)(ws + colsum_off), &cs_done, P.opt, blocked))) return r;
    S_cs = S;
    if ((r = launch_reduce_slabs_unperm(slabs, G * H, S, H, H, dw_hh[l], on))) return r;
    if ((r = launch_gemm_tn_slabs(dgm, inp, slabs, G, I, TB, CSN_BF16, on, &S, nullptr, nullptr, P.opt, blocked))) return r;
    if ((r = launch_reduce_slabs_unperm(slabs, G * I, S, H, I, dw_ih[l], on))) return r;
    // bias gradient = column sums of dgates: partial sums come out of the dW_hh GEMM when its kernel provides them
    if (!cs_done) {
      if (blocked) return fail(CSN_ERR_UNSUPPORTED, "backward_persist: single-copy mode without the weight-gradient kernel's column sums");
      if ((r = launch_colsum_partial(ws + L.dgates, TB, G, CSN_BF16, ws + colsum_off, on))) return r;
      S_cs = colsum_chunks();
    }
    if ((r = launch_reduce_slabs_unperm((const float*)(ws + colsum_off), G, S_cs, H, 1, db_ih[l], on))) return r;
    CSN_HIP_CHECK(hipMemcpyAsync(db_hh[l], db_ih[l], (size_t)G * 4, hipMemcpyDeviceToDevice, on));
    return CSN_OK;
  };
  // CSN_WGRAD_OVERLAP (off by default -- measured, and worse): an upper layer is done `lag` launches before the
  // bottom one; its weight-gradient GEMMs can go to a LOWEST-priority stream of the plan right then, to fill the CUs
  // the last launches leave free (the XCDs of the finished layer's groups + the 8 spare CUs of the others).  On
  // MI355X the stream priority does not keep the GEMM's workgroups off the CUs the next recurrence launch needs: its
  // workgroups (one per CU, all of a group resident before anyone advances) wait for GEMM workgroups to retire --
  // 14.4 vs 11.0 ms per step at cfg2.
  bool wg_done[8] = {false, false, false, false, false, false, false, false};
  const bool wg_overlap = NL > 1 && w.tn_scratch2 != 0 && P.opt.wgrad_overlap;
  bool wg_side = false;
  if (wg_overlap && (rc = side_ctx(P.sc))) return rc;

  PersistBwdArgs a{};
  a.error_flag = (unsigned*)(ws + w.status);
  a.B = B; a.H = H; a.T = T; a.Bpad = Bpad; a.MT = MT;
  a.xcd_groups = 1;
  a.grid_slices = 32;
  a.rotate = !P.opt.no_rotate;
  a.data_polls = P.opt.bwd_flags ? 0 : (P.opt.dpoll_no_hint ? 2 : 1);
#ifdef CSN_SLAB_TAGS
  if (a.data_polls && P.opt.tags_no_rearm) a.data_polls |= 4;
#endif
  // CSN_BWD_SINGLE_COPY (experiments library only): ONE copy of dgates (lstm_bwd_persist.hip, SINGLE) --
  // the hand-off slabs, one per step, fragment-major, are what the GEMMs after the recurrence read, and the row-major
  // copy is not written (4 of the 12 store instructions of a step).  Possible when every reader of dgates takes the
  // block layout: the weight gradients on the 256 x 256 kernel (which also delivers the bias gradient), the input
  // gradients of the upper layers inside the launch (gemm_beside.h), and nobody wants dx of layer 0.
  const int64_t I0 = d->I;
  const bool single = a.data_polls != 0 && P.opt.bwd_single_copy && dx == nullptr && !wg_overlap && B == Bpad &&
                      (NL == 1 || (beside && nch > lag)) && (int64_t)T * Bpad * G * 2 < ((int64_t)1 << 31) &&
                      gemm_tn_takes_blocked_a(G, H, TB, P.opt) && gemm_tn_takes_blocked_a(G, I0, TB, P.opt);
  a.single_copy = single_copy = single ? 1 : 0;
  P.dgates_copies = single ? 1 : 2;
  if (a.data_polls) {
    // the hand-off slabs that are polled before a kernel has armed them start as sentinel: the ring of 4, or (single
    // copy) the slabs of steps T-1 and T-2 -- slab s <= T-3 is armed by its producers at step s+2
    const int first = single ? (T >= 2 ? T - 2 : 0) : 0, count = single ? T - first : 4;
    for (int l = 0; l < NL; ++l)
      CSN_HIP_CHECK(hipMemsetAsync(ws + w.layer[l].dg_blk_all + (size_t)first * Bpad * G * 2, 0xff, (size_t)count * Bpad * G * 2, st));
  }
  int n_launch = 0;
  BesideGemm pending[3];           // GEMMs of the chunks finished by the previous launch
  int npending = 0;
  if ((rc = prof_mark(g_prof, 2, st))) return rc;
  for (int dg = 0; dg < ndiag; ++dg) {
    int lay[4], chk[4], ns = 0;
    for (int l = NL - 1; l >= 0; --l) {
      const int c = dg - lag * (NL - 1 - l);      // reverse chunk index of layer l on this diagonal
      if (c < 0 || c >= nch) continue;
      const LayerWs& L = w.layer[l];
      PersistBwdSlot& S = a.slot[ns];
      S.wt_blk = (const bf16_t*)(ws + L.whht_blk);
      S.gates = (const bf16_t*)(ws + L.gates);
      S.c_all = (const float*)(ws + L.c_all);
      if (l == NL - 1) {
        S.dy = dy_tm;
        S.dy_last = dy_tm ? nullptr : dy_last;
      } else {
        S.dy = (const float*)(ws + w.layer[l + 1].dx);
        S.dy_last = nullptr;
      }
      S.zeros = (const float*)(ws + w.zeros_bh);
      S.dc_carry = (float*)(ws + L.dc_carry);
      S.dgates = (bf16_t*)(ws + L.dgates);
      S.dg_blk_all = (bf16_t*)(ws + L.dg_blk_all);
      S.flags = (unsigned*)(ws + L.bflags);
      S.t_hi = T - 1 - c * Cz;
      S.nsteps = (c * Cz + Cz <= T) ? Cz : T - c * Cz;
      lay[ns] = l;
      chk[ns] = c;
      ++ns;
    }
    if (ns == 0) {
      // a diagonal without a layer in range (one chunk per layer, T <= chunk, and the layers two launches apart): nothing
      // to recur over, but the input-gradient GEMMs the previous launch left for "the next launch" still have to run
      for (int i = 0; i < npending; ++i) {
        const BesideGemm& g = pending[i];
        if (g.a_blocked) return fail(CSN_ERR_UNSUPPORTED, "backward_persist: fragment-major dgates without a launch to carry their GEMM");
        if ((rc = gemm_nt(g.A, g.Bt, g.bias, g.C, g.M, g.N, g.K, CSN_BF16, CSN_F32, 0, st, P.opt))) return rc;
      }
      npending = 0;
      continue;
    }
    a.nslots = ns;
    a.ngemm = npending;
    for (int i = 0; i < npending; ++i) a.gemm[i] = pending[i];
    npending = 0;
    a.agree = try_local ? (unsigned long long*)(ws + w.agree_b) + (size_t)dg * 8 : nullptr;
    if ((rc = prof_pair(g_prof, 1, false, st))) return rc;
    if ((rc = launch_bwd_persist(a, st))) return rc;
    if ((rc = prof_pair(g_prof, 1, true, st))) return rc;
    ++n_launch;
    for (int i = 0; i < ns; ++i) {
      const int l = lay[i];
      if (l == 0) continue;
      // dx_l[chunk] = dgates_l[chunk] (interleaved K) * W_ih;  Bt = W_ih^T [I, 4H']
      const LayerWs& L = w.layer[l];
      const int t_hi = T - 1 - chk[i] * Cz, t_lo = t_hi - a.slot[i].nsteps + 1;
      const bf16_t* Ag = (const bf16_t*)(ws + (a.single_copy ? L.dg_blk_all : L.dgates)) + (size_t)t_lo * B * G;
      float* Cg = (float*)(ws + L.dx) + (size_t)t_lo * B * H;
      const int64_t Mg = (int64_t)(t_hi - t_lo + 1) * B;
      if (beside) {
        pending[npending++] = BesideGemm{Ag, (const bf16_t*)(ws + L.wiht), Cg, (int)Mg, H, (int)G, nullptr, nullptr, 0, a.single_copy};
      } else {
        if ((rc = gemm_nt(Ag, ws + L.wiht, nullptr, Cg, Mg, H, G, CSN_BF16, CSN_F32, 0, st, P.opt))) return rc;
      }
    }
    if (wg_overlap) {
      for (int i = 0; i < ns; ++i) {
        const int l = lay[i];
        if (l == 0 || chk[i] != nch - 1) continue;          // layer l has just walked its last chunk (dgates_l complete)
        if ((rc = hand_off(&P.sc, st, P.sc.wgrad))) return rc;
        if ((rc = weight_grads(l, P.sc.wgrad, w.tn_scratch2, w.colsum2))) return rc;
        wg_done[l] = true;
        wg_side = true;
      }
    }
  }
  if ((rc = prof_mark(g_prof, 3, st))) return rc;
  g_prof.launches[1] = n_launch;
  g_prof.cells[1] = T * NL;
  g_prof.have[1] = g_prof.on;

  if (dx != nullptr) {
    const LayerWs& L = w.layer[0];
    if ((rc = gemm_nt(ws + L.dgates, ws + L.wiht, nullptr, ws + L.dx, TB, d->I, G, CSN_BF16, CSN_F32, 0,
                          st, P.opt)))
      return rc;
    tb_to_bt_kernel<<<grid_for(TB * d->I), 256, 0, st>>>((const float*)(ws + L.dx), dx, B, T, d->I);
    CSN_LAUNCH_CHECK();
  }
  // (every recurrence launch has been enqueued by now: what a gradient-ready callback starts -- a collective on another
  // stream -- runs beside the remaining layers' weight-gradient GEMMs, never beside a one-workgroup-per-CU launch)
  for (int l = NL - 1; l >= 0; --l) {
    if (wg_done[l]) continue;
    if ((rc = weight_grads(l, st, w.tn_scratch, w.colsum))) return rc;
    if (!wg_side) P.grads_ready(l);
  }
  if (wg_side) {
    if ((rc = hand_off(&P.sc, P.sc.wgrad, st))) return rc;     // the caller's stream resumes after the side work
    for (int l = NL - 1; l >= 0; --l) P.grads_ready(l);
  }
  return CSN_OK;
}

// =============================================================================================
extern "C" int csn_lstm_forward(csnLstmPlan* Pp, const float* x, int64_t x_stride_b, int64_t x_stride_t,
                                const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                                const float* const* b_hh, void* workspace, float* y_last, float* y_all,
                                csnStream_t stream) {
  CSN_REQUIRE(Pp != nullptr, "csn_lstm_forward: null plan");
  Plan& P = *Pp;
  const csnLstmDesc* d = &P.d;
  CSN_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && workspace, "csn_lstm_forward: null pointer");
  CSN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "csn_lstm_forward: workspace must be 256-B aligned");
  CSN_REQUIRE(y_last || y_all, "csn_lstm_forward: no output requested");
  for (int l = 0; l < d->L; ++l)
    CSN_REQUIRE(w_ih[l] && w_hh[l] && b_ih[l] && b_hh[l], "csn_lstm_forward: null parameter pointer, layer %d", l);
  int dev = -1;
  CSN_HIP_CHECK(hipGetDevice(&dev));
  CSN_REQUIRE(dev == P.device, "csn_lstm_forward: plan was created on device %d, current device is %d", P.device, dev);
  hipStream_t st = as_stream(stream);
  const WsLayout& w = P.w;
  const int training = P.training;
  char* ws = (char*)workspace;
  const int B = d->B, T = d->T, H = d->H, dt = d->dtype;
  const size_t es = dtype_size(dt);
  int rc;
  // (the status word is NOT cleared here: it stays raised from the first timed-out hand-off until
  // csn_lstm_status_clear, so a check at the end of an epoch / a timed region covers every step in it)
  if (w.il)
    rc = forward_il(P, ws, x, x_stride_b, x_stride_t, w_ih, w_hh, b_ih, b_hh, training, stream);
  else if (w.f32_persist)
    rc = forward_f32p(P, ws, x, x_stride_b, x_stride_t, w_ih, w_hh, b_ih, b_hh, training, stream);
  else
    rc = forward_v1(P, ws, x, x_stride_b, x_stride_t, w_ih, w_hh, b_ih, b_hh, training, stream);
  if (rc) return rc;
  const LayerWs& top = w.layer[d->L - 1];
  if (y_last)
    if ((rc = launch_upcast(ws + top.h_all + (size_t)T * B * H * es, dt, y_last, (int64_t)B * H, st))) return rc;
  if (y_all) {
    const int64_t n = (int64_t)T * B * H;
    if (dt == CSN_BF16)
      gather_y_all_kernel<bf16_t><<<grid_for(n), 256, 0, st>>>((const bf16_t*)(ws + top.h_all), y_all, B, T, H);
    else
      gather_y_all_kernel<float><<<grid_for(n), 256, 0, st>>>((const float*)(ws + top.h_all), y_all, B, T, H);
    CSN_LAUNCH_CHECK();
  }
  return CSN_OK;
}

extern "C" int csn_lstm_backward(csnLstmPlan* Pp, const float* dy_last, const float* dy_all, void* workspace,
                                 float* const* dw_ih, float* const* dw_hh, float* const* db_ih, float* const* db_hh,
                                 float* dx, csnStream_t stream) {
  CSN_REQUIRE(Pp != nullptr, "csn_lstm_backward: null plan");
  Plan& P = *Pp;
  const csnLstmDesc* d = &P.d;
  CSN_REQUIRE(P.training, "csn_lstm_backward: the plan was created with training = 0");
  CSN_REQUIRE(workspace && dw_ih && dw_hh && db_ih && db_hh, "csn_lstm_backward: null pointer");
  CSN_REQUIRE(dy_last || dy_all, "csn_lstm_backward: no incoming gradient");
  for (int l = 0; l < d->L; ++l)
    CSN_REQUIRE(dw_ih[l] && dw_hh[l] && db_ih[l] && db_hh[l], "csn_lstm_backward: null gradient pointer, layer %d", l);
  int dev = -1;
  CSN_HIP_CHECK(hipGetDevice(&dev));
  CSN_REQUIRE(dev == P.device, "csn_lstm_backward: plan was created on device %d, current device is %d", P.device, dev);
  hipStream_t st = as_stream(stream);
  const WsLayout& w = P.w;
  char* ws = (char*)workspace;
  const int B = d->B, T = d->T, H = d->H;
  const int64_t TB = (int64_t)T * B;

  // gradient w.r.t. the top layer's outputs, time-major.  With only dy_last, no buffer is needed.
  const float* dy_tm = nullptr;
  if (dy_all) {
    float* buf = (float*)(ws + w.dy_tm);
    bt_to_tb_kernel<<<grid_for(TB * H), 256, 0, st>>>(dy_all, buf, B, T, H);
    CSN_LAUNCH_CHECK();
    if (dy_last) {
      add_rows_kernel<<<grid_for((int64_t)B * H), 256, 0, st>>>(dy_last, buf + (size_t)(T - 1) * B * H, (int64_t)B * H);
      CSN_LAUNCH_CHECK();
    }
    dy_tm = buf;
  }
  if (w.il) return backward_il(P, ws, dy_last, dy_tm, dw_ih, dw_hh, db_ih, db_hh, dx, stream);
  if (w.f32_persist) return backward_f32p(P, ws, dy_last, dy_tm, dw_ih, dw_hh, db_ih, db_hh, dx, stream);
  return backward_v1(P, ws, dy_last, dy_tm, dw_ih, dw_hh, db_ih, db_hh, dx, stream);
}
