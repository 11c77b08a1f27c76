// K3 orchestration: stacked LSTM forward / backward over the whole sequence.
// Replaces nn.LSTM(batch_first=True).forward / autograd backward as called at
// /root/reference/LSTMDistill.py:118,132 and /root/reference/LSTMDistillRetreival.py:91,103.
//
// Layout in HBM (all inside the caller-provided workspace; [T,B,*] time-major so that one
// timestep is one contiguous slab that the per-step cell kernel streams):
//   per layer l:  Wih_c[4H,I_l]  Whh_c[4H,H]  WhhT_c[H,4H]  WihT_c[I_l,4H]   compute dtype copies
//                 bias[4H] f32 (= b_ih + b_hh)
//                 xproj[T,B,4H] f32          input projection + bias, one big GEMM per layer
//                 gates[T,B,4H] dtype        post-activation i,f,g,o (saved for backward)
//                 c_all[T+1,B,H] f32, h_all[T+1,B,H] dtype   (slot 0 = zero initial state)
//                 dgates[T,B,4H] dtype       pre-activation gradients (backward)
//   shared:       x_c[T,B,I] dtype (time-major copy of the input), dx_buf[T,B,H] f32 (gradient
//                 flowing into the layer below), dc_carry[B,H] f32, dy_tm[T,B,H] f32 (optional),
//                 GEMM split-K slabs, column-sum partials.
// The recurrence is a stream of per-timestep launches on one HIP stream (a launch boundary
// is the cheapest grid-wide hand-off on this chip, ~1.5 us); the two big
// non-recurrent contractions per layer (input projection, weight gradients) are single GEMMs.
#include <vector>

#include "csn_common.h"

namespace csn {

int launch_cell_fwd(const void* h_prev, const void* w_hh, const float* xproj, int64_t xproj_ld, const float* c_prev,
                    void* gates_out, float* c_out, void* h_out, int B, int H, int dtype, hipStream_t st);
int launch_cell_bwd(const void* dg_next, const void* w_hh_t, const float* dy, int64_t dy_ld, const void* gates,
                    const float* c, const float* c_prev, float* dc_carry, void* dg_out, int B, int H, int dtype,
                    hipStream_t st);

struct LayerWs {
  size_t wih, whh, whht, wiht, bias, xproj, gates, c_all, h_all, dgates;
};
struct WsLayout {
  LayerWs layer[8];
  size_t x_c, dx_buf[2], dc_carry, dy_tm, tn_scratch, colsum, total;
};

static WsLayout make_layout(const csnLstmDesc& d, int training) {
  WsLayout w{};
  size_t off = 0;
  const size_t es = dtype_size(d.dtype);
  auto take = [&](size_t bytes) {
    size_t o = off;
    off += align_up(bytes, 256);
    return o;
  };
  const size_t TB = (size_t)d.T * d.B, H = d.H, G = 4 * (size_t)d.H;
  size_t tn_bytes = 0;
  for (int l = 0; l < d.L; ++l) {
    const size_t I = l == 0 ? d.I : d.H;
    LayerWs& L = w.layer[l];
    L.wih = take(G * I * es);
    L.whh = take(G * H * es);
    L.whht = take(G * H * es);
    L.wiht = take(G * I * es);
    L.bias = take(G * 4);
    L.xproj = take(TB * G * 4);
    L.gates = take(TB * G * es);
    L.c_all = take((TB + d.B) * H * 4);
    L.h_all = take((TB + d.B) * H * es);
    L.dgates = training ? take(TB * G * es) : 0;
    size_t a = csn_gemm_tn_scratch_bytes(G, I, TB), b = csn_gemm_tn_scratch_bytes(G, H, TB);
    if (a > tn_bytes) tn_bytes = a;
    if (b > tn_bytes) tn_bytes = b;
  }
  w.x_c = take(TB * d.I * es);
  if (training) {
    const size_t widest = (size_t)(d.I > d.H ? d.I : d.H);
    w.dx_buf[0] = take(TB * widest * 4);
    w.dx_buf[1] = take(TB * widest * 4);
    w.dc_carry = take((size_t)d.B * H * 4);
    w.dy_tm = take(TB * H * 4);
    w.tn_scratch = take(tn_bytes);
    w.colsum = take(colsum_scratch_bytes(G));
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

// dst[t_sel][b][h] += src[b][h]   (adds dy_last into the time-major dy buffer)
__global__ void add_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] += src[i];
}

static inline unsigned grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return (unsigned)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace csn

using namespace csn;

extern "C" size_t csn_lstm_workspace_bytes(const csnLstmDesc* d, int training) {
  if (check_desc("csn_lstm_workspace_bytes", d) != CSN_OK) return 0;
  return make_layout(*d, training).total;
}

extern "C" int csn_lstm_forward(const csnLstmDesc* d, const float* x, int64_t x_stride_b, int64_t x_stride_t,
                                const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                                const float* const* b_hh, void* workspace, int training, float* y_last, float* y_all,
                                csnStream_t stream) {
  if (int rc = check_desc("csn_lstm_forward", d)) return rc;
  CSN_REQUIRE(x && w_ih && w_hh && b_ih && b_hh && workspace, "csn_lstm_forward: null pointer");
  CSN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "csn_lstm_forward: workspace must be 256-B aligned");
  CSN_REQUIRE(y_last || y_all, "csn_lstm_forward: no output requested");
  hipStream_t st = as_stream(stream);
  const WsLayout w = make_layout(*d, training);
  char* ws = (char*)workspace;
  const int B = d->B, T = d->T, H = d->H, dt = d->dtype;
  const int64_t G = 4 * (int64_t)H, TB = (int64_t)T * B;
  const size_t es = dtype_size(dt);
  int rc;

  // time-major copy of the input in the compute dtype: x_c[t][b][i]
  if ((rc = launch_cast_strided(x, x_stride_b, x_stride_t, B, T, d->I, ws + w.x_c, dt, st))) return rc;

  for (int l = 0; l < d->L; ++l) {
    const LayerWs& L = w.layer[l];
    const int64_t I = l == 0 ? d->I : H;
    CSN_REQUIRE(w_ih[l] && w_hh[l] && b_ih[l] && b_hh[l], "csn_lstm_forward: null parameter pointer, layer %d", l);
    // compute-dtype copies of the weights (+ transposes for the backward pass)
    if ((rc = launch_cast(w_ih[l], ws + L.wih, G * I, dt, st))) return rc;
    if ((rc = launch_cast(w_hh[l], ws + L.whh, G * H, dt, st))) return rc;
    if (training) {
      if ((rc = launch_transpose_cast(w_hh[l], G, H, ws + L.whht, dt, st))) return rc;
      if ((rc = launch_transpose_cast(w_ih[l], G, I, ws + L.wiht, dt, st))) return rc;
    }
    if ((rc = launch_add_vec(b_ih[l], b_hh[l], (float*)(ws + L.bias), G, st))) return rc;

    // xproj[T*B, 4H] = inp[T*B, I] * W_ih^T + bias
    const void* inp = l == 0 ? (const void*)(ws + w.x_c)
                             : (const void*)(ws + w.layer[l - 1].h_all + (size_t)B * H * es);
    if ((rc = csn_gemm_nt(inp, ws + L.wih, (const float*)(ws + L.bias), ws + L.xproj, TB, G, I, dt, CSN_F32, 0,
                          stream)))
      return rc;

    CSN_HIP_CHECK(hipMemsetAsync(ws + L.h_all, 0, (size_t)B * H * es, st));
    CSN_HIP_CHECK(hipMemsetAsync(ws + L.c_all, 0, (size_t)B * H * 4, st));
    for (int t = 0; t < T; ++t) {
      const char* h_prev = ws + L.h_all + (size_t)t * B * H * es;
      const float* c_prev = (const float*)(ws + L.c_all) + (size_t)t * B * H;
      rc = launch_cell_fwd(h_prev, ws + L.whh, (const float*)(ws + L.xproj) + (size_t)t * B * G, G, c_prev,
                           training ? ws + L.gates + (size_t)t * B * G * es : nullptr,
                           (float*)(ws + L.c_all) + (size_t)(t + 1) * B * H,
                           ws + L.h_all + (size_t)(t + 1) * B * H * es, B, H, dt, st);
      if (rc) return rc;
    }
  }
  const LayerWs& top = w.layer[d->L - 1];
  if (y_last)
    if ((rc = launch_upcast(ws + top.h_all + (size_t)T * B * H * es, dt, y_last, (int64_t)B * H, st))) return rc;
  if (y_all) {
    if (dt == CSN_BF16)
      gather_y_all_kernel<bf16_t><<<grid_for(TB * H), 256, 0, st>>>((const bf16_t*)(ws + top.h_all), y_all, B, T, H);
    else
      gather_y_all_kernel<float><<<grid_for(TB * H), 256, 0, st>>>((const float*)(ws + top.h_all), y_all, B, T, H);
    CSN_LAUNCH_CHECK();
  }
  return CSN_OK;
}

extern "C" int csn_lstm_backward(const csnLstmDesc* d, const float* dy_last, const float* dy_all, void* workspace,
                                 float* const* dw_ih, float* const* dw_hh, float* const* db_ih, float* const* db_hh,
                                 float* dx, csnStream_t stream) {
  if (int rc = check_desc("csn_lstm_backward", d)) return rc;
  CSN_REQUIRE(workspace && dw_ih && dw_hh && db_ih && db_hh, "csn_lstm_backward: null pointer");
  CSN_REQUIRE(dy_last || dy_all, "csn_lstm_backward: no incoming gradient");
  hipStream_t st = as_stream(stream);
  const WsLayout w = make_layout(*d, 1);
  char* ws = (char*)workspace;
  const int B = d->B, T = d->T, H = d->H, dt = d->dtype;
  const int64_t G = 4 * (int64_t)H, TB = (int64_t)T * B;
  const size_t es = dtype_size(dt);
  int rc;

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

  for (int l = d->L - 1; l >= 0; --l) {
    const LayerWs& L = w.layer[l];
    const int64_t I = l == 0 ? d->I : H;
    CSN_REQUIRE(dw_ih[l] && dw_hh[l] && db_ih[l] && db_hh[l], "csn_lstm_backward: null gradient pointer, layer %d", l);
    const bool top = (l == d->L - 1);
    const float* dy_src = top ? dy_tm : (const float*)(ws + w.dx_buf[l & 1]);   // [T,B,H] or null
    CSN_HIP_CHECK(hipMemsetAsync(ws + w.dc_carry, 0, (size_t)B * H * 4, st));
    for (int t = T - 1; t >= 0; --t) {
      const float* dy_t = dy_src ? dy_src + (size_t)t * B * H : ((top && t == T - 1) ? dy_last : nullptr);
      const void* dg_next = (t == T - 1) ? nullptr : (const void*)(ws + L.dgates + (size_t)(t + 1) * B * G * es);
      rc = launch_cell_bwd(dg_next, ws + L.whht, dy_t, H, ws + L.gates + (size_t)t * B * G * es,
                           (const float*)(ws + L.c_all) + (size_t)(t + 1) * B * H,
                           (const float*)(ws + L.c_all) + (size_t)t * B * H, (float*)(ws + w.dc_carry),
                           ws + L.dgates + (size_t)t * B * G * es, B, H, dt, st);
      if (rc) return rc;
    }
    // weight gradients: contraction over all T*B rows
    const void* inp = l == 0 ? (const void*)(ws + w.x_c)
                             : (const void*)(ws + w.layer[l - 1].h_all + (size_t)B * H * es);
    if ((rc = csn_gemm_tn(ws + L.dgates, ws + L.h_all, dw_hh[l], G, H, TB, dt, ws + w.tn_scratch, stream))) return rc;
    if ((rc = csn_gemm_tn(ws + L.dgates, inp, dw_ih[l], G, I, TB, dt, ws + w.tn_scratch, stream))) return rc;
    if ((rc = launch_colsum(ws + L.dgates, TB, G, dt, db_ih[l], ws + w.colsum, st))) return rc;
    CSN_HIP_CHECK(hipMemcpyAsync(db_hh[l], db_ih[l], (size_t)G * 4, hipMemcpyDeviceToDevice, st));
    // gradient flowing to the layer below: dx[T*B, I] = dgates[T*B, 4H] * W_ih   (Bt = W_ih^T [I,4H])
    if (l > 0) {
      if ((rc = csn_gemm_nt(ws + L.dgates, ws + L.wiht, nullptr, ws + w.dx_buf[(l - 1) & 1], TB, I, G, dt, CSN_F32, 0,
                            stream)))
        return rc;
    } else if (dx) {
      float* tmp = (float*)(ws + w.dx_buf[1]);   // time-major [T,B,I], then re-laid batch-first for the caller
      if ((rc = csn_gemm_nt(ws + L.dgates, ws + L.wiht, nullptr, tmp, TB, I, G, dt, CSN_F32, 0, stream))) return rc;
      tb_to_bt_kernel<<<grid_for(TB * I), 256, 0, st>>>(tmp, dx, B, T, (int)I);
      CSN_LAUNCH_CHECK();
    }
  }
  return CSN_OK;
}
