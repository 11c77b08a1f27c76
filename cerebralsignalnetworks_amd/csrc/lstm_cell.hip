// K3: one LSTM cell step, forward and backward, as a recurrent GEMM with the gate math fused
// into its epilogue.  Replaces the per-timestep body of nn.LSTM (ATen) that the reference
// reaches at /root/reference/LSTMDistill.py:118,132 and LSTMDistillRetreival.py:91,103.
//
// Forward step:   a = xproj_t + h_{t-1} W_hh^T ; i,f,o = sigmoid, g = tanh ; c_t = f c_{t-1} + i g ;
//                 h_t = o tanh(c_t)                               (gate order i,f,g,o)
// Backward step:  dh = dy_t + dgates_{t+1} W_hh ; do, dc, di, df, dg ; dgates_t (pre-activation grads)
//
// Decomposition: a wave owns a 16 (batch rows) x 16 (hidden units) cell tile; a workgroup is
// 4 waves = 64 batch rows x 16 units; grid = (H/16, ceil(B/64)).  For a tile the four gate
// pre-activations of one (row, unit) land in the SAME lane and register index of four MFMA
// accumulators (W_hh rows u, H+u, 2H+u, 3H+u are four B-tiles sharing one A fragment), so the
// whole gate computation is lane-local: no shuffles, no LDS.  MFMA operands are swapped
// (W as the A operand, h as the B operand) so that each lane holds 4 CONSECUTIVE units of one
// batch row: xproj / c loads and gates / c / h stores are 16-byte (f32) or 8-byte (bf16) vectors.
// Operand fragments are k-contiguous in memory for both W_hh[4H,H] and h[B,H] and are loaded
// straight from L2 to VGPRs (16 B per lane); the step is latency-bound, not MFMA-bound.
//
// bf16 variant: v_mfma_f32_16x16x32_bf16, f32 accumulate, f32 cell state, bf16 h / gates.
// f32 variant : v_mfma_f32_16x16x4_f32 (exact f32 fma chain) -- the parity path.
#include "csn_common.h"
#include "lstm_cell_common.h"

namespace csn {

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  typedef bf16x8 type;
  static constexpr int kStep = 32;  // k covered by one fragment load (8 per lane x 4 lane groups)
  static __device__ __forceinline__ type load(const bf16_t* row_ptr, int k0, int lane) {
    return *reinterpret_cast<const bf16x8*>(row_ptr + k0 + 8 * (lane >> 4));
  }
  static __device__ __forceinline__ type zero() { return (bf16x8){0, 0, 0, 0, 0, 0, 0, 0}; }
  static __device__ __forceinline__ f32x4 mma(const type& a, const type& b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  }
};
template <> struct Frag<float> {
  typedef f32x4 type;
  static constexpr int kStep = 16;  // 4 per lane x 4 lane groups
  static __device__ __forceinline__ type load(const float* row_ptr, int k0, int lane) {
    return *reinterpret_cast<const f32x4*>(row_ptr + k0 + 4 * (lane >> 4));
  }
  static __device__ __forceinline__ type zero() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
  static __device__ __forceinline__ f32x4 mma(const type& a, const type& b, f32x4 acc) {
    // MFMA jj contracts k = k0 + 4*(lane>>4) + jj over the 4 lane groups; A and B use the same map.
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[jj], b[jj], acc, 0, 0, 0);
    return acc;
  }
};

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
lstm_cell_fwd_kernel(const T* __restrict__ h_prev, const T* __restrict__ w_hh, const float* __restrict__ xproj,
                     int64_t xproj_ld, const float* __restrict__ c_prev, T* __restrict__ gates_out,
                     float* __restrict__ c_out, T* __restrict__ h_out, int B, int H) {
  typedef Frag<T> F;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u0 = blockIdx.x * 16;
  const int mrow = blockIdx.y * 64 + wave * 16 + (lane & 15);   // batch row whose h fragment this lane loads
  const bool row_ok = mrow < B;
  const T* hrow = h_prev + (int64_t)(row_ok ? mrow : 0) * H;
  const T* wrow[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) wrow[g] = w_hh + ((int64_t)g * H + u0 + (lane & 15)) * H;

  f32x4 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (h_prev != nullptr) {
#pragma unroll 4
    for (int k0 = 0; k0 < H; k0 += F::kStep) {
      typename F::type hf = row_ok ? F::load(hrow, k0, lane) : F::zero();
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typename F::type wf = F::load(wrow[g], k0, lane);
        acc[g] = F::mma(wf, hf, acc[g]);   // D[row = unit][col = batch]
      }
    }
  }

  // lane holds batch row (lane & 15) and units u0 + (lane>>4)*4 + r, r = 0..3
  if (!row_ok) return;
  const int ub = u0 + (lane >> 4) * 4;
  float pre[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float xp[4];
    Vec4<float>::load(xproj + (int64_t)mrow * xproj_ld + (int64_t)g * H + ub, xp);
#pragma unroll
    for (int r = 0; r < 4; ++r) pre[g][r] = acc[g][r] + xp[r];
  }
  float cp[4] = {0.f, 0.f, 0.f, 0.f};
  if (c_prev != nullptr) Vec4<float>::load(c_prev + (int64_t)mrow * H + ub, cp);
  float gi[4], gf[4], gg[4], go[4], cn[4], hn[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    gi[r] = sigmoid_f32(pre[0][r]);
    gf[r] = sigmoid_f32(pre[1][r]);
    gg[r] = tanh_f32(pre[2][r]);
    go[r] = sigmoid_f32(pre[3][r]);
    cn[r] = gf[r] * cp[r] + gi[r] * gg[r];
    hn[r] = go[r] * tanh_f32(cn[r]);
  }
  if (gates_out != nullptr) {
    T* gp = gates_out + (int64_t)mrow * 4 * H + ub;
    Vec4<T>::store(gp, gi);
    Vec4<T>::store(gp + H, gf);
    Vec4<T>::store(gp + 2 * (int64_t)H, gg);
    Vec4<T>::store(gp + 3 * (int64_t)H, go);
  }
  Vec4<float>::store(c_out + (int64_t)mrow * H + ub, cn);
  Vec4<T>::store(h_out + (int64_t)mrow * H + ub, hn);
}

// ------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
lstm_cell_bwd_kernel(const T* __restrict__ dg_next, const T* __restrict__ w_hh_t, const float* __restrict__ dy,
                     int64_t dy_ld, const T* __restrict__ gates, const float* __restrict__ c,
                     const float* __restrict__ c_prev, float* __restrict__ dc_carry, T* __restrict__ dg_out, int B,
                     int H) {
  typedef Frag<T> F;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u0 = blockIdx.x * 16;
  const int mrow = blockIdx.y * 64 + wave * 16 + (lane & 15);
  const bool row_ok = mrow < B;
  const int K = 4 * H;

  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (dg_next != nullptr) {
    const T* drow = dg_next + (int64_t)(row_ok ? mrow : 0) * K;
    const T* wrow = w_hh_t + (int64_t)(u0 + (lane & 15)) * K;   // W_hh^T[u][k]
#pragma unroll 8
    for (int k0 = 0; k0 < K; k0 += F::kStep) {
      typename F::type df = row_ok ? F::load(drow, k0, lane) : F::zero();
      typename F::type wf = F::load(wrow, k0, lane);
      acc = F::mma(wf, df, acc);   // D[row = unit][col = batch] = sum_k W_hh^T[u][k] dg_next[b][k]
    }
  }
  if (!row_ok) return;
  const int ub = u0 + (lane >> 4) * 4;
  float dh[4] = {acc[0], acc[1], acc[2], acc[3]};
  if (dy != nullptr) {
    float d[4];
    Vec4<float>::load(dy + (int64_t)mrow * dy_ld + ub, d);
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[r] += d[r];
  }
  float gi[4], gf[4], gg[4], go[4], cc[4], cp[4] = {0.f, 0.f, 0.f, 0.f}, dcn[4];
  const T* gp = gates + (int64_t)mrow * K + ub;
  Vec4<T>::load(gp, gi);
  Vec4<T>::load(gp + H, gf);
  Vec4<T>::load(gp + 2 * (int64_t)H, gg);
  Vec4<T>::load(gp + 3 * (int64_t)H, go);
  Vec4<float>::load(c + (int64_t)mrow * H + ub, cc);
  if (c_prev != nullptr) Vec4<float>::load(c_prev + (int64_t)mrow * H + ub, cp);
  Vec4<float>::load(dc_carry + (int64_t)mrow * H + ub, dcn);
  float dai[4], daf[4], dag[4], dao[4], dcarry[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float tc = tanh_f32(cc[r]);
    const float d_o = dh[r] * tc;
    const float dc = dh[r] * go[r] * (1.0f - tc * tc) + dcn[r];
    dai[r] = dc * gg[r] * gi[r] * (1.0f - gi[r]);
    daf[r] = dc * cp[r] * gf[r] * (1.0f - gf[r]);
    dag[r] = dc * gi[r] * (1.0f - gg[r] * gg[r]);
    dao[r] = d_o * go[r] * (1.0f - go[r]);
    dcarry[r] = dc * gf[r];
  }
  T* op = dg_out + (int64_t)mrow * K + ub;
  Vec4<T>::store(op, dai);
  Vec4<T>::store(op + H, daf);
  Vec4<T>::store(op + 2 * (int64_t)H, dag);
  Vec4<T>::store(op + 3 * (int64_t)H, dao);
  Vec4<float>::store(dc_carry + (int64_t)mrow * H + ub, dcarry);
}


int launch_cell_fwd(const void* h_prev, const void* w_hh, const float* xproj, int64_t xproj_ld, const float* c_prev,
                    void* gates_out, float* c_out, void* h_out, int B, int H, int dtype, hipStream_t st) {
  dim3 grid((unsigned)(H / 16), (unsigned)((B + 63) / 64));
  if (dtype == CSN_BF16)
    lstm_cell_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)h_prev, (const bf16_t*)w_hh, xproj, xproj_ld,
                                                       c_prev, (bf16_t*)gates_out, c_out, (bf16_t*)h_out, B, H);
  else
    lstm_cell_fwd_kernel<float><<<grid, 256, 0, st>>>((const float*)h_prev, (const float*)w_hh, xproj, xproj_ld, c_prev,
                                                      (float*)gates_out, c_out, (float*)h_out, B, H);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

int launch_cell_bwd(const void* dg_next, const void* w_hh_t, const float* dy, int64_t dy_ld, const void* gates,
                    const float* c, const float* c_prev, float* dc_carry, void* dg_out, int B, int H, int dtype,
                    hipStream_t st) {
  dim3 grid((unsigned)(H / 16), (unsigned)((B + 63) / 64));
  if (dtype == CSN_BF16)
    lstm_cell_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)dg_next, (const bf16_t*)w_hh_t, dy, dy_ld,
                                                       (const bf16_t*)gates, c, c_prev, dc_carry, (bf16_t*)dg_out, B, H);
  else
    lstm_cell_bwd_kernel<float><<<grid, 256, 0, st>>>((const float*)dg_next, (const float*)w_hh_t, dy, dy_ld,
                                                      (const float*)gates, c, c_prev, dc_carry, (float*)dg_out, B, H);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

}  // namespace csn

using namespace csn;

static int check_cell_args(const char* fn, int B, int H, int dtype) {
  CSN_REQUIRE(B > 0 && H > 0, "%s: bad shape B=%d H=%d", fn, B, H);
  CSN_REQUIRE(H % 32 == 0, "%s: H=%d must be a multiple of 32", fn, H);
  CSN_REQUIRE(dtype == CSN_F32 || dtype == CSN_BF16, "%s: bad dtype %d", fn, dtype);
  return CSN_OK;
}

extern "C" int csn_lstm_cell_forward(const void* h_prev, const void* w_hh, const float* xproj, int64_t xproj_ld,
                                     const float* c_prev, void* gates_out, float* c_out, void* h_out, int B, int H,
                                     int dtype, csnStream_t stream) {
  if (int rc = check_cell_args("csn_lstm_cell_forward", B, H, dtype)) return rc;
  CSN_REQUIRE(w_hh && xproj && c_out && h_out, "csn_lstm_cell_forward: null pointer");
  CSN_REQUIRE(xproj_ld % 4 == 0, "csn_lstm_cell_forward: xproj_ld must be a multiple of 4");
  return launch_cell_fwd(h_prev, w_hh, xproj, xproj_ld, c_prev, gates_out, c_out, h_out, B, H, dtype,
                         as_stream(stream));
}

extern "C" int csn_lstm_cell_backward(const void* dgates_next, const void* w_hh_t, const float* dy, int64_t dy_ld,
                                      const void* gates, const float* c, const float* c_prev, float* dc_carry,
                                      void* dgates_out, int B, int H, int dtype, csnStream_t stream) {
  if (int rc = check_cell_args("csn_lstm_cell_backward", B, H, dtype)) return rc;
  CSN_REQUIRE(gates && c && dc_carry && dgates_out, "csn_lstm_cell_backward: null pointer");
  CSN_REQUIRE(dgates_next == nullptr || w_hh_t != nullptr, "csn_lstm_cell_backward: w_hh_t is null");
  CSN_REQUIRE(dy == nullptr || dy_ld % 4 == 0, "csn_lstm_cell_backward: dy_ld must be a multiple of 4");
  return launch_cell_bwd(dgates_next, w_hh_t, dy, dy_ld, gates, c, c_prev, dc_carry, dgates_out, B, H, dtype,
                         as_stream(stream));
}
