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

#define CSN_CELL_FWD_BIND                                                                                    \
  const CellFwdOne& q_ = batch.p[blockIdx.z];                                                                \
  const T* __restrict__ h_prev = (const T*)q_.h_prev; const T* __restrict__ w_hh = (const T*)q_.w_hh;        \
  const float* __restrict__ xproj = q_.xproj; const int64_t xproj_ld = q_.xproj_ld;                          \
  const float* __restrict__ c_prev = q_.c_prev; T* __restrict__ gates_out = (T*)q_.gates_out;                \
  float* __restrict__ c_out = q_.c_out; T* __restrict__ h_out = (T*)q_.h_out
#define CSN_CELL_BWD_BIND                                                                                    \
  const CellBwdOne& q_ = batch.p[blockIdx.z];                                                                \
  const T* __restrict__ dg_next = (const T*)q_.dg_next; const T* __restrict__ w_hh_t = (const T*)q_.w_hh_t;  \
  const float* __restrict__ dy = q_.dy; const int64_t dy_ld = q_.dy_ld; const T* __restrict__ gates = (const T*)q_.gates; \
  const float* __restrict__ c = q_.c; const float* __restrict__ c_prev = q_.c_prev;                          \
  float* __restrict__ dc_carry = q_.dc_carry; T* __restrict__ dg_out = (T*)q_.dg_out

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
lstm_cell_fwd_kernel(CellFwdBatch batch, int B, int H) {
  CSN_CELL_FWD_BIND;
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
lstm_cell_bwd_kernel(CellBwdBatch batch, int B, int H) {
  CSN_CELL_BWD_BIND;
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


// ------------------------------------------------------------------------------------------
// K-split forms (round 3): the same arithmetic with the chip filled AND the operands reused.  The kernels above give every
// 16 x 16 cell tile to ONE wave that walks all of K -- at B = 256, H = 768 that is 768 tiles on 1024 SIMDs, one long chain
// of dependent MFMAs each (768 of them in the backward, K = 4H) fed by loads nothing overlaps: 33 / 64 us per step and
// layer in float32 (12 - 23 % of the float32 MFMA rate).  A first K-split (one 16 x 16 tile per 4-wave workgroup) filled the
// chip but re-read W_hh once per 16 batch rows: 188 / 302 MB per launch through L2 -> CU at the fabric's 8 TB/s, 26 / 36 us.
// Here a workgroup owns 64 rows x 16 units (forward, 4 waves) or 64 rows x 32 units (backward, 8 waves); wave w contracts
// K slice w for ALL of the tile (each weight fragment feeds 4 row groups), the partial tiles meet in LDS and every wave
// finishes one of them (forward: row group w; backward: (row group, unit tile) w).  Same operand map and epilogue as above;
// the float32 sum is formed from 4 / 8 partial sums.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void cell_fwd_epilogue(f32x4 (&acc)[4], int mrow, int ub, const float* xproj, int64_t xproj_ld,
                                                  const float* c_prev, T* gates_out, float* c_out, T* h_out, int H) {
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

template <typename T>
__global__ void __launch_bounds__(256)
lstm_cell_fwd_ks_kernel(CellFwdBatch batch, int B, int H) {
  CSN_CELL_FWD_BIND;
  typedef Frag<T> F;
  // partial tiles on their way to the wave that finishes them: [dst row group][src wave, without dst][gate][lane]
  __shared__ f32x4 part[4][3][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u0 = blockIdx.x * 16, m0 = blockIdx.y * 64;
  f32x4 acc[4][4];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[rg][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (h_prev != nullptr) {
    const T* hrow[4];
    bool ok[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int m = m0 + rg * 16 + (lane & 15);
      ok[rg] = m < B;
      hrow[rg] = h_prev + (int64_t)(ok[rg] ? m : 0) * H;
    }
    const T* wrow[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) wrow[g] = w_hh + ((int64_t)g * H + u0 + (lane & 15)) * H;
    // Each k-step takes ONE 128-byte line per operand row (two fragment steps: a 16-byte fragment load covers 64 bytes of
    // each of its 16 rows; with the second half requested a phase later the line had left L1 again and crossed the
    // L2 -> CU fabric twice).  All loads unconditional (rows beyond B read row 0: their columns of the product are never
    // stored; a select on a loaded value puts a wait for the load right behind it).
    const int kq = H >> 2, k_beg = wave * kq, n = kq / (2 * F::kStep);
    // two stages: the lines of k-step i + 1 are requested before the MFMAs of k-step i (round 4; as one stage -- load, wait,
    // multiply -- every k-step exposed its whole load latency: 24 us per cell at cfg2 against 10.7 us of f32 MFMA issue).
    // Same k order, same sums.
    typename F::type wf[2][2][4], hf[2][2][4];
    auto load = [&](int b, int i) {
      const int k0 = k_beg + i * 2 * F::kStep;
#pragma unroll
      for (int hfl = 0; hfl < 2; ++hfl) {
#pragma unroll
        for (int g = 0; g < 4; ++g) wf[b][hfl][g] = F::load(wrow[g], k0 + hfl * F::kStep, lane);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) hf[b][hfl][rg] = F::load(hrow[rg], k0 + hfl * F::kStep, lane);
      }
    };
    auto mma = [&](int b) {
#pragma unroll
      for (int hfl = 0; hfl < 2; ++hfl)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[rg][g] = F::mma(wf[b][hfl][g], hf[b][hfl][rg], acc[rg][g]);
    };
    load(0, 0);
    int i = 0;
    for (; i + 2 <= n; i += 2) {
      load(1, i + 1);
      mma(0);
      load(0, i + 2 < n ? i + 2 : n - 1);          // (the last one is a harmless re-request of the final lines)
      mma(1);
    }
    if (i < n) mma(0);
  }
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
    if (rg != wave) {
#pragma unroll
      for (int g = 0; g < 4; ++g) part[rg][wave - (wave > rg ? 1 : 0)][g][lane] = acc[rg][g];
    }
  __syncthreads();
  // wave w finishes row group w.  The four partial sums are added in K order whatever the finishing wave is: every batch
  // row sees the same summation order (copies of a segment in different row groups give the same bits)
  f32x4 mine[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f32x4 own = wave == 0 ? acc[0][g] : wave == 1 ? acc[1][g] : wave == 2 ? acc[2][g] : acc[3][g];
    f32x4 tot = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int slot = w - (w > wave ? 1 : 0);
      const f32x4 other = part[wave][slot < 3 ? slot : 2][g][lane];
      tot += (w == wave) ? own : other;
    }
    mine[g] = tot;
  }
  const int mrow = m0 + wave * 16 + (lane & 15);
  if (mrow >= B) return;
  cell_fwd_epilogue<T>(mine, mrow, u0 + (lane >> 4) * 4, xproj, xproj_ld, c_prev, gates_out, c_out, h_out, H);
}

template <typename T>
__global__ void __launch_bounds__(512)
lstm_cell_bwd_ks_kernel(CellBwdBatch batch, int B, int H) {
  CSN_CELL_BWD_BIND;
  typedef Frag<T> F;
  // 32 rows x 32 units per workgroup, K = 4H in 8 slices (one per wave): 192 workgroups at B 256 / H 768 -- 64 x 32 tiles
  // would read less (113 instead of 151 MB per launch) but fill only 96 CUs
  __shared__ f32x4 part[4][7][64];                        // [dst = rg * 2 + ut][src wave, without dst][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int K = 4 * H;
  f32x4 acc[2][2];
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int ut = 0; ut < 2; ++ut) acc[rg][ut] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (dg_next != nullptr) {
    const T* drow[2];
    bool ok[2];
#pragma unroll
    for (int rg = 0; rg < 2; ++rg) {
      const int m = m0 + rg * 16 + (lane & 15);
      ok[rg] = m < B;
      drow[rg] = dg_next + (int64_t)(ok[rg] ? m : 0) * K;
    }
    const T* wrow[2];
#pragma unroll
    for (int ut = 0; ut < 2; ++ut) wrow[ut] = w_hh_t + (int64_t)(u0 + ut * 16 + (lane & 15)) * K;   // W_hh^T[u][k]
    const int ks = K >> 3, k_beg = wave * ks, n = ks / (2 * F::kStep);
    typename F::type wf[3][2][2], df[3][2][2];               // three stages of one 128-byte line per row, as in the forward
    auto load = [&](int b, int i) {
#pragma unroll
      for (int hfl = 0; hfl < 2; ++hfl) {
        const int k0 = k_beg + (2 * i + hfl) * F::kStep;
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) wf[b][hfl][ut] = F::load(wrow[ut], k0, lane);
#pragma unroll
        for (int rg = 0; rg < 2; ++rg) df[b][hfl][rg] = F::load(drow[rg], k0, lane);      // (rows beyond B: row 0, never stored)
      }
    };
    auto mma = [&](int b) {
#pragma unroll
      for (int hfl = 0; hfl < 2; ++hfl)
#pragma unroll
        for (int rg = 0; rg < 2; ++rg)
#pragma unroll
          for (int ut = 0; ut < 2; ++ut) acc[rg][ut] = F::mma(wf[b][hfl][ut], df[b][hfl][rg], acc[rg][ut]);
    };
    auto clamp = [&](int i) { return i < n ? i : n - 1; };
    load(0, 0);
    load(1, clamp(1));
    int i = 0;
    for (; i + 3 <= n; i += 3) {
      load(2, clamp(i + 2));
      mma(0);
      load(0, clamp(i + 3));
      mma(1);
      load(1, clamp(i + 4));
      mma(2);
    }
    if (i < n) mma(0);
    if (i + 1 < n) mma(1);
  }
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int ut = 0; ut < 2; ++ut) {
      const int dst = rg * 2 + ut;
      if (dst != wave) part[dst][wave - (wave > dst ? 1 : 0)][lane] = acc[rg][ut];
    }
  __syncthreads();
  if (wave >= 4) return;
  // wave w < 4 finishes (row group w >> 1, unit tile w & 1); partial sums added in K order whatever the finishing wave is
  f32x4 own = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int rg = 0; rg < 2; ++rg)
#pragma unroll
    for (int ut = 0; ut < 2; ++ut)
      if (rg * 2 + ut == wave) own = acc[rg][ut];
  f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const int slot = w - (w > wave ? 1 : 0);
    const f32x4 other = part[wave][slot < 7 ? slot : 6][lane];
    sum += (w == wave) ? own : other;
  }
  const int mrow = m0 + (wave >> 1) * 16 + (lane & 15);
  if (mrow >= B) return;
  const int ub = u0 + (wave & 1) * 16 + (lane >> 4) * 4;
  float dh[4] = {sum[0], sum[1], sum[2], sum[3]};
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

// the K-split kernels need K slices (forward: H / 4; backward: 4H / 8 = H / 2) that are whole numbers of 128-byte lines
static bool cell_ks_ok(int H, int dtype) { return H % (8 * (dtype == CSN_BF16 ? 32 : 16)) == 0; }     // whole lines per K slice

int launch_cell_fwd_batch(const CellFwdBatch& b, int np, int B, int H, int dtype, hipStream_t st) {
  CSN_REQUIRE(np >= 1 && np <= 4, "launch_cell_fwd_batch: %d problems", np);
  if (cell_ks_ok(H, dtype)) {
    dim3 gridk((unsigned)(H / 16), (unsigned)((B + 63) / 64), (unsigned)np);
    if (dtype == CSN_BF16) lstm_cell_fwd_ks_kernel<bf16_t><<<gridk, 256, 0, st>>>(b, B, H);
    else lstm_cell_fwd_ks_kernel<float><<<gridk, 256, 0, st>>>(b, B, H);
  } else {
    dim3 grid((unsigned)(H / 16), (unsigned)((B + 63) / 64), (unsigned)np);
    if (dtype == CSN_BF16) lstm_cell_fwd_kernel<bf16_t><<<grid, 256, 0, st>>>(b, B, H);
    else lstm_cell_fwd_kernel<float><<<grid, 256, 0, st>>>(b, B, H);
  }
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}
int launch_cell_fwd(const void* h_prev, const void* w_hh, const float* xproj, int64_t xproj_ld, const float* c_prev,
                    void* gates_out, float* c_out, void* h_out, int B, int H, int dtype, hipStream_t st) {
  CellFwdBatch b{};
  b.p[0] = CellFwdOne{h_prev, w_hh, xproj, xproj_ld, c_prev, gates_out, c_out, h_out};
  return launch_cell_fwd_batch(b, 1, B, H, dtype, st);
}

int launch_cell_bwd_batch(const CellBwdBatch& b, int np, int B, int H, int dtype, hipStream_t st) {
  CSN_REQUIRE(np >= 1 && np <= 4, "launch_cell_bwd_batch: %d problems", np);
  if (cell_ks_ok(H, dtype)) {
    dim3 gridk((unsigned)(H / 32), (unsigned)((B + 31) / 32), (unsigned)np);
    if (dtype == CSN_BF16) lstm_cell_bwd_ks_kernel<bf16_t><<<gridk, 512, 0, st>>>(b, B, H);
    else lstm_cell_bwd_ks_kernel<float><<<gridk, 512, 0, st>>>(b, B, H);
  } else {
    dim3 grid((unsigned)(H / 16), (unsigned)((B + 63) / 64), (unsigned)np);
    if (dtype == CSN_BF16) lstm_cell_bwd_kernel<bf16_t><<<grid, 256, 0, st>>>(b, B, H);
    else lstm_cell_bwd_kernel<float><<<grid, 256, 0, st>>>(b, B, H);
  }
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}
int launch_cell_bwd(const void* dg_next, const void* w_hh_t, const float* dy, int64_t dy_ld, const void* gates,
                    const float* c, const float* c_prev, float* dc_carry, void* dg_out, int B, int H, int dtype,
                    hipStream_t st) {
  CellBwdBatch b{};
  b.p[0] = CellBwdOne{dg_next, w_hh_t, dy, dy_ld, gates, c, c_prev, dc_carry, dg_out};
  return launch_cell_bwd_batch(b, 1, B, H, dtype, st);
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
