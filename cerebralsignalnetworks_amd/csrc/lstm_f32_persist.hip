// K3, exact-float32 path: weight-stationary recurrence.  Replaces the per-timestep launches of lstm_cell.hip's K-split
// kernels (one launch per wavefront diagonal: ~1 000 launches per pass at cfg2, each re-reading its 9.4 MB of W_hh from
// L2 and paying a launch boundary for 10 us of float32 MFMA work) behind the same entry points -- the float32 form of
// what nn.LSTM computes at /root/reference/LSTMDistill.py:118,132 and LSTMDistillRetreival.py:91,103.
//
// One launch = one layer, all T steps, a block of 64-row M-tiles.  A workgroup (4 waves) owns 64 batch rows x 16 hidden
// units for the whole sequence; wave w keeps the K quarter w of the workgroup's weight rows in registers
// (v_mfma_f32_16x16x4_f32 fragments: H / 4 registers per lane -- 192 at H = 768, 256 at H = 1024), multiplies it by the
// matching K quarter of the tile's h_{t-1} (forward) / dgates_{t+1} (backward) rows, the four partial tiles meet in LDS and
// wave w finishes row group w: the operand map, the k order, the order of the four partial sums and the gate math are
// those of lstm_cell_fwd_ks_kernel, so the forward is BIT-IDENTICAL to the per-step launches (CSN_NO_PERSIST).  The
// backward splits K = 4H four ways (one gate block per wave) where the per-step kernel splits it eight ways: same
// arithmetic, a different grouping of the float32 partial sums.
//
// Hand-off between the H / 16 workgroups of a tile, per step: the payload is a fragment-major copy of h_t / dgates_t (one
// slot per step: 1 KB blocks in the operand layout of the consumers' MFMAs, written next to the row-major arrays the GEMMs
// read), stored write-through (sc1) and read with sc1 loads (this CU's L1 bypassed); a producer drains its stores
// (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, one lane raises the producer's word of the step's flag line
// (relaxed, agent scope); a consumer wave polls exactly the words of the producers whose units / gate block it
// contracts.  ONLY the payload is stored in front of the drain: the saved tensors / row-major results (gates, c, h, dgates:
// streaming stores to cold HBM lines, acknowledged microseconds later) follow the flag, and the next step's inputs are
// requested there too -- with them in front of the drain a forward step took 19.9 instead of 16 us.  The placement-independent flag protocol of lstm_fwd_persist.hip, nothing else: at 10 us of MFMA issue per
// step the 2 - 3 us of hand-off are not where the time is.  Every spin is bounded (status word 0 on time-out).
// All workgroups of a launch must be co-resident: the grid is at most 256 (one workgroup per CU: > 256 registers per
// lane), larger batches walk their M-tiles in blocks (rows are independent).
#include "lstm_f32_persist.h"

#include "lstm_cell_common.h"

#ifdef CSN_PSTAMPS        // `make diag`: phase times of ONE workgroup's wave 0, summed over the steps of a launch (tools/insitu_stamps.py)
__device__ unsigned long long g_f32stamps[16];      // forward 0..6, backward 8..14: wait | loads + MFMA | partials + barrier | sums + gate math + store issue | drain | barrier + flag
#define CSN_F32STAMP(i)                                                        \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    if (tid == 0 && blockIdx.x == 5) {                                         \
      const unsigned long long now_ = wall_clock64();                          \
      atomicAdd(&g_f32stamps[i], now_ - last_);                                \
      last_ = now_;                                                            \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#else
#define CSN_F32STAMP(i)
#endif

namespace csn {

typedef __attribute__((ext_vector_type(4))) unsigned fu32x4;
static constexpr unsigned long long kF32SpinTimeoutTicks = 20000000ull;   // 0.2 s of the 100 MHz wall clock

#ifndef CSN_F32_ABL
#define CSN_F32_ABL 0      // timing-only ablations (tools/abl_build.sh): 1 no MFMAs, 2 no operand loads, 4 cheap gate math, 8 no polls, 16 no drain, 32 saved-tensor stores in front of the drain
#endif
__device__ __forceinline__ f32x4 f32p_load_sc1(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
#if CSN_F32_ABL & 2
  return (f32x4){1e-3f * (float)(byte_off & 7), 1e-3f, -1e-3f, 2e-3f};
#endif
  const fu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16);      // aux 16 = sc1
  return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ void f32p_store_wt(__amdgpu_buffer_rsrc_t rsrc, int byte_off, const f32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(fu32x4, v), rsrc, byte_off, 0, 16);   // sc1 = write-through
}
__device__ __forceinline__ f32x4 f32p_mfma(float a, float b, const f32x4& c) {
#if CSN_F32_ABL & 1
  return (f32x4){c[0] + a * b, c[1], c[2], c[3]};
#endif
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// saved tensors / row-major results: streaming (non-temporal) stores; bit 64 of CSN_F32_ABL: plain stores (timing comparison)
__device__ __forceinline__ void f32p_cold_store(f32x4* p, const f32x4& v) {
#if CSN_F32_ABL & 64
  *p = v;
#else
  nt_store(p, v);
#endif
}

// wait until every polled word is raised (lane i < n watches word i of `line`); bounded
__device__ __forceinline__ void f32p_wait_flags(const unsigned* line, int n, int lane, unsigned* error_flag) {
#if CSN_F32_ABL & 8
  return;
#endif
  const unsigned* fl = line + (lane < n ? lane : 0);
  const unsigned long long t_begin = wall_clock64();
  while (!__all(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    if (wall_clock64() - t_begin > kF32SpinTimeoutTicks) {
      __hip_atomic_store(error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward: H = 64 * KS
// ------------------------------------------------------------------------------------------
template <int KS>
__global__ void __launch_bounds__(256) lstm_fwd_f32_persist_kernel(F32PersistFwdArgs a) {
  constexpr int H = 64 * KS, G = 4 * H;
  constexpr int NSL = H / 16;              // producer slices of a tile
  constexpr int NPW = NSL / 4;             // ... whose units fall into one wave's K quarter
  constexpr int RING = KS < 4 ? KS : 4;    // 16-wide k-steps of h in flight per wave (3 and 6: no faster)
  // partial tiles on their way to the wave that finishes them: [dst row group][src wave, without dst][gate][lane]
  __shared__ f32x4 part[4][3][4][64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = (int)(blockIdx.x % (unsigned)a.MT), slice = (int)(blockIdx.x / (unsigned)a.MT);
  const int B = a.B, T = a.T;
  const int u0 = slice * 16, m0 = (a.mt0 + mt) * 64;
  const int k_beg = wave * (H / 4);

  // stationary operand: rows (gate g, unit u0 + (lane & 15)), this wave's K quarter
  f32x4 wf[KS][4];
#pragma unroll
  for (int i = 0; i < KS; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      wf[i][g] = *reinterpret_cast<const f32x4*>(a.w_hh + ((int64_t)g * H + u0 + (lane & 15)) * H + k_beg + i * 16 + 4 * (lane >> 4));

  // the moving operand comes from the fragment-major copy of h (h_blk): block (16-row group, 16-unit k-step) = 1 KB, lane l's
  // 16 bytes at l * 16 -- what a producer wave holds in its lanes after the gate math IS one such block, so both the
  // store and the load of a fragment are one contiguous 1 KB access (the row-major slot gives 16 x 64-byte pieces per
  // instruction, each a request of its own to L2 with the L1 bypassed: the backward's 786 KB per step and workgroup
  // took 34 us that way, three times its MFMA time)
  const int hoff = (((m0 >> 4) * (H / 16)) + wave * KS) * 1024 + lane * 16;      // row group rg, k-step i: + (rg * (H / 16) + i) * 1024
  // the (row, 4 units) this lane finishes, every step
  const int mrow = m0 + wave * 16 + (lane & 15);
  const bool row_ok = mrow < B;
  const int mr = row_ok ? mrow : 0;
  const int ub = u0 + (lane >> 4) * 4;
  float cp[4];
  {
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(a.c_all + (int64_t)mr * H + ub);
    cp[0] = c0[0]; cp[1] = c0[1]; cp[2] = c0[2]; cp[3] = c0[3];
  }
  const size_t flag_step = (size_t)a.MT_total * kF32FlagLine;
  unsigned* const flags = a.flags + (size_t)(a.mt0 + mt) * kF32FlagLine;
  const int bslot_bytes = a.MT_total * 64 * H * 4;       // one fragment-major slot: all row groups, padded

#ifdef CSN_PSTAMPS
  unsigned long long last_ = wall_clock64();
#endif
  for (int t = 0; t < T; ++t) {
    f32x4 acc[4][4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[rg][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 xp[4];
    auto request_x = [&]() {
#pragma unroll
      for (int g = 0; g < 4; ++g)
        xp[g] = nt_load(reinterpret_cast<const f32x4*>(a.xproj + ((int64_t)t * B + mr) * G + (int64_t)g * H + ub));
    };
    if (t > 0) {
      // h_{t-1} = slot t: this wave contracts the units of slices [wave * NPW, (wave + 1) * NPW)
      f32p_wait_flags(flags + (size_t)t * flag_step + wave * NPW, NPW, lane, a.error_flag);
      CSN_F32STAMP(0);
      const __amdgpu_buffer_rsrc_t hsrc =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.h_blk + (size_t)t * (bslot_bytes / 4)), 0, bslot_bytes, 0x00020000);
      f32x4 hf[RING][4];
      auto issue = [&](int i) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) hf[i % RING][rg] = f32p_load_sc1(hsrc, hoff + (rg * (H / 16) + i) * 1024);
      };
#pragma unroll
      for (int i = 0; i < RING; ++i) {
        issue(i);
        __builtin_amdgcn_sched_barrier(0);
      }
      // this step's input projection: requested behind the first operand groups, used 10 us later in the epilogue
      request_x();
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        // (jj outermost: 16 independent accumulators between two MFMAs into the same one; every accumulator still
        // receives its products in ascending k, the order of lstm_cell_fwd_ks_kernel)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[rg][g] = f32p_mfma(wf[i][g][jj], hf[i % RING][rg][jj], acc[rg][g]);
        __builtin_amdgcn_sched_barrier(0);
        if (i + RING < KS) {
          issue(i + RING);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else {
      request_x();
    }
    CSN_F32STAMP(1);

#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      if (rg != wave) {
#pragma unroll
        for (int g = 0; g < 4; ++g) part[rg][wave - (wave > rg ? 1 : 0)][g][lane] = acc[rg][g];
      }
    __syncthreads();
    CSN_F32STAMP(2);
    // wave w finishes row group w; the four partial sums are added in K order whatever the finishing wave is
    float pre[4][4];
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
#pragma unroll
      for (int r = 0; r < 4; ++r) pre[g][r] = tot[r] + xp[g][r];
    }
    float gi[4], gf[4], gg[4], go[4], cn[4], hn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#if CSN_F32_ABL & 4
      gi[r] = fminf(fmaxf(pre[0][r] * 0.25f + 0.5f, 0.f), 1.f);
      gf[r] = fminf(fmaxf(pre[1][r] * 0.25f + 0.5f, 0.f), 1.f);
      gg[r] = fminf(fmaxf(pre[2][r], -1.f), 1.f);
      go[r] = fminf(fmaxf(pre[3][r] * 0.25f + 0.5f, 0.f), 1.f);
      cn[r] = gf[r] * cp[r] + gi[r] * gg[r];
      hn[r] = go[r] * fminf(fmaxf(cn[r], -1.f), 1.f);
#else
      gi[r] = sigmoid_f32(pre[0][r]);
      gf[r] = sigmoid_f32(pre[1][r]);
      gg[r] = tanh_f32(pre[2][r]);
      go[r] = sigmoid_f32(pre[3][r]);
      cn[r] = gf[r] * cp[r] + gi[r] * gg[r];
      hn[r] = go[r] * tanh_f32(cn[r]);
#endif
      cp[r] = cn[r];
    }
    auto cold_stores = [&]() {
      if (row_ok) {
        if (a.gates != nullptr) {
          f32x4* gp = reinterpret_cast<f32x4*>(a.gates + ((int64_t)t * B + mrow) * G + ub);
          f32p_cold_store(gp, (f32x4){gi[0], gi[1], gi[2], gi[3]});
          f32p_cold_store(gp + H / 4, (f32x4){gf[0], gf[1], gf[2], gf[3]});
          f32p_cold_store(gp + 2 * (H / 4), (f32x4){gg[0], gg[1], gg[2], gg[3]});
          f32p_cold_store(gp + 3 * (H / 4), (f32x4){go[0], go[1], go[2], go[3]});
        }
        f32p_cold_store(reinterpret_cast<f32x4*>(a.c_all + ((int64_t)(t + 1) * B + mrow) * H + ub), (f32x4){cn[0], cn[1], cn[2], cn[3]});
        f32p_cold_store(reinterpret_cast<f32x4*>(a.h_all + ((int64_t)(t + 1) * B + mrow) * H + ub), (f32x4){hn[0], hn[1], hn[2], hn[3]});
      }
    };
#if CSN_F32_ABL & 32
    cold_stores();      // (timing comparison: round 4's first form, every store in front of the drain)
#endif
    {
      // the hand-off payload: this wave's finished 16 x 16 tile as ONE fragment block, written through (rows beyond B:
      // whatever the clamped inputs gave -- finite, read only into their own, never stored, columns)
      const __amdgpu_buffer_rsrc_t hdst =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.h_blk + (size_t)(t + 1) * (bslot_bytes / 4)), 0, bslot_bytes, 0x00020000);
      f32p_store_wt(hdst, ((((m0 >> 4) + wave) * (H / 16)) + slice) * 1024 + lane * 16, (f32x4){hn[0], hn[1], hn[2], hn[3]});
    }
    CSN_F32STAMP(3);
    // publish: every storing wave drains, workgroup barrier (also frees `part`), one lane signals
#if !(CSN_F32_ABL & 16)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    CSN_F32STAMP(4);
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + (size_t)(t + 1) * flag_step + slice, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#if !(CSN_F32_ABL & 32)
    cold_stores();
#endif
    CSN_F32STAMP(5);
  }
}

// ------------------------------------------------------------------------------------------
// backward: dh_t = dy_t + dgates_{t+1} W_hh ; gate derivatives ; dgates_t.  K = 4H, wave w contracts gate block w.
// ------------------------------------------------------------------------------------------
template <int KS>
__global__ void __launch_bounds__(256) lstm_bwd_f32_persist_kernel(F32PersistBwdArgs a) {
  constexpr int H = 64 * KS, G = 4 * H;
  constexpr int NSL = H / 16;
  constexpr int KSB = H / 16;              // 16-wide k-steps of one gate block
  constexpr int RING = KS >= 16 ? 6 : (KSB < 4 ? KSB : 4);      // k-steps of dgates in flight per wave (measured at H = 768: 2 - 4 equal, 6 / 8 / 12 / 16 slower by 1 / 2 / 5 / 27 %: registers, not latency)
  __shared__ f32x4 part[4][3][64];         // [dst row group][src wave, without dst][lane]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mt = (int)(blockIdx.x % (unsigned)a.MT), slice = (int)(blockIdx.x / (unsigned)a.MT);
  const int B = a.B, T = a.T;
  const int u0 = slice * 16, m0 = (a.mt0 + mt) * 64;
  const int k_beg = wave * H;

  // stationary operand: W_hh^T rows u0 + (lane & 15), this wave's gate block
  f32x4 wt[KSB];
#pragma unroll
  for (int i = 0; i < KSB; ++i)
    wt[i] = *reinterpret_cast<const f32x4*>(a.w_hh_t + (int64_t)(u0 + (lane & 15)) * G + k_beg + i * 16 + 4 * (lane >> 4));

  // fragment-major copy of dgates (see the forward): block (16-row group, k-step = gate * H / 16 + unit tile)
  const int doff = (((m0 >> 4) * (G / 16)) + wave * KSB) * 1024 + lane * 16;
  const int mrow = m0 + wave * 16 + (lane & 15);
  const bool row_ok = mrow < B;
  const int mr = row_ok ? mrow : 0;
  const int ub = u0 + (lane >> 4) * 4;
  float dcn[4] = {0.f, 0.f, 0.f, 0.f};      // dL/dc carried from step t + 1
  // bias gradient = column sums of dgates: this lane's (row, 4 units x 4 gates) summed over the steps as they are produced
  // (a pass of its own over the 1.57 GB of dgates per layer took 0.26 ms at cfg2)
  float bsum[4][4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) bsum[g][r] = 0.f;
  const size_t flag_step = (size_t)a.MT_total * kF32FlagLine;
  unsigned* const flags = a.flags + (size_t)(a.mt0 + mt) * kF32FlagLine;
  const int bslot_bytes = a.MT_total * 64 * G * 4;

  // saved tensors of a step + its incoming gradient: always the same unconditional loads, requested one step EARLY (right
  // behind the publish of the step before): they come from HBM, and queued behind the first operand groups of their own
  // step -- vector-memory operations return in order -- they held the later groups back (the ring's 8 groups cover 1.7 us)
  f32x4 sg[4], scc, scp, sdy;
  auto request_saved = [&](int tt) {
    const f32x4* gp = reinterpret_cast<const f32x4*>(a.gates + ((int64_t)tt * B + mr) * G + ub);
#pragma unroll
    for (int g = 0; g < 4; ++g) sg[g] = nt_load(gp + g * (H / 4));
    scc = nt_load(reinterpret_cast<const f32x4*>(a.c_all + ((int64_t)(tt + 1) * B + mr) * H + ub));
    scp = nt_load(reinterpret_cast<const f32x4*>(a.c_all + ((int64_t)tt * B + mr) * H + ub));
    const float* dsrc = a.dy != nullptr ? a.dy + (int64_t)tt * B * H : ((a.dy_last != nullptr && tt == T - 1) ? a.dy_last : a.zeros);
    sdy = nt_load(reinterpret_cast<const f32x4*>(dsrc + (int64_t)mr * H + ub));
  };
  request_saved(T - 1);
#ifdef CSN_PSTAMPS
  unsigned long long last_ = wall_clock64();
#endif
  for (int t = T - 1; t >= 0; --t) {
    f32x4 acc[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) acc[rg] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (t < T - 1) {
      // dgates_{t+1}: every slice of the tile produced a part of this wave's gate block
      f32p_wait_flags(flags + (size_t)(t + 1) * flag_step, NSL, lane, a.error_flag);
      CSN_F32STAMP(8);
      const __amdgpu_buffer_rsrc_t dsrc_r =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.dg_blk + (size_t)(t + 1) * (bslot_bytes / 4)), 0, bslot_bytes, 0x00020000);
      f32x4 df[RING][4];
      auto issue = [&](int i) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) df[i % RING][rg] = f32p_load_sc1(dsrc_r, doff + (rg * (G / 16) + i) * 1024);
      };
#pragma unroll
      for (int i = 0; i < RING; ++i) {
        issue(i);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < KSB; ++i) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) acc[rg] = f32p_mfma(wt[i][jj], df[i % RING][rg][jj], acc[rg]);
        __builtin_amdgcn_sched_barrier(0);
        if (i + RING < KSB) {
          issue(i + RING);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    CSN_F32STAMP(9);

#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      if (rg != wave) part[rg][wave - (wave > rg ? 1 : 0)][lane] = acc[rg];
    __syncthreads();
    CSN_F32STAMP(10);
    const f32x4 own = wave == 0 ? acc[0] : wave == 1 ? acc[1] : wave == 2 ? acc[2] : acc[3];
    f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int slot = w - (w > wave ? 1 : 0);
      const f32x4 other = part[wave][slot < 3 ? slot : 2][lane];
      sum += (w == wave) ? own : other;
    }
    float dai[4], daf[4], dag[4], dao[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dh = sum[r] + sdy[r];
      const float gi = sg[0][r], gf = sg[1][r], gg = sg[2][r], go = sg[3][r];
#if CSN_F32_ABL & 4
      const float tc = fminf(fmaxf(scc[r], -1.f), 1.f);
#else
      const float tc = tanh_f32(scc[r]);
#endif
      const float d_o = dh * tc;
      const float dc = dh * go * (1.0f - tc * tc) + dcn[r];
      dai[r] = dc * gg * gi * (1.0f - gi);
      daf[r] = dc * scp[r] * gf * (1.0f - gf);
      dag[r] = dc * gi * (1.0f - gg * gg);
      dao[r] = d_o * go * (1.0f - go);
      dcn[r] = dc * gf;
      if (row_ok) {
        bsum[0][r] += dai[r];
        bsum[1][r] += daf[r];
        bsum[2][r] += dag[r];
        bsum[3][r] += dao[r];
      }
    }
    {
      // the hand-off payload first: four fragment blocks (one per gate), written through
      const __amdgpu_buffer_rsrc_t ddst =
          __builtin_amdgcn_make_buffer_rsrc((void*)(a.dg_blk + (size_t)t * (bslot_bytes / 4)), 0, bslot_bytes, 0x00020000);
      const int o = ((((m0 >> 4) + wave) * (G / 16)) + slice) * 1024 + lane * 16;
      f32p_store_wt(ddst, o, (f32x4){dai[0], dai[1], dai[2], dai[3]});
      f32p_store_wt(ddst, o + KSB * 1024, (f32x4){daf[0], daf[1], daf[2], daf[3]});
      f32p_store_wt(ddst, o + 2 * KSB * 1024, (f32x4){dag[0], dag[1], dag[2], dag[3]});
      f32p_store_wt(ddst, o + 3 * KSB * 1024, (f32x4){dao[0], dao[1], dao[2], dao[3]});
    }
    auto cold_stores = [&]() {
    if (row_ok) {      // the row-major result the GEMMs behind the recurrence read
      f32x4* op = reinterpret_cast<f32x4*>(a.dgates + ((int64_t)t * B + mrow) * G + ub);
      f32p_cold_store(op, (f32x4){dai[0], dai[1], dai[2], dai[3]});
      f32p_cold_store(op + H / 4, (f32x4){daf[0], daf[1], daf[2], daf[3]});
      f32p_cold_store(op + 2 * (H / 4), (f32x4){dag[0], dag[1], dag[2], dag[3]});
      f32p_cold_store(op + 3 * (H / 4), (f32x4){dao[0], dao[1], dao[2], dao[3]});
    }
    };
#if CSN_F32_ABL & 32
    cold_stores();      // (timing comparison: round 4's first form, every store in front of the drain)
#endif
    CSN_F32STAMP(11);
#if !(CSN_F32_ABL & 16)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    CSN_F32STAMP(12);
    __syncthreads();
    if (tid == 0) __hip_atomic_store(flags + (size_t)t * flag_step + slice, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t > 0) request_saved(t - 1);      // (in front of the cold stores: the other order is 0.15 ms per launch slower)
#if !(CSN_F32_ABL & 32)
    cold_stores();
#endif
    CSN_F32STAMP(13);
  }
  // the 16 rows of this wave's row group sit in lanes (l & 15) of each 16-lane group: xor-tree, then lane (l & 15) == 0 of each
  // group writes 4 units x 4 gates of the row group's partial sum; the row groups are added in fixed order by the host side
  if (a.bias_part != nullptr) {
    float* const dst = a.bias_part + (size_t)((a.mt0 + mt) * 4 + wave) * G;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float x = bsum[g][r];
        x += __shfl_xor(x, 1);
        x += __shfl_xor(x, 2);
        x += __shfl_xor(x, 4);
        x += __shfl_xor(x, 8);
        v[r] = x;
      }
      if ((lane & 15) == 0) *reinterpret_cast<f32x4*>(dst + g * H + ub) = (f32x4){v[0], v[1], v[2], v[3]};
    }
  }
}

// ------------------------------------------------------------------------------------------
bool f32_persist_supported(int B, int H) {
  if (!(H == 128 || H == 256 || H == 384 || H == 512 || H == 768 || H == 1024)) return false;
  return B >= 1 && ((int64_t)B + 63) / 64 * 64 * 4 * H * 4 < ((int64_t)1 << 31);      // one fragment-major dgates slot within a buffer resource's 32-bit offsets
}
int f32_persist_tiles_per_launch(int H) { return 256 / (H / 16); }

template <int KS>
static int launch_fwd_t(const F32PersistFwdArgs& a, hipStream_t st) {
  lstm_fwd_f32_persist_kernel<KS><<<dim3((unsigned)(a.MT * (64 * KS / 16))), dim3(256), 0, st>>>(a);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}
template <int KS>
static int launch_bwd_t(const F32PersistBwdArgs& a, hipStream_t st) {
  lstm_bwd_f32_persist_kernel<KS><<<dim3((unsigned)(a.MT * (64 * KS / 16))), dim3(256), 0, st>>>(a);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

int launch_fwd_f32_persist(const F32PersistFwdArgs& a, int H, hipStream_t st) {
  CSN_REQUIRE(f32_persist_supported(a.B, H), "launch_fwd_f32_persist: B=%d H=%d not covered", a.B, H);
  CSN_REQUIRE(a.MT >= 1 && a.MT <= f32_persist_tiles_per_launch(H) && a.mt0 >= 0 && a.mt0 + a.MT <= a.MT_total &&
                  a.MT_total == (a.B + 63) / 64 && a.T >= 1 && a.h_blk != nullptr,
              "launch_fwd_f32_persist: bad tile block (mt0 %d, MT %d of %d)", a.mt0, a.MT, a.MT_total);
  switch (H) {
    case 128: return launch_fwd_t<2>(a, st);
    case 256: return launch_fwd_t<4>(a, st);
    case 384: return launch_fwd_t<6>(a, st);
    case 512: return launch_fwd_t<8>(a, st);
    case 768: return launch_fwd_t<12>(a, st);
    default: return launch_fwd_t<16>(a, st);
  }
}
int launch_bwd_f32_persist(const F32PersistBwdArgs& a, int H, hipStream_t st) {
  CSN_REQUIRE(f32_persist_supported(a.B, H), "launch_bwd_f32_persist: B=%d H=%d not covered", a.B, H);
  CSN_REQUIRE(a.MT >= 1 && a.MT <= f32_persist_tiles_per_launch(H) && a.mt0 >= 0 && a.mt0 + a.MT <= a.MT_total &&
                  a.MT_total == (a.B + 63) / 64 && a.T >= 1 && a.zeros != nullptr && a.dg_blk != nullptr,
              "launch_bwd_f32_persist: bad tile block (mt0 %d, MT %d of %d)", a.mt0, a.MT, a.MT_total);
  switch (H) {
    case 128: return launch_bwd_t<2>(a, st);
    case 256: return launch_bwd_t<4>(a, st);
    case 384: return launch_bwd_t<6>(a, st);
    case 512: return launch_bwd_t<8>(a, st);
    case 768: return launch_bwd_t<12>(a, st);
    default: return launch_bwd_t<16>(a, st);
  }
}

}  // namespace csn

#ifdef CSN_PSTAMPS
extern "C" int csn_debug_read_f32stamps(unsigned long long* out) {      // read and reset (diag library only)
  unsigned long long z[16] = {};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_f32stamps), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_f32stamps), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif
