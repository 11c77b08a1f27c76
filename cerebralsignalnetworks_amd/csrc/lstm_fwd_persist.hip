// K3 forward, weight-stationary form: ONE launch advances a layer through a chunk of timesteps.
// Replaces the per-step body of nn.LSTM reached at /root/reference/LSTMDistill.py:118,132.
//
// Why: in the per-step launches of lstm_cell_blk.hip, 60 % of the operand bytes a workgroup loads
// per step are its W_hh slice, re-read every step at the per-CU L2 rate (~55 GB/s per CU measured).
// Here a workgroup owns a (64 rows x 4*NQ units) tile for the whole chunk, keeps its W_hh slice in
// registers (each wave: its K quarter of all 4*NQ gate-row tiles = KS*NQ fragments, 144 VGPRs at
// H = 768) and its cell state c in registers, and per step only streams its 64 rows of h_{t-1}.
//
// The step-to-step hand-off of h between the workgroups of one M-tile (the gridDim.x workgroups
// that share 64 batch rows; different M-tiles never talk) is the placement-independent form of
// cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms", row 1:
//   producer: h slice stored WRITE-THROUGH (8-byte agent-scope atomic stores = global_store sc1),
//             every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE lane adds 1 to
//             the arrival counter of (step, M-tile) with an agent-scope atomic;
//   consumer: ONE lane polls that counter with relaxed agent-scope loads (+ s_sleep), workgroup
//             barrier, then every load of the handed-off bytes is an sc1 (L1-bypassing) buffer load.
// On top of that every step's h lives at its OWN address (h_blk_all[t], never reused inside a
// forward), so no L1 / L2 in the chip can hold an older copy of a line being handed off.
// Measured per step (tools/persist_bench.hip, 2 layers side by side, H = 768): wait 3.0 us,
// h loads + MFMA 2.6 us, LDS reduction 0.5 us, epilogue 1.9 us, drain + signal 0.6 us = 8.6 us.
// Splitting the 64 rows into two alternating 32-row halves to hide the wait was tried and is slower
// (12.8 us per step: the half-size loads are latency-bound and the hand-off latency, ~3 us, is as
// long as a half step), so the whole tile advances together.
// All gridDim.x * gridDim.y workgroups must be co-resident (1 per CU: 128 at B = 256, H = 768);
// every spin is bounded: after kSpinTimeoutTicks the workgroup raises *error_flag (sticky: later
// waits return immediately) so a scheduling accident ends in a reported error, never in a hang.
#include "csn_common.h"
#include "lstm_cell_common.h"
#include "lstm_cell_blk.h"

#ifdef CSN_PSTAMPS
// diagnostic build only (tools/persist_bench.hip): per-phase wall-clock sums of workgroup (0,0)
__device__ unsigned long long g_pstamps[8];
#define CSN_PSTAMP(i)                                                          \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    if (tid == 0 && blockIdx.x == 3 && blockIdx.y == 1) {                      \
      const unsigned long long now_ = wall_clock64();                          \
      atomicAdd(&g_pstamps[i], now_ - last_);                                  \
      last_ = now_;                                                            \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#else
#define CSN_PSTAMP(i)
#endif

namespace csn {

static constexpr unsigned long long kSpinTimeoutTicks = 20000000ull;   // 0.2 s of the 100 MHz wall clock

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ bf16x8 load_sc1_b128(__amdgpu_buffer_rsrc_t rsrc, int byte_off) {
  // aux = 16: sc1 (agent-coherent, bypasses this CU's L1; MI355X_MICROARCH.md visibility table)
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16);
  union { u32x4 u; bf16x8 b; } cvt;
  cvt.u = v;
  return cvt.b;
}

__device__ __forceinline__ void store_wt_b64(bf16_t* p, const float (&v)[4]) {
  union { bf16x4 b; unsigned long long u; } cvt;
  cvt.b = (bf16x4){(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), cvt.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NQ, int KS>
__global__ void __launch_bounds__(256) lstm_fwd_persist_kernel(PersistFwdArgs a) {
  constexpr int NT = 4 * NQ;
  constexpr int NPAIR = 64 * NQ;
  constexpr int NPASS = (NPAIR + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float4 red[];   // [4][NT][65]
  const int B = a.B, H = a.H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int u0 = blockIdx.x * 4 * NQ, m0 = blockIdx.y * 64, mt = blockIdx.y, MT = gridDim.y;
  const int kblocks = H >> 5;
  const int ks_beg = wave * KS;                        // KS = kblocks / 4 k-steps per wave
  const size_t slab = (size_t)a.Bpad * H;              // elements of one fragment-major h slab

  // ---- stationary operands: this wave's K quarter of the workgroup's W_hh rows --------------
  bf16x8 wreg[KS][NQ];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int j = 0; j < NQ; ++j)
      wreg[ks][j] = *reinterpret_cast<const bf16x8*>(a.w_blk + ((int64_t)((u0 >> 2) + j) * kblocks + ks_beg + ks) * 512 + lane * 8);

  // ---- cell state of the (row, unit-quad) pairs this thread owns ------------------------------
  float4 cst[NPASS];
  int prow[NPASS], puq[NPASS], prl[NPASS], pj[NPASS];
  bool pok[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int p = tid + ps * 256;
    prl[ps] = p / NQ;
    pj[ps] = p % NQ;
    prow[ps] = m0 + prl[ps];
    puq[ps] = u0 + 4 * pj[ps];
    pok[ps] = p < NPAIR && prow[ps] < B;
    cst[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pok[ps] && a.t0 > 0)
      cst[ps] = *reinterpret_cast<const float4*>(a.c_all + ((size_t)a.t0 * B + prow[ps]) * H + puq[ps]);
  }

  const __amdgpu_buffer_rsrc_t hsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)a.h_blk_all, 0, (int)((size_t)(a.T + 1) * slab * 2), 0x00020000);
  const unsigned n_producers = gridDim.x;
#ifdef CSN_PSTAMPS
  unsigned long long last_ = wall_clock64();
#endif

  for (int s = 0; s < a.nsteps; ++s) {
    const int t = a.t0 + s;
    // this step's input projection, requested before the wait so its HBM latency hides under it
    float4 xp[NPASS][4];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
      if (pok[ps]) {
        const float4* xr = reinterpret_cast<const float4*>(a.xproj + ((size_t)t * B + prow[ps]) * 4 * H + 4 * (size_t)puq[ps]);
#pragma unroll
        for (int q = 0; q < 4; ++q) xp[ps][q] = nt_load(xr + q);
      }

    f32x4 acc[4][NQ];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int j = 0; j < NQ; ++j) acc[rg][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (t > 0) {
      // wait until every workgroup of this M-tile has published its slice of h_{t-1} (slot t)
      if (tid == 0) {
        const unsigned* cnt = a.counters + (size_t)t * MT + mt;
        const unsigned long long t_begin = wall_clock64();
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_producers) {
          __builtin_amdgcn_s_sleep(1);
          if (__hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          if (wall_clock64() - t_begin > kSpinTimeoutTicks) {
            __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      __syncthreads();
      CSN_PSTAMP(0);   // wait for h_{t-1}
      const int base = (int)(((size_t)t * slab + ((size_t)(m0 >> 4) * kblocks + ks_beg) * 512 + lane * 8) * 2);
      bf16x8 hf[KS][4];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) hf[ks][rg] = load_sc1_b128(hsrc, base + (rg * kblocks + ks) * 1024);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
          for (int j = 0; j < NQ; ++j)
            acc[rg][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][j], hf[ks][rg], acc[rg][j], 0, 0, 0);
    }
    CSN_PSTAMP(1);     // h loads + MFMA

#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int j = 0; j < NQ; ++j)
        red[(wave * NT + rg * NQ + j) * 65 + lane] = make_float4(acc[rg][j][0], acc[rg][j][1], acc[rg][j][2], acc[rg][j][3]);
    __syncthreads();
    CSN_PSTAMP(2);     // LDS write + barrier

#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      if (!pok[ps]) continue;
      const int rl = prl[ps], j = pj[ps], row = prow[ps], uq = puq[ps];
      float gi[4], gf[4], gg[4], go[4], cn[4], hn[4];
      const float cpv[4] = {cst[ps].x, cst[ps].y, cst[ps].z, cst[ps].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = ((rl >> 4) * NQ + j) * 65 + (rl & 15) + 16 * q;
        float4 sum = red[idx];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) {
          const float4 v = red[w2 * NT * 65 + idx];
          sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        gi[q] = fast_sigmoid(sum.x + xp[ps][q].x);
        gf[q] = fast_sigmoid(sum.y + xp[ps][q].y);
        gg[q] = fast_tanh(sum.z + xp[ps][q].z);
        go[q] = fast_sigmoid(sum.w + xp[ps][q].w);
        cn[q] = gf[q] * cpv[q] + gi[q] * gg[q];
        hn[q] = go[q] * fast_tanh(cn[q]);
      }
      cst[ps] = make_float4(cn[0], cn[1], cn[2], cn[3]);
      // the hand-off payload first, write-through
      store_wt_b64(a.h_blk_all + (size_t)(t + 1) * slab + blk_offset(row, uq, H), hn);
      if (a.gates != nullptr) {
        bf16x8 lo = {(bf16_t)gi[0], (bf16_t)gf[0], (bf16_t)gg[0], (bf16_t)go[0], (bf16_t)gi[1], (bf16_t)gf[1], (bf16_t)gg[1], (bf16_t)go[1]};
        bf16x8 hi = {(bf16_t)gi[2], (bf16_t)gf[2], (bf16_t)gg[2], (bf16_t)go[2], (bf16_t)gi[3], (bf16_t)gf[3], (bf16_t)gg[3], (bf16_t)go[3]};
        bf16x8* gp = reinterpret_cast<bf16x8*>(a.gates + ((size_t)t * B + row) * 4 * H + 4 * (size_t)uq);
        __builtin_nontemporal_store(lo, gp);
        __builtin_nontemporal_store(hi, gp + 1);
      }
      __builtin_nontemporal_store((f32x4){cn[0], cn[1], cn[2], cn[3]},
                                  reinterpret_cast<f32x4*>(a.c_all + ((size_t)(t + 1) * B + row) * H + uq));
      __builtin_nontemporal_store((bf16x4){(bf16_t)hn[0], (bf16_t)hn[1], (bf16_t)hn[2], (bf16_t)hn[3]},
                                  reinterpret_cast<bf16x4*>(a.h_all + ((size_t)(t + 1) * B + row) * H + uq));
    }
    CSN_PSTAMP(3);     // epilogue (LDS reads, math, store issue)
    // publish: every storing wave drains, workgroup barrier (also frees `red`), one lane signals
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CSN_PSTAMP(4);     // drain + barrier
    if (tid == 0)
      __hip_atomic_fetch_add(a.counters + (size_t)(t + 1) * MT + mt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    CSN_PSTAMP(5);     // signal
  }
}

bool fwd_persist_supported(int B, int H, int dtype) {
  if (dtype != CSN_BF16 || H % 128 != 0 || getenv("CSN_NO_PERSIST") != nullptr) return false;
  const int nq = (H % 24 == 0) ? 6 : 8, ks = H / 128;
  const bool shape = (nq == 6 && (ks == 6 || ks == 3)) || (nq == 8 && (ks == 4 || ks == 2 || ks == 1));
  // all workgroups of a launch must be co-resident: one per CU, and two layers run side by side
  const int wgs = (H / (4 * nq)) * ((B + 63) / 64);
  return shape && wgs <= 128;
}

template <int NQ, int KS>
static int launch_persist_t(const PersistFwdArgs& a, hipStream_t st) {
  const size_t lds = (size_t)4 * 4 * NQ * 65 * sizeof(float4);
  static bool attr_done = false;
  if (!attr_done) {
    CSN_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_persist_kernel<NQ, KS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  dim3 grid((unsigned)(a.H / (4 * NQ)), (unsigned)((a.B + 63) / 64));
  lstm_fwd_persist_kernel<NQ, KS><<<grid, 256, lds, st>>>(a);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

int launch_fwd_persist(const PersistFwdArgs& a, hipStream_t st) {
  const int nq = (a.H % 24 == 0) ? 6 : 8, ks = a.H / 128;
  if (nq == 6 && ks == 6) return launch_persist_t<6, 6>(a, st);
  if (nq == 6 && ks == 3) return launch_persist_t<6, 3>(a, st);
  if (nq == 8 && ks == 4) return launch_persist_t<8, 4>(a, st);
  if (nq == 8 && ks == 2) return launch_persist_t<8, 2>(a, st);
  if (nq == 8 && ks == 1) return launch_persist_t<8, 1>(a, st);
  return fail(CSN_ERR_UNSUPPORTED, "launch_fwd_persist: no kernel for H=%d", a.H);
}

}  // namespace csn

#ifdef CSN_PSTAMPS
// diagnostic build only (make diag): read and clear the per-phase tick sums
extern "C" int csn_debug_read_pstamps(unsigned long long* out) {
  unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamps), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_pstamps), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif
