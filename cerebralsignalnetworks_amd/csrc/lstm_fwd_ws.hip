// K3 forward, weight-stationary, WAVE-SPECIALISED form for H = 768 (cfg2): the per-step body of nn.LSTM reached at
// /root/reference/LSTMDistill.py:118,132 -- same contract, workspace and hand-off protocol as lstm_fwd_persist.hip /
// lstm_fwd_ns.hip, which stay as cross-checks.
//
// Why another body.  In every other weight-stationary forward a step is ONE serial chain per workgroup,
//     poll -> h tile in (0.6 us to the first bytes) -> MFMA (2.0 us) -> gate math -> stores -> drain -> flag -> 0.5 us
// of which only the 2.0 us are matrix-core time (6.4 us per step measured), and all four waves walk every phase together,
// so while the gate math runs the MFMA pipes idle and vice versa.  Pipelining two row halves through the SAME waves
// (kp_recurrence) hid the hand-off latency but not that: 6.0 us of work per step remained.  Here the phases get waves
// of their own and the 64-row tile is cut into FOUR independent 16-row chains (a chain = one MFMA N-tile of batch rows):
//
//   waves 0-3  (one per SIMD)  MFMA waves.  Wave w keeps K-quarter w of the workgroup's 96 gate rows (24 units x
//              i,f,g,o) in 144 accumulator registers.  Per chain and step: poll the chain's flag line (scalar loads,
//              a queue of their own), fetch its 16 x 192 slice of h_{t-1} STRAIGHT INTO REGISTERS as six 1 KB operand
//              fragments (no LDS staging: a K-quarter of a 16-row chain is private to the wave), 36 MFMAs, the six
//              partial tiles to LDS, bump the chain's LDS counter.  The fetch of the next chain is issued before the
//              MFMAs of the current one.
//   waves 4-7  gate waves, one per chain.  Off the critical path: the input part of the pre-activations -- layer 0:
//              bias + x_t W_ih^T by 24 MFMAs of its own against 96 registers of W_ih; layers above: the projection
//              read from HBM.  On it: wait for the chain's counter, add the four K-quarter partials, gate math in the
//              accumulator layout ((i, f, g, o) of one cell in one lane; c stays in registers), a wave-private LDS
//              transpose, stores with consecutive lanes on consecutive bytes (the h hand-off block first), a COUNTED
//              vmcnt wait that covers only that first store, the chain's flag.  Then the next step's input request.
//
// No workgroup barrier anywhere in the loop: the MFMA waves hear from the gate waves only through the global flags
// (their own workgroup's among the 32), the gate waves from the MFMA waves through monotonic LDS counters; the partial
// buffer of a chain cannot be overwritten early because its next MFMAs wait for the flag its own gate wave sets after
// reading it.  Chains of one workgroup run out of phase, so the MFMA pipes see work while every other latency
// (flag propagation, fetch, gate math, drain) of the other chains is in flight.
//
// Registers: 8 waves per CU -> 256 per wave, AGPRs + VGPRs.  MFMA waves: 144 + (24 acc + 2 x 24 operand + addressing);
// gate waves: 96 + (24 pre-activations + 16 input + 6 c + 16 of partials at a time + math).
#include "csn_common.h"
#include "lstm_cell_common.h"
#include "lstm_cell_blk.h"
#include "lstm_ns_util.h"

#ifdef CSN_PSTAMPS
#ifndef CSN_STAMP_BLOCK
#define CSN_STAMP_BLOCK 11
#endif
__device__ unsigned long long g_wstamps[16];
// one MFMA wave (wave 0: slots 0..) and one gate wave (wave 4: slots 8..) of one workgroup
#define CSN_WSTAMP(i)                                                          \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    if (stamping_ && lane == 0) {                                              \
      const unsigned long long now_ = wall_clock64();                          \
      atomicAdd(&g_wstamps[i], now_ - last_);                                  \
      last_ = now_;                                                            \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#define CSN_WSTAMP_INIT(cond)                                                  \
  const bool stamping_ = (cond) && blockIdx.x == CSN_STAMP_BLOCK;              \
  unsigned long long last_ = wall_clock64();                                   \
  (void)stamping_; (void)last_
#else
#define CSN_WSTAMP(i)
#define CSN_WSTAMP_INIT(cond)
#endif

namespace csn {

static constexpr int kWsKB = 24;                                   // k-blocks of H = 768
static constexpr int kWsKQ = 6;                                    // k-blocks of one K-quarter
static constexpr int kWsAF = 28;                                   // weight fragments (of 36) kept in AGPRs, the rest in VGPRs
static constexpr int kWsPartBytes = 4 * 4 * 6144;                  // [chain][K-quarter] six 1 KB partial tiles
static constexpr int kWsStageChain = 16 * (208 + 112 + 80);        // gates (bf16 x 4) + c (f32) + h (bf16) of 16 rows x 24 units, padded rows
static constexpr int kWsLdsBytes = kWsPartBytes + 4 * kWsStageChain + 64;

typedef __attribute__((address_space(3))) unsigned lds_u32;

__device__ __forceinline__ bool ws_error_seen(const PersistFwdArgs& a) {
  return __hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// ---------------------------------------------------------------------------------------------------------------
// MFMA wave kq: K-quarter kq of all 6 gate-row tiles of the slice, every chain of the tile in turn
__device__ __forceinline__ void ws_mfma_wave(const PersistFwdArgs& a, const PersistFwdSlot& S, char* smem, int slice,
                                             int mt, int kq) {
  constexpr int KB = kWsKB, KQ = kWsKQ;
  const int lane = threadIdx.x & 63;
  const int H = a.H, MT = a.MT;
  const size_t slab = (size_t)a.Bpad * H;
  const int t_first = S.t0, nsteps = S.nsteps;
  const int rot = a.rotate ? __builtin_amdgcn_readfirstlane((slice * KQ) / (H / 24)) : 0;

  // stationary operand: register p holds k-block kq KQ + (p + rot) % KQ of the 6 tiles
  // (the compiler splits the 256 registers of a wave 128 : 128 between VGPRs and AGPRs: 32 of the 36 fragments live
  // in AGPRs, 4 in VGPRs -- an MFMA takes its A operand from either)
  bf16x8 wa[kWsAF], wv[36 - kWsAF];
#pragma unroll
  for (int p = 0; p < KQ; ++p) {
    int kb = p + rot;
    kb = (kb >= KQ ? kb - KQ : kb) + kq * KQ;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(S.w_blk + ((int64_t)(6 * slice + j) * KB + kb) * 512 + lane * 8);
      if (p * 6 + j < kWsAF) wa[p * 6 + j] = w;
      else wv[p * 6 + j - kWsAF] = w;
    }
  }
  const __amdgpu_buffer_rsrc_t hrsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.h_blk_all), 0, __builtin_amdgcn_readfirstlane((int)((size_t)(a.T + 1) * slab * 2)), 0x00020000);
  lds_u32* const cnt = (lds_u32*)(smem + kWsPartBytes + 4 * kWsStageChain);

  // Is h_{t-1} of chain q complete?  The chain's GATE wave watches the global flag line (it has the time) and posts the
  // step into ready[q] in LDS: a check costs this wave an LDS read, not an L2 round trip.
  lds_u32* const ready = cnt + 4;
  auto is_ready = [&](int t, int q) {
    return __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ready + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= t;
  };
  auto wait_ready = [&](int t, int q) {
    const unsigned long long t_begin = wall_clock64();
    while (!is_ready(t, q)) {
      __builtin_amdgcn_s_sleep(1);
      if (wall_clock64() - t_begin > 2 * kNsSpinTimeoutTicks) {       // (the gate waves bound their own polls; this only ends a lost wave)
        __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  };
  // One operand buffer: the fragment of k-block p is re-requested for the NEXT item right behind the 6 MFMAs that
  // read it for the current one (the 128 VGPRs of a wave do not hold two buffers beside the accumulators and the
  // weight fragments that do not fit the AGPRs) -- so the next item's flag is polled before the current MFMAs start.
  bf16x8 hf[KQ];
  auto hoff_of = [&](int t, int q) {
    return __builtin_amdgcn_readfirstlane((int)(((size_t)t * slab + ((size_t)(mt * 4 + q) * KB + kq * KQ) * 512) * 2));
  };
  auto load_block = [&](int p, int sbase) {
    int kb = p + rot;
    kb = kb >= KQ ? kb - KQ : kb;
    hf[p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(hrsrc, lane * 16, sbase + kb * 1024, 16));   // sc1: hand-off data
  };
  auto mfma_block = [&](f32x4 (&acc)[6], int p) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      if (p * 6 + j < kWsAF) ns_mfma<false>(acc[j], wa[p * 6 + j], hf[p]);
      else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(wv[p * 6 + j - kWsAF]), "v"(hf[p]));
    }
  };

  const int N = 4 * nsteps;
  CSN_WSTAMP_INIT(kq == 0);
  if (t_first > 0) {
    wait_ready(t_first, 0);
    const int sb = hoff_of(t_first, 0);
#pragma unroll
    for (int p = 0; p < KQ; ++p) load_block(p, sb);
  }
  for (int i = 0; i < N; ++i) {
    const int t = t_first + (i >> 2), q = i & 3;
    const int tn = t_first + ((i + 1) >> 2), qn = (i + 1) & 3;
    const bool next_valid = i + 1 < N && tn > 0;
    const int sbn = hoff_of(tn, qn);
    // the next item's operand is requested between the MFMAs of this one if its chain is ready NOW; otherwise after
    // them, in one go, as soon as it is
    bool requested = false;
    if (t > 0) {
      f32x4 acc[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      ns_wait_vmcnt<0>();
      CSN_WSTAMP(1);   // operand fragments of the current item
      __builtin_amdgcn_sched_barrier(0);
      ns_mfma_fence();
      if (next_valid && is_ready(tn, qn)) {
        requested = true;
#pragma unroll
        for (int p = 0; p < KQ; ++p) {
          mfma_block(acc, p);
          __builtin_amdgcn_sched_barrier(0);
          load_block(p, sbn);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int p = 0; p < KQ; ++p) mfma_block(acc, p);
      }
      ns_mfma_fence();
      CSN_WSTAMP(2);   // MFMAs + requests
      __builtin_amdgcn_sched_barrier(0);
      f32x4* const part = reinterpret_cast<f32x4*>(smem + (size_t)(q * 4 + kq) * 6144);
#pragma unroll
      for (int j = 0; j < 6; ++j) part[j * 64 + lane] = acc[j];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_sched_barrier(0);
    CSN_WSTAMP(3);     // partials to LDS + counter
    if (next_valid && !requested) {
      wait_ready(tn, qn);
      CSN_WSTAMP(0);   // waiting for the next chain
#pragma unroll
      for (int p = 0; p < KQ; ++p) load_block(p, sbn);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// gate wave of chain q: rows 16 q .. 16 q + 15 of the tile, all 24 units of the slice
template <bool FUSED>
__device__ __forceinline__ void ws_gate_wave(const PersistFwdArgs& a, const PersistFwdSlot& S, char* smem, int slice,
                                             int mt, bool local, int q) {
  constexpr int P = FUSED ? 4 : 6;                     // 16-byte registers of one input request
  constexpr int xkb = 4;                               // fused form: I = 128 (checked by the launcher)
  const int lane = threadIdx.x & 63;
  const int B = a.B, H = a.H, MT = a.MT;
  const int u0 = slice * 24, r0 = mt * 64 + 16 * q;    // first unit / first batch row of this wave's cells
  const size_t slab = (size_t)a.Bpad * H;
  bf16_t* const gates = S.gates;
  float* const c_all = S.c_all;
  bf16_t* const h_all = S.h_all;
  unsigned* const flags = S.flags + ((size_t)mt * 4 + q) * kPersistFlagLine;
  const size_t flag_step = (size_t)MT * 4 * kPersistFlagLine;
  const int t_first = S.t0, nsteps = S.nsteps;
  const bool full = r0 + 16 <= B;                      // every row of the chain is a real batch row
  char* const sg = smem + kWsPartBytes + (size_t)q * kWsStageChain;   // gates [16 rows][208 B]: 24 units x (i, f, g, o) bf16
  char* const sc = sg + 16 * 208;                                      // c     [16 rows][112 B]: 24 units f32
  char* const sh = sc + 16 * 112;                                      // h     [16 rows][ 80 B]: 24 units bf16
  lds_u32* const cnt = (lds_u32*)(smem + kWsPartBytes + 4 * kWsStageChain);
  lds_u32* const ready = cnt + 4;
  // (a chunk that does not start the sequence: everything before it was published by launches that have completed)
  if (t_first > 0 && lane == 0) __hip_atomic_store(ready + q, (unsigned)t_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

  // the 6 cells of this lane: row r0 + (lane & 15), units u0 + 4 j + (lane >> 4)
  const int unit_q = u0 + (lane >> 4);
  const int row = r0 + (lane & 15);
  const int rowc = row < B ? row : B - 1;
  float cst[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) cst[j] = t_first > 0 ? c_all[((size_t)t_first * B + rowc) * H + unit_q + 4 * j] : 0.0f;

  bf16x8 wih[FUSED ? xkb : 1][6];
  f32x4 biasv[6];
  if constexpr (FUSED) {
#pragma unroll
    for (int kb = 0; kb < xkb; ++kb)
#pragma unroll
      for (int j = 0; j < 6; ++j)
        wih[kb][j] = *reinterpret_cast<const bf16x8*>(S.wih_blk + ((int64_t)(6 * slice + j) * xkb + kb) * 512 + lane * 8);
#pragma unroll
    for (int j = 0; j < 6; ++j)
      biasv[j] = *reinterpret_cast<const f32x4*>(S.bias + 16 * (size_t)(6 * slice + j) + 4 * (lane >> 4));
  }

  f32x4 nxt[P];
  const unsigned xslab = FUSED ? (unsigned)a.Bpad * (unsigned)S.I : 0u;
  const __amdgpu_buffer_rsrc_t in_rsrc = FUSED
      ? __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.x_blk), 0, __builtin_amdgcn_readfirstlane((int)((size_t)a.T * xslab * 2)), 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.xproj), 0, __builtin_amdgcn_readfirstlane((int)((size_t)a.T * B * 16 * H)), 0x00020000);
  const int xvoff = (int)(((size_t)rowc * 4 * H + 4 * (size_t)unit_q) * 4);
  auto request_input = [&](int t) {
    if constexpr (FUSED) {
      const int sbase = __builtin_amdgcn_readfirstlane((int)(((size_t)t * xslab + (size_t)(mt * 4 + q) * xkb * 512) * 2));
#pragma unroll
      for (int kb = 0; kb < xkb; ++kb) nxt[kb] = ns_bload_nt_f32x4(in_rsrc, lane * 16 + kb * 1024, sbase);
    } else {
      const int sbase = __builtin_amdgcn_readfirstlane((int)((size_t)t * B * 16 * H));
#pragma unroll
      for (int j = 0; j < 6; ++j) nxt[j] = ns_bload_nt_f32x4(in_rsrc, xvoff + j * 64, sbase);
    }
  };
  request_input(t_first);

  const __amdgpu_buffer_rsrc_t hdst_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.h_blk_all), 0, __builtin_amdgcn_readfirstlane((int)((size_t)(a.T + 1) * slab * 2)), 0x00020000);

  CSN_WSTAMP_INIT(q == 0);
#ifdef CSN_PSTAMPS
  const unsigned long long core0_ = __builtin_amdgcn_s_memtime(), wall0_ = wall_clock64();
#endif
  for (int s = 0; s < nsteps; ++s) {
    const int t = t_first + s;
    // ---- input part of the pre-activations (nothing here depends on h_{t-1})
    f32x4 pre[6];
    CSN_WSTAMP(14);    // input request
    ns_wait_vmcnt<0>();
    CSN_WSTAMP(8);     // input arrives (and every store of the step before has landed)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FUSED) {
#pragma unroll
      for (int j = 0; j < 6; ++j) pre[j] = biasv[j];
      ns_mfma_fence();
#pragma unroll
      for (int kb = 0; kb < xkb; ++kb)
#pragma unroll
        for (int j = 0; j < 6; ++j) ns_mfma<false>(pre[j], wih[kb][j], __builtin_bit_cast(bf16x8, nxt[kb]));
      ns_mfma_fence();
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) pre[j] = nxt[j];
    }
    __builtin_amdgcn_sched_barrier(0);

    CSN_WSTAMP(9);     // x MFMAs
    // ---- the four K-quarter partials of this chain and step
    {
      const unsigned want = 4u * (unsigned)(s + 1);
      const unsigned long long t_begin = wall_clock64();
      while (__hip_atomic_load(cnt + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t_begin > 2 * kNsSpinTimeoutTicks) {       // (the MFMA waves bound their own waits; this only ends a lost wave)
          __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    CSN_WSTAMP(10);    // wait for the partials
    if (t > 0) {
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const f32x4* const part = reinterpret_cast<const f32x4*>(smem + (size_t)(q * 4) * 6144) + j * 64 + lane;
        const f32x4 p0 = part[0], p1 = part[6144 / 16], p2 = part[2 * 6144 / 16], p3 = part[3 * 6144 / 16];
        pre[j] = pre[j] + ((p0 + p1) + (p2 + p3));
      }
    }
    // ---- gate math in place; results into the wave's transpose area
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const f32x4 v = pre[j];
      const float gi = fast_sigmoid(v[0]), gf = fast_sigmoid(v[1]), gg = fast_tanh(v[2]), go = fast_sigmoid(v[3]);
      const float cn = gf * cst[j] + gi * gg;
      const float hn = go * fast_tanh(cn);
      cst[j] = cn;
      const int r = lane & 15, unit = 4 * j + (lane >> 4);
      *reinterpret_cast<bf16x4*>(sg + r * 208 + unit * 8) = (bf16x4){(bf16_t)gi, (bf16_t)gf, (bf16_t)gg, (bf16_t)go};
      *reinterpret_cast<float*>(sc + r * 112 + unit * 4) = cn;
      *reinterpret_cast<bf16_t*>(sh + r * 80 + unit * 2) = (bf16_t)hn;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    CSN_WSTAMP(11);    // partial sums + gate math + transpose writes
    // ---- out with consecutive lanes on consecutive bytes; the hand-off block first
    if (lane < 48) {
      const int r = lane & 15, c8 = lane >> 4;              // 16 consecutive rows of a block are 256 contiguous bytes
      const nu32x4 v = *reinterpret_cast<const nu32x4*>(sh + r * 80 + c8 * 16);
      const unsigned hoff = (unsigned)(((size_t)(t + 1) * slab + blk_offset(r0 + r, u0 + 8 * c8, H)) * 2);
      if (local) ns_store_b128<false>(hdst_rsrc, hoff, v);
      else ns_store_b128<true>(hdst_rsrc, hoff, v);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (lane < 48) {
      const int r2 = lane / 3, c2 = lane % 3;
      if (r0 + r2 < B)
        nt_store(reinterpret_cast<nu32x4*>(h_all + ((size_t)(t + 1) * B + r0 + r2) * H + u0 + 8 * c2),
                 *reinterpret_cast<const nu32x4*>(sh + r2 * 80 + c2 * 16));
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = lane + 64 * k, r2 = idx / 6, ch = idx % 6;
      if (idx < 96 && r0 + r2 < B)
        nt_store(reinterpret_cast<nu32x4*>(c_all + ((size_t)(t + 1) * B + r0 + r2) * H + u0 + 4 * ch),
                 *reinterpret_cast<const nu32x4*>(sc + r2 * 112 + ch * 16));
    }
    if (gates != nullptr) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int idx = lane + 64 * k, r2 = idx / 12, ch = idx % 12;
        if (r0 + r2 < B)
          nt_store(reinterpret_cast<nu32x4*>(gates + ((size_t)t * B + r0 + r2) * 4 * H + 4 * (size_t)u0 + 8 * ch),
                   *reinterpret_cast<const nu32x4*>(sg + r2 * 208 + ch * 16));
      }
    }
    CSN_WSTAMP(12);    // transpose reads + store issue
    // only the hand-off store has to have landed before the flag: with every row real, the 3 (+3) stores behind it
    // were all issued (non-empty lane sets), so a counted wait covers exactly that first one
    if (full) {
      if (gates != nullptr) ns_wait_vmcnt<6>();
      else ns_wait_vmcnt<3>();
    } else {
      ns_wait_vmcnt<0>();
    }
    CSN_WSTAMP(13);    // hand-off store landed
    if (lane == 0) {
      unsigned* fl = flags + (size_t)(t + 1) * flag_step + slice;
      if (local) *fl = 1u;
      else __hip_atomic_store(fl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_sched_barrier(0);
    request_input(s + 1 < nsteps ? t + 1 : t);
    __builtin_amdgcn_sched_barrier(0);
    // ---- watch the chain's flag line for the MFMA waves: all 32 slices have published rows 16 q .. 16 q + 15 of h_t
    if (s + 1 < nsteps) {
      const unsigned* line = ns_uniform(flags + (size_t)(t + 1) * flag_step);
      const unsigned long long t_begin = wall_clock64();
      while (ns_flags_set_scalar(line, 32) < 32) {
        const unsigned long long waited = wall_clock64() - t_begin;
        if (waited > kNsSpinTimeoutTicks) {
          __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        if (waited > 2000ull && ws_error_seen(a)) break;     // (after 20 us of waiting: has someone else given up?)
      }
      if (lane == 0) __hip_atomic_store(ready + q, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      CSN_WSTAMP(15);  // flag line of the next step complete
    }
  }
#ifdef CSN_PSTAMPS
  if (stamping_ && lane == 0) {                       // shader clock / 100 MHz wall clock over the chunk
    atomicAdd(&g_wstamps[5], __builtin_amdgcn_s_memtime() - core0_);
    atomicAdd(&g_wstamps[6], wall_clock64() - wall0_);
  }
#endif
}

// FUSE: may a slot of this launch be the fused layer 0?  (run-time, workgroup-uniform: one launch advances layer 0 and
// the layers above it)
template <bool FUSE>
__global__ void __launch_bounds__(512) lstm_fwd_ws_kernel(PersistFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int MT = a.MT;
  const int tid = threadIdx.x;
  constexpr int nslices = 32;
  int grp, slice;
  if (a.xcd_groups) {
    grp = blockIdx.x & 7;
    slice = blockIdx.x >> 3;
    if (grp >= a.nslots * MT || slice >= nslices) return;
  } else {
    grp = blockIdx.x / nslices;
    slice = blockIdx.x % nslices;
  }
  const PersistFwdSlot& S = a.slot[grp / MT];
  const int mt = grp % MT;

  // ---- is this group on one XCD?  (lstm_fwd_persist.hip)
  int* const shared_word = reinterpret_cast<int*>(smem + kWsPartBytes + 4 * kWsStageChain + 32);
  lds_u32* const cnt = (lds_u32*)(smem + kWsPartBytes + 4 * kWsStageChain);
  if (tid < 8) cnt[tid] = 0u;                          // 4 partial counters + 4 ready words
  bool local = false;
  if (a.xcd_groups && a.agree != nullptr) {
    if (tid == 0) {
      const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 7u;        // hwreg(HW_REG_XCC_ID, 0, 4)
      const unsigned long long mine = 1ull | (1ull << (8 + 6 * xcc));
      __hip_atomic_fetch_add(a.agree + grp, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t_begin = wall_clock64();
      unsigned long long v;
      while (((v = __hip_atomic_load(a.agree + grp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0xffull) <
             (unsigned long long)nslices) {
        __builtin_amdgcn_s_sleep(1);
        if (ws_error_seen(a)) break;
        if (wall_clock64() - t_begin > kNsSpinTimeoutTicks) {
          __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      *reinterpret_cast<volatile int*>(shared_word) = (int)(((v >> (8 + 6 * xcc)) & 63ull) == (unsigned long long)nslices);
    }
    __syncthreads();
    local = *reinterpret_cast<volatile int*>(shared_word) != 0;
  }
  __syncthreads();                                   // (the counters are zero before any wave takes its role)

  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave < 4) {
    ws_mfma_wave(a, S, smem, slice, mt, wave);
    return;
  }
  if constexpr (FUSE) {
    if (__builtin_amdgcn_readfirstlane((int)(S.x_blk != nullptr)) != 0) {
      ws_gate_wave<true>(a, S, smem, slice, mt, local, wave - 4);
      return;
    }
  }
  ws_gate_wave<false>(a, S, smem, slice, mt, local, wave - 4);
}

template <bool FUSE>
static int launch_ws_t(const PersistFwdArgs& a, hipStream_t st) {
  if (int rc = ensure_dyn_lds<&lstm_fwd_ws_kernel<FUSE>>(kWsLdsBytes)) return rc;
  PersistFwdArgs b = a;
  b.grid_slices = 32;
  const unsigned grid = b.xcd_groups ? 8u * 32u : 32u * (unsigned)(b.MT * b.nslots);
  lstm_fwd_ws_kernel<FUSE><<<dim3(grid), 512, kWsLdsBytes, st>>>(b);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

int launch_fwd_ws(const PersistFwdArgs& a, hipStream_t st) {
  CSN_REQUIRE(a.H == 768 && a.chains == 4, "launch_fwd_ws: built for H = 768 with four chains per tile");
  CSN_REQUIRE(a.nslots >= 1 && a.nslots <= 4 && a.MT >= 1 && a.ngemm == 0, "launch_fwd_ws: bad slot / GEMM count");
  if (a.xcd_groups) CSN_REQUIRE(a.nslots * a.MT <= 8, "launch_fwd_ws: groups do not fit 8 XCDs");
  bool fused = false;
  for (int i = 0; i < a.nslots; ++i) {
    fused = fused || a.slot[i].x_blk != nullptr;
    CSN_REQUIRE(a.slot[i].x_blk == nullptr || a.slot[i].I == 128, "launch_fwd_ws: the fused input projection takes I = 128");
    CSN_REQUIRE(a.slot[i].xproj_bf16 == 0, "launch_fwd_ws: f32 input projection only");
  }
  return fused ? launch_ws_t<true>(a, st) : launch_ws_t<false>(a, st);
}

}  // namespace csn

#ifdef CSN_PSTAMPS
extern "C" int csn_debug_read_wstamps(unsigned long long* out) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamps), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_wstamps), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif
