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
//   waves 4-7  gate waves, one per chain; they own everything of the chain that is latency.  Off the critical path:
//              the input part of the pre-activations -- layer 0: bias + x_t W_ih^T by 24 MFMAs of its own against 96
//              registers of W_ih; layers above: the projection read from HBM.  On it: wait for the chain's counter,
//              add the MFMA waves' tiles, gate math in the accumulator layout ((i, f, g, o) of one cell in one lane;
//              c stays in registers), a wave-private LDS transpose, stores with consecutive lanes on consecutive
//              bytes (the h hand-off block first), a full store drain, the
//              chain's flag.  Then it watches the flag line of the next step (scalar loads), and as soon as all 32
//              slices have published, brings the chain's 16 x 768 slice of h_t into LDS (24 LDS-DMA pieces of 1 KB
//              that already have the operand layout; the next input request goes out BEHIND them and a counted wait
//              covers the pieces only) and posts ready[q].
//   waves 0-3  (one per SIMD)  MFMA waves: no global memory access at all in the loop.  The 6 gate-row tiles x 2
//              K-halves of the slice are dealt 3 : 3 : 3 : 3 (wave 0: tile 0 whole + tile 1 lower K-half, wave 1: tile 1
//              upper half + tile 2 whole, waves 2, 3 the same on tiles 3-5): 36 weight fragments = 144 registers per
//              wave.  Per chain and step: wait for ready[q] (an LDS read), stream the 24 operand fragments from LDS
//              (ds_read_b128, four in flight), 36 MFMAs into two accumulators, the two tiles to LDS, bump the chain's
//              LDS counter.  About 0.35 us per item: the MFMA waves are idle most of the time -- the point is that
//              no chain ever waits for another chain's memory latency.
//
// No workgroup barrier anywhere in the loop: gate wave -> MFMA waves through ready[q], MFMA waves -> gate wave through
// the monotonic counter cnt[q]; the h buffer and the tile slots of a chain cannot be overwritten early because each
// side writes them only after the other has signalled that it is done with the step before.  Chains of one workgroup
// run out of phase and independently of each other.
//
// (First version, measured: MFMA waves K-split 4 ways, each fetching its private K-quarter of the chain straight into
// registers and polling the flags itself -- 481 us per launch; with the polls moved to the gate waves 356 us: an MFMA
// wave that waits 1 us for ITS operand blocks the three other chains.  Hence LDS as the meeting point.)
//
// Registers: 8 waves per CU -> 256 per wave, which the compiler splits 128 VGPRs : 128 AGPRs.  MFMA waves: 28 weight
// fragments in AGPRs + 8 in VGPRs (an MFMA takes its A operand from either), 2 accumulators, 4 operand fragments.
// Gate waves: 96 AGPRs of W_ih + 24 pre-activations + 16-24 input + 6 c + math.
// LDS: 4 x 24 KB h buffers + 4 x 8 KB tile slots (reused as the chain's transpose area) + 64 B of counters.
#include "../csn_common.h"
#include "../lstm_cell_common.h"
#include "../lstm_cell_blk.h"
#include "../lstm_ns_util.h"

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
static constexpr int kWsAF = 28;                                   // weight fragments (of 36) kept in AGPRs, the rest in VGPRs
static constexpr int kWsHBuf = kWsKB * 1024;                       // one chain's 16 x 768 slice of h, 24 operand fragments
static constexpr int kWsSlots = 8 * 1024;                          // one chain's 8 tile slots (f32 16 x 16 tiles); reused as its transpose area
static constexpr int kWsStageChain = 16 * (208 + 112 + 80);        // gates (bf16 x 4) + c (f32) + h (bf16) of 16 rows x 24 units, padded rows
static_assert(kWsStageChain <= kWsSlots, "the transpose area lives in the chain's tile slots");
static constexpr int kWsCtlOff = 4 * kWsHBuf + 4 * kWsSlots;       // cnt[4] | ready[4] | scratch word
static constexpr int kWsLdsBytes = kWsCtlOff + 64 + 96 * 4;        // + the slice's 96 bias values (fused layer 0)

typedef __attribute__((address_space(3))) unsigned lds_u32;

__device__ __forceinline__ bool ws_error_seen(const PersistFwdArgs& a) {
  return __hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
}

// ---------------------------------------------------------------------------------------------------------------
// MFMA wave w: one whole gate-row tile + one K-half of the tile it shares with its neighbour, every chain in turn
template <int odd>
__device__ __forceinline__ void ws_mfma_wave(const PersistFwdArgs& a, const PersistFwdSlot& S, char* smem, int slice, int w) {
  constexpr int KB = kWsKB, KH = KB / 2;
  const int lane = threadIdx.x & 63;
  const int t_first = S.t0, nsteps = S.nsteps;
  const int tile_full = 6 * slice + 3 * (w >> 1) + 2 * odd;       // rows of the interleaved 4H axis
  const int tile_half = 6 * slice + 3 * (w >> 1) + 1;
  const int kb_half0 = odd * KH;                                   // the K-half of the shared tile this wave multiplies

  // stationary operand: fragment f < 24 is k-block f of the whole tile, fragment 24 + i k-block kb_half0 + i of the
  // shared one
  bf16x8 wa[kWsAF], wv[36 - kWsAF];
#pragma unroll
  for (int f = 0; f < 36; ++f) {
    const int tile = f < KB ? tile_full : tile_half;
    const int kb = f < KB ? f : kb_half0 + (f - KB);
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(S.w_blk + ((int64_t)tile * KB + kb) * 512 + lane * 8);
    if (f < kWsAF) wa[f] = v;
    else wv[f - kWsAF] = v;
  }
  lds_u32* const cnt = (lds_u32*)(smem + kWsCtlOff);
  lds_u32* const ready = cnt + 4;
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  auto is_ready = [&](int t, int q) {
    return __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(ready + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >= t;
  };
  // (first: the accumulator starts from the literal 0 -- no register initialisation that would have to be kept, with
  // wait states the hazard recogniser cannot see are needed, in front of an inline-asm MFMA)
  auto mfma_frag = [&](f32x4& acc, int f, const bf16x8& h, bool first) {
    static_assert(kWsAF > kWsKB, "the first fragment of either tile lives in AGPRs");
    if (first) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(wa[f]), "v"(h));
    else if (f < kWsAF) ns_mfma<false>(acc, wa[f], h);
    else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wv[f - kWsAF]), "v"(h));
  };

  const int N = 4 * nsteps;
  CSN_WSTAMP_INIT(w == 0);
  for (int i = 0; i < N; ++i) {
    const int t = t_first + (i >> 2), q = i & 3;
    if (t > 0) {
      // ---- h_{t-1} of the chain is in LDS (its gate wave posts the step)
      {
        const unsigned long long t_begin = wall_clock64();
        while (!is_ready(t, q)) {
          __builtin_amdgcn_s_sleep(1);
          if (wall_clock64() - t_begin > 2 * kNsSpinTimeoutTicks) {     // (the gate waves bound their own polls; this only ends a lost wave)
            __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      CSN_WSTAMP(0);   // waiting for the chain
      unsigned zero_;
      asm volatile("v_mov_b32 %0, 0" : "=v"(zero_));
      const unsigned hb = lds_base + (unsigned)(q * kWsHBuf) + (unsigned)lane * 16u + zero_;
      f32x4 acc_full, acc_half;
      bf16x8 hf[4];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < 4; ++d) hf[d] = ns_lds_read_b128(hb + (unsigned)(d * 1024));
      ns_mfma_fence();
#pragma unroll
      for (int kb = 0; kb < KB; ++kb) {
        const int behind = KB - 1 - kb < 3 ? KB - 1 - kb : 3;         // reads issued after the one this k-block needs
        if (behind == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
        else if (behind == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else if (behind == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_frag(acc_full, kb, hf[kb & 3], kb == 0);
        if (odd ? kb >= KH : kb < KH) mfma_frag(acc_half, KB + (odd ? kb - KH : kb), hf[kb & 3], kb == (odd ? KH : 0));
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 4 < KB) hf[kb & 3] = ns_lds_read_b128(hb + (unsigned)((kb + 4) * 1024));
        __builtin_amdgcn_sched_barrier(0);
      }
      // (the wait states between the last MFMA and the first read of its result, tied to the registers so that no
      // use can be scheduled in front of them)
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc_full), "+v"(acc_half));
      CSN_WSTAMP(1);   // LDS reads + MFMAs
      __builtin_amdgcn_sched_barrier(0);
      // slots of the chain: wave 0 -> 0 (tile 0), 1 (tile 1 lower); wave 1 -> 2 (tile 1 upper), 3 (tile 2); waves 2, 3 -> 4..7
      f32x4* const slot = reinterpret_cast<f32x4*>(smem + 4 * kWsHBuf + (size_t)q * kWsSlots) + (size_t)(2 * w) * 64 + lane;
      slot[odd ? 64 : 0] = acc_full;
      slot[odd ? 0 : 64] = acc_half;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt + q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __builtin_amdgcn_sched_barrier(0);
    CSN_WSTAMP(2);     // tiles to LDS + counter
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
  char* const hbuf = smem + (size_t)q * kWsHBuf;                       // the chain's slice of h_{t-1}: 24 operand fragments
  char* const slots = smem + 4 * kWsHBuf + (size_t)q * kWsSlots;       // the MFMA waves' 8 tiles; then this wave's transpose area:
  char* const sg = slots;                                              // gates [16 rows][208 B]: 24 units x (i, f, g, o) bf16
  char* const sc = sg + 16 * 208;                                      // c     [16 rows][112 B]: 24 units f32
  char* const sh = sc + 16 * 112;                                      // h     [16 rows][ 80 B]: 24 units bf16
  lds_u32* const cnt = (lds_u32*)(smem + kWsCtlOff);
  lds_u32* const ready = cnt + 4;

  // the 6 cells of this lane: row r0 + (lane & 15), units u0 + 4 j + (lane >> 4)
  const int unit_q = u0 + (lane >> 4);
  const int row = r0 + (lane & 15);
  const int rowc = row < B ? row : B - 1;
  float cst[6];
#pragma unroll
  for (int j = 0; j < 6; ++j) cst[j] = t_first > 0 ? c_all[((size_t)t_first * B + rowc) * H + unit_q + 4 * j] : 0.0f;

  bf16x8 wih[FUSED ? xkb : 1][6];
  const float* const lds_bias = reinterpret_cast<const float*>(smem + kWsCtlOff + 64);   // (written by the kernel prologue)
  if constexpr (FUSED) {
#pragma unroll
    for (int kb = 0; kb < xkb; ++kb)
#pragma unroll
      for (int j = 0; j < 6; ++j)
        wih[kb][j] = *reinterpret_cast<const bf16x8*>(S.wih_blk + ((int64_t)(6 * slice + j) * xkb + kb) * 512 + lane * 8);
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

  const __amdgpu_buffer_rsrc_t hdst_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.h_blk_all), 0, __builtin_amdgcn_readfirstlane((int)((size_t)(a.T + 1) * slab * 2)), 0x00020000);
  // h_{t-1} of the chain (slab t, row group 4 mt + q: 24 contiguous 1 KB fragments) -> LDS, then the input of step t
  // BEHIND it; the counted wait covers the 24 pieces only (vmcnt counts in issue order), then the MFMA waves may start
  auto fetch_and_post = [&](int t) {
    const int sbase = __builtin_amdgcn_readfirstlane((int)(((size_t)t * slab + (size_t)(mt * 4 + q) * kWsKB * 512) * 2));
#pragma unroll
    for (int kb = 0; kb < kWsKB; ++kb) ns_dma16_sc1(hdst_rsrc, hbuf + kb * 1024, lane * 16, sbase + kb * 1024);
    __builtin_amdgcn_sched_barrier(0);
    request_input(t);
    __builtin_amdgcn_sched_barrier(0);
    ns_wait_vmcnt<P>();
    __builtin_amdgcn_sched_barrier(0);
#ifndef CSN_WS_NO_SETTLE
    __builtin_amdgcn_s_sleep(4);
#endif
    __builtin_amdgcn_sched_barrier(0);
    if (lane == 0) __hip_atomic_store(ready + q, (unsigned)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  // (a chunk that does not start the sequence: everything before it was published by launches that have completed)
  if (t_first > 0) fetch_and_post(t_first);
  else request_input(t_first);

  CSN_WSTAMP_INIT(q == 0);
#ifdef CSN_PSTAMPS
  const unsigned long long core0_ = __builtin_amdgcn_s_memtime(), wall0_ = wall_clock64();
#endif
  for (int s = 0; s < nsteps; ++s) {
    const int t = t_first + s;
    // ---- input part of the pre-activations (nothing here depends on h_{t-1})
    f32x4 pre[6];
    ns_wait_vmcnt<0>();
    CSN_WSTAMP(8);     // input arrives (and every store of the step before has landed)
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FUSED) {
#pragma unroll
      for (int j = 0; j < 6; ++j) pre[j] = *reinterpret_cast<const f32x4*>(lds_bias + 16 * j + 4 * (lane >> 4));
      ns_mfma_fence();
#pragma unroll
      for (int kb = 0; kb < xkb; ++kb)
#pragma unroll
        for (int j = 0; j < 6; ++j) ns_mfma<false>(pre[j], wih[kb][j], __builtin_bit_cast(bf16x8, nxt[kb]));
      asm volatile("s_nop 15\n\ts_nop 7" : "+v"(pre[0]), "+v"(pre[1]), "+v"(pre[2]), "+v"(pre[3]), "+v"(pre[4]), "+v"(pre[5]));
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) pre[j] = nxt[j];
    }
    __builtin_amdgcn_sched_barrier(0);

    CSN_WSTAMP(9);     // x MFMAs
    // ---- the MFMA waves' tiles of this chain and step
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
    CSN_WSTAMP(10);    // wait for the tiles
    if (t > 0) {
      // slots: 0 tile 0 | 1, 2 the K-halves of tile 1 | 3 tile 2 | 4 tile 3 | 5, 6 the K-halves of tile 4 | 7 tile 5
      const f32x4* const sl = reinterpret_cast<const f32x4*>(slots) + lane;
      const f32x4 s0 = sl[0], s1 = sl[64], s2 = sl[128], s3 = sl[192], s4 = sl[256], s5 = sl[320], s6 = sl[384], s7 = sl[448];
      pre[0] = pre[0] + s0;
      pre[1] = pre[1] + (s1 + s2);
      pre[2] = pre[2] + s3;
      pre[3] = pre[3] + s4;
      pre[4] = pre[4] + (s5 + s6);
      pre[5] = pre[5] + s7;
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
    // every store of the step has landed before the flag (a counted wait that covered only the hand-off store was
    // measured here first; counted waits on stores are not relied upon any more, see lstm_fwd_persist.hip)
    ns_wait_vmcnt<0>();
    CSN_WSTAMP(13);    // hand-off store landed
    if (lane == 0) {
      unsigned* fl = flags + (size_t)(t + 1) * flag_step + slice;
      if (local) *fl = 1u;
      else __hip_atomic_store(fl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- watch the chain's flag line: once all 32 slices have published rows 16 q .. 16 q + 15 of h_t, bring them in
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
      CSN_WSTAMP(15);  // flag line of the next step complete
      fetch_and_post(t + 1);
      CSN_WSTAMP(14);  // h tile into LDS
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
  int* const shared_word = reinterpret_cast<int*>(smem + kWsCtlOff + 32);
  lds_u32* const cnt = (lds_u32*)(smem + kWsCtlOff);
  if (tid < 8) cnt[tid] = 0u;                          // 4 tile counters + 4 ready words
  if constexpr (FUSE) {
    if (S.x_blk != nullptr && tid >= 64 && tid < 160)
      reinterpret_cast<float*>(smem + kWsCtlOff + 64)[tid - 64] = S.bias[96 * (size_t)slice + (tid - 64)];
  }
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
    if (wave & 1) ws_mfma_wave<1>(a, S, smem, slice, wave);
    else ws_mfma_wave<0>(a, S, smem, slice, wave);
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
