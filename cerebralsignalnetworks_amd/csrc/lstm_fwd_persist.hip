// K3 forward, weight-stationary form: ONE launch advances a layer through a chunk of timesteps.
// Replaces the per-step body of nn.LSTM reached at /root/reference/LSTMDistill.py:118,132.
//
// Why: in the per-step launches of lstm_cell_blk.hip, 60 % of the operand bytes a workgroup loads
// per step are its W_hh slice, re-read every step at the per-CU L2 rate (~55 GB/s per CU measured).
// Here a workgroup owns a (64 rows x 4*NQ units) tile for the whole chunk, keeps its W_hh slice in
// registers (each wave: its K quarter of all 4*NQ gate-row tiles = KS*NQ fragments, 144 VGPRs at
// H = 768) and its cell state c in registers, and per step only streams its 64 rows of h_{t-1}.
//
// One launch advances up to 4 layers, each through its own chunk (a wavefront diagonal over chunks).  A
// hand-off GROUP is one layer's M-tile (64 batch rows): its nslices workgroups exchange h every step,
// different groups never talk.  Two hand-off forms, same kernel, bit-identical results:
//
//  * L2-local (grouped launch, xcd_groups != 0): group = blockIdx.x % 8.  The hardware deals consecutive
//    workgroups round-robin over the 8 XCDs, so a group's workgroups share one XCD and one L2 -- at cfg2
//    2 layers x 4 M-tiles = 8 groups of 32 workgroups = one group per XCD, one workgroup per CU.  That is
//    an observation, not a contract, so each group VERIFIES it at run time: every workgroup adds
//    (1, 1 << 6*XCC_ID) to the group's agreement word with an agent-scope atomic and waits for all
//    arrivals; all then read the same word and take the same decision.  If the group is on one XCD:
//      producer: h slice with PLAIN stores (they stay in that L2), every storing wave s_waitcnt vmcnt(0),
//                workgroup barrier, one lane sets the workgroup's word of the group's flag line (plain);
//      consumer: each wave polls the flags of exactly the producers whose units it multiplies (its K
//                quarter) with sc1 loads (L1 bypassed, L2-served) and then loads h with sc1 loads.
//    Measured per step at cfg2 (tools/persist_bench.hip mode 2): wait 0.7 us, h loads + MFMA 1.8,
//    LDS reduction 0.6, epilogue 2.1, drain + signal 0.5 = 6.2 us (placement-independent form: 8.2 us).
//  * placement-independent (any grouping; cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md
//    "Valid forms", row 1): the h slice and the flag are written through (agent-scope = sc1 stores), the
//    polls and every load of the handed-off bytes are sc1 loads.
//
// In both forms every step's h lives at its OWN address (h_blk_all[t], never reused inside a forward),
// so no L1 / L2 in the chip can hold an older copy of a line being handed off, and nothing is queued in
// a wave's vector-memory pipe ahead of its polls and h loads (results return in issue order: the next
// step's input projection is requested behind the MFMAs, a full step early -- with it in front of the
// poll the wait was 2.5 us instead of 0.7).
// Splitting the 64 rows into two alternating 32-row halves to hide the wait was tried and is slower
// (12.8 us per step: the half-size loads are latency-bound), so the whole tile advances together.
// All workgroups of a launch must be co-resident (1 per CU);
// every spin is bounded: after kSpinTimeoutTicks the workgroup raises *error_flag (sticky: later
// waits return immediately) so a scheduling accident ends in a reported error, never in a hang.
#include "csn_common.h"
#include "lstm_cell_common.h"
#include "lstm_cell_blk.h"

#ifdef CSN_PSTAMPS
#ifndef CSN_STAMP_BLOCK
#define CSN_STAMP_BLOCK 11     // group 3 (layer 0 at cfg2), slice 1; 15 = group 7 (layer 1)
#endif
// diagnostic build only (tools/persist_bench.hip): per-phase wall-clock sums of workgroup (0,0)
__device__ unsigned long long g_pstamps[16];
__device__ unsigned long long g_span[5] = {~0ull, 0ull, ~0ull, 0ull, 0ull};
__device__ unsigned long long g_grp_alive[32];      // [0..7] ticks alive summed per group (slice 0 only), [8..15] launches, [16..23] ticks in the step loop, [24..31] steps
#define CSN_PSTAMP(i)                                                          \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    if (tid == 0 && blockIdx.x == CSN_STAMP_BLOCK) {                           \
      const unsigned long long now_ = wall_clock64();                          \
      atomicAdd(&g_pstamps[i], now_ - last_);                                  \
      last_ = now_;                                                            \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#else
#define CSN_PSTAMP(i)
#endif
#ifdef CSN_PSTAMPS
// (diagnostic: how long the launch's waves were alive -- first entry to last exit over all workgroups -- to set against
// the duration rocprofv3 reports for the dispatch, which also holds the dispatch itself and the end-of-kernel cache
// write-back)
__device__ __forceinline__ void span_exit(unsigned long long t_entry) {
  if (threadIdx.x != 0) return;
  const unsigned long long t_exit = wall_clock64();
  atomicMin(&g_span[0], t_entry);
  atomicMax(&g_span[1], t_entry);
  atomicMin(&g_span[2], t_exit);
  atomicMax(&g_span[3], t_exit);
  __threadfence();
  if (atomicAdd(&g_span[4], 1ull) + 1ull == (unsigned long long)gridDim.x) {      // the last workgroup out
    __threadfence();
    const unsigned long long e0 = atomicAdd(&g_span[0], 0ull), e1 = atomicAdd(&g_span[1], 0ull);
    const unsigned long long x0 = atomicAdd(&g_span[2], 0ull), x1 = atomicAdd(&g_span[3], 0ull);
    atomicAdd(&g_pstamps[10], x1 - e0);      // waves alive
    atomicAdd(&g_pstamps[11], e1 - e0);      // entry skew
    atomicAdd(&g_pstamps[12], x1 - x0);      // exit skew
    atomicAdd(&g_pstamps[13], 1ull);         // launches
    g_span[0] = ~0ull; g_span[1] = 0ull; g_span[2] = ~0ull; g_span[3] = 0ull; g_span[4] = 0ull;
    __threadfence();
  }
}
#endif


namespace csn {

// float4 slots per accumulator tile in the reduction buffer: 64 + padding.  The epilogue's lanes walk (row, unit quad j)
// with j fastest and read slot (tile j, row): with 65 a lane's 16-byte bank group is (j + row) mod 16 -- three lanes of a
// 16-lane group on one group; with 71 it is (7 j + row) mod 16, distinct except for one rare pair.
static constexpr int kRedTile = 71;
static constexpr unsigned long long kSpinTimeoutTicks = 20000000ull;   // 0.2 s of the 100 MHz wall clock

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

__device__ __forceinline__ bf16x8 load_sc1_b128(__amdgpu_buffer_rsrc_t rsrc, int byte_off, int soff = 0) {
  // aux = 16: sc1 (agent-coherent, bypasses this CU's L1; MI355X_MICROARCH.md visibility table)
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, soff, 16);
  union { u32x4 u; bf16x8 b; } cvt;
  cvt.u = v;
  return cvt.b;
}

// CSN_SLAB_TAGS (`make tags`, a DEBUG library): every 8-byte hand-off piece carries, in the lowest mantissa bit of
// its first element, bit 2 of the step it belongs to.  The previous occupant of a ring slot is the piece of step
// t - 4, whose tag is the opposite one -- so a consumer that is served a STALE occupant (data, not the sentinel: the
// one case the sentinel proof cannot see) raises status word 2.  The tag perturbs h by one bf16 ulp: a detection
// build, not a parity build.
__device__ __forceinline__ unsigned long long pack_h_piece(const float (&v)[4], int tag) {
  union { bf16x4 b; unsigned long long u; } cvt;
  cvt.b = (bf16x4){(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
#ifdef CSN_SLAB_TAGS
  cvt.u = (cvt.u & ~1ull) | (unsigned long long)(tag & 1);
#endif
  return cvt.u;
}

__device__ __forceinline__ void store_wt_b64(bf16_t* p, const float (&v)[4], int tag) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), pack_h_piece(v, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void store_plain_b64(bf16_t* p, const float (&v)[4], int tag) {
  *reinterpret_cast<unsigned long long*>(p) = pack_h_piece(v, tag);
}

// Bounded wait of one lane until *p (read with relaxed agent-scope = sc1 loads, which bypass this CU's L1)
// satisfies `done`; returns the last value read.  On time-out raises the sticky error flag.
template <typename T, typename Pred>
__device__ __forceinline__ T bounded_poll(const T* p, unsigned* error_flag, Pred done) {
  const unsigned long long t_begin = wall_clock64();
  T v;
  while (!done(v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
    __builtin_amdgcn_s_sleep(1);
    if (__hip_atomic_load(error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
    if (wall_clock64() - t_begin > kSpinTimeoutTicks) {
      __hip_atomic_store(error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
  return v;
}

#ifndef CSN_FWD_RING
#define CSN_FWD_RING 4
#endif
template <int NQ, int KS, bool DPOLL>
__global__ void __launch_bounds__(256) lstm_fwd_persist_kernel(PersistFwdArgs a) {
  constexpr int NT = 4 * NQ;
  constexpr int NPAIR = 64 * NQ;
  constexpr int NPASS = (NPAIR + 255) / 256;
  // The epilogue's work items are (row, unit-quad) pairs, one per thread and pass.  At NQ = 6 there are 384 of them: a full
  // pass and 128 left over -- as a second pass of whole pairs those kept waves 0 and 1 busy for a full pass while waves 2
  // and 3 waited at the barrier (an ablation without them: 212 -> 185 us per launch).  HALFQ: the 128 quads of that last
  // pass are split into unit PAIRS over all 256 threads (thread t: quad 256 NFULL + t / 2, units 2 (t & 1), +1), so every
  // wave runs half a pass; the two lanes of a quad meet through a DPP swap for the one 8-byte hand-off piece.
  constexpr bool HALFQ = (NPAIR % 256) == 128;
  constexpr int NFULL = NPAIR / 256;
  extern __shared__ __attribute__((aligned(16))) float4 red[];   // [4][NT][kRedTile], then [NQ][4] bias
  const int B = a.B, H = a.H, MT = a.MT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CSN_PSTAMPS
  const unsigned long long t_entry_ = wall_clock64();
#endif
  const int nslices = H / (4 * NQ);
  // hand-off group = (slot, M-tile); a workgroup's place in it = its slice of the hidden units
  int grp, slice;
  if (a.xcd_groups) {
    grp = blockIdx.x & 7;
    slice = blockIdx.x >> 3;
    if (grp >= a.nslots * MT) {
#ifdef CSN_PSTAMPS
      span_exit(t_entry_);
#endif
      return;
    }
  } else {
    grp = blockIdx.x / nslices;
    slice = blockIdx.x % nslices;
  }
  const PersistFwdSlot& S = a.slot[grp / MT];
  const int mt = grp % MT;
  const int u0 = slice * 4 * NQ, m0 = mt * 64;
  const int kblocks = H >> 5;
  const int ks_beg = wave * KS;                        // KS = kblocks / 4 k-steps per wave
  const size_t slab = (size_t)a.Bpad * H;              // elements of one fragment-major h slab
  const bf16_t* const w_blk = S.w_blk;
  const float* const xproj = S.xproj;
  bf16_t* const gates = S.gates;
  float* const c_all = S.c_all;
  bf16_t* const h_all = S.h_all;
  bf16_t* const h_blk_all = S.h_blk_all;
  unsigned* const flags = S.flags + (size_t)mt * kPersistFlagLine;    // [T+1][MT][line]: word i = slice i has published
  const size_t flag_step = (size_t)MT * kPersistFlagLine;
  const int t_first = S.t0, nsteps = S.nsteps;
  // Hand-off by DATA (a.data_polls): the slabs form a ring of 4 whose unwritten regions hold a sentinel (all ones: two
  // bf16 NaNs, which h = o tanh(c) never is); a consumer watches one word per producer and then verifies every
  // 8-byte piece it loaded, so the producers neither drain their stores nor set a flag.  See the step loop.
  constexpr bool dpoll = DPOLL;      // (compile time: as a run-time switch its branches slowed the flag form down by 4 %)
  // fused input projection: wave w contracts k-block w of the input features (I <= 128: at most one per wave)
  const bf16_t* const x_blk = S.x_blk;
  const bool fused = x_blk != nullptr;
  const int xkb = S.I >> 5;
  // (wave index through readfirstlane: the condition around the x MFMAs must be a SCALAR branch -- MFMA ignores
  // EXEC, so under a mere EXEC mask it would execute in every wave)
  const bool xwave = fused && __builtin_amdgcn_readfirstlane(wave) < xkb;

  // ---- is this group on one XCD?  Every workgroup adds (1, 1 << its XCC field) to the group's word and
  // waits for all `nslices` arrivals; all of them then see the same word and take the same decision.
  bool local = false;
  if (a.xcd_groups && a.agree != nullptr) {
    if (tid == 0) {
      const unsigned xcc = __builtin_amdgcn_s_getreg(6164) & 7u;        // hwreg(HW_REG_XCC_ID, 0, 4)
      const unsigned long long mine = 1ull | (1ull << (8 + 6 * xcc));
      __hip_atomic_fetch_add(a.agree + grp, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long v = bounded_poll(a.agree + grp, a.error_flag, [&](unsigned long long x) {
        return (unsigned)(x & 0xffull) >= (unsigned)nslices;
      });
      ((__attribute__((address_space(3))) int*)(__attribute__((address_space(3))) void*)red)[0] = (int)(((v >> (8 + 6 * xcc)) & 63ull) == (unsigned long long)nslices);
    }
    __syncthreads();
    local = ((__attribute__((address_space(3))) int*)(__attribute__((address_space(3))) void*)red)[0] != 0;    // (LDS address space: no flat_ instruction in these kernels -- FLAT retires out of order)
    __syncthreads();
  }

  // ---- stationary operands: this wave's K quarter of the workgroup's W_hh rows --------------
  // The workgroups of a group all read the same h slab; each walks its k-blocks from its own offset (register i
  // holds k-block (i + rot) % KS) so that their requests spread over the L2 channels (lstm_bwd_persist.hip).
  const int rot = a.rotate ? (slice * KS) / nslices : 0;
  bf16x8 wreg[KS][NQ];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const int kk = (ks + rot) % KS;
#pragma unroll
    for (int j = 0; j < NQ; ++j)
      wreg[ks][j] = *reinterpret_cast<const bf16x8*>(w_blk + ((int64_t)((u0 >> 2) + j) * kblocks + ks_beg + kk) * 512 + lane * 8);
  }

  // Two register sets of 8 x 16 bytes with a mode-dependent meaning (the modes never mix inside a workgroup):
  //   plain: cur = this step's input projection xp[pass][quad], nxt = the next step's (in flight)
  //   fused: cur = this wave's W_ih fragments (stationary), nxt[0..3] = x fragments of the step (in flight a step early)
  static_assert(NQ <= 8 && NPASS * 4 <= 8, "register sets too small");
  f32x4 cur[8], nxt[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) cur[i] = nxt[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float* const bias_lds = reinterpret_cast<float*>(red + 4 * NT * kRedTile);      // [NQ][4] float4, behind the reduction buffer
  if (fused) {
    if (xwave) {
#pragma unroll
      for (int j = 0; j < NQ; ++j)
        cur[j] = *reinterpret_cast<const f32x4*>(S.wih_blk + ((int64_t)((u0 >> 2) + j) * xkb + wave) * 512 + lane * 8);
    }
    // the bias of this workgroup's gate rows in accumulator order: entry (j, lane >> 4) = rows 16 j + 4 (lane >> 4) ..+3
    if (tid < NQ * 4)
      reinterpret_cast<float4*>(bias_lds)[tid] =
          *reinterpret_cast<const float4*>(S.bias + 4 * (size_t)u0 + 16 * (tid >> 2) + 4 * (tid & 3));
    __syncthreads();
  }
  const unsigned xslab = (unsigned)a.Bpad * (unsigned)S.I;    // elements of one x slab

  // ---- cell state of the (row, unit-quad) pairs this thread owns ------------------------------
  float4 cst[NPASS];
  int prow[NPASS], puq[NPASS], prl[NPASS], pj[NPASS];
  bool pok[NPASS];
  const int hq = tid & 1;                       // split pass: which unit pair of the quad
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const bool hp = HALFQ && ps == NFULL;
    const int p = hp ? NFULL * 256 + (tid >> 1) : tid + ps * 256;
    prl[ps] = p / NQ;
    pj[ps] = p % NQ;
    prow[ps] = m0 + prl[ps];
    puq[ps] = u0 + 4 * pj[ps];                  // (the quad's first unit, also in the split pass)
    pok[ps] = p < NPAIR && prow[ps] < B;
    cst[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pok[ps] && t_first > 0) {
      const float* cp = c_all + ((size_t)t_first * B + prow[ps]) * H + puq[ps];
      if (hp) {
        const f32x2 c2 = *reinterpret_cast<const f32x2*>(cp + 2 * hq);
        cst[ps] = make_float4(c2[0], c2[1], 0.f, 0.f);
      } else {
        cst[ps] = *reinterpret_cast<const float4*>(cp);
      }
    }
  }

  const __amdgpu_buffer_rsrc_t hsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)h_blk_all, 0, (int)((size_t)(a.T + 1) * slab * 2), 0x00020000);
#ifdef CSN_PSTAMPS
  unsigned long long last_ = wall_clock64();
  if (tid == 0 && blockIdx.x == CSN_STAMP_BLOCK) atomicAdd(&g_pstamps[6], last_ - t_entry_);   // prologue of this launch
  const unsigned long long t_loop_ = last_;
#endif

  auto request_input = [&](int t) {
    if (fused) {
      if (xwave) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
          nxt[rg] = nt_load(reinterpret_cast<const f32x4*>(x_blk + (size_t)t * xslab + ((size_t)((m0 >> 4) + rg) * xkb + wave) * 512 + lane * 8));
      }
      return;
    }
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps)
      if (pok[ps]) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(xproj + ((size_t)t * B + prow[ps]) * 4 * H + 4 * (size_t)puq[ps]);
        if (HALFQ && ps == NFULL) {
#pragma unroll
          for (int q = 0; q < 2; ++q) nxt[ps * 4 + q] = nt_load(xr + 2 * hq + q);
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) nxt[ps * 4 + q] = nt_load(xr + q);
        }
      }
  };
  request_input(t_first);

  for (int s = 0; s < nsteps; ++s) {
    const int t = t_first + s;
    // this step's input projection (or input) was requested a step ago (below, behind the MFMAs): nothing may sit
    // in the vector-memory queue ahead of the flag polls and the h loads -- results return in issue order, so a
    // queued HBM read would add its latency to every hand-off (measured: wait 2.5 -> 0.5 us per step)
    bf16x8 xf[4];
    if (fused) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) xf[rg] = __builtin_bit_cast(bf16x8, nxt[rg]);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }

    // slab (ring slot) this step reads h_{t-1} from / writes h_t to
    const size_t slot_in = dpoll ? (size_t)(t & 3) : (size_t)t;
    const size_t slot_out = dpoll ? (size_t)((t + 1) & 3) : (size_t)(t + 1);

    f32x4 acc[4][NQ];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int j = 0; j < NQ; ++j) acc[rg][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (fused && wave == 0) {                 // one wave's partial sum starts from the bias
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const float4 bz = reinterpret_cast<const float4*>(bias_lds)[j * 4 + (lane >> 4)];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) acc[rg][j] = (f32x4){bz.x, bz.y, bz.z, bz.w};
      }
    }

    if (t > 0) {
      // Wait for h_{t-1} (slot t).  A wave only multiplies its K quarter of h, i.e. the units of the
      // nslices/4 producer slices [wave * nslices/4, ...): it polls exactly those flags of the group's flag
      // line, lane i the flag of its i-th producer (sc1 loads: L1 bypassed, L2-served), and then loads --
      // no workgroup barrier, the polling wave is the loading wave.
      {
        const int npw = nslices >> 2;
        // flags: word i of the (t, M-tile) line.  Data polls: four words per producer, one from the LAST store
        // instruction of each of its waves (thread 64 w + 63 in the last pass in which it owns a pair) -- a hint
        // that the whole region is there; the proof is the check of every piece below
        const unsigned* fl;
        if constexpr (dpoll) {
          const int pi = lane < 4 * npw ? lane >> 2 : 0, pw = lane & 3;
          int pp = 64 * pw + 63;
#pragma unroll
          for (int ps = 1; ps < NPASS; ++ps)
            if (64 * pw + 63 + ps * 256 < NPAIR) pp = 64 * pw + 63 + ps * 256;
          if (HALFQ) pp = NFULL * 256 + 32 * pw + 31;     // split pass: the quad of the wave's last even lane (thread 64 pw + 62)
          fl = reinterpret_cast<const unsigned*>(h_blk_all + slot_in * slab +
                                                 blk_offset(m0 + pp / NQ, (wave * npw + pi) * 4 * NQ + 4 * (pp % NQ), H));
        } else {
          fl = flags + (size_t)t * flag_step + wave * npw + (lane < npw ? lane : 0);
        }
        const unsigned not_yet = dpoll ? 0xffffffffu : 0u;
        const unsigned long long t_begin = wall_clock64();
        // (data_polls == 2, a test switch: no hint, load straight away -- every step then goes through the re-read path)
        while (!(dpoll && CSN_DPOLL_MODE(a.data_polls) == 2) && !__all(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != not_yet)) {
          __builtin_amdgcn_s_sleep(1);
          if (__hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          if (wall_clock64() - t_begin > kSpinTimeoutTicks) {
            __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      CSN_PSTAMP(0);   // wait for h_{t-1}
      const int base = (int)((slot_in * slab + ((size_t)(m0 >> 4) * kblocks + ks_beg) * 512 + lane * 8) * 2);
      // issue order = consumption order (k-step major), pinned, so the MFMAs of k-step ks wait only for its
      // own 4 loads (counted vmcnt) while the younger ones are still in flight
      // k-blocks of h in flight per wave (registers).  NO instantiation may spill: hipcc (ROCm 7.2) spills a 4-dword
      // fragment as 3 dwords to scratch + 1 "reload reuse" dword to an AGPR and then restores only the three -- found in
      // the H = 512 instantiation (NQ 8, KS 4), whose W_hh fragments came back with a foreign 4th dword (0.8 % errors in
      // h, build-dependent).  Ring depth per instantiation so that every one fits its 512 registers; `make` fails on
      // scratch use in these kernels (tools/check_spills.py).
      constexpr int RING_MAX = (NQ == 8 && KS == 4) ? 2 : ((NQ == 6 && KS == 6 && !DPOLL) ? 3 : CSN_FWD_RING);
      constexpr int RING = KS > RING_MAX ? RING_MAX : KS;
      bf16x8 hf[RING][4];
      // (the rotated k-block walk as ONE running scalar offset: k-block (i + rot) % KS of the i-th group issued; as
      // per-load constants the compiler kept them in SGPRs across the steps)
      int rot_t = rot;
      asm volatile("" : "+s"(rot_t));
      int kbo = rot_t * 1024;
      auto issue_group = [&](int slot) {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) hf[slot][rg] = load_sc1_b128(hsrc, base + rg * kblocks * 1024, kbo);
        kbo = kbo + 1024 == KS * 1024 ? 0 : kbo + 1024;
      };
#pragma unroll
      for (int ks = 0; ks < RING; ++ks) {
        issue_group(ks);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        if (dpoll) {
          // every 8-byte piece (one producer store) of the four fragments is data, not the sentinel; otherwise the
          // watched word was ahead of its neighbours: re-read this k-block until it is whole
          auto whole = [&]() {
            bool ok = true;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const u32x4 u = __builtin_bit_cast(u32x4, hf[ks % RING][rg]);
              ok = ok && u[0] != 0xffffffffu && u[2] != 0xffffffffu;
            }
            return __all(ok);
          };
          if (__builtin_expect(!whole(), 0)) {
            const unsigned long long t_begin = wall_clock64();
            do {
              __builtin_amdgcn_s_sleep(1);
#pragma unroll
              for (int rg = 0; rg < 4; ++rg) hf[ks % RING][rg] = load_sc1_b128(hsrc, base + (rg * kblocks + (ks + rot) % KS) * 1024);
              if (wall_clock64() - t_begin > kSpinTimeoutTicks) {
                __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
              }
            } while (!whole() && __hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u);
          }
        }
#ifdef CSN_SLAB_TAGS
        if (dpoll) {        // h_{t-1} was tagged with bit 2 of t - 1; anything else in a non-sentinel piece is a stale occupant
          const unsigned want = (unsigned)(((t - 1) >> 2) & 1);
          bool stale = false;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const u32x4 u = __builtin_bit_cast(u32x4, hf[ks % RING][rg]);
            stale |= (u[0] != 0xffffffffu && (u[0] & 1u) != want) | (u[2] != 0xffffffffu && (u[2] & 1u) != want);
          }
          if (__any(stale) && lane == 0) __hip_atomic_store(a.error_flag + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
          for (int j = 0; j < NQ; ++j)
            acc[rg][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][j], hf[ks % RING][rg], acc[rg][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (ks + RING < KS) {
          issue_group(ks % RING);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (xwave) {                               // x_t W_ih^T, requested a step ago
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int j = 0; j < NQ; ++j)
          acc[rg][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur[j]), xf[rg], acc[rg][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // next step's input projection (or input): in flight during this step's reduction and epilogue, landed
    // before the next poll
    if (s + 1 < nsteps) request_input(t + 1);
    __builtin_amdgcn_sched_barrier(0);
    CSN_PSTAMP(1);     // h loads + MFMA

#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int j = 0; j < NQ; ++j)
        red[(wave * NT + rg * NQ + j) * kRedTile + lane] = make_float4(acc[rg][j][0], acc[rg][j][1], acc[rg][j][2], acc[rg][j][3]);
    __syncthreads();
    CSN_PSTAMP(2);     // LDS write + barrier

    // Data polls: re-arm a ring slot with the sentinel.  Ring of 4: h_s lives in slot (s + 1) & 3; slot (t + 3) & 3 holds
    // h_{t-2} and will hold h_{t+2}.
    //   safe to overwrite: past the barrier above all four waves have seen the watched word of h_{t-1} from every
    //     producer of the group, and a producer writes h_{t-1} only after it has read h_{t-2}.  (NOT at the top of the
    //     step: a workgroup that has not waited for h_{t-1} yet knows nothing about slower neighbours still reading
    //     h_{t-2} -- found as a time-out of the one-launch-per-layer form, where the skew is larger.)
    //   visible in time: a consumer looks at this slot at step t+3, after it has consumed this workgroup's h_{t+1};
    //     h_{t+1} is stored behind the loads of step t+1, whose return (vmcnt retires in order) means these stores
    //     were acknowledged -- so the sentinel is in L2 before h_{t+1} is even issued, with no wait spent on it.
    //     (A ring of 3 -- re-arming the slot of h_{t+1} -- would leave only issue order between the sentinel and the
    //     h_t the consumer has to see first.)
    //   never over data: h_{t+2} is stored two steps from now, by this same wave.
    if (dpoll && !CSN_DPOLL_NO_REARM(a.data_polls)) {
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        if (HALFQ && ps == NFULL ? hq != 0 : tid + ps * 256 >= NPAIR) continue;      // (rows beyond B are padding rows of the slab: re-armed like the rest; split pass: one lane per quad)
        unsigned long long* sp = reinterpret_cast<unsigned long long*>(
            h_blk_all + (size_t)((t + 3) & 3) * slab + blk_offset(prow[ps], puq[ps], H));
        if (local) *sp = ~0ull;
        else __hip_atomic_store(sp, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __builtin_amdgcn_sched_barrier(0);
    }


#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      if (!pok[ps]) {
        // padding rows of the last M-tile: nobody computes them, but with data polls their pieces must stop being
        // the sentinel (zeros: the rows feed only their own, never stored, outputs)
        if (dpoll && (HALFQ && ps == NFULL ? hq == 0 : tid + ps * 256 < NPAIR)) {
          unsigned long long* zp = reinterpret_cast<unsigned long long*>(h_blk_all + slot_out * slab + blk_offset(prow[ps], puq[ps], H));
#ifdef CSN_SLAB_TAGS
          const unsigned long long zv = (unsigned long long)((t >> 2) & 1);
#else
          const unsigned long long zv = 0ull;
#endif
          if (local) *zp = zv;
          else __hip_atomic_store(zp, zv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        continue;
      }
      const bool hp = HALFQ && ps == NFULL;           // (a constant once the pass loop is unrolled)
      const int nq = hp ? 2 : 4;                         // units this thread finishes in this pass
      const int q0 = hp ? 2 * hq : 0;                    // first of them inside the quad
      const int rl = prl[ps], j = pj[ps], row = prow[ps], uq = puq[ps];
      float gi[4], gf[4], gg[4], go[4], cn[4], hn[4];
      const float cpv[4] = {cst[ps].x, cst[ps].y, cst[ps].z, cst[ps].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (q >= nq) continue;
        const int idx = ((rl >> 4) * NQ + j) * kRedTile + (rl & 15) + 16 * (q0 + q);
        float4 sum = red[idx];
#pragma unroll
        for (int w2 = 1; w2 < 4; ++w2) {
          const float4 v = red[w2 * NT * kRedTile + idx];
          sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        }
        const f32x4 xq = fused ? (f32x4){0.f, 0.f, 0.f, 0.f} : cur[ps * 4 + q];
        gi[q] = fast_sigmoid(sum.x + xq[0]);
        gf[q] = fast_sigmoid(sum.y + xq[1]);
        gg[q] = fast_tanh(sum.z + xq[2]);
        go[q] = fast_sigmoid(sum.w + xq[3]);
        cn[q] = gf[q] * cpv[q] + gi[q] * gg[q];
        hn[q] = go[q] * fast_tanh(cn[q]);
      }
#ifdef CSN_PSTAMPS
      if (ps == 0) CSN_PSTAMP(8);      // (diagnostic: pass 0 LDS reads + gate math)
#endif
      bf16_t* hdst = h_blk_all + slot_out * slab + blk_offset(row, uq, H);
      if (hp) {
        cst[ps] = make_float4(cn[0], cn[1], 0.f, 0.f);
        // the quad's 8-byte hand-off piece: the even lane collects its neighbour's two values (quad_perm [1,0,3,2]) and
        // stores -- one store per piece, as in the full passes (the consumers' proof is per 8-byte piece)
        union { bf16_t b[2]; unsigned u; } pk;
        pk.b[0] = (bf16_t)hn[0];
        pk.b[1] = (bf16_t)hn[1];
        const unsigned other = (unsigned)__builtin_amdgcn_update_dpp(0, (int)pk.u, 0xB1, 0xf, 0xf, false);
        if (hq == 0) {
          unsigned long long piece = (unsigned long long)pk.u | ((unsigned long long)other << 32);
#ifdef CSN_SLAB_TAGS
          piece = (piece & ~1ull) | (unsigned long long)((t >> 2) & 1);
#endif
          if (local) *reinterpret_cast<unsigned long long*>(hdst) = piece;
          else __hip_atomic_store(reinterpret_cast<unsigned long long*>(hdst), piece, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (gates != nullptr) {
          bf16x8 lo = {(bf16_t)gi[0], (bf16_t)gf[0], (bf16_t)gg[0], (bf16_t)go[0], (bf16_t)gi[1], (bf16_t)gf[1], (bf16_t)gg[1], (bf16_t)go[1]};
          __builtin_nontemporal_store(lo, reinterpret_cast<bf16x8*>(gates + ((size_t)t * B + row) * 4 * H + 4 * (size_t)(uq + q0)));
        }
        __builtin_nontemporal_store((f32x2){cn[0], cn[1]}, reinterpret_cast<f32x2*>(c_all + ((size_t)(t + 1) * B + row) * H + uq + q0));
        __builtin_nontemporal_store(pk.u, reinterpret_cast<unsigned*>(h_all + ((size_t)(t + 1) * B + row) * H + uq + q0));
      } else {
        cst[ps] = make_float4(cn[0], cn[1], cn[2], cn[3]);
        // the hand-off payload first: plain stores stay in this XCD's L2 (L2-local groups), write-through otherwise
        if (local) store_plain_b64(hdst, hn, t >> 2);
        else store_wt_b64(hdst, hn, t >> 2);
        if (gates != nullptr) {
          bf16x8 lo = {(bf16_t)gi[0], (bf16_t)gf[0], (bf16_t)gg[0], (bf16_t)go[0], (bf16_t)gi[1], (bf16_t)gf[1], (bf16_t)gg[1], (bf16_t)go[1]};
          bf16x8 hi = {(bf16_t)gi[2], (bf16_t)gf[2], (bf16_t)gg[2], (bf16_t)go[2], (bf16_t)gi[3], (bf16_t)gf[3], (bf16_t)gg[3], (bf16_t)go[3]};
          bf16x8* gp = reinterpret_cast<bf16x8*>(gates + ((size_t)t * B + row) * 4 * H + 4 * (size_t)uq);
          __builtin_nontemporal_store(lo, gp);
          __builtin_nontemporal_store(hi, gp + 1);
        }
        __builtin_nontemporal_store((f32x4){cn[0], cn[1], cn[2], cn[3]},
                                    reinterpret_cast<f32x4*>(c_all + ((size_t)(t + 1) * B + row) * H + uq));
        __builtin_nontemporal_store((bf16x4){(bf16_t)hn[0], (bf16_t)hn[1], (bf16_t)hn[2], (bf16_t)hn[3]},
                                    reinterpret_cast<bf16x4*>(h_all + ((size_t)(t + 1) * B + row) * H + uq));
      }
#ifdef CSN_PSTAMPS
      if (ps == 0) CSN_PSTAMP(9);      // (diagnostic: pass 0 store issue)
#endif
    }
    CSN_PSTAMP(3);     // epilogue (LDS reads, math, store issue)
    // publish: every storing wave drains, workgroup barrier (also frees `red`), one lane signals.
    // (A COUNTED wait -- vmcnt(4): "all but the four saved-tensor stores behind the last hand-off store" -- was 1.7 %
    // faster per launch and passed every test, until a semantically neutral reordering of the prologue made the
    // H = 512 instantiation return stale rows with it and correct ones without it: whatever the cause -- store
    // acknowledgements of different kinds overtaking each other, or compiler-placed spill traffic in the count --
    // the hand-off may not depend on it.  Full drain.)
    if (!dpoll) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CSN_PSTAMP(4);     // drain + barrier
    if (!dpoll && tid == 0) {
      unsigned* fl = flags + (size_t)(t + 1) * flag_step + slice;
      // L2-local groups: a plain store, kept in the one L2 all readers poll; otherwise written through
      if (local) *fl = 1u;
      else __hip_atomic_store(fl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    CSN_PSTAMP(5);     // signal
  }
#ifdef CSN_PSTAMPS
  __syncthreads();
  if (tid == 0 && slice == 0) {
    const unsigned long long now_ = wall_clock64();
    atomicAdd(&g_grp_alive[grp & 7], now_ - t_entry_);
    atomicAdd(&g_grp_alive[8 + (grp & 7)], 1ull);
    atomicAdd(&g_grp_alive[16 + (grp & 7)], now_ - t_loop_);
    atomicAdd(&g_grp_alive[24 + (grp & 7)], (unsigned long long)nsteps);
  }
  span_exit(t_entry_);
#endif
}

// The weight-stationary kernels place one workgroup per CU and need all of a launch co-resident: they are used
// only on a full MI355X (256 CUs; smaller partitions take the per-timestep launches).
static bool device_has_256_cus() {       // asked when a layout is made (plan creation), not per launch
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  return hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 256;
}

bool fwd_persist_supported(int B, int H, int dtype, const Options& opt) {
  if (dtype != CSN_BF16 || H % 128 != 0 || opt.no_persist) return false;
  if (!device_has_256_cus()) return false;
  const int nq = (H % 24 == 0) ? 6 : 8, ks = H / 128;
  const bool shape = (nq == 6 && (ks == 6 || ks == 3)) || (nq == 8 && (ks == 4 || ks == 2 || ks == 1));
  // all workgroups of a launch must be co-resident: one per CU, and two layers run side by side
  const int wgs = (H / (4 * nq)) * ((B + 63) / 64);
  return shape && wgs <= 128;
}

int fwd_persist_slices(int H) { return H / (4 * ((H % 24 == 0) ? 6 : 8)); }

template <int NQ, int KS>
static int launch_persist_t(const PersistFwdArgs& a, hipStream_t st) {
  const size_t lds = (size_t)(4 * 4 * NQ * kRedTile + 4 * NQ) * sizeof(float4);
  if (int rc = ensure_dyn_lds<&lstm_fwd_persist_kernel<NQ, KS, false>>((int)lds)) return rc;
  if (int rc = ensure_dyn_lds<&lstm_fwd_persist_kernel<NQ, KS, true>>((int)lds)) return rc;
  const unsigned nslices = (unsigned)(a.H / (4 * NQ));
  const unsigned grid = a.xcd_groups ? 8u * nslices : nslices * (unsigned)(a.MT * a.nslots);
  if (a.data_polls) lstm_fwd_persist_kernel<NQ, KS, true><<<dim3(grid), 256, lds, st>>>(a);
  else lstm_fwd_persist_kernel<NQ, KS, false><<<dim3(grid), 256, lds, st>>>(a);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

int launch_fwd_persist(const PersistFwdArgs& a, hipStream_t st) {
  CSN_REQUIRE(a.nslots >= 1 && a.nslots <= 4 && a.MT >= 1, "launch_fwd_persist: bad slot count");
  CSN_REQUIRE(fwd_persist_slices(a.H) % 4 == 0 && fwd_persist_slices(a.H) <= kPersistFlagLine,
              "launch_fwd_persist: H=%d gives %d slices (need a multiple of 4, at most %d)", a.H, fwd_persist_slices(a.H),
              kPersistFlagLine);
  if (a.xcd_groups)
    CSN_REQUIRE(a.nslots * a.MT <= 8 && fwd_persist_slices(a.H) <= 32, "launch_fwd_persist: groups do not fit 8 XCDs");
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
extern "C" int csn_debug_read_grp_alive(unsigned long long* out) {
  unsigned long long z[32] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_grp_alive), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_grp_alive), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
extern "C" int csn_debug_read_pstamps(unsigned long long* out) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pstamps), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_pstamps), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif
