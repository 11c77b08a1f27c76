// K3 backward recurrence, weight-stationary form: ONE launch walks up to 4 layers, each backwards through its
// own chunk of timesteps.  Replaces the per-step body of the autograd backward of nn.LSTM reached from
// loss.backward() at /root/reference/LstmDistillFromDinoV2Train.py:127 (model at LSTMDistill.py:118,132).
//
// Per step and layer:  dh_t = dgates_{t+1} W_hh + dy_t,  then the gate derivatives dgates_t [B, 4H] from dh_t,
// the carried dc and the saved forward tensors.  In the per-diagonal launches of lstm_cell_blk.hip every
// workgroup re-reads its W_hh^T slice (295 KB) every step; here a workgroup owns a (64 rows x 16*NUT units)
// tile for a whole chunk, keeps its W_hh^T slice in registers (each wave: its quarter of K' = 4H for all
// 16*NUT units = KS*NUT fragments, 192 VGPRs at H = 768) and the carried dc in registers, and per step
// streams only the 64 rows of dgates_{t+1} (a ring of RING k-blocks in flight per wave).
//
// The step-to-step hand-off of dgates between the workgroups of a group (one layer's 64-row M-tile) is the
// one of lstm_fwd_persist.hip, both forms (L2-local per XCD group, verified at run time; placement-
// independent write-through otherwise), with per-step slabs dg_blk_all[t] that are never reused inside a
// backward, per-wave polling of exactly the producers a wave's K quarter covers, and nothing queued in a
// wave's vector-memory pipe ahead of its polls (the saved tensors of step t-1 are requested behind the
// MFMAs of step t).  All workgroups of a launch must be co-resident (1 per CU); every spin is bounded and
// raises the sticky error flag instead of hanging.
#include "csn_common.h"
#include "lstm_cell_common.h"
#include "lstm_cell_blk.h"
#include "gemm_beside.h"

#ifdef CSN_PSTAMPS
#ifndef CSN_STAMP_BLOCK
#define CSN_STAMP_BLOCK 11     // group 3 (layer 0 at cfg2), slice 1; 15 = group 7 (layer 1)
#endif
__device__ unsigned long long g_bstamps[16];
#define CSN_BSTAMP(i)                                                          \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    if (tid == 0 && blockIdx.x == CSN_STAMP_BLOCK) {                           \
      const unsigned long long now_ = wall_clock64();                          \
      atomicAdd(&g_bstamps[i], now_ - last_);                                  \
      last_ = now_;                                                            \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#else
#define CSN_BSTAMP(i)
#endif

namespace csn {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

static constexpr unsigned long long kBwdSpinTimeoutTicks = 20000000ull;

// Reduction buffer: float4 slot of (batch row r of the 16-row group, unit q of the 4-unit tile) inside an accumulator
// tile.  The accumulator layout would put it at r + 16 q -- but the epilogue's 16-lane groups are 2 consecutive rows x
// 8 unit quads (2 tiles x 4 units), and r + 16 q puts the 4 units of a row on ONE 16-byte bank group (16 float4 = one
// turn of the 64 banks), the two tiles one group apart: eight lanes on the same group.  pi(r) + 17 q with
// pi(r) = r / 2 + 8 (r mod 2) and a tile pitch of 4 mod 16 gives the 16 lanes 16 different groups, and a writing
// 16-lane group (q fixed, r = 0..15) as well.
static constexpr int kBwdRedTile = 68;
__device__ __forceinline__ int bwd_red_pos(int r, int q) { return (r >> 1) + 8 * (r & 1) + 17 * q; }   // 0.2 s of the 100 MHz wall clock

typedef __attribute__((ext_vector_type(4))) unsigned bu32x4;

__device__ __forceinline__ bf16x8 bload_sc1_b128(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, int soff = 0) {
  bu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)byte_off, soff, 16);   // aux 16 = sc1: L1 bypassed
  union { bu32x4 u; bf16x8 b; } cvt;
  cvt.u = v;
  return cvt.b;
}
template <bool WT>
__device__ __forceinline__ void bstore_b128(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, const bf16x8& v, int soff = 0) {
  union { bu32x4 u; bf16x8 b; } cvt;
  cvt.b = v;
  __builtin_amdgcn_raw_buffer_store_b128(cvt.u, rsrc, (int)byte_off, soff, WT ? 16 : 0);   // sc1 = write-through
}

template <int NUT, int KS, bool DPOLL, bool SINGLE = false>
__global__ void __launch_bounds__(256) lstm_bwd_persist_kernel(PersistBwdArgs a) {
#ifndef CSN_BWD_RING
#define CSN_BWD_RING 5
#endif
  constexpr int RING = KS >= 32 ? 3 : (KS < CSN_BWD_RING ? KS : CSN_BWD_RING);   // k-blocks of dgates in flight per wave (3 at H = 1024: register budget)
  constexpr int NT = 4 * NUT;                    // accumulator tiles per wave (4 row groups x NUT unit tiles)
  constexpr int QPR = 4 * NUT;                   // unit quads per row of the tile
  constexpr int NPAIR = 64 * QPR;                // (row, unit-quad) pairs
  constexpr int NPASS = NPAIR / 256;
  static_assert(NPAIR % 256 == 0, "tile must split evenly over the 256 threads");
  extern __shared__ __attribute__((aligned(16))) float4 red[];   // [4][NT][kBwdRedTile]
  const int B = a.B, H = a.H, MT = a.MT, T = a.T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef CSN_PSTAMPS
  const unsigned long long t_entry_ = wall_clock64();
#endif
  const int nslices = H / (16 * NUT);
  int grp, slice;
  if (a.xcd_groups) {
    grp = blockIdx.x & 7;
    slice = blockIdx.x >> 3;
    const int ngroups = a.nslots * MT, gs = a.grid_slices;
    if (grp >= ngroups || slice >= nslices) {
      // no recurrence work for this workgroup: it walks the tiles of the launch's input-gradient GEMMs.  Workers of
      // one XCD (equal blockIdx.x % 8) get consecutive indices, so they work on neighbouring tiles.
      if (a.ngemm > 0) {
        const int idle_here = gs - nslices;
        const unsigned base = grp <= ngroups ? (unsigned)(grp * idle_here)
                                             : (unsigned)(ngroups * idle_here + (grp - ngroups) * gs);
        const unsigned worker = base + (unsigned)(grp < ngroups ? slice - nslices : slice);
        const unsigned nworkers = (unsigned)(ngroups * idle_here + (8 - ngroups) * gs);
        for (int i = 0; i < a.ngemm; ++i) beside_gemm_tiles(a.gemm[i], reinterpret_cast<char*>(red), worker, nworkers);
      }
      return;
    }
  } else {
    grp = blockIdx.x / nslices;
    slice = blockIdx.x % nslices;
  }
  const PersistBwdSlot& S = a.slot[grp / MT];
  const int mt = grp % MT;
  const int u0 = slice * 16 * NUT, m0 = mt * 64;
  const int K = 4 * H, kblocks = K >> 5;
  const int ks_beg = wave * KS;                         // KS = kblocks / 4 k-blocks per wave
  const size_t slab = (size_t)a.Bpad * K;               // elements of one fragment-major dgates slab
  const bf16_t* const wt_blk = S.wt_blk;
  const bf16_t* const gates = S.gates;
  const float* const c_all = S.c_all;
  const float* const dy = S.dy;
  const float* const dy_last = S.dy_last;
  bf16_t* const dgates = S.dgates;
  bf16_t* const dg_blk_all = S.dg_blk_all;
  unsigned* const flags = S.flags + (size_t)mt * kPersistFlagLine;
  const size_t flag_step = (size_t)MT * kPersistFlagLine;
  const int t_hi = S.t_hi, nsteps = S.nsteps;
  // hand-off by DATA (see lstm_fwd_persist.hip): ring of 4 slabs, dgates_s in slot s & 3, unwritten regions hold the
  // all-ones sentinel; no store drain and no flag on the producer side
  constexpr bool dpoll = DPOLL;
  // SINGLE (a.single_copy; experiments library only): one copy of dgates.  Every step has a slab of its own (dgates_s at dg_blk_all + s slab), which
  // the weight- and input-gradient GEMMs read in place (fragment-major A); the row-major copy is not written.  The
  // hand-off protocol is the ring's with `& 3` dropped: a producer arms ITS region of slab t - 2 at step t (the host arms
  // slabs T-1 and T-2), so every argument about the order of arming, publishing and polling carries over unchanged.
  static_assert(!SINGLE || DPOLL, "single-copy mode hands off by data");
  auto slab_of = [](int s) { return SINGLE ? s : (s & 3); };

  // ---- is this group on one XCD?  (see lstm_fwd_persist.hip)
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
        if (__hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
        if (wall_clock64() - t_begin > kBwdSpinTimeoutTicks) {
          __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      ((__attribute__((address_space(3))) int*)(__attribute__((address_space(3))) void*)red)[0] = (int)(((v >> (8 + 6 * xcc)) & 63ull) == (unsigned long long)nslices);
    }
    __syncthreads();
    local = ((__attribute__((address_space(3))) int*)(__attribute__((address_space(3))) void*)red)[0] != 0;    // (LDS address space: no flat_ instruction in these kernels -- FLAT retires out of order)
    __syncthreads();
  }

  // ---- stationary operand: this wave's K' quarter of the workgroup's rows of W_hh^T ----------------
  // The workgroups of a group all stream the same slab; each starts its walk over the k-blocks at its own
  // offset (register i holds k-block (i + rot) % KS), so that at any moment they pull different lines and the
  // requests spread over all L2 channels instead of queueing on the few that hold one 24 KB window.
  const int rot = a.rotate ? (slice * KS) / nslices : 0;
  bf16x8 wreg[KS][NUT];
#pragma unroll
  for (int kb = 0; kb < KS; ++kb) {
    const int kk = (kb + rot) % KS;
#pragma unroll
    for (int ut = 0; ut < NUT; ++ut)
      wreg[kb][ut] = *reinterpret_cast<const bf16x8*>(wt_blk + ((int64_t)((u0 >> 4) + ut) * kblocks + ks_beg + kk) * 512 + lane * 8);
  }

  // ---- the (row, unit-quad) pairs this thread owns: carried dc and the forward's c in registers ----
  int prow[NPASS], puq[NPASS], prl[NPASS], pjq[NPASS];
  bool pok[NPASS];
  float4 dcn[NPASS], cc[NPASS];
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    const int p = tid + ps * 256;
    prl[ps] = p / QPR;
    pjq[ps] = p % QPR;
    prow[ps] = m0 + prl[ps];
    puq[ps] = u0 + 4 * pjq[ps];
    pok[ps] = prow[ps] < B;
    dcn[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
    cc[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pok[ps]) {
      dcn[ps] = *reinterpret_cast<const float4*>(S.dc_carry + (size_t)prow[ps] * H + puq[ps]);
      cc[ps] = *reinterpret_cast<const float4*>(c_all + ((size_t)(t_hi + 1) * B + prow[ps]) * H + puq[ps]);
    }
  }

  // saved tensors of one step, requested a full step early (behind the MFMAs of the step before)
  bf16x8 gt_n[NPASS][2];
  float4 cpv_n[NPASS], dy_n[NPASS];
  auto request_saved = [&](int t) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      cpv_n[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      dy_n[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (pok[ps]) {
        const bf16x8* gp = reinterpret_cast<const bf16x8*>(gates + ((size_t)t * B + prow[ps]) * K + 4 * (size_t)puq[ps]);
        gt_n[ps][0] = nt_load(gp);
        gt_n[ps][1] = nt_load(gp + 1);
        cpv_n[ps] = nt_load(reinterpret_cast<const float4*>(c_all + ((size_t)t * B + prow[ps]) * H + puq[ps]));
        // always ONE unconditional load: a load under a run-time condition makes the compiler wait vmcnt(0) where the
        // paths join, i.e. for every saved tensor just requested, in the middle of the MFMA phase (measured on the
        // top layer, which has no per-step dy: 8.1 instead of 6.0 us per step)
        const float* dsrc = dy != nullptr ? dy + (size_t)t * B * H : ((dy_last != nullptr && t == T - 1) ? dy_last : S.zeros);
        dy_n[ps] = nt_load(reinterpret_cast<const float4*>(dsrc + (size_t)prow[ps] * H + puq[ps]));
      }
    }
  };
  request_saved(t_hi);
  // data polls: ONE buffer resource over the ring of 4 slabs, the slab of a step is a scalar offset -- a resource per
  // slab and use (read, write, re-arm) ran the kernel out of SGPRs; flags: a resource per slab (T slabs can exceed 2 GiB)
  const __amdgpu_buffer_rsrc_t ring_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)dg_blk_all, 0, (int)((size_t)(SINGLE ? T : 4) * slab * 2), 0x00020000);
  const int slab_bytes = (int)(slab * 2);
#ifdef CSN_PSTAMPS
  unsigned long long last_ = wall_clock64();
  if (tid == 0 && blockIdx.x == CSN_STAMP_BLOCK) atomicAdd(&g_bstamps[6], last_ - t_entry_);   // prologue of this launch
#endif

  // data polls: sentinel over this workgroup's region of the slab at byte offset arm_off of the ring / slab array
  auto arm_region = [&](int arm_off) {
    const bf16x8 sent = __builtin_bit_cast(bf16x8, (u32x4){0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      const unsigned o0 = (unsigned)(blk_offset(prow[ps], 4 * (int64_t)puq[ps], K) * 2);
      const unsigned o1 = (unsigned)(blk_offset(prow[ps], 4 * (int64_t)puq[ps] + 8, K) * 2);
      if (local) {
        bstore_b128<false>(ring_rsrc, o0, sent, arm_off);
        bstore_b128<false>(ring_rsrc, o1, sent, arm_off);
      } else {
        bstore_b128<true>(ring_rsrc, o0, sent, arm_off);
        bstore_b128<true>(ring_rsrc, o1, sent, arm_off);
      }
    }
  };
  for (int s = 0; s < nsteps; ++s) {
    const int t = t_hi - s;
    bf16x8 gt[NPASS][2];
    float4 cpv[NPASS], dyv[NPASS];
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      gt[ps][0] = gt_n[ps][0];
      gt[ps][1] = gt_n[ps][1];
      cpv[ps] = cpv_n[ps];
      dyv[ps] = dy_n[ps];
    }

    f32x4 acc[4][NUT];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int ut = 0; ut < NUT; ++ut) acc[rg][ut] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (t < T - 1) {
      // Wait for dgates_{t+1} (slab t+1).  This wave contracts k' in [wave H, (wave+1) H) = the units of the
      // nslices/4 producer slices [wave nslices/4, ...): it polls exactly those flags (sc1 loads) and then
      // loads -- the polling wave is the loading wave, no workgroup barrier.
      {
        const int npw = nslices >> 2;
        // flags: word i of the (t+1, M-tile) line.  Data polls: four words per producer, one from the LAST store
        // instruction of each of its waves (thread 64 w + 63, last pass, second piece) -- a hint that the whole
        // region is there (watching the first word stored made the consumers start early and redo the phase)
        const unsigned* fl;
        if constexpr (dpoll) {
          const int pi = lane < 4 * npw ? lane >> 2 : 0, pw = lane & 3;
          const int pp = 64 * pw + 63 + (NPASS - 1) * 256;                 // (NPAIR is a multiple of 256: every thread has a pair in the last pass)
          const int64_t prow_w = m0 + pp / QPR, pcol_w = 4 * (int64_t)((wave * npw + pi) * 16 * NUT + 4 * (pp % QPR)) + 8;
          fl = reinterpret_cast<const unsigned*>(dg_blk_all + (size_t)slab_of(t + 1) * slab + blk_offset(prow_w, pcol_w, K));
        } else {
          fl = flags + (size_t)(t + 1) * flag_step + wave * npw + (lane < npw ? lane : 0);
        }
        const unsigned not_yet = dpoll ? 0xffffffffu : 0u;
        const unsigned long long t_begin = wall_clock64();
        // (data_polls == 2, a test switch: no hint, load straight away -- every step then goes through the re-read path)
        while (!(dpoll && CSN_DPOLL_MODE(a.data_polls) == 2) && !__all(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != not_yet)) {
          __builtin_amdgcn_s_sleep(1);
          if (__hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          if (wall_clock64() - t_begin > kBwdSpinTimeoutTicks) {
            __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      CSN_BSTAMP(0);   // wait for dgates_{t+1}
      const __amdgpu_buffer_rsrc_t slabs_rsrc = dpoll ? ring_rsrc : __builtin_amdgcn_make_buffer_rsrc(
          (void*)(dg_blk_all + (size_t)(t + 1) * slab), 0, slab_bytes, 0x00020000);
      const int src_off = dpoll ? __builtin_amdgcn_readfirstlane(slab_of(t + 1) * slab_bytes) : 0;
      // (the k-block walk offset re-enters the step as an opaque scalar: left visible as a loop invariant, the compiler
      // keeps the 24 rotated block offsets of every load in SGPRs across the steps and runs out of them)
      int rot_t = rot;
      asm volatile("" : "+s"(rot_t));
      const unsigned base = (unsigned)((((size_t)(m0 >> 4) * kblocks + ks_beg) * 512 + lane * 8) * 2);
      // ring of RING k-blocks: issue order = consumption order (pinned), so the MFMAs of a k-block wait only
      // for its own 4 loads while the next RING-1 k-blocks are in flight.
      // Data polls: every piece that was multiplied must have been data, not the sentinel -- checked ONCE behind the
      // MFMAs (a test + branch per k-block cost 1 us per step here, a running minimum over the loaded words 0.8): if
      // the watched words were ahead of their neighbours, the whole phase is redone.
      const unsigned long long t_phase = wall_clock64();
      bool again = false;
      int redone = 0;          // data polls: phases redone in this step; from the second on the operand is inspected
      bool proven = false;     // every piece of the operand was seen to be data (no sentinel left)
      do {
        if (again) {
          __builtin_amdgcn_s_sleep(1);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int ut = 0; ut < NUT; ++ut) acc[rg][ut] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        bf16x8 df[RING][4];
        // (the rotated k-block walk as ONE running scalar offset: k-block (i + rot) % KS of the i-th group issued)
        int kbo = rot_t * 1024;
        auto issue_group = [&](int slot) {
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
            df[slot][rg] = bload_sc1_b128(slabs_rsrc, base + (unsigned)(rg * kblocks) * 1024u, src_off + kbo);
          kbo = kbo + 1024 == KS * 1024 ? 0 : kbo + 1024;
        };
#pragma unroll
        for (int kb = 0; kb < RING; ++kb) {
          issue_group(kb);
          __builtin_amdgcn_sched_barrier(0);
        }
#ifdef CSN_SLAB_TAGS
        bool stale = false;       // (debug library, see lstm_fwd_persist.hip: a non-sentinel piece with the wrong step tag)
#ifdef CSN_SLAB_TAGS_DUMP         // where the first stale piece was seen (CSN_TAGS_VERBOSE): a dozen values kept live across the unrolled
        int stale_kb = -1;        // k-block loop -- with them the H = 768 / 1024 instantiations of the tags build spilled 488 / 868
        unsigned stale_u0 = 0;    // bytes per lane, which the spill gate (now applied to that build too) does not allow; opt-in
#endif
#endif
#pragma unroll
        for (int kb = 0; kb < KS; ++kb) {
#ifdef CSN_SLAB_TAGS
          if constexpr (dpoll) {
            const unsigned want = (unsigned)(((t + 1) >> 2) & 1);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
              const u32x4 u = __builtin_bit_cast(u32x4, df[kb % RING][rg]);
              const bool bad = u[0] != 0xffffffffu && (u[0] & 1u) != want;
#ifdef CSN_SLAB_TAGS_DUMP
              if (bad && stale_kb < 0) { stale_kb = kb * 4 + rg; stale_u0 = u[0]; }
#endif
              stale |= bad;
            }
          }
#endif
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int ut = 0; ut < NUT; ++ut)   // D[row = unit][col = batch row]
              acc[rg][ut] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[kb][ut], df[kb % RING][rg], acc[rg][ut], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (kb + RING < KS) {
            issue_group(kb % RING);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#ifdef CSN_SLAB_TAGS
        if (__any(stale) && lane == 0) __hip_atomic_store(a.error_flag + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef CSN_SLAB_TAGS_DUMP
        if (stale) {     // first detection of the workspace: where it was (read back by csn_lstm_status_read of the debug library)
          if (atomicCAS(a.error_flag + 7, 0u, 1u) == 0u) {
            unsigned* dbg = a.error_flag + 8;
            dbg[0] = (unsigned)t; dbg[1] = (unsigned)s; dbg[2] = (unsigned)grp; dbg[3] = (unsigned)slice; dbg[4] = (unsigned)wave;
            dbg[5] = (unsigned)lane; dbg[6] = (unsigned)stale_kb; dbg[7] = stale_u0; dbg[8] = (unsigned)local; dbg[9] = (unsigned)redone;
            dbg[10] = (unsigned)SINGLE; dbg[11] = (unsigned)(((t + 1) >> 2) & 1); dbg[12] = (unsigned)rot_t; dbg[13] = (unsigned)nsteps;
            dbg[14] = __builtin_amdgcn_s_getreg(6164) & 7u; dbg[15] = (unsigned)__popcll(__ballot(stale));
          }
        }
#endif
#endif
        again = false;
        if constexpr (dpoll) {
          // the sentinel is a pair of bf16 NaNs: a piece that was still the sentinel when it was multiplied has
          // poisoned every accumulator element of its batch row -- one sum over the 32 accumulator registers finds it
          float chk = 0.f;
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int ut = 0; ut < NUT; ++ut) chk += (acc[rg][ut][0] + acc[rg][ut][1]) + (acc[rg][ut][2] + acc[rg][ut][3]);
          if (__builtin_expect(!__all(chk == chk), 0)) {
#ifdef CSN_PSTAMPS
            if (lane == 0) atomicAdd(&g_bstamps[8 + wave], 1ull);        // (diagnostic: phases redone, per wave, all workgroups)
#endif
            if (proven) {
              // the operand was all data and the product is still not finite: a genuine non-finite gradient (NaN / Inf
              // input, Inf - Inf in the accumulators).  The reference propagates it; so do we -- no spin, and a status
              // bit of its own (word 1) so that the host can tell it from a hand-off that timed out (word 0)
              if (lane == 0) __hip_atomic_store(a.error_flag + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
              if (++redone >= 2) {
                // cold path: re-read every piece this wave multiplies and look for the sentinel by bit pattern (a piece
                // is ONE 16-byte producer store; data NaNs are canonical 0x7fc0 / 0xffc0 patterns, never all ones)
                bool sentinel = false;
                int kbo2 = rot_t * 1024;
#pragma unroll 1
                for (int kb = 0; kb < KS; ++kb) {
#pragma unroll
                  for (int rg = 0; rg < 4; ++rg) {
                    const u32x4 u = __builtin_bit_cast(u32x4, bload_sc1_b128(slabs_rsrc, base + (unsigned)(rg * kblocks) * 1024u, src_off + kbo2));
                    sentinel |= (u[0] == 0xffffffffu) | (u[3] == 0xffffffffu);
                  }
                  kbo2 = kbo2 + 1024 == KS * 1024 ? 0 : kbo2 + 1024;
                }
                proven = !__any(sentinel);     // a written piece stays written until this workgroup publishes dgates_t
              }
              again = __hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
              if (wall_clock64() - t_phase > kBwdSpinTimeoutTicks) {
                __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                again = false;
              }
            }
          }
        }
      } while (again);
    }
    // the next step's saved tensors: in flight during this step's reduction and epilogue
    if (s + 1 < nsteps) request_saved(t - 1);
    __builtin_amdgcn_sched_barrier(0);
    CSN_BSTAMP(1);     // dgates loads + MFMA

    // lane holds batch row (lane & 15) of row group rg, units 16 ut + (lane >> 4) * 4 + r
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int ut = 0; ut < NUT; ++ut)
        red[(wave * NT + rg * NUT + ut) * kBwdRedTile + bwd_red_pos(lane & 15, lane >> 4)] = make_float4(acc[rg][ut][0], acc[rg][ut][1], acc[rg][ut][2], acc[rg][ut][3]);
    __syncthreads();
    CSN_BSTAMP(2);     // LDS write + barrier

    // data polls: re-arm slot (t - 2) & 3 (it holds dgates_{t+2}: every producer has published dgates_{t+1}, so all of
    // them have read it; it is looked at again at step t-3, after this workgroup's dgates_{t-1} was consumed, which
    // is stored behind loads that retire these stores -- the argument of lstm_fwd_persist.hip, mirrored in time)
    if (dpoll && !CSN_DPOLL_NO_REARM(a.data_polls) && (!SINGLE || t >= 2)) {
      arm_region(__builtin_amdgcn_readfirstlane((SINGLE ? t - 2 : ((t + 2) & 3)) * slab_bytes));
      __builtin_amdgcn_sched_barrier(0);
    }

    const __amdgpu_buffer_rsrc_t slabs_rsrc = dpoll ? ring_rsrc : __builtin_amdgcn_make_buffer_rsrc(
        (void*)(dg_blk_all + (size_t)t * slab), 0, slab_bytes, 0x00020000);
    const int dst_off = dpoll ? __builtin_amdgcn_readfirstlane(slab_of(t) * slab_bytes) : 0;
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
      if (!pok[ps]) {
        if (dpoll) {        // padding rows of the last M-tile: zeros instead of the sentinel (they feed only their own outputs)
#ifdef CSN_SLAB_TAGS
          const unsigned ztag = (unsigned)((t >> 2) & 1);
#else
          const unsigned ztag = 0u;
#endif
          const bf16x8 z = __builtin_bit_cast(bf16x8, (u32x4){ztag, 0u, 0u, 0u});
          const unsigned z0 = (unsigned)(blk_offset(prow[ps], 4 * (int64_t)puq[ps], K) * 2);
          const unsigned z1 = (unsigned)(blk_offset(prow[ps], 4 * (int64_t)puq[ps] + 8, K) * 2);
          if (local) {
            bstore_b128<false>(slabs_rsrc, z0, z, dst_off);
            bstore_b128<false>(slabs_rsrc, z1, z, dst_off);
          } else {
            bstore_b128<true>(slabs_rsrc, z0, z, dst_off);
            bstore_b128<true>(slabs_rsrc, z1, z, dst_off);
          }
        }
        continue;
      }
      const int rl = prl[ps], jq = pjq[ps], row = prow[ps], uq = puq[ps];
      const int idx = ((rl >> 4) * NUT + (jq >> 2)) * kBwdRedTile + bwd_red_pos(rl & 15, jq & 3);
      float4 sm = red[idx];
#pragma unroll
      for (int w2 = 1; w2 < 4; ++w2) {
        const float4 v = red[w2 * NT * kBwdRedTile + idx];
        sm.x += v.x; sm.y += v.y; sm.z += v.z; sm.w += v.w;
      }
      const float dh[4] = {sm.x + dyv[ps].x, sm.y + dyv[ps].y, sm.z + dyv[ps].z, sm.w + dyv[ps].w};
      const float cv[4] = {cc[ps].x, cc[ps].y, cc[ps].z, cc[ps].w};
      const float cpr[4] = {cpv[ps].x, cpv[ps].y, cpv[ps].z, cpv[ps].w};
      const float dcv[4] = {dcn[ps].x, dcn[ps].y, dcn[ps].z, dcn[ps].w};
      float out[16], dcarry[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bf16x8& g8 = gt[ps][q >> 1];
        const float gi = (float)g8[(q & 1) * 4 + 0], gf = (float)g8[(q & 1) * 4 + 1];
        const float gg = (float)g8[(q & 1) * 4 + 2], go = (float)g8[(q & 1) * 4 + 3];
        const float tc = fast_tanh(cv[q]);
        const float d_o = dh[q] * tc;
        const float dc = dh[q] * go * (1.0f - tc * tc) + dcv[q];
        out[4 * q + 0] = dc * gg * gi * (1.0f - gi);
        out[4 * q + 1] = dc * cpr[q] * gf * (1.0f - gf);
        out[4 * q + 2] = dc * gi * (1.0f - gg * gg);
        out[4 * q + 3] = d_o * go * (1.0f - go);
        dcarry[q] = dc * gf;
      }
      bf16x8 lo, hi;
#pragma unroll
      for (int e = 0; e < 8; ++e) { lo[e] = (bf16_t)out[e]; hi[e] = (bf16_t)out[8 + e]; }
      // the hand-off payload first: plain stores stay in this XCD's L2 (L2-local groups), write-through otherwise
      const unsigned o0 = (unsigned)(blk_offset(row, 4 * (int64_t)uq, K) * 2);
      const unsigned o1 = (unsigned)(blk_offset(row, 4 * (int64_t)uq + 8, K) * 2);
#ifdef CSN_SLAB_TAGS      // each 16-byte hand-off piece carries bit 2 of its step in the lowest mantissa bit of its first element
      {
        u32x4 ul = __builtin_bit_cast(u32x4, lo), uh = __builtin_bit_cast(u32x4, hi);
        const unsigned tag = (unsigned)((t >> 2) & 1);
        ul[0] = (ul[0] & ~1u) | tag;
        uh[0] = (uh[0] & ~1u) | tag;
        lo = __builtin_bit_cast(bf16x8, ul);
        hi = __builtin_bit_cast(bf16x8, uh);
      }
#endif
      if (local) {
        bstore_b128<false>(slabs_rsrc, o0, lo, dst_off);
        bstore_b128<false>(slabs_rsrc, o1, hi, dst_off);
      } else {
        bstore_b128<true>(slabs_rsrc, o0, lo, dst_off);
        bstore_b128<true>(slabs_rsrc, o1, hi, dst_off);
      }
      if constexpr (!SINGLE) {
        bf16x8* op = reinterpret_cast<bf16x8*>(dgates + ((size_t)t * B + row) * K + 4 * (size_t)uq);
        nt_store(op, lo);
        nt_store(op + 1, hi);
      }
      dcn[ps] = make_float4(dcarry[0], dcarry[1], dcarry[2], dcarry[3]);
      cc[ps] = cpv[ps];                       // c_{t-1} is the next step's c
    }
    CSN_BSTAMP(3);     // epilogue
    if (!dpoll) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (full drain, not a counted one: see lstm_fwd_persist.hip)
    __syncthreads();
    CSN_BSTAMP(4);     // drain + barrier
    if (!dpoll && tid == 0) {
      unsigned* fl = flags + (size_t)t * flag_step + slice;
      if (local) *fl = 1u;
      else __hip_atomic_store(fl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    CSN_BSTAMP(5);     // signal
  }

#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps)
    if (pok[ps]) *reinterpret_cast<float4*>(S.dc_carry + (size_t)prow[ps] * H + puq[ps]) = dcn[ps];
}

bool bwd_persist_supported(int B, int H, int dtype, const Options& opt) {
  if (dtype != CSN_BF16 || opt.no_persist || opt.no_persist_bwd) return false;
  return H == 128 || H == 256 || H == 384 || H == 512 || H == 768 || H == 1024;
}
int bwd_persist_slices(int H) { return H / 32; }

template <int NUT, int KS>
static int launch_bwd_persist_t(const PersistBwdArgs& a, hipStream_t st) {
  size_t lds = (size_t)4 * 4 * NUT * kBwdRedTile * sizeof(float4);
  if (int rc = ensure_dyn_lds<&lstm_bwd_persist_kernel<NUT, KS, false>>((int)kBesideLdsBytes + 64)) return rc;
  if (int rc = ensure_dyn_lds<&lstm_bwd_persist_kernel<NUT, KS, true>>((int)kBesideLdsBytes + 64)) return rc;
#ifdef CSN_EXPERIMENTS
  if (int rc = ensure_dyn_lds<&lstm_bwd_persist_kernel<NUT, KS, true, true>>((int)kBesideLdsBytes + 64)) return rc;
#endif
  const unsigned nslices = (unsigned)(a.H / (16 * NUT));
  PersistBwdArgs b = a;
  if (b.xcd_groups) {
    if (b.ngemm > 0) {
      lds = kBesideLdsBytes + 64;            // the GEMM workers' staging ring (+ the claim word); also keeps every workgroup alone on its CU
      if (b.grid_slices < (int)nslices) b.grid_slices = (int)nslices;
    } else {
      b.grid_slices = (int)nslices;
    }
  }
  const unsigned grid = b.xcd_groups ? 8u * (unsigned)b.grid_slices : nslices * (unsigned)(b.MT * b.nslots);
#ifdef CSN_EXPERIMENTS
  if (b.data_polls && b.single_copy) lstm_bwd_persist_kernel<NUT, KS, true, true><<<dim3(grid), 256, lds, st>>>(b);
  else
#endif
  if (b.data_polls) lstm_bwd_persist_kernel<NUT, KS, true><<<dim3(grid), 256, lds, st>>>(b);
  else lstm_bwd_persist_kernel<NUT, KS, false><<<dim3(grid), 256, lds, st>>>(b);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

int launch_bwd_persist(const PersistBwdArgs& a, hipStream_t st) {
  CSN_REQUIRE(a.nslots >= 1 && a.nslots <= 4 && a.MT >= 1, "launch_bwd_persist: bad slot count");
  const int ns = bwd_persist_slices(a.H);
  CSN_REQUIRE(ns % 4 == 0 && ns <= kPersistFlagLine, "launch_bwd_persist: H=%d gives %d slices", a.H, ns);
  if (a.xcd_groups) CSN_REQUIRE(a.nslots * a.MT <= 8, "launch_bwd_persist: groups do not fit 8 XCDs");
  CSN_REQUIRE(a.ngemm >= 0 && a.ngemm <= 3 && (a.ngemm == 0 || a.xcd_groups), "launch_bwd_persist: bad GEMM list");
#ifndef CSN_EXPERIMENTS
  CSN_REQUIRE(!a.single_copy, "launch_bwd_persist: the single-copy form lives in the experiments library");
#endif
  CSN_REQUIRE(!a.single_copy || (a.data_polls && a.B == a.Bpad && (int64_t)a.T * a.Bpad * a.H * 8 < (int64_t)1 << 31),
              "launch_bwd_persist: single-copy mode needs data polls, B %% 64 == 0 and T slabs below 2 GiB");
  for (int i = 0; i < a.ngemm; ++i)
    CSN_REQUIRE(a.gemm[i].K % 64 == 0 && a.gemm[i].N % 4 == 0 && a.gemm[i].M > 0, "launch_bwd_persist: GEMM %d shape", i);
  switch (a.H) {
    case 1024: return launch_bwd_persist_t<2, 32>(a, st);
    case 768: return launch_bwd_persist_t<2, 24>(a, st);
    case 512: return launch_bwd_persist_t<2, 16>(a, st);
    case 384: return launch_bwd_persist_t<2, 12>(a, st);
    case 256: return launch_bwd_persist_t<2, 8>(a, st);
    case 128: return launch_bwd_persist_t<2, 4>(a, st);
  }
  return fail(CSN_ERR_UNSUPPORTED, "launch_bwd_persist: no kernel for H=%d", a.H);
}

}  // namespace csn

#ifdef CSN_PSTAMPS
extern "C" int csn_debug_read_bstamps(unsigned long long* out) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bstamps), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_bstamps), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif
