// K3 forward, weight-stationary, N-SPLIT form: the forward recurrence kernel for H = 1024 (cfg4), and -- behind
// CSN_FWD_NSPLIT -- an alternative to the K-split kernel of lstm_fwd_persist.hip for the other widths.
// Replaces the per-step body of nn.LSTM reached at /root/reference/LSTMDistill.py:118,132 (same contract, same
// workspace, same hand-off protocol as lstm_fwd_persist.hip, whose K-split kernel stays as the cross-check).
//
// What the K-split kernel paid per step at cfg2 (in-kernel stamps, profiles/): 0.5 us of LDS reduction over the 4
// K-quarter partial tiles (100 KB written + read), a 1.8 us epilogue on a (row, unit-quad) thread mapping that leaves
// a third of the threads idle in its second pass, and -- a matter of register budget -- no kernel at all for H = 1024.
// Here a workgroup owns 64 batch rows x 32 units (128 interleaved gate rows = 8 MFMA tiles) and the waves split
// the GATE ROWS, not K: wave w holds tiles 2w, 2w+1 of W_hh for ALL of K in registers (8 * H/32 VGPRs: 192 at
// H = 768, 256 at H = 1024) and accumulates the complete pre-activations of its 64 x 8 cells in 32 registers.
//   * No partial sums, no LDS reduction: the accumulator layout of v_mfma_f32_16x16x32_bf16 with interleaved gate
//     rows puts (i, f, g, o) of ONE cell in the 4 registers of one lane, so the gate math runs in place, balanced
//     over all 256 lanes (8 cells each), the cell state c stays in registers in the same layout.
//   * Every wave needs the whole h_{t-1} tile (64 rows x H): it is brought in ONCE per workgroup and step by LDS-DMA
//     (global_load_lds, sc1), 1 KB fragment blocks that already have the operand layout, and read by all 4 waves
//     with conflict-free ds_read_b128.  Same L2 -> CU bytes per step as before (the per-CU L2 port is what bounds
//     the operand stream), 4x the LDS reads, which overlap the MFMAs.
//   * The DMA is issued in 4 groups of K-blocks with counted vmcnt waits: the MFMAs of group g run while the groups
//     behind it are still landing; the next step's input (projection, or x for the fused layer 0) is requested
//     behind the DMAs, never in front of a poll.
//   * Results leave through a 32 KB workgroup-wide LDS transpose so that every global store is a 16-byte-per-lane
//     store in which consecutive lanes write consecutive bytes (a lane-per-row mapping -- 64 rows, 16 bytes each per
//     instruction -- was measured first: 2.2 us of store issue + 1.9 us of drain per step)
// Barriers per step: one per DMA group + one inside the transpose + one before the flag.  Hand-off, flags, XCD agreement, bounded spins:
// identical to lstm_fwd_persist.hip (L2-local inside an XCD-resident group, verified per launch; placement-
// independent write-through otherwise).
//
// 32 units per workgroup: H/32 workgroups per hand-off group (24 at H = 768, leaving 8 CUs per XCD to the layer-1
// input-projection GEMM carried by the launch, see gemm_beside.h; 32 at H = 1024).
#include <algorithm>

#include "csn_common.h"
#include "lstm_cell_common.h"
#include "lstm_cell_blk.h"
#include "gemm_beside.h"
#include "lstm_ns_util.h"

#ifdef CSN_PSTAMPS
#ifndef CSN_STAMP_BLOCK
#define CSN_STAMP_BLOCK 11
#endif
__device__ unsigned long long g_nstamps[16];
#define CSN_NSTAMP(i)                                                          \
  do {                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                         \
    if (tid == 0 && blockIdx.x == CSN_STAMP_BLOCK) {                           \
      const unsigned long long now_ = wall_clock64();                          \
      atomicAdd(&g_nstamps[i], now_ - last_);                                  \
      last_ = now_;                                                            \
    }                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                         \
  } while (0)
#else
#define CSN_NSTAMP(i)
#endif

namespace csn {

// The recurrence of one workgroup over its chunk.  FUSED (compile time): layer 0 multiplies x_t itself.  The two
// forms are separate instantiations called from one kernel (a launch advances layer 0 AND the layers above it):
// merged into one body with a run-time flag, the two meanings of the input registers made the compiler wait for
// the next step's input -- and with it for every DMA queued before it -- right after requesting it.
template <int KB, bool FUSED>
__device__ __forceinline__ void ns_recurrence(const PersistFwdArgs& a, const PersistFwdSlot& S, char* smem, int slice,
                                              int mt, bool local) {
  constexpr int GI = KB / 4;                       // k-blocks per DMA group = DMA instructions per wave and group
  constexpr bool AIA = KB <= 24;                   // accumulators in AGPRs beside the weights (ns_mfma)
  // H = 1024: the h tile (128 KB per step) comes in by LDS-DMA, all 32 pieces of a wave in flight at once.  Through registers
  // (the form below, kept for H <= 768) this width has staging registers for ONE group of 8 k-blocks: 32 KB in flight per
  // CU, i.e. four serial round trips to L2 per step -- in-kernel stamps: 5.2 us for the h stream + 1.8 us of MFMAs.  A DMA
  // piece costs ~100 cycles of issue (which is why the narrower widths, whose registers hold two groups, stay with
  // register loads), but needs no register and no ds_write, so nothing limits the bytes in flight (round 4)
#if defined(CSN_NS_NO_HDMA)      // timing A/B (tools/abl_build.sh)
  constexpr bool HDMA = false;
#else
  constexpr bool HDMA = KB >= 32;
#endif
  constexpr int P = FUSED ? 16 : 8;                // 16-byte registers of one input request
  static_assert(KB % 4 == 0, "4 load groups");
  const int B = a.B, H = a.H, MT = a.MT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef CSN_PSTAMPS
  unsigned long long last_ = wall_clock64();
#endif
  const int nslices = H >> 5;
  const int u0 = slice * 32, m0 = mt * 64;
  const size_t slab = (size_t)a.Bpad * H;              // elements of one fragment-major h slab
  bf16_t* const gates = S.gates;
  float* const c_all = S.c_all;
  bf16_t* const h_all = S.h_all;
  unsigned* const flags = S.flags + (size_t)mt * kPersistFlagLine;    // [T+1][MT][line]: word i = slice i has published
  const size_t flag_step = (size_t)MT * kPersistFlagLine;
  const int t_first = S.t0, nsteps = S.nsteps;
  constexpr int xkb = 4;                               // fused form: I = 128 input features (4 k-blocks), checked by the launcher
  char* const stage = smem + (size_t)KB * 4096;
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // ---- stationary operands: this wave's two gate-row tiles of W_hh for all of K.  Register p holds k-block
  // (p + rot) % KB: the workgroups of a group all pull the same slab and each walks it from its own offset, so that
  // at any moment they are on different lines (lstm_bwd_persist.hip).
  // (LDS-DMA form: rotation in steps of 4 k-blocks, so that four consecutive walk positions are four consecutive KB of the
  // slab -- one M0 / one scalar offset per four pieces, the rest in the instruction offset -- and never straddle the wrap)
  const int rot = a.rotate ? (__builtin_amdgcn_readfirstlane((slice * KB) / nslices) & (HDMA ? ~3 : ~0)) : 0;
  const int tile0 = (u0 >> 2) + 2 * wave;              // first of this wave's two 16-row tiles of the interleaved 4H axis
  bf16x8 wreg[KB][2];
#pragma unroll
  for (int p = 0; p < KB; ++p) {
    int kb = p + rot;
    kb = kb >= KB ? kb - KB : kb;
#pragma unroll
    for (int j = 0; j < 2; ++j)
      wreg[p][j] = *reinterpret_cast<const bf16x8*>(S.w_blk + ((int64_t)(tile0 + j) * KB + kb) * 512 + lane * 8);
  }
  bf16x8 wih[FUSED ? 4 : 1][2];
  // (the bias of the fused layer 0 lives in LDS, in the 32 pad bytes of the gates rows of the transpose area -- entry e
  // = (wave, tile j, lane >> 4) at row e: as registers it was 8 more than the H = 1024 instantiation has, and a spilling
  // instantiation is not acceptable, see lstm_fwd_persist.hip)
  char* const bias_lds = stage + 256 + (size_t)((2 * wave) * 4 + (lane >> 4)) * 288;      // + 4 * 288 for tile 1
  if constexpr (FUSED) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        wih[kb][j] = *reinterpret_cast<const bf16x8*>(S.wih_blk + ((int64_t)(tile0 + j) * xkb + kb) * 512 + lane * 8);
    // accumulator layout: lane holds gate rows 4 (lane >> 4) .. +3 of a tile = (i, f, g, o) of unit (lane >> 4)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if ((lane & 15) == 0)
        *reinterpret_cast<f32x4*>(bias_lds + j * 4 * 288) =
            *reinterpret_cast<const f32x4*>(S.bias + 16 * (size_t)(tile0 + j) + 4 * (lane >> 4));
    __builtin_amdgcn_wave_barrier();         // written and read by this wave only
  }

  // ---- the 8 cells of this lane: rows m0 + 16 rg + (lane & 15), units u0 + 8 wave + 4 j + (lane >> 4)
  const int unit_q = u0 + 8 * wave + (lane >> 4);          // + 4 j
  int rowc[4];                                             // row, clamped into the batch (padding rows compute on row B-1's inputs)
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    const int r = m0 + 16 * rg + (lane & 15);
    rowc[rg] = r < B ? r : B - 1;
  }
  // The cell state of the lane's 8 cells: registers -- except at H = 1024 (KB = 32: 256 W_hh registers), where it stays
  // in the c rows of the transpose area between the steps (each lane reads back, before the gate math, exactly the
  // words it wrote a step ago; nothing else writes them): the 8 registers are the difference between fitting and spilling
  constexpr bool CLDS = KB == 32;
  float cst[4][2];
#pragma unroll
  for (int rg = 0; rg < 4; ++rg)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float c0 = t_first > 0 ? c_all[((size_t)t_first * B + rowc[rg]) * H + unit_q + 4 * j] : 0.0f;
      if constexpr (CLDS)
        *reinterpret_cast<float*>(stage + 64 * 288 + (rg * 16 + (lane & 15)) * 144 + (8 * wave + 4 * j + (lane >> 4)) * 4) = c0;
      else
        cst[rg][j] = c0;
    }

  // next step's input, requested a step early (behind the DMAs): plain layers the projection of the 8 cells
  // (8 x 16 B), the fused layer 0 the x fragments of all 64 rows (16 x 16 B).  Buffer loads: the block offsets are
  // wave-uniform (SGPRs), the per-lane part is one register per row group.
  f32x4 nxt[P];
  const bool xbf = !FUSED && __builtin_amdgcn_readfirstlane(S.xproj_bf16) != 0;
  const unsigned xslab = FUSED ? (unsigned)a.Bpad * (unsigned)S.I : 0u;
  // Buffer resources are based at THIS LAUNCH's first step and sized for its steps: byte offsets inside them are 32-bit
  // (the float32 projection is 4 MB per step at B = 256, H = 1024 -- 4 GiB, where offsets from step 0 wrapped, at T = 1024)
  const size_t in_step_bytes = FUSED ? (size_t)xslab * 2 : (size_t)B * H * (xbf ? 8 : 16);
  const __amdgpu_buffer_rsrc_t in_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)ns_uniform((FUSED ? (const char*)S.x_blk : (const char*)S.xproj) + (size_t)t_first * in_step_bytes), 0,
      __builtin_amdgcn_readfirstlane((int)((size_t)nsteps * in_step_bytes)), 0x00020000);
  int xvoff[4];                                            // plain: byte offset of (row, first unit) inside a step's [B, 4H] f32 slab
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) xvoff[rg] = (int)(((size_t)rowc[rg] * 4 * H + 4 * (size_t)unit_q) * 4);
  auto request_input = [&](int t) {
    if constexpr (FUSED) {
      // the 64 rows x 128 features of x_t are 16 contiguous 1 KB fragment blocks: one base, constant offsets
      const int sbase = __builtin_amdgcn_readfirstlane((int)(((size_t)(t - t_first) * xslab + (size_t)(m0 >> 4) * xkb * 512) * 2));
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) nxt[rg * 4 + kb] = ns_bload_nt_f32x4(in_rsrc, lane * 16 + (rg * 4 + kb) * 1024, sbase);
    } else if (xbf) {
      // bf16 projection: 8 bytes per cell, widened on arrival
      const int sbase = __builtin_amdgcn_readfirstlane((int)((size_t)(t - t_first) * B * 8 * H));
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const nu32x2 v = __builtin_amdgcn_raw_buffer_load_b64(in_rsrc, (xvoff[rg] >> 1) + j * 32, sbase, 2);
          nxt[rg * 2 + j] = (f32x4){__builtin_bit_cast(float, v[0] << 16), __builtin_bit_cast(float, v[0] & 0xffff0000u),
                                    __builtin_bit_cast(float, v[1] << 16), __builtin_bit_cast(float, v[1] & 0xffff0000u)};
        }
    } else {
      const int sbase = __builtin_amdgcn_readfirstlane((int)((size_t)(t - t_first) * B * 16 * H));
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int j = 0; j < 2; ++j) nxt[rg * 2 + j] = ns_bload_nt_f32x4(in_rsrc, xvoff[rg] + j * 64, sbase);
    }
  };
  if constexpr (FUSED && HDMA) {
    // The x tile of a step (64 rows x 128 features, 16 KB) is the same for all workgroups of the group and comes from HBM:
    // requested a step ahead it still arrived 0.6 us late (in-kernel stamps: the x MFMAs waited, and with them every h
    // piece queued behind the x loads).  Workgroup i therefore pulls the tile of step t_first + i into the group's L2 now
    // (default cache policy, LDS-DMA into the idle h tile: no registers; the data itself is not used), so that the
    // requests of the steps hit L2.
    if (slice < nsteps) {
      const int pbase = __builtin_amdgcn_readfirstlane((int)((((size_t)slice * xslab + (size_t)(m0 >> 4) * xkb * 512) * 2) + (size_t)wave * 4096));
      char* const dst = smem + (size_t)wave * 4096;
      ns_dma16_sc1<0, 0>(in_rsrc, dst, lane * 16, pbase);
      ns_dma16_sc1<1024, 0>(in_rsrc, dst, lane * 16, pbase);
      ns_dma16_sc1<2048, 0>(in_rsrc, dst, lane * 16, pbase);
      ns_dma16_sc1<3072, 0>(in_rsrc, dst, lane * 16, pbase);
    }
  }
  request_input(t_first);

  const __amdgpu_buffer_rsrc_t hdst_rsrc =        // (slabs t_first .. t_first + nsteps: read h_{t-1} from slab t, write h_t to slab t + 1)
      __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.h_blk_all + (size_t)t_first * slab), 0,
                                        __builtin_amdgcn_readfirstlane((int)((size_t)(nsteps + 1) * slab * 2)), 0x00020000);

  // Fused layer 0: the sum of a step starts from bias + x_t W_ih^T, and that part does not depend on h_{t-1} -- it is computed
  // at the END of the step before, behind the publish, i.e. in the time the workgroup would spend waiting for its neighbours'
  // h anyway (round 4: in-kernel stamps had the x MFMAs and their wait on the critical path of the slower layer of every
  // launch).  Same terms in the same order as before (bias, x k-blocks 0..3, then h): same bits.
  f32x4 acc[4][2];
  auto start_from_x = [&]() {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[rg][j] = *reinterpret_cast<const f32x4*>(bias_lds + j * 4 * 288);
    ns_mfma_fence();
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int j = 0; j < 2; ++j) ns_mfma<AIA, !AIA>(acc[rg][j], wih[FUSED ? kb : 0][j], __builtin_bit_cast(bf16x8, nxt[FUSED ? rg * 4 + kb : 0]));
    ns_mfma_fence();
    __builtin_amdgcn_sched_barrier(0);
  };
  if constexpr (FUSED) start_from_x();        // (the launch's first step: the one place that waits for x)

  for (int s = 0; s < nsteps; ++s) {
    const int t = t_first + s;
    if constexpr (!FUSED) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[rg][j] = nxt[rg * 2 + j];      // plain: the sum starts from the input projection
      ns_mfma_fence();
    }

    // LDS address of this lane's 16 bytes in block 0; the opaque zero is re-made every step so that the 4 KB fragment
    // addresses derived from it stay one add each instead of being hoisted into 4 KB registers for good
    unsigned zero_;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_));
    const unsigned hbase = lds_base + (unsigned)lane * 16u + zero_;
    // the h tile comes in through REGISTERS: sc1 buffer loads (the form MI355X_MICROARCH.md's hand-off table is
    // measured for) of this wave's row group, 16-byte ds_writes into the tile, in 4 groups of GI k-blocks with two
    // groups in flight.  (LDS-DMA was built first: 24 buffer_load ... lds per wave and step cost 1.05 us of ISSUE
    // time alone -- about 100 cycles each -- against about 10 for a register load + a ds_write.)
    constexpr int SB = KB >= 32 ? 1 : 2;                     // groups in flight (register budget)
    bf16x8 stg[SB][GI];
    int kb_next = rot;                                       // k-block of walk position p, kept as a running scalar
    const int sbase = __builtin_amdgcn_readfirstlane((int)(((size_t)(t - t_first) * slab + (size_t)((m0 >> 4) + wave) * KB * 512) * 2));
    auto issue_group = [&](int buf) {
#if defined(CSN_NS_ABL) && CSN_NS_ABL >= 1
      return;                                               // ablation (timing only): no h loads
#endif
#pragma unroll
      for (int i = 0; i < GI; ++i) {
        stg[buf][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(hdst_rsrc, lane * 16, sbase + kb_next * 1024, 16));
        kb_next = kb_next + 1 == KB ? 0 : kb_next + 1;
      }
    };
    if (t > 0) {
      // wait for h_{t-1} (slot t): every wave polls the group's flag line (lane i the flag of slice i; sc1 loads)
      {
        const unsigned* fl = flags + (size_t)t * flag_step + (lane < nslices ? lane : 0);
        const unsigned long long t_begin = wall_clock64();
        while (!__all(__hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
          __builtin_amdgcn_s_sleep(1);
          if (__hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
          if (wall_clock64() - t_begin > kNsSpinTimeoutTicks) {
            __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
          }
        }
      }
      CSN_NSTAMP(0);   // wait for h_{t-1}
      if constexpr (HDMA) {
        // walk position p of this wave's row group <- k-block (p + rot) % KB, 1 KB per instruction, sc1 (L1 bypassed)
#pragma unroll
        for (int p = 0; p < KB; p += 4) {
          char* const dst = smem + ((size_t)wave * KB + p) * 1024;
          const int so = sbase + kb_next * 1024;
          ns_dma16_sc1<0>(hdst_rsrc, dst, lane * 16, so);
          ns_dma16_sc1<1024>(hdst_rsrc, dst, lane * 16, so);
          ns_dma16_sc1<2048>(hdst_rsrc, dst, lane * 16, so);
          ns_dma16_sc1<3072>(hdst_rsrc, dst, lane * 16, so);
          kb_next = kb_next + 4 >= KB ? kb_next + 4 - KB : kb_next + 4;
        }
      } else {
        issue_group(0);
        if (SB > 1) issue_group(1);
      }
      __builtin_amdgcn_sched_barrier(0);
      CSN_NSTAMP(8);   // first two groups requested
    }
    if constexpr (!FUSED) {
      // plain layers: the next step's projection (8 registers) is requested now, behind the first h loads, and has the
      // whole step to arrive from HBM
      request_input(s + 1 < nsteps ? t + 1 : t);
      __builtin_amdgcn_sched_barrier(0);
    }

    if (t > 0) {
      // MFMAs of one group: fragment reads in inline asm with counted lgkmcnt waits, two k-blocks ahead (one ahead
      // left every read's latency half exposed: 0.70 us per group of 48 MFMAs instead of 0.33)
      // (the fused body at H = 1024 keeps two: with three it is no faster, and its registers are the tightest of the library)
      constexpr int HD = (KB >= 32 && (!HDMA || FUSED)) ? 2 : 3;      // fragment buffers (register budget; the DMA form has no staging registers): reads run HD - 1 k-blocks ahead
      auto mfma_group = [&](int g) {
        bf16x8 hf[HD][4];
#pragma unroll
        for (int d = 0; d < HD - 1; ++d)
          if (d < GI) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) hf[d][rg] = ns_lds_read_b128(hbase + (unsigned)((rg * KB + g * GI + d) * 1024));
          }
#pragma unroll
        for (int i = 0; i < GI; ++i) {
          const int p = g * GI + i;
          if (i + HD - 1 < GI) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
              hf[(i + HD - 1) % HD][rg] = ns_lds_read_b128(hbase + (unsigned)((rg * KB + p + HD - 1) * 1024));
          }
          // reads still allowed in flight: those of the k-blocks after this one that have been issued
          const int ahead = (GI - 1 - i) < (HD - 1) ? (GI - 1 - i) : (HD - 1);
          if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
          else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int j = 0; j < 2; ++j) ns_mfma<AIA>(acc[rg][j], wreg[p][j], hf[i % HD][rg]);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
      if constexpr (HDMA) {
        // counted waits: behind the pieces of group g this wave has issued the (3 - g) GI pieces of the later groups and,
        // in the plain layers, the P loads of the next step's projection (vector-memory operations retire in order)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          ns_wait_group<KB, FUSED ? 0 : P>(g);
          __builtin_amdgcn_s_barrier();            // everybody's pieces of group g are in the tile
          __builtin_amdgcn_sched_barrier(0);
          mfma_group(g);
          __builtin_amdgcn_sched_barrier(0);
          CSN_NSTAMP(10 + g > 12 ? 12 : 10 + g);
        }
      } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        // group g has landed in this wave's registers (the compiler counts the vmcnt: only the next group's loads are
        // younger) -> into the tile; its registers take group g + 2; then the MFMAs of group g - 1, published by the
        // barrier of the round before; then this round's barrier publishes group g
#if !defined(CSN_NS_ABL) || CSN_NS_ABL < 1
#pragma unroll
        for (int i = 0; i < GI; ++i)
          *reinterpret_cast<bf16x8*>(smem + ((size_t)wave * KB + g * GI + i) * 1024 + lane * 16) = stg[g % SB][i];
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (g + SB < 4) issue_group(g % SB);
        __builtin_amdgcn_sched_barrier(0);
        if (g > 0) mfma_group(g - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        CSN_NSTAMP(10 + g > 12 ? 12 : 10 + g);
      }
      mfma_group(3);
      }
    }
    ns_mfma_fence();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (FUSED) {
      // (fused layer 0: the 64 registers of x are free only now; requested later than this -- after the gate math --
      // its latency lands in the publish's store drain, i.e. in front of the flag: measured twice, rounds 2 and 4)
      request_input(s + 1 < nsteps ? t + 1 : t);
      __builtin_amdgcn_sched_barrier(0);
    }
    CSN_NSTAMP(1);     // DMA + MFMA

    // ---- gate math in place; results into the workgroup's transpose area (padded rows: every ds_write covers all
    // banks evenly), then -- one barrier later -- out again in a thread mapping in which consecutive lanes store
    // consecutive bytes: a row of the tile is 256 B of gates, 128 B of c, 64 B of h, and the hand-off slab takes
    // the 4 blocks of this workgroup's k-block as 4 contiguous 1 KB runs
    {
      char* const sg = stage;                                  // gates [64 rows][288 B]: 32 units x (i, f, g, o) bf16
      char* const sc = stage + 64 * 288;                       // c     [64 rows][144 B]: 32 units f32
      char* const sh = stage + 64 * 288 + 64 * 144;            // h     [64 rows][ 80 B]: 32 units bf16
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x4 v = acc[rg][j];
          const float gi = fast_sigmoid(v[0]), gf = fast_sigmoid(v[1]), gg = fast_tanh(v[2]), go = fast_sigmoid(v[3]);
          const int row = rg * 16 + (lane & 15), unit = 8 * wave + 4 * j + (lane >> 4);
          float cprev;
          if constexpr (CLDS) cprev = *reinterpret_cast<const float*>(sc + row * 144 + unit * 4);
          else cprev = cst[rg][j];
          const float cn = gf * cprev + gi * gg;
          const float hn = go * fast_tanh(cn);
          if constexpr (!CLDS) cst[rg][j] = cn;
          *reinterpret_cast<bf16x4*>(sg + row * 288 + unit * 8) = (bf16x4){(bf16_t)gi, (bf16_t)gf, (bf16_t)gg, (bf16_t)go};
          *reinterpret_cast<float*>(sc + row * 144 + unit * 4) = cn;
          *reinterpret_cast<bf16_t*>(sh + row * 80 + unit * 2) = (bf16_t)hn;
        }
      CSN_NSTAMP(2);   // gate math + transpose writes
      __syncthreads();
      // the hand-off payload first: plain stores stay in this XCD's L2 (L2-local groups), write-through otherwise
      {
        const int rg = tid >> 6, r15 = tid & 15, q = (tid >> 4) & 3;
        const nu32x4 v = *reinterpret_cast<const nu32x4*>(sh + (rg * 16 + r15) * 80 + q * 16);
        const unsigned hoff = (unsigned)(((size_t)(t + 1 - t_first) * slab + ((size_t)((m0 >> 4) + rg) * KB + slice) * 512) * 2) + (unsigned)(tid & 63) * 16u;
        if (local) ns_store_b128<false>(hdst_rsrc, hoff, v);
        else ns_store_b128<true>(hdst_rsrc, hoff, v);
      }
      {
        const int row = tid >> 2, q = tid & 3;
        if (m0 + row < B)
          nt_store(reinterpret_cast<nu32x4*>(h_all + ((size_t)(t + 1) * B + m0 + row) * H + u0 + 8 * q),
                   *reinterpret_cast<const nu32x4*>(sh + row * 80 + q * 16));
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = tid + 256 * k, row = idx >> 3, ch = idx & 7;
        if (m0 + row < B)
          nt_store(reinterpret_cast<nu32x4*>(c_all + ((size_t)(t + 1) * B + m0 + row) * H + u0 + 4 * ch),
                   *reinterpret_cast<const nu32x4*>(sc + row * 144 + ch * 16));
      }
      if (gates != nullptr) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int idx = tid + 256 * k, row = idx >> 4, ch = idx & 15;
          if (m0 + row < B)
            nt_store(reinterpret_cast<nu32x4*>(gates + ((size_t)t * B + m0 + row) * 4 * H + 4 * (size_t)u0 + 8 * ch),
                     *reinterpret_cast<const nu32x4*>(sg + row * 288 + ch * 16));
        }
      }
    }
    CSN_NSTAMP(3);     // transpose reads + store issue
    // publish: every storing wave drains, workgroup barrier (also: every wave is done with the h tile in LDS, the
    // next step's DMA may overwrite it), one lane signals
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CSN_NSTAMP(4);     // drain + barrier
    if (tid == 0) {
      unsigned* fl = flags + (size_t)(t + 1) * flag_step + slice;
      if (local) *fl = 1u;
      else __hip_atomic_store(fl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    CSN_NSTAMP(5);     // signal
    if constexpr (FUSED) {
      if (s + 1 < nsteps) start_from_x();      // bias + x_{t+1} W_ih^T: x_{t+1} was requested behind this step's MFMAs, the drain above covered it
      CSN_NSTAMP(9);   // x MFMAs of the next step
    }
  }
}

#ifdef CSN_EXPERIMENTS
#include "experiments/lstm_fwd_kn.inc.h"
#endif

static constexpr int kKnLdsExtra = 4 * 6144 + 64 * (208 + 112 + 80);      // exchange + transpose area behind the h tile

// FUSE: may a slot of this launch be the fused layer 0?  Whether a workgroup's slot IS fused is a run-time,
// workgroup-uniform fact (one launch advances layer 0 and the layers above it).
// KN: the K2 x N2 body (24 units per workgroup) instead of the N-split one (32 units)
template <int KB, bool FUSE, bool KN>
__global__ void __launch_bounds__(256) lstm_fwd_ns_kernel(PersistFwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [4 rg][KB] 1 KB blocks of h, then the 28 KB transpose
  const int H = a.H, MT = a.MT;
  const int tid = threadIdx.x;
#ifdef CSN_PSTAMPS
  const unsigned long long t_entry_ = wall_clock64();
#endif
  const int nslices = KN ? H / 24 : H >> 5;
  int grp, slice;
  if (a.xcd_groups) {
    grp = blockIdx.x & 7;
    slice = blockIdx.x >> 3;
    const int ngroups = a.nslots * MT, gs = a.grid_slices;
    if (grp >= ngroups || slice >= nslices) {
      // no recurrence work for this workgroup: it walks the tiles of the launch's input-projection GEMMs (the chunk
      // the layer below finished one launch ago); workers of one XCD get consecutive indices
      if (a.ngemm > 0) {
        const int idle_here = gs - nslices;
        const unsigned base = grp <= ngroups ? (unsigned)(grp * idle_here)
                                             : (unsigned)(ngroups * idle_here + (grp - ngroups) * gs);
        const unsigned worker = base + (unsigned)(grp < ngroups ? slice - nslices : slice);
        const unsigned nworkers = (unsigned)(ngroups * idle_here + (8 - ngroups) * gs);
        for (int i = 0; i < a.ngemm; ++i) beside_gemm_tiles(a.gemm[i], smem, worker, nworkers);
      }
      return;
    }
  } else {
    grp = blockIdx.x / nslices;
    slice = blockIdx.x % nslices;
  }
  const PersistFwdSlot& S = a.slot[grp / MT];
  const int mt = grp % MT;

  // ---- is this group on one XCD?  (lstm_fwd_persist.hip)
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
        if (wall_clock64() - t_begin > kNsSpinTimeoutTicks) {
          __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      ((__attribute__((address_space(3))) int*)(__attribute__((address_space(3))) void*)smem)[0] = (int)(((v >> (8 + 6 * xcc)) & 63ull) == (unsigned long long)nslices);
    }
    __syncthreads();
    local = ((__attribute__((address_space(3))) int*)(__attribute__((address_space(3))) void*)smem)[0] != 0;    // (LDS address space: no flat_ instruction in these kernels -- FLAT retires out of order)
    __syncthreads();
  }
#ifdef CSN_PSTAMPS
  if (tid == 0 && blockIdx.x == CSN_STAMP_BLOCK) atomicAdd(&g_nstamps[6], wall_clock64() - t_entry_);
#endif
  bool done = false;
  if constexpr (FUSE) {
    if (__builtin_amdgcn_readfirstlane((int)(S.x_blk != nullptr)) != 0) {
#ifdef CSN_EXPERIMENTS
      if constexpr (KN) {
        if (a.half_tiles) kp_recurrence<KB, true>(a, S, smem, slice, mt, local);
        else kn_recurrence<KB, true>(a, S, smem, slice, mt, local);
      } else
#endif
        ns_recurrence<KB, true>(a, S, smem, slice, mt, local);
      done = true;
    }
  }
  if (!done) {
#ifdef CSN_EXPERIMENTS
    if constexpr (KN) {
      if (a.half_tiles) kp_recurrence<KB, false>(a, S, smem, slice, mt, local);
      else kn_recurrence<KB, false>(a, S, smem, slice, mt, local);
    } else
#endif
      ns_recurrence<KB, false>(a, S, smem, slice, mt, local);
  }
  // chunk finished: take tiles of the launch's GEMMs that are still unclaimed (counter mode only)
  for (int i = 0; i < a.ngemm; ++i)
    if (a.gemm[i].counter != nullptr) beside_gemm_tiles(a.gemm[i], smem, 0u, 1u);
}

static bool ns_device_has_256_cus() {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  return hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus >= 256;
}

int fwd_ns_slices(int H);
// all workgroups of a launch co-resident, each alone on its CU
bool fwd_ns_supported(int B, int H, int dtype, const Options& opt) {
  if (dtype != CSN_BF16 || opt.no_persist || opt.fwd_ksplit) return false;
  if (!(H == 128 || H == 256 || H == 384 || H == 512 || H == 768 || H == 1024)) return false;
  // where both forward kernels exist they are equally fast (cfg2: 11.2 vs 11.0 ms per step, profiles/r02_c): the
  // K-split one stays the default there; the N-split one is the only weight-stationary forward at H = 1024
  if (H != 1024 && !opt.fwd_nsplit && !(opt.fwd_ws && H == 768)) return false;
#ifndef CSN_EXPERIMENTS
  if (H == 768) return false;      // the K2 x N2 body lives in `make experiments` only; the K-split kernel is the one at H = 768
#endif
  if (!ns_device_has_256_cus()) return false;
  return fwd_ns_slices(H) * ((B + 63) / 64) <= 128;
}

template <int KB, bool FUSE, bool KN>
static int launch_ns_t(const PersistFwdArgs& a, hipStream_t st) {
  size_t lds = (size_t)KB * 4096 + (KN ? kKnLdsExtra : kNsStageBytes);
  if (int rc = ensure_dyn_lds<&lstm_fwd_ns_kernel<KB, FUSE, KN>>((int)(lds > kBesideLdsBytes + 64 ? lds : kBesideLdsBytes + 64))) return rc;
  const unsigned nslices = (unsigned)(KN ? a.H / 24 : a.H / 32);
  PersistFwdArgs b = a;
  if (b.xcd_groups) {
    if (b.ngemm > 0) {
      if (lds < kBesideLdsBytes + 64) lds = kBesideLdsBytes + 64;     // the GEMM workers' staging ring + the claim word
      if (b.grid_slices < (int)nslices) b.grid_slices = (int)nslices;
    } else {
      b.grid_slices = (int)nslices;
    }
  }
  const unsigned grid = b.xcd_groups ? 8u * (unsigned)b.grid_slices : nslices * (unsigned)(b.MT * b.nslots);
  lstm_fwd_ns_kernel<KB, FUSE, KN><<<dim3(grid), 256, lds, st>>>(b);
  CSN_LAUNCH_CHECK();
  return CSN_OK;
}

// which body runs: the K2 x N2 one at H = 768 (32 workgroups per group), else the N-split one
static bool ns_use_kn(int H) { return H == 768; }
int fwd_ns_slices(int H) { return ns_use_kn(H) ? H / 24 : H / 32; }

int launch_fwd_ns(const PersistFwdArgs& a, hipStream_t st) {
  CSN_REQUIRE(a.nslots >= (a.ngemm > 0 ? 0 : 1) && a.nslots <= 4 && a.MT >= 1, "launch_fwd_ns: bad slot count");
  const int ns = fwd_ns_slices(a.H);
  CSN_REQUIRE(ns <= kPersistFlagLine, "launch_fwd_ns: H=%d gives %d slices", a.H, ns);
  if (a.xcd_groups) CSN_REQUIRE(a.nslots * a.MT <= 8, "launch_fwd_ns: groups do not fit 8 XCDs");
  CSN_REQUIRE(a.ngemm >= 0 && a.ngemm <= 3 && (a.ngemm == 0 || a.xcd_groups), "launch_fwd_ns: bad GEMM list");
  bool fused = false;          // does any slot of the launch multiply x_t itself?
  for (int i = 0; i < a.nslots; ++i) {
    fused = fused || a.slot[i].x_blk != nullptr;
    CSN_REQUIRE(a.slot[i].x_blk == nullptr || a.slot[i].I == 128, "launch_fwd_ns: the fused input projection takes I = 128");
    // the kernel's buffer resources are based at the launch's first step and addressed with 32-bit byte offsets: the
    // steps of ONE launch (CSN_LSTM_CHUNK) must keep the float32 projection ([B, 4H] per step), the fragment-major input
    // ([Bpad, I] bf16) and the h slabs ([Bpad, H] bf16, nsteps + 1 of them) below 4 GiB each
    const unsigned long long n = (unsigned long long)a.slot[i].nsteps;
    const unsigned long long worst = std::max({n * (unsigned long long)a.B * a.H * 16ull,
                                               n * (unsigned long long)a.Bpad * (unsigned long long)a.slot[i].I * 2ull,
                                               (n + 1ull) * (unsigned long long)a.Bpad * a.H * 2ull});
    CSN_REQUIRE(worst < (1ull << 32), "launch_fwd_ns: %llu steps per launch at B=%d H=%d address %llu bytes from the launch's base "
                "(32-bit offsets): lower CSN_LSTM_CHUNK", n, a.B, a.H, worst);
  }
#ifdef CSN_EXPERIMENTS
  if (a.H == 768) return fused ? launch_ns_t<24, true, true>(a, st) : launch_ns_t<24, false, true>(a, st);
#endif
#define CSN_NS_CASE(KBV)                                                   \
  case KBV * 32:                                                           \
    return fused ? launch_ns_t<KBV, true, false>(a, st) : launch_ns_t<KBV, false, false>(a, st)
  switch (a.H) {
    CSN_NS_CASE(4);
    CSN_NS_CASE(8);
    CSN_NS_CASE(12);
    CSN_NS_CASE(16);
    CSN_NS_CASE(32);      // (H = 1024, fused: W_ih in VGPRs -- lstm_ns_util.h:ns_mfma<.., WV>)
  }
#undef CSN_NS_CASE
  return fail(CSN_ERR_UNSUPPORTED, "launch_fwd_ns: no kernel for H=%d", a.H);
}

}  // namespace csn

#ifdef CSN_PSTAMPS
extern "C" int csn_debug_read_nstamps(unsigned long long* out) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nstamps), sizeof(z)) != hipSuccess) return 1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_nstamps), z, sizeof(z)) != hipSuccess) return 1;
  return 0;
}
#endif
