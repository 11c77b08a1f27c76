// Forward bodies that LOSE to the default kernels at the widths they were built for (DESIGN.md section 3.4 (o), K2 x N2):
// the K2 x N2 body at H = 768 and its half-pipelined variant (CSN_FWD_HALVES).  Not part of libcsn_hip.so: included by
// lstm_fwd_ns.hip only in `make experiments` (-DCSN_EXPERIMENTS -> lib/libcsn_hip_experiments.so), kept as lab notes
// with their bit-identity tests (tests/test_gpu_experiments.py, run only when that library has been built).
#pragma once
// ---------------------------------------------------------------------------------------------------------------
// K2 x N2 form (H % 24 == 0: 384, 768): the MFMA phase of a step runs at the MFMA rate -- 14 ns per
// v_mfma_f32_16x16x32_bf16 at the clock the chip holds under MFMA load (tools/mfma_rate.hip: 16.3 shader cycles =
// 12 - 14 ns, i.e. about 1.2 GHz) -- so the step time is (MFMAs per wave) x 14 ns + everything that is not MFMA.  The
// N-split body above has the cheap "everything else" (2.7 us against 4.4 us of the K-split kernel) but, with 32
// units per workgroup, only 24 workgroups per group: 192 MFMAs per wave = 2.7 us where the K-split kernel's 32
// workgroups need 144 = 2.0 us.  This body keeps 32 workgroups per group (64 rows x 24 units = 6 gate-row tiles):
// wave (kh, th) multiplies K-half kh of tiles 3 th .. 3 th + 2 (144 MFMAs, 36 weight fragments = 144 AGPRs), the two
// K-halves swap the half of their partial tiles the other one finishes (24 KB through LDS instead of the 100 KB
// 4-way reduction), and each wave runs the gate math in place on 2 row groups x 3 tiles = 6 cells per lane.
template <int KB, bool FUSED>
__device__ __forceinline__ void kn_recurrence(const PersistFwdArgs& a, const PersistFwdSlot& S, char* smem, int slice,
                                              int mt, bool local) {
  constexpr int KH = KB / 2;                       // k-blocks of one K-half
  constexpr int GQ = KH / 4;                       // k-blocks per load group (4 groups); a wave loads 2 row groups of them
  constexpr int GI = 2 * GQ;                       // loads per wave and group
  static_assert(KH % 4 == 0, "4 load groups per K-half");
  constexpr int P = FUSED ? 8 : 6;                 // 16-byte registers of one input request
  const int B = a.B, H = a.H, MT = a.MT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave >> 1, th = wave & 1;
#ifdef CSN_PSTAMPS
  unsigned long long last_ = wall_clock64();
#endif
  const int nslices = H / 24;
  const int u0 = slice * 24, m0 = mt * 64;
  const size_t slab = (size_t)a.Bpad * H;
  bf16_t* const gates = S.gates;
  float* const c_all = S.c_all;
  bf16_t* const h_all = S.h_all;
  unsigned* const flags = S.flags + (size_t)mt * kPersistFlagLine;
  const size_t flag_step = (size_t)MT * kPersistFlagLine;
  const int t_first = S.t0, nsteps = S.nsteps;
  constexpr int xkb = 4;                               // fused form: I = 128 (checked by the launcher)
  // LDS: h tile [2 kh][4 rg][KH] 1 KB blocks | partial-sum exchange 4 x 6 KB | transpose area
  char* const exch = smem + (size_t)KB * 4096;
  char* const stage = exch + 4 * 6144;
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  // ---- stationary operands: K-half kh of this wave's 3 gate-row tiles; register p holds k-block kh KH + (p + rot) % KH
  const int rot = a.rotate ? __builtin_amdgcn_readfirstlane((slice * KH) / nslices) : 0;
  const int tile0 = 6 * slice + 3 * th;                // first of this wave's three 16-row tiles of the interleaved 4H axis
  bf16x8 wreg[KH][3];
#pragma unroll
  for (int p = 0; p < KH; ++p) {
    int kb = p + rot;
    kb = (kb >= KH ? kb - KH : kb) + kh * KH;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      wreg[p][j] = *reinterpret_cast<const bf16x8*>(S.w_blk + ((int64_t)(tile0 + j) * KB + kb) * 512 + lane * 8);
  }
  bf16x8 wih[FUSED ? 2 : 1][3];                        // fused: x k-blocks 2 kh, 2 kh + 1
  f32x4 biasv[3];
  if constexpr (FUSED) {
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        wih[k2][j] = *reinterpret_cast<const bf16x8*>(S.wih_blk + ((int64_t)(tile0 + j) * xkb + 2 * kh + k2) * 512 + lane * 8);
#pragma unroll
    for (int j = 0; j < 3; ++j)
      biasv[j] = *reinterpret_cast<const f32x4*>(S.bias + 16 * (size_t)(tile0 + j) + 4 * (lane >> 4));
  }

  // ---- the 6 cells this lane FINISHES: row groups 2 kh, 2 kh + 1; units u0 + 12 th + 4 j + (lane >> 4)
  const int unit_q = u0 + 12 * th + (lane >> 4);           // + 4 j
  int rowc[2];
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) {
    const int r = m0 + 16 * (2 * kh + r2) + (lane & 15);
    rowc[r2] = r < B ? r : B - 1;
  }
  float cst[2][3];
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      cst[r2][j] = t_first > 0 ? c_all[((size_t)t_first * B + rowc[r2]) * H + unit_q + 4 * j] : 0.0f;

  // next step's input: plain layers the projection of the 6 cells (6 x 16 B); the fused layer 0 the x fragments of
  // all 4 row groups for this wave's 2 input k-blocks (8 x 16 B)
  f32x4 nxt[P];
  const unsigned xslab = FUSED ? (unsigned)a.Bpad * (unsigned)S.I : 0u;
  const __amdgpu_buffer_rsrc_t in_rsrc = FUSED
      ? __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.x_blk), 0, __builtin_amdgcn_readfirstlane((int)((size_t)a.T * xslab * 2)), 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.xproj), 0, __builtin_amdgcn_readfirstlane((int)((size_t)a.T * B * 16 * H)), 0x00020000);
  int xvoff[2];
#pragma unroll
  for (int r2 = 0; r2 < 2; ++r2) xvoff[r2] = (int)(((size_t)rowc[r2] * 4 * H + 4 * (size_t)unit_q) * 4);
  auto request_input = [&](int t) {
    if constexpr (FUSED) {
      const int sbase = __builtin_amdgcn_readfirstlane((int)(((size_t)t * xslab + (size_t)(m0 >> 4) * xkb * 512) * 2) + 2 * kh * 1024);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) nxt[rg * 2 + k2] = ns_bload_nt_f32x4(in_rsrc, lane * 16 + (rg * 4 + k2) * 1024, sbase);
    } else {
      const int sbase = __builtin_amdgcn_readfirstlane((int)((size_t)t * B * 16 * H));
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int j = 0; j < 3; ++j) nxt[r2 * 3 + j] = ns_bload_nt_f32x4(in_rsrc, xvoff[r2] + j * 64, sbase);
    }
  };
  request_input(t_first);

  const __amdgpu_buffer_rsrc_t hdst_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.h_blk_all), 0, __builtin_amdgcn_readfirstlane((int)((size_t)(a.T + 1) * slab * 2)), 0x00020000);

  for (int s = 0; s < nsteps; ++s) {
    const int t = t_first + s;
    f32x4 acc[4][3];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[rg][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ns_mfma_fence();

    unsigned zero_;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_));
    const unsigned hbase = lds_base + (unsigned)(kh * 4 * KH * 1024) + (unsigned)lane * 16u + zero_;   // this wave's K-half of the tile
    // this wave brings in row groups 2 th, 2 th + 1 of K-half kh: registers (sc1 buffer loads) -> ds_write, 4 groups
    bf16x8 stg[2][GI];
    int kb_next = rot;
    const int sbase = __builtin_amdgcn_readfirstlane((int)(((size_t)t * slab + ((size_t)((m0 >> 4) + 2 * th) * KB + kh * KH) * 512) * 2));
    auto issue_group = [&](int buf) {
#pragma unroll
      for (int i = 0; i < GQ; ++i) {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
          stg[buf][i * 2 + r2] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(hdst_rsrc, lane * 16 + r2 * KB * 1024, sbase + kb_next * 1024, 16));
        kb_next = kb_next + 1 == KH ? 0 : kb_next + 1;
      }
    };
    if (t > 0) {
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
      issue_group(0);
      issue_group(1);
      __builtin_amdgcn_sched_barrier(0);
      CSN_NSTAMP(8);
    }
    if constexpr (FUSED) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
          for (int j = 0; j < 3; ++j) ns_mfma<true>(acc[rg][j], wih[k2][j], __builtin_bit_cast(bf16x8, nxt[rg * 2 + k2]));
      __builtin_amdgcn_sched_barrier(0);
      CSN_NSTAMP(9);
      request_input(s + 1 < nsteps ? t + 1 : t);            // (its 8 registers were just consumed)
      __builtin_amdgcn_sched_barrier(0);
    }

    if (t > 0) {
      auto mfma_group = [&](int g) {
        bf16x8 hf[3][4];
#pragma unroll
        for (int d = 0; d < 2; ++d)
          if (d < GQ) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) hf[d][rg] = ns_lds_read_b128(hbase + (unsigned)((rg * KH + g * GQ + d) * 1024));
          }
#pragma unroll
        for (int i = 0; i < GQ; ++i) {
          const int p = g * GQ + i;
          if (i + 2 < GQ) {
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) hf[(i + 2) % 3][rg] = ns_lds_read_b128(hbase + (unsigned)((rg * KH + p + 2) * 1024));
          }
          const int ahead = (GQ - 1 - i) < 2 ? (GQ - 1 - i) : 2;
          if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
          else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int j = 0; j < 3; ++j) ns_mfma<true>(acc[rg][j], wreg[p][j], hf[i % 3][rg]);
          __builtin_amdgcn_sched_barrier(0);
        }
      };
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int i = 0; i < GQ; ++i)
#pragma unroll
          for (int r2 = 0; r2 < 2; ++r2)
            *reinterpret_cast<bf16x8*>(smem + ((size_t)(kh * 4 + 2 * th + r2) * KH + g * GQ + i) * 1024 + lane * 16) = stg[g & 1][i * 2 + r2];
        __builtin_amdgcn_sched_barrier(0);
        if (g + 2 < 4) issue_group(g & 1);
        __builtin_amdgcn_sched_barrier(0);
        if (g > 0) mfma_group(g - 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        CSN_NSTAMP(10 + g > 12 ? 12 : 10 + g);
      }
      mfma_group(3);
    }
    ns_mfma_fence();
    __builtin_amdgcn_sched_barrier(0);
    CSN_NSTAMP(1);     // loads + MFMA

    // ---- the two K-halves swap what the other one finishes: wave (kh, th) gives away row groups 2 (1 - kh) + {0, 1}
    {
      f32x4* const mine = reinterpret_cast<f32x4*>(exch + (size_t)wave * 6144);
      const f32x4* const theirs = reinterpret_cast<const f32x4*>(exch + (size_t)(wave ^ 2) * 6144);
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int j = 0; j < 3; ++j) mine[(r2 * 3 + j) * 64 + lane] = kh == 0 ? acc[2 + r2][j] : acc[r2][j];   // (static indices: kh is a scalar)
      __syncthreads();
      // gate math in place on the 6 finished cells; results into the workgroup's transpose area
      char* const sg = stage;                                  // gates [64 rows][208 B]: 24 units x (i, f, g, o) bf16
      char* const sc = stage + 64 * 208;                       // c     [64 rows][112 B]: 24 units f32
      char* const sh = stage + 64 * 208 + 64 * 112;            // h     [64 rows][ 80 B]: 24 units bf16
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const f32x4 o = theirs[(r2 * 3 + j) * 64 + lane];
          const f32x4 own = kh == 0 ? acc[r2][j] : acc[2 + r2][j];
          const f32x4 in = FUSED ? biasv[j] : nxt[r2 * 3 + j];
          const f32x4 v = own + o + in;
          const float gi = fast_sigmoid(v[0]), gf = fast_sigmoid(v[1]), gg = fast_tanh(v[2]), go = fast_sigmoid(v[3]);
          const float cn = gf * cst[r2][j] + gi * gg;
          const float hn = go * fast_tanh(cn);
          cst[r2][j] = cn;
          const int row = (2 * kh + r2) * 16 + (lane & 15), unit = 12 * th + 4 * j + (lane >> 4);
          *reinterpret_cast<bf16x4*>(sg + row * 208 + unit * 8) = (bf16x4){(bf16_t)gi, (bf16_t)gf, (bf16_t)gg, (bf16_t)go};
          *reinterpret_cast<float*>(sc + row * 112 + unit * 4) = cn;
          *reinterpret_cast<bf16_t*>(sh + row * 80 + unit * 2) = (bf16_t)hn;
        }
      if constexpr (!FUSED) {
        __builtin_amdgcn_sched_barrier(0);
        request_input(s + 1 < nsteps ? t + 1 : t);          // (its 6 registers were just consumed)
      }
      CSN_NSTAMP(2);   // exchange + gate math + transpose writes
      __syncthreads();
      // out again with consecutive lanes on consecutive bytes; the hand-off payload first
      if (tid < 192) {
        const int row = tid & 63, c8 = tid >> 6;              // 16 consecutive rows of a block are 256 contiguous bytes
        const nu32x4 v = *reinterpret_cast<const nu32x4*>(sh + row * 80 + c8 * 16);
        const unsigned hoff = (unsigned)(((size_t)(t + 1) * slab + blk_offset(m0 + row, u0 + 8 * c8, H)) * 2);
        if (local) ns_store_b128<false>(hdst_rsrc, hoff, v);
        else ns_store_b128<true>(hdst_rsrc, hoff, v);
        const int row2 = tid / 3, c2 = tid % 3;
        if (m0 + row2 < B)
          nt_store(reinterpret_cast<nu32x4*>(h_all + ((size_t)(t + 1) * B + m0 + row2) * H + u0 + 8 * c2),
                   *reinterpret_cast<const nu32x4*>(sh + row2 * 80 + c2 * 16));
      }
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = tid + 256 * k, row = idx / 6, ch = idx % 6;
        if (idx < 384 && m0 + row < B)
          nt_store(reinterpret_cast<nu32x4*>(c_all + ((size_t)(t + 1) * B + m0 + row) * H + u0 + 4 * ch),
                   *reinterpret_cast<const nu32x4*>(sc + row * 112 + ch * 16));
      }
      if (gates != nullptr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int idx = tid + 256 * k, row = idx / 12, ch = idx % 12;
          if (m0 + row < B)
            nt_store(reinterpret_cast<nu32x4*>(gates + ((size_t)t * B + m0 + row) * 4 * H + 4 * (size_t)u0 + 8 * ch),
                     *reinterpret_cast<const nu32x4*>(sg + row * 208 + ch * 16));
        }
      }
    }
    CSN_NSTAMP(3);     // transpose reads + store issue
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    CSN_NSTAMP(4);     // drain + barrier
    if (tid == 0) {
      unsigned* fl = flags + (size_t)(t + 1) * flag_step + slice;
      if (local) *fl = 1u;
      else __hip_atomic_store(fl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    CSN_NSTAMP(5);     // signal
  }
}
// ---------------------------------------------------------------------------------------------------------------
// K2 x N2 form, PIPELINED over the two 32-row halves of the tile (the default forward at H = 768).
// A step of the bodies above is a serial chain: poll -> loads (0.6 us to the first bytes) -> MFMA -> gate math -> stores
// -> drain -> flag -> (0.5 us until the others see it) -> poll ...: only 2.0 of its 6.4 us are MFMA time.  The chain
// cannot be shortened, but two of them can share a workgroup: the 64-row tile is handed off as two 32-row halves with
// flags of their own, and the step is software-pipelined so that every latency of one half lies under work of the
// other --
//      poll A, request h_A | finish B (swap partials, gate math, stores, drain, flag B) | MFMA A     <- flag B travels
//      poll B, request h_B | finish A (                  ...                 , flag A) | MFMA B     <- flag A travels
// The weights are shared (one workgroup, one register set): nothing is duplicated, unlike two workgroups per CU
// (tried: 256 registers per wave do not hold the 144 weight registers + the rest -- 44 to 131 spills).
template <int KB, bool FUSED>
__device__ __forceinline__ void kp_recurrence(const PersistFwdArgs& a, const PersistFwdSlot& S, char* smem, int slice,
                                              int mt, bool local) {
  constexpr int KH = KB / 2;                       // k-blocks of one K-half
  constexpr int LG = KH / 2;                       // loads per wave and group: a wave brings in ONE row group of a half, 2 groups
  static_assert(KH % 2 == 0, "2 load groups per half");
  const int B = a.B, H = a.H, MT = a.MT;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kh = wave >> 1, th = wave & 1;
#ifdef CSN_PSTAMPS
  unsigned long long last_ = wall_clock64();
#endif
  const int nslices = H / 24;
  const int u0 = slice * 24, m0 = mt * 64;
  const size_t slab = (size_t)a.Bpad * H;
  bf16_t* const gates = S.gates;
  float* const c_all = S.c_all;
  bf16_t* const h_all = S.h_all;
  // flag lines: [T+1][MT][2 halves][line]
  unsigned* const flags = S.flags + (size_t)mt * 2 * kPersistFlagLine;
  const size_t flag_step = (size_t)MT * 2 * kPersistFlagLine;
  const int t_first = S.t0, nsteps = S.nsteps;
  constexpr int xkb = 4;
  char* const exch = smem + (size_t)KB * 4096;         // 4 waves x 3 KB
  char* const stage = exch + 4 * 3072;                 // 32 rows x (208 + 112 + 80)
  const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;

  const int rot = a.rotate ? __builtin_amdgcn_readfirstlane((slice * KH) / nslices) : 0;
  const int tile0 = 6 * slice + 3 * th;
  bf16x8 wreg[KH][3];
#pragma unroll
  for (int p = 0; p < KH; ++p) {
    int kb = p + rot;
    kb = (kb >= KH ? kb - KH : kb) + kh * KH;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      wreg[p][j] = *reinterpret_cast<const bf16x8*>(S.w_blk + ((int64_t)(tile0 + j) * KB + kb) * 512 + lane * 8);
  }
  bf16x8 wih[FUSED ? 2 : 1][3];
  f32x4 biasv[3];
  if constexpr (FUSED) {
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        wih[k2][j] = *reinterpret_cast<const bf16x8*>(S.wih_blk + ((int64_t)(tile0 + j) * xkb + 2 * kh + k2) * 512 + lane * 8);
#pragma unroll
    for (int j = 0; j < 3; ++j)
      biasv[j] = *reinterpret_cast<const f32x4*>(S.bias + 16 * (size_t)(tile0 + j) + 4 * (lane >> 4));
  }

  // the 3 cells this lane finishes in each half: row group 2 hf + kh, units u0 + 12 th + 4 j + (lane >> 4)
  const int unit_q = u0 + 12 * th + (lane >> 4);
  int rowc[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    const int r = m0 + 16 * (2 * hf + kh) + (lane & 15);
    rowc[hf] = r < B ? r : B - 1;
  }
  float cst[2][3];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      cst[hf][j] = t_first > 0 ? c_all[((size_t)t_first * B + rowc[hf]) * H + unit_q + 4 * j] : 0.0f;

  // next step's input per half: plain layers the projection of the 3 cells (3 x 16 B); the fused layer 0 the x
  // fragments of the half's 2 row groups for this wave's 2 input k-blocks (4 x 16 B)
  constexpr int P = FUSED ? 4 : 3;
  f32x4 nxt[2][P];
  const unsigned xslab = FUSED ? (unsigned)a.Bpad * (unsigned)S.I : 0u;
  const __amdgpu_buffer_rsrc_t in_rsrc = FUSED
      ? __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.x_blk), 0, __builtin_amdgcn_readfirstlane((int)((size_t)a.T * xslab * 2)), 0x00020000)
      : __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.xproj), 0, __builtin_amdgcn_readfirstlane((int)((size_t)a.T * B * 16 * H)), 0x00020000);
  int xvoff[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) xvoff[hf] = (int)(((size_t)rowc[hf] * 4 * H + 4 * (size_t)unit_q) * 4);
  auto request_input = [&](int hf, int t) {
    if constexpr (FUSED) {
      const int sbase = __builtin_amdgcn_readfirstlane((int)(((size_t)t * xslab + (size_t)(m0 >> 4) * xkb * 512) * 2) + 2 * kh * 1024);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) nxt[hf][r * 2 + k2] = ns_bload_nt_f32x4(in_rsrc, lane * 16 + ((2 * hf + r) * 4 + k2) * 1024, sbase);
    } else {
      const int sbase = __builtin_amdgcn_readfirstlane((int)((size_t)t * B * 16 * H));
#pragma unroll
      for (int j = 0; j < 3; ++j) nxt[hf][j] = ns_bload_nt_f32x4(in_rsrc, xvoff[hf] + j * 64, sbase);
    }
  };
  request_input(0, t_first);
  request_input(1, t_first);

  const __amdgpu_buffer_rsrc_t hdst_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void*)ns_uniform(S.h_blk_all), 0, __builtin_amdgcn_readfirstlane((int)((size_t)(a.T + 1) * slab * 2)), 0x00020000);

  f32x4 acc[4][3];
  bf16x8 stg[2][LG];

  // ---- poll the flags of half hf for step t and request its h rows (this wave: row group 2 hf + th, K-half kh)
  auto poll_and_request = [&](int hf, int t) {
    {
      const unsigned* line = ns_uniform(flags + (size_t)t * flag_step + hf * kPersistFlagLine);
      const unsigned long long t_begin = wall_clock64();
      while (ns_flags_set_scalar(line, nslices) < nslices) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t_begin > kNsSpinTimeoutTicks) {
          __hip_atomic_store(a.error_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        if ((wall_clock64() - t_begin) > 2000ull &&           // (every 20 us of waiting: has someone else given up?)
            __hip_atomic_load(a.error_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)
          break;
      }
    }
    const int sbase = __builtin_amdgcn_readfirstlane((int)(((size_t)t * slab + ((size_t)((m0 >> 4) + 2 * hf + th) * KB + kh * KH) * 512) * 2));
    int kb_next = rot;
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int i = 0; i < LG; ++i) {
        stg[g][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(hdst_rsrc, lane * 16, sbase + kb_next * 1024, 16));
        kb_next = kb_next + 1 == KH ? 0 : kb_next + 1;
      }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- accumulators of half hf: zero, (fused) + x_t W_ih^T
  auto start_half = [&](int hf) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[2 * hf + r][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    ns_mfma_fence();
    if constexpr (FUSED) {
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int j = 0; j < 3; ++j) ns_mfma<true>(acc[2 * hf + r][j], wih[k2][j], __builtin_bit_cast(bf16x8, nxt[hf][r * 2 + k2]));
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- h rows of half hf: registers -> LDS tile, then the MFMAs (2 groups of LG k-blocks)
  auto mfma_half = [&](int hf) {
    unsigned zero_;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_));
    const unsigned hbase = lds_base + (unsigned)((kh * 4 + 2 * hf) * KH * 1024) + (unsigned)lane * 16u + zero_;
    auto mfma_group = [&](int g) {
      bf16x8 hfr[3][2];
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 2; ++r) hfr[d][r] = ns_lds_read_b128(hbase + (unsigned)((r * KH + g * LG + d) * 1024));
#pragma unroll
      for (int i = 0; i < LG; ++i) {
        const int p = g * LG + i;
        if (i + 2 < LG) {
#pragma unroll
          for (int r = 0; r < 2; ++r) hfr[(i + 2) % 3][r] = ns_lds_read_b128(hbase + (unsigned)((r * KH + p + 2) * 1024));
        }
        const int ahead = (LG - 1 - i) < 2 ? (LG - 1 - i) : 2;
        if (ahead == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int j = 0; j < 3; ++j) ns_mfma<true>(acc[2 * hf + r][j], wreg[p][j], hfr[i % 3][r]);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll
    for (int g = 0; g < 2; ++g) {
#pragma unroll
      for (int i = 0; i < LG; ++i)
        *reinterpret_cast<bf16x8*>(smem + ((size_t)(kh * 4 + 2 * hf + th) * KH + g * LG + i) * 1024 + lane * 16) = stg[g][i];
      __builtin_amdgcn_sched_barrier(0);
      if (g > 0) mfma_group(g - 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    mfma_group(1);
    ns_mfma_fence();
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- finish half hf of step t: swap partial tiles between the K-halves, gate math, transpose, stores, flag
  auto finish_half = [&](int hf, int t, bool more) {
    f32x4* const mine = reinterpret_cast<f32x4*>(exch + (size_t)wave * 3072);
    const f32x4* const theirs = reinterpret_cast<const f32x4*>(exch + (size_t)(wave ^ 2) * 3072);
#pragma unroll
    for (int j = 0; j < 3; ++j) mine[j * 64 + lane] = kh == 0 ? acc[2 * hf + 1][j] : acc[2 * hf][j];   // row group 2 hf + (1 - kh)
    __syncthreads();
    char* const sg = stage;                                  // gates [32 rows][208 B]
    char* const sc = stage + 32 * 208;                       // c     [32 rows][112 B]
    char* const sh = stage + 32 * 208 + 32 * 112;            // h     [32 rows][ 80 B]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const f32x4 o = theirs[j * 64 + lane];
      const f32x4 own = kh == 0 ? acc[2 * hf][j] : acc[2 * hf + 1][j];
      const f32x4 in = FUSED ? biasv[j] : nxt[hf][j];
      const f32x4 v = own + o + in;
      const float gi = fast_sigmoid(v[0]), gf = fast_sigmoid(v[1]), gg = fast_tanh(v[2]), go = fast_sigmoid(v[3]);
      const float cn = gf * cst[hf][j] + gi * gg;
      const float hn = go * fast_tanh(cn);
      cst[hf][j] = cn;
      const int row = kh * 16 + (lane & 15), unit = 12 * th + 4 * j + (lane >> 4);
      *reinterpret_cast<bf16x4*>(sg + row * 208 + unit * 8) = (bf16x4){(bf16_t)gi, (bf16_t)gf, (bf16_t)gg, (bf16_t)go};
      *reinterpret_cast<float*>(sc + row * 112 + unit * 4) = cn;
      *reinterpret_cast<bf16_t*>(sh + row * 80 + unit * 2) = (bf16_t)hn;
    }
    __syncthreads();
    const int mh = m0 + 32 * hf;                              // first row of the half
    if (tid < 96) {
      const int row = tid & 31, c8 = tid >> 5;                // 16 consecutive rows of a block are 256 contiguous bytes
      const nu32x4 v = *reinterpret_cast<const nu32x4*>(sh + row * 80 + c8 * 16);
      const unsigned hoff = (unsigned)(((size_t)(t + 1) * slab + blk_offset(mh + row, u0 + 8 * c8, H)) * 2);
      if (local) ns_store_b128<false>(hdst_rsrc, hoff, v);
      else ns_store_b128<true>(hdst_rsrc, hoff, v);
      const int row2 = tid / 3, c2 = tid % 3;
      if (mh + row2 < B)
        nt_store(reinterpret_cast<nu32x4*>(h_all + ((size_t)(t + 1) * B + mh + row2) * H + u0 + 8 * c2),
                 *reinterpret_cast<const nu32x4*>(sh + row2 * 80 + c2 * 16));
    }
    if (tid < 192) {
      const int row = tid / 6, ch = tid % 6;
      if (mh + row < B)
        nt_store(reinterpret_cast<nu32x4*>(c_all + ((size_t)(t + 1) * B + mh + row) * H + u0 + 4 * ch),
                 *reinterpret_cast<const nu32x4*>(sc + row * 112 + ch * 16));
    }
    if (gates != nullptr) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = tid + 256 * k, row = idx / 12, ch = idx % 12;
        if (idx < 384 && mh + row < B)
          nt_store(reinterpret_cast<nu32x4*>(gates + ((size_t)t * B + mh + row) * 4 * H + 4 * (size_t)u0 + 8 * ch),
                   *reinterpret_cast<const nu32x4*>(sg + row * 208 + ch * 16));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      unsigned* fl = flags + (size_t)(t + 1) * flag_step + hf * kPersistFlagLine + slice;
      if (local) *fl = 1u;
      else __hip_atomic_store(fl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the half's next input (HBM): requested only now, BEHIND the drain and the flag -- in front of them, the drain
    // (vmcnt counts in order) waited for its HBM latency and the other half's poll queued behind it: 7.5 us per step
    __builtin_amdgcn_sched_barrier(0);
    if (more) request_input(hf, t + 1);
    __builtin_amdgcn_sched_barrier(0);
  };

  for (int s = 0; s < nsteps; ++s) {
    const int t = t_first + s;
    const bool more = s + 1 < nsteps;
    // ---- half A of step t; half B of step t-1 finishes under A's load latency
    if (t > 0) poll_and_request(0, t);
    CSN_NSTAMP(0);
    if (s > 0) finish_half(1, t - 1, true);
    CSN_NSTAMP(2);
    start_half(0);
    if (t > 0) mfma_half(0);
    CSN_NSTAMP(1);
    // ---- half B of step t; half A finishes under B's load latency
    if (t > 0) poll_and_request(1, t);
    CSN_NSTAMP(8);
    finish_half(0, t, more);
    CSN_NSTAMP(3);
    start_half(1);
    if (t > 0) mfma_half(1);
    CSN_NSTAMP(9);
  }
  finish_half(1, t_first + nsteps - 1, false);
}
