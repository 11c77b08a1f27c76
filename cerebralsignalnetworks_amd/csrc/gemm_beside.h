// A bf16 NT GEMM as a DEVICE function for a 256-thread workgroup that is alone on its CU: C[M,N] (f32) =
// A[M,K] * Bt[N,K]^T, 256 x 128 tiles walked by `nworkers` workgroups.  It runs inside the weight-stationary
// backward launch (lstm_bwd_persist.hip) on the workgroups that launch would otherwise leave idle -- the input
// gradient dx of the chunk the layer above finished one launch ago -- so it needs no stream, event or flag: both
// of its dependencies are kernel boundaries.
//
// Structure = gemm_nt_256_kernel (gemm.hip): ring of 3 LDS-DMA stages of 64 contraction columns (A 32 KB + B 16 KB,
// 128-byte rows, XOR swizzle applied on the source side), counted vmcnt + raw barrier, fragment reads in inline
// asm with counted lgkmcnt; here 4 waves (2 x 2) of 128 x 64 outputs each.  K % 64 == 0, N % 4 == 0.
#pragma once
#include "csn_common.h"
#include "lstm_cell_common.h"
#include "lstm_cell_blk.h"

namespace csn {

static constexpr int kBesideStages = 3;
static constexpr unsigned kBesideStageBytes = 49152;
static constexpr unsigned kBesideLdsBytes = kBesideStages * kBesideStageBytes;

__device__ __forceinline__ bf16x8 beside_lds_read_b128(unsigned addr) {
  bf16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}
__device__ __forceinline__ void beside_glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// `smem`: the workgroup's dynamic LDS (at LDS offset 0, >= kBesideLdsBytes); worker = this workgroup's index
// among the nworkers workgroups that share the GEMM (workgroups of one XCD should have consecutive indices).
// g.counter != null: tiles are handed out by an atomic counter (one word per GEMM, zeroed by the host), so workgroups
// of different speed -- including recurrence workgroups that have finished their chunk -- share the work evenly; the
// word at smem + kBesideLdsBytes (the caller allocates kBesideLdsBytes + 64) broadcasts the claimed tile.
__device__ __forceinline__ void beside_gemm_tiles(const BesideGemm& g, char* smem, unsigned worker, unsigned nworkers) {
  constexpr int NSTAGE = kBesideStages;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t M = g.M, N = g.N, K = g.K;
  const unsigned ntn = (unsigned)((N + 127) / 128);
  const unsigned ntiles = ntn * (unsigned)((M + 255) / 256);
  const int nk = (int)(K / 64);
  const unsigned sw = (unsigned)((lane & 15) >> 1);
  const unsigned a_base = (unsigned)((wm * 128 + (lane & 15)) * 128) + ((((unsigned)lane >> 4) ^ sw) << 4);
  const unsigned b_base = 32768u + (unsigned)((wn * 64 + (lane & 15)) * 128) + ((((unsigned)lane >> 4) ^ sw) << 4);

  // (an LDS-address-space pointer: a generic one made this a flat_store / flat_load pair, and FLAT instructions retire
  // out of order on the counters the counted waits below rely on)
  typedef __attribute__((address_space(3))) unsigned lds_u32;
  lds_u32* const claim = (lds_u32*)(__attribute__((address_space(3))) void*)(smem + kBesideLdsBytes);
  for (unsigned lid = worker;; lid += nworkers) {
    __syncthreads();     // every wave has left the previous tile's last stage (and has read the previous claim)
    if (g.counter != nullptr) {
      if (tid == 0) *claim = __hip_atomic_fetch_add(g.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      lid = *claim;
    }
    if (lid >= ntiles) break;
    const int64_t m0 = (int64_t)(lid / ntn) * 256, n0 = (int64_t)(lid % ntn) * 128;
    // staging: a 1 KB instruction fills 8 rows x 8 chunks; A has 32 per stage (8 per wave), B 16 (4 per wave);
    // lane -> row r = lane >> 3, LDS chunk position c = lane & 7 <- global chunk c ^ ((row >> 1) & 7)
    const int srow = lane >> 3;
    auto issue = [&](int kt) {
      char* a_s = smem + (kt % NSTAGE) * kBesideStageBytes;
      char* b_s = a_s + 32768;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = (4 * i + wave) * 8 + srow;
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        int64_t am = m0 + row;
        am = am < M ? am : M - 1;
        // (a_blocked: A in the recurrence's fragment-major layout -- the same 16-byte pieces at other addresses)
        const bf16_t* asrc = g.a_blocked ? g.A + blk_offset(am, (int64_t)kt * 64 + ch * 8, K)
                                         : g.A + am * K + (int64_t)kt * 64 + ch * 8;
        beside_glds16(asrc, a_s + (4 * i + wave) * 1024);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = (4 * i + wave) * 8 + srow;
        const int ch = (lane & 7) ^ ((row >> 1) & 7);
        int64_t bn = n0 + row;
        bn = bn < N ? bn : N - 1;
        beside_glds16(g.Bt + bn * K + (int64_t)kt * 64 + ch * 8, b_s + (4 * i + wave) * 1024);
      }
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int h = 0; h < NSTAGE - 1; ++h)
      if (h < nk) issue(h);
    for (int kt = 0; kt < nk; ++kt) {
      const int ahead = nk - 1 - kt;            // stages issued after kt: min(NSTAGE - 2, ahead) stay in flight
      if (ahead >= NSTAGE - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 * (NSTAGE - 2)) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (kt + NSTAGE - 1 < nk) issue(kt + NSTAGE - 1);
      const unsigned sb = (unsigned)(kt % NSTAGE) * kBesideStageBytes;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const unsigned xo = kk ? 64u : 0u;
        bf16x8 af[8], bfr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bfr[j] = beside_lds_read_b128(((b_base ^ xo) + sb) + j * 2048u);
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = beside_lds_read_b128(((a_base ^ xo) + sb) + i * 2048u);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          switch (7 - i) {   // LDS reads complete in order: A tile i is ready once at most 7 - i younger reads are out
            case 7: asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory"); break;
            default: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }

#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t m = m0 + wm * 128 + i * 16 + (lane & 15);
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
        if (n + 3 < N) {
          float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
          if (g.bias != nullptr) {
            const float4 bz = *reinterpret_cast<const float4*>(g.bias + n);
            o.x += bz.x; o.y += bz.y; o.z += bz.z; o.w += bz.w;
          }
          if (g.c_bf16)
            *reinterpret_cast<bf16x4*>(reinterpret_cast<bf16_t*>(g.C) + m * N + n) = (bf16x4){(bf16_t)o.x, (bf16_t)o.y, (bf16_t)o.z, (bf16_t)o.w};
          else
            *reinterpret_cast<float4*>(g.C + m * N + n) = o;
        } else if (g.c_bf16) {
          for (int r = 0; r < 4; ++r)
            if (n + r < N) reinterpret_cast<bf16_t*>(g.C)[m * N + n + r] = (bf16_t)(acc[i][j][r] + (g.bias != nullptr ? g.bias[n + r] : 0.0f));
        } else {
          for (int r = 0; r < 4; ++r)
            if (n + r < N) g.C[m * N + n + r] = acc[i][j][r] + (g.bias != nullptr ? g.bias[n + r] : 0.0f);
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace csn
