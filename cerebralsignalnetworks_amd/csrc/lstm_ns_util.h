// Device helpers shared by the N-split family of weight-stationary forward kernels (lstm_fwd_ns.hip, lstm_fwd_ws.hip).
#pragma once
#include "csn_common.h"

namespace csn {

static constexpr unsigned long long kNsSpinTimeoutTicks = 20000000ull;   // 0.2 s of the 100 MHz wall clock
static constexpr int kNsStageBytes = 64 * (288 + 144 + 80);               // padded rows of gates (bf16 x 4) + c (f32) + h (bf16) of 64 x 32 cells

typedef __attribute__((ext_vector_type(4))) unsigned nu32x4;
typedef __attribute__((ext_vector_type(2))) unsigned nu32x2;

// (OFF: instruction offset, added to the global AND the LDS address: four consecutive 1 KB pieces share one M0 / soffset;
// AUX: cache policy bits of the load -- 16 = sc1 for hand-off data, 0 = default for data that should stay in L2)
template <int OFF = 0, int AUX = 16>
__device__ __forceinline__ void ns_dma16_sc1(__amdgpu_buffer_rsrc_t rsrc, void* lds, int voffset, int soffset) {
  // buffer_load_dwordx4 ... lds: 16 bytes per lane, global (rsrc + soffset + voffset) -> LDS at lds + lane * 16.
  // Every block offset is wave-uniform (an SGPR), the only VGPR is lane * 16; aux 16 = sc1 (this CU's L1 is bypassed:
  // hand-off data)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voffset, soffset, OFF, AUX);
}
__device__ __forceinline__ f32x4 ns_bload_nt_f32x4(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset) {
  nu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, soffset, 2);     // aux 2 = nt (streamed once)
  return __builtin_bit_cast(f32x4, v);
}
__device__ __forceinline__ bf16x8 ns_lds_read_b128(unsigned addr) {
  bf16x8 v;
#if defined(CSN_NS_ABL) && CSN_NS_ABL >= 2
  v = __builtin_bit_cast(bf16x8, (nu32x4){addr, addr + 1u, addr + 2u, addr + 3u});   // ablation (timing only): no LDS reads
#else
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
#endif
  return v;
}
// The MFMA in inline asm with the stationary operand constrained to the ACCUMULATOR registers.  Left to the compiler,
// the 8 * H/32 weight registers of a wave overflow the 256 architectural VGPRs, are parked in AGPRs as "spills" and
// copied back (8 v_accvgpr_read per k-block) in front of the MFMAs that use them: the MFMA phase of a step ran at 40
// cycles per MFMA instead of 16 (in-kernel stamps with the loads and the LDS reads ablated: unchanged).  CDNA3/4 MFMAs
// read A/B operands from AGPRs directly, so the weights simply LIVE there.  AIA: the accumulator is in AGPRs too
// (H <= 768: 8 * 24 + 32 + 32 <= 256); at H = 1024 the weights alone fill the 256 AGPRs and it stays in VGPRs.
// WV: this stationary operand lives in VGPRs (the fused layer 0 at H = 1024: W_hh fills all 256 AGPRs, so the 32 registers
// of W_ih must NOT carry an "a" constraint -- with one, the compiler kept 8 W_hh fragments in VGPRs and copied each into
// a[32:39] straight in front of its first MFMA; the hazard recogniser does not see an MFMA inside an asm block, so no
// wait state separated `v_accvgpr_write_b32 a35, ...` from the MFMA reading a[32:35]: row group 0 of every tile
// multiplied stale dwords.  tools/check_asm_hazards.py fails the build on that pattern)
template <bool AIA, bool WV = false>
__device__ __forceinline__ void ns_mfma(f32x4& acc, const bf16x8& w, const bf16x8& h) {
#if defined(CSN_NS_HAZARD_DEMO)      // `make hazard_demo` (never shipped): round 3's form for the A/B of DESIGN.md section 3.8 --
  constexpr bool wv = false;        // every stationary operand asks for the accumulator file again
#else
  constexpr bool wv = WV;
#endif
  if constexpr (AIA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(h));
  else if constexpr (wv) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(h));
#if defined(CSN_NS_HAZARD_DEMO) && CSN_NS_HAZARD_DEMO == 2      // ... plus the two wait states the recogniser would have inserted
  else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(h));
#else
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(h));
#endif
}
// (the hazard recogniser does not see inside inline asm: explicit wait states where a VALU result feeds the first MFMA
// of a phase, and where the last MFMA's result is read back -- 16-pass MFMA: up to 18 wait states)
__device__ __forceinline__ void ns_mfma_fence() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }

template <bool WT>
__device__ __forceinline__ void ns_store_b128(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, const nu32x4& v) {
  __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (int)byte_off, 0, WT ? 16 : 0);   // sc1 = write-through
}

// A pointer that is the same in every lane, told to the compiler (it arrives through a dynamically indexed kernel
// argument, which the compiler otherwise keeps in VGPRs: a buffer resource built from it would be "divergent" and every
// buffer instruction wrapped in a waterfall loop)
template <typename T>
__device__ __forceinline__ T* ns_uniform(T* p) {
  const unsigned long long v = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return reinterpret_cast<T*>(((unsigned long long)hi << 32) | lo);
}

// All 32 words of one 128-byte flag line through the SCALAR memory path (s_load_dwordx16 x 2, glc: the scalar cache is
// bypassed, L2 serves it): scalar loads have a queue of their own, so a poll does not wait behind whatever the wave
// has in its vector-memory queue (which returns in order -- HBM input loads ahead of a vector poll add their whole
// latency to it).  Returns the number of non-zero words among the first n.
typedef __attribute__((ext_vector_type(16))) unsigned nu32x16;
__device__ __forceinline__ int ns_flags_set_scalar(const unsigned* line, int n) {
  nu32x16 lo, hi;
  asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc\n\ts_waitcnt lgkmcnt(0)"
               : "=&s"(lo), "=&s"(hi)
               : "s"(line)
               : "memory");
  int c = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    c += (i < n && lo[i] != 0u) ? 1 : 0;
    c += (16 + i < n && hi[i] != 0u) ? 1 : 0;
  }
  return c;
}

template <int N>
__device__ __forceinline__ void ns_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// the counted waits of one DMA group: at most `REST` DMAs of later groups + the P input loads are outstanding
template <int KB, int P>
__device__ __forceinline__ void ns_wait_group(int g) {
  constexpr int GI = KB / 4;
  if (g == 0) ns_wait_vmcnt<3 * GI + P>();
  else if (g == 1) ns_wait_vmcnt<2 * GI + P>();
  else if (g == 2) ns_wait_vmcnt<GI + P>();
  else ns_wait_vmcnt<P>();
}


}  // namespace csn
