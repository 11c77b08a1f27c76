// Argument blocks of the multi-problem cell kernels (lstm_cell_blk.hip) and their launchers.
#pragma once
#include "csn_common.h"

namespace csn {

struct CellFwdProb {
  const bf16_t* h_prev_blk;  // fragment-major [Bpad, H] or null (zero state)
  const bf16_t* w_blk;       // fragment-major W_hh, interleaved rows [4H, H]
  const float* xproj;        // [B, 4H] interleaved columns (x W_ih^T + b), row stride 4H
  const float* c_prev;       // [B, H] or null
  bf16_t* gates_out;         // [B, 4H] interleaved, or null
  float* c_out;              // [B, H]
  bf16_t* h_out;             // [B, H] row-major
  bf16_t* h_out_blk;         // fragment-major [Bpad, H]
};
struct CellFwdArgs {
  CellFwdProb p[4];
  int B, H;
};

struct CellBwdProb {
  const bf16_t* dg_next_blk;  // fragment-major [Bpad, 4H'] or null
  const bf16_t* wt_blk;       // fragment-major W_hh^T [H, 4H'] (k interleaved)
  const float* dy;            // [B, H] (row stride dy_ld) or null
  int64_t dy_ld;
  const bf16_t* gates;        // [B, 4H] interleaved
  const float* c;             // [B, H]
  const float* c_prev;        // [B, H] or null
  float* dc_carry;            // [B, H] in/out
  bf16_t* dg_out;             // [B, 4H] interleaved, row-major
  bf16_t* dg_out_blk;         // fragment-major [Bpad, 4H']
};
struct CellBwdArgs {
  CellBwdProb p[4];
  int B, H;
};

// weight-stationary forward over a chunk of timesteps of ONE layer (lstm_fwd_persist.hip)
struct PersistFwdArgs {
  const bf16_t* w_blk;     // fragment-major W_hh, interleaved rows [4H, H]
  const float* xproj;      // [T, B, 4H] interleaved
  bf16_t* gates;           // [T, B, 4H] interleaved, or null
  float* c_all;            // [T+1, B, H]  (slot t+1 = c_t)
  bf16_t* h_all;           // [T+1, B, H]  row-major
  bf16_t* h_blk_all;       // [T+1][Bpad * H] fragment-major slabs (slot t+1 = h_t); never reused in a forward
  unsigned* counters;      // [T+1][MT] arrivals per (slot, M-tile), zeroed before the first chunk
  unsigned* error_flag;    // sticky: a bounded spin gave up
  int B, H, T, t0, nsteps, Bpad;
};
bool fwd_persist_supported(int B, int H, int dtype);
int launch_fwd_persist(const PersistFwdArgs& a, hipStream_t st);

bool cell_blk_supported(int H, int dtype);
int launch_cell_fwd_il(const CellFwdArgs& a, int nprob, hipStream_t st);
int launch_cell_bwd_il(const CellBwdArgs& a, int nprob, hipStream_t st);
int launch_blockify(const float* src, int64_t ld_r, int64_t ld_k, int64_t R, int64_t K, int perm_r, int perm_k,
                    int64_t H, void* dst, hipStream_t st);
int launch_permute_rows_cast(const float* src, int64_t H, int64_t I, void* dst, hipStream_t st);
int launch_transpose_perm_cast(const float* src, int64_t H, int64_t I, void* dst, hipStream_t st);
int launch_bias_perm_sum(const float* a, const float* b, int64_t H, float* dst, hipStream_t st);
int launch_reduce_slabs_unperm(const float* slabs, int64_t slab_stride, int S, int64_t H, int64_t C, float* out,
                               hipStream_t st);

}  // namespace csn
