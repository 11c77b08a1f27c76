// Weight-stationary recurrence of the exact-float32 path (lstm_f32_persist.hip): one launch walks ALL timesteps of one
// layer for a block of batch rows; W_hh lives in registers, the workgroups of a 64-row tile hand h_t / dgates_t to each
// other through HBM-resident row-major slots + one flag word per producer and step.
#pragma once
#include "csn_common.h"

namespace csn {

static constexpr int kF32FlagLine = 64;      // words per (step, M-tile) flag line: at most 64 producer slices (H <= 1024)

struct F32PersistFwdArgs {
  const float* w_hh;       // [4H][H] (gate-major rows i, f, g, o: the reference's layout)
  const float* xproj;      // [T][B][4H], bias included
  float* gates;            // [T][B][4H] activated gates (training) or null
  float* c_all;            // [T+1][B][H]; slot 0 = c_0
  float* h_all;            // [T+1][B][H]; slot 0 = h_0 = 0 (never read: step 0 skips the recurrent product)
  float* h_blk;            // [T+1][MT_total * 4 row groups][H / 16][64 lanes][4]: fragment-major copy of h, the hand-off payload
  unsigned* flags;         // [T+1][MT_total][kF32FlagLine], zeroed per forward; word s of line (t, mt) != 0: slice s published h_{t-1}
  unsigned* error_flag;    // status word 0: a hand-off timed out
  int B, T;
  int MT_total, mt0, MT;   // M-tiles (64 rows) of the batch; first tile and number of tiles of THIS launch
};

struct F32PersistBwdArgs {
  const float* w_hh_t;     // [H][4H] = W_hh^T
  const float* gates;      // [T][B][4H]
  const float* c_all;      // [T+1][B][H]
  const float* dy;         // [T][B][H] gradient w.r.t. this layer's outputs, or null
  const float* dy_last;    // [B][H] gradient w.r.t. the last output only (dy == null), or null
  const float* zeros;      // [B][H] zeros (steps without an incoming gradient still take ONE unconditional load)
  float* dgates;           // [T][B][4H] pre-activation gradients (row-major: what the GEMMs read)
  float* dg_blk;           // [T][MT_total * 4 row groups][4H / 16][64 lanes][4]: fragment-major copy, the hand-off payload
  float* bias_part;        // [MT_total * 4 row groups][4H]: column sums of dgates over this launch's steps, per 16-row group (or null)
  unsigned* flags;         // [T][MT_total][kF32FlagLine], zeroed per backward
  unsigned* error_flag;
  int B, T;
  int MT_total, mt0, MT;
};

bool f32_persist_supported(int B, int H);
int f32_persist_tiles_per_launch(int H);                       // M-tiles one launch can hold (all its workgroups co-resident)
int launch_fwd_f32_persist(const F32PersistFwdArgs& a, int H, hipStream_t st);
int launch_bwd_f32_persist(const F32PersistBwdArgs& a, int H, hipStream_t st);

}  // namespace csn
