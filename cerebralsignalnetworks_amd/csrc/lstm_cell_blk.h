// Argument blocks of the multi-problem cell kernels (lstm_cell_blk.hip) and their launchers.
#pragma once
#include "csn_common.h"

// Debug library (-DCSN_SLAB_TAGS, `make tags`): bit 2 of data_polls = fault injection "do not re-arm the ring slots", so
// that consumers ARE served stale occupants and the tag check can be seen to fire.  The product build has neither.
#ifdef CSN_SLAB_TAGS
#define CSN_DPOLL_MODE(x) ((x) & 3)
#define CSN_DPOLL_NO_REARM(x) (((x) & 4) != 0)
#else
#define CSN_DPOLL_MODE(x) (x)
#define CSN_DPOLL_NO_REARM(x) false
#endif

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

// weight-stationary forward (lstm_fwd_persist.hip): ONE launch advances up to 4 layers, each through its
// own chunk of timesteps (a wavefront diagonal over chunks).
struct PersistFwdSlot {
  const bf16_t* w_blk;     // fragment-major W_hh, interleaved rows [4H, H]
  const float* xproj;      // [T, B, 4H] interleaved (float32; bf16 when xproj_bf16)
  bf16_t* gates;           // [T, B, 4H] interleaved, or null
  float* c_all;            // [T+1, B, H]  (slot t+1 = c_t)
  bf16_t* h_all;           // [T+1, B, H]  row-major
  bf16_t* h_blk_all;       // [T+1][Bpad * H] fragment-major slabs (slot t+1 = h_t); never reused in a forward
  unsigned* flags;         // [T+1][MT][kPersistFlagLine]: word i of line (t, mt) != 0 once slice i has published h_{t-1}; zeroed per forward
  // fused input projection (layer 0, I % 32 == 0, I <= 128): xproj is ignored, the workgroup keeps its W_ih rows
  // in registers too and multiplies x_t itself
  const bf16_t* x_blk;     // [T][Bpad * I] fragment-major slabs of the input, or null
  const bf16_t* wih_blk;   // fragment-major W_ih, interleaved rows [4H, I]
  const float* bias;       // [4H] interleaved b_ih + b_hh
  int I;
  int t0, nsteps;
  int xproj_bf16;          // N-split kernel only: xproj holds bf16 (written by the GEMM the previous launch carried)
};
// C[M,N] (f32) = A[M,K] * Bt[N,K]^T (+ bias[N]), bf16 operands: run by the workgroups of a weight-stationary launch
// that have no recurrence work (gemm_beside.h)
struct BesideGemm {
  const bf16_t* A;
  const bf16_t* Bt;
  float* C;
  int M, N, K;
  const float* bias;       // [N] added to every row, or null
  unsigned* counter;       // atomic tile counter (zeroed by the host), or null: tiles dealt round-robin to the workers
  int c_bf16;              // != 0: C is bf16 (the forward's input projection, consumed only by lstm_fwd_ns.hip)
  int a_blocked;           // != 0: A is in the fragment-major 16 x 32 block layout (blk_offset; M % 16 == 0), not row-major
};
static constexpr int kPersistFlagLine = 32;    // one 128-byte line per (slot, M-tile): at most 32 slices
struct PersistFwdArgs {
  PersistFwdSlot slot[4];
  int nslots;
  // N-split kernel only (lstm_fwd_ns.hip): input-projection GEMMs of the chunks the layers below finished one
  // launch ago, walked by the workgroups of the launch that have no recurrence work; the grid is 8 * grid_slices
  BesideGemm gemm[3];
  int ngemm;
  int grid_slices;
  int half_tiles;          // K2 x N2 body pipelined over 32-row halves: the flag lines are [T+1][MT][2][line]
  int data_polls;          // K-split kernel: hand-off by sentinel data in a ring of 4 slabs (no flags, no store drain); the host fills the 4 slabs with 0xff per forward
  int chains;              // 4: the wave-specialised body (lstm_fwd_ws.hip), four 16-row chains per tile: flag lines [T+1][MT][4][line]
  // xcd_groups != 0: 1-D grid of 8 * nslices workgroups; the workgroups that share (blockIdx.x % 8) form one
  // hand-off group (a slot's M-tile) -- under the round-robin dispatch they share an XCD, which each group
  // verifies at run time through agree[group] (zeroed, one set of 8 words per launch) before it uses the
  // L2-local hand-off.  xcd_groups == 0: groups are contiguous block ranges, placement-independent hand-off.
  int xcd_groups;
  int rotate;              // != 0: each workgroup walks the k-blocks from its own offset (changes the summation order)
  unsigned long long* agree;
  unsigned* error_flag;    // sticky: a bounded spin gave up
  int B, H, T, Bpad, MT;
};
bool fwd_persist_supported(int B, int H, int dtype, const Options& opt);
// N-split weight-stationary forward (lstm_fwd_ns.hip): the default where it applies
bool fwd_ns_supported(int B, int H, int dtype, const Options& opt);
int fwd_ns_slices(int H);
int launch_fwd_ns(const PersistFwdArgs& a, hipStream_t st);
int launch_fwd_ws(const PersistFwdArgs& a, hipStream_t st);   // H = 768, a.chains == 4
int fwd_persist_slices(int H);   // workgroups per hand-off group
int launch_fwd_persist(const PersistFwdArgs& a, hipStream_t st);

// weight-stationary backward recurrence (lstm_bwd_persist.hip): ONE launch walks up to 4 layers, each backwards
// through its own chunk of timesteps; grouping / hand-off as in PersistFwdArgs.
struct PersistBwdSlot {
  const bf16_t* wt_blk;    // fragment-major W_hh^T [H, 4H'] (k interleaved)
  const bf16_t* gates;     // [T, B, 4H] interleaved (saved by the forward)
  const float* c_all;      // [T+1, B, H]  (slot t+1 = c_t)
  const float* dy;         // [T, B, H] gradient w.r.t. this layer's output, or null
  const float* dy_last;    // [B, H] added at t = T-1 when dy is null, or null
  const float* zeros;      // [B, H] of zeros: what a step reads when it has no incoming gradient
  float* dc_carry;         // [B, H] carried dc, in/out across launches
  bf16_t* dgates;          // [T, B, 4H] interleaved, row-major (GEMM operand)
  bf16_t* dg_blk_all;      // [T][Bpad * 4H] fragment-major slabs (slot t = dgates_t); never reused in a backward
  unsigned* flags;         // [T][MT][kPersistFlagLine], zeroed per backward
  int t_hi, nsteps;        // steps t_hi, t_hi - 1, ..., t_hi - nsteps + 1
};
struct PersistBwdArgs {
  PersistBwdSlot slot[4];
  int nslots;
  BesideGemm gemm[3];      // input-gradient GEMMs of the chunks the layers above finished one launch ago
  int ngemm;
  int grid_slices;         // xcd_groups: the grid is 8 * grid_slices workgroups (>= slices per group)
  int xcd_groups;
  int rotate;              // != 0: each workgroup walks the k-blocks from its own offset (changes the summation order)
  int data_polls;          // hand-off by sentinel data in a ring of 4 slabs (the host fills them with 0xff per backward)
  int single_copy;         // (experiments library) != 0: ONE copy of dgates -- every step's hand-off slab has an address of its own (dg_blk_all[t],
                           // sentinel-armed two steps ahead by its producer) and IS what the weight- and input-gradient
                           // GEMMs read; the row-major copy is not written (4 of the 12 store instructions of a step)
  unsigned long long* agree;
  unsigned* error_flag;
  int B, H, T, Bpad, MT;
};
bool bwd_persist_supported(int B, int H, int dtype, const Options& opt);
int bwd_persist_slices(int H);
int launch_bwd_persist(const PersistBwdArgs& a, hipStream_t st);

// one launch for all layout-preparation jobs of a forward (lstm_cell_blk.hip: prep_multi_kernel)
enum { kPrepBlockify = 0, kPrepPermRows, kPrepTransPerm, kPrepBias, kPrepCastX, kPrepBlockifyX };
static constexpr int kPrepMaxJobs = 32;
struct PrepJob {
  int kind;
  const float* a;
  const float* b;
  void* dst;
  int64_t n0, n1, n2, s0, s1, H;     // meaning per kind (see the kernel)
  int perm_r, perm_k;
  int64_t work;                      // work items (threads) of the job
  unsigned blk_begin, blk_count;     // filled by the launcher
};
struct PrepArgs {
  PrepJob job[kPrepMaxJobs];
  int njobs;
};
int launch_prep_multi(PrepArgs& A, hipStream_t st);

bool cell_blk_supported(int H, int dtype, const Options& opt);
int launch_cell_fwd_il(const CellFwdArgs& a, int nprob, hipStream_t st, int max_nk);
int launch_cell_bwd_il(const CellBwdArgs& a, int nprob, hipStream_t st);
int launch_blockify_x(const float* x, int64_t xsb, int64_t xst, int B, int T, int I, void* dst, hipStream_t st);
int launch_blockify(const float* src, int64_t ld_r, int64_t ld_k, int64_t R, int64_t K, int perm_r, int perm_k,
                    int64_t H, void* dst, hipStream_t st);
int launch_permute_rows_cast(const float* src, int64_t H, int64_t I, void* dst, hipStream_t st);
int launch_transpose_perm_cast(const float* src, int64_t H, int64_t I, void* dst, hipStream_t st);
int launch_bias_perm_sum(const float* a, const float* b, int64_t H, float* dst, hipStream_t st);
int launch_reduce_slabs_unperm(const float* slabs, int64_t slab_stride, int S, int64_t H, int64_t C, float* out,
                               hipStream_t st);

}  // namespace csn
