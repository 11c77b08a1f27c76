/*
 * csn_hip.h -- C ABI of libcsn_hip.so: the MI355X (gfx950) implementation of the
 * EEG -> stacked-LSTM -> distillation hot path of Vi-Sri/CerebralSignalNetworks.
 *
 * The reference is pure Python and has no FFI/plugin interface of its own (SURVEY.md
 * section 8b): its boundary is a set of Python call sites that reach third-party native code
 * (scipy.signal, torch.nn.LSTM/ATen, faiss).  Each entry point below replaces one of
 * those call sites and cites it.  Signatures carry only plain pointers and sizes:
 * every `const T*` / `T*` is a DEVICE pointer unless marked [host]; `stream` is a
 * hipStream_t passed as void* (NULL = the default stream).  All work is enqueued on
 * `stream`; nothing synchronises the device.  Return value: 0 on success, non-zero
 * csnStatus otherwise, with a message in csn_last_error().  Shape / alignment
 * violations are rejected on the host before any launch.
 *
 * Tensors are dense row-major unless strides are given.  dtype codes: CSN_F32, CSN_BF16.
 */
#ifndef CSN_HIP_H
#define CSN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* csnStream_t;

enum csnStatus {
  CSN_OK = 0,
  CSN_ERR_INVALID_ARGUMENT = 1,
  CSN_ERR_HIP = 2,
  CSN_ERR_UNSUPPORTED = 3
};

enum csnDtype { CSN_F32 = 0, CSN_BF16 = 1 };

/* ABI version of this header; bumped on any signature change. */
#define CSN_ABI_VERSION 5
int csn_abi_version(void);
/* Thread-local message for the last non-zero status returned on this thread. */
const char* csn_last_error(void);
/* Name of the device code object's target ("gfx950"). */
const char* csn_target_arch(void);

/* ------------------------------------------------------------------------------------
 * K1+K2  fused band-pass + per-channel z-score.
 * Replaces: scipy.signal.butter/lfilter design+apply named by utils/EEGFilters.py:2,26
 * (applied causally, in second-order sections) followed by EEGDataset.normlizeEEG,
 * utils/PerilsEEGDataset.py:454-461, for every channel of a segment, and the
 * `.t()` re-layout of EEGDataset.__getitem__, utils/PerilsEEGDataset.py:549.
 *   x      [B,C,T] float32 (channel-first, as stored on disk: ConvertToPth.py:170-201)
 *   sos    [host] [nsec,6] float64 rows (b0,b1,b2,a0,a1,a2), nsec <= 8; nsec == 0 = no filter
 *   ddof   0 (numpy path, PerilsEEGDataset.py:555-562) or 1 (torch path, :576-579)
 *   y      out_dtype, laid out [B,T,C] (time_major=0) or [T,B,C] (time_major=1)
 * IIR state and statistics are carried in float64.  Stateless: what depends on the coefficients alone (the
 * chunk-scan basis) is computed on the host per call and travels with the kernel arguments -- no device buffer is
 * kept between calls, so calls with different coefficients on different streams / threads / devices are independent.
 * ---------------------------------------------------------------------------------- */
int csn_eeg_bandpass_znorm(const float* x, int B, int C, int T,
                           const double* sos, int nsec, int ddof,
                           void* y, int out_dtype, int time_major, csnStream_t stream);

/* Zero-phase variant.  Replaces: signal.filtfilt(b, a, eeg[s, :, c]) for every (s, c) in
 * Utilities.remove_noise, utils/Utilities.py:411-428 (Butterworth order 4, 1-50 Hz; scipy defaults:
 * odd extension, padlen = 3*max(len(a),len(b)) = 3*(2*nsec+1), steady-state initial conditions),
 * evaluated on the second-order-section cascade in float64.
 *   x, y    [S,T,C] float32 (samples x time x channels, the layout remove_noise takes), T > padlen
 *   scratch csn_eeg_filtfilt_scratch_bytes(S,T,C,nsec) bytes of device memory */
size_t csn_eeg_filtfilt_scratch_bytes(int S, int T, int C, int nsec);
int csn_eeg_filtfilt(const float* x, int S, int T, int C, const double* sos, int nsec,
                     float* y, void* scratch, csnStream_t stream);

/* ------------------------------------------------------------------------------------
 * K3  stacked LSTM, zero initial state, gate order i,f,g,o, nn.LSTM parameter layout.
 * Replaces: nn.LSTM(input, hidden, num_layers, batch_first=True) forward/backward at
 * LSTMDistill.py:118,132 and LSTMDistillRetreival.py:91,103 (the body of the absent
 * models.lstm.Model, LstmDistillFromDinoV2Train.py:323).
 * ---------------------------------------------------------------------------------- */
typedef struct csnLstmDesc {
  int32_t B;      /* batch                                       */
  int32_t T;      /* time steps                                  */
  int32_t I;      /* input features (EEG channels)               */
  int32_t H;      /* hidden size; multiple of 32                 */
  int32_t L;      /* stacked layers, 1..8                        */
  int32_t dtype;  /* CSN_BF16: bf16 MFMA operands, f32 accumulate/state; CSN_F32: exact f32 MFMA */
} csnLstmDesc;

/* A PLAN holds everything host-side that a stacked-LSTM problem of one shape needs: the workspace layout, the
 * diagnostic switches (environment, read once here), library-owned side streams, an event pool, profiling events.
 * The library keeps NO mutable global state: plans are independent of each other, so different host threads /
 * streams / devices use different plans freely.  One plan must not be used from two threads at once, and it is
 * bound to the device that was current when it was created.  training != 0: the forward keeps what
 * csn_lstm_backward needs.  (There is nothing like this in the reference: torch's nn.LSTM hides the same state in
 * cuDNN/MIOpen descriptors and the autograd graph.) */
typedef struct csnLstmPlan csnLstmPlan;
int csn_lstm_plan_create(const csnLstmDesc* d, int training, csnLstmPlan** out);
void csn_lstm_plan_destroy(csnLstmPlan* plan);
/* Bytes of device scratch ("workspace") a forward (+ backward) of this plan needs; 256-B aligned base.  The
 * caller owns it; one plan may be used with several workspaces (e.g. several forwards awaiting their backward). */
size_t csn_lstm_plan_workspace_bytes(const csnLstmPlan* plan);
/* Which kernels the plan runs: 0 generic per-step cells (exact f32 / odd shapes), 1 per-diagonal bf16 launches,
 * 2 weight-stationary forward, 3 weight-stationary forward and backward (bf16), 4 weight-stationary forward and
 * backward of the exact-float32 path (CSN_F32, H in {128, 256, 384, 512, 768, 1024}, a whole MI355X: one launch per layer and
 * block of batch rows; its workspace also holds fragment-major copies of h and of the gate gradients). */
int csn_lstm_plan_path(const csnLstmPlan* plan);
/* Copies of the gate gradients the plan's LAST csn_lstm_backward wrote per step: 2 = the fragment-major hand-off slab
 * and a row-major copy for the GEMMs behind the recurrence (always, in this library); 1 = the hand-off slabs alone,
 * read in place by those GEMMs (experiments library under CSN_BWD_SINGLE_COPY); 0 = no backward has
 * run, or a path without hand-off slabs.  Diagnostic: lets a test see which form it compared. */
int csn_lstm_plan_dgates_copies(const csnLstmPlan* plan);
/* Name of the device function that advances the recurrence on this plan's path: which = 0 forward, 1 backward
 * ("lstm_fwd_persist_kernel", "lstm_fwd_ns_kernel", "lstm_bwd_persist_kernel", "lstm_cell_fwd_il_kernel", ... -- the
 * names a rocprofv3 kernel trace shows, without template arguments).  Diagnostic: bench.py labels its roofline object
 * and looks up the committed counter passes with it.  Static storage; NULL for a null plan / other `which`. */
const char* csn_lstm_plan_kernel_name(const csnLstmPlan* plan, int which);
/* Same number without a plan (what csn_lstm_plan_workspace_bytes would return for a plan created now). */
size_t csn_lstm_workspace_bytes(const csnLstmDesc* d, int training);

/* Once per workspace, before its first forward (enqueued on `stream`): zeroes the status word and the regions the
 * kernels only ever read as zero (the zero initial state h_0 / c_0 of every layer, ...).  A forward / backward
 * re-zeroes per call only what it dirties (flag lines, carried dc). */
int csn_lstm_workspace_init(const csnLstmPlan* plan, void* workspace, csnStream_t stream);

/* x: element (b,t,i) at x[b*x_stride_b + t*x_stride_t + i] (float32).
 * w_ih/w_hh/b_ih/b_hh: [host] arrays of L device pointers to float32 parameters
 *   weight_ih_l{k}[4H,I_k], weight_hh_l{k}[4H,H], bias_ih_l{k}[4H], bias_hh_l{k}[4H].
 * y_last: [B,H] float32 = output of the top layer at t = T-1.
 * y_all : optional (may be NULL) [B,T,H] float32, every step of the top layer. */
int csn_lstm_forward(csnLstmPlan* plan,
                     const float* x, int64_t x_stride_b, int64_t x_stride_t,
                     const float* const* w_ih, const float* const* w_hh,
                     const float* const* b_ih, const float* const* b_hh,
                     void* workspace, float* y_last, float* y_all, csnStream_t stream);

/* dy_last: [B,H] float32 gradient w.r.t. y_last (may be NULL).
 * dy_all : optional [B,T,H] float32 gradient w.r.t. y_all (may be NULL).
 * dw_ih/dw_hh/db_ih/db_hh: [host] arrays of L device pointers, float32, OVERWRITTEN.
 * dx: optional [B,T,I] float32 (dense), gradient w.r.t. x (may be NULL). */
int csn_lstm_backward(csnLstmPlan* plan,
                      const float* dy_last, const float* dy_all,
                      void* workspace,
                      float* const* dw_ih, float* const* dw_hh,
                      float* const* db_ih, float* const* db_hh,
                      float* dx, csnStream_t stream);

/* Gradient-ready notification (data-parallel training: the reference gets the overlap of its gradient all-reduce with
 * the backward from DistributedDataParallel's autograd hooks, LstmDistillation.py:445; this is the same hook at the C
 * boundary).  csn_lstm_backward calls fn(user, layer) on the CALLING host thread, once per layer, top layer first,
 * each time at a point where every kernel that writes dw_ih/dw_hh/db_ih/db_hh of `layer` has been enqueued on (or
 * ordered before) `stream`: work the callback enqueues behind `stream` -- e.g. an all-reduce of that layer's gradients
 * on a communication stream that waits on an event recorded there -- then runs beside the remaining layers' weight-
 * gradient GEMMs.  The weight-stationary recurrence launches (one workgroup per CU, all co-resident) are all enqueued
 * BEFORE the first call, so nothing the callback starts can share the device with them.  The callback must not call
 * back into this plan.  fn = NULL removes it. */
typedef void (*csnGradReadyFn)(void* user, int layer);
int csn_lstm_plan_set_grad_callback(csnLstmPlan* plan, csnGradReadyFn fn, void* user);

/* The workspace's status word: 0 = ok.  Bit CSN_STATUS_TIMEOUT: a bounded in-kernel wait of a weight-stationary
 * kernel gave up at some point since the word was last cleared (the results of that forward / backward and of every
 * later one are invalid).  Bit CSN_STATUS_NONFINITE: a NaN / Inf gradient reached the backward recurrence (the
 * operand was proven to be data, its product was not finite): the gradients are non-finite exactly as the reference's
 * autograd would leave them -- a diverged run, not a device fault.  Only the weight-stationary backward with the
 * hand-off by data (csn_lstm_plan_path() == 3, not under CSN_BWD_FLAGS) inspects its operands and can raise this bit;
 * on every other path (exact-f32, odd shapes, sequences too long for the slab ring) non-finite gradients simply
 * propagate into dw / dx, and a caller that wants the check on every path tests its gradient buffer itself
 * (trainer.check_device_status does).  It is STICKY: no forward or backward clears it.  csn_lstm_status_clear zeroes it (enqueued on
 * `stream`): call it once after allocating a workspace and after a reported error has been handled.
 * csn_lstm_status_read is a blocking device -> host read.  csn_lstm_status_raise is fault injection for tests of
 * the error path: it leaves the word exactly as a timed-out wait does (every later bounded wait then returns at
 * once, so nothing hangs; results are garbage by construction). */
#define CSN_STATUS_TIMEOUT 1
#define CSN_STATUS_NONFINITE 2
#define CSN_STATUS_STALE_SLOT 4   /* debug library (make tags) only: a hand-off ring slot served its previous occupant */
int csn_lstm_status_clear(const csnLstmPlan* plan, void* workspace, csnStream_t stream);
int csn_lstm_status_read(const csnLstmPlan* plan, const void* workspace, int* status);
int csn_lstm_status_raise(const csnLstmPlan* plan, void* workspace, csnStream_t stream);

/* Optional timing of the plan's recurrence kernels with HIP events recorded on the caller's stream in its most
 * recent forward / backward: around every weight-stationary launch (the time reported is the sum over the
 * launches, the GEMMs between them excluded), or around the whole launch loop of the per-timestep cell
 * kernels.  csn_lstm_profile_read synchronises on those events; *_launches = recurrence launches;
 * *_cells = cell problems (layer-steps) those launches advanced. */
int csn_lstm_profile_enable(csnLstmPlan* plan, int on);
int csn_lstm_profile_read(csnLstmPlan* plan, double* fwd_ms, int* fwd_launches, int* fwd_cells,
                          double* bwd_ms, int* bwd_launches, int* bwd_cells);

/* ------------------------------------------------------------------------------------
 * Building blocks of K3, exported so that each can be parity-tested on its own.
 * ---------------------------------------------------------------------------------- */
/* C[M,N] (+)= A[M,K] * Bt[N,K]^T (+ bias[N]).  dtype = type of A and Bt; C is out_dtype.
 * accumulate != 0 adds into C (float32 C only).  Replaces the input-projection /
 * input-gradient GEMMs inside ATen's LSTM. */
int csn_gemm_nt(const void* A, const void* Bt, const float* bias, void* C,
                int64_t M, int64_t N, int64_t K, int dtype, int out_dtype, int accumulate,
                csnStream_t stream);
/* C[M,N] = A[K,M]^T * B[K,N]  (weight-gradient form; float32 C, overwritten).
 * scratch: device buffer of csn_gemm_tn_scratch_bytes(M,N,K) bytes (split-K slabs). */
size_t csn_gemm_tn_scratch_bytes(int64_t M, int64_t N, int64_t K);
int csn_gemm_tn(const void* A, const void* B, float* C,
                int64_t M, int64_t N, int64_t K, int dtype, void* scratch, csnStream_t stream);

/* One LSTM cell step.  h_prev/h_out/gates are `dtype`; xproj (= x_t W_ih^T + b_ih + b_hh),
 * c_prev, c_out float32.  gates_out (may be NULL) receives post-activation i,f,g,o [B,4H]. */
int csn_lstm_cell_forward(const void* h_prev, const void* w_hh, const float* xproj, int64_t xproj_ld,
                          const float* c_prev, void* gates_out, float* c_out, void* h_out,
                          int B, int H, int dtype, csnStream_t stream);
/* One backward cell step: dh = dy + dgates_next * W_hh (w_hh_t = W_hh^T [H,4H]); writes
 * dgates_out [B,4H] and updates dc_carry [B,H] in place.  dgates_next / dy may be NULL. */
int csn_lstm_cell_backward(const void* dgates_next, const void* w_hh_t,
                           const float* dy, int64_t dy_ld,
                           const void* gates, const float* c, const float* c_prev,
                           float* dc_carry, void* dgates_out,
                           int B, int H, int dtype, csnStream_t stream);

/* ------------------------------------------------------------------------------------
 * K5  1 - mean_b cos(student_b, teacher_b), dim=1, eps=1e-8, and its gradient.
 * Replaces: CosineSimilarityLoss.forward, LstmDistillFromDinoV2Train.py:36-43.
 *   loss: [1] float32; dstudent: optional [B,D] float32 = grad_scale * dloss/dstudent.
 *   scratch: csn_cosine_loss_scratch_bytes(B) bytes of device memory owned by the caller (the per-row cosines; 8-byte
 *   aligned) -- like csn_l2_topk's: the library holds no buffer of its own between calls.
 * ---------------------------------------------------------------------------------- */
size_t csn_cosine_loss_scratch_bytes(int B);
int csn_cosine_loss(const float* student, const float* teacher, int B, int D,
                    float* loss, float* dstudent, float grad_scale, void* scratch, csnStream_t stream);

/* ------------------------------------------------------------------------------------
 * Optimiser step of the hot loop over ONE flat float32 parameter / gradient / state buffer (16-byte aligned).
 * Replaces: torch.optim.RMSprop(model.parameters(), lr=...).step(), LstmDistillFromDinoV2Train.py:329,373, with
 * that call's defaults: square_avg <- alpha square_avg + (1 - alpha) g^2 ; p <- p - lr g / (sqrt(square_avg) + eps)
 * (alpha 0.99, eps 1e-8; no momentum, not centred, no weight decay).
 * ---------------------------------------------------------------------------------- */
int csn_rmsprop_step(float* params, const float* grads, float* square_avg, int64_t n,
                     float lr, float alpha, float eps, csnStream_t stream);

/* ------------------------------------------------------------------------------------
 * K7  Barlow-Twins reduction over the cross-correlation matrix c[D,D] (float32):
 *   out[0] = sum_i (c_ii - 1)^2,  out[1] = sum_{i!=j} c_ij^2      (float32[2])
 * Replaces: EEG-BarlowNetworks/net.py:6-9,39-40.
 * ---------------------------------------------------------------------------------- */
int csn_barlow_offdiag_sqsum(const float* c, int D, float* out, csnStream_t stream);

/* ------------------------------------------------------------------------------------
 * K8  exact squared-L2 top-k.  Replaces: faiss.IndexFlatL2(d).add / .search(k),
 * utils/Utilities.py:45-55.  gallery [Ng,D], query [Nq,D] float32; out_idx [Nq,k]
 * int64, out_dist [Nq,k] float32, ascending, ties -> lower gallery index.
 * scratch: device buffer of csn_l2_topk_scratch_bytes(Ng,Nq) bytes.  k <= 64.
 * ---------------------------------------------------------------------------------- */
size_t csn_l2_topk_scratch_bytes(int64_t Ng, int64_t Nq);
int csn_l2_topk(const float* gallery, const float* query, int64_t Ng, int64_t Nq, int D, int k,
                int64_t* out_idx, float* out_dist, void* scratch, csnStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CSN_HIP_H */
