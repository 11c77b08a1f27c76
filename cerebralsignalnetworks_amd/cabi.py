"""ctypes binding of libcsn_hip.so (the C ABI declared in include/csn_hip.h).

This is the only place the Python host touches native code.  There is no CPU fallback:
if the library is missing or a call fails, an exception is raised.  Tensors are passed
as raw device pointers (``tensor.data_ptr()``) together with the current HIP stream of
PyTorch; PyTorch itself is used only for device memory, streams and torch.distributed.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSN_LIB_PATH") or os.path.join(_HERE, "lib", "libcsn_hip.so")

CSN_F32, CSN_BF16 = 0, 1
STATUS_TIMEOUT, STATUS_NONFINITE, STATUS_STALE_SLOT = 1, 2, 4      # bits of csn_lstm_status_read (include/csn_hip.h)
ABI_VERSION = 5

_c_void_p, _c_int, _c_i64, _c_size_t, _c_float = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                                  ctypes.c_size_t, ctypes.c_float)


class LstmDesc(ctypes.Structure):
    _fields_ = [("B", ctypes.c_int32), ("T", ctypes.c_int32), ("I", ctypes.c_int32),
                ("H", ctypes.c_int32), ("L", ctypes.c_int32), ("dtype", ctypes.c_int32)]


# name -> (restype, argtypes): every symbol include/csn_hip.h declares
SIGNATURES = {
    "csn_abi_version": (_c_int, []),
    "csn_last_error": (ctypes.c_char_p, []),
    "csn_target_arch": (ctypes.c_char_p, []),
    "csn_eeg_bandpass_znorm": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_double), _c_int,
                                        _c_int, _c_void_p, _c_int, _c_int, _c_void_p]),
    "csn_eeg_filtfilt_scratch_bytes": (_c_size_t, [_c_int, _c_int, _c_int, _c_int]),
    "csn_eeg_filtfilt": (_c_int, [_c_void_p, _c_int, _c_int, _c_int, ctypes.POINTER(ctypes.c_double), _c_int,
                                  _c_void_p, _c_void_p, _c_void_p]),
    "csn_lstm_plan_create": (_c_int, [ctypes.POINTER(LstmDesc), _c_int, ctypes.POINTER(_c_void_p)]),
    "csn_lstm_plan_destroy": (None, [_c_void_p]),
    "csn_lstm_plan_workspace_bytes": (_c_size_t, [_c_void_p]),
    "csn_lstm_plan_path": (_c_int, [_c_void_p]),
    "csn_lstm_plan_dgates_copies": (_c_int, [_c_void_p]),
    "csn_lstm_plan_kernel_name": (ctypes.c_char_p, [_c_void_p, _c_int]),
    "csn_lstm_plan_set_grad_callback": (_c_int, [_c_void_p, _c_void_p, _c_void_p]),
    "csn_lstm_workspace_bytes": (_c_size_t, [ctypes.POINTER(LstmDesc), _c_int]),
    "csn_lstm_forward": (_c_int, [_c_void_p, _c_void_p, _c_i64, _c_i64,
                                  ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                                  ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                                  _c_void_p, _c_void_p, _c_void_p, _c_void_p]),
    "csn_lstm_backward": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p,
                                   ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                                   ctypes.POINTER(_c_void_p), ctypes.POINTER(_c_void_p),
                                   _c_void_p, _c_void_p]),
    "csn_lstm_workspace_init": (_c_int, [_c_void_p, _c_void_p, _c_void_p]),
    "csn_lstm_status_clear": (_c_int, [_c_void_p, _c_void_p, _c_void_p]),
    "csn_lstm_status_read": (_c_int, [_c_void_p, _c_void_p, ctypes.POINTER(_c_int)]),
    "csn_lstm_status_raise": (_c_int, [_c_void_p, _c_void_p, _c_void_p]),
    "csn_lstm_profile_enable": (_c_int, [_c_void_p, _c_int]),
    "csn_lstm_profile_read": (_c_int, [_c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_c_int),
                                       ctypes.POINTER(_c_int), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_c_int),
                                       ctypes.POINTER(_c_int)]),
    "csn_gemm_nt": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_i64, _c_i64, _c_i64,
                             _c_int, _c_int, _c_int, _c_void_p]),
    "csn_gemm_tn_scratch_bytes": (_c_size_t, [_c_i64, _c_i64, _c_i64]),
    "csn_gemm_tn": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_i64, _c_i64, _c_i64, _c_int, _c_void_p, _c_void_p]),
    "csn_lstm_cell_forward": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_i64, _c_void_p, _c_void_p, _c_void_p,
                                       _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "csn_lstm_cell_backward": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_i64, _c_void_p, _c_void_p, _c_void_p,
                                        _c_void_p, _c_void_p, _c_int, _c_int, _c_int, _c_void_p]),
    "csn_cosine_loss_scratch_bytes": (_c_size_t, [_c_int]),
    "csn_cosine_loss": (_c_int, [_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p, _c_float, _c_void_p, _c_void_p]),
    "csn_rmsprop_step": (_c_int, [_c_void_p, _c_void_p, _c_void_p, _c_i64, _c_float, _c_float, _c_float, _c_void_p]),
    "csn_barlow_offdiag_sqsum": (_c_int, [_c_void_p, _c_int, _c_void_p, _c_void_p]),
    "csn_l2_topk_scratch_bytes": (_c_size_t, [_c_i64, _c_i64]),
    "csn_l2_topk": (_c_int, [_c_void_p, _c_void_p, _c_i64, _c_i64, _c_int, _c_int, _c_void_p, _c_void_p,
                             _c_void_p, _c_void_p]),
}


GRAD_READY_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int)      # csnGradReadyFn(user, layer)


class CsnError(RuntimeError):
    pass


_lib = None


def load():
    """Loads libcsn_hip.so (after torch, so both share one HIP runtime).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CsnError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.csn_abi_version() != ABI_VERSION:
        raise CsnError(f"libcsn_hip.so ABI {lib.csn_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise CsnError(f"libcsn_hip status {rc}: {load().csn_last_error().decode()}")


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dt(dtype):
    if dtype in (torch.float32, "f32", "float32", CSN_F32):
        return CSN_F32
    if dtype in (torch.bfloat16, "bf16", "bfloat16", CSN_BF16):
        return CSN_BF16
    raise CsnError(f"unsupported dtype {dtype}")


def torch_dtype(code):
    return torch.bfloat16 if code == CSN_BF16 else torch.float32


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise CsnError("libcsn_hip needs device tensors (no CPU fallback)")


# ------------------------------------------------------------------------------------------
def eeg_bandpass_znorm(x_bct, sos, ddof=0, out_dtype=torch.float32, time_major=False):
    """x[B,C,T] float32 (device) -> y[B,T,C] or [T,B,C]; sos = [nsec,6] (host array)."""
    import numpy as np
    _need_cuda(x_bct)
    x = x_bct.contiguous()
    if x.dtype != torch.float32:
        raise CsnError("eeg_bandpass_znorm expects float32 input")
    B, C, T = x.shape
    sos = np.ascontiguousarray(np.asarray(sos, dtype=np.float64).reshape(-1, 6)) if sos is not None else np.zeros((0, 6))
    y = torch.empty((T, B, C) if time_major else (B, T, C), dtype=out_dtype, device=x.device)
    with torch.cuda.device(x.device):
        _check(load().csn_eeg_bandpass_znorm(_ptr(x), B, C, T, sos.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                             sos.shape[0], int(ddof), _ptr(y), _dt(out_dtype), int(time_major), _stream()))
    return y


def eeg_filtfilt(x_stc, sos):
    """Zero-phase band-pass of eeg[S,T,C] float32 (device); sos = [nsec,6] host array."""
    import numpy as np
    _need_cuda(x_stc)
    x = x_stc.float().contiguous()
    S, T, C = x.shape
    sos = np.ascontiguousarray(np.asarray(sos, dtype=np.float64).reshape(-1, 6))
    lib = load()
    scratch = torch.empty(max(1, lib.csn_eeg_filtfilt_scratch_bytes(S, T, C, sos.shape[0])), dtype=torch.uint8,
                          device=x.device)
    y = torch.empty_like(x)
    with torch.cuda.device(x.device):
        _check(lib.csn_eeg_filtfilt(_ptr(x), S, T, C, sos.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), sos.shape[0],
                                    _ptr(y), _ptr(scratch), _stream()))
    return y


def gemm_nt(a, bt, bias=None, out_dtype=torch.float32, out=None, accumulate=False):
    _need_cuda(a, bt)
    a, bt = a.contiguous(), bt.contiguous()
    M, K = a.shape
    N = bt.shape[0]
    assert bt.shape[1] == K and a.dtype == bt.dtype
    c = out if out is not None else torch.empty((M, N), dtype=out_dtype, device=a.device)
    _check(load().csn_gemm_nt(_ptr(a), _ptr(bt), _ptr(bias), _ptr(c), M, N, K, _dt(a.dtype), _dt(c.dtype),
                              int(accumulate), _stream()))
    return c


def gemm_tn(a_km, b_kn):
    _need_cuda(a_km, b_kn)
    a, b = a_km.contiguous(), b_kn.contiguous()
    K, M = a.shape
    N = b.shape[1]
    assert b.shape[0] == K and a.dtype == b.dtype
    lib = load()
    scratch = torch.empty(max(1, lib.csn_gemm_tn_scratch_bytes(M, N, K)), dtype=torch.uint8, device=a.device)
    c = torch.empty((M, N), dtype=torch.float32, device=a.device)
    _check(lib.csn_gemm_tn(_ptr(a), _ptr(b), _ptr(c), M, N, K, _dt(a.dtype), _ptr(scratch), _stream()))
    return c


def lstm_cell_forward(h_prev, w_hh, xproj, c_prev, want_gates=True):
    _need_cuda(w_hh, xproj)
    B, G = xproj.shape
    H = G // 4
    dt = w_hh.dtype
    gates = torch.empty((B, G), dtype=dt, device=xproj.device) if want_gates else None
    c_out = torch.empty((B, H), dtype=torch.float32, device=xproj.device)
    h_out = torch.empty((B, H), dtype=dt, device=xproj.device)
    _check(load().csn_lstm_cell_forward(_ptr(h_prev), _ptr(w_hh), _ptr(xproj), G, _ptr(c_prev), _ptr(gates),
                                        _ptr(c_out), _ptr(h_out), B, H, _dt(dt), _stream()))
    return h_out, c_out, gates


def lstm_cell_backward(dgates_next, w_hh_t, dy, gates, c, c_prev, dc_carry):
    _need_cuda(gates, c, dc_carry)
    B, G = gates.shape
    H = G // 4
    out = torch.empty_like(gates)
    _check(load().csn_lstm_cell_backward(_ptr(dgates_next), _ptr(w_hh_t), _ptr(dy), H, _ptr(gates), _ptr(c),
                                         _ptr(c_prev), _ptr(dc_carry), _ptr(out), B, H, _dt(gates.dtype), _stream()))
    return out


def _ptr_array(tensors):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class LstmPlan:
    """One stacked-LSTM problem: a native plan handle (csn_lstm_plan_create: layout, switches, side streams, events
    -- all per plan, the library has no global state) + one workspace; forward()/backward() enqueue on the
    current stream."""

    def __init__(self, B, T, I, H, L, dtype, device, training=True):
        self.desc = LstmDesc(B, T, I, H, L, _dt(dtype))
        self.training = bool(training)
        self.device = torch.device(device)
        lib = load()
        handle = _c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.csn_lstm_plan_create(ctypes.byref(self.desc), int(self.training), ctypes.byref(handle)))
        self._plan = handle
        nbytes = lib.csn_lstm_plan_workspace_bytes(self._plan)
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        off = (-self.workspace.data_ptr()) % 256
        self._ws_ptr = ctypes.c_void_p(self.workspace.data_ptr() + off)
        self.busy = False
        with torch.cuda.device(self.device):      # torch.empty memory: status word, zero initial state, ...
            _check(lib.csn_lstm_workspace_init(self._plan, self._ws_ptr, _stream()))

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan and _lib is not None:
            _lib.csn_lstm_plan_destroy(plan)

    def key(self):
        d = self.desc
        return (d.B, d.T, d.I, d.H, d.L, d.dtype, self.training)

    def path(self):
        """0 generic cells, 1 per-diagonal bf16 launches, 2 weight-stationary forward, 3 + weight-stationary backward, 4 the
        exact-float32 path's weight-stationary recurrence."""
        return load().csn_lstm_plan_path(self._plan)

    def kernel_names(self):
        """(forward, backward) recurrence kernel of this plan's path, as a rocprofv3 kernel trace names them."""
        lib = load()
        return tuple((lib.csn_lstm_plan_kernel_name(self._plan, k) or b"").decode() for k in (0, 1))

    def set_grad_callback(self, fn):
        """fn(layer) is called on this thread from inside backward() once layer's gradient kernels are enqueued (top layer
        first); None removes it.  The ctypes thunk is kept alive by the plan."""
        if getattr(self, "_grad_cb_fn", None) is fn and (fn is None or getattr(self, "_grad_cb", None) is not None):
            return                                  # unchanged since the last backward: keep the installed thunk
        self._grad_cb_fn = fn
        self._grad_cb = GRAD_READY_FN(lambda _user, layer: fn(int(layer))) if fn is not None else None
        _check(load().csn_lstm_plan_set_grad_callback(self._plan, ctypes.cast(self._grad_cb, _c_void_p) if fn is not None
                                                      else None, None))

    def dgates_copies(self):
        """Copies of the gate gradients the last backward wrote per step (csn_hip.h): 1, 2, or 0 before any backward."""
        return load().csn_lstm_plan_dgates_copies(self._plan)

    def forward(self, x_bti, w_ih, w_hh, b_ih, b_hh, want_all=False):
        d = self.desc
        _need_cuda(x_bti)
        if x_bti.dtype != torch.float32 or x_bti.stride(2) != 1:
            x_bti = x_bti.float().contiguous()
        assert x_bti.shape == (d.B, d.T, d.I), (tuple(x_bti.shape), (d.B, d.T, d.I))
        y_last = torch.empty((d.B, d.H), dtype=torch.float32, device=x_bti.device)
        y_all = torch.empty((d.B, d.T, d.H), dtype=torch.float32, device=x_bti.device) if want_all else None
        ws = [[p.detach() for p in group] for group in (w_ih, w_hh, b_ih, b_hh)]
        for group in ws:
            for p in group:
                assert p.dtype == torch.float32 and p.is_contiguous() and p.is_cuda
        with torch.cuda.device(self.device):
            _check(load().csn_lstm_forward(self._plan, _ptr(x_bti), x_bti.stride(0), x_bti.stride(1),
                                           _ptr_array(ws[0]), _ptr_array(ws[1]), _ptr_array(ws[2]), _ptr_array(ws[3]),
                                           self._ws_ptr, _ptr(y_last), _ptr(y_all), _stream()))
        return y_last, y_all

    def backward(self, dy_last, dy_all, grads, dx=None):
        """grads: 4 lists (dw_ih, dw_hh, db_ih, db_hh) of float32 device tensors, overwritten."""
        if dy_last is not None:
            dy_last = dy_last.float().contiguous()
        if dy_all is not None:
            dy_all = dy_all.float().contiguous()
        with torch.cuda.device(self.device):
            _check(load().csn_lstm_backward(self._plan, _ptr(dy_last), _ptr(dy_all), self._ws_ptr,
                                            _ptr_array(grads[0]), _ptr_array(grads[1]), _ptr_array(grads[2]),
                                            _ptr_array(grads[3]), _ptr(dx), _stream()))

    # ---- sticky status word of the workspace ------------------------------------------------------
    def clear_status(self):
        with torch.cuda.device(self.device):
            _check(load().csn_lstm_status_clear(self._plan, self._ws_ptr, _stream()))

    def inject_timeout(self):
        """Test hook: leaves the status word as a timed-out in-kernel wait would."""
        with torch.cuda.device(self.device):
            _check(load().csn_lstm_status_raise(self._plan, self._ws_ptr, _stream()))

    def status(self, clear=False):
        """Blocking: 0 if every in-kernel hand-off since the last clear completed; bit STATUS_TIMEOUT = a bounded wait
        gave up, bit STATUS_NONFINITE = a NaN / Inf gradient reached the backward (the word is sticky: an event in ANY
        forward / backward since the last clear keeps it raised)."""
        out = _c_int(0)
        _check(load().csn_lstm_status_read(self._plan, self._ws_ptr, ctypes.byref(out)))
        if clear and out.value != 0:
            self.clear_status()
        return out.value

    # ---- per-plan event timing of the recurrence launches -----------------------------------------
    def profile_enable(self, on=True):
        _check(load().csn_lstm_profile_enable(self._plan, int(on)))

    def profile_read(self):
        """-> dict(fwd_ms, fwd_launches, fwd_cells, bwd_ms, bwd_launches, bwd_cells) of the last fwd/bwd."""
        fm, bm = ctypes.c_double(), ctypes.c_double()
        fl, fc, bl, bc = _c_int(), _c_int(), _c_int(), _c_int()
        _check(load().csn_lstm_profile_read(self._plan, ctypes.byref(fm), ctypes.byref(fl), ctypes.byref(fc),
                                            ctypes.byref(bm), ctypes.byref(bl), ctypes.byref(bc)))
        return dict(fwd_ms=fm.value, fwd_launches=fl.value, fwd_cells=fc.value,
                    bwd_ms=bm.value, bwd_launches=bl.value, bwd_cells=bc.value)


def cosine_loss(student, teacher, want_grad=True, grad_scale=1.0):
    _need_cuda(student, teacher)
    s, t = student.float().contiguous(), teacher.float().contiguous()
    B, D = s.shape
    loss = torch.empty(1, dtype=torch.float32, device=s.device)
    ds = torch.empty_like(s) if want_grad else None
    lib = load()
    scratch = torch.empty(lib.csn_cosine_loss_scratch_bytes(B) // 8, dtype=torch.float64, device=s.device)   # caller-owned
    with torch.cuda.device(s.device):
        _check(lib.csn_cosine_loss(_ptr(s), _ptr(t), B, D, _ptr(loss), _ptr(ds), float(grad_scale), _ptr(scratch), _stream()))
    return loss, ds


def rmsprop_step(params_flat, grads_flat, square_avg_flat, lr, alpha=0.99, eps=1e-8):
    """In place, on the current stream: one fused pass over the flat float32 buffers."""
    _need_cuda(params_flat, grads_flat, square_avg_flat)
    n = params_flat.numel()
    assert grads_flat.numel() == n and square_avg_flat.numel() == n
    assert params_flat.dtype == grads_flat.dtype == square_avg_flat.dtype == torch.float32
    with torch.cuda.device(params_flat.device):
        _check(load().csn_rmsprop_step(_ptr(params_flat), _ptr(grads_flat), _ptr(square_avg_flat), n, float(lr), float(alpha),
                                       float(eps), _stream()))


def barlow_offdiag_sqsum(c):
    _need_cuda(c)
    c = c.float().contiguous()
    out = torch.empty(2, dtype=torch.float32, device=c.device)
    _check(load().csn_barlow_offdiag_sqsum(_ptr(c), c.shape[0], _ptr(out), _stream()))
    return out


def l2_topk(gallery, query, k):
    _need_cuda(gallery, query)
    g, q = gallery.float().contiguous(), query.float().contiguous()
    Ng, D = g.shape
    Nq = q.shape[0]
    lib = load()
    scratch = torch.empty(lib.csn_l2_topk_scratch_bytes(Ng, Nq), dtype=torch.uint8, device=g.device)
    idx = torch.empty((Nq, k), dtype=torch.int64, device=g.device)
    dist = torch.empty((Nq, k), dtype=torch.float32, device=g.device)
    _check(lib.csn_l2_topk(_ptr(g), _ptr(q), Ng, Nq, D, k, _ptr(idx), _ptr(dist), _ptr(scratch), _stream()))
    return dist, idx
