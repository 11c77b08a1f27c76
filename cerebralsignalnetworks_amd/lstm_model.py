"""Host mirror of the reference's LSTM encoders, running on the HIP kernels.

``Model`` fills the place of the reference's missing ``models/lstm.py`` (imported at
/root/reference/LstmDistillFromDinoV2Train.py:5, constructed at :323,
LstmDistillFromDinoV2TrainSpampinato.py:368, LstmDistillFromDinoV2Eval.py:308) with the
call-site contract of SURVEY.md section 8(b): ``Model(input_size, lstm_size, lstm_layers, output_size,
include_top)``; ``forward(x[B,T,C]) -> [B,output_size]`` or ``([B,output_size],
[B,n_classes])``.  Parameter names and layouts follow the in-tree precedent
(/root/reference/LSTMDistill.py:118-120): ``lstm.{weight_ih,weight_hh,bias_ih,bias_hh}_l{k}``,
``fc.*``, ``class_pred.*`` -- checkpoints round-trip with a stock ``nn.LSTM``.
"""
import torch
import torch.nn as nn

from . import cabi


class _Lease:
    """Holds a plan (= its workspace) for one autograd node.  Released by the node's backward -- or, when the
    graph is dropped without a backward (a skipped step, a validation pass without no_grad, an exception between
    forward and backward), when the node itself is collected, so such a forward never pins a workspace for good."""

    def __init__(self, plan):
        self.plan = plan
        plan.busy = True

    def release(self):
        if self.plan is not None:
            self.plan.busy = False
            self.plan = None

    __del__ = release


class _LstmFunction(torch.autograd.Function):
    """Stacked LSTM over libcsn_hip.  A training forward keeps its state in a workspace that stays
    checked out until the matching backward has run, so several forwards (e.g. the multi-crop views of
    the DINO trainer) can be outstanding at once."""

    @staticmethod
    def forward(ctx, x, owner, want_all, training, L, *params):
        w_ih, w_hh, b_ih, b_hh = params[0:L], params[L:2 * L], params[2 * L:3 * L], params[3 * L:4 * L]
        plan = owner._checkout(x.shape[0], x.shape[1], x.device, training)
        y_last, y_all = plan.forward(x, w_ih, w_hh, b_ih, b_hh, want_all=want_all)
        ctx.lease, ctx.L, ctx.want_all = _Lease(plan), L, want_all
        ctx.owner = owner
        ctx.need_dx = x.requires_grad
        ctx.x_shape = x.shape
        ctx.param_like = params
        if not training:
            ctx.lease.release()
        if want_all:
            return y_last, y_all
        return y_last, y_last.new_empty(0)

    @staticmethod
    def backward(ctx, dy_last, dy_all):
        L, plan = ctx.L, ctx.lease.plan
        if plan is None:
            raise RuntimeError("HipLSTM: second backward through one forward -- its workspace was handed back after the "
                               "first (retain_graph / double backward are not supported)")
        owner = ctx.owner
        direct = owner.direct_grads and all(p.grad is not None and p.grad.is_contiguous() and p.grad.dtype == torch.float32
                                            for p in ctx.param_like)
        if direct:
            # the library OVERWRITES its gradient outputs: written straight into the parameters' .grad (the views into the
            # trainer's flat buffer) this forward's contribution needs no temporaries and no accumulation pass -- valid
            # while this is the only forward of the step that uses these parameters (the trainer's contract)
            grads = [[p.grad for p in ctx.param_like[g * L:(g + 1) * L]] for g in range(4)]
        else:
            grads = [[torch.empty_like(p) for p in ctx.param_like[g * L:(g + 1) * L]] for g in range(4)]
        dx = torch.empty(ctx.x_shape, dtype=torch.float32, device=dy_last.device) if ctx.need_dx else None
        plan.set_grad_callback(owner.grad_ready_hook if direct else None)
        plan.backward(dy_last, dy_all if ctx.want_all else None, grads, dx=dx)
        ctx.lease.release()
        if direct:
            return (dx, None, None, None, None, *([None] * (4 * L)))
        flat = [g for group in grads for g in group]
        return (dx, None, None, None, None, *flat)


class HipLSTM(nn.Module):
    """nn.LSTM(batch_first=True) parameter-compatible stacked LSTM on the HIP path.

    ``compute_dtype``: torch.bfloat16 (bf16 MFMA operands, f32 accumulate and cell state -- the
    fast path) or torch.float32 (exact-f32 MFMA -- the parity path).
    """

    MAX_IDLE_PLANS = 4      # workspaces are large (10 GB at cfg2): keep only a few idle ones

    def __init__(self, input_size, hidden_size, num_layers=1, compute_dtype=torch.bfloat16):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.compute_dtype = compute_dtype
        ref = nn.LSTM(input_size, hidden_size, num_layers=num_layers, batch_first=True)   # same init + key names
        for name, p in ref.named_parameters():
            self.register_parameter(name, nn.Parameter(p.detach().clone()))
        self._plans = {}        # key -> list of plans; plan.busy marks a forward awaiting its backward
        # set by a trainer that owns the gradient buffers (trainer.DistillTrainer): the backward writes each parameter's
        # gradient straight into its .grad and calls grad_ready_hook(layer) as soon as a layer's gradients are enqueued
        self.direct_grads = False
        self.grad_ready_hook = None

    def _checkout(self, B, T, device, training):
        key = (B, T, str(device), bool(training), self.compute_dtype)
        pool = self._plans.setdefault(key, [])
        for plan in pool:
            if not plan.busy:
                return plan
        idle = [(k, pl) for k, lst in self._plans.items() for pl in lst if not pl.busy and k != key]
        while len(idle) >= self.MAX_IDLE_PLANS:
            k, pl = idle.pop(0)
            self._plans[k].remove(pl)
        plan = cabi.LstmPlan(B, T, self.input_size, self.hidden_size, self.num_layers, self.compute_dtype, device,
                             training=training)
        pool.append(plan)
        return plan

    def all_plans(self):
        return [pl for lst in self._plans.values() for pl in lst]

    def forward(self, x, want_all=False):
        if not x.is_cuda:
            raise cabi.CsnError("HipLSTM runs on the GPU only (no CPU fallback); move the module and input to cuda")
        L = self.num_layers
        params = [getattr(self, f"{n}_l{k}") for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh") for k in range(L)]
        # (grad mode is off inside Function.forward, so "is a backward coming" is decided here)
        training = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        y_last, y_all = _LstmFunction.apply(x, self, want_all, training, L, *params)
        return (y_all, y_last) if want_all else y_last


class Model(nn.Module):
    """``models.lstm.Model``.  ``head``/``fc`` may be reassigned by
    ``MultiCropWrapper`` (utils/utils.py:607-612) without breaking forward."""

    def __init__(self, input_size=128, lstm_size=128, lstm_layers=1, output_size=128, include_top=True,
                 n_classes=40, compute_dtype=torch.bfloat16):
        super().__init__()
        self.input_size, self.lstm_size, self.lstm_layers = input_size, lstm_size, lstm_layers
        self.output_size, self.include_top = output_size, include_top
        self.lstm = HipLSTM(input_size, lstm_size, lstm_layers, compute_dtype=compute_dtype)
        self.fc = nn.Linear(lstm_size, output_size)
        if include_top:
            self.class_pred = nn.Linear(output_size, n_classes)

    def forward(self, x):
        last = self.lstm(x)                    # [B, H] = top layer at the last timestep
        feat = self.fc(last)
        if self.include_top and hasattr(self, "class_pred"):
            return feat, self.class_pred(feat)
        return feat


class LSTMModel(nn.Module):
    """/root/reference/LSTMDistillRetreival.py:85-110 / LSTMDistill.py:112-142 on the HIP LSTM.

    Keeps the reference's ``x.view(B, C, T)`` (a reshape, not a transpose: the sequence then runs
    over the *channel* axis with ``input_size`` = time samples).  ``all_steps=True`` gives the
    LSTMDistill.py variant: fc on every step, class_pred, ReLU on the features.
    """

    def __init__(self, input_size, hidden_size, n_layers=2, out_features=384, number_of_classes=None,
                 all_steps=False, compute_dtype=torch.bfloat16):
        super().__init__()
        self.hidden_size, self.n_layer, self.input_size, self.all_steps = hidden_size, n_layers, input_size, all_steps
        self.lstm = HipLSTM(input_size, hidden_size, n_layers, compute_dtype=compute_dtype)
        self.fc = nn.Linear(hidden_size, out_features)
        if number_of_classes:
            self.class_pred = nn.Linear(out_features, number_of_classes)

    def forward(self, x):
        batch_size, timespan, channels = x.size()
        x = x.reshape(batch_size, channels, timespan)
        if self.all_steps:
            y_all, _ = self.lstm(x, want_all=True)
            feat = self.fc(y_all)
            cls_pred = self.class_pred(feat)
            return nn.functional.relu(feat), cls_pred
        return self.fc(self.lstm(x))


class CustomModel(nn.Module):
    """/root/reference/utils/CustomModel.py:4-17 (3-layer MLP; keys fc.0/2/4.*)."""

    def __init__(self, input_size, output_size):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(input_size, 2000), nn.ReLU(), nn.Linear(2000, 2000), nn.ReLU(),
                                nn.Linear(2000, output_size))

    def forward(self, x):
        return self.fc(x)
