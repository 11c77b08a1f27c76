"""Oracle: stacked LSTM written out as an explicit recurrence (test infrastructure only).

Restates the arithmetic of ``torch.nn.LSTM(input, hidden, num_layers, batch_first=True)``
as the reference uses it (/root/reference/LSTMDistill.py:112-142,
/root/reference/LSTMDistillRetreival.py:85-110): zero initial state, gate order
i,f,g,o, parameters ``weight_ih_l{k}[4H,I] weight_hh_l{k}[4H,H] bias_ih_l{k}[4H]
bias_hh_l{k}[4H]``, followed by ``fc = Linear(H, D)`` on the last timestep (or
all timesteps) and the optional ``class_pred = Linear(D, n_classes)``.

Forward and the hand-derived backward are plain numpy (float64 by default) so
that the HIP kernels are checked against something that shares no code with
them.  Pinned against ``torch.nn.LSTM`` CPU outputs/grads in
tests/golden/lstm_*.npz.
"""
import numpy as np


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_forward(x_btc, params, num_layers, dtype=np.float64, return_saved=False):
    """x[B,T,I]; params: dict with torch.nn.LSTM key names (numpy arrays).

    Returns y[B,T,H] of the top layer (and the per-layer saved tensors).
    """
    x = np.asarray(x_btc, dtype=dtype)
    B, T, _ = x.shape
    saved = []
    inp = x
    for l in range(num_layers):
        w_ih = np.asarray(params[f"weight_ih_l{l}"], dtype)
        w_hh = np.asarray(params[f"weight_hh_l{l}"], dtype)
        bias = np.asarray(params[f"bias_ih_l{l}"], dtype) + np.asarray(params[f"bias_hh_l{l}"], dtype)
        H = w_hh.shape[1]
        h = np.zeros((B, H), dtype)
        c = np.zeros((B, H), dtype)
        hs = np.empty((B, T, H), dtype)
        gates = np.empty((B, T, 4 * H), dtype)
        cs = np.empty((B, T, H), dtype)
        for t in range(T):
            a = inp[:, t, :] @ w_ih.T + h @ w_hh.T + bias
            i = _sigmoid(a[:, 0 * H:1 * H])
            f = _sigmoid(a[:, 1 * H:2 * H])
            g = np.tanh(a[:, 2 * H:3 * H])
            o = _sigmoid(a[:, 3 * H:4 * H])
            c = f * c + i * g
            h = o * np.tanh(c)
            hs[:, t] = h
            cs[:, t] = c
            gates[:, t] = np.concatenate([i, f, g, o], axis=1)
        saved.append(dict(inp=inp, hs=hs, cs=cs, gates=gates))
        inp = hs
    if return_saved:
        return inp, saved
    return inp


def lstm_backward(dy_bth, params, saved, num_layers, dtype=np.float64):
    """Backward of :func:`lstm_forward`.  dy[B,T,H] = dLoss/dy(top layer, every step).

    Returns (dx[B,T,I], grads dict with torch key names).
    """
    grads = {}
    dout = np.asarray(dy_bth, dtype)
    for l in reversed(range(num_layers)):
        w_ih = np.asarray(params[f"weight_ih_l{l}"], dtype)
        w_hh = np.asarray(params[f"weight_hh_l{l}"], dtype)
        s = saved[l]
        inp, hs, cs, gates = s["inp"], s["hs"], s["cs"], s["gates"]
        B, T, H = hs.shape
        dh_rec = np.zeros((B, H), dtype)
        dc_next = np.zeros((B, H), dtype)
        da_all = np.empty((B, T, 4 * H), dtype)
        for t in reversed(range(T)):
            i = gates[:, t, 0 * H:1 * H]
            f = gates[:, t, 1 * H:2 * H]
            g = gates[:, t, 2 * H:3 * H]
            o = gates[:, t, 3 * H:4 * H]
            c = cs[:, t]
            c_prev = cs[:, t - 1] if t > 0 else np.zeros_like(c)
            tc = np.tanh(c)
            dh = dout[:, t] + dh_rec
            do = dh * tc
            dc = dh * o * (1.0 - tc * tc) + dc_next
            di = dc * g
            df = dc * c_prev
            dg = dc * i
            da = np.concatenate([di * i * (1 - i), df * f * (1 - f), dg * (1 - g * g), do * o * (1 - o)], axis=1)
            da_all[:, t] = da
            dh_rec = da @ w_hh
            dc_next = dc * f
        da2 = da_all.reshape(B * T, 4 * H)
        h_prev = np.concatenate([np.zeros((B, 1, H), dtype), hs[:, :-1]], axis=1).reshape(B * T, H)
        grads[f"weight_ih_l{l}"] = da2.T @ inp.reshape(B * T, -1)
        grads[f"weight_hh_l{l}"] = da2.T @ h_prev
        grads[f"bias_ih_l{l}"] = da2.sum(axis=0)
        grads[f"bias_hh_l{l}"] = da2.sum(axis=0)
        dout = (da2 @ w_ih).reshape(B, T, -1)
    return dout, grads


def model_forward(x_btc, params, num_layers, include_top=False, dtype=np.float64, return_saved=False):
    """``models.lstm.Model`` contract (precedent
    LSTMDistillRetreival.py:103-108): fc(lstm(x)[:, -1, :]) -> [B,D]; with
    ``include_top`` also class_pred(fc_out) -> [B,n_classes] (LSTMDistill.py:139-140).
    Param keys: ``lstm.*``, ``fc.weight/bias``, ``class_pred.weight/bias``.
    """
    lstm_p = {k[len("lstm."):]: v for k, v in params.items() if k.startswith("lstm.")}
    y, saved = lstm_forward(x_btc, lstm_p, num_layers, dtype, return_saved=True)
    last = y[:, -1, :]
    feat = last @ np.asarray(params["fc.weight"], dtype).T + np.asarray(params["fc.bias"], dtype)
    out = (feat,)
    if include_top:
        cls = feat @ np.asarray(params["class_pred.weight"], dtype).T + np.asarray(params["class_pred.bias"], dtype)
        out = (feat, cls)
    result = out if include_top else feat
    if return_saved:
        return result, dict(lstm=saved, y=y, last=last, feat=feat)
    return result


def model_backward(dfeat, params, saved, num_layers, dcls=None, dtype=np.float64):
    """Backward through head + LSTM.  Returns grads dict with ``Model`` key names."""
    grads = {}
    dfeat = np.asarray(dfeat, dtype).copy()
    feat, last, y = saved["feat"], saved["last"], saved["y"]
    if dcls is not None:
        dcls = np.asarray(dcls, dtype)
        grads["class_pred.weight"] = dcls.T @ feat
        grads["class_pred.bias"] = dcls.sum(axis=0)
        dfeat = dfeat + dcls @ np.asarray(params["class_pred.weight"], dtype)
    grads["fc.weight"] = dfeat.T @ last
    grads["fc.bias"] = dfeat.sum(axis=0)
    dlast = dfeat @ np.asarray(params["fc.weight"], dtype)
    dy = np.zeros_like(y)
    dy[:, -1, :] = dlast
    lstm_p = {k[len("lstm."):]: v for k, v in params.items() if k.startswith("lstm.")}
    dx, g = lstm_backward(dy, lstm_p, saved["lstm"], num_layers, dtype)
    for k, v in g.items():
        grads["lstm." + k] = v
    return dx, grads


def reference_lstm_model(x_b_ts_ch, params, num_layers, target=None, dtype=np.float64):
    """``LSTMModel`` of LSTMDistillRetreival.py:85-110: the input [B, timespan, channels] is *viewed* (memory
    reinterpreted, not transposed) as [B, channels, timespan] (:97-98), i.e. the recurrence runs over ``channels``
    steps with ``timespan`` features; zero initial state; fc on the last step.  With ``target``: also the
    CosineSimilarityLoss (LstmDistillFromDinoV2Train.py:36-43) and every parameter gradient.
    """
    from . import losses
    x = np.ascontiguousarray(np.asarray(x_b_ts_ch))
    B, TS, CH = x.shape
    seq = x.reshape(B, CH, TS)
    feat, saved = model_forward(seq, params, num_layers, dtype=dtype, return_saved=True)
    if target is None:
        return feat
    loss = losses.cosine_similarity_loss(feat, target)
    _, grads = model_backward(losses.cosine_similarity_loss_grad(feat, target), params, saved, num_layers, dtype=dtype)
    return feat, loss, grads


def reference_lstm_model_all_steps(x_b_ts_ch, params, num_layers, dfeat=None, dcls=None, dtype=np.float64):
    """``LSTMModel`` of LSTMDistill.py:112-142: same view, then fc on EVERY step, class_pred on the un-rectified
    fc output, ReLU on the returned features (:137-141).  With (dfeat, dcls) = dLoss/d(returned feat, cls):
    the parameter gradients.
    """
    x = np.ascontiguousarray(np.asarray(x_b_ts_ch))
    B, TS, CH = x.shape
    seq = x.reshape(B, CH, TS)
    lstm_p = {k[len("lstm."):]: v for k, v in params.items() if k.startswith("lstm.")}
    y, saved = lstm_forward(seq, lstm_p, num_layers, dtype, return_saved=True)            # [B, CH, H]
    wf, bf = np.asarray(params["fc.weight"], dtype), np.asarray(params["fc.bias"], dtype)
    wc, bc = np.asarray(params["class_pred.weight"], dtype), np.asarray(params["class_pred.bias"], dtype)
    pre = y @ wf.T + bf
    cls = pre @ wc.T + bc
    feat = np.maximum(pre, 0.0)
    if dfeat is None:
        return feat, cls
    dpre = np.asarray(dfeat, dtype) * (pre > 0) + np.asarray(dcls, dtype) @ wc
    grads = {"class_pred.weight": np.einsum("btn,btd->nd", np.asarray(dcls, dtype), pre),
             "class_pred.bias": np.asarray(dcls, dtype).sum(axis=(0, 1)),
             "fc.weight": np.einsum("btd,bth->dh", dpre, y), "fc.bias": dpre.sum(axis=(0, 1))}
    _, g = lstm_backward(dpre @ wf, lstm_p, saved, num_layers, dtype)
    for k, v in g.items():
        grads["lstm." + k] = v
    return feat, cls, grads


def init_params(input_size, hidden, num_layers, out_features, n_classes=None, seed=43, dtype=np.float32):
    """Deterministic numpy init with nn.LSTM / nn.Linear's U(-1/sqrt(fan), 1/sqrt(fan)) ranges."""
    rng = np.random.default_rng(seed)
    p = {}
    k = 1.0 / np.sqrt(hidden)
    for l in range(num_layers):
        i_sz = input_size if l == 0 else hidden
        p[f"lstm.weight_ih_l{l}"] = rng.uniform(-k, k, (4 * hidden, i_sz)).astype(dtype)
        p[f"lstm.weight_hh_l{l}"] = rng.uniform(-k, k, (4 * hidden, hidden)).astype(dtype)
        p[f"lstm.bias_ih_l{l}"] = rng.uniform(-k, k, (4 * hidden,)).astype(dtype)
        p[f"lstm.bias_hh_l{l}"] = rng.uniform(-k, k, (4 * hidden,)).astype(dtype)
    p["fc.weight"] = rng.uniform(-k, k, (out_features, hidden)).astype(dtype)
    p["fc.bias"] = rng.uniform(-k, k, (out_features,)).astype(dtype)
    if n_classes:
        kc = 1.0 / np.sqrt(out_features)
        p["class_pred.weight"] = rng.uniform(-kc, kc, (n_classes, out_features)).astype(dtype)
        p["class_pred.bias"] = rng.uniform(-kc, kc, (n_classes,)).astype(dtype)
    return p
