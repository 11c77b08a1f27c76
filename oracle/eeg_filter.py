"""Oracle: per-segment EEG band-pass + per-channel z-score (test infrastructure only).

Restates, in float64 numpy:
  * the filter *design* of /root/reference/utils/EEGFilters.py:10-26
    (Butterworth order 3/4/5, band 0.1-60 Hz normalised by fs/2) -- the design
    call itself is ``scipy.signal.butter`` exactly as the reference makes it,
    but in second-order-section form (the (b,a) form of
    this band is unstable for order >= 4);
  * a causal cascade of direct-form-II-transposed biquads along time (the
    arithmetic of ``scipy.signal.sosfilt``; the reference imports ``lfilter``
    at EEGFilters.py:2 and never calls it);
  * the zero-phase variant the reference does apply,
    /root/reference/utils/Utilities.py:411-428 (butter-4, 1-50 Hz, ``filtfilt``);
  * ``normlizeEEG`` of /root/reference/utils/PerilsEEGDataset.py:454-461:
    ``(x - mean_t) / std_t`` per channel (numpy path ddof=0, torch path ddof=1).

Pinned against scipy 1.15.3 outputs in tests/golden/filter_*.npz.
"""
import numpy as np

LOW_CUTOFF_HZ = 0.1   # EEGFilters.py:10
HIGH_CUTOFF_HZ = 60.0  # EEGFilters.py:11
ORDERS = (3, 4, 5)     # EEGFilters.py:19


def design_bandpass_sos(fs, order=3, low=LOW_CUTOFF_HZ, high=HIGH_CUTOFF_HZ):
    """SOS form of ``butter(order, [low/(fs/2), high/(fs/2)], btype='bandpass')``."""
    from scipy.signal import butter
    nyq = fs / 2.0
    return np.asarray(butter(order, [low / nyq, high / nyq], btype="bandpass", output="sos"),
                      dtype=np.float64)


def sosfilt_rows(sos, x):
    """Causal biquad cascade along the last axis, float64, zero initial state.

    x: [..., T].  Vectorised over rows, sequential over time and sections:
        y = b0*x + s1 ; s1 = b1*x - a1*y + s2 ; s2 = b2*x - a2*y
    """
    sos = np.asarray(sos, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    lead = x.shape[:-1]
    T = x.shape[-1]
    cur = x.reshape(-1, T).copy()
    for sec in sos:
        b0, b1, b2, a0, a1, a2 = sec
        b0, b1, b2, a1, a2 = b0 / a0, b1 / a0, b2 / a0, a1 / a0, a2 / a0
        s1 = np.zeros(cur.shape[0])
        s2 = np.zeros(cur.shape[0])
        out = np.empty_like(cur)
        for n in range(T):
            xn = cur[:, n]
            yn = b0 * xn + s1
            s1 = b1 * xn - a1 * yn + s2
            s2 = b2 * xn - a2 * yn
            out[:, n] = yn
        cur = out
    return cur.reshape(*lead, T)


def zscore_rows(y, ddof=0):
    """normlizeEEG (PerilsEEGDataset.py:454-461): (x-mean)/std over time, no epsilon."""
    y = np.asarray(y, dtype=np.float64)
    mean = y.mean(axis=-1, keepdims=True)
    std = y.std(axis=-1, ddof=ddof, keepdims=True)
    return (y - mean) / std


def dataset_item(raw_ct, time_low, time_high, filter_channels=(), channel_wise_norm=False, mean=None, std=None,
                 means_c=None, stds_c=None):
    """The EEG part of ``EEGDataset.__getitem__`` (utils/PerilsEEGDataset.py:541-573; utils/EEGDataset.py:539-567):
    stored [C, T_raw] -> ``.float().t()`` -> window [time_low:time_high].  With ``filter_channels``: those columns
    only, each optionally z-scored by ``normlizeEEG`` on a numpy array (ddof 0), and the result transposed BACK to
    channel-first [len(filter_channels), T] (:565 ``.t()``).  ``mean``/``std``: Perils scalar dataset-level
    normalisation (:572-573).  ``means_c``/``stds_c`` [C,1]: the Spampinato per-channel form, applied to the
    channel-first record before the transpose (utils/EEGDataset.py:543-544)."""
    rec = np.asarray(raw_ct, np.float32)
    if means_c is not None:
        rec = (rec - np.asarray(means_c, np.float32)) / np.asarray(stds_c, np.float32)
    eeg = rec.T
    if len(filter_channels) > 0:
        out = np.zeros((time_high - time_low, len(filter_channels)), np.float32)
        for j, ch in enumerate(filter_channels):
            out[:, j] = eeg[time_low:time_high, ch]
            if channel_wise_norm:
                col = out[:, j]
                out[:, j] = (col - col.mean()) / col.std()
        eeg = out.T
    else:
        eeg = eeg[time_low:time_high, :]
    if mean is not None:
        eeg = (eeg - np.float32(mean)) / np.float32(std)
    return eeg


def eeg_bandpass_znorm(x_bct, sos, ddof=0, time_major=False):
    """x[B,C,T] -> filtered + z-scored, laid out for the LSTM.

    Returns float64 [B,T,C] (``time_major=False``, the reference's ``.t()``
    layout, PerilsEEGDataset.py:549) or [T,B,C].
    """
    y = zscore_rows(sosfilt_rows(sos, x_bct), ddof=ddof)   # [B,C,T]
    if time_major:
        return np.ascontiguousarray(np.transpose(y, (2, 0, 1)))
    return np.ascontiguousarray(np.transpose(y, (0, 2, 1)))


# ---------------------------------------------------------------------------
# zero-phase variant (Utilities.remove_noise, Utilities.py:411-428)
# ---------------------------------------------------------------------------
def _lfilter_ba(b, a, x, zi):
    """Direct-form-II-transposed (b,a) filter along last axis with initial state."""
    n = len(a)
    b = np.asarray(b, np.float64) / a[0]
    a = np.asarray(a, np.float64) / a[0]
    z = np.array(zi, dtype=np.float64, copy=True)          # [rows, n-1]
    y = np.empty_like(x)
    for t in range(x.shape[-1]):
        xt = x[:, t]
        yt = z[:, 0] + b[0] * xt
        for k in range(n - 2):
            z[:, k] = z[:, k + 1] + xt * b[k + 1] - yt * a[k + 1]
        z[:, n - 2] = xt * b[n - 1] - yt * a[n - 1]
        y[:, t] = yt
    return y


def _lfilter_zi(b, a):
    """Steady-state DF2T state for a unit step (scipy.signal.lfilter_zi's linear system)."""
    b = np.asarray(b, np.float64) / a[0]
    a = np.asarray(a, np.float64) / a[0]
    n = len(a)
    # zi = A zi + B with A = companion(a).T; solved by the same closed-form recursion scipy
    # uses (the band is ill-conditioned enough that a generic solve differs at 1e-5).
    B = b[1:] - a[1:] * b[0]
    zi = np.zeros(n - 1)
    zi[0] = B.sum() / (1.0 + a[1:].sum())
    asum, csum = 1.0, 0.0
    for k in range(1, n - 1):
        asum += a[k]
        csum += b[k] - a[k] * b[0]
        zi[k] = asum * zi[0] - csum
    return zi


def filtfilt_rows(b, a, x):
    """scipy.signal.filtfilt(b, a, x) defaults: odd extension, padlen=3*max(len(a),len(b))."""
    x = np.asarray(x, np.float64)
    lead = x.shape[:-1]
    T = x.shape[-1]
    x2 = x.reshape(-1, T)
    padlen = 3 * max(len(a), len(b))
    left = 2 * x2[:, :1] - x2[:, padlen:0:-1]
    right = 2 * x2[:, -1:] - x2[:, -2:-(padlen + 2):-1]
    ext = np.concatenate([left, x2, right], axis=1)
    zi = _lfilter_zi(b, a)
    y = _lfilter_ba(b, a, ext, zi[None, :] * ext[:, :1])
    y = _lfilter_ba(b, a, y[:, ::-1], zi[None, :] * y[:, -1:])
    y = y[:, ::-1][:, padlen:-padlen]
    return np.ascontiguousarray(y).reshape(*lead, T)


def remove_noise(eeg_stc, sampling_rate):
    """Utilities.remove_noise (Utilities.py:411-428): eeg[S,T,C] -> zero-phase butter-4 1-50 Hz."""
    from scipy.signal import butter
    nyq = 0.5 * sampling_rate
    b, a = butter(4, [1.0 / nyq, 50.0 / nyq], btype="band")
    x = np.transpose(np.asarray(eeg_stc, np.float64), (0, 2, 1))     # [S,C,T]
    return np.transpose(filtfilt_rows(b, a, x), (0, 2, 1))


def synthetic_eeg(n, channels=128, samples=500, fs=1000.0, freq=40.0, amp=0.5, seed=43):
    """Synthetic segments, recipe of /root/reference/utils/GenerateRandomEEGNoise.py:4-19
    (N(0,1) + 0.5*sin(2*pi*40*t)), float32 [n, C, T]; numpy generator (seeded)."""
    rng = np.random.default_rng(seed)
    t = np.arange(samples) / fs
    x = rng.standard_normal((n, channels, samples)) + amp * np.sin(2 * np.pi * freq * t)
    return x.astype(np.float32)


def classwise_channel_norm(eeg_nct, class_ids, time_low=0, compat_stale_index=False, stored_float32=False):
    """Class-wise, channel-wise normalisation of a dataset, restating
    /root/reference/utils/PerilsEEGDataset.py:464-507 (``transformEEGDataToChannelWiseNorm``).

    Statistics, as the reference computes them: per class (in order of first appearance), per channel, the mean
    over that class's segments of the per-segment mean and of the per-segment numpy std (ddof 0) over the time
    window ``__getitem__`` returns.  ``eeg_nct`` here IS that window (``[N, C, T]``, already sliced at
    ``time_low``), so normalising it element-wise equals slicing the reference's normalised full-length record.

    compat_stale_index=False: what the function is written to do -- every segment of the class normalised with
    its class's per-channel statistics.
    compat_stale_index=True: what the reference's code leaves behind (pinned by executing it,
    tests/golden/ref_preproc.npz).  Three things combine: (a) the record is stored channel-first ``[C, T_raw]`` but
    ``:503-506`` index it as ``eeg[:, ch]``, so "channel" ch's statistics are applied to the raw TIME sample ch of
    every channel; (b) ``:507`` stores into ``self.subsetData[i]`` with ``i`` left over from the first loop (= N-1),
    so entry N-1 is re-pointed at the transformed array of every record visited; (c) ``.float().cpu().numpy()``
    (:499-500) copies a float64 record but ALIASES a float32 one -- with float32 storage (``stored_float32``) every
    visited record is therefore also changed in place, and with float64 storage only the arrays the function
    itself created (float32) are: entry N-1, when visited after it has been re-pointed, transforms that array again.
    """
    x = np.array(eeg_nct, dtype=np.float32, copy=True)
    N, C, T = x.shape
    order = []
    for k in class_ids:
        if int(k) not in order:
            order.append(int(k))
    stats = {}
    for k in order:
        idx = [i for i in range(N) if int(class_ids[i]) == k]
        means = np.array([[x[i, c].mean() for c in range(C)] for i in idx], dtype=np.float32)
        stds = np.array([[x[i, c].std() for c in range(C)] for i in idx], dtype=np.float32)
        stats[k] = (idx, means.mean(axis=0), stds.mean(axis=0))
    if not compat_stale_index:
        out = x.copy()
        for k in order:
            idx, mm, ms = stats[k]
            for i in idx:
                out[i] = (x[i] - mm[:, None]) / ms[:, None]
        return out

    def transform(rec, k):                       # in place, raw columns ch = 0..C-1 of the channel-first record
        _, mm, ms = stats[k]
        for ch in range(C):
            col = ch - time_low
            if 0 <= col < T:
                rec[:, col] = (rec[:, col] - mm[ch]) / ms[ch]

    recs = [x[i].copy() for i in range(N)]       # physical arrays; entries point at them
    aliased = [bool(stored_float32)] * N         # does .float().cpu().numpy() alias this array?
    slot = list(range(N))
    for k in order:
        for i in stats[k][0]:
            phys = slot[i]
            if aliased[phys]:
                transform(recs[phys], k)
            else:
                recs.append(recs[phys].copy())
                aliased.append(True)             # created by torch.from_numpy(float32 array)
                phys = len(recs) - 1
                transform(recs[phys], k)
            slot[N - 1] = phys
    return np.stack([recs[slot[i]] for i in range(N)])
