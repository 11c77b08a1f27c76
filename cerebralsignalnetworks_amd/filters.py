"""Host side of K1+K2: filter design + the fused HIP band-pass / z-score.

Mirrors /root/reference/utils/EEGFilters.py:4-28 (``EEGFilters(fs)`` with attributes
``fs, low_cutoff, high_cutoff, low_cutoff_norm, high_cutoff_norm``).  The reference only
*designs* Butterworth band-passes (orders 3/4/5, 0.1-60 Hz) and discards the coefficients;
here the design is kept in second-order sections (the polynomial form of this band is
unstable for order >= 4) and ``apply`` runs the fused HIP kernel.
"""
import numpy as np
import torch
from scipy.signal import butter

from . import cabi


class EEGFilters:
    orders = (3, 4, 5)   # EEGFilters.py:19

    def __init__(self, fs, order=3) -> None:
        self.low_cutoff = 0.1      # EEGFilters.py:10
        self.high_cutoff = 60.0    # EEGFilters.py:11
        self.fs = fs
        self.low_cutoff_norm = self.low_cutoff / (self.fs / 2)
        self.high_cutoff_norm = self.high_cutoff / (self.fs / 2)
        self.order = order
        self.Butterworth_sos = {
            o: np.asarray(butter(o, [self.low_cutoff_norm, self.high_cutoff_norm], btype="bandpass", output="sos"),
                          dtype=np.float64)
            for o in self.orders
        }

    @property
    def sos(self):
        return self.Butterworth_sos[self.order]

    def apply(self, eeg_bct, ddof=0, out_dtype=torch.float32, time_major=False):
        """eeg[B,C,T] float32 on the GPU -> band-passed, per-channel z-scored [B,T,C] (or [T,B,C])."""
        return eeg_bandpass_znorm(eeg_bct, self.sos, ddof=ddof, out_dtype=out_dtype, time_major=time_major)


def eeg_bandpass_znorm(eeg_bct, sos, ddof=0, out_dtype=torch.float32, time_major=False):
    return cabi.eeg_bandpass_znorm(eeg_bct, sos, ddof=ddof, out_dtype=out_dtype, time_major=time_major)


def remove_noise(eeg_data, sampling_rate):
    """``Utilities.remove_noise`` (/root/reference/utils/Utilities.py:411-428) on the GPU: Butterworth order 4,
    1-50 Hz, zero-phase (forward-backward) filtering of eeg[S,T,C]; returns a tensor of the same shape."""
    nyquist_freq = 0.5 * sampling_rate
    sos = butter(4, [1.0 / nyquist_freq, 50.0 / nyquist_freq], btype='band', output='sos')
    return cabi.eeg_filtfilt(eeg_data, sos)
