"""MI355X-native EEG -> LSTM -> DINOv2-distillation hot path (drop-in for the reference's
models.lstm.Model / EEG filtering / retrieval call sites).  Native code lives in
``csrc/`` behind the C ABI of ``include/csn_hip.h``; see DESIGN.md."""
from . import cabi  # noqa: F401
from .filters import EEGFilters, eeg_bandpass_znorm, remove_noise  # noqa: F401
from .lstm_model import Model, LSTMModel, CustomModel  # noqa: F401
from .losses import (CosineSimilarityLoss, FeatureDistributionLoss, loss_fn_kd,  # noqa: F401
                     BarlowTwinsLoss, HyperParams)
from .retrieval import evaluate, l2_search  # noqa: F401
