"""Drop-in for the reference's missing ``models/lstm.py`` (imported at LstmDistillFromDinoV2Train.py:5)."""
from cerebralsignalnetworks_amd.lstm_model import Model  # noqa: F401
