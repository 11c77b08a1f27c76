from cerebralsignalnetworks_amd.retrieval import evaluate, evaluate_full  # noqa: F401
from cerebralsignalnetworks_amd.filters import remove_noise  # noqa: F401,E402


class Utilities:
    """Holder with the reference's method name (Utilities().remove_noise, Utilities.py:411)."""

    def remove_noise(self, eeg_data, sampling_rate):
        return remove_noise(eeg_data, sampling_rate)
