"""Import-path shim for /root/reference/utils/EEGDataset.py (the Spampinato dataset: split file, subject filter,
per-channel dataset-level normalisation at load)."""
import functools

from cerebralsignalnetworks_amd import dataset as _ds


class EEGDataset(_ds.EEGDataset):
    __init__ = functools.partialmethod(_ds.EEGDataset.__init__, flavour="spampinato")
