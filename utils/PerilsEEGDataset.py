"""Import-path shim for /root/reference/utils/PerilsEEGDataset.py (all records, scalar dataset-level statistics)."""
import functools

from cerebralsignalnetworks_amd import dataset as _ds


class EEGDataset(_ds.EEGDataset):
    __init__ = functools.partialmethod(_ds.EEGDataset.__init__, flavour="perils")
