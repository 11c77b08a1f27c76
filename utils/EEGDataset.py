from cerebralsignalnetworks_amd.dataset import EEGDataset  # noqa: F401
