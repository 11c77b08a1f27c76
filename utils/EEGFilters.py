from cerebralsignalnetworks_amd.filters import EEGFilters  # noqa: F401
