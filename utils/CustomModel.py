from cerebralsignalnetworks_amd.lstm_model import CustomModel  # noqa: F401
