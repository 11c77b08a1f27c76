from cerebralsignalnetworks_amd.retrieval import evaluate, evaluate_full  # noqa: F401
