"""Import-path shim for /root/reference/utils/utils.py: ``from utils import utils`` (LstmDistillFromDinoV2Train.py:9,
utils/PerilsEEGDataset.py:9, LstmDistillation.py:14) resolves here.  The runtime helpers live in
cerebralsignalnetworks_amd/runtime.py; the DINO pieces and LARS in dino.py / losses.py.  Not provided (out of the hot
path): the image augmentations, ViT weight loaders, PCA / mAP retrieval metrics, ``multi_scale``."""
from cerebralsignalnetworks_amd.runtime import (  # noqa: F401
    SmoothedValue, MetricLogger, reduce_dict, is_dist_avail_and_initialized, get_world_size, get_rank, is_main_process,
    save_on_master, setup_for_distributed, init_distributed_mode, clip_gradients, cancel_gradients_last_layer,
    get_params_groups, has_batchnorms, bool_flag, fix_random_seeds, restart_from_checkpoint, accuracy)
from cerebralsignalnetworks_amd.dino import MultiCropWrapper, cosine_scheduler  # noqa: F401
from cerebralsignalnetworks_amd.losses import LARS  # noqa: F401
