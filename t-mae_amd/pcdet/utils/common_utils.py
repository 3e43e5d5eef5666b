"""The pcdet.utils.common_utils entry points the training driver uses (common_utils.py:148-241)."""
import logging
import os
import random

import numpy as np
import torch
import torch.distributed as dist


def create_logger(log_file=None, rank=0, log_level=logging.INFO):
    logger = logging.getLogger(__name__)
    logger.setLevel(log_level if rank == 0 else 'ERROR')
    fmt = logging.Formatter('%(asctime)s  %(levelname)5s  %(message)s')
    if not logger.handlers:
        console = logging.StreamHandler()
        console.setLevel(log_level if rank == 0 else 'ERROR')
        console.setFormatter(fmt)
        logger.addHandler(console)
        if log_file is not None:
            fh = logging.FileHandler(filename=log_file)
            fh.setLevel(log_level if rank == 0 else 'ERROR')
            fh.setFormatter(fmt)
            logger.addHandler(fh)
    logger.propagate = False
    return logger


def set_random_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def init_dist_pytorch(tcp_port=None, local_rank=None, backend='nccl'):
    """One process per GPU; 'nccl' is RCCL on ROCm.  Reads the torchrun environment."""
    local_rank = int(os.environ.get('LOCAL_RANK', local_rank or 0))
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
    if not dist.is_initialized():
        dist.init_process_group(backend=backend)
    return dist.get_world_size(), dist.get_rank()


def get_dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1
