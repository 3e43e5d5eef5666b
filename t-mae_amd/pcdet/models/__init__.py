"""build_network / load_data_to_gpu / model_fn_decorator with the reference's contract
(pcdet/models/__init__.py:10-41)."""
from collections import namedtuple

import numpy as np
import torch

from .detectors import build_detector


def build_network(model_cfg, num_class, dataset, logger=None):
    return build_detector(model_cfg=model_cfg, num_class=num_class, dataset=dataset, logger=logger)


def load_data_to_gpu(batch_dict):
    for key, val in batch_dict.items():
        if not isinstance(val, np.ndarray):
            continue
        if key in ['frame_id', 'metadata', 'calib', 'image_shape', 'image_pad_shape', 'image_rescale_shape']:
            continue
        batch_dict[key] = torch.from_numpy(val).float().cuda()


def model_fn_decorator():
    ModelReturn = namedtuple('ModelReturn', ['loss', 'tb_dict', 'disp_dict'])

    def model_func(model, batch_dict, **kwargs):
        load_data_to_gpu(batch_dict)
        ret_dict, tb_dict, disp_dict = model(batch_dict)
        loss = ret_dict['loss'].mean()
        (model if hasattr(model, 'update_global_step') else model.module).update_global_step()
        return ModelReturn(loss, tb_dict, disp_dict)

    return model_func
