"""Detector template + TMAE detector with the reference's registry contract.

Mirror of pcdet/models/detectors/detector3d_template.py:15-100,365-451 and t_mae.py:4-34: modules are looked
up by NAME in per-package registries, called in order on ``batch_dict``; train mode returns
``({'loss': loss}, tb_dict, disp_dict)``.  Checkpoints are ``{'model_state': state_dict, ...}``.
"""
import os

import torch
import torch.nn as nn


class Detector3DTemplate(nn.Module):
    def __init__(self, model_cfg, num_class, dataset, logger=None):
        super().__init__()
        self.model_cfg, self.num_class, self.dataset, self.logger = model_cfg, num_class, dataset, logger
        self.class_names = dataset.class_names
        self.register_buffer('global_step', torch.LongTensor(1).zero_())
        self.module_topology = ['vfe', 'backbone_3d', 'backbone_2d', 'dense_head']   # the entries the T-MAE configs use

    @property
    def mode(self):
        return 'TRAIN' if self.training else 'TEST'

    def update_global_step(self):
        self.global_step += 1

    def build_networks(self):
        info = {
            'module_list': [],
            'num_rawpoint_features': self.dataset.point_feature_encoder.num_point_features,
            'num_point_features': self.dataset.point_feature_encoder.num_point_features,
            'grid_size': self.dataset.grid_size,
            'point_cloud_range': self.dataset.point_cloud_range,
            'voxel_size': self.dataset.voxel_size,
        }
        for name in self.module_topology:
            module, info = getattr(self, 'build_%s' % name)(model_info_dict=info)
            self.add_module(name, module)
        for unsupported in ('MAP_TO_BEV', 'PFE', 'POINT_HEAD', 'ROI_HEAD', 'IMG_BACKBONE'):
            if self.model_cfg.get(unsupported, None) is not None:
                raise NotImplementedError(f'MODEL.{unsupported}: not used by the T-MAE configs (SURVEY 8f)')
        return info['module_list']

    def build_vfe(self, model_info_dict):
        from . import registry
        if self.model_cfg.get('VFE', None) is None:
            return None, model_info_dict
        m = registry.VFE[self.model_cfg.VFE.NAME](
            model_cfg=self.model_cfg.VFE, num_point_features=model_info_dict['num_rawpoint_features'],
            point_cloud_range=model_info_dict['point_cloud_range'], voxel_size=model_info_dict['voxel_size'],
            grid_size=model_info_dict['grid_size'])
        model_info_dict['num_point_features'] = m.get_output_feature_dim()
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def build_backbone_3d(self, model_info_dict):
        from . import registry
        if self.model_cfg.get('BACKBONE_3D', None) is None:
            return None, model_info_dict
        m = registry.BACKBONES_3D[self.model_cfg.BACKBONE_3D.NAME](
            model_cfg=self.model_cfg.BACKBONE_3D, input_channels=model_info_dict['num_point_features'],
            grid_size=model_info_dict['grid_size'], voxel_size=model_info_dict['voxel_size'],
            point_cloud_range=model_info_dict['point_cloud_range'])
        model_info_dict['module_list'].append(m)
        model_info_dict['num_point_features'] = m.num_point_features
        return m, model_info_dict

    def build_backbone_2d(self, model_info_dict):
        from . import registry
        if self.model_cfg.get('BACKBONE_2D', None) is None:
            return None, model_info_dict
        m = registry.BACKBONES_2D[self.model_cfg.BACKBONE_2D.NAME](
            model_cfg=self.model_cfg.BACKBONE_2D, input_channels=model_info_dict.get('num_bev_features', None))
        model_info_dict['module_list'].append(m)
        model_info_dict['num_bev_features'] = m.num_bev_features
        return m, model_info_dict

    def build_dense_head(self, model_info_dict):
        from . import registry
        if self.model_cfg.get('DENSE_HEAD', None) is None:
            return None, model_info_dict
        m = registry.DENSE_HEADS[self.model_cfg.DENSE_HEAD.NAME](
            model_cfg=self.model_cfg.DENSE_HEAD, input_channels=model_info_dict['num_bev_features'],
            num_class=self.num_class if not self.model_cfg.DENSE_HEAD.CLASS_AGNOSTIC else 1,
            class_names=self.class_names, grid_size=model_info_dict['grid_size'],
            point_cloud_range=model_info_dict['point_cloud_range'],
            predict_boxes_when_training=self.model_cfg.get('ROI_HEAD', False),
            voxel_size=model_info_dict.get('voxel_size', False))
        model_info_dict['module_list'].append(m)
        return m, model_info_dict

    def forward(self, **kwargs):
        raise NotImplementedError

    # ---- checkpoints (detector3d_template.py:398-451; train_utils.py:245-270)
    def _load_state_dict(self, model_state_disk, *, strict=True):
        """detector3d_template.py:365-396: keep entries whose name and shape match; sparse-conv weights saved in
        another spconv layout are adapted first -- (..., c_in, c_out) -> (..., c_out, c_in) by a transpose, or the
        spconv-1 kernel-major layout (k1, k2[, k3], c_in, c_out) -> (c_out, k1, k2[, k3], c_in)."""
        from .sparse import SparseConvolution
        state_dict = self.state_dict()
        spconv_keys = {f'{n}.weight' for n, m in self.named_modules() if isinstance(m, SparseConvolution)}
        update = {}
        for key, val in model_state_disk.items():
            if key in spconv_keys and key in state_dict and state_dict[key].shape != val.shape:
                native = val.transpose(-1, -2)
                if native.shape == state_dict[key].shape:
                    val = native.contiguous()
                else:
                    implicit = val.permute(val.dim() - 1, *range(val.dim() - 1))
                    if implicit.shape == state_dict[key].shape:
                        val = implicit.contiguous()
            if key in state_dict and state_dict[key].shape == val.shape:
                update[key] = val
        if strict:
            self.load_state_dict(update)
        else:
            state_dict.update(update)
            self.load_state_dict(state_dict)
        return state_dict, update

    def load_params_from_file(self, filename, logger=None, to_cpu=False):
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        ckpt = torch.load(filename, map_location='cpu' if to_cpu else None, weights_only=False)
        own, update = self._load_state_dict(ckpt['model_state'], strict=False)
        if logger is not None:
            if ckpt.get('version', None) is not None:
                logger.info('==> Checkpoint trained from version: %s' % ckpt['version'])
            for k in own:
                if k not in update:
                    logger.info('Not updated weight %s: %s' % (k, str(own[k].shape)))
            logger.info('==> Done (loaded %d/%d)' % (len(update), len(own)))

    def load_params_with_optimizer(self, filename, to_cpu=False, optimizer=None, logger=None):
        if not os.path.isfile(filename):
            raise FileNotFoundError(filename)
        ckpt = torch.load(filename, map_location='cpu' if to_cpu else None, weights_only=False)
        self._load_state_dict(ckpt['model_state'], strict=True)
        if optimizer is not None and ckpt.get('optimizer_state') is not None:
            try:
                optimizer.load_state_dict(ckpt['optimizer_state'])
            except ValueError as e:          # another optimizer / parameter set: keep the weights, restart the moments
                msg = f'optimizer_state of {filename} not loaded ({e}); Adam moments start from zero'
                (logger.warning if logger is not None else print)(msg)
        # 'epoch' = number of epochs already trained (train_utils.py:217-232 saves cur_epoch + 1 and the reference
        # resumes with range(epoch, total), tools/train.py:261-312)
        return ckpt.get('it', 0.0), ckpt.get('epoch', -1)


class TMAE(Detector3DTemplate):
    def __init__(self, model_cfg, num_class, dataset, logger=None):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset, logger=logger)
        self.module_list = self.build_networks()

    def forward(self, batch_dict):
        from .. import ops
        with ops.defer_bn_updates():            # running statistics: one multi-tensor update per forward pass
            for cur_module in self.module_list:
                batch_dict = cur_module(batch_dict)
        if self.training:
            loss, tb_dict, disp_dict = self.get_training_loss()
            return {'loss': loss}, tb_dict, disp_dict
        return self.post_processing(batch_dict)

    def post_processing(self, batch_dict):
        return {}, {}

    def get_training_loss(self):
        loss_rpn, tb_dict = self.backbone_3d.get_loss()
        # the reference calls loss.item() here (t_mae.py:31): a host sync per step; keep the tensor instead
        tb_dict = {'loss_rpn': loss_rpn.detach(), **tb_dict}
        return loss_rpn, tb_dict, {}


class CenterPoint(Detector3DTemplate):
    """pcdet/models/detectors/centerpoint.py:4-50 (training path)."""

    def __init__(self, model_cfg, num_class, dataset, logger=None):
        super().__init__(model_cfg=model_cfg, num_class=num_class, dataset=dataset, logger=logger)
        self.module_list = self.build_networks()

    def forward(self, batch_dict):
        from .. import ops
        with ops.defer_bn_updates():            # running statistics: one multi-tensor update per forward pass
            for cur_module in self.module_list:
                batch_dict = cur_module(batch_dict)
        if self.training:
            loss, tb_dict, disp_dict = self.get_training_loss()
            return {'loss': loss}, tb_dict, disp_dict
        return self.post_processing(batch_dict)

    def post_processing(self, batch_dict):
        """centerpoint.py:35-50 + Detector3DTemplate.generate_recall_record (detector3d_template.py:319-363)."""
        from .. import ops
        thresh_list = self.model_cfg.POST_PROCESSING.RECALL_THRESH_LIST
        final = batch_dict['final_box_dicts']
        recall = {}
        for index in range(int(batch_dict['batch_size'])):
            if 'gt_boxes' not in batch_dict:
                continue
            if not recall:
                recall = {'gt_num': 0}
                for t in thresh_list:
                    recall['recall_roi_%s' % str(t)] = 0
                    recall['recall_rcnn_%s' % str(t)] = 0
            gt = batch_dict['gt_boxes'][index]
            k = gt.shape[0] - 1
            nz = (gt.abs().sum(1) != 0).nonzero()
            gt = gt[:int(nz.max()) + 1] if nz.numel() else gt[:0]
            if gt.shape[0] > 0:
                boxes = final[index]['pred_boxes']
                iou = ops.boxes_iou3d_gpu(boxes[:, 0:7], gt[:, 0:7]) if boxes.shape[0] > 0 else None
                for t in thresh_list:
                    if iou is not None:
                        recall['recall_rcnn_%s' % str(t)] += int((iou.max(dim=0)[0] > t).sum())
                recall['gt_num'] += gt.shape[0]
        return final, recall

    def get_training_loss(self):
        loss_rpn, tb_dict = self.dense_head.get_loss()
        tb_dict = {'loss_rpn': loss_rpn.detach(), **tb_dict}      # the reference calls .item() here: a host sync
        return loss_rpn, tb_dict, {}
