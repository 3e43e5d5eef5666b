"""adam_onecycle: Adam(beta2 0.99) with decoupled weight decay applied before the step and a one-cycle
cosine schedule of lr and beta1.  Same update rule as the reference's OptimWrapper/OneCycle
(tools/train_utils/optimization/fastai_optim.py:135-152, learning_schedules_fastai.py:44-77,
optimization/__init__.py:19-32) without the per-layer-group Python bookkeeping: one foreach multiply for
the decay and one fused Adam launch."""
import math

import torch


def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)


class AdamOneCycle:
    """p <- p * (1 - wd*lr) for every trainable parameter (true_wd, bn_wd), then Adam.step()."""

    def __init__(self, params, lr=3e-3, wd=0.01, betas=(0.9, 0.99)):
        self.params = [p for p in params if p.requires_grad]
        fused = all(p.is_cuda for p in self.params) and len(self.params) > 0
        self.opt = torch.optim.Adam(self.params, lr=lr, betas=betas, weight_decay=0.0, fused=fused)
        self.wd = wd
        self.lr, self.mom = lr, betas[0]

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, v):
        self._lr = float(v)
        for g in self.opt.param_groups:
            g['lr'] = self._lr

    @property
    def mom(self):
        return self._mom

    @mom.setter
    def mom(self, v):
        self._mom = float(v)
        for g in self.opt.param_groups:
            g['betas'] = (self._mom, g['betas'][1])

    def zero_grad(self, set_to_none=True):
        self.opt.zero_grad(set_to_none=set_to_none)

    @torch.no_grad()
    def step(self):
        ps = [p for p in self.params if p.grad is not None]
        if ps and self.wd != 0.0:
            torch._foreach_mul_(ps, 1.0 - self.wd * self._lr)
        self.opt.step()

    def state_dict(self):
        return self.opt.state_dict()

    def load_state_dict(self, sd):
        self.opt.load_state_dict(sd)


class OneCycle:
    """lr: low -> lr_max over pct_start of the steps, then -> low/1e4; beta1: moms[0] -> moms[1] -> moms[0]."""

    def __init__(self, optimizer, total_step, lr_max, moms, div_factor, pct_start):
        self.optimizer, self.total_step = optimizer, max(int(total_step), 1)
        low = lr_max / div_factor
        split = int(self.total_step * pct_start)
        self.lr_phases = [(0, split, low, lr_max), (split, self.total_step, lr_max, low / 1e4)]
        self.mom_phases = [(0, split, moms[0], moms[1]), (split, self.total_step, moms[1], moms[0])]
        optimizer.lr, optimizer.mom = low, moms[0]

    def step(self, step):
        for start, end, a, b in self.lr_phases:
            if step >= start and end > start:
                self.optimizer.lr = annealing_cos(a, b, (step - start) / (end - start))
        for start, end, a, b in self.mom_phases:
            if step >= start and end > start:
                self.optimizer.mom = annealing_cos(a, b, (step - start) / (end - start))


def build_optimizer(model, optim_cfg):
    if optim_cfg.OPTIMIZER != 'adam_onecycle':
        raise NotImplementedError('the T-MAE recipe uses OPTIMIZER: adam_onecycle (t_mae_ssl.yaml:188)')
    return AdamOneCycle(model.parameters(), lr=3e-3, wd=optim_cfg.WEIGHT_DECAY, betas=(0.9, 0.99))


def build_scheduler(optimizer, total_iters_each_epoch, total_epochs, last_epoch, optim_cfg):
    total = total_iters_each_epoch * total_epochs
    return OneCycle(optimizer, total, optim_cfg.LR, list(optim_cfg.MOMS), optim_cfg.DIV_FACTOR,
                    optim_cfg.PCT_START), None
