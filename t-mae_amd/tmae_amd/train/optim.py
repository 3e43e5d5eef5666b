"""adam_onecycle: Adam(beta2 0.99) with decoupled weight decay applied before the step and a one-cycle
cosine schedule of lr and beta1 -- the update rule AND the parameter grouping of the reference's
OptimWrapper / OneCycle (tools/train_utils/optimization/__init__.py:19-32, fastai_optim.py:16-27,99-152,
learning_schedules_fastai.py:12-77), without the per-group Python loops: on the GPU the decay and the Adam update of
ALL parameter tensors are ONE launch of the library's multi-tensor kernel (tmae_adam_step, csrc/optim.hip; torch's
fused Adam + foreach multiply need ~36 launches for this model); CPU parameters (tests, small tools) take torch's Adam.

Grouping (pinned by tests/golden/O1_optimizer.npz, generated from the reference's own classes):
`flatten_model` keeps the LEAF modules of the model (modules without children) in depth-first order,
`split_bn_bias` puts the parameters of the BatchNorm leaves in param group 1 and all other leaves' in group 0.
A parameter owned directly by a module that also has children belongs to no leaf and is therefore in NO group:
in the T-MAE model these are `in_proj_weight`, `in_proj_bias` and `tau` of every attention module (54 tensors,
2 665 746 of the 11 793 218 parameters) -- the reference never updates or decays them.  `train_nonleaf=True`
(not the reference's behaviour) appends them to group 0.
The state_dict is torch.optim.Adam's with these two groups, so `optimizer_state` of a reference checkpoint
(train_utils.py:245-270) loads as it is and the other way round.
"""
import math

import numpy as np
import torch
import torch.nn as nn

_BN_TYPES = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm)


def annealing_cos(start, end, pct):
    return end + (start - end) / 2 * (math.cos(math.pi * pct) + 1)


def leaf_modules(module):
    """Depth-first list of the modules without children (optimization/__init__.py:20-26 `flatten_model`)."""
    kids = list(module.children())
    if not kids:
        return [module]
    out = []
    for k in kids:
        out += leaf_modules(k)
    return out


def reference_param_groups(model, train_nonleaf=False):
    """(non-BN parameters, BN parameters, parameters in neither) in the reference's order; requires_grad only."""
    plain, bn, seen = [], [], set()
    for leaf in leaf_modules(model):
        dst = bn if isinstance(leaf, _BN_TYPES) else plain
        for p in leaf.parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                dst.append(p)
    rest = [p for p in model.parameters() if p.requires_grad and id(p) not in seen]
    if train_nonleaf:
        plain, rest = plain + rest, []
    return plain, bn, rest


class AdamOneCycle:
    """p <- p * (1 - wd*lr) for every parameter of the two groups (true_wd, bn_wd; also those without a gradient,
    fastai_optim.py:139-150), then Adam.step().  `source` = an nn.Module (reference grouping) or an iterable of
    parameters (one plain group; tests and small tools)."""

    def __init__(self, source, lr=3e-3, wd=0.01, betas=(0.9, 0.99), train_nonleaf=False):
        if isinstance(source, nn.Module):
            plain, bn, rest = reference_param_groups(source, train_nonleaf)
            self.params = [p for p in source.parameters() if p.requires_grad]       # everything that gets a gradient
        else:
            plain, bn, rest = [p for p in source if p.requires_grad], [], []
            self.params = list(plain)
        self.unoptimized = rest
        self.decayed = plain + bn
        fused = len(self.decayed) > 0 and all(p.is_cuda for p in self.decayed)
        # two groups even when one is empty: the layout of the reference's optimizer_state
        self.opt = torch.optim.Adam([{'params': plain}, {'params': bn}], lr=lr, betas=betas, weight_decay=0.0,
                                    fused=fused)
        self.wd = wd
        self.lr, self.mom = lr, betas[0]
        # GPU path: one multi-tensor launch (fp32 contiguous CUDA parameters on one device); torch.optim.Adam stays the
        # owner of the state (exp_avg, exp_avg_sq, step) so that state_dict / load_state_dict keep the reference's format
        self._native = fused and all(p.dtype == torch.float32 and p.is_contiguous() and p.device == self.decayed[0].device
                                     for p in self.decayed)
        self._steps = {}             # id(p) -> steps taken (host mirror of the state's step tensors)
        self._table = None           # (device table, device chunk map, chunks)
        self._pinned, self._chunk0 = None, None      # event-guarded pinned staging buffers of the table (_lib.PinnedStager)
        self._warned = False

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

    @property
    def param_groups(self):
        return self.opt.param_groups

    def skip_unread_gradients(self):
        """Stop autograd from computing the gradients of the parameters in no group (NOT the reference's behaviour,
        off by default; OPTIMIZATION.SKIP_UNREAD_GRADIENTS).  Under AMP the reference's step never reads them
        (train_utils.py:86-97: gradient clipping only in the non-AMP branch, GradScaler looks at the optimizer's own
        parameters), so the weights evolve identically -- but `p.grad` of those parameters stays None, and with
        clipping (non-AMP) they DO enter the total norm: do not use it there."""
        for p in self.unoptimized:
            p.requires_grad_(False)
        self.params = [p for p in self.params if p.requires_grad]

    def zero_grad(self, set_to_none=True):
        # every parameter that receives a gradient, also the ones the optimizer does not own (the reference's
        # zero_grad leaves those to accumulate for ever, which nothing reads)
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self, copy_dtype=None):
        """copy_dtype (torch.bfloat16 under autocast): the GPU kernel also refreshes the low-precision copies of the
        parameters that tmae_amd.ops serves to the autocast forward (ops.cast_param), instead of multi-tensor copy
        launches after the step; whatever it does not write, ops.refresh_param_copies still does."""
        if self._native and self._native_step(copy_dtype):
            return
        if self.decayed and self.wd != 0.0:
            torch._foreach_mul_(self.decayed, 1.0 - self.wd * self._lr)
        for p in self.decayed:               # host mirror of the step counts: read a loaded state BEFORE torch moves it
            if p.grad is not None and id(p) not in self._steps and len(self.opt.state.get(p, {})):
                self._state_of(p)
        self.opt.step()
        for p in self.decayed:
            if p.grad is not None:
                self._steps[id(p)] = self._steps.get(id(p), 0) + 1

    def _state_of(self, p):
        st = self.opt.state[p]
        if len(st) == 0:                      # torch.optim.Adam._init_group (fused: the step lives on the device)
            st['step'] = torch.zeros((), dtype=torch.float32, device=p.device)
            st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            self._steps[id(p)] = 0
        if id(p) not in self._steps:          # a state that load_state_dict brought in: read every step count at once
            ps = [q for q in self.decayed if id(q) not in self._steps and 'step' in self.opt.state.get(q, {})]
            tens = [q for q in ps if torch.is_tensor(self.opt.state[q]['step'])]
            if tens:
                vals = torch.stack([self.opt.state[q]['step'].detach().reshape(()).float().to(p.device) for q in tens]).cpu().tolist()
                for q, v in zip(tens, vals):
                    self._steps[id(q)] = int(v)
            for q in ps:
                if id(q) not in self._steps:                  # a plain number (older checkpoints)
                    self._steps[id(q)] = int(self.opt.state[q]['step'])
                    self.opt.state[q]['step'] = torch.tensor(float(self._steps[id(q)]), dtype=torch.float32, device=q.device)
        return st

    def _native_step(self, copy_dtype=None):
        """Decay + Adam for every tensor in one launch; every table entry carries its own step number, so tensors that
        missed a gradient in some earlier step (and lag behind, as in torch.optim.Adam) stay on this path.  Returns
        False (caller takes torch's path) only for gradients the kernel cannot read (dtype / layout / device)."""
        from .._lib import lib, check, PinnedStager
        key, any_grad = [], False
        copies = [None] * len(self.decayed)
        if copy_dtype == torch.bfloat16:
            from .. import ops
            copies = ops.plain_copy_targets(self.decayed, copy_dtype)
        for p in self.decayed:
            g = p.grad
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous() or g.device != p.device):
                if not self._warned:
                    self._warned = True
                    import warnings
                    warnings.warn('AdamOneCycle: a gradient is not fp32 / contiguous / on the parameter\'s device; '
                                  'this step runs on torch.optim.Adam instead of tmae_adam_step')
                return False
        new_steps = {}
        for p, cp in zip(self.decayed, copies):
            cptr = 0 if cp is None else cp.data_ptr()
            g = p.grad
            if g is not None:
                st = self._state_of(p)
                any_grad = True
                if st['step'].device != p.device:
                    st['step'] = st['step'].to(device=p.device, dtype=torch.float32)
                new_steps[id(p)] = self._steps[id(p)] + 1
                key.append((p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(),
                            st['step'].data_ptr(), p.numel(), cptr, new_steps[id(p)]))
            else:
                key.append((p.data_ptr(), 0, 0, 0, 0, p.numel(), cptr, 0))
        if not any_grad and self.wd == 0.0:
            return True
        key = tuple(key)
        # Gradients are fresh allocations every step and the step numbers move, so the table is re-sent every step --
        # from PINNED memory with an asynchronous copy.  (torch.tensor(list, device=...) is a pageable, synchronous
        # host-to-device copy: it made the host wait for the whole backward pass here, ~30 ms per step, and the next
        # step's forward then started on an empty queue.)
        dev = self.decayed[0].device
        if self._table is None or self._table[0].shape[0] != len(key):
            nchs = np.array([(k_[5] + 4095) // 4096 for k_ in key], dtype=np.int64)
            cmap = np.repeat(np.arange(len(key), dtype=np.int32), nchs)
            self._chunk0 = np.concatenate([[0], np.cumsum(nchs)[:-1]]).astype(np.int64)
            self._pinned = PinnedStager((len(key), 8))
            dev_tab = torch.empty((len(key), 8), dtype=torch.int64, device=dev)
            dev_map = torch.from_numpy(cmap).to(dev)        # static (numels only): sent once
            self._table = (dev_tab, dev_map, int(nchs.sum()))
        rows = np.array(key, dtype=np.int64)
        rows[:, 5] |= self._chunk0 << 40
        # a staging buffer is rewritten only once the copy last queued out of it has run (the host may be steps ahead)
        self._pinned.upload(rows, self._table[0])
        tab, cmap, chunks = self._table
        g0 = self.opt.param_groups[0]
        check(lib.tmae_adam_step(tab.data_ptr(), cmap.data_ptr(), chunks, self._lr, self._mom, g0['betas'][1], g0['eps'],
                                 self.wd, torch.cuda.current_stream(tab.device).cuda_stream), 'tmae_adam_step')
        self._steps.update(new_steps)
        # the kernel wrote through raw pointers: tell autograd (the bf16 copies / folded weights of tmae_amd.ops are keyed
        # on the parameters' version counters, exactly as an in-place torch op would have moved them)
        torch.autograd.graph.increment_version(self.decayed)
        if copy_dtype == torch.bfloat16:
            ops.mark_copies_written([p for p, cp in zip(self.decayed, copies) if cp is not None])
        return True

    def state_dict(self):
        return self.opt.state_dict()

    def load_state_dict(self, sd):
        """torch.optim.Adam's state dict with the reference's two param groups.  A state whose groups do not fit
        (another optimizer type, a model with a different parameter set) is refused with a ValueError that names
        the mismatch; `Detector3DTemplate.load_params_with_optimizer` turns that into a logged warning."""
        mine = [len(g['params']) for g in self.opt.param_groups]
        theirs = [len(g['params']) for g in sd.get('param_groups', [])]
        if mine != theirs:
            raise ValueError(f'optimizer_state has param groups of sizes {theirs}, this model needs {mine} '
                             f'(non-BatchNorm / BatchNorm leaves, fastai_optim.py:16-27)')
        lr, mom = self._lr, self._mom
        self.opt.load_state_dict(sd)
        self._steps, self._table = {}, None              # the host mirror of the step counts is re-read from the new state
        g0 = self.opt.param_groups[0]
        self._lr, self._mom = float(g0.get('lr', lr)), float(g0['betas'][0])
        for g in self.opt.param_groups:              # the reference leaves weight_decay 0 in the groups (true_wd)
            g['weight_decay'] = 0.0


class OneCycle:
    """lr: low -> lr_max over pct_start of the steps, then -> low/1e4; beta1: moms[0] -> moms[1] -> moms[0]
    (learning_schedules_fastai.py:44-77; a step in the second phase is set by the second phase alone)."""

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
    opt = AdamOneCycle(model, lr=3e-3, wd=optim_cfg.WEIGHT_DECAY, betas=(0.9, 0.99),
                       train_nonleaf=bool(optim_cfg.get('TRAIN_NONLEAF_PARAMS', False)))
    if optim_cfg.get('SKIP_UNREAD_GRADIENTS', False):
        opt.skip_unread_gradients()
    return opt


def build_scheduler(optimizer, total_iters_each_epoch, total_epochs, last_epoch, optim_cfg):
    total = total_iters_each_epoch * total_epochs
    return OneCycle(optimizer, total, optim_cfg.LR, list(optim_cfg.MOMS), optim_cfg.DIV_FACTOR,
                    optim_cfg.PCT_START), None
