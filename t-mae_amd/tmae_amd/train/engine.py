"""One training step of the T-MAE pre-training loop (tools/train_utils/train_utils.py:59-100) and the
data-parallel wrapper (tools/train.py:283-289): one process per GPU, DDP over RCCL ('nccl' on ROCm),
bf16 autocast (no GradScaler needed), decoupled-decay Adam one-cycle."""

import torch
import torch.distributed as dist


def build_model_from_cfg(cfg, dataset, logger=None):
    from pcdet.models import build_network
    return build_network(model_cfg=cfg.MODEL, num_class=len(cfg.CLASS_NAMES), dataset=dataset, logger=logger)


def wrap_ddp(model, local_rank):
    """Frame-pair data parallel: gradients (11.8 M params, 47 MB fp32) are all-reduced bucket-wise during
    backward; BatchNorm statistics stay per-rank (SYNC_BN off in the shipped configs).
    Buckets of 25 MB: the decoder / stage-3 half of the gradients is on the xGMI ring while the backward of stages 1-2
    still runs, and only the last bucket (~20 MB, ~0.5 ms on 8 GPUs) is exposed before the optimizer step; one 64 MB
    bucket (round 1) could not start before the last gradient existed."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    if next(model.parameters()).is_cuda:
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=True,
                                                         gradient_as_bucket_view=True, bucket_cap_mb=25)
    return torch.nn.parallel.DistributedDataParallel(model)


_GC_SETTLED = False


def settle_gc():
    """Once, after the first training step: collect what set-up and the first step left behind and FREEZE the survivors (the
    interpreter's ~275 k long-lived objects: torch itself, the module tree, the optimizer state, cached index tables) out of the
    cyclic collector's sight.  Python's full collection is triggered by allocation COUNTS, walks every tracked object and, with
    that heap, takes 80 - 190 ms in the launching thread: in round 5 it fell on the 10th step of every run started as
    `python bench.py --warmup 5 --steps 20` -- one stalled step, +3.5 ms on the 20-step mean, queue drained, GPU idle -- and moved
    with every unrelated edit that changed the number of objects created at import (profiles/round5_gc_stall.txt).  After the
    freeze a full collection only looks at objects made since (about a thousand), i.e. it stays automatic and costs microseconds."""
    global _GC_SETTLED
    if _GC_SETTLED:
        return
    import gc
    gc.collect()
    gc.freeze()
    _GC_SETTLED = True


def unsettle_gc():
    """Undo settle_gc(): the frozen objects go back under the collector (gc.unfreeze) and the next train_one_step(settle=True)
    freezes again.  For a long-lived process that builds model after model (a sweep, a notebook): call it when a model and its
    optimizer are dropped, or their reference cycles stay in the permanent generation for the life of the process."""
    global _GC_SETTLED
    import gc
    gc.unfreeze()
    _GC_SETTLED = False


def train_one_step(model, optimizer, scheduler, batch_dict, it, model_func, amp_dtype=torch.bfloat16,
                   grad_norm_clip=None, settle=None, next_batch=None):
    """lr_scheduler.step -> zero_grad -> autocast forward -> backward (DDP all-reduce overlaps) ->
    [clip, non-AMP branch only: train_utils.py:88-93] -> optimizer.step.
    settle: freeze the interpreter's long-lived objects out of the cyclic collector after the first step (settle_gc: a
    process-wide, one-way change until unsettle_gc()).  None = the TMAE_SETTLE_GC environment switch, default ON -- the
    training drivers (tools/train.py, bench.py) are one-model processes; a host that builds many models passes False or calls
    unsettle_gc() between them.
    next_batch: the batch_dict the NEXT call will be given (the same dict object): its voxelisation is enqueued between this
    step's forward and backward (TemporalDynVFE.prefetch), so that the next step does not begin with a host stall."""
    if scheduler is not None:
        scheduler.step(it)
    optimizer.zero_grad(set_to_none=True)
    use_amp = amp_dtype is not None
    with torch.autocast('cuda', dtype=amp_dtype if use_amp else torch.bfloat16, enabled=use_amp):
        loss, tb_dict, disp_dict = model_func(model, batch_dict)
    if next_batch is not None:
        vfe = getattr(getattr(model, 'module', model), 'vfe', None)
        if vfe is not None and hasattr(vfe, 'prefetch'):
            vfe.prefetch(next_batch)
    loss.backward()
    if not use_amp and grad_norm_clip:
        torch.nn.utils.clip_grad_norm_(model.parameters(), grad_norm_clip)
    if use_amp and hasattr(optimizer, '_native_step'):
        optimizer.step(copy_dtype=amp_dtype)       # the one-launch step also refreshes the bf16 copies it can
    else:
        optimizer.step()
    if use_amp:
        from .. import ops
        ops.refresh_param_copies(optimizer.params if hasattr(optimizer, 'params') else model.parameters(), amp_dtype)
    if settle is None:
        import os
        settle = os.environ.get('TMAE_SETTLE_GC', '1') != '0'
    if settle:
        settle_gc()                                 # first call only
    return loss, tb_dict, disp_dict
