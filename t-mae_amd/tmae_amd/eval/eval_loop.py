"""eval_one_epoch (tools/eval_utils/eval_utils.py:24-161): no-grad forward over the evaluation split, recall record,
prediction dictionaries, merge over ranks, ONCE AP.  The reference merges through pickle files on a shared file system
and two barriers (pcdet/utils/common_utils.py:244-265); here it is one all_gather_object."""
import pickle
import time
from pathlib import Path

import torch
import torch.distributed as dist


def statistics_info(cfg, ret_dict, metric, disp_dict):
    for key in metric.keys():
        if key in ret_dict:
            metric[key] += ret_dict[key]
    t = cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST[0]
    disp_dict['recall_%s' % str(t)] = '(%d, %d) / %d' % (metric['recall_roi_%s' % str(t)],
                                                         metric['recall_rcnn_%s' % str(t)], metric['gt_num'])


def merge_results_dist(result_part, size):
    """Every rank's list, interleaved in sampler order (rank r holds samples r, r + W, ...) and cut to `size`; rank 0
    gets the merged list, the others None (common_utils.py:244-265 semantics)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return result_part[:size]
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, result_part)
    if dist.get_rank() != 0:
        return None
    ordered = []
    for i in range(max(len(p) for p in parts)):
        for p in parts:
            if i < len(p):
                ordered.append(p[i])
    return ordered[:size]


def eval_one_epoch(cfg, model, dataloader, epoch_id, logger, dist_test=False, save_to_file=False, result_dir=None,
                   amp_dtype=None):
    from pcdet.models import load_data_to_gpu
    result_dir = Path(result_dir)
    result_dir.mkdir(parents=True, exist_ok=True)
    metric = {'gt_num': 0}
    for t in cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST:
        metric['recall_roi_%s' % str(t)] = 0
        metric['recall_rcnn_%s' % str(t)] = 0
    dataset = dataloader.dataset
    class_names = dataset.class_names
    det_annos = []
    logger.info('*************** EPOCH %s EVALUATION *****************' % epoch_id)
    rank = dist.get_rank() if (dist_test and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist_test and dist.is_initialized()) else 1
    was_training = model.training
    model.eval()
    start, run_time = time.time(), 0.0
    for batch_dict in dataloader:
        torch.cuda.synchronize()
        t0 = time.time()
        load_data_to_gpu(batch_dict)
        with torch.no_grad(), torch.autocast('cuda', dtype=amp_dtype or torch.bfloat16, enabled=amp_dtype is not None):
            pred_dicts, ret_dict = model(batch_dict)
        torch.cuda.synchronize()
        run_time += time.time() - t0
        disp_dict = {}
        statistics_info(cfg, ret_dict, metric, disp_dict)
        det_annos += dataset.generate_prediction_dicts(batch_dict, pred_dicts, class_names, output_path=None)
    model.train(was_training)
    if dist_test:
        det_annos = merge_results_dist(det_annos, len(dataset))
        metrics = merge_results_dist([metric], world)
    else:
        metrics = [metric]
    n_local = max(len(dataset) / world, 1)
    logger.info('*************** Performance of EPOCH %s *****************' % epoch_id)
    logger.info('Run time per sample: %.4f second.' % (run_time / n_local))
    logger.info('Generate label finished(sec_per_example: %.4f second).' % ((time.time() - start) / n_local))
    if rank != 0:
        return {}
    total = dict(metrics[0])
    for other in metrics[1:]:
        for k, v in other.items():
            total[k] += v
    ret = {}
    gt_num = total['gt_num']
    for t in cfg.MODEL.POST_PROCESSING.RECALL_THRESH_LIST:
        for kind in ('roi', 'rcnn'):
            r = total['recall_%s_%s' % (kind, str(t))] / max(gt_num, 1)
            logger.info('recall_%s_%s: %f' % (kind, t, r))
            ret['recall/%s_%s' % (kind, str(t))] = r
    n_obj = sum(len(a['name']) for a in det_annos)
    logger.info('Average predicted number of objects(%d samples): %.3f' % (len(det_annos), n_obj / max(1, len(det_annos))))
    with open(result_dir / 'result.pkl', 'wb') as f:
        pickle.dump(det_annos, f)
    result_str, result_dict = dataset.evaluation(det_annos, class_names,
                                                 eval_metric=cfg.MODEL.POST_PROCESSING.get('EVAL_METRIC', 'once'),
                                                 output_path=result_dir)
    logger.info(result_str)
    ret.update(result_dict)
    logger.info('Result is save to %s' % result_dir)
    logger.info('****************Evaluation done.*****************')
    return ret
