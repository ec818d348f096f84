"""compute_knn_loss_components.  Mirror of liso/slim/slim_loss/knn_loss.py:9-82."""
import torch

from liso_amd.slim.slim_loss.knn_wrapper import NearestPointLoss, compute_flow_loss_a_to_b
from liso_amd.utils.config import AttrDict as Munch


def compute_knn_loss_components(pcl_t0, valid_mask_t0, pcl_t1, valid_mask_t1, *, prediction, loss_cfg, model_cfg, bev_extent,
                                knn_indices=None, query_order_indices=None):
    assert pcl_t0.ndim == 3 and pcl_t0.size(2) == 3 and pcl_t1.ndim == 3 and pcl_t1.size(2) == 3
    assert prediction.static_flow.ndim == 3 and prediction.static_flow.size(2) == 3
    types = {"aggregated"}  # reference :27-43
    if loss_cfg.artificial_labels.cross_entropy_penalty > 0.0:
        types.add("dynamic")
        types.add("static_aggr" if loss_cfg.artificial_labels.use_static_aggr_flow else "static")
    if loss_cfg.knn_on_dynamic_penalty != 0.0:
        types.add("dynamic")
    if loss_cfg.knn_on_static_penalty != 0.0:
        types.add("static_aggr" if model_cfg.use_static_aggr_flow_for_aggr_flow else "static")
    types = sorted(types)  # deterministic order (the reference iterates a set)
    nan = float("nan")
    pcl_t0 = torch.where(valid_mask_t0[..., None], pcl_t0, torch.full_like(pcl_t0, nan))  # :44-45 (out of place)
    pcl_t1 = torch.where(valid_mask_t1[..., None], pcl_t1, torch.full_like(pcl_t1, nan))
    flows = [torch.where(valid_mask_t0[..., None], prediction["%s_flow" % t], torch.full_like(pcl_t0, nan)) for t in types]
    bs = flows[0].size(0)
    n = len(types)
    loss, knn = compute_flow_loss_a_to_b(torch.cat([pcl_t0] * n, 0), torch.cat([pcl_t1] * n, 0), torch.cat(flows, 0),
                                         loss_function=NearestPointLoss(bev_extent=bev_extent, **loss_cfg.knn_loss),
                                         nearest_dist_mode=loss_cfg.knn_dist_measure,
                                         knn_indices=None if knn_indices is None else list(knn_indices) * n,
                                         query_order_indices=None if query_order_indices is None else list(query_order_indices) * n)
    return {t: {"loss": loss[i * bs:(i + 1) * bs], "knn": Munch(**{k: v[i * bs:(i + 1) * bs] for k, v in knn.items()})}
            for i, t in enumerate(types)}
