"""mirror of liso/slim/slim_loss/artificial_labels_pytorch.py:11-64"""
import torch


def constant_labels(*, static_knn_results, dynamic_knn_results, knn_dist_sqr_key: str):
    lab = (static_knn_results[knn_dist_sqr_key] <= dynamic_knn_results[knn_dist_sqr_key]).to(torch.float)
    return lab, torch.ones_like(lab)


def compute_artificial_label_loss(*, prediction, knn_results, loss_cfg):
    assert loss_cfg.artificial_labels.knn_mode == "point" and loss_cfg.artificial_labels.weight_mode in {"constant"}
    key = "static_aggr" if loss_cfg.artificial_labels.use_static_aggr_flow else "static"
    lab, w = constant_labels(static_knn_results=knn_results[key]["knn"], dynamic_knn_results=knn_results["dynamic"]["knn"],
                             knn_dist_sqr_key="nearest_dist_sqr")
    return torch.nn.BCELoss(reduction="none")(prediction["staticness"], lab) * w.detach()
