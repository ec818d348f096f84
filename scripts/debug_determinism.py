import torch
from liso_amd.datasets.synthetic import slim_pair
from liso_amd.trainer import LisoLoopTrainer
from liso_amd.utils.config import apply_slim_simple_knn_training, default_cfg
dev = torch.device("cuda")
s0, s1 = slim_pair(2, dev, n_points=120000, grid=512, bev_range_m=100.0)
res = []
for rep in range(2):
    cfg = apply_slim_simple_knn_training(default_cfg(grid=512, bev_range_m=100.0))
    torch.manual_seed(0)
    tr = LisoLoopTrainer(cfg, dev, compute_dtype=torch.bfloat16, total_steps=8)
    with torch.no_grad():
        f1 = tr.slim.infer_point_flow_t0_t1(s0, s1).clone()
        f2 = tr.slim.infer_point_flow_t0_t1(s0, s1).clone()
    b1, _ = tr.mine_boxes(s0, s1)
    from liso_amd.datasets.targets import render_center_targets
    out = (128, 128)
    tg = render_center_targets(b1.pos.float(), b1.dims.float().clamp(min=1e-3), b1.rot.float(), b1.valid, out, (100.0, 100.0))
    tr.detector.model.train()
    l1, _, _ = tr.detector.loss(s0["pcl_full_no_ground_ta"], tg)
    l2, _, _ = tr.detector.loss(s0["pcl_full_no_ground_ta"], tg)
    res.append((f1, b1, tg, float(l1), float(l2)))
    print("rep", rep, "flow same-trainer bitwise", torch.equal(f1, f2), "loss twice", float(l1), float(l2))
print("flow across trainers max diff", float((res[0][0] - res[1][0]).abs().max()), "pos diff", float((res[0][1].pos - res[1][1].pos).abs().max()),
      "rot diff", float((res[0][1].rot - res[1][1].rot).abs().max()))
for k in res[0][2]:
    print("target", k, float((res[0][2][k].float() - res[1][2][k].float()).abs().max()))
