"""TEST INFRASTRUCTURE ONLY: fp32 torch-CPU restatement of the detector rows C1-C4, functional over a
state_dict so that it shares no code with the product modules.  Pinned by tests/golden/detector_*.npz, generated from
the reference's own modules by tests/golden/make_detector_golden.py.
"""
import torch
import torch.nn.functional as F


def _bn(x, sd, p, training, momentum=0.1, eps=1e-5):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], training,
                        momentum, eps)


def rpn_forward(sd, x, layer_nums, ds_strides, us_strides, training, prefix=""):
    """liso/networks/centerpoint/rpn.py:113-146: per stage ZeroPad2d(1)+conv3x3(stride)+BN+ReLU, n x (conv3x3+BN+ReLU),
    deblock (conv k=1/s stride 1/s | convT k=s stride s)+BN+ReLU, concat."""
    ups = []
    start = len(layer_nums) - len(us_strides)
    for i, n in enumerate(layer_nums):
        b = f"{prefix}blocks.{i}."
        x = F.conv2d(F.pad(x, (1, 1, 1, 1)), sd[b + "1.weight"], None, stride=ds_strides[i])
        x = F.relu(_bn(x, sd, b + "2", training))
        for j in range(n):
            x = F.conv2d(x, sd[b + f"{4 + 3 * j}.weight"], None, padding=1)
            x = F.relu(_bn(x, sd, b + f"{5 + 3 * j}", training))
        if i - start >= 0:
            d = f"{prefix}deblocks.{i - start}."
            s = us_strides[i - start]
            if s > 1:
                u = F.conv_transpose2d(x, sd[d + "0.weight"], None, stride=int(s))
            else:
                k = int(round(1 / s))
                u = F.conv2d(x, sd[d + "0.weight"], None, stride=k)
            ups.append(F.relu(_bn(u, sd, d + "1", training)))
    return torch.cat(ups, dim=1)


def center_head_forward(sd, x, heads, training, prefix=""):
    """liso/networks/centerpoint/center_head.py:60-65,109-117: shared conv3x3+BN+ReLU; per head conv3x3+BN+ReLU+conv3x3"""
    p = prefix + "shared_conv."
    x = F.relu(_bn(F.conv2d(x, sd[p + "0.weight"], sd[p + "0.bias"], padding=1), sd, p + "1", training))
    out = {}
    for h in heads:
        q = f"{prefix}tasks.0.{h}."
        y = F.relu(_bn(F.conv2d(x, sd[q + "0.weight"], sd[q + "0.bias"], padding=1), sd, q + "1", training))
        out[h] = F.conv2d(y, sd[q + "3.weight"], sd[q + "3.bias"], padding=1)
    return out


def decode(raw, pillar_centers, bev_range_m, z_min, z_max):
    """simple_net.py:111-151 + output_modification.py:4-45,58-127 for the centerpoint overlay
    (pos=tanh/local_relative_offset, dims=softplus/abs size, rot=vector, probs=none).  raw: dict of NHWC maps."""
    act = {"pos": torch.tanh(raw["pos"]), "dims": F.softplus(raw["dims"]), "rot": raw["rot"], "probs": raw["probs"]}
    H, W = raw["pos"].shape[1:3]
    res = (torch.tensor(bev_range_m) / torch.tensor([H, W])).to(raw["pos"].device)
    xy = pillar_centers[None] + res * 0.5 * act["pos"][..., :2]
    z = z_min + 0.5 * (act["pos"][..., [-1]] + 1.0) * (z_max - z_min)
    s, c = torch.split(act["rot"], 1, dim=-1)
    dec = {"pos": torch.cat([xy, z], -1), "dims": act["dims"], "rot": torch.atan2(s, c), "probs": act["probs"]}
    return dec, act


def centerpoint_loss(dec, act, gt, center_mask, ignore_mask, rot_weights):
    """liso/losses/centerpoint_loss.py:13-136 (+ :165-200 focal) with boolean-mask indexing exactly as written there."""
    out = {}
    num_pos = torch.clip(center_mask.sum(), min=1.0)
    logits = act["probs"]
    pp, pn = torch.sigmoid(logits), torch.sigmoid(-logits)
    pos_l = 0.5 * torch.pow(pn, 2.0) * F.logsigmoid(logits)
    neg_l = 0.5 * torch.pow(pp, 2.0) * torch.pow(1.0 - gt["probs"], 4.0) * F.logsigmoid(-logits)
    out["probs"] = -(pos_l[center_mask & ~ignore_mask].sum() + neg_l[~center_mask & ~ignore_mask].sum()) / num_pos
    sel = center_mask & ~ignore_mask
    if sel.sum() > 0:
        w = torch.maximum(rot_weights[sel], torch.tensor(0.1, device=sel.device))
        w = w / torch.maximum(w.sum(), torch.tensor(1.0, device=sel.device))
        out["rot"] = 10 * torch.sum(F.l1_loss(act["rot"][sel], gt["rot"][sel], reduction="none") * w)
        out["dims"] = F.l1_loss(dec["dims"][sel], gt["dims"][sel]).sum() / num_pos
        out["pos"] = F.l1_loss(dec["pos"][sel], gt["pos"][sel]).sum() / num_pos
    return out


def rotation_regulariser(act):
    """liso/kabsch/main_utils.py:51-58"""
    n = torch.norm(act["rot"], dim=-1)
    return F.mse_loss(n, torch.ones_like(n))
