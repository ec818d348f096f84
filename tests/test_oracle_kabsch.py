"""CPU: pins oracle/kabsch.py against fixtures produced by the reference's own python (make_kabsch_golden.py)."""
import os

import numpy as np
import torch

from oracle import kabsch as OK


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "kabsch_reference.npz"))


def _rel(a, b):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-12)


def test_symm_ortho_forward_backward(golden_dir):
    g = _g(golden_dir)
    A = torch.from_numpy(g["so_A"]).requires_grad_(True)
    R = OK.symm_ortho(A)
    assert _rel(R, g["so_R"]) < 1e-10
    (R * torch.from_numpy(g["so_G"])).sum().backward()
    assert _rel(A.grad, g["so_gradA"]) < 1e-8


def test_weighted_pc_alignment(golden_dir):
    g = _g(golden_dir)
    w = torch.from_numpy(g["wpa_w"]).requires_grad_(True)
    T, nep = OK.weighted_pc_alignment(torch.from_numpy(g["wpa_p0"]), torch.from_numpy(g["wpa_p1"]), w)
    assert _rel(T, g["wpa_T"]) < 1e-6 and bool(nep) == bool(g["wpa_nep"])
    (T * torch.from_numpy(g["wpa_GT"])).sum().backward()
    assert _rel(w.grad, g["wpa_grad_w"]) < 1e-4
    few = torch.zeros(4000)
    few[:2] = 1.0
    T2, nep2 = OK.weighted_pc_alignment(torch.from_numpy(g["wpa_p0"]), torch.from_numpy(g["wpa_p1"]), few)
    assert bool(nep2) and _rel(T2, g["wpa_few_T"]) < 1e-5


def test_kabsch_decoder(golden_dir):
    g = _g(golden_dir)
    args = [torch.from_numpy(g[k]) for k in ("kd_pos", "kd_dims", "kd_rot", "kd_pts", "kd_valid", "kd_flow")]
    T, cum, w = OK.kabsch_trafos(*args)
    S = g["kd_pos"].shape[1]
    assert _rel(T[:, :S], g["kd_fgT"]) < 1e-5 and _rel(T[:, S:], g["kd_bgT"]) < 1e-5
    assert _rel(cum[:, :S], g["kd_fgc"]) < 1e-5 and _rel(cum[:, S:], g["kd_bgc"]) < 1e-5
    assert _rel(w.sum(-1), g["kd_fgw_sum"]) < 1e-5 and _rel(w[:, :, ::50], g["kd_fgw_sample"]) < 1e-5
    far = [torch.tensor([[[4000.0, 4000.0, 0.0]]]), torch.tensor([[[1.0, 1.0, 1.0]]]), torch.zeros(1, 1, 1)]
    T2, cum2, _ = OK.kabsch_trafos(*far, args[3][:1], args[4][:1], args[5][:1], softness="sigmoid")
    assert _rel(cum2[:, :1], g["kd_far_fgc"]) < 1e-5 and _rel(T2[:, :1], g["kd_far_fgT"]) < 1e-4
    assert _rel(T2[:, 1:], g["kd_far_bgT"]) < 1e-5
