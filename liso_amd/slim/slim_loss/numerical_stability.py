"""mirror of liso/slim/slim_loss/numerical_stability.py:7-53"""
import torch


def numerically_stable_lin_comb_exps(*, exps, weights, mask=None, dim: int = -1, keepdims: bool = False):
    if mask is not None:
        min_exp = torch.min(exps, dim=dim, keepdims=True)[0]
        exps = torch.where(mask, exps, min_exp)
        weights = weights * mask
    max_exp = torch.max(exps, dim=dim, keepdims=True)[0]
    max_kd = max_exp if keepdims else torch.squeeze(max_exp, dim=dim)
    return max_kd, (torch.exp(exps - max_exp) * weights).sum(dim=dim, keepdims=keepdims)


def normalized_sigmoid_sum(logits, mask=None):
    """sigmoid(x) = exp(-relu(-x)) * sigmoid(|x|), normalised to sum 1 over the masked entries"""
    neg = -torch.relu(-logits)
    weights = torch.sigmoid(torch.abs(logits))
    denom_exp, denom = numerically_stable_lin_comb_exps(exps=neg, weights=weights, mask=mask, keepdims=True)
    if mask is not None:
        weights = weights * mask
        all_masked = ~mask.any(dim=-1, keepdims=True)
        denom = torch.where(all_masked, torch.ones_like(denom), denom)
        neg = torch.where(mask, neg, denom_exp)
    return torch.exp(neg - denom_exp) * weights / denom
