"""Host-side logic of the SLIM training path that does not need a GPU: the deferred weight-gradient mechanism of the RAFT
update block (liso_amd/slim/model/deferred_wgrad.py) against plain autograd, in fp64 on the CPU (there the convolutions and
the ConvGRU gates run the reference's op sequence)."""
import torch

from liso_amd.slim.model.deferred_wgrad import conv2d_pair, deferred_weight_gradients
from liso_amd.slim.model.update import ConvGRU, SmallUpdateBlock
from liso_amd.utils.config import default_cfg


def _run(ub, defer, n_it=4, B=2, h=8, w=8):
    g = torch.Generator().manual_seed(0)
    net0 = torch.randn(B, 96, h, w, dtype=torch.double, generator=g)
    inp = torch.randn(B, 64, h, w, dtype=torch.double, generator=g)
    corrs = [torch.randn(B, 196, h, w, dtype=torch.double, generator=g) for _ in range(n_it)]
    for p in ub.parameters():
        p.grad = None
    net = net0.clone().requires_grad_(True)
    flow, logits = torch.zeros(B, 2, h, w, dtype=torch.double), torch.zeros(B, 4, h, w, dtype=torch.double)
    loss, n = 0.0, net
    with deferred_weight_gradients(ub, enabled=defer) as st:
        assert (st is not None) == defer
        for it in range(n_it):
            n, df, dl, _ = ub(n, inp, corrs[it], flow.detach(), logits.detach(), None)
            flow, logits = flow.detach() + df, logits.detach() + dl
            loss = loss + flow.square().mean() + 0.5 * logits.square().mean()
    loss.backward()
    return float(loss.detach()), [p.grad.clone() for p in ub.parameters()], net.grad.clone()


def test_deferred_weight_gradients_are_exact_in_fp64():
    torch.manual_seed(0)
    ub = SmallUpdateBlock(default_cfg(grid=128).SLIM).double()
    l0, g0, n0 = _run(ub, False)
    l1, g1, n1 = _run(ub, True)
    assert l0 == l1 and torch.equal(n0, n1)
    assert max(float((a - b).abs().max()) for a, b in zip(g0, g1)) < 1e-14
    assert all(float(a.abs().max()) > 0 for a in g0)
    # a second backward pass through a fresh context works (the state is per forward)
    l2, g2, _ = _run(ub, True)
    assert l2 == l1 and all(torch.equal(a, b) for a, b in zip(g1, g2))


def test_deferral_is_off_without_grad_and_keeps_state_dict_keys():
    ub = SmallUpdateBlock(default_cfg(grid=128).SLIM)
    keys = set(ub.state_dict())
    assert "gru.convz.weight" in keys and "gru.convr.bias" in keys and not any("merged" in k for k in keys)
    with torch.no_grad():
        with deferred_weight_gradients(ub) as st:
            assert st is None
    for p in ub.gru.convz.parameters():
        p.requires_grad_(False)
    with deferred_weight_gradients(ub) as st:  # a frozen layer: plain autograd for everything
        assert st is None


def test_merged_convolution_pair_equals_the_two_convolutions():
    torch.manual_seed(2)
    gru = ConvGRU(hidden_dim=8, input_dim=8 + 5).double()
    x = torch.randn(2, 13, 6, 7, dtype=torch.double)
    both = conv2d_pair(gru.convz, gru.convr, x)
    assert torch.allclose(both, torch.cat([gru.convz(x), gru.convr(x)], dim=1), atol=1e-13)
    with deferred_weight_gradients(gru):
        y = conv2d_pair(gru.convz, gru.convr, x)
    wgt = torch.randn_like(y)
    (y * wgt).sum().backward()
    gz, gr = gru.convz.weight.grad.clone(), gru.convr.weight.grad.clone()
    for p in gru.parameters():
        p.grad = None
    (torch.cat([gru.convz(x), gru.convr(x)], dim=1) * wgt).sum().backward()
    assert torch.allclose(gz, gru.convz.weight.grad, atol=1e-12) and torch.allclose(gr, gru.convr.weight.grad, atol=1e-12)
