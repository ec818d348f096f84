"""Weight gradients of the RAFT update block, computed once per step instead of once per iteration.

The update block (liso/slim/model/update.py:98-164) runs its 13 convolutions `num_iters` (6) times with the same weights.
Autograd then launches 6 weight-gradient convolutions, 6 bias reductions and 5 gradient accumulations per parameter
(78 + 78 + 130 launches per direction batch; the 64x64 maps leave most of the 256 CUs idle in each).  The weight gradient is
a sum over samples, so here every convolution's backward only computes the data gradient and stashes (input, grad_output);
when the backward of the first iteration is through, ONE weight-gradient convolution per layer runs over the stashed pairs
concatenated along the batch axis.  Same arithmetic, other summation order (fp32 sums over 6x the samples in one kernel).

Ordering: a 1-element token is threaded through the convolutions in forward order (each consumes the previous token and
emits the next), so autograd runs their backwards in exactly the reverse order and the gate that owns the parameters last.
"""
import contextlib

import torch
from torch import nn

from liso_amd.utils import host_ops

_ACTIVE = None


class _State:
    """`ops`: tuples of 1 or 2 Conv2d.  A pair shares its input and geometry (ConvGRU's convz / convr): it runs as ONE
    convolution with the weights concatenated along the output channels."""

    def __init__(self, ops):
        self.ops = ops
        self.layers = [op[0] for op in ops]  # geometry (stride / padding / dilation) of every op
        self.index = {tuple(id(m) for m in op): i for i, op in enumerate(ops)}
        self.stash = [[] for _ in ops]
        self.merged = {}  # op index -> (weight, bias) concatenated once per step
        self.packs = {}
        self.token = None
        self.direct_accumulate = False
        self.params = []

    def packed(self, li, for_dgrad, dtype):
        """the kernels' packed panels of op `li` (weights are constant while the state lives: one pack per step and use)"""
        from liso_amd.utils import mfma_conv as MC

        key = (li, bool(for_dgrad), dtype)
        if key not in self.packs:
            op = self.ops[li]
            # (a single layer: hand over the Parameter itself, so that a trainer's batched per-step pack -- mfma_conv.set_step_packs --
            # can serve it; merged pairs are temporaries and pack on their own)
            w = op[0].weight if len(op) == 1 else self.weights(li)[0]
            self.packs[key] = MC.pack_weights(w, MC.ConvSpec.of(self.layers[li]), for_dgrad, MC._mode(dtype))
        return self.packs[key]

    def weights(self, li):
        op = self.ops[li]
        if len(op) == 1:
            return op[0].weight.detach(), op[0].bias.detach()
        if li not in self.merged:
            self.merged[li] = (torch.cat([m.weight.detach() for m in op], dim=0), torch.cat([m.bias.detach() for m in op], dim=0))
        return self.merged[li]


def _pair(v):
    return [int(v[0]), int(v[1])] if isinstance(v, (tuple, list)) else [int(v), int(v)]


class _ParamGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, state, *params):
        ctx.state = state
        ctx.save_for_backward(*params)
        return params[0].new_zeros(1)

    @staticmethod
    def backward(ctx, _token_grad):
        params = ctx.saved_tensors
        st = ctx.state
        grads = []
        pi = 0
        for li, op in enumerate(st.ops):
            mine = params[pi:pi + 2 * len(op)]  # (w, b) of every layer of the op
            pi += 2 * len(op)
            items, st.stash[li] = st.stash[li], []
            if not items:
                grads += [torch.zeros_like(p) for p in mine]
                continue
            layer = op[0]
            w, b = st.weights(li)
            x = items[0][0] if len(items) == 1 else torch.cat([i[0] for i in items], dim=0)
            gy = items[0][1] if len(items) == 1 else torch.cat([i[1] for i in items], dim=0)
            res = None
            if _own_kernels(x, w, layer):  # ONE weight-gradient launch over the stacked iterations (split over pixels)
                from liso_amd.utils import mfma_conv as MC

                res = MC.conv_wgrad(x, gy, tuple(w.shape), MC.ConvSpec.of(layer), want_bias=True)
            if res is not None:
                gw, gb = res
            else:
                if x.is_cuda:
                    raise NotImplementedError(f"liso_amd: no device weight-gradient kernel for {tuple(w.shape)} on {tuple(x.shape)} {x.dtype}")
                _, gw, gb = torch.ops.aten.convolution_backward(gy, x, w, [b.numel()], _pair(layer.stride), _pair(layer.padding),
                                                                _pair(layer.dilation), False, [0, 0], 1, [False, True, True])
            if len(op) == 1:
                grads += [gw, gb]
            else:
                sizes = [m.weight.shape[0] for m in op]
                for gwi, gbi in zip(torch.split(gw, sizes, dim=0), torch.split(gb, sizes, dim=0)):
                    grads += [gwi, gbi]
        st.merged.clear()
        st.packs.clear()
        if st.direct_accumulate:
            # hipGraph capture: add into the (pre-existing) .grad buffers here, on the capturing stream.  Handing the gradients
            # to autograd's AccumulateGrad nodes would run them on the stream those nodes were created on (the warm-up's), a
            # cross-stream hop that a capture cannot contain: wrong values / a crash, depending on the sizes (measured).
            with torch.no_grad():
                for src, g in zip(st.params, grads):
                    if src.grad is None:
                        src.grad = g  # (a fresh tensor of this launch, or a slice of one: the trainer gathers it into its flat buffer)
                    else:
                        src.grad.add_(g)
            return (None,) * (1 + len(params))
        return (None, *grads)


def _own_kernels(x, w, layer):
    from liso_amd.utils import mfma_conv as MC

    # (False on the device only for geometries that `conv2d` below hands to mfma_conv.conv2d -- the 2-channel conv_flow1, zero-padded to
    # one 16-byte channel group there; host tensors: the CPU test tier, host_ops)
    return MC.on_device(x) and MC.supported(x, w, MC.ConvSpec.of(layer))


class _ConvDeferred(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, token, w, b, state, li, relu):
        layer = state.layers[li]
        ctx.state, ctx.li, ctx.relu = state, li, bool(relu)
        ctx.own = _own_kernels(x, w, layer)
        ctx.set_materialize_grads(False)
        if ctx.own:
            from liso_amd.utils import mfma_conv as MC

            y, _ = MC.conv_forward(x, w, b, MC.ConvSpec.of(layer), out_relu=ctx.relu, packed=state.packed(li, False, x.dtype))
        else:
            y = host_ops.conv2d(x, w, b, layer.stride, layer.padding, layer.dilation)
            y = torch.relu(y) if ctx.relu else y
        ctx.save_for_backward(x, w, y if ctx.relu else None)
        return y, token.view_as(token)  # alias: no launch

    @staticmethod
    def backward(ctx, gy, token_grad):
        x, w, y = ctx.saved_tensors
        st, li = ctx.state, ctx.li
        layer = st.layers[li]
        gx = None
        if gy is not None:
            if ctx.relu:
                gy = torch.ops.aten.threshold_backward(gy, y, 0.0)  # gy where y > 0 else 0, one launch
            if ctx.needs_input_grad[0]:
                if ctx.own:
                    from liso_amd.utils import mfma_conv as MC

                    gx = MC.conv_dgrad(gy, w, MC.ConvSpec.of(layer), tuple(x.shape), packed=st.packed(li, True, gy.dtype))
                else:
                    gx = torch.ops.aten.convolution_backward(gy, x, w, None, _pair(layer.stride), _pair(layer.padding),
                                                             _pair(layer.dilation), False, [0, 0], 1, [True, False, False])[0]
            st.stash[li].append((x, gy))
        if token_grad is None:  # the last convolution of the chain: nothing consumed its token
            token_grad = st.zero
        return gx, token_grad, None, None, None, None, None


@contextlib.contextmanager
def deferred_weight_gradients(module: nn.Module, enabled=True, direct_accumulate=False):
    """Inside the context, `conv2d(layer, x)` of any Conv2d of `module` (groups == 1, with bias) defers its weight gradient.
    `direct_accumulate`: the gate adds the weight gradients into `param.grad` itself instead of returning them to autograd
    (for hipGraph capture; bypasses gradient hooks, so not for DistributedDataParallel)."""
    global _ACTIVE
    layers = [m for m in module.modules() if isinstance(m, nn.Conv2d)]
    ok = (enabled and torch.is_grad_enabled() and layers and all(m.groups == 1 and m.bias is not None and m.weight.requires_grad
                                                                  and m.bias.requires_grad and m.padding_mode == "zeros"
                                                                  for m in layers))
    if not ok or _ACTIVE is not None:
        yield None
        return
    pairs = [tuple(p) for m in module.modules() for p in getattr(m, "merged_convs", ())]
    paired = {id(l) for p in pairs for l in p}
    ops = pairs + [(m,) for m in layers if id(m) not in paired]
    st = _State(ops)
    st.direct_accumulate = bool(direct_accumulate)
    st.params = [p for op in ops for m in op for p in (m.weight, m.bias)]
    st.token = _ParamGate.apply(st, *st.params)
    st.zero = st.token.detach()
    _ACTIVE = st
    try:
        yield st
    finally:
        _ACTIVE = None


def conv2d(layer: nn.Conv2d, x, relu=False):
    """relu?(layer(x)); inside `deferred_weight_gradients` the weight gradient is deferred to the end of the backward pass"""
    st = _ACTIVE
    if st is None or (id(layer),) not in st.index or not _own_kernels(x, layer.weight, layer):
        # (not deferred: layers outside the context, and the 2-channel conv_flow1, which mfma_conv.conv2d runs on the own kernels
        # with zero-padded channels -- its tiny weight gradient is computed per iteration)
        from liso_amd.utils import mfma_conv as MC

        return MC.conv2d(layer, x, relu)
    li = st.index[(id(layer),)]
    y, st.token = _ConvDeferred.apply(x, st.token, *st.weights(li), st, li, relu)
    return y


def conv2d_pair(layer_a: nn.Conv2d, layer_b: nn.Conv2d, x, relu=False):
    """relu?(cat([layer_a(x), layer_b(x)], dim=1)) as one convolution (same input, same geometry)"""
    st = _ACTIVE
    key = (id(layer_a), id(layer_b))
    own = _own_kernels(x, layer_a.weight, layer_a)
    if st is None or key not in st.index or not own:
        from liso_amd.utils import mfma_conv as MC

        if own:
            return MC.fused_conv(x, None, [layer_a, layer_b], out_relu=relu)[0]
        if x.is_cuda:  # channel counts outside the kernels' 16-byte groups (none of the networks'): mfma_conv.conv2d pads them, per layer
            return torch.cat([MC.conv2d(layer_a, x, relu), MC.conv2d(layer_b, x, relu)], dim=1)
        w = torch.cat([layer_a.weight, layer_b.weight], dim=0)
        b = torch.cat([layer_a.bias, layer_b.bias], dim=0)
        y = host_ops.conv2d(x, w, b, layer_a.stride, layer_a.padding, layer_a.dilation)
        return torch.relu(y) if relu else y
    li = st.index[key]
    y, st.token = _ConvDeferred.apply(x, st.token, *st.weights(li), st, li, relu)
    return y
