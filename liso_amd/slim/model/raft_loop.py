"""All RAFT iterations of SLIM's update loop as ONE autograd node (training).

Reference: liso/slim/model/raft.py:178-259 (the loop: detach state, correlation lookup, update block, state += deltas) and
liso/slim/model/update.py:29-164 (motion encoder, ConvGRU, flow / classification heads).

Until round 5 the training step ran this loop op by op through autograd: ~50 nodes per iteration, and around the own convolution
kernels ~300 framework launches per step -- 86 gradient accumulations (every tensor with two consumers), 68 ReLU backward passes
(threshold_backward), ~60 concatenations (the reference's torch.cat calls and the stacking of six iterations' activations for the
deferred weight gradients), slice / pad / layout copies: 26.5 % of the step's kernel time (profiles/r05_aten_sources_slim.txt).
Here the loop is what the inference path already was (update.py: forward_inference) plus its adjoint, sequenced by hand:

* every activation of every iteration lives in a few STACKED channels-last buffers [iteration * batch, h, w, C]; the convolutions write
  their channel ranges (`conv_forward(out=...)`), so none of the reference's concatenations exists, and a layer's weight gradient is ONE
  launch over the stacked iterations with no stacking pass;
* [h | inp | out | class | flow | r*h] share one buffer per iteration: channels [0, 304) are the ConvGRU's `hx`, [96, 400) its
  `[x, r*h]` (convq on filters whose input channels are permuted accordingly); conv_class2 | conv_flow2 and the two heads' output
  convolutions are block-diagonal single launches, conv_class1 | conv_flow1 one 7x7 launch on the 8-float state pixel -- the added
  products are exact zeros, every filter sees its own channels in their own order;
* the state inputs (flow, logits) are detached per iteration in the reference (raft.py:189-193): an iteration's outputs receive gradient
  from its own deltas only, so the loss gradient of iteration i's low-resolution flow / logits IS the gradient of the heads' merged
  output of iteration i, and the motion encoder's 7x7 layers need no data gradient;
* backward: eight data gradients per iteration on the own kernels; between them `liso_rows_combine_f32` does in one launch what
  autograd did in three (sum of the consumers' gradients, ReLU mask, placement into the stacked buffer the weight gradient reads), and
  the ConvGRU gates have row-strided adjoints (include/liso_slim.h).

Numbers equal the op-by-op path's up to fp32 summation order (tests/test_gpu_slim.py::test_raft_loop_node_equals_op_by_op_autograd).
"""
import ctypes
import os

import torch

from liso_amd import _lib as L
from liso_amd.slim.model.raft_code import corr as C
from liso_amd.utils import mfma_conv as MC


def _dims(ub):
    me, gru, fh, hd = ub.motion_encoder, ub.gru, ub.static_flow_head, ub.classification_head
    return dict(ch=gru.convz.out_channels, co=me.conv.out_channels, cc=me.conv_class2.out_channels, cf=me.conv_flow2.out_channels,
                cq=me.conv_stat_corr1.out_channels, k1c=me.conv_class1.out_channels, k1f=me.conv_flow1.out_channels,
                hf=fh.conv1.out_channels, hc=hd.conv1.out_channels, P=me.conv_stat_corr1.in_channels)


def applicable(ub, net, inp, radius):
    """the configuration the node implements (the reference's default: class logits, no static-aggregation weights) on fp32 GPU tensors"""
    if os.environ.get("LISO_RAFT_LOOP", "1") == "0" or not net.is_cuda or net.dtype != torch.float32 or inp.dtype != torch.float32:
        return False
    if not ub.predict_logits or ub.cfg.model.predict_weight_for_static_aggregation or radius > 3:
        return False
    me, gru, fh, hd = ub.motion_encoder, ub.gru, ub.static_flow_head, ub.classification_head
    d = _dims(ub)
    ch, ci = net.shape[1], inp.shape[1]

    def geo(conv, k, pad, cin=None, cout=None):
        return (tuple(conv.kernel_size) == (k, k) and tuple(conv.padding) == (pad, pad) and tuple(conv.stride) == (1, 1)
                and tuple(conv.dilation) == (1, 1) and conv.groups == 1 and conv.bias is not None
                and (cin is None or conv.in_channels == cin) and (cout is None or conv.out_channels == cout))

    return (d["ch"] == ch and geo(me.conv_stat_corr1, 1, 0) and geo(me.conv_class1, 7, 3, 4) and geo(me.conv_flow1, 7, 3, 2)
            and geo(me.conv_class2, 3, 1, d["k1c"]) and geo(me.conv_flow2, 3, 1, d["k1f"]) and geo(me.conv, 3, 1, d["cq"] + d["cf"] + d["cc"])
            and geo(gru.convz, 3, 1, ch + ci + d["co"] + d["cc"] + d["cf"], ch) and geo(gru.convr, 3, 1, gru.convz.in_channels, ch)
            and geo(gru.convq, 3, 1, gru.convz.in_channels, ch) and geo(fh.conv1, 3, 1, ch) and geo(hd.conv1, 3, 1, ch)
            and geo(fh.conv2, 3, 1, d["hf"], 2) and geo(hd.conv2, 3, 1, d["hc"], 4)
            and all(v % 8 == 0 for v in (ch, ci, d["co"], d["cc"], d["cf"], d["cq"], d["k1c"], d["k1f"], d["hf"], d["hc"])) and d["P"] % 4 == 0)


def _params(ub):
    me, gru, fh, hd = ub.motion_encoder, ub.gru, ub.static_flow_head, ub.classification_head
    layers = (me.conv_stat_corr1, me.conv_class1, me.conv_flow1, me.conv_class2, me.conv_flow2, me.conv, gru.convz, gru.convr, gru.convq,
              fh.conv1, hd.conv1, fh.conv2, hd.conv2)
    return layers, [p for m in layers for p in (m.weight, m.bias)]


_S1, _S3, _S7 = MC.ConvSpec(1, 1, 1, 0), MC.ConvSpec(3, 3, 1, 1), MC.ConvSpec(7, 7, 1, 3)


def _merged_buffers(ub):
    """the loop's merged filters as persistent buffers on the update block (allocated and zeroed once per device / dtype: the
    off-diagonal blocks of the block-diagonal ones are never written again)"""
    me, gru = ub.motion_encoder, ub.gru
    d = _dims(ub)
    ch, co, cc, cf, cq, k1c, k1f, hf, hc = (d[k] for k in ("ch", "co", "cc", "cf", "cq", "k1c", "k1f", "hf", "hc"))
    dev = me.conv.weight.device
    hit = getattr(ub, "_raft_loop_buffers", None)
    if hit is not None and hit["dev"] == dev:
        return hit
    z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)  # noqa: E731
    cin = gru.convz.in_channels
    hit = {"dev": dev,
           "c71": (z(k1c + k1f, 8, 7, 7), z(k1c + k1f)), "cf": (z(cc + cf, k1c + k1f, 3, 3), z(cc + cf)),
           "conv": (z(co, cq + cc + cf, 3, 3), None), "zr": (z(2 * ch, cin, 3, 3), z(2 * ch)), "q": (z(ch, cin, 3, 3), None),
           "pair": (z(hf + hc, ch, 3, 3), z(hf + hc)), "hd": (z(8, hf + hc, 3, 3), z(8))}
    ub._raft_loop_buffers = hit
    return hit


def _merged_weights(ub, ci):
    """the loop's filters from the modules' parameters (detached): permuted / concatenated / block-diagonal, see the module docstring.
    ONE launch places every block (liso_multi_copy_rows) into the persistent buffers of `_merged_buffers`."""
    me, gru, fh, hd = ub.motion_encoder, ub.gru, ub.static_flow_head, ub.classification_head
    d = _dims(ub)
    ch, cc, cf, cq, k1c, hf = (d[k] for k in ("ch", "cc", "cf", "cq", "k1c", "hf"))
    B = _merged_buffers(ub)
    w71, b71 = B["c71"]
    wcf, bcf = B["cf"]
    wcv, _ = B["conv"]
    wzr, bzr = B["zr"]
    wq, _ = B["q"]
    wpr, bpr = B["pair"]
    whd, bhd = B["hd"]
    P = lambda m: (m.weight.detach(), m.bias.detach())  # noqa: E731
    (wc1, bc1), (wf1, bf1), (wc2, bc2), (wf2, bf2) = P(me.conv_class1), P(me.conv_flow1), P(me.conv_class2), P(me.conv_flow2)
    wc, _ = P(me.conv)
    (wz, bz), (wr, br), (wqq, _) = P(gru.convz), P(gru.convr), P(gru.convq)
    (wh1f, bh1f), (wh1c, bh1c), (wh2f, bh2f), (wh2c, bh2c) = P(fh.conv1), P(hd.conv1), P(fh.conv2), P(hd.conv2)
    cx = wqq.shape[1] - ch
    jobs = [
        (w71[:k1c, 0:4], wc1), (w71[k1c:, 4:6], wf1), (b71[:k1c], bc1), (b71[k1c:], bf1),
        (wcf[:cc, :k1c], wc2), (wcf[cc:, k1c:], wf2), (bcf[:cc], bc2), (bcf[cc:], bf2),
        # `conv` reads (corr, class, flow) instead of the reference's (corr, flow, class); `convq` reads (x, r*h) instead of (r*h, x)
        (wcv[:, :cq], wc[:, :cq]), (wcv[:, cq:cq + cc], wc[:, cq + cf:cq + cf + cc]), (wcv[:, cq + cc:], wc[:, cq:cq + cf]),
        (wzr[:ch], wz), (wzr[ch:], wr), (bzr[:ch], bz), (bzr[ch:], br),
        (wq[:, :cx], wqq[:, ch:]), (wq[:, cx:], wqq[:, :ch]),
        (wpr[:hf], wh1f), (wpr[hf:], wh1c), (bpr[:hf], bh1f), (bpr[hf:], bh1c),
        (whd[0:4, hf:], wh2c), (whd[4:6, :hf], wh2f), (bhd[0:4], bh2c), (bhd[4:6], bh2f),
    ]
    L.copy_blocks(jobs)
    return {
        "corr1": (me.conv_stat_corr1.weight.detach(), me.conv_stat_corr1.bias.detach(), _S1),
        "c71": (w71, b71, _S7), "cf": (wcf, bcf, _S3), "conv": (wcv, me.conv.bias.detach(), _S3), "zr": (wzr, bzr, _S3),
        "q": (wq, gru.convq.bias.detach(), _S3), "pair": (wpr, bpr, _S3), "hd": (whd, bhd, _S3),
    }


def _nchw(t):
    """logical [B, C, H, W] view of a channels-last buffer (slice) [B, H, W, C]"""
    return t.permute(0, 3, 1, 2)


def _combine(a, b=None, c=None, mask=None, out=None, accumulate=False):
    """out[..., :] = (a + b + c) * (mask > 0) over pixel rows; every operand a [N, H, W, C] tensor or channel slice of one"""
    n_pix, ch = a.shape[0] * a.shape[1] * a.shape[2], a.shape[3]
    if out is None:
        out = torch.empty(a.shape, dtype=torch.float32, device=a.device)

    def ps(t):
        if t is None:
            return None, 0
        assert t.shape == a.shape and t.stride(3) == 1 and t.dtype == torch.float32, (t.shape, a.shape, t.stride())
        s = t.stride(2)
        assert t.stride(1) == t.shape[2] * s and (t.shape[0] == 1 or t.stride(0) == t.shape[1] * t.shape[2] * s), (t.shape, t.stride())
        return L.ptr(t), s

    (pa, sa), (pb, sb), (pc, sc), (pm, sm), (po, so) = ps(a), ps(b), ps(c), ps(mask), ps(out)
    with torch.cuda.device(a.device):
        L.check(L.lib().liso_rows_combine_f32(n_pix, ch, pa, sa, pb, sb, pc, sc, pm, sm, po, so, int(bool(accumulate)), L.stream_ptr()),
                "rows_combine")
    return out


class _RaftLoop(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ub, correlation, n_it, direct, coords0, net0, inp, token, *params):
        d = _dims(ub)
        ch, co, cc, cf, cq, k1c, k1f, hf, hc, P = (d[k] for k in ("ch", "co", "cc", "cf", "cq", "k1c", "k1f", "hf", "hc", "P"))
        b, _, H, W = net0.shape
        ci = inp.shape[1]
        dev = net0.device
        o_inp, o_out, o_cls, o_rh = ch, ch + ci, ch + ci + co, ch + ci + co + cc + cf
        T = o_rh + ch
        N, npix, hw = n_it * b, b * H * W, H * W
        E = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        BIG, M, C1 = E((n_it + 1) * b, H, W, T), E(N, H, W, cq + cc + cf), E(N, H, W, k1c + k1f)
        F8, CORR = E((n_it + 1) * b, H, W, 8), E(N, H, W, P)
        ZR, Z, CQ, HID, BOTH = E(N, H, W, 2 * ch), E(N, H, W, ch), E(N, H, W, ch), E(N, H, W, hf + hc), E(N, H, W, 8)
        COORDS, FLOW, LOGITS = E(n_it + 1, b, 2, H, W), E(n_it, b, 2, H, W), E(n_it, b, 4, H, W)
        F8[:b].zero_()  # the loop starts from zero flow and zero logits (raft.py:150-154)
        COORDS[0].copy_(coords0)
        BIG[:b, ..., :ch].copy_(net0.permute(0, 2, 3, 1))
        BIG.view(n_it + 1, b, H, W, T)[:n_it, ..., o_inp:o_out].copy_(inp.permute(0, 2, 3, 1))  # (one broadcast copy: inp is constant)
        wts = _merged_weights(ub, ci)
        mode = MC._mode(torch.float32)
        pk = {k: MC.pack_weights(w, spec, False, mode) for k, (w, _, spec) in wts.items()}

        def conv(key, x, buf, off, relu):
            w, bias, spec = wts[key]
            MC.conv_forward(_nchw(x), w, bias, spec, out_relu=relu, packed=pk[key], out=(buf, off))

        lib = L.lib()
        f1, levels = correlation._fmap1_d, correlation._levels_d
        cfg = None
        for it in range(n_it):
            s, sn = slice(it * b, (it + 1) * b), slice((it + 1) * b, (it + 2) * b)
            big = BIG[s]
            _, cfg = C.lookup_forward(f1, levels, COORDS[it], correlation.radius, out=CORR[s])
            conv("corr1", CORR[s], M[s], 0, True)
            conv("c71", F8[s], C1[s], 0, True)
            conv("cf", C1[s], big, o_cls, True)
            _combine(big[..., o_cls:o_rh], out=M[s][..., cq:])  # (class, flow) features: the motion encoder's `conv` reads them too
            conv("conv", M[s], big, o_out, True)
            conv("zr", big[..., :o_rh], ZR[s], 0, False)
            with torch.cuda.device(dev):
                L.check(lib.liso_gru_in_rows_f32(npix, ch, L.ptr(ZR[s]), 2 * ch, L.ptr(big), T, L.ptr(Z[s]),
                                                 ctypes.c_void_p(big.data_ptr() + 4 * o_rh), T, L.stream_ptr()), "gru_in_rows")
            conv("q", big[..., ch:], CQ[s], 0, False)
            with torch.cuda.device(dev):
                L.check(lib.liso_gru_out_rows_train_f32(npix, ch, L.ptr(CQ[s]), ch, L.ptr(Z[s]), L.ptr(big), T, L.ptr(BIG[sn]), T,
                                                        L.stream_ptr()), "gru_out_rows_train")
            conv("pair", BIG[sn][..., :ch], HID[s], 0, True)
            conv("hd", HID[s], BOTH[s], 0, False)
            with torch.cuda.device(dev):
                L.check(lib.liso_raft_state_step_train_f32(b, hw, L.ptr(BOTH[s]), L.ptr(coords0), L.ptr(COORDS[it]), L.ptr(COORDS[it + 1]),
                                                           L.ptr(F8[s]), L.ptr(F8[sn]), L.ptr(FLOW[it]), L.ptr(LOGITS[it]), L.stream_ptr()),
                        "raft_state_step_train")
        ctx.ub, ctx.correlation, ctx.n_it, ctx.direct, ctx.cfg = ub, correlation, n_it, bool(direct), cfg
        ctx.level_hw = [l.shape[1] * l.shape[2] for l in levels]
        ctx.dims = (b, H, W, ci, T)
        ctx.buffers = (BIG, M, C1, F8, CORR, ZR, Z, CQ, HID, COORDS)
        ctx.wts = wts
        ctx.params = params if direct else None
        ctx.set_materialize_grads(False)
        return FLOW, LOGITS

    @staticmethod
    def backward(ctx, g_flow, g_logits):
        ub, correlation, n_it = ctx.ub, ctx.correlation, ctx.n_it
        d = _dims(ub)
        ch, co, cc, cf, cq, k1c, k1f, hf, hc, P = (d[k] for k in ("ch", "co", "cc", "cf", "cq", "k1c", "k1f", "hf", "hc", "P"))
        b, H, W, ci, T = ctx.dims
        BIG, M, C1, F8, CORR, ZR, Z, CQ, HID, COORDS = ctx.buffers
        ctx.buffers = None
        dev = BIG.device
        o_out, o_cls, o_rh = ch + ci, ch + ci + co, ch + ci + co + cc + cf
        cx = ci + co + cc + cf  # channels of the ConvGRU's x
        N, npix, hw = n_it * b, b * H * W, H * W
        E = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
        wts = ctx.wts
        mode = MC._mode(torch.float32)
        pk = {k: MC.pack_weights(w, spec, True, mode) for k, (w, _, spec) in wts.items() if k != "c71"}
        lib = L.lib()
        if g_flow is None:
            g_flow = torch.zeros((n_it, b, 2, H, W), dtype=torch.float32, device=dev)
        if g_logits is None:
            g_logits = torch.zeros((n_it, b, 4, H, W), dtype=torch.float32, device=dev)
        G8 = E(N, H, W, 8)
        with torch.cuda.device(dev):
            L.check(lib.liso_raft_pack_output_grads_f32(N, hw, L.ptr(g_flow.float().contiguous()), L.ptr(g_logits.float().contiguous()),
                                                        L.ptr(G8), L.stream_ptr()), "raft_pack_output_grads")
        D_HID, D_CQ, D_ZR, D_OUT = E(N, H, W, hf + hc), E(N, H, W, ch), E(N, H, W, 2 * ch), E(N, H, W, co)
        D_CF, D_C1, D_CORRF = E(N, H, W, cc + cf), E(N, H, W, k1c + k1f), E(N, H, W, cq)
        g_inp = E(b, H, W, ci)
        g_z, g_ha, g_hb = E(b, H, W, ch), E(b, H, W, ch), E(b, H, W, ch)

        def dgrad(key, dy, c_in):
            w, _, spec = wts[key]
            return MC.conv_dgrad(_nchw(dy), w, spec, (b, c_in, H, W), packed=pk[key]).permute(0, 2, 3, 1)  # -> [b, H, W, c_in] contiguous

        g_h = None  # gradient of the hidden state handed to the next iteration
        for it in reversed(range(n_it)):
            s, sn = slice(it * b, (it + 1) * b), slice((it + 1) * b, (it + 2) * b)
            big = BIG[s]
            # heads (update.py:107-127): both = conv_hd(relu(conv_pair(h_new)))
            _combine(dgrad("hd", G8[s], hf + hc), mask=HID[s], out=D_HID[s])
            g_hn = dgrad("pair", D_HID[s], ch)
            if g_h is not None:
                g_hn = _combine(g_hn, g_h)
            # ConvGRU (update.py:29-37)
            with torch.cuda.device(dev):
                L.check(lib.liso_gru_out_rows_bwd_f32(npix, ch, L.ptr(CQ[s]), ch, L.ptr(Z[s]), L.ptr(big), T, L.ptr(g_hn), ch, L.ptr(D_CQ[s]), ch,
                                                      L.ptr(g_z), L.ptr(g_ha), L.stream_ptr()), "gru_out_rows_bwd")
            d_xrh = dgrad("q", D_CQ[s], T - ch)  # [x | r*h]
            with torch.cuda.device(dev):
                L.check(lib.liso_gru_in_rows_bwd_f32(npix, ch, L.ptr(ZR[s]), 2 * ch, L.ptr(big), T, L.ptr(Z[s]), L.ptr(g_z),
                                                     ctypes.c_void_p(d_xrh.data_ptr() + 4 * cx), T - ch, L.ptr(D_ZR[s]), 2 * ch, L.ptr(g_hb),
                                                     L.stream_ptr()), "gru_in_rows_bwd")
            d_hx = dgrad("zr", D_ZR[s], o_rh)  # [h | x]
            g_h = _combine(g_ha, g_hb, d_hx[..., :ch])
            _combine(d_xrh[..., :ci], d_hx[..., ch:ch + ci], out=g_inp, accumulate=it != n_it - 1)
            # motion encoder (update.py:74-96): x = [inp | out | class | flow]; class / flow also feed `conv`
            _combine(d_xrh[..., ci:ci + co], d_hx[..., ch + ci:ch + ci + co], mask=big[..., o_out:o_cls], out=D_OUT[s])
            d_m = dgrad("conv", D_OUT[s], cq + cc + cf)  # [corr | class | flow]
            _combine(d_xrh[..., ci + co:cx], d_hx[..., ch + ci + co:ch + cx], d_m[..., cq:], mask=big[..., o_cls:o_rh], out=D_CF[s])
            _combine(dgrad("cf", D_CF[s], k1c + k1f), mask=C1[s], out=D_C1[s])  # (no data gradient behind the 7x7: the state is detached)
            _combine(d_m[..., :cq], mask=M[s][..., :cq], out=D_CORRF[s])
            d_corr = dgrad("corr1", D_CORRF[s], P)
            C.lookup_backward(correlation._state, ctx.cfg, ctx.level_hw, COORDS[it], d_corr)

        # ---- weight gradients: one launch per layer over the stacked iterations ------------------------------------------------------
        def wgrad(x, dy, shape, spec):
            res = MC.conv_wgrad(_nchw(x), _nchw(dy), shape, spec, want_bias=True)
            if res is None:
                raise NotImplementedError(f"liso_amd: no device weight-gradient kernel for {shape} on {tuple(x.shape)}")
            return res

        me, gru, fh, hd = ub.motion_encoder, ub.gru, ub.static_flow_head, ub.classification_head
        w_hd, b_hd = wgrad(HID, G8, (8, hf + hc, 3, 3), _S3)
        w_pair, b_pair = wgrad(BIG[b:(n_it + 1) * b][..., :ch], D_HID, (hf + hc, ch, 3, 3), _S3)
        w_q, b_q = wgrad(BIG[:N][..., ch:], D_CQ, (ch, T - ch, 3, 3), _S3)
        w_zr, b_zr = wgrad(BIG[:N][..., :o_rh], D_ZR, (2 * ch, o_rh, 3, 3), _S3)
        w_cv, b_cv = wgrad(M, D_OUT, (co, cq + cc + cf, 3, 3), _S3)
        w_cf, b_cf = wgrad(C1, D_CF, (cc + cf, k1c + k1f, 3, 3), _S3)
        w_c1, b_c1 = wgrad(F8[:N][..., 0:4], D_C1[..., :k1c], (k1c, 4, 7, 7), _S7)
        w_f1, b_f1 = wgrad(F8[:N][..., 4:8], D_C1[..., k1c:], (k1f, 4, 7, 7), _S7)
        w_co, b_co = wgrad(CORR, D_CORRF, (cq, P, 1, 1), _S1)
        # the merged / permuted gradients back to the modules' own layouts: row ranges are views, the column blocks are placed by ONE launch
        like = lambda p_: torch.empty(p_.shape, dtype=torch.float32, device=p_.device)  # noqa: E731
        g_f1, g_c2, g_f2, g_cv, g_q, g_fh2, g_hd2 = (like(m.weight) for m in (me.conv_flow1, me.conv_class2, me.conv_flow2, me.conv, gru.convq,
                                                                                fh.conv2, hd.conv2))
        L.copy_blocks([
            (g_f1, w_f1[:, :2]), (g_c2, w_cf[:cc, :k1c]), (g_f2, w_cf[cc:, k1c:]),
            (g_cv[:, :cq], w_cv[:, :cq]), (g_cv[:, cq:cq + cf], w_cv[:, cq + cc:]), (g_cv[:, cq + cf:], w_cv[:, cq:cq + cc]),
            (g_q[:, :ch], w_q[:, cx:]), (g_q[:, ch:], w_q[:, :cx]), (g_fh2, w_hd[4:6, :hf]), (g_hd2, w_hd[0:4, hf:]),
        ])
        grads = {
            me.conv_stat_corr1: (w_co, b_co), me.conv_class1: (w_c1, b_c1), me.conv_flow1: (g_f1, b_f1),
            me.conv_class2: (g_c2, b_cf[:cc]), me.conv_flow2: (g_f2, b_cf[cc:]), me.conv: (g_cv, b_cv),
            gru.convz: (w_zr[:ch], b_zr[:ch]), gru.convr: (w_zr[ch:], b_zr[ch:]), gru.convq: (g_q, b_q),
            fh.conv1: (w_pair[:hf], b_pair[:hf]), hd.conv1: (w_pair[hf:], b_pair[hf:]),
            fh.conv2: (g_fh2, b_hd[4:6]), hd.conv2: (g_hd2, b_hd[0:4]),
        }
        layers, _ = _params(ub)
        flat = [g for m in layers for g in grads[m]]
        if ctx.direct:
            # hipGraph capture: add into the (pre-existing) .grad buffers here, on the capturing stream -- autograd's AccumulateGrad nodes
            # would run on the stream they were created on (deferred_wgrad.py)
            with torch.no_grad():
                for p, g in zip(ctx.params, flat):
                    if p.grad is None:
                        p.grad = g.contiguous()
                    else:
                        p.grad.add_(g)
            flat = [None] * len(flat)
        g_net0 = _nchw(g_h) if ctx.needs_input_grad[5] else None
        g_inp_out = _nchw(g_inp) if ctx.needs_input_grad[6] else None
        g_tok = G8.new_zeros(1) if ctx.needs_input_grad[7] else None
        return (None, None, None, None, None, g_net0, g_inp_out, g_tok, *flat)


def raft_loop(ub, correlation, coords0, net, inp, n_it, direct=False):
    """-> (flows [n_it, b, 2, h, w] = coords1 - coords0 after every iteration, logits [n_it, b, 4, h, w]): what
    raft_outputs._RaftOutputs assembles into the network outputs of all iterations"""
    _, params = _params(ub)
    return _RaftLoop.apply(ub, correlation, int(n_it), bool(direct), coords0.detach().float().contiguous(), net, inp, correlation._token,
                           *params)
