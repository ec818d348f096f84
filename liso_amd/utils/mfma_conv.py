"""Convolutions of the BEV networks on the gfx950 matrix cores (C ABI: include/liso_conv.h).

Host side of the own implicit-GEMM kernels that replace torch.nn.functional.conv2d / conv_transpose2d (cuDNN in the
reference, MIOpen on ROCm) for
    liso/networks/centerpoint/rpn.py:113-146, center_head.py:60-117   (bf16 tensors, fp32 accumulation)
    liso/slim/model/update.py:96-164, extractor.py:211-297            (fp32 tensors as bf16 hi/lo pairs, "F32X3")
Tensors are logical NCHW with channels-last strides (physical [B, H, W, C]).  `fused_conv` is the autograd entry point:
    y_raw = conv( relu?(bn(x_raw)) ) + bias          -- the BatchNorm-apply + ReLU of the PRODUCING layer runs in the
                                                        convolution's prologue; its backward (liso_bn_relu_bwd) follows
                                                        the data gradient inside this function's backward
and optionally emits the per-block partial sums from which `finalize_bn` derives the BatchNorm statistics of y_raw
(no separate statistics pass over the activation).
"""
import ctypes
import os
import weakref

import torch

from liso_amd import _lib as L


def on_device(x):
    """True: `x` is a GPU tensor and goes through the own kernels -- and only through them: a device tensor in a dtype they do not
    take raises here instead of reaching a library.  False: a HOST tensor (the CPU test tier steps the modules through
    liso_amd/utils/host_ops.py)."""
    if not x.is_cuda:
        return False
    if x.dtype not in (torch.bfloat16, torch.float32):
        raise TypeError(f"liso_amd: device convolutions take bfloat16 / float32 tensors, got {x.dtype}")
    return True


class ConvSpec:
    """geometry of one nn.Conv2d / nn.ConvTranspose2d (square stride / padding, dilation 1, groups 1)"""

    def __init__(self, kh, kw, stride=1, padding=0, transposed=False):
        self.kh, self.kw, self.stride, self.padding, self.transposed = int(kh), int(kw), int(stride), int(padding), bool(transposed)

    @staticmethod
    def of(conv):
        st = conv.stride[0] if isinstance(conv.stride, (tuple, list)) else conv.stride
        pd = conv.padding[0] if isinstance(conv.padding, (tuple, list)) else conv.padding
        assert tuple(conv.dilation) == (1, 1) and conv.groups == 1
        return ConvSpec(conv.kernel_size[0], conv.kernel_size[1], st, pd, isinstance(conv, torch.nn.ConvTranspose2d))

    def out_hw(self, h, w):
        if self.transposed:
            return (h - 1) * self.stride - 2 * self.padding + self.kh, (w - 1) * self.stride - 2 * self.padding + self.kw
        return (h + 2 * self.padding - self.kh) // self.stride + 1, (w + 2 * self.padding - self.kw) // self.stride + 1


# Arithmetic of fp32 tensors: "x3" = three bf16 MFMAs per product (2^-16 per product, the production default of the SLIM networks)
# or "exact" = native fp32 MFMA (v_mfma_f32_32x32x2_f32: 2^-24 per product, the reference's fp32 semantics; the parity mode).
# Process-wide; LISO_CONV_FP32=exact sets the initial value.  bf16 tensors are not affected.
_FP32_MODE = "exact" if os.environ.get("LISO_CONV_FP32", "x3") == "exact" else "x3"


def set_fp32_mode(mode):
    """"x3" | "exact" -> the previous mode"""
    global _FP32_MODE
    assert mode in ("x3", "exact"), mode
    prev, _FP32_MODE = _FP32_MODE, mode
    return prev


def fp32_mode():
    return _FP32_MODE


class fp32_arithmetic:
    """with fp32_arithmetic("exact"): ... -- forward AND backward of the graph built inside should run inside the block"""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = set_fp32_mode(self.mode)
        return self

    def __exit__(self, *exc):
        set_fp32_mode(self.prev)
        return False


def _mode(dtype):
    if dtype == torch.bfloat16:
        return L.CONV_BF16
    if dtype == torch.float32:
        return L.CONV_F32 if _FP32_MODE == "exact" else L.CONV_F32X3
    raise L.LisoHipError(f"mfma conv: unsupported dtype {dtype}")


def _base_desc(B, hi, wi, ci, x_ps, ho, wo, co, y_ps, y_off, mode, out_f32, in_relu, out_relu, w_taps):
    d = L.ConvDesc()
    d.batch, d.hi, d.wi, d.ci, d.x_pix_stride = B, hi, wi, ci, x_ps
    d.ho, d.wo, d.co, d.y_pix_stride, d.y_ch_off = ho, wo, co, y_ps, y_off
    d.mode, d.out_f32, d.in_relu, d.out_relu, d.w_taps = mode, int(out_f32), int(in_relu), int(out_relu), w_taps
    return d


_DESC_CACHE = {}
_COPY_LOG = None  # debugging: set to {} to count the layout copies `as_nhwc` has to make, by shape and caller


def _cached(fn):
    def wrapper(spec, *args, **kw):
        key = (fn.__name__, spec.kh, spec.kw, spec.stride, spec.padding, spec.transposed, args, tuple(sorted(kw.items())))
        d = _DESC_CACHE.get(key)
        if d is None:
            d = _DESC_CACHE[key] = fn(spec, *args, **kw)
        return d
    return wrapper


@_cached
def gather_desc(spec, B, hi, wi, ci, x_ps, ho, wo, co, y_ps, y_off, mode, out_f32=False, in_relu=False, out_relu=False):
    """out[v] = sum_taps in[v * s + (k - p)] * w[k]: forward of a convolution, data gradient of a transposed convolution"""
    d = _base_desc(B, hi, wi, ci, x_ps, ho, wo, co, y_ps, y_off, mode, out_f32, in_relu, out_relu, spec.kh * spec.kw)
    d.hv, d.wv, d.isy, d.isx, d.osy, d.osx = ho, wo, spec.stride, spec.stride, 1, 1
    d.n_classes = 1
    d.class_tap_begin[0], d.class_tap_begin[1] = 0, spec.kh * spec.kw
    d.n_taps = spec.kh * spec.kw
    assert d.n_taps <= L.CONV_MAX_TAPS
    for ky in range(spec.kh):
        for kx in range(spec.kw):
            t = ky * spec.kw + kx
            d.tap_dy[t], d.tap_dx[t], d.tap_w[t] = ky - spec.padding, kx - spec.padding, t
    return d


@_cached
def scatter_desc(spec, B, hi, wi, ci, x_ps, ho, wo, co, y_ps, y_off, mode, out_f32=False, in_relu=False, out_relu=False):
    """out[i * s - p + k] += in[i] * w[k], evaluated per output-parity class: forward of a transposed convolution, data
    gradient of a convolution.  (hi, wi, ci) describe the tensor that is READ, (ho, wo, co) the one that is written."""
    s, p = spec.stride, spec.padding
    d = _base_desc(B, hi, wi, ci, x_ps, ho, wo, co, y_ps, y_off, mode, out_f32, in_relu, out_relu, spec.kh * spec.kw)
    d.hv, d.wv = (ho + s - 1) // s, (wo + s - 1) // s
    d.isy, d.isx, d.osy, d.osx = 1, 1, s, s
    n, cls = 0, 0
    d.class_tap_begin[0] = 0
    for py in range(s):
        for px in range(s):
            taps = [(ky, kx) for ky in range(spec.kh) for kx in range(spec.kw) if (py + p - ky) % s == 0 and (px + p - kx) % s == 0]
            if not taps:  # kernel smaller than the stride: no input reaches this output parity (the caller zero-fills it)
                continue
            for ky, kx in taps:
                assert n < L.CONV_MAX_TAPS
                d.tap_dy[n], d.tap_dx[n], d.tap_w[n] = (py + p - ky) // s, (px + p - kx) // s, ky * spec.kw + kx
                n += 1
            d.class_ooy[cls], d.class_oox[cls] = py, px
            cls += 1
            d.class_tap_begin[cls] = n
    assert cls <= L.CONV_MAX_CLASSES
    d.n_classes, d.n_taps = cls, n
    d.sparse_output = cls < s * s  # (python-side attribute) some output positions are never written
    return d


def _pix_stride(v):
    B, H, W, C = v.shape
    if W > 1:
        return v.stride(2)
    if H > 1:
        return v.stride(1)
    if B > 1:
        return v.stride(0)
    return C


def as_nhwc(t, vec):
    """physical NHWC view of a logical-NCHW tensor + its pixel stride (a copy is made unless the layout already is
    channels-last, possibly as a channel slice of a wider tensor, with a pixel stride that keeps 16-B alignment)"""
    v = t.permute(0, 2, 3, 1)
    B, H, W, C = v.shape
    ps = _pix_stride(v)
    regular = (v.stride(3) == 1 or C == 1) and ps >= C and (W == 1 or v.stride(2) == ps) and (H == 1 or v.stride(1) == W * ps) and \
        (B == 1 or v.stride(0) == H * W * ps) and ps % vec == 0 and v.data_ptr() % 16 == 0
    if not regular:
        if _COPY_LOG is not None:
            import traceback
            fr = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack(limit=7)[:-1] if "mfma_conv" not in f.filename][-3:]
            _COPY_LOG[(tuple(t.shape), tuple(t.stride()), tuple(fr))] = _COPY_LOG.get((tuple(t.shape), tuple(t.stride()), tuple(fr)), 0) + 1
        v = t.contiguous(memory_format=torch.channels_last).permute(0, 2, 3, 1)
        if v.stride(3) != 1 and C > 1:  # (channels_last of a C == 1 tensor can report odd strides)
            v = v.contiguous()
        ps = C
    return v, ps


_PACK_CACHE = {}


_SHARED_DEPTH = [0]


class shared_gpu:
    """`with shared_gpu():` -- the launches planned inside (incl. those captured into hipGraphs) share the GPU with other streams'
    kernels: convolution plans never take a CU's whole LDS for one block (include/liso_conv.h: LISO_CONV_OPT_SHARED_GPU).  The LISO
    loop's pipeline runs its steps inside; stand-alone trainers keep the single-stream plans.  Results do not depend on it beyond
    the summation order inside a convolution."""

    def __enter__(self):
        _SHARED_DEPTH[0] += 1
        if _SHARED_DEPTH[0] == 1:
            L.check(L.lib().liso_conv_set_option(L.CONV_OPT_SHARED_GPU, 1), "conv_set_option")
        return self

    def __exit__(self, *exc):
        _SHARED_DEPTH[0] -= 1
        if _SHARED_DEPTH[0] == 0:
            L.check(L.lib().liso_conv_set_option(L.CONV_OPT_SHARED_GPU, 0), "conv_set_option")
        return False


_WGRAD_SIDE = None  # {"stream", "keep", "used"} while weight gradients are issued on a side stream (wgrad_side)


class wgrad_side:
    """`with wgrad_side(stream):` around a backward pass -- the weight-gradient launches of the fused convolutions (those that write
    straight into their parameter's gradient buffer: nothing for autograd to consume) go to `stream`, forked off the backward pass's
    stream by an event behind the gradient they read and joined when the block exits.  Nothing in the backward chain reads a weight
    gradient: the chain (data gradient -> BatchNorm backward -> data gradient ...) no longer waits for 19 weight-gradient + slab-reduction
    launches, which run next to it -- inside a captured hipGraph as a parallel branch.  The tensors those launches read are kept alive
    until the join (no allocator reuse under them)."""

    def __init__(self, stream):
        self.stream = stream

    def __enter__(self):
        global _WGRAD_SIDE
        self.prev = _WGRAD_SIDE
        _WGRAD_SIDE = {"stream": self.stream, "keep": [], "used": False} if self.stream is not None else None
        return self

    def __exit__(self, *exc):
        global _WGRAD_SIDE
        st, _WGRAD_SIDE = _WGRAD_SIDE, self.prev
        if st is not None and st["used"]:
            torch.cuda.current_stream(st["stream"].device).wait_stream(st["stream"])
            st["keep"].clear()
        return False


class roles_cus:
    """`with roles_cus(n):` -- the persistent blocks of the 3x3 / stride-1 convolution launches planned inside (incl. those captured into
    hipGraphs) occupy at most n compute units (include/liso_conv.h: LISO_CONV_OPT_ROLES_CUS; n = 0: no limit).  Not re-entrant."""

    def __init__(self, n):
        self.n = int(n)

    def __enter__(self):
        if self.n:
            L.check(L.lib().liso_conv_set_option(L.CONV_OPT_ROLES_CUS, self.n), "conv_set_option")
        return self

    def __exit__(self, *exc):
        if self.n:
            L.check(L.lib().liso_conv_set_option(L.CONV_OPT_ROLES_CUS, 0), "conv_set_option")
        return False


_PACK_RECORD = None  # dict while the pack requests of a step are being recorded
_STEP_PACKS = None   # dict while a step runs on panels that were packed by ONE batched launch


def record_pack_jobs(on=True):
    """start (-> None) / stop (-> list of jobs) recording which (Parameter, geometry, direction, arithmetic) panels a step asks
    for.  A trainer records one warm-up step, then opens every captured step with `set_step_packs(batched_pack(jobs))`: one
    launch instead of one per layer and direction (57 for the CenterPoint backbone + head)."""
    global _PACK_RECORD
    if on:
        _PACK_RECORD = {}
        return None
    jobs, _PACK_RECORD = list((_PACK_RECORD or {}).values()), None
    return jobs


def batched_pack(jobs):
    """-> {key: packed panels} for `set_step_packs`; one liso_conv_pack_weights_batched call"""
    out, arr = {}, (L.ConvPackJob * len(jobs))()
    keep = []
    for i, (key, weight, spec, for_dgrad, mode) in enumerate(jobs):
        w = weight.detach()
        if w.dtype != torch.float32 or not w.is_contiguous():
            w = w.float().contiguous()
        keep.append(w)
        d0, d1 = w.shape[0], w.shape[1]
        same = spec.transposed == bool(for_dgrad)
        K, N = (d1, d0) if same else (d0, d1)
        dst = torch.empty(L.lib().liso_conv_packed_bytes(K, N, spec.kh * spec.kw, mode), dtype=torch.uint8, device=w.device)
        arr[i] = L.ConvPackJob(w.data_ptr(), dst.data_ptr(), d0, d1, spec.kh, spec.kw, int(spec.transposed), int(bool(for_dgrad)), mode)
        out[key] = dst
    if jobs:
        with torch.cuda.device(jobs[0][1].device):
            L.check(L.lib().liso_conv_pack_weights_batched(arr, len(jobs), L.stream_ptr()), "conv_pack_weights_batched")
    return out


def set_step_packs(packs):
    global _STEP_PACKS
    _STEP_PACKS = packs


def pack_weights(weight, spec, for_dgrad, mode):
    """torch-layout fp32 master weights -> the kernels' packed bf16 panels (one launch).  The result is cached per weight
    tensor until the tensor is modified in place (`_version`: optimizer steps bump it), so frozen networks pack once and
    the forward / backward of one training step share the panels of a layer."""
    L.require_cuda(weight)
    if not isinstance(weight, torch.nn.Parameter):  # (temporaries: their storage is recycled, no stable identity)
        return _pack_weights(weight, spec, for_dgrad, mode)
    pkey = (id(weight), spec.kh, spec.kw, spec.transposed, bool(for_dgrad), mode)
    if _STEP_PACKS is not None and pkey in _STEP_PACKS:
        return _STEP_PACKS[pkey]
    if _PACK_RECORD is not None:
        _PACK_RECORD[pkey] = (pkey, weight, spec, bool(for_dgrad), mode)
    if weight.requires_grad and torch.cuda.is_current_stream_capturing():
        # a captured training step must re-pack on every replay (the optimizer changes the weights in between): never let a
        # cache hit elide the pack launch from the graph
        return _pack_weights(weight, spec, for_dgrad, mode)
    key = (id(weight), spec.transposed, bool(for_dgrad), mode)
    hit = _PACK_CACHE.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == weight._version and hit[2] == weight.data_ptr():
        return hit[3]
    out = _pack_weights(weight, spec, for_dgrad, mode)
    # the entry dies with the Parameter object (a recycled id() can therefore never hit a stale panel)
    _PACK_CACHE[key] = (weakref.ref(weight, lambda _r, k=key: _PACK_CACHE.pop(k, None)), weight._version, weight.data_ptr(), out)
    return out


def _pack_weights(weight, spec, for_dgrad, mode):
    w = weight.detach()
    if w.dtype != torch.float32 or not w.is_contiguous():
        w = w.float().contiguous()
    d0, d1 = w.shape[0], w.shape[1]
    same = spec.transposed == bool(for_dgrad)
    K, N = (d1, d0) if same else (d0, d1)
    nbytes = L.lib().liso_conv_packed_bytes(K, N, spec.kh * spec.kw, mode)
    out = torch.empty(nbytes, dtype=torch.uint8, device=w.device)
    with torch.cuda.device(w.device):
        L.check(L.lib().liso_conv_pack_weights(L.ptr(w), d0, d1, spec.kh, spec.kw, int(spec.transposed), int(bool(for_dgrad)), mode,
                                               L.ptr(out), L.stream_ptr()), "conv_pack_weights")
    return out


def _vec(mode):
    return 8 if mode == L.CONV_BF16 else 4


def _timer_name(mode, kind, d=None):
    """timer family of a launch; forward / data-gradient launches of conv_roles_kernel get their own families (bench.py's `roofline`
    is about ONE kernel: rocprofv3 lists conv_roles_kernel and conv_igemm_kernel separately, so does the timer)"""
    name = ("conv_bf16_" if mode == L.CONV_BF16 else "conv_f32x3_" if mode == L.CONV_F32X3 else "conv_f32_") + kind
    if d is not None and L.TIMER.enabled:
        kind = L.lib().liso_conv_kernel_kind(ctypes.byref(d))
        name += "_roles" if kind == 1 else "_direct" if kind in (2, 3) else ""
    return name


def _flops(d):
    taps = d.n_taps if d.n_classes == 1 else d.n_taps / d.n_classes  # taps per output pixel (average over parity classes)
    return 2.0 * d.batch * d.ho * d.wo * d.co * d.ci * taps


def _bytes(d, wgrad=False):
    """algorithmic HBM bytes of one launch: the tensor that is read + the tensor that is written once each + the weights
    (forward / data gradient: packed bf16 panels, 2 B x planes; weight gradient: x and dy read once, fp32 dW written)"""
    es = 2 if d.mode == L.CONV_BF16 else 4
    w = d.w_taps * d.ci * d.co
    if wgrad:
        return d.batch * (d.hi * d.wi * d.ci + d.ho * d.wo * d.co) * es + 4 * w
    out_es = 4 if (d.out_f32 or d.mode != L.CONV_BF16) else 2
    return d.batch * (d.hi * d.wi * d.ci * es + d.ho * d.wo * d.co * out_es) + w * (2 if d.mode == L.CONV_BF16 else 4)


def conv_forward(x, weight, bias, spec, in_scale=None, in_shift=None, in_relu=False, out_relu=False, out_dtype=None,
                 want_stats=False, stats_shift=None, packed=None, affine_batch_stride=0, occupancy=None, out=None):
    """x: logical [B,Ci,H,W] (channels-last storage preferred) -> (y logical [B,Co,Ho,Wo] channels-last, stats_partial | None).
    `affine_batch_stride` > 0: in_scale / in_shift hold one vector per sample, that many elements apart (InstanceNorm).
    `occupancy`: fp32 [B,1,H,W] / [B,H,W], 0 where x is exactly zero in every channel (the pillar canvas's occupancy map): blocks
    whose whole input window is unoccupied skip their loads and MFMAs (bit-identical result; no prologue allowed)."""
    L.require_cuda(x, weight)
    mode = _mode(x.dtype)
    xv, xps = as_nhwc(x, _vec(mode))
    B, hi, wi, ci = xv.shape
    co = weight.shape[1] if spec.transposed else weight.shape[0]
    ho, wo = spec.out_hw(hi, wi)
    out_dtype = out_dtype or x.dtype
    if mode != L.CONV_BF16:
        assert out_dtype == torch.float32
    out_f32 = out_dtype == torch.float32
    if out is not None:
        # `out` = (channels-last buffer [B, ho, wo, C_total], first channel): the result becomes channels [first, first + co) of that
        # buffer -- several convolutions write one concatenated map without a concatenation pass
        buf, y_off = out
        assert buf.is_contiguous() and tuple(buf.shape[:3]) == (B, ho, wo) and buf.dtype == out_dtype and y_off + co <= buf.shape[3], \
            (tuple(buf.shape), (B, ho, wo, co), y_off)
        y_ps = buf.shape[3]
        assert y_ps % _vec(mode) == 0 and y_off % 8 == 0
        y_base, y = buf, buf[..., y_off:y_off + co]
    else:
        y = torch.empty((B, ho, wo, co), dtype=out_dtype, device=x.device)
        y_base, y_ps, y_off = y, co, 0
    build = scatter_desc if spec.transposed else gather_desc
    d = build(spec, B, hi, wi, ci, xps, ho, wo, co, y_ps, y_off, mode, out_f32, in_relu, out_relu)
    if affine_batch_stride:
        d = L.ConvDesc.from_buffer_copy(d)  # (descriptors are cached and shared: never edit them in place)
        d.in_affine_batch_stride = int(affine_batch_stride)
    if packed is None:
        packed = pack_weights(weight, spec, False, mode)
    lib = L.lib()
    stats = None
    if want_stats:
        rows = lib.liso_conv_stats_rows(ctypes.byref(d))
        if rows <= 0:
            raise L.LisoHipError("conv_forward: unsupported geometry")
        stats = torch.empty((rows, 2, (co + 63) // 64 * 64), dtype=torch.float32, device=x.device)
    b = bias.detach() if bias is not None else None
    if b is not None and (b.dtype != torch.float32 or not b.is_contiguous()):
        b = b.float().contiguous()
    occ = None
    if occupancy is not None and in_scale is None and spec.transposed is False:
        occ = occupancy
        assert occ.dtype == torch.float32 and occ.is_contiguous() and occ.numel() == B * hi * wi, (occ.shape, occ.dtype)
    with torch.cuda.device(x.device):
        L.check(L.TIMER.launch(_timer_name(mode, "fwd", d), lambda: lib.liso_conv_forward_sparse(
            ctypes.byref(d), L.ptr(xv), L.ptr(packed), L.ptr(b) if b is not None else None,
            L.ptr(in_scale) if in_scale is not None else None, L.ptr(in_shift) if in_shift is not None else None, L.ptr(y_base),
            L.ptr(stats) if stats is not None else None, L.ptr(stats_shift) if stats_shift is not None else None,
            L.ptr(occ) if occ is not None else None, L.stream_ptr()),
            units=_flops(d), nbytes=_bytes(d)), "conv_forward")
    return y.permute(0, 3, 1, 2), stats


def _pad_out_channels(dy, weight, spec, vec):
    """the kernels read channels in 16-B groups: a gradient with 1-3 channels (the head's output convolutions) is padded with
    zero channels (and the weights with zero filters) -- a few KB"""
    co = dy.shape[1]
    pad = (-co) % vec
    if pad == 0:
        return dy, weight
    dy = torch.nn.functional.pad(dy, (0, 0, 0, 0, 0, pad))
    if weight is not None:
        weight = torch.nn.functional.pad(weight.detach(), (0, 0, 0, 0, 0, pad) if spec.transposed else (0, 0, 0, 0, 0, 0, 0, pad))
    return dy, weight


def conv_dgrad(dy, weight, spec, x_shape, out_dtype=None, packed=None):
    """dy: logical [B,Co,Ho,Wo] -> dx logical [B,Ci,Hi,Wi] (gradient w.r.t. the convolution's INPUT x')"""
    L.require_cuda(dy, weight)
    mode = _mode(dy.dtype)
    # 1-3 output channels (the head's last convolutions): dy gets zero channels up to one 16-B group.  A panel packed from the
    # UNPADDED weight is already what the kernel needs (its reduction axis is zero-padded to 16 channels either way), so the weight
    # is only padded when it still has to be packed here.
    dy, weight = _pad_out_channels(dy, weight if packed is None else None, spec, _vec(mode))
    gv, gps = as_nhwc(dy, _vec(mode))
    B, ho, wo, co = gv.shape
    _, ci, hi, wi = x_shape
    out_dtype = out_dtype or dy.dtype
    out_f32 = out_dtype == torch.float32
    build = gather_desc if spec.transposed else scatter_desc
    d = build(spec, B, ho, wo, co, gps, hi, wi, ci, ci, 0, mode, out_f32, False, False)
    alloc = torch.zeros if getattr(d, "sparse_output", False) else torch.empty
    if not spec.transposed and ((ho - 1) * spec.stride - spec.padding + spec.kh < hi or (wo - 1) * spec.stride - spec.padding + spec.kw < wi):
        # trailing input rows / columns that no output window reaches: gradient 0.  (The last window ends at input row
        # (ho - 1) s - p + kh - 1; a non-zero remainder of (hi + 2p - kh) / s alone does not mean uncovered rows -- 3x3 / 2 / 1 on an
        # even map reaches every row -- and the fill it used to trigger was a pass over the whole gradient.)
        alloc = torch.zeros
    dx = alloc((B, hi, wi, ci), dtype=out_dtype, device=dy.device)
    if packed is None:
        packed = pack_weights(weight, spec, True, mode)
    with torch.cuda.device(dy.device):
        L.check(L.TIMER.launch(_timer_name(mode, "dgrad", d), lambda: L.lib().liso_conv_forward(
            ctypes.byref(d), L.ptr(gv), L.ptr(packed), None, None, None, L.ptr(dx), None, None, L.stream_ptr()), units=_flops(d), nbytes=_bytes(d)),
            "conv_dgrad")
    return dx.permute(0, 3, 1, 2)


def conv_wgrad(x, dy, weight_shape, spec, in_scale=None, in_shift=None, in_relu=False, want_bias=True, out_dw=None, out_db=None,
               co_true=None):
    """-> (dw fp32 in torch's layout `weight_shape`, dbias fp32 [Co] | None); None if the geometry is not supported by the
    kernels (the caller raises: there is no library fallback).  `out_dw` / `out_db`: dense fp32 tensors to write into (overwritten)."""
    L.require_cuda(x, dy)
    if (x.shape[1] == 4 and x.dtype == torch.float32 and not spec.transposed and spec.stride == 1 and spec.kh == spec.kw and spec.kh in (5, 7)
            and spec.padding == spec.kh // 2 and in_scale is None and tuple(weight_shape[1:]) == (4, spec.kh, spec.kw)
            and os.environ.get("LISO_WGRAD_SMALLCI", "1") != "0"):
        res = _conv_wgrad_smallci(x, dy, weight_shape, spec, want_bias, out_dw, out_db, co_true)
        if res is not None:
            return res
    # (7x7 kernels on DENSE inputs with many channels run the row-of-taps MFMA kernel below, which re-stages the halo tile once per
    # kernel row: 1.9 ms on the encoders' stem.  None of the networks' layers gets here -- the stem's canvas is sparse and takes
    # conv_wgrad_sparse (0.11 ms), the motion encoder's 2-4-channel layers (update.py:57,66) take _conv_wgrad_smallci above -- and
    # there is no library route for a device tensor: a caller without an occupancy map pays the slow kernel.)
    mode = _mode(x.dtype)
    if dy.dtype != x.dtype:
        dy = dy.to(x.dtype)
    if co_true is None:  # (`co_true` given: the caller already padded dy with zero channels, see _FusedConv.backward)
        co_true = dy.shape[1]
        dy, _ = _pad_out_channels(dy, None, spec, _vec(mode))
    xv, xps = as_nhwc(x, _vec(mode))
    gv, gps = as_nhwc(dy, _vec(mode))
    B, hi, wi, ci = xv.shape
    _, ho, wo, co = gv.shape
    build = scatter_desc if spec.transposed else gather_desc
    d = build(spec, B, hi, wi, ci, xps, ho, wo, co, co, 0, mode, True, in_relu, False)
    if co != co_true:  # zero-padded gradient channels: the kernel computes them, the reduction writes the layer's true filters only
        d = L.ConvDesc.from_buffer_copy(d)
        d.wgrad_co = co_true
    lib = L.lib()
    nbytes = lib.liso_conv_wgrad_workspace_bytes(ctypes.byref(d))
    if nbytes == 0:
        return None
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = out_dw if out_dw is not None else torch.empty(weight_shape, dtype=torch.float32, device=x.device)
    db = (out_db if out_db is not None else torch.empty(co_true, dtype=torch.float32, device=x.device)) if want_bias else None
    with torch.cuda.device(x.device):
        L.check(L.TIMER.launch(_timer_name(mode, "wgrad"), lambda: lib.liso_conv_wgrad(
            ctypes.byref(d), L.ptr(xv), L.ptr(in_scale) if in_scale is not None else None,
            L.ptr(in_shift) if in_shift is not None else None, L.ptr(gv), gps, int(spec.transposed), L.ptr(dw),
            L.ptr(db) if db is not None else None, L.ptr(ws), nbytes, L.stream_ptr()), units=_flops(d), nbytes=_bytes(d, True)), "conv_wgrad")
    return dw, db


def _conv_wgrad_smallci(x, dy, weight_shape, spec, want_bias, out_dw, out_db, co_true):
    """the motion encoder's 7x7 layers on 2-4 input channels (update.py:57,66): liso_conv_wgrad_smallci_f32 -- a wave per input channel,
    a lane per output channel, the filter's k x k accumulators and a sliding input window in registers (no 64-channel tile padding)"""
    if dy.dtype != torch.float32:
        dy = dy.float()
    xv, xps = as_nhwc(x, 4)
    gv, gps = as_nhwc(dy, 4)
    B, hi, wi, _ = xv.shape
    co = co_true if co_true is not None else gv.shape[3]
    if gv.shape[1:3] != (hi, wi) or weight_shape[0] != co:
        return None
    lib = L.lib()
    nbytes = lib.liso_conv_wgrad_smallci_workspace_bytes(B, hi, wi, co, spec.kh)
    if nbytes == 0:
        return None
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = out_dw if out_dw is not None else torch.empty(weight_shape, dtype=torch.float32, device=x.device)
    db = (out_db if out_db is not None else torch.empty(co, dtype=torch.float32, device=x.device)) if want_bias else None
    with torch.cuda.device(x.device):
        L.check(L.TIMER.launch("conv_wgrad_smallci", lambda: lib.liso_conv_wgrad_smallci_f32(
            L.ptr(xv), xps, L.ptr(gv), gps, B, hi, wi, co, spec.kh, L.ptr(dw), L.ptr(db) if db is not None else None, L.ptr(ws), nbytes,
            L.stream_ptr()), units=2.0 * B * hi * wi * co * 4 * spec.kh * spec.kw), "conv_wgrad_smallci")
    return dw, db


def conv_wgrad_sparse(x, occupancy, dy, weight_shape, spec, want_bias=True):
    """weight (and bias) gradient of a convolution whose fp32 input is a sparse canvas with an occupancy map (the encoders' 7x7 / 2
    stem): only occupied cells are visited (liso_conv_wgrad_sparse_f32) -> (dw fp32 `weight_shape`, dbias | None), or None when the
    geometry is not covered (the caller takes the dense path)"""
    L.require_cuda(x, dy, occupancy)
    if x.dtype != torch.float32 or spec.transposed or occupancy.dtype != torch.float32:
        return None
    if dy.dtype != torch.float32:
        dy = dy.float()
    xv, xps = as_nhwc(x, 4)
    gv, gps = as_nhwc(dy, 4)
    B, hi, wi, ci = xv.shape
    _, ho, wo, co = gv.shape
    occ = occupancy.contiguous()
    if occ.numel() != B * hi * wi:
        return None
    d = gather_desc(spec, B, hi, wi, ci, xps, ho, wo, co, co, 0, _mode(x.dtype), True, False, False)
    lib = L.lib()
    nbytes = lib.liso_conv_wgrad_sparse_workspace_bytes(ctypes.byref(d))
    if nbytes == 0:
        return None
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    dw = torch.empty(weight_shape, dtype=torch.float32, device=x.device)
    db = torch.empty(co, dtype=torch.float32, device=x.device) if want_bias else None
    with torch.cuda.device(x.device):
        L.check(L.TIMER.launch("conv_wgrad_sparse", lambda: lib.liso_conv_wgrad_sparse_f32(
            ctypes.byref(d), L.ptr(xv), L.ptr(occ), L.ptr(gv), gps, L.ptr(dw), L.ptr(db) if db is not None else None, L.ptr(ws), nbytes,
            L.stream_ptr()), units=_flops(d), nbytes=_bytes(d, True)), "conv_wgrad_sparse")
    return dw, db


def supported(x, weight, spec):
    """can the own kernels run this convolution (forward, data and weight gradient)?"""
    if not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float32):
        return False
    vec = 8 if x.dtype == torch.bfloat16 else 4
    ci = x.shape[1]
    co = weight.shape[1] if spec.transposed else weight.shape[0]
    if ci % vec or spec.kh * spec.kw > L.CONV_MAX_TAPS:
        return False
    if spec.transposed and (spec.kh != spec.stride or spec.padding != 0):
        return False
    return co >= 1


# ---- BatchNorm folding ---------------------------------------------------------------------------------------------------
class BnFold:
    """BatchNorm2d(+ReLU) layers that have NOT been applied yet: the consumer convolution applies them in its prologue.
    One group per BatchNorm module, covering consecutive channel ranges of the raw tensor (several groups = the raw tensor
    is a channel concatenation of separately normalised maps).  Per group: `stats` fp32 [4C] = scale | shift | mean |
    invstd (the layout of include/liso_bn.h) and the module's gamma / beta parameters, which receive their gradients
    through the consumer's backward."""

    def __init__(self, groups, relu=True, training=True):
        self.groups, self.relu, self.training = list(groups), bool(relu), bool(training)
        self._ss = None

    @property
    def channels(self):
        return sum(g["gamma"].shape[0] for g in self.groups)

    def scale_shift(self):
        if self._ss is None:
            if len(self.groups) == 1:
                st, C = self.groups[0]["stats"], self.groups[0]["gamma"].shape[0]
                self._ss = (st[:C], st[C:2 * C])
            else:
                self._ss = (torch.cat([g["stats"][:g["gamma"].shape[0]] for g in self.groups]).contiguous(),
                            torch.cat([g["stats"][g["gamma"].shape[0]:2 * g["gamma"].shape[0]] for g in self.groups]).contiguous())
        return self._ss

    def group(self, k):
        return BnFold([self.groups[k]], self.relu, self.training)

    @staticmethod
    def cat(folds):
        assert all(f.relu == folds[0].relu and f.training == folds[0].training for f in folds)
        return BnFold([g for f in folds for g in f.groups], folds[0].relu, folds[0].training)

    def params(self):
        return [t for g in self.groups for t in (g["gamma"], g["beta"])]


def finalize_bn(stats_partial, n_pixels, bn, stats_shift=None, channel_offset=0):
    """per-block partial sums of a conv_forward(..., want_stats=True) -> BnFold of `bn` in training mode (batch statistics,
    running statistics updated with the module's momentum, like torch.nn.BatchNorm2d.forward).  `channel_offset`: first
    channel of this BatchNorm inside the convolution's output (several BatchNorms behind one merged convolution)."""
    C = bn.num_features
    stats = torch.empty(4 * C, dtype=torch.float32, device=stats_partial.device)
    rows, _, cop = stats_partial.shape
    mom = bn.momentum if bn.momentum is not None else 0.1
    track = bn.track_running_stats and bn.training
    part = ctypes.c_void_p(stats_partial.data_ptr() + 4 * channel_offset)
    with torch.cuda.device(stats_partial.device):
        L.check(L.lib().liso_conv_bn_finalize(
            part, rows, C, cop, int(n_pixels), L.ptr(stats_shift) if stats_shift is not None else None,
            L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(bn.running_mean) if track else None, L.ptr(bn.running_var) if track else None,
            float(mom), float(bn.eps), L.ptr(stats), L.stream_ptr()), "conv_bn_finalize")
    return BnFold([{"stats": stats, "gamma": bn.weight, "beta": bn.bias}], True, True)


def eval_bn_fold(bn, relu=True):
    """BnFold from the running statistics (eval mode)"""
    with torch.no_grad():
        invstd = torch.rsqrt(bn.running_var.float() + bn.eps)
        scale = bn.weight.detach().float() * invstd
        shift = bn.bias.detach().float() - bn.running_mean.float() * scale
        stats = torch.cat([scale, shift, bn.running_mean.float(), invstd]).contiguous()
    return BnFold([{"stats": stats, "gamma": bn.weight, "beta": bn.bias}], relu, False)


_DIRECT_GRADS = False
_DIRECT_TOUCHED = set()  # parameters written in place during the current step: a second contribution goes through autograd's add


def set_direct_grads(on, keep_touched=False):
    """While on, the backward of `_FusedConv` writes the gradients of leaf parameters that already own a dense fp32 `.grad` buffer
    (a trainer's flat gradient views, zeroed at the start of the step) straight into that buffer -- the weight-gradient reduction,
    the bias reduction and the BatchNorm backward take it as their output pointer -- and reports None to autograd: no AccumulateGrad
    `add` per parameter (104 launches per detector step).  The first contribution of a step overwrites the (zeroed) buffer, any
    further one to the same parameter is returned to autograd and added."""
    global _DIRECT_GRADS
    _DIRECT_GRADS = bool(on)
    if not keep_touched:  # (`keep_touched`: the second half of a backward pass that was split in two, see GradCut)
        _DIRECT_TOUCHED.clear()


def direct_touched():
    """ids of the parameters whose gradient was written in place since the last `set_direct_grads(True)`"""
    return set(_DIRECT_TOUCHED)


def _direct_target(p):
    """the dense fp32 .grad buffer of leaf parameter `p`, or None"""
    if not _DIRECT_GRADS or not isinstance(p, torch.nn.Parameter):
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape or id(p) in _DIRECT_TOUCHED:
        return None
    _DIRECT_TOUCHED.add(id(p))  # (a BatchNorm consumed by two layers -- block output -> next block and deblock -- contributes twice)
    return g


_BN_TICKETS = {}  # device index -> [int32 zeros, {layer key: slot}]


def _bn_ticket(key, device):
    """the layer's persistent device counter for liso_bn_relu_bwd_ticket (zero between calls), or None: the pool of a device is
    created on the first EAGER call (memory allocated while a graph is being captured belongs to that graph's pool) and, like every
    trainer here, a captured step is preceded by an eager warm-up pass; LISO_BN_TICKET=1 enables it"""
    if os.environ.get("LISO_BN_TICKET", "0") != "1":
        # MEASURED (round 4): slower than the separate finalize launch it saves -- detector step 4.57-4.63 vs 4.46 ms, loop 5.66 vs
        # 5.65 ms (with plain stores + __threadfence(): 4.84 / 6.03 ms -- a device-scope release fence writes back the XCD's L2).
        # Inside a hipGraph the extra launch costs ~2 us of stream time; the last block's 256-thread finalize over L2-bypassing
        # loads costs more.  Off unless asked for.
        return None
    ent = _BN_TICKETS.get(device.index)
    if ent is None:
        if torch.cuda.is_current_stream_capturing():
            return None
        ent = _BN_TICKETS[device.index] = [torch.zeros(4096, dtype=torch.int32, device=device), {}]
    slot = ent[1].get(key)
    if slot is None:
        if len(ent[1]) >= ent[0].numel():
            return None
        slot = ent[1][key] = len(ent[1])
    return ent[0][slot:slot + 1]


def _rows_view(t, vec):
    """physical [B,H,W,C] view of a logical-NCHW tensor whose (pixel, channel) rows are regular -- dense, or a channel slice of a wider
    channels-last tensor -- and its row stride in elements; (None, 0) otherwise"""
    v = t.permute(0, 2, 3, 1)
    B, H, W, C = v.shape
    ps = _pix_stride(v)
    regular = (v.stride(3) == 1 or C == 1) and ps >= C and (W == 1 or v.stride(2) == ps) and (H == 1 or v.stride(1) == W * ps) and \
        (B == 1 or v.stride(0) == H * W * ps) and ps % vec == 0 and v.data_ptr() % 16 == 0
    return (v, ps) if regular else (None, 0)


def _bn_backward_group(g, x_raw, grp, relu, training, out=None):
    """gradient through relu?(bn(x_raw)) of ONE BatchNorm given g = dL/d(output): -> (dx_raw, dgamma, dbeta); g, x_raw logical NCHW.
    Channel slices of wider channels-last tensors are read in place (liso_bn_relu_bwd_strided); `out`: a logical-NCHW tensor (e.g. the
    group's channel slice of the concatenated input gradient) to write dx_raw into."""
    C = grp["gamma"].shape[0]
    vec = 8 if x_raw.dtype == torch.bfloat16 else 4
    xv, xs = _rows_view(x_raw, vec)
    if xv is None:
        xv, xs = x_raw.permute(0, 2, 3, 1).contiguous(), C
    if g.dtype != xv.dtype:
        g = g.to(xv.dtype)
    gv, gs = _rows_view(g, vec)
    if gv is None:
        gv, gs = g.permute(0, 2, 3, 1).contiguous(), C
    M = xv.numel() // C
    lib = L.lib()
    dxv, ds = _rows_view(out, vec) if out is not None else (None, 0)
    if dxv is None:
        dxv, ds = torch.empty(xv.shape, dtype=xv.dtype, device=xv.device), C
    tg, tb = _direct_target(grp["gamma"]), _direct_target(grp["beta"])
    direct = tg is not None and tb is not None
    gg = tg if direct else torch.empty(C, dtype=torch.float32, device=xv.device)
    gb = tb if direct else torch.empty(C, dtype=torch.float32, device=xv.device)
    nbytes = lib.liso_bn_workspace_bytes(C)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=xv.device)
    bf = int(xv.dtype == torch.bfloat16)
    units = 5 * M * C * xv.element_size()
    with torch.cuda.device(xv.device):
        if (xs, gs, ds) != (C, C, C):
            L.check(L.TIMER.launch("bn_bwd", lambda: lib.liso_bn_relu_bwd_strided(
                L.ptr(gv), gs, L.ptr(xv), xs, bf, M, C, L.ptr(grp["gamma"]), L.ptr(grp["stats"]), int(training), int(relu), L.ptr(dxv), ds,
                L.ptr(gg), L.ptr(gb), L.ptr(ws), nbytes, L.stream_ptr()), units=units), "bn_relu_bwd_strided")
        else:
            args = (L.ptr(gv), L.ptr(xv), bf, M, C, L.ptr(grp["gamma"]), L.ptr(grp["stats"]), int(training), int(relu), L.ptr(dxv), L.ptr(gg),
                    L.ptr(gb), L.ptr(ws), nbytes)
            ticket = _bn_ticket(grp.get("ticket_key", id(grp["gamma"])), xv.device)
            if ticket is not None:  # two launches: the reduction's last block also finalises (include/liso_bn.h)
                L.check(L.TIMER.launch("bn_bwd", lambda: lib.liso_bn_relu_bwd_ticket(*args, L.ptr(ticket), L.stream_ptr()), units=units),
                        "bn_relu_bwd_ticket")
            else:
                L.check(L.TIMER.launch("bn_bwd", lambda: lib.liso_bn_relu_bwd(*args, L.stream_ptr()), units=units), "bn_relu_bwd")
    return dxv.permute(0, 3, 1, 2), (None if direct else gg), (None if direct else gb)


def _bn_backward(g, x_raw, fold):
    """-> (dx_raw logical NCHW, [dgamma0, dbeta0, dgamma1, dbeta1, ...])"""
    if len(fold.groups) == 1:
        dx, gg, gb = _bn_backward_group(g, x_raw, fold.groups[0], fold.relu, fold.training)
        return dx, [gg, gb]
    # several BatchNorms over consecutive channel ranges of one raw tensor (the deblocks' concatenation in front of the head's shared
    # convolution, the four heads' hidden maps): BatchNorm is per channel, so ONE backward over all channels with the groups'
    # gamma / statistics concatenated (two small launches) replaces a slice copy of x and g, three kernels per group and the
    # concatenation of the partial results; the per-group parameter gradients are slices of its two output vectors
    Cs = [grp["gamma"].shape[0] for grp in fold.groups]
    if sum(Cs) > 256:  # (the BatchNorm kernels take up to 256 channels per call: e.g. the deblocks' 3 x 128-channel concatenation)
        # every group reads its channel slice of g / x_raw in place and writes its slice of ONE input-gradient tensor: no slice
        # copies, no concatenation
        B, _, H, W = x_raw.shape
        dx_full = torch.empty((B, H, W, sum(Cs)), dtype=x_raw.dtype, device=x_raw.device).permute(0, 3, 1, 2)
        grads, a = [], 0
        for grp in fold.groups:
            C = grp["gamma"].shape[0]
            dx, gg, gb = _bn_backward_group(g[:, a:a + C], x_raw[:, a:a + C], grp, fold.relu, fold.training, out=dx_full[:, a:a + C])
            if dx.data_ptr() != dx_full[:, a:a + C].data_ptr():  # (irregular layout: the group wrote a tensor of its own)
                dx_full[:, a:a + C].copy_(dx)
            grads += [gg, gb]
            a += C
        return dx_full, grads
    gam = torch.cat([grp["gamma"].detach() for grp in fold.groups])
    stats = torch.cat([grp["stats"][k * c:(k + 1) * c] for k in range(4) for grp, c in zip(fold.groups, Cs)])  # scale | shift | mean | invstd
    dx, gg, gb = _bn_backward_group(g, x_raw, {"gamma": gam, "beta": None, "stats": stats, "ticket_key": id(fold.groups[0]["gamma"])},
                                    fold.relu, fold.training)
    grads, a = [], 0
    for c in Cs:
        grads += [gg[a:a + c], gb[a:a + c]]
        a += c
    return dx, grads


class _FusedConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x_raw, weight, bias, meta, *fold_params):
        """meta: dict(spec, fold (BnFold | None; `fold_params` = its gamma / beta tensors, passed so that autograd routes
        their gradients), out_dtype, want_stats, stats_shift, out_relu); the partial statistics are handed back through
        meta['stats_partial']"""
        spec, fold = meta["spec"], meta["fold"]
        sc, sh = fold.scale_shift() if fold is not None else (None, None)
        occ = meta.get("occupancy") if fold is None else None  # (sparse pillar canvas: sparse / tile-skipping forward, cell list backward)
        res = None
        if occ is not None and meta.get("out_dtype") in (None, x_raw.dtype):
            # a stride-2 convolution on the pillar canvas (SLIM stem, the detector's first layer): only occupied cells are multiplied
            res = _sparse_stem(x_raw, occ, weight, bias, spec, "batch" if meta.get("want_stats", False) else "none",
                               meta.get("out_relu", False), stats_shift=meta.get("stats_shift"))
        sparse_ws = None
        if res is not None:
            y, part, sparse_ws = res
        else:
            y, part = conv_forward(x_raw, weight, bias, spec, sc, sh, in_relu=fold.relu if fold is not None else False,
                                   out_relu=meta.get("out_relu", False), out_dtype=meta.get("out_dtype"),
                                   want_stats=meta.get("want_stats", False), stats_shift=meta.get("stats_shift"), occupancy=occ,
                                   out=meta.get("out"))
        meta["stats_partial"] = part
        relu = bool(meta.get("out_relu", False))
        ctx.save_for_backward(x_raw, weight, y if relu else None)
        ctx.meta = {"spec": spec, "fold": fold, "has_bias": bias is not None, "n_fold_params": len(fold_params), "relu": relu,
                    "bias_param": bias if isinstance(bias, torch.nn.Parameter) else None, "occupancy": occ, "sparse_ws": sparse_ws}
        return y

    @staticmethod
    def backward(ctx, dy):
        x_raw, weight, y = ctx.saved_tensors
        spec, fold = ctx.meta["spec"], ctx.meta["fold"]
        if dy.dtype != x_raw.dtype:
            dy = dy.to(x_raw.dtype)
        if ctx.meta["relu"]:  # the ReLU ran in the convolution's epilogue: its mask is the sign of the stored output
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)  # dy where y > 0 else 0, one launch
        sc, sh = fold.scale_shift() if fold is not None else (None, None)
        dw = db = dx = None
        fold_grads = [None] * ctx.meta["n_fold_params"]
        co_true = dy.shape[1]
        if co_true % _vec(_mode(dy.dtype)):  # 1-3 channels: one zero-padded copy serves the weight AND the data gradient
            dy, _ = _pad_out_channels(dy, None, spec, _vec(_mode(dy.dtype)))
        if ctx.needs_input_grad[1] or (ctx.meta["has_bias"] and ctx.needs_input_grad[2]):
            tw = _direct_target(weight)
            tb = _direct_target(ctx.meta["bias_param"]) if tw is not None and ctx.meta["has_bias"] else None
            if ctx.meta["has_bias"] and tb is None:
                tw = None  # (both or none: one launch produces both)
            res = None
            if ctx.meta.get("occupancy") is not None and co_true == dy.shape[1]:
                res = conv_wgrad_sparse(x_raw, ctx.meta["occupancy"], dy, tuple(weight.shape), spec, want_bias=ctx.meta["has_bias"])
                tw = None if res is not None else tw
            if res is None:
                side = _WGRAD_SIDE if (tw is not None and dy.is_cuda) else None
                if side is not None:  # (written in place: autograd never sees the result -- see wgrad_side)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dy.device))
                    with torch.cuda.stream(side["stream"]):
                        side["stream"].wait_event(ev)
                        res = conv_wgrad(x_raw, dy, tuple(weight.shape), spec, sc, sh, in_relu=fold.relu if fold is not None else False,
                                         want_bias=ctx.meta["has_bias"], out_dw=tw, out_db=tb, co_true=co_true)
                    side["keep"].append((x_raw, dy, sc, sh))
                    side["used"] = True
                else:
                    res = conv_wgrad(x_raw, dy, tuple(weight.shape), spec, sc, sh, in_relu=fold.relu if fold is not None else False,
                                     want_bias=ctx.meta["has_bias"], out_dw=tw, out_db=tb, co_true=co_true)
            if res is None:  # (none of the networks' layers; there is no library route for a device tensor)
                raise NotImplementedError(f"liso_amd: no device weight-gradient kernel for {tuple(weight.shape)} stride {spec.stride} "
                                          f"on {tuple(x_raw.shape)} {x_raw.dtype}")
            else:
                dw, db = res
                if tw is not None and dw is tw:  # written in place: nothing for autograd to accumulate
                    dw, db = None, None
            if dw is not None:
                dw = dw.to(weight.dtype)
            if db is not None:
                db = db.to(weight.dtype)
        if ctx.needs_input_grad[0] or any(ctx.needs_input_grad[4:]):
            g = None
            if ctx.meta.get("occupancy") is not None and fold is None and co_true == dy.shape[1]:
                g = _sparse_dgrad(dy, ctx.meta["occupancy"], weight, spec, tuple(x_raw.shape), x_raw.dtype, lists=ctx.meta.get("sparse_ws"))
            if g is None:
                g = conv_dgrad(dy, weight, spec, tuple(x_raw.shape))
            if fold is not None:
                dx, fold_grads = _bn_backward(g, x_raw, fold)
            else:
                dx = g
        return (dx, dw, db, None, *fold_grads)


def fused_conv(x_raw, fold, conv, out_bn=None, out_dtype=None, out_relu=False, spec=None, occupancy=None, out=None):
    """y_raw = conv(relu?(bn(x_raw))) (+ bias).  `fold`: BnFold pending on x_raw or None.  `conv`: one nn.Conv2d /
    nn.ConvTranspose2d, or a list of nn.Conv2d with the same geometry and input (run as ONE convolution with the filters
    concatenated along the output channels).  `out_bn`: the BatchNorm2d (list: one per convolution of the list) that follows
    -> returns (y_raw, BnFold) (training: batch statistics from the convolution's epilogue; eval: running statistics);
    without it returns (y_raw, None)."""
    convs = list(conv) if isinstance(conv, (list, tuple)) else [conv]
    bns = (list(out_bn) if isinstance(out_bn, (list, tuple)) else [out_bn]) if out_bn is not None else None
    spec = spec or ConvSpec.of(convs[0])
    if len(convs) == 1:
        weight, bias = convs[0].weight, convs[0].bias
    elif torch.is_grad_enabled() and any(c.weight.requires_grad for c in convs):
        weight = torch.cat([c.weight for c in convs], dim=0)
        bias = torch.cat([c.bias for c in convs], dim=0) if convs[0].bias is not None else None
    else:  # frozen / inference: the concatenated filters are built once per weight version (their packed panels are cached with them)
        key = tuple((id(c), c.weight._version, c.weight.data_ptr(), None if c.bias is None else c.bias._version) for c in convs)
        hit = getattr(convs[0], "_liso_merged_weights", None)
        if hit is None or hit[0] != key:
            w = torch.nn.Parameter(torch.cat([c.weight.detach() for c in convs], dim=0), requires_grad=False)
            b = torch.cat([c.bias.detach() for c in convs], dim=0) if convs[0].bias is not None else None
            hit = convs[0]._liso_merged_weights = (key, w, b)
        weight, bias = hit[1], hit[2]
    training_bn = bns is not None and (bns[0].training or not bns[0].track_running_stats)
    # (`out`: (channels-last buffer, first channel) the raw output is written into, see conv_forward)
    meta = {"spec": spec, "fold": fold, "out_dtype": out_dtype, "want_stats": training_bn, "out_relu": out_relu, "occupancy": occupancy,
            "out": out if occupancy is None else None}
    if training_bn and len(bns) == 1 and bns[0].track_running_stats:
        # any per-channel constant close to the mean keeps the sums well conditioned: the running mean.  (The finalize kernel
        # reads stats_shift[c] before the same thread updates running_mean[c]: passing the live buffer is safe.)
        meta["stats_shift"] = bns[0].running_mean
    params = fold.params() if fold is not None else []
    y = _FusedConv.apply(x_raw, weight, bias, meta, *params)
    if bns is None:
        return y, None
    if not training_bn:
        return y, BnFold.cat([eval_bn_fold(b, relu=True) for b in bns])
    n = y.shape[0] * y.shape[2] * y.shape[3]
    folds, off = [], 0
    for b in bns:
        if b.training and b.track_running_stats and not getattr(b, "_liso_counter_deferred", False):
            b.num_batches_tracked += 1
        folds.append(finalize_bn(meta["stats_partial"], n, b, meta.get("stats_shift"), channel_offset=off))
        off += b.num_features
    return y, BnFold.cat(folds)


# ---- InstanceNorm folding (inference) ----------------------------------------------------------------------------------------
class InFold:
    """InstanceNorm2d (+ReLU) that has NOT been applied yet to a raw convolution output: per-sample `stats` fp32 [B, 4C] =
    scale | shift | mean | invstd (liso_conv_in_finalize).  Consumers apply it on the fly: the next convolution in its prologue
    (per-sample vectors), a residual tail in `residual_relu`.  No autograd: the SLIM encoders use it under no_grad."""

    def __init__(self, stats, channels, relu=True):
        self.stats, self.channels, self.relu = stats, int(channels), bool(relu)

    @property
    def scale(self):
        return self.stats  # element [b * 4C + c]

    @property
    def shift(self):
        return self.stats[:, self.channels:]  # element [b * 4C + C + c]: same stride, offset C

    @property
    def stride(self):
        return 4 * self.channels


def _norm_kind(norm):
    """'none' | 'instance' | None (a layer this path does not fold)"""
    if isinstance(norm, torch.nn.Sequential) and len(norm) == 0:
        return "none"
    if isinstance(norm, torch.nn.InstanceNorm2d) and not norm.track_running_stats:
        return "instance"
    return None


@torch.no_grad()
def conv_in(x_raw, fold, conv, norm, relu=True, spec=None, occupancy=None):
    """inference: conv(pending(x_raw)) followed by `norm` (InstanceNorm2d or nothing) and an optional ReLU.
    -> (y_raw, InFold | None): with InstanceNorm the output stays raw and the normalisation (+ReLU) pending; without a
    normalisation the ReLU runs in the convolution's epilogue."""
    spec = spec or ConvSpec.of(conv)
    kind = _norm_kind(norm)
    kw = {}
    if fold is not None:
        kw = dict(in_scale=fold.scale, in_shift=fold.shift, in_relu=fold.relu, affine_batch_stride=fold.stride)
    if fold is None and occupancy is not None:
        res = _sparse_stem(x_raw, occupancy, conv.weight, conv.bias, spec, kind, relu)
        if res is not None:
            y, part, _ = res
            if kind == "none":
                return y, None
            return y, _in_fold_from_partial(y, part, norm, relu)
        kw["occupancy"] = occupancy
    if kind == "none":
        y, _ = conv_forward(x_raw, conv.weight, conv.bias, spec, out_relu=relu, **kw)
        return y, None
    y, part = conv_forward(x_raw, conv.weight, conv.bias, spec, want_stats=True, **kw)
    return y, _in_fold_from_partial(y, part, norm, relu)


def _in_fold_from_partial(y, part, norm, relu):
    """per-block partial sums `part` [rows, 2, co_pad] of the raw output y -> its pending InstanceNorm (+ReLU)"""
    B, C, H, W = y.shape
    stats = torch.empty((B, 4 * C), dtype=torch.float32, device=y.device)
    rows, _, cop = part.shape
    with torch.cuda.device(y.device):
        L.check(L.lib().liso_conv_in_finalize(L.ptr(part), rows // B, B, C, cop, H * W,
                                              L.ptr(norm.weight) if norm.affine else None, L.ptr(norm.bias) if norm.affine else None,
                                              float(norm.eps), L.ptr(stats), L.stream_ptr()), "conv_in_finalize")
    return InFold(stats, C, relu)


SPARSE_STEM_MAX_CELLS = 40960   # capacity of the cell lists per sample (the voxeliser emits at most 40000 pillars per sweep)
_SPARSE_OVERFLOW = {}           # device index -> int32 [1]: set by the kernels when a batch held more occupied cells than the capacity


def sparse_stem_overflowed(device):
    """did any sparse stem convolution on `device` drop cells since the process started (one device -> host read)?"""
    t = _SPARSE_OVERFLOW.get(_dev_index(torch.device(device)))
    return bool(t is not None and int(t.item()) != 0)


def reset_sparse_stem_overflow(device):
    """clear the sticky overflow flag of `device` (tests that overflow the cell lists on purpose)"""
    t = _SPARSE_OVERFLOW.get(_dev_index(torch.device(device)))
    if t is not None:
        t.zero_()


def _dev_index(dev):
    """torch.device('cuda') has index None: the flag table is keyed by the index the kernels' tensors report"""
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _sparse_geometry(x_raw, weight, spec):
    """(is_bf16, k, co) if the sparse-canvas kernels cover this convolution, else None: 7x7 / 2 / 3 with 32 filters on fp32 tensors (the
    SLIM stem) or 3x3 / 2 / 1 with 64 filters on bf16 / fp32 tensors (the detector's first layer), 64 input channels, an even canvas
    height and a width that is a multiple of 64; fp32 tensors only in F32X3 arithmetic (the exact-fp32 mode keeps the dense kernels)"""
    if os.environ.get("LISO_SPARSE_STEM", "1") == "0" or spec.transposed or x_raw.dim() != 4:
        return None
    B, C, H, W = x_raw.shape
    geo = (spec.kh, spec.kw, spec.stride, spec.padding, weight.shape[0])
    bf = x_raw.dtype == torch.bfloat16
    if C != 64 or tuple(weight.shape[1:]) != (64, spec.kh, spec.kw) or H % 2 or W % 64:
        return None
    if geo == (7, 7, 2, 3, 32) and x_raw.dtype == torch.float32 and fp32_mode() == "x3":
        return False, 7, 32
    if geo == (3, 3, 2, 1, 64) and (bf or (x_raw.dtype == torch.float32 and fp32_mode() == "x3")):
        return bf, 3, 64
    return None


def _sparse_flag(dev):
    idx = _dev_index(dev)
    flag = _SPARSE_OVERFLOW.get(idx)
    if flag is None:
        if torch.cuda.is_current_stream_capturing():
            return None  # (first use inside a capture: the persistent flag cannot be created here; a warm-up pass creates it)
        flag = _SPARSE_OVERFLOW[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
    return flag


def _occupancy_f32(occupancy, n):
    occ = occupancy if occupancy.dtype == torch.float32 else occupancy.float()
    occ = occ.contiguous()
    return occ if occ.numel() == n else None


def _sparse_stem(x_raw, occupancy, weight, bias, spec, kind, relu, stats_shift=None):
    """A stride-2 convolution on the pillar canvas in its sparse form (liso_sparse_conv_forward): only occupied cells are multiplied.
    `kind`: "none" (no statistics; `relu` in the epilogue) | "instance" / "batch" (raw output + per-block statistics partial sums,
    shifted by `stats_shift`).  -> (y logical NCHW, partial sums [blocks, 2, co] | None) or None (the caller takes the dense kernel)."""
    geo = _sparse_geometry(x_raw, weight, spec)
    if geo is None:
        return None
    bf, k, co = geo
    B, C, H, W = x_raw.shape
    occ = _occupancy_f32(occupancy, B * H * W)
    dev = x_raw.device
    flag = _sparse_flag(dev)
    if occ is None or flag is None:
        return None
    xv, xps = as_nhwc(x_raw, 8 if bf else 4)
    lib = L.lib()
    cap = min(SPARSE_STEM_MAX_CELLS, H * W)
    nbytes = lib.liso_sparse_conv_workspace_bytes(B, H, W, k, co, cap, 0)
    if nbytes == 0:
        return None
    mode = L.CONV_BF16 if bf else L.CONV_F32X3
    packed = pack_weights(weight, spec, False, mode)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    ho, wo = H // 2, W // 2
    y = torch.empty((B, ho, wo, co), dtype=x_raw.dtype, device=dev)
    groups = lib.liso_sparse_conv_stat_groups(H, W, co)
    if groups <= 0:
        return None
    part = torch.empty((B * ho * wo * co // (4096 * groups), 2, co), dtype=torch.float32, device=dev) if kind != "none" else None
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("conv_sparse_stem", lambda: lib.liso_sparse_conv_forward(
            L.ptr(xv), xps, int(bf), L.ptr(occ), L.ptr(packed), L.ptr(bias) if bias is not None else None, B, H, W, k, co, cap,
            int(bool(relu) and kind == "none"), L.ptr(y), L.ptr(part) if part is not None else None,
            L.ptr(stats_shift) if (stats_shift is not None and part is not None) else None, L.ptr(flag), L.ptr(ws), nbytes, L.stream_ptr()),
            units=B * ho * wo * co * y.element_size() + B * H * W * 4), "sparse_conv_forward")  # (bytes: the dense output written once + the occupancy map)
    return y.permute(0, 3, 1, 2), part, ws


def _sparse_dgrad(dy, occupancy, weight, spec, x_shape, x_dtype, lists=None):
    """data gradient of that convolution at the occupied cells (zeros elsewhere: nothing reads them -- the pillar encoder's backward
    gathers the canvas gradient at its pillars) -> dx logical NCHW, or None (dense data gradient)"""
    B, C, H, W = x_shape
    if dy.dtype != x_dtype:
        return None
    fake = _SparseShape(x_shape, x_dtype)
    geo = _sparse_geometry(fake, weight, spec)
    if geo is None:
        return None
    bf, k, co = geo
    occ = _occupancy_f32(occupancy, B * H * W)
    dev = dy.device
    flag = _sparse_flag(dev)
    if occ is None or flag is None:
        return None
    gv, gps = as_nhwc(dy, 8 if bf else 4)
    if gv.shape[1:3] != (H // 2, W // 2) or gv.shape[3] != co:
        return None
    lib = L.lib()
    cap = min(SPARSE_STEM_MAX_CELLS, H * W)
    nbytes = lib.liso_sparse_conv_workspace_bytes(B, H, W, k, co, cap, 1)
    if nbytes == 0:
        return None
    packed = pack_weights(weight, spec, True, L.CONV_BF16 if bf else L.CONV_F32X3)
    reuse = lists is not None and lists.numel() >= nbytes  # (the forward call's workspace on this canvas: its cell lists are reused)
    ws = lists if reuse else torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dx = torch.zeros((B, H, W, C), dtype=x_dtype, device=dev)
    with torch.cuda.device(dev):
        L.check(L.TIMER.launch("conv_sparse_dgrad", lambda: lib.liso_sparse_conv_dgrad(
            L.ptr(gv), gps, int(bf), L.ptr(occ), L.ptr(packed), B, H, W, k, co, cap, L.ptr(dx), C, L.ptr(flag), L.ptr(ws), ws.numel(),
            int(reuse), L.stream_ptr()), units=gv.numel() * gv.element_size()), "sparse_conv_dgrad")
        # (bytes the LAUNCH moves, lower bound: dy read once; the rows it writes at the occupied cells are not counted -- their number
        # is only known on the device -- and the zero fill of the canvas gradient is torch.zeros above, outside the timed launch)
    return dx.permute(0, 3, 1, 2)


class _SparseShape:
    """shape / dtype stand-in for `_sparse_geometry` where only the input's shape is at hand (backward)"""

    def __init__(self, shape, dtype):
        self.shape, self.dtype = tuple(shape), dtype

    def dim(self):
        return len(self.shape)


@torch.no_grad()
def residual_relu(a_raw, a_fold, b_raw, b_fold):
    """relu(fa(a) + fb(b)) with the pending InstanceNorm (+ReLU) of either branch applied on the fly; NHWC fp32 in and out"""
    av, _ = as_nhwc(a_raw, 4)
    bv, _ = as_nhwc(b_raw, 4)
    assert av.shape == bv.shape and av.is_contiguous() and bv.is_contiguous() and av.dtype == torch.float32
    B, H, W, C = av.shape
    out = torch.empty_like(av)
    sa = (L.ptr(a_fold.scale), L.ptr(a_fold.shift), a_fold.stride, int(a_fold.relu)) if a_fold is not None else (None, None, 0, 0)
    sb = (L.ptr(b_fold.scale), L.ptr(b_fold.shift), b_fold.stride, int(b_fold.relu)) if b_fold is not None else (None, None, 0, 0)
    with torch.cuda.device(av.device):
        L.check(L.TIMER.launch("residual_affine_relu", lambda: L.lib().liso_residual_affine_relu_f32(
            L.ptr(av), *sa, L.ptr(bv), *sb, L.ptr(out), B, H * W, C, L.stream_ptr()), units=12 * av.numel()), "residual_affine_relu")
    return out.permute(0, 3, 1, 2)


class _AddRelu(torch.autograd.Function):
    """relu(a + b) of two fp32 channels-last maps as ONE launch (the tail of a residual block in training: `self.relu(x + y)`,
    liso/slim/model/extractor.py:38); backward: the ReLU mask from the stored output, the same tensor for both branches"""

    @staticmethod
    def forward(ctx, a, b):
        out = residual_relu(a, None, b, None)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, g):
        (out,) = ctx.saved_tensors
        gm = torch.ops.aten.threshold_backward(g, out, 0.0)
        return gm, gm


def add_relu(a, b):
    if (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape == b.shape and a.dim() == 4
            and a.shape[1] % 4 == 0):
        return _AddRelu.apply(a, b)
    return torch.relu(a + b)


class _Materialize(torch.autograd.Function):
    """relu?(bn(x_raw)) as a tensor, for consumers that are not convolutions (API parity: RPN.forward returns a tensor)"""

    @staticmethod
    def forward(ctx, x_raw, fold, *params):
        sc, sh = fold.scale_shift()
        C = x_raw.shape[1]
        y = x_raw.float() * sc.view(1, C, 1, 1) + sh.view(1, C, 1, 1)
        y = torch.relu(y) if fold.relu else y
        ctx.save_for_backward(x_raw)
        ctx.fold = fold
        return y.to(x_raw.dtype)

    @staticmethod
    def backward(ctx, g):
        (x_raw,) = ctx.saved_tensors
        dx, grads = _bn_backward(g, x_raw, ctx.fold)
        return (dx, None, *grads)


def materialize(x_raw, fold):
    if fold is None:
        return x_raw
    return _Materialize.apply(x_raw, fold, *fold.params())


class _SliceCat(torch.autograd.Function):
    """`full` [B, sum C_k, H, W] already holds the maps `parts` in consecutive channel ranges (their convolutions wrote them there:
    conv_forward(out=...)): the concatenation without a copy.  Backward: the channel slices of the gradient, as views."""

    @staticmethod
    def forward(ctx, full, *parts):
        ctx.sizes = [p.shape[1] for p in parts]
        return full.detach()

    @staticmethod
    def backward(ctx, g):
        return (None, *torch.split(g, ctx.sizes, dim=1))


def slice_cat(full, parts):
    """the concatenated map `full` whose channel ranges `parts` were written in place, connected to the parts' autograd history"""
    if torch.is_grad_enabled() and any(p.requires_grad for p in parts):
        return _SliceCat.apply(full, *parts)
    return full


class GradCut:
    """Splits a backward pass in two at one activation: `split(x)` hands the consumers a detached leaf, `finish()` -- called after
    the loss's backward() has filled that leaf's gradient -- continues the backward pass into the layers that produced x.  A
    data-parallel trainer captures the two halves as two hipGraphs and starts the all-reduce of the gradients the first half
    produced between the two replays (liso_amd/trainer.py); the arithmetic is that of the single backward pass."""

    def __init__(self):
        self.upstream, self.leaf = None, None

    def split(self, x):
        if not x.requires_grad:
            return x
        self.upstream = x
        self.leaf = x.detach().requires_grad_(True)
        return self.leaf

    def finish(self):
        up, leaf, self.upstream, self.leaf = self.upstream, self.leaf, None, None
        if up is not None and leaf.grad is not None:
            up.backward(leaf.grad)


def conv2d(layer, x, relu=False, occupancy=None, out=None):
    """relu?(layer(x)) for an nn.Conv2d / nn.ConvTranspose2d on the own kernels (host tensors -- the CPU test tier -- through
    host_ops; a device tensor whose geometry the kernels do not cover raises).  `occupancy`: fp32 [B,1,H,W] / [B,H,W] map with 0
    where x is zero in every channel (the pillar canvas): empty tiles are skipped forward, the weight gradient walks occupied cells"""
    spec = ConvSpec.of(layer)
    if on_device(x) and supported(x, layer.weight, spec):
        return fused_conv(x, None, layer, out_relu=relu, spec=spec, occupancy=occupancy if x.dtype == torch.float32 else None, out=out)[0]
    if x.is_cuda and not spec.transposed:
        # 1-3 (7) input channels -- the motion encoder's conv_flow1: 7x7 on the 2-channel flow, liso/slim/model/update.py:53-60 --
        # the kernels read channels in 16-B groups: zero channels (and zero filter slices) up to one group, then the own kernel
        vec = 8 if x.dtype == torch.bfloat16 else 4
        pad = (-x.shape[1]) % vec
        xp = torch.nn.functional.pad(x.permute(0, 2, 3, 1), (0, pad)).permute(0, 3, 1, 2)  # one launch: dense NHWC rows of one 16-B group
        if torch.is_grad_enabled() and layer.weight.requires_grad:
            wp = torch.nn.functional.pad(layer.weight, (0, 0, 0, 0, 0, pad))
        else:  # frozen / inference: the padded filter is built once per weight version (its packed panels are cached with it)
            hit = getattr(layer, "_liso_padded_weight", None)
            if hit is None or hit[0] != layer.weight._version or hit[1].device != layer.weight.device:
                wp = torch.nn.Parameter(torch.nn.functional.pad(layer.weight.detach(), (0, 0, 0, 0, 0, pad)), requires_grad=False)
                layer._liso_padded_weight = (layer.weight._version, wp)
            wp = layer._liso_padded_weight[1]
        if supported(xp, wp, spec):
            import types
            return fused_conv(xp, None, types.SimpleNamespace(weight=wp, bias=layer.bias), out_relu=relu, spec=spec, out=out)[0]
    if x.is_cuda:
        raise NotImplementedError(f"liso_amd: no device kernel for {type(layer).__name__} {tuple(layer.weight.shape)} stride {spec.stride} "
                                  f"padding {spec.padding} on {tuple(x.shape)} {x.dtype}")
    assert out is None, "conv2d(out=...): only on the own kernels"
    from liso_amd.utils import host_ops

    y = host_ops.module_forward(layer, x)
    return torch.relu(y) if relu else y
