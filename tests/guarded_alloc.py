"""Guard-band allocation for the `-m gpu` canary tests (GPU AddressSanitizer is not available on this pool).

Inside `guarded()` every tensor the package's wrappers allocate on the GPU with torch.empty / zeros / ones / full / *_like /
Tensor.new_* is carved out of a larger buffer with GUARD bytes of a canary pattern in front of and behind it.  A kernel that
writes outside a caller-provided output or workspace -- fixed box slots, cell lists, the 40 000-pillar cap, kNN buckets, split-K
slabs -- lands in a guard (the caching allocator would have handed it a neighbouring live tensor instead, silently).  `check()`
synchronises and verifies every guard; the bases stay alive until then, so no guard is reused by a later allocation.
Reads outside a buffer are not detected.
"""
import contextlib

import torch

GUARD = 4096  # bytes on either side (a multiple of every alignment the kernels assume)
CANARY = 0xC7


class _Registry:
    def __init__(self):
        self.live = []  # (base uint8 tensor, payload bytes, description)

    def alloc(self, shape, dtype, device, what):
        dtype = dtype or torch.get_default_dtype()
        numel = 1
        for s in shape:
            numel *= int(s)
        item = torch.empty((), dtype=dtype).element_size()
        nbytes = numel * item
        pad = (-nbytes) % 256  # keep the rear guard 256-B aligned like the front one
        base = _REAL["empty"](GUARD + nbytes + pad + GUARD, dtype=torch.uint8, device=device)
        base[:GUARD] = CANARY
        base[GUARD + nbytes:] = CANARY
        self.live.append((base, nbytes, what))
        return base[GUARD:GUARD + nbytes].view(dtype).view(tuple(int(s) for s in shape))

    def check(self):
        torch.cuda.synchronize()
        bad = []
        for base, nbytes, what in self.live:
            front_ok = bool((base[:GUARD] == CANARY).all())
            rear_ok = bool((base[GUARD + nbytes:] == CANARY).all())
            if not (front_ok and rear_ok):
                side = ("front " if not front_ok else "") + ("rear" if not rear_ok else "")
                bad.append(f"{what}: {nbytes} B payload, {side.strip()} guard overwritten")
        n = len(self.live)
        self.live = []
        assert not bad, "out-of-bounds device writes:\n  " + "\n  ".join(bad)
        return n


_REAL = {}


def _shape_of(args, kwargs):
    if "size" in kwargs:
        return tuple(kwargs["size"])
    if len(args) == 1 and isinstance(args[0], (tuple, list, torch.Size)):
        return tuple(args[0])
    return tuple(args)


@contextlib.contextmanager
def guarded():
    """with guarded() as g: ...; g.check()  (check() may be called several times; leaving the block checks once more)"""
    reg = _Registry()
    names = ["empty", "zeros", "ones", "full", "empty_like", "zeros_like", "ones_like"]
    for n in names:
        _REAL[n] = getattr(torch, n)
    real_new = {n: getattr(torch.Tensor, n) for n in ("new_empty", "new_zeros", "new_ones", "new_full")}

    def is_cuda(device):
        return device is not None and torch.device(device).type == "cuda"

    def make(fill):
        def f(*args, dtype=None, device=None, **kw):
            if not is_cuda(device) or kw.get("out") is not None or kw.get("pin_memory") or kw.get("memory_format") not in (None, torch.contiguous_format):
                return _REAL["empty" if fill is None else ("zeros" if fill == 0 else "ones")](*args, dtype=dtype, device=device, **kw)
            t = reg.alloc(_shape_of(args, kw), dtype, device, f"torch.{'empty' if fill is None else ('zeros' if fill == 0 else 'ones')}")
            if fill is not None:
                t.fill_(fill)
            return t.requires_grad_(True) if kw.get("requires_grad") else t
        return f

    def full(size, fill_value, *, dtype=None, device=None, **kw):
        if not is_cuda(device) or kw.get("out") is not None:
            return _REAL["full"](size, fill_value, dtype=dtype, device=device, **kw)
        if dtype is None:
            dtype = torch.get_default_dtype() if isinstance(fill_value, float) else (torch.bool if isinstance(fill_value, bool) else torch.int64)
        return reg.alloc(tuple(size), dtype, device, "torch.full").fill_(fill_value)

    def make_like(fill):
        def f(t, *, dtype=None, device=None, **kw):
            dev = device if device is not None else t.device
            contiguous_like = kw.get("memory_format") in (None, torch.contiguous_format, torch.preserve_format) and t.is_contiguous()
            if not is_cuda(dev) or not contiguous_like:
                return _REAL[("empty" if fill is None else ("zeros" if fill == 0 else "ones")) + "_like"](t, dtype=dtype, device=device, **kw)
            o = reg.alloc(tuple(t.shape), dtype or t.dtype, dev, "torch.*_like")
            if fill is not None:
                o.fill_(fill)
            return o
        return f

    def make_new(fill):
        def f(self, *args, dtype=None, device=None, **kw):
            dev = device if device is not None else self.device
            if not is_cuda(dev):
                return real_new["new_empty" if fill is None else ("new_zeros" if fill == 0 else "new_ones")](self, *args, dtype=dtype, device=device, **kw)
            o = reg.alloc(_shape_of(args, kw), dtype or self.dtype, dev, "Tensor.new_*")
            if fill is not None:
                o.fill_(fill)
            return o
        return f

    def new_full(self, size, fill_value, *, dtype=None, device=None, **kw):
        dev = device if device is not None else self.device
        if not is_cuda(dev):
            return real_new["new_full"](self, size, fill_value, dtype=dtype, device=device, **kw)
        return reg.alloc(tuple(size), dtype or self.dtype, dev, "Tensor.new_full").fill_(fill_value)

    torch.empty, torch.zeros, torch.ones, torch.full = make(None), make(0), make(1), full
    torch.empty_like, torch.zeros_like, torch.ones_like = make_like(None), make_like(0), make_like(1)
    torch.Tensor.new_empty, torch.Tensor.new_zeros, torch.Tensor.new_ones, torch.Tensor.new_full = make_new(None), make_new(0), make_new(1), new_full
    try:
        yield reg
        reg.check()
    finally:
        for n in names:
            setattr(torch, n, _REAL[n])
        for n, fn in real_new.items():
            setattr(torch.Tensor, n, fn)
