"""AdamW over flat buffers: the detector's parameter update as ONE launch of liso_adamw_step_f32 (include/liso_optim.h).

Drop-in for `torch.optim.AdamW(params, lr, weight_decay=0.01)` as the reference builds it (liso/liso_cli.py:792-823): a
torch.optim.Optimizer with one parameter group carrying `lr` and `betas`, so `OneCycleLR` drives it unchanged (it rewrites
the group's lr and beta1 before every step), and `state_dict()` keeps torch's layout (per parameter: step, exp_avg,
exp_avg_sq).  Every parameter's `.data` and `.grad` become strided views (the parameter's own strides, channels-last
included) into flat fp32 buffers with one element order, and so do both moments: the update is a single HBM-bound pass and
data-parallel training all-reduces `flat_grad` in one RCCL call."""
import torch

from liso_amd import _lib as L


def _dense(p):
    """a tensor whose elements tile one contiguous block (contiguous in some dimension order)"""
    return p.is_contiguous() or p.is_contiguous(memory_format=torch.channels_last) or \
        (p.dim() == 5 and p.is_contiguous(memory_format=torch.channels_last_3d))


class FlatAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = [p for p in params if p.requires_grad]
        assert len(params) > 0
        dev = params[0].device
        L.require_cuda(*params)
        assert all(p.dtype == torch.float32 and p.device == dev for p in params), "fp32 master parameters on one device"
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4  # 16-byte aligned views; the gaps stay zero in all four buffers
        self.numel = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_exp_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_exp_avg_sq = torch.zeros(off, dtype=torch.float32, device=dev)
        self._step = 0
        self.grad_scale = 1.0  # factor on the gradients inside the update (data parallelism: 1 / world size behind a SUM all-reduce)
        self.offsets = {id(p): o for p, o in zip(params, offs)}  # first element of every parameter inside the flat buffers
        for p, o in zip(params, offs):
            if not _dense(p.data):
                p.data = p.data.contiguous()
            size, stride = p.shape, p.stride()
            view = lambda flat: torch.as_strided(flat, size, stride, o)  # noqa: E731
            view(self.flat_param).copy_(p.data)
            p.data = view(self.flat_param)
            g = view(self.flat_grad)
            if p.grad is not None:
                g.copy_(p.grad)
            p.grad = g
            self.state[p] = {"step": torch.tensor(0.0), "exp_avg": view(self.flat_exp_avg), "exp_avg_sq": view(self.flat_exp_avg_sq)}

    def zero_grad(self, set_to_none=False):
        """one memset; the .grad views stay (set_to_none is ignored: autograd accumulates into the flat buffer)"""
        self.flat_grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        assert len(self.param_groups) == 1, "one parameter group (the reference's optimizer has one)"
        g = self.param_groups[0]
        self._step += 1
        with torch.cuda.device(self.flat_param.device):
            L.check(L.TIMER.launch("adamw_flat", lambda: L.lib().liso_adamw_step_scaled_f32(
                L.ptr(self.flat_param), L.ptr(self.flat_grad), L.ptr(self.flat_exp_avg), L.ptr(self.flat_exp_avg_sq), self.numel,
                float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                float(self.grad_scale), self._step, L.stream_ptr()), units=28 * self.numel), "adamw_step")
        # the kernel wrote through a raw pointer: tell autograd / the packed-weight cache that every parameter changed
        torch.autograd.graph.increment_version(g["params"])

    def state_dict(self):
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        views = {p: (st["exp_avg"], st["exp_avg_sq"]) for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        step = 0
        for p, (m, v) in views.items():  # keep the moments inside the flat buffers
            st = self.state[p]
            if "exp_avg" in st and "exp_avg_sq" in st:
                m.copy_(st["exp_avg"]), v.copy_(st["exp_avg_sq"])
                step = max(step, int(float(st.get("step", 0))))
            else:  # torch.optim.AdamW saved before its first step: empty per-parameter state = zero moments, step 0
                m.zero_(), v.zero_()
            st["exp_avg"], st["exp_avg_sq"] = m, v
            st.setdefault("step", torch.tensor(0.0))
        self._step = step


class FlatRMSprop(torch.optim.Optimizer):
    """torch.optim.RMSprop(params, lr) with its defaults (alpha 0.99, eps 1e-8, no weight decay, no momentum, not centered) -- what
    liso/slim/experiment.py:200-219 builds for SLIM -- over flat buffers: every parameter's `.data`, `.grad` and `square_avg` are
    views into three flat fp32 buffers, the update is ONE launch of liso_rmsprop_step_f32 (torch's multi-tensor form: five foreach
    launches per chunk of tensors, 13 launches and 0.28 ms per SLIM step), data parallelism all-reduces `flat_grad` in one call.
    One parameter group carrying `lr`, so LambdaLR drives it unchanged; `state_dict()` keeps torch's layout (step, square_avg)."""

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8):
        params = [p for p in params if p.requires_grad]
        assert len(params) > 0
        dev = params[0].device
        L.require_cuda(*params)
        assert all(p.dtype == torch.float32 and p.device == dev for p in params), "fp32 master parameters on one device"
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=0, momentum=0, centered=False))
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.numel = off
        self.flat_param = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_square_avg = torch.zeros(off, dtype=torch.float32, device=dev)
        self._step = 0
        self.grad_scale = 1.0
        self.offsets = {id(p): o for p, o in zip(params, offs)}
        self._grad_views = {}
        for p, o in zip(params, offs):
            if not _dense(p.data):
                p.data = p.data.contiguous()
            size, stride = p.shape, p.stride()
            view = lambda flat: torch.as_strided(flat, size, stride, o)  # noqa: E731
            view(self.flat_param).copy_(p.data)
            p.data = view(self.flat_param)
            g = view(self.flat_grad)
            if p.grad is not None:
                g.copy_(p.grad)
            p.grad = g
            self._grad_views[id(p)] = g
            self.state[p] = {"step": torch.tensor(0.0), "square_avg": view(self.flat_square_avg)}

    def grad_view(self, p):
        """the parameter's slice of `flat_grad` (its `.grad` unless a caller parked `.grad = None` for a captured backward pass)"""
        return self._grad_views[id(p)]

    def zero_grad(self, set_to_none=False):
        """one memset.  `set_to_none`: every `.grad` becomes None, so that autograd hands each parameter's gradient over as a tensor of
        its own instead of launching one `add_` per parameter into the zeroed buffer (accumulate_grad.h); `step()` then moves them
        into their slices with one launch per 48 tensors.  Otherwise the `.grad` views stay / come back."""
        self.flat_grad.zero_()
        for p in self.param_groups[0]["params"]:
            if set_to_none:
                p.grad = None
            elif p.grad is not self._grad_views[id(p)]:
                p.grad = self._grad_views[id(p)]

    def collect_grads(self):
        """gradients that live in tensors of their own (after zero_grad(set_to_none=True)) -> their slices of `flat_grad`, one
        liso_gather_f32 launch per 48 tensors; `.grad` is the flat view again afterwards (a parameter without a gradient keeps its
        zeros: torch skips it, here its square_avg decays and the parameter does not move)"""
        import ctypes

        jobs = []
        for p in self.param_groups[0]["params"]:
            v = self._grad_views[id(p)]
            if p.grad is not None and p.grad is not v:
                g = p.grad if p.grad.dtype == torch.float32 and _dense(p.grad) and p.grad.stride() == v.stride() else None
                if g is None:  # (a layout autograd chose differently: through the view)
                    v.copy_(p.grad)
                else:
                    jobs.append((g, v))
            p.grad = v
        if jobs:
            n = len(jobs)
            src = (ctypes.c_void_p * n)(*[g.data_ptr() for g, _ in jobs])
            dst = (ctypes.c_void_p * n)(*[f.data_ptr() for _, f in jobs])
            cnt = (ctypes.c_size_t * n)(*[g.numel() for g, _ in jobs])
            with torch.cuda.device(self.flat_param.device):
                L.check(L.lib().liso_gather_f32(n, src, dst, cnt, L.stream_ptr()), "gather_f32")
        return [g for g, _ in jobs]

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None and len(self.param_groups) == 1
        g = self.param_groups[0]
        keep = self.collect_grads()  # noqa: F841  (alive until the launch is queued)
        self._step += 1
        with torch.cuda.device(self.flat_param.device):
            L.check(L.TIMER.launch("rmsprop_flat", lambda: L.lib().liso_rmsprop_step_f32(
                L.ptr(self.flat_param), L.ptr(self.flat_grad), L.ptr(self.flat_square_avg), self.numel, float(g["lr"]), float(g["alpha"]),
                float(g["eps"]), float(self.grad_scale), L.stream_ptr()), units=20 * self.numel), "rmsprop_step")
        torch.autograd.graph.increment_version(g["params"])

    def state_dict(self):
        for st in self.state.values():
            st["step"] = torch.tensor(float(self._step))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        views = {p: st["square_avg"] for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        step = 0
        for p, v in views.items():
            st = self.state[p]
            if "square_avg" in st:
                v.copy_(st["square_avg"])
                step = max(step, int(float(st.get("step", 0))))
            else:
                v.zero_()
            st["square_avg"] = v
            st.setdefault("step", torch.tensor(0.0))
        self._step = step
