"""liso_adamw_step_f32 / FlatAdamW (include/liso_optim.h) against the reference's optimizer itself: torch.optim.AdamW driven by
OneCycleLR exactly as liso/liso_cli.py:792-823 builds them, run on the CPU in fp32."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(64, 64, 3, 3), (64,), (7,), (128, 64, 3, 3), (3, 64, 1, 1), (1,), (9, 5), (2, 64, 2, 2)]


def _make(seed, device):
    g = torch.Generator().manual_seed(seed)
    ps = []
    for i, s in enumerate(SHAPES):
        t = torch.randn(s, generator=g) * (0.05 if len(s) > 1 else 1.0)
        if len(s) == 4 and i % 2 == 0:
            t = t.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(t.to(device)))
    return ps


def _sched(opt, total):
    return torch.optim.lr_scheduler.OneCycleLR(optimizer=opt, max_lr=1e-3, pct_start=0.4, base_momentum=0.85, max_momentum=0.95,
                                               div_factor=10.0, total_steps=total)


def test_flat_adamw_follows_torch_adamw_with_onecycle():
    from liso_amd.utils.flat_adamw import FlatAdamW

    dev = torch.device("cuda")
    ref_p, my_p = _make(0, "cpu"), _make(0, dev)
    ref = torch.optim.AdamW(ref_p, lr=1e-3, weight_decay=0.01, foreach=False)
    mine = FlatAdamW(my_p, lr=1e-3, weight_decay=0.01)
    assert all(p.data_ptr() >= mine.flat_param.data_ptr() for p in my_p)
    assert all(a.stride() == b.stride() and a.grad.stride() == a.stride() for a, b in zip(my_p, _make(0, dev)))
    rs, ms = _sched(ref, 12), _sched(mine, 12)
    g = torch.Generator().manual_seed(5)
    for step in range(10):
        v0 = my_p[0]._version
        mine.zero_grad()
        for a, b in zip(ref_p, my_p):
            grad = torch.randn(a.shape, generator=g) * (10.0 ** ((step % 3) - 2))
            if step == 4:
                grad[grad.abs() < 0.5 * grad.abs().max()] = 0.0  # exact zeros: the update is -lr * 0 / (sqrt(v) + eps)
            a.grad = grad.clone()
            b.grad.add_(grad.to(dev))  # autograd accumulates into the flat views the same way
        ref.step(), mine.step()
        rs.step(), ms.step()
        assert my_p[0]._version > v0  # the packed-weight cache keys on it
        assert mine.param_groups[0]["lr"] == ref.param_groups[0]["lr"] and mine.param_groups[0]["betas"] == ref.param_groups[0]["betas"]
        for a, b in zip(ref_p, my_p):
            err = (a.detach() - b.detach().cpu()).abs().max() / a.detach().abs().max()
            assert float(err) <= 2e-6, (step, tuple(a.shape), float(err))
    sd = mine.state_dict()
    st = sd["state"][0]
    assert float(st["step"]) == 10.0
    assert torch.allclose(st["exp_avg"].cpu(), ref.state[ref_p[0]]["exp_avg"], rtol=1e-5, atol=1e-9)
    assert torch.allclose(st["exp_avg_sq"].cpu(), ref.state[ref_p[0]]["exp_avg_sq"], rtol=1e-5, atol=1e-12)
    # the gaps between the 16-byte aligned views never move
    used = torch.zeros(mine.numel, dtype=torch.bool, device=dev)
    for p in my_p:
        off = (p.data_ptr() - mine.flat_param.data_ptr()) // 4
        used[off:off + p.numel()] = True
    assert float(mine.flat_param[~used].abs().sum()) == 0.0


def test_flat_adamw_state_dict_round_trip_continues_identically():
    from liso_amd.utils.flat_adamw import FlatAdamW

    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(9)
    grads = [[torch.randn(s, generator=g).to(dev) for s in SHAPES] for _ in range(6)]

    def run(split):
        ps = _make(3, dev)
        opt = FlatAdamW(ps, lr=1e-3, weight_decay=0.01)
        for k in range(6):
            if k == split:  # save, rebuild, load
                sd = opt.state_dict()
                ps2 = [torch.nn.Parameter(p.detach().clone()) for p in ps]
                opt = FlatAdamW(ps2, lr=1e-3, weight_decay=0.01)
                opt.load_state_dict(sd)
                ps = ps2
            opt.zero_grad()
            for p, gr in zip(ps, grads[k]):
                p.grad.add_(gr)
            opt.step()
        return [p.detach().clone() for p in ps]

    a, b = run(None), run(3)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


def test_adamw_rejects_bad_arguments():
    from liso_amd import _lib as L

    x = torch.zeros(64, device="cuda")
    lib = L.lib()
    assert lib.liso_adamw_step_f32(L.ptr(x), L.ptr(x), L.ptr(x), None, 64, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, L.stream_ptr()) == -1
    assert lib.liso_adamw_step_f32(L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), 64, 1e-3, 0.9, 0.999, 1e-8, 0.01, 0, L.stream_ptr()) == -1
    assert lib.liso_adamw_step_f32(L.ptr(x[1:]), L.ptr(x), L.ptr(x), L.ptr(x), 60, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, L.stream_ptr()) == -1
    assert lib.liso_adamw_step_f32(L.ptr(x), L.ptr(x), L.ptr(x), L.ptr(x), 0, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, L.stream_ptr()) == 0


def test_batched_gradient_gather_moves_every_tensor():
    """liso_gather_f32: more tensors than one launch's table (48) of odd sizes -> their destinations inside one flat buffer"""
    import ctypes

    from liso_amd import _lib as L

    torch.manual_seed(2)
    sizes = [1, 3, 64, 1152, 36864, 7, 128] * 9  # 63 tensors
    srcs = [torch.randn(n, device="cuda") for n in sizes]
    flat = torch.zeros(sum(sizes) + 5, device="cuda")
    dsts, off = [], 0
    for n in sizes:
        dsts.append(flat[off:off + n])
        off += n
    n = len(sizes)
    src = (ctypes.c_void_p * n)(*[t.data_ptr() for t in srcs])
    dst = (ctypes.c_void_p * n)(*[t.data_ptr() for t in dsts])
    cnt = (ctypes.c_size_t * n)(*sizes)
    L.check(L.lib().liso_gather_f32(n, src, dst, cnt, L.stream_ptr()), "gather")
    assert torch.equal(flat[:off], torch.cat(srcs)) and float(flat[off:].abs().max()) == 0.0


def test_multi_copy_equals_tensor_copies():
    """_lib.multi_copy: many device-to-device copies in one launch (include/liso_optim.h: liso_multi_copy) -- every dtype the staging
    code moves, odd byte counts, misaligned slices, empty tensors, more segments than one launch takes, and the pairs it must leave to
    copy_ (host sources, dtype changes, strided views)."""
    import torch

    from liso_amd import _lib as L

    g = torch.Generator().manual_seed(0)
    pairs, want = [], []
    for k in range(60):
        dt = [torch.float32, torch.int32, torch.uint8, torch.float64, torch.bool, torch.bfloat16, torch.int64][k % 7]
        n = [0, 1, 3, 17, 255, 4096, 100003][k % 7 if k % 5 else (k // 5) % 7]
        src = (torch.rand(n + 3, generator=g) * 100).to(dt).cuda()
        dst = torch.zeros(n + 5, dtype=dt, device="cuda")
        s_, d_ = src[3:], dst[1:1 + n]  # (misaligned starts)
        pairs.append((d_, s_))
        want.append((dst, torch.cat([dst[:1].clone(), s_.clone(), dst[1 + n:].clone()])))
    host = torch.arange(10, dtype=torch.float32)
    d_host = torch.zeros(10, device="cuda")
    pairs.append((d_host, host))                                   # host source
    d_cast = torch.zeros(6, device="cuda", dtype=torch.float64)
    pairs.append((d_cast, torch.arange(6, device="cuda", dtype=torch.float32)))  # dtype change
    base = torch.zeros(8, 4, device="cuda")
    pairs.append((base[:, 1], torch.arange(8, device="cuda", dtype=torch.float32)))  # strided destination
    L.multi_copy(pairs)
    torch.cuda.synchronize()
    for dst, exp in want:
        assert torch.equal(dst, exp)
    assert torch.equal(d_host.cpu(), host) and torch.equal(d_cast.cpu(), torch.arange(6, dtype=torch.float64))
    assert torch.equal(base[:, 1].cpu(), torch.arange(8, dtype=torch.float32)) and float(base[:, 0].abs().max()) == 0


@pytest.mark.gpu
def test_copy_blocks_places_filter_blocks_like_slice_assignments():
    """_lib.copy_blocks (include/liso_optim.h: liso_multi_copy_rows): blocks of merged filters -- row ranges, column ranges, both,
    1-D biases, more jobs than one launch takes -- land where slice assignments put them and nothing else is touched."""
    import torch

    from liso_amd import _lib as L

    g = torch.Generator().manual_seed(1)
    R = lambda *s: torch.rand(*s, generator=g).cuda()  # noqa: E731
    jobs, checks = [], []
    for k in range(45):
        co, ci, kk = [1, 2, 4, 64, 96][k % 5], [2, 8, 64, 33][k % 4], [1, 3, 7][k % 3]
        big = R(co + 5, ci + 7, kk, kk)
        ref = big.clone()
        blk = R(co, ci, kk, kk)
        jobs.append((big[2:2 + co, 3:3 + ci], blk))
        ref[2:2 + co, 3:3 + ci] = blk
        checks.append((big, ref))
        out = torch.zeros(co, ci, kk, kk, device="cuda")
        other = R(co + 3, ci + 2, kk, kk)
        jobs.append((out, other[1:1 + co, 2:2 + ci]))  # strided source
        checks.append((out, other[1:1 + co, 2:2 + ci].clone()))
    bias = torch.zeros(13, device="cuda")
    b1, b2 = R(5), R(8)
    jobs += [(bias[:5], b1), (bias[5:], b2)]
    L.copy_blocks(jobs)
    torch.cuda.synchronize()
    for (got, exp) in checks:
        assert torch.equal(got, exp)
    assert torch.equal(bias, torch.cat([b1, b2]))
    with pytest.raises(AssertionError):
        L.copy_blocks([(torch.zeros(4, 3, device="cuda")[:, :2], torch.zeros(4, 2, 2, device="cuda")[:, :, 0])])  # inner stride 2


@pytest.mark.gpu
def test_flat_rmsprop_follows_torch_rmsprop():
    """FlatRMSprop (one liso_rmsprop_step_f32 launch over flat buffers) against torch.optim.RMSprop with the reference's settings
    (liso/slim/experiment.py:200-219: defaults + a LambdaLR schedule): 20 steps on tensors of odd sizes and channels-last filters,
    gradients arriving as `.grad` views, as tensors of their own (zero_grad(set_to_none=True)) and not at all for one parameter"""
    from liso_amd.utils.flat_adamw import FlatRMSprop

    g = torch.Generator().manual_seed(3)
    shapes = [(64, 2, 7, 7), (96,), (33, 5), (192, 304, 3, 3), (7,)]
    base = [torch.randn(s, generator=g) for s in shapes]
    pa = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    pb = [torch.nn.Parameter(t.clone().cuda()) for t in base]
    pb[3].data = pb[3].data.contiguous(memory_format=torch.channels_last)
    oa = torch.optim.RMSprop(pa, lr=1e-2)
    ob = FlatRMSprop(pb, lr=1e-2)
    sched = lambda o: torch.optim.lr_scheduler.LambdaLR(o, lambda k: 1.0 / (1 + 0.1 * k))  # noqa: E731
    sa, sb = sched(oa), sched(ob)
    for step in range(20):
        grads = [torch.randn(s, generator=g).cuda() * (10.0 ** (step % 3 - 1)) for s in shapes]
        oa.zero_grad(set_to_none=True)
        ob.zero_grad(set_to_none=step % 2 == 0)
        for k, (a, b, gr) in enumerate(zip(pa, pb, grads)):
            if k == 4 and step < 10:
                continue  # (no gradient for this parameter during the first steps)
            a.grad = gr.clone()
            if step % 2 == 0:
                b.grad = gr.clone()
            else:
                b.grad.copy_(gr)
        oa.step(), ob.step()
        sa.step(), sb.step()
    for k, (a, b) in enumerate(zip(pa, pb)):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (k, float((a - b).abs().max()))
    sd = ob.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "square_avg"} and float(sd["state"][0]["step"]) == 20.0
    assert torch.allclose(oa.state_dict()["state"][0]["square_avg"], sd["state"][0]["square_avg"], rtol=2e-6, atol=1e-12)
