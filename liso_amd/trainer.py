"""Training steps of the hot path.

DetectorTrainer.step mirrors the per-iteration body of liso/kabsch/liso_cli.py:362-618 for the CenterPoint-pillar
network (forward -> centerpoint_loss x cm_loss_weight + rotation regulariser -> backward -> AdamW -> OneCycleLR,
optimizer factory liso_cli.py:792-823).  The reference is single-GPU; data parallelism (one process per GPU,
gradient all-reduce over RCCL/xGMI in a single flat bucket overlapped with backward, per-rank BatchNorm) is the
only addition.
"""
import contextlib
import os

import torch
import torch.distributed as dist

from liso_amd import _lib as L

from liso_amd.losses.centerpoint_loss import centerpoint_loss, rotation_vec_on_unit_circle
from liso_amd.networks.simple_net.simple_net import BoxLearner


def get_optimizer_scheduler(cfg, box_predictor, total_steps=None, flat=False):
    """liso_cli.py:792-823: AdamW(lr, weight_decay 0.01) + OneCycleLR(pct_start 0.4, momentum 0.85-0.95, div 10); training on
    ground truth runs one cycle over `num_training_steps` (+2, as in the reference), training on mined boxes one cycle per
    weight-drop period with final_div_factor 10.  `total_steps` overrides the cycle length (benchmarks)."""
    if flat:  # the same update as one launch over flat buffers (liso_amd/utils/flat_adamw.py, include/liso_optim.h)
        from liso_amd.utils.flat_adamw import FlatAdamW

        opt = FlatAdamW(box_predictor.parameters(), lr=cfg.optimization.learning_rate, weight_decay=0.01)
    else:
        opt = torch.optim.AdamW(box_predictor.parameters(), lr=cfg.optimization.learning_rate, weight_decay=0.01)
    common = dict(optimizer=opt, max_lr=cfg.optimization.learning_rate, pct_start=0.4, base_momentum=0.85, max_momentum=0.95,
                  div_factor=10.0)
    source = cfg.data.setdefault("train_on_box_source", "gt")
    if source == "gt":
        sched = torch.optim.lr_scheduler.OneCycleLR(total_steps=(total_steps or cfg.optimization.num_training_steps) + 2, **common)
    elif source == "mined":
        assert cfg.optimization.rounds.active, "assuming this"
        rounds = cfg.optimization.rounds
        sched = torch.optim.lr_scheduler.OneCycleLR(
            total_steps=(total_steps or rounds.steps_per_round * rounds.drop_net_weights_every_nth_round) + 2, final_div_factor=10,
            **common)
    else:
        raise NotImplementedError(source)
    return opt, sched


def get_slim_optimizer_scheduler(slim_cfg, params, flat=False):
    """slim/experiment.py:200-219: RMSprop(lr = initial) (or Adam) + linear warm-up over `warm_up.step_length` steps, then
    linear decay to 5 % of the initial rate at `iterations.train`.  `flat` (GPU): RMSprop over flat buffers, one update launch
    (liso_amd/utils/flat_adamw.py: FlatRMSprop)."""
    from liso_amd.utils.learning_rate import get_polynomial_decay_schedule_with_warmup

    if slim_cfg.optimizer == "rmsprop" and flat:
        from liso_amd.utils.flat_adamw import FlatRMSprop

        opt = FlatRMSprop(list(params), lr=slim_cfg.learning_rate.initial)
    elif slim_cfg.optimizer == "rmsprop":
        opt = torch.optim.RMSprop(params, lr=slim_cfg.learning_rate.initial)
    elif slim_cfg.optimizer == "adam":
        opt = torch.optim.Adam(params, lr=slim_cfg.learning_rate.initial)
    else:
        raise AssertionError("only rmsprop/adam supported")
    sched = get_polynomial_decay_schedule_with_warmup(
        optimizer=opt, num_warmup_steps=slim_cfg.learning_rate.warm_up.step_length,
        num_training_steps=slim_cfg.iterations.train, lr_end=slim_cfg.learning_rate.initial * 0.05)
    return opt, sched


def gather_gradients(device, params, add=False):
    """`params`: (parameter, its flat-buffer gradient view) whose `.grad` was None during the backward pass.  Their gradient
    tensors -> the flat views by one launch per 48 tensors (liso_gather_f32); `.grad` is the flat view again afterwards.  `add`: the
    second half of a split backward pass adds (nothing arrives there for these parameters today).  -> the source tensors (keep them
    alive until the launch has run: inside a capture they belong to the graph's pool anyway)."""
    from liso_amd import _lib as L
    import ctypes

    jobs = []
    for p_, flat in params:
        g = p_.grad
        p_.grad = flat
        if g is None or g is flat:
            continue
        if add or g.dtype != torch.float32 or g.stride() != flat.stride() or g.shape != flat.shape:
            flat.add_(g) if add else flat.copy_(g)
            continue
        jobs.append((g, flat))
    if not jobs:
        return []
    n = len(jobs)
    src = (ctypes.c_void_p * n)(*[g.data_ptr() for g, _ in jobs])
    dst = (ctypes.c_void_p * n)(*[f.data_ptr() for _, f in jobs])
    cnt = (ctypes.c_size_t * n)(*[g.numel() for g, _ in jobs])
    with torch.cuda.device(device):
        L.check(L.lib().liso_gather_f32(n, src, dst, cnt, L.stream_ptr()), "gather_f32")
    return [g for g, _ in jobs]


_SIDE_STREAMS = {}


def side_stream(device, role, priority=0):
    """The process's side stream for `role` on `device`: ONE HIP stream per (device, role), shared by every trainer of the process.

    HIP binds a stream to one of a few hardware queues when it is first used, in order of first use.  A second LisoLoopTrainer that
    CREATED three more streams next to a first, already stepped one found its streams on queues that its own caller's stream or each
    other already occupied: its three pipeline stages took turns on a queue and its step ran at 11.3 ms instead of 4.37 (the first
    trainer, stepped again afterwards: 4.37; either trainer without overlap: 9.1-9.4; the second with its side streams replaced by
    fresh high-priority ones: 4.39 -- scripts/second_trainer_bisect.py, round 6; rounds 4-5 had seen it as "a trainer built second
    is 1.5x slower, cause not found" and moved bench.py's parity legs into child processes).  Streams are queues, not state: trainers
    of one process share them, an idle trainer leaves nothing on them, two active ones interleave on them."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), role)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=priority)
    return st


class DetectorTrainer:
    def __init__(self, cfg, device, compute_dtype=torch.float32, total_steps=None, fused_loss=None, use_graph=False, exact=None,
                 grad_buckets=None):
        """`grad_buckets` (graph path, several ranks): 2 (default) = the step is captured as TWO graphs, cut behind the backbone's first
        block (mfma_conv.GradCut): the gradients of everything downstream of the cut -- 97 % of the parameters -- are all-reduced
        between the two replays and the collective overlaps block 0's backward pass (the layers at the largest resolution), the
        pillar encoder's backward and the second, small all-reduce; 1 = one graph, one all-reduce behind it.  (ROCm refuses external
        events inside a captured graph -- "External events are disallowed in rocm" -- so the signal cannot come from inside ONE
        graph.)  A single rank always runs one graph.
        `exact` (with compute_dtype float32): True = fp32 convolutions on the native fp32 MFMA (2^-24 per product, the reference's
        fp32 semantics; the parity configuration), False = as bf16 hi/lo pairs (F32X3, 2^-16 per product), None = leave the
        process-wide setting (liso_amd.utils.mfma_conv.set_fp32_mode) as it is.
        `fused_loss`: activations + decode + CenterPoint loss in one HIP pass (include/liso_detector.h) instead of
        ~140 torch launches; default = whenever the configuration is the overlay the kernel implements and the
        network runs on the GPU.
        `use_graph`: forward + loss + backward of one step (no device->host sync on that path) are captured once per input
        shape into a hipGraph and replayed; new clouds / targets are copied into the captured input buffers.  Gradients live
        in one flat buffer: data parallelism is ONE RCCL all-reduce of that buffer after the replay (no DDP wrapper), then
        the eager AdamW / OneCycleLR step."""
        from liso_amd.losses import fused_centerpoint

        self.cfg, self.device = cfg, device
        if device.type == "cuda":  # (persistent device flag of the sparse canvas convolutions: must exist before the first capture)
            from liso_amd.utils import mfma_conv as _MC
            _MC._sparse_flag(device)
        self.use_graph = bool(use_graph) and device.type == "cuda"
        self._graph, self._graph2, self._graph_sig, self._capture_stream = None, None, None, None
        self._wgrad_stream = None
        self.fused_loss = (fused_centerpoint.supports(cfg) and device.type == "cuda") if fused_loss is None else fused_loss
        self.net = BoxLearner(cfg).to(device)
        self.net.model.set_compute_dtype(compute_dtype)
        from liso_amd.utils import mfma_conv as MC
        if exact is not None:
            MC.set_fp32_mode("exact" if exact else "x3")
        if self.use_graph and not self.fused_loss:
            # only the fused loss keeps the captured region free of hipMemsetAsync nodes (graph_safety.py): torch's multi-block
            # reductions (centerpoint_loss) put them into the graph
            from liso_amd.utils.graph_safety import require_node_replay

            require_node_replay("DetectorTrainer(use_graph=True) without the fused loss")
        # (the filters stay contiguous fp32 master weights: the kernels pack their panels from them; channels-last parameters would
        # cost a layout copy per pack and strided gradient accumulations -- ~170 extra launches per step, measured)
        self.model = self.net
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        # on the GPU the parameters, their gradients and the AdamW moments are views into four flat buffers: the update is one
        # launch, the gradient all-reduce one collective, zero_grad one memset
        self.optimizer, self.lr_scheduler = get_optimizer_scheduler(cfg, self.net, total_steps, flat=device.type == "cuda")
        if self.world > 1 and not self.use_graph:
            # ~19 MB of fp32 gradients: one flat bucket, launched as backward reaches the first layer's grads
            self.model = torch.nn.parallel.DistributedDataParallel(
                self.net, device_ids=[device.index] if device.type == "cuda" else None, bucket_cap_mb=64,
                broadcast_buffers=False, gradient_as_bucket_view=device.type != "cuda")
        self.n_grad_buckets = 1
        if self.use_graph:
            self._flat_grad = self.optimizer.flat_grad  # data parallelism: SUM all-reduce of this buffer, 1 / world inside the AdamW launch
            if self.world > 1:  # replicas start identical (what the DDP constructor would do)
                for t in list(self.net.parameters()) + list(self.net.buffers()):
                    dist.broadcast(t.data, src=0)
                self.optimizer.grad_scale = 1.0 / self.world
                self._setup_gradient_buckets(grad_buckets)
            elif grad_buckets == 2:  # (one rank, asked for explicitly: the two-graph step without a collective -- tests)
                self._setup_gradient_buckets(2)
        from liso_amd.networks.centerpoint.fused_bn import defer_batch_counters
        self._bn_counters = defer_batch_counters(self.net.model.rpn) + defer_batch_counters(self.net.model.center_head)

    def _setup_gradient_buckets(self, grad_buckets):
        """two gradient buckets: [pillar encoder | backbone block 0] and [everything else] -- the second one is complete when the
        backward pass arrives at the output of block 0 (rpn.py: grad_cut), with block 0's backward (the layers at the largest
        resolution) and the pillar encoder's still to come"""
        from liso_amd.utils import mfma_conv as MC

        rpn = self.net.model.rpn
        if grad_buckets is None:
            grad_buckets = int(os.environ.get("LISO_GRAD_BUCKETS", "2"))
        if grad_buckets < 2 or len(rpn.blocks) < 2 or not hasattr(self.optimizer, "offsets"):
            return
        first = next(rpn.blocks[1].parameters())
        split = self.optimizer.offsets[id(first)]
        # (the flat buffer follows net.parameters(): pillar encoder, blocks 0.., deblocks, head)
        order = [id(p_) for p_ in self.net.parameters() if p_.requires_grad]
        assert all(self.optimizer.offsets[i] < split for i in order[:order.index(id(first))])
        assert all(self.optimizer.offsets[i] >= split for i in order[order.index(id(first)):])
        self._bucket_split = int(split)
        self._grad_cut = MC.GradCut()
        self.n_grad_buckets = 2

    def _reduce_gradients(self, rest_of_backward):
        """SUM all-reduce of the flat gradient buffer over the ranks (the mean's 1 / world is applied inside the AdamW launch).
        Called behind the replay of the (first) graph; `rest_of_backward()` = what is still to run: the second graph (two buckets)
        and the pillar encoder's eager backward."""
        if getattr(self, "skip_collective", False):  # bench.py's measurement of the collective's exposed cost (replicas diverge)
            rest_of_backward()
            return
        if self.n_grad_buckets < 2:
            rest_of_backward()
            dist.all_reduce(self._flat_grad)
            return
        # (the collective runs on the process group's own stream, which waits for the caller's stream as it stands NOW: behind the
        # first graph; the caller's stream goes on with the rest of the backward pass and waits for both collectives at the end)
        w1 = dist.all_reduce(self._flat_grad[self._bucket_split:], async_op=True)
        rest_of_backward()
        w2 = dist.all_reduce(self._flat_grad[:self._bucket_split], async_op=True)
        w1.wait()
        w2.wait()

    def loss(self, pcls, targets, canvas=None):
        """liso_cli.py:452-614.  `canvas`: precomputed pillar canvas (bev, occupancy) -- the hipGraph path"""
        if self.model.training:
            from liso_amd.networks.centerpoint.fused_bn import step_batch_counters
            step_batch_counters(self._bn_counters)
        cfg = self.cfg
        sup = cfg.loss.supervised.supervised_on_clusters
        gt_maps = {a: targets[a] for a in sup.attrs}
        mask = targets["center_bool_mask"]
        if self.fused_loss:
            from liso_amd.losses.fused_centerpoint import fused_centerpoint_loss

            _, _, raw, _ = self.model(None, pcls, None, centermaps_gt=None, decode=False, canvas=canvas)
            total, losses = fused_centerpoint_loss(
                cfg=cfg, raw_box_maps=raw, gt_maps=gt_maps, gt_center_mask=mask,
                ignore_region_is_true_mask=targets.get("ignore_region_is_true_mask", None),
                pillar_center_coors_m=self.net.pillar_center_coors_m)
            return total, losses, None
        pred_boxes, decoded, activated, aux = self.model(None, pcls, None, centermaps_gt=None)
        ignore = targets.get("ignore_region_is_true_mask", torch.zeros_like(mask))
        losses = centerpoint_loss(loss_cfg=cfg.loss, raw_activated_pred_box_maps=activated, decoded_pred_box_maps=decoded,
                                  gt_maps=gt_maps, gt_center_mask=mask,
                                  rotation_loss_weights_map=torch.ones_like(gt_maps["probs"]),
                                  box_prediction_cfg=cfg.box_prediction, ignore_region_is_true_mask=ignore)
        total = 0.0
        for v in losses.values():
            total = total + sup.weight * v
        rr = cfg.box_prediction.rotation_representation
        if rr.method == "vector":  # main_utils.py:119-134
            total = total + rotation_vec_on_unit_circle(activated) * rr.regul_weight
        return total, losses, pred_boxes

    def step(self, pcls, targets, prep=None):
        """`prep` (extension): `self.net.model.pfn.prepare(pcls)` issued earlier (LisoLoopTrainer: on another stream, next to the previous
        step) -- the weight-independent half of the pillar encoder"""
        self.model.train()
        if self.use_graph:
            return self._graph_step(pcls, targets, prep)
        self.optimizer.zero_grad(set_to_none=True)
        total, losses, _ = self.loss(pcls, targets)
        total.backward()
        self.optimizer.step()
        self.lr_scheduler.step()
        return total.detach()

    def eager_pass(self, pcls, targets):
        """forward + loss + backward with eager launches and NO update: no collective, no optimizer / scheduler step (bench.py
        times individual kernels this way next to graph-replayed steps; every rank may call it independently)"""
        self.model.train()
        saved = [p.grad for p in self.net.parameters()]
        for p in self.net.parameters():
            p.grad = None
        bufs = {k: v.clone() for k, v in self.net.state_dict().items() if "running" in k or "num_batches" in k}
        ctx = self.model.no_sync() if hasattr(self.model, "no_sync") else contextlib.nullcontext()
        with ctx:
            total, _, _ = self.loss(pcls, targets)
            total.backward()
        with torch.no_grad():
            for k, v in self.net.state_dict().items():
                if k in bufs:
                    v.copy_(bufs[k])
        for p, g in zip(self.net.parameters(), saved):
            p.grad = g
        return total.detach()

    # ---- hipGraph path ----------------------------------------------------------------------------------------------
    # The graph holds backbone + head + fused loss, forward and backward (~95 % of the step's launches).  The pillar encoder
    # (voxelise + PFN + scatter; 10 launches) runs eagerly around it.  Rounds 2-4: replaying ITS launches from a graph while other
    # pillar-encoder calls ran eagerly in the same process (the LISO loop's SLIM inference) ended in a GPU memory fault, bisected to
    # exactly that combination (scripts/try_loop_graph3.py PART=pfn).  Cause (round 5): rocPRIM's radix sort inside the voxeliser --
    # the same library path whose captured launches fault for > 1 M keys (see _capture_slim below) and whose memset nodes do not
    # survive replays (utils/graph_safety.py).  The voxeliser no longer sorts (csrc/pillars.hip: per-cell segments + arrival rank):
    # the reproducer replays cleanly, and the guard-band runs of tests/test_gpu_canaries.py find no out-of-bounds write in any kernel
    # of the encoder, at and beyond its capacities.  The encoder still runs eagerly for a different reason: its launches carry the
    # raw clouds' lengths (host offsets) as kernel arguments, which a captured graph would freeze.
    def _pillars(self, pcls, out=None, prep=None):
        """`out`: (canvas rows [B, gx, gy, 64], occupancy) to write into -- the graph's static inputs (no copy afterwards)"""
        bev, occ = self.net.model.pfn(pcl_t0=pcls, img_t0=None, out=out, prep=prep)
        return bev, occ

    def _capture(self, pcls, targets):
        dev = self.device
        # snapshot FIRST: the shape-probing pillar pass below runs the PFN BatchNorm in training mode and must not count as a batch
        buffers = {k: v.clone() for k, v in self.net.state_dict().items() if v.is_floating_point() or v.dtype == torch.long}
        with torch.no_grad():
            bev, occ = self._pillars(pcls)
        self._static_bev = bev.detach().clone().requires_grad_(True)
        self._static_occ = occ.detach().clone()
        self._static_bev.grad = None
        self._static_targets = {k: v.to(dev).clone() for k, v in targets.items()}
        quiet = hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch")
        if quiet:  # the flat gradient views are created on the default stream, warm-up and capture run on a side stream
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        if self._capture_stream is None:
            self._capture_stream = side_stream(dev, "flow")  # (the loop trainer lends the same one: SLIM inference's stream)
        # LISO_WGRAD_SIDE=1 (opt-in, measured slower): the weight gradients as a parallel branch of the captured backward pass
        # (mfma_conv.wgrad_side) -- nothing in the backward chain reads them.  Results identical; but every fork edge of a replayed
        # hipGraph costs ~240 us here: detector replay 7.15 vs 2.54 ms (19 forks + 1 join), loop 6.44 vs 4.38 ms per step.
        if self._wgrad_stream is None and os.environ.get("LISO_WGRAD_SIDE", "0") == "1":
            self._wgrad_stream = side_stream(dev, "wgrad")
        side = self._capture_stream  # (kept alive with the graph)
        side.wait_stream(torch.cuda.current_stream(dev))

        from liso_amd.utils import mfma_conv as MC

        self._pack_jobs = None
        self._gather_params = None

        cut = self._grad_cut if self.n_grad_buckets == 2 else None
        rpn = self.net.model.rpn

        def body(part=None):
            """part None: the whole step; 1: forward + loss + backward down to the cut; 2: the backward pass above the cut"""
            if part in (None, 1):
                self._flat_grad.zero_()
                # (no zeroed gradient buffer for the canvas: with .grad = None autograd keeps the first layer's data gradient as the
                # leaf's gradient -- a 134-MB fill and a 400-MB accumulation into zeros per step at B = 4 otherwise)
                self._static_bev.grad = None
                if self._pack_jobs:  # the forward / data-gradient panels of every layer from ONE launch (recorded in the warm-up)
                    self._step_packs = MC.batched_pack(self._pack_jobs)
            MC.set_step_packs(getattr(self, "_step_packs", None))
            det_cus = MC.roles_cus(getattr(self, "roles_cus", 0))  # (experiment knob of the pipelined loop: LisoLoopTrainer.detector_cus)
            det_cus.__enter__()
            MC.set_direct_grads(True, keep_touched=part == 2)  # gradients of conv / BatchNorm parameters land in the flat buffer without an add each
            gathered = self._gather_params or []
            for p_, _ in gathered:  # (autograd then KEEPS the gradient tensor it is handed instead of adding it into the zeroed slice)
                p_.grad = None
            try:
                if part in (None, 1):
                    rpn.grad_cut = cut
                    try:
                        total, _, _ = self.loss(None, self._static_targets, canvas=(self._static_bev, self._static_occ))
                    finally:
                        rpn.grad_cut = None
                    with MC.wgrad_side(self._wgrad_stream):
                        total.backward()
                    self._body_loss = total.detach()
                if cut is not None and part in (None, 2):
                    with MC.wgrad_side(self._wgrad_stream):
                        cut.finish()
            finally:
                det_cus.__exit__(None, None, None)
                MC.set_step_packs(None)
                MC.set_direct_grads(False, keep_touched=True)
                self._gather_gradients(gathered, add=part == 2)
            return self._body_loss

        self._gather_params = None
        with torch.cuda.stream(side):  # warm-up off the capture: lazy initialisations, allocator pools
            MC.record_pack_jobs(True)
            body()
            self._pack_jobs = MC.record_pack_jobs(False)
            if os.environ.get("LISO_GATHER_GRADS", "1") != "0":
                # parameters whose gradient no kernel wrote in place during that pass (merged head convolutions, sliced BatchNorm
                # vectors, block-diagonal filters): autograd would launch one `add_` each into the zeroed flat buffer -- instead
                # their gradient tensors are collected behind the backward pass by ONE launch (liso_gather_f32)
                in_place = MC.direct_touched()
                outside = {id(p_) for p_ in self.net.model.pfn.parameters()}  # (the pillar encoder's backward runs outside the graph)
                self._gather_params = [(p_, p_.grad) for p_ in self.net.parameters()
                                       if p_.requires_grad and p_.grad is not None and id(p_) not in in_place and id(p_) not in outside]
            body()
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():  # the warm-up passes must not count as training steps (BatchNorm statistics / counters)
            for k, v in self.net.state_dict().items():
                if k in buffers:
                    v.copy_(buffers[k])
        self._graph, self._graph2 = torch.cuda.CUDAGraph(), None
        # capture on the warm-up's stream: the AccumulateGrad nodes of the parameters were created there; a capture on another
        # stream forks into it and the replay computes garbage (measured: loss 63 instead of 1082)
        if cut is None:
            with torch.cuda.graph(self._graph, stream=side):
                self._static_loss = body()
        else:  # two graphs sharing one memory pool: the second one reads the leaf gradient and the saved tensors of the first
            with torch.cuda.graph(self._graph, stream=side):
                self._static_loss = body(1)
            self._graph2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph2, stream=side, pool=self._graph.pool()):
                body(2)
        with torch.no_grad():
            for k, v in self.net.state_dict().items():
                if k in buffers:
                    v.copy_(buffers[k])
        if quiet:
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(True)

    def _gather_gradients(self, params, add=False):
        self._gather_keepalive = gather_gradients(self.device, params, add)

    def _graph_step(self, pcls, targets, prep=None):
        # what the graph consumes: the [B, 64, gx, gy] canvas (batch size; grid and dtype are fixed per trainer) and the target maps.
        # The clouds themselves never enter it (the pillar encoder runs eagerly in front): their point counts are not part of the key.
        tsig = targets.shapes() if isinstance(targets, _BatchedTargets) else tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(targets.items()))
        sig = (len(pcls), tsig)
        if self._graph is None or sig != self._graph_sig:
            self._graph = None
            self._capture(pcls, targets)
            self._graph_sig = sig
        # the eager pillar encoder writes straight into the graph's input buffers (its own backward runs on saved feature rows)
        bev, occ = self._pillars(pcls, out=(self._static_bev.detach().permute(0, 2, 3, 1), self._static_occ), prep=prep)
        with torch.no_grad():
            stage = []  # (one launch for all target tensors: _lib.multi_copy)
            if isinstance(targets, _BatchedTargets):
                off = {}
                for t in targets.per_sample:  # (running offset per key: samples may carry more than one row)
                    for k, v in t.items():
                        o = off.get(k, 0)
                        stage.append((self._static_targets[k][o:o + v.shape[0]], v))
                        off[k] = o + v.shape[0]
            else:
                for k, v in targets.items():
                    stage.append((self._static_targets[k], v))
            L.multi_copy(stage)
        self._graph.replay()

        def rest_of_backward():
            if self._graph2 is not None:  # (two gradient buckets: the backward pass above the cut)
                self._graph2.replay()
            if bev.requires_grad:  # the pillar encoder's own parameters: its backward runs eagerly on the replayed d loss / d canvas
                bev.backward(self._static_bev.grad)

        if self.world > 1:
            self._reduce_gradients(rest_of_backward)
        else:
            rest_of_backward()
        self.optimizer.step()
        self.lr_scheduler.step()
        return self._static_loss.clone()


class SlimTrainer:
    """SLIM self-supervised train step, mirror of liso/slim/experiment.py:834-919 (`train_one_step`) with the optimizer /
    schedule factory of :200-219: RMSprop(lr 1e-4) + linear warm-up (2000) then linear decay to 5 %; the loss is the
    un-weighted sum over the 6 RAFT iterations.  The reference cloud of every kNN query is bucketed on the device once
    per step (the reference rebuilds a host KD-tree for each of the 12+ queries).

    `use_graph=True`: forward + loss + backward of one step (≈7 800 kernel launches, no device->host sync) are captured
    once per input shape into a hipGraph and replayed; new sweeps are copied into the captured input buffers.  Gradients
    live in one flat buffer: data parallelism is one RCCL all-reduce of that buffer after the replay (no DDP wrapper),
    then the eager RMSprop step."""

    def __init__(self, cfg, device, num_train_samples=1000, use_graph=False, channels_last=False, exact=None):
        """`exact`: True = every fp32 convolution on the native fp32 MFMA (parity configuration), False = bf16 hi/lo pairs (F32X3,
        the production default), None = leave the process-wide setting (mfma_conv.set_fp32_mode)"""
        from liso_amd.slim.model.slim import SLIM
        if exact is not None:
            from liso_amd.utils import mfma_conv as MC
            MC.set_fp32_mode("exact" if exact else "x3")
        self.cfg, self.slim_cfg, self.device = cfg, cfg.SLIM, device
        if device.type == "cuda":  # (persistent device flag of the sparse canvas convolutions: must exist before the first capture)
            from liso_amd.utils import mfma_conv as _MC
            _MC._sparse_flag(device)
        self.net = SLIM(cfg, num_train_samples=num_train_samples).to(device)
        if channels_last:  # measured slower than NCHW filters on gfx950 (65 vs 59 ms per step): MIOpen's fp32 Winograd is NCHW
            # the pillar canvas is channels-last storage; keep filters in the same layout so MIOpen's NHWC kernels run
            # without a transpose before and after every convolution
            self.net.raft_network.to(memory_format=torch.channels_last)
        self.model = self.net
        self.use_graph = bool(use_graph) and device.type == "cuda"
        self.world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if self.world > 1 and not self.use_graph:
            self.model = torch.nn.parallel.DistributedDataParallel(
                self.net, device_ids=[device.index] if device.type == "cuda" else None, bucket_cap_mb=64,
                broadcast_buffers=False, gradient_as_bucket_view=True)
        # (flat buffers unless DistributedDataParallel owns the gradients as views of ITS buckets)
        flat = device.type == "cuda" and self.model is self.net and os.environ.get("LISO_FLAT_RMSPROP", "1") != "0"
        self.optimizer, self.lr_scheduler = get_slim_optimizer_scheduler(self.slim_cfg, self.net.parameters(), flat=flat)
        self._flat_opt = hasattr(self.optimizer, "flat_grad")
        import numpy as np
        half = 0.5 * np.array(cfg.data.bev_range_m, dtype=np.float32)
        self.bev_extent = np.concatenate([-half, half], axis=0)
        self._graph, self._graph_sig = None, None
        if self.use_graph:
            from liso_amd.utils.graph_safety import require_node_replay

            require_node_replay("SlimTrainer(use_graph=True)")
            # once-per-step weight gradients of the update block (deferred_wgrad.py): inside a captured graph the gate adds them
            # into the flat gradient buffer itself ("direct"); returned to autograd they would go through AccumulateGrad nodes
            # bound to the warm-up's stream -- a cross-stream hop the capture cannot contain (wrong losses / crashes, measured)
            self.net.raft_network.defer_update_block_wgrad = "direct"
            params = [p for p in self.net.parameters() if p.requires_grad]
            if self._flat_opt:  # the optimizer's own flat gradient buffer (parameters, gradients, square averages: one element order)
                self._flat_grad = self.optimizer.flat_grad
            else:
                self._flat_grad = torch.zeros(sum(p.numel() for p in params), dtype=torch.float32, device=device)
                off = 0
                for p in params:  # gradients are views into one flat buffer: one memset, one all-reduce
                    p.grad = self._flat_grad[off:off + p.numel()].view_as(p)
                    off += p.numel()
            # parameters whose gradients are produced INSIDE the captured step (everything but the pillar encoder, whose backward
            # runs eagerly behind the replay): with .grad = None during the captured backward pass autograd keeps the gradient
            # tensors instead of launching one add_ per parameter into the zeroed flat buffer (~100 launches per step), and one
            # gather launch per 48 tensors moves them into their slices (DetectorTrainer does the same for its few such parameters)
            outside = {id(p) for p in self.net.raft_network.pp_layer.parameters()} if hasattr(self.net.raft_network, "pp_layer") else set()
            self._gather_params = [(p, p.grad) for p in params if id(p) not in outside] \
                if os.environ.get("LISO_GATHER_GRADS", "1") != "0" else []
            if self.world > 1:  # replicas start identical (what the DDP constructor would do)
                for t in list(self.net.parameters()) + list(self.net.buffers()):
                    dist.broadcast(t.data, src=0)

    def _inputs(self, sample_t0, sample_t1):
        pc1, m1 = sample_t0["pcl_ta"]["pcl"].to(self.device), sample_t0["pcl_ta"]["pcl_is_valid"].to(self.device)
        pc2, m2 = sample_t1["pcl_ta"]["pcl"].to(self.device), sample_t1["pcl_ta"]["pcl_is_valid"].to(self.device)
        return pc1, m1, pc2, m2

    def loss(self, sample_t0, sample_t1, all_valid=None, canvases=None, gather_plan=None):
        """`all_valid` = (bool, bool): whether every row of the two loss clouds is a real point; evaluated here (the step's
        only device->host sync, before any work is queued) unless the caller already knows.  `canvases`: precomputed pillar
        canvases of both sweeps (the hipGraph path keeps the pillar encoder outside of the graph)."""
        from liso_amd.slim.slim_loss.knn_graph import KnnIndex
        from liso_amd.slim.slim_loss.slim_loss_adaptor import selfsupervisedSlimSingleScaleLoss

        pc1, m1, pc2, m2 = self._inputs(sample_t0, sample_t1)
        if all_valid is None:
            all_valid = (bool(m1.all()), bool(m2.all()))
        ext = [float(v) for v in self.bev_extent]
        # bucket both clouds before the network runs
        idx1 = [KnnIndex(pc1[b][:, :3], extent=ext, all_rows_finite=True) for b in range(pc1.shape[0])] if all_valid[0] else None
        idx2 = [KnnIndex(pc2[b][:, :3], extent=ext, all_rows_finite=True) for b in range(pc2.shape[0])] if all_valid[1] else None
        if canvases is None:
            preds_fw, preds_bw = self.model(sample_t0, sample_t1, None)
        else:
            preds_fw, preds_bw = self.model(sample_t0, sample_t1, None, canvases=canvases, gather_plan=gather_plan)
        kw = dict(moving_thresh_module=self.net.moving_dynamicness_threshold, loss_cfg=self.slim_cfg.losses.unsupervised,
                  model_cfg=self.slim_cfg.model, bev_extent=self.bev_extent, metrics_collector={})
        stacked = getattr(self.net, "stacked_predictions", None)
        if stacked is not None and not self.slim_cfg.model.use_static_aggr_flow_for_aggr_flow:
            # experiment.py:834-919 sums the loss over the RAFT iterations.  Every term of it is a mean over the points /
            # samples of one iteration, so with the iterations stacked along the batch axis (same clouds, same masks) ONE
            # evaluation gives exactly sum_i loss_i / n_it.  (Not used when the loss also updates the dynamicness
            # threshold: that update is sequential in the iterations.)
            sfw, sbw, n_it = stacked
            rep = lambda t: t.repeat(n_it, *([1] * (t.dim() - 1)))  # noqa: E731
            total = n_it * selfsupervisedSlimSingleScaleLoss(
                pc1=rep(pc1), valid_mask_pc1=rep(m1), pc2=rep(pc2), valid_mask_pc2=rep(m2), pred_fw=sfw, pred_bw=sbw,
                knn_index_pc1=None if idx1 is None else idx1 * n_it, knn_index_pc2=None if idx2 is None else idx2 * n_it, **kw)
            return total, preds_fw, preds_bw
        total = torch.zeros(1, device=self.device)
        for pfw, pbw in zip(preds_fw, preds_bw):
            total = total + selfsupervisedSlimSingleScaleLoss(pc1=pc1, valid_mask_pc1=m1, pc2=pc2, valid_mask_pc2=m2, pred_fw=pfw,
                                                              pred_bw=pbw, knn_index_pc1=idx1, knn_index_pc2=idx2, **kw)
        return total, preds_fw, preds_bw

    def step(self, sample_t0, sample_t1, eager=False, update=True):
        """`update=False` (with `eager=True`): forward + loss + backward only -- no collective, no optimizer / scheduler step
        (bench.py times individual kernels this way after a graph-replayed region; every rank may call it independently)."""
        self.model.train()
        if self.use_graph and not eager:
            return self._graph_step(sample_t0, sample_t1)
        total, _, _ = self.loss(sample_t0, sample_t1)
        if self.use_graph:
            self._flat_grad.zero_()
        else:
            self.optimizer.zero_grad(set_to_none=True)
        if not update:
            with self.model.no_sync() if hasattr(self.model, "no_sync") else contextlib.nullcontext():
                total.backward()
            return total.detach()
        total.backward()
        self._reduce_and_update()
        return total.detach()

    def _reduce_and_update(self):
        if self.use_graph and self.world > 1:
            dist.all_reduce(self._flat_grad)
            self._flat_grad.div_(self.world)
        self.optimizer.step()
        self.lr_scheduler.step()

    # ---- hipGraph path ----------------------------------------------------------------------------------------------
    @staticmethod
    def _map_tensors(obj, fn):
        if torch.is_tensor(obj):
            return fn(obj)
        if isinstance(obj, dict):
            return {k: SlimTrainer._map_tensors(v, fn) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(SlimTrainer._map_tensors(v, fn) for v in obj)
        return obj

    @staticmethod
    def _copy_tensors(dst, src, _acc=None):
        """dst <- src over a tree of tensors: the device-to-device ones as ONE launch (_lib.multi_copy)"""
        top = _acc is None
        acc = [] if top else _acc
        if torch.is_tensor(dst):
            acc.append((dst, src))
        elif isinstance(dst, dict):
            for k in dst:
                SlimTrainer._copy_tensors(dst[k], src[k], acc)
        elif isinstance(dst, (list, tuple)):
            for d, s in zip(dst, src):
                SlimTrainer._copy_tensors(d, s, acc)
        if top and acc:
            L.multi_copy(acc)

    def _signature(self, sample_t0, sample_t1, all_valid):
        shapes = []
        self._map_tensors((sample_t0, sample_t1), lambda t: shapes.append((tuple(t.shape), t.dtype)) or t)
        return (tuple(shapes), all_valid)

    def _capture(self, sample_t0, sample_t1, all_valid):
        dev = self.device
        self._static = self._map_tensors((sample_t0, sample_t1), lambda t: t.to(dev).clone())
        s0, s1 = self._static
        buffers = {k: v.clone() for k, v in self.net.state_dict().items() if v.is_floating_point() or v.dtype == torch.long}
        # The pillar encoder stays OUTSIDE the graph (as in DetectorTrainer): a graph that holds its launches faults -- a GPU
        # memory access fault inside a later replay -- once a few thousand eager launches (the optimizer's, here) have run
        # between replays (scripts/debug_slim_graph_fault*.py: replays alone, copies, allocations, the scheduler are harmless;
        # RMSprop.step() with lr = 0 is enough; independent of the convolution backend).  The canvases become graph inputs, their
        # gradients graph outputs, and the encoder's own backward runs eagerly on them.
        with torch.no_grad():
            canv = self._pillars(sample_t0, sample_t1)
        self._static_canv = tuple(c.detach().clone() for c in canv)
        for i in (0, 2):
            self._static_canv[i].requires_grad_(True)
            self._static_canv[i].grad = None
        quiet = hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch")
        if quiet:
            # the flat gradient views are created on the default stream, warm-up and capture run on side streams
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        side = side_stream(dev, "flow")
        side.wait_stream(torch.cuda.current_stream(dev))
        # So does the decoder's point -> cell plan: a torch.sort of 12 x B x N (2.9 M) keys.  A captured torch.sort of more than
        # ~1 M keys is what makes the replay fault after a few thousand unrelated eager launches (scripts/debug_pillar_graph_fault.py:
        # 1.0 M keys replay cleanly, 1.5 M fault; rocPRIM's large-input radix sort on this ROCm build) -- the root cause of the
        # "graph + eager launches" faults of rounds 1-2.  The plan depends on the sweeps only: built eagerly, copied in.
        self._static_plan = None
        meta = getattr(self.net, "gather_plan_meta", None)
        if meta is None:
            with torch.no_grad():
                self.net(s0, s1, None, canvases=self._static_canv)
            meta = getattr(self.net, "gather_plan_meta", None)
        if meta is not None:
            self._static_plan = self.net.build_gather_plan(s0, s1, *meta)
            self._static_plan.lin64  # (its lazily built int64 copy is a graph input too)

        from liso_amd.utils import mfma_conv as MC

        self._pack_jobs = None

        # several ranks + the dynamicness threshold learned from the loss (static aggregation): its updates all-reduce a histogram
        # increment each -- a collective cannot sit inside the graph -- so the captured step only RECORDS the per-rank increments and
        # `_graph_step` reduces and applies them behind the replay (movavg_cls_threshold.py: apply_deferred)
        thr = self.net.moving_dynamicness_threshold
        defer = self.world > 1 and bool(self.slim_cfg.model.use_static_aggr_flow_for_aggr_flow)
        self._thr_items = None

        def body():
            self._flat_grad.zero_()
            for i in (0, 2):  # (autograd keeps the stems' data gradients as the canvases' gradients: no fill, one add less)
                self._static_canv[i].grad = None
            if self._pack_jobs:  # every layer's forward / data-gradient panels from ONE launch (recorded in the warm-up)
                MC.set_step_packs(MC.batched_pack(self._pack_jobs))
            thr._defer = [] if defer else None
            for p_, _ in self._gather_params:
                p_.grad = None
            try:
                total, _, _ = self.loss(s0, s1, all_valid, canvases=self._static_canv, gather_plan=self._static_plan)
                total.backward()
            finally:
                MC.set_step_packs(None)
                self._thr_items, thr._defer = thr._defer, None
                self._gather_keepalive = gather_gradients(self.device, self._gather_params)
            return total.detach()

        with torch.cuda.stream(side):  # warm-up off the capture: MIOpen / rocBLAS pick their kernels, caches fill
            MC.record_pack_jobs(True)
            body()
            self._pack_jobs = MC.record_pack_jobs(False)
            body()
        torch.cuda.current_stream(dev).wait_stream(side)
        with torch.no_grad():  # the warm-up passes must not count as training steps (BN / threshold statistics)
            for k, v in self.net.state_dict().items():
                if k in buffers:
                    v.copy_(buffers[k])
        self._graph = torch.cuda.CUDAGraph()
        # capture on the warm-up's stream: the AccumulateGrad nodes of the parameters live there (a capture on another stream
        # forks into it: wrong results, measured on the detector step)
        with torch.cuda.graph(self._graph, stream=side):
            self._static_loss = body()
        with torch.no_grad():  # capture does not execute, but restore anyway in case the backend ran eagerly
            for k, v in self.net.state_dict().items():
                if k in buffers:
                    v.copy_(buffers[k])
        if quiet:  # process-wide switch: only silenced while capturing
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(True)

    def capture(self, sample_t0, sample_t1):
        """Capture the graph for inputs shaped like these (no collective is issued: callers running several ranks can
        agree on the outcome before the first step).  Raises whatever the runtime raises if capture is refused."""
        _, m1, _, m2 = self._inputs(sample_t0, sample_t1)
        all_valid = (bool(m1.all()), bool(m2.all()))
        sig = self._signature(sample_t0, sample_t1, all_valid)
        if self._graph is None or sig != self._graph_sig:
            self._graph = None
            self._capture(sample_t0, sample_t1, all_valid)
            self._graph_sig = sig
            return True
        return False

    def _pillars(self, sample_t0, sample_t1):
        from liso_amd.slim.model.slim import get_network_input_pcls

        dev = self.device
        return self.net.raft_network.encode_pillars(get_network_input_pcls(self.cfg, sample_t0, "ta", to_device=dev),
                                                    get_network_input_pcls(self.cfg, sample_t1, "ta", to_device=dev))

    def _graph_step(self, sample_t0, sample_t1):
        if not self.capture(sample_t0, sample_t1):
            self._copy_tensors(self._static, (sample_t0, sample_t1))
        canv = self._pillars(sample_t0, sample_t1)  # eager, with autograd: the encoder's parameters get their gradients below
        with torch.no_grad():
            for d, c in zip(self._static_canv, canv):
                d.copy_(c.detach(), non_blocking=True)
            if self._static_plan is not None:
                fresh = self.net.build_gather_plan(sample_t0, sample_t1, *self.net.gather_plan_meta)
                for name in ("lin", "sorted_lin", "order", "seg_rank"):
                    getattr(self._static_plan, name).copy_(getattr(fresh, name), non_blocking=True)
                self._static_plan.lin64.copy_(fresh.lin, non_blocking=True)
        self._graph.replay()
        if self._thr_items:  # (several ranks: the threshold updates the captured step recorded, reduced over the ranks and applied)
            with torch.no_grad():
                self.net.moving_dynamicness_threshold.apply_deferred(self._thr_items)
        live = [(canv[i], self._static_canv[i].grad) for i in (0, 2) if canv[i].requires_grad]
        if live:
            torch.autograd.backward([c for c, _ in live], [g for _, g in live])
        self._reduce_and_update()
        return self._static_loss.clone()  # the captured output is overwritten by the next replay


class _Prefetched:
    """what the pipeline of LisoLoopTrainer holds for one announced sample pair: the point flow of stage A and, after stage B, the
    target maps / boxes and the cluster count (pinned host tensor) -- each guarded by the event recorded behind its work"""
    __slots__ = ("pair", "flow", "done", "targets", "boxes", "cluster_count")

    def __init__(self, pair, flow, done, targets=None, boxes=None, cluster_count=None):
        self.pair, self.flow, self.done = pair, flow, done
        self.targets, self.boxes, self.cluster_count = targets, boxes, cluster_count

    def is_for(self, pair):
        return self.pair[0] is pair[0] and self.pair[1] is pair[1]


def _packed_inputs(tensors):
    """static copies of a graph's input tensors as typed views of ONE byte buffer, so that a replay refreshes all of them with a
    single concatenation launch: -> {"in": {name: view}, "in_flat": uint8 buffer | None, "order": names}.  Segments whose byte size
    is not a multiple of 16 go last (every view then starts 16-B aligned); tensors that are not contiguous or live elsewhere make
    the caller fall back to one copy per tensor (in_flat None)."""
    names = sorted(tensors, key=lambda k: (tensors[k].numel() * tensors[k].element_size()) % 16 != 0)
    sizes = [tensors[k].numel() * tensors[k].element_size() for k in names]
    ok = all(tensors[k].is_contiguous() and tensors[k].numel() > 0 for k in names) and sum(1 for z in sizes if z % 16) <= 1
    if not ok:
        return {"in": {k: v.clone() for k, v in tensors.items()}, "in_flat": None, "order": names}
    flat = torch.empty(sum(sizes), dtype=torch.uint8, device=tensors[names[0]].device)
    views, off = {}, 0
    for k, z in zip(names, sizes):
        t = tensors[k]
        views[k] = flat[off:off + z].view(t.dtype).view(t.shape)
        views[k].copy_(t)
        off += z
    return {"in": views, "in_flat": flat, "order": names}


def _pack_tensors(named):
    """[(name, tensor)] -> (flat uint8 tensor = ONE torch.cat launch, layout) with every segment aligned for its dtype: 8-byte types
    first, then 4-byte, then the rest"""
    order = sorted(range(len(named)), key=lambda i: -named[i][1].element_size())
    layout, parts, off = [], [], 0
    for i in order:
        name, t = named[i]
        t = t.contiguous()
        u = t.view(torch.uint8) if t.dtype != torch.bool else t.view(torch.uint8)
        layout.append((name, off, u.numel(), t.dtype, tuple(t.shape)))
        parts.append(u.reshape(-1))
        off += u.numel()
    return torch.cat(parts), layout


def _unpack_tensors(flat, layout):
    out = {}
    for name, off, nbytes, dtype, shape in layout:
        seg = flat[off:off + nbytes]
        out[name] = (seg.view(torch.uint8).view(dtype) if dtype != torch.bool else seg.view(torch.bool)).view(shape)
    return out


class _BatchedTargets(dict):
    """target maps of a batch given as one dict per sample ([1, ...] tensors): behaves like the concatenated dict (built lazily,
    one torch.cat per key) and lets the graph path copy sample by sample into its captured inputs without concatenating"""

    def __init__(self, per_sample):
        super().__init__()
        self.per_sample = list(per_sample)
        for k in self.per_sample[0]:
            dict.__setitem__(self, k, None)

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if v is None:
            v = torch.cat([t[k] for t in self.per_sample], dim=0)
            dict.__setitem__(self, k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def shapes(self):
        rows = {k: sum(t[k].shape[0] for t in self.per_sample) for k in self.per_sample[0]}
        return tuple((k, (rows[k],) + tuple(v.shape[1:]), v.dtype) for k, v in sorted(self.per_sample[0].items()))


class LisoLoopTrainer:
    """One fused LISO iteration per sample pair: SLIM forward (no_grad) -> per-point flow ->
    FlowClusterDetector (BEV dynamicness, DBSCAN, region moments, z-fit, filters, Kabsch heading/velocity) -> rotated NMS
    (pre 1000 / post 100 / IoU 0.1, liso_config.yml:4,27-28) -> CenterPoint target maps -> detector train step.
    The reference runs these stages as separate jobs that exchange files (flow export: slim/experiment.py:363-471;
    box mining + box DB: tracker/; training: liso_cli.py); here the tensors stay in HBM from the sweep to the gradient.
    Box-DB augmentation and tracking between the stages are outside this loop."""

    def __init__(self, cfg, device, compute_dtype=torch.float32, total_steps=None, slim_state_dict=None, use_graph=False,
                 overlap=False, infer_batch=2, flow_ahead=0, exact=None):
        """`use_graph`: the frozen SLIM inference (one capture per input shape) and the detector's forward+loss+backward are
        replayed from hipGraphs; the flow clustering in between stays eager (its box count sizes the padded Shape).
        `overlap`: step(pair_i, upcoming=(pair_i+1, pair_i+2)) runs the iteration as a three-stage software pipeline on
        three HIP streams (see _stage_a / _stage_b below); results are those of the one-stream loop.
        `infer_batch`: pairs per SLIM inference replay; `flow_ahead`: steps by which an inference batch is issued BEFORE stage B needs
        its first flow (0: just in time -- stage B of the next two pairs then queues behind a whole batch replay).  The pipeline uses
        up to `2 + flow_ahead + infer_batch - 1` announced pairs."""
        from liso_amd.networks.flow_cluster_detector.flow_cluster_detector import FlowClusterDetector
        from liso_amd.slim.model.slim import SLIM

        self.cfg, self.device = cfg, device
        if exact is not None:  # arithmetic of the fp32 convolutions (SLIM; the detector too when compute_dtype is float32)
            from liso_amd.utils import mfma_conv as MC
            MC.set_fp32_mode("exact" if exact else "x3")
        if device.type == "cuda":  # (persistent device flag of the sparse canvas convolutions: must exist before the first capture)
            from liso_amd.utils import mfma_conv as _MC
            _MC._sparse_flag(device)
        self.use_graph = bool(use_graph) and device.type == "cuda"
        self._graph_infer = self.use_graph and use_graph in (True, "infer")
        self._graph_det = self.use_graph and use_graph in (True, "detector")
        import collections
        self._infer_graph, self._infer_graphs = None, collections.OrderedDict()
        self._mine_graphs = collections.OrderedDict()
        self._graph_mine = self.use_graph and use_graph is True
        # real sweeps differ in their point count from sample to sample: the padded loss clouds are grown to the next multiple of
        # `infer_point_bucket` rows (NaN rows, pcl_is_valid False, pillar_coors -1: the dataset's own collate padding,
        # torch_dataset_commons.py:380-431) so that the inference graph's input signature repeats, and at most `max_infer_graphs`
        # captured graphs (each with its static inputs and a private memory pool) stay resident, least recently used first out
        tcfg = cfg.data.tracking_cfg
        self.infer_point_bucket = int(tcfg.setdefault("infer_point_bucket", 8192))
        self.max_infer_graphs = int(tcfg.setdefault("max_infer_graphs", 8))
        # stage B's captured graphs are keyed on the same bucket-padded shapes (plus the full cloud's padded length): resident graphs
        # per mining stream, and the number of captures after which a new signature runs the eager fixed-slot path instead of evicting
        # a graph whose replays may still be in flight
        self.max_mine_graphs = int(tcfg.setdefault("max_mine_graphs", 6))
        self.mine_capture_budget = int(tcfg.setdefault("mine_capture_budget", 24))
        self.mine_capture_window = int(tcfg.setdefault("mine_capture_window", 400))
        self.mine_captures, self.mine_eager_fallbacks = 0, 0
        self._mine_calls, self._mine_capture_calls, self._mine_fallback_warned = 0, [], False
        self._views = collections.OrderedDict()  # id(sample) -> (sample, bucket-padded view); the samples themselves are never edited
        self.overlap = bool(overlap) and device.type == "cuda"
        self.infer_batch, self.flow_ahead = int(infer_batch), int(flow_ahead)
        self.box_capacity, self.capacity_overflows = int(cfg.data.tracking_cfg.setdefault("flow_cluster_capacity", 64)), 0
        self._flow_stream, self._mine_stream, self._main_used_static = None, None, None
        self._inputs_ready = None  # event on the caller's stream at the head of a step: the side streams' reads of `upcoming` wait on it
        self._flows, self._mined = [], []
        self.slim = SLIM(cfg, num_train_samples=1000).to(device)
        if slim_state_dict is not None:
            self.slim.load_state_dict(slim_state_dict)
        self.slim.eval()
        if os.environ.get("LISO_ENC_IMAGES"):  # experiment: encoder passes of at most this many images inside an inference batch
            self.slim.raft_network.encoder_images_per_pass = int(os.environ["LISO_ENC_IMAGES"])
        for p_ in self.slim.parameters():  # frozen: its packed convolution panels are built once and never re-packed
            p_.requires_grad_(False)
        self.cluster_detector = FlowClusterDetector(cfg).to(device)
        self.detector = DetectorTrainer(cfg, device, compute_dtype=compute_dtype, total_steps=total_steps, use_graph=self._graph_det)
        if device.type == "cuda":
            # three streams in all (the caller's, A, B): HIP multiplexes streams onto 4 hardware queues, and two streams on one
            # queue take turns.  Graph capture / warm-up of both networks borrows stream A.
            # (one set per device and PROCESS, not per trainer: side_stream)
            self._flow_stream = side_stream(device, "flow")
            self._mine_stream = side_stream(device, "mine0", priority=-1)  # many tiny kernels + host reads: dispatch first
            # stage B is a chain of ~100 dependent small launches; under contention it is the slowest stage (the host waits ~1.4 ms
            # per step for it).  LISO_MINE_STREAMS=2 alternates consecutive pairs between two mining streams (two chains in
            # flight): 4.46-4.65 instead of 4.6-4.8 ms per step on one GPU, but that is the 4th stream -- one more (RCCL's, in a
            # multi-GPU run) and two streams share a hardware queue (measured with 3 mining streams: 6.4 ms).  Default 1.
            n_mine = max(1, int(os.environ.get("LISO_MINE_STREAMS", "1")))
            self._mine_streams = [self._mine_stream] + [side_stream(device, f"mine{k}", priority=-1) for k in range(1, n_mine)]
            self._mine_turn = 0
            self.detector._capture_stream = self._flow_stream
        # compute units the captured SLIM inference's persistent 3x3 convolutions may take while the pipeline overlaps it with the
        # detector step (0 = all): half the chip by default, see _infer_flow_padded
        self.infer_cus = 0
        if device.type == "cuda" and self.overlap and os.environ.get("LISO_DETECTOR_CUS"):
            self.detector.roles_cus = int(os.environ["LISO_DETECTOR_CUS"])  # (experiment: the detector's persistent convolutions capped too)
        if device.type == "cuda" and self.overlap:
            n_cu = torch.cuda.get_device_properties(device).multi_processor_count
            self.infer_cus = int(os.environ.get("LISO_INFER_CUS", str(n_cu // 2)))
            if self.infer_cus >= n_cu:
                self.infer_cus = 0
        self._pillar_prep = None  # (clouds, pfn.prepare(clouds), event) of the next detector step
        self._prep_ahead = os.environ.get("LISO_PREP_AHEAD", "1") != "0"
        tc = cfg.data.tracking_cfg
        self.pre_nms, self.post_nms = tc.max_num_boxes_before_nms, tc.max_num_boxes_after_nms
        self.nms_iou = cfg.setdefault("nms_iou_threshold", 0.1)

    @torch.no_grad()
    def mine_boxes(self, sample_t0, sample_t1):
        """-> (Shape [B,K] after NMS, padded with zeros; point flow [B,N,3])"""
        flow = self._infer_flow(sample_t0, sample_t1)  # one direction, last RAFT iteration
        _, boxes = self._targets_from_flow(sample_t0, flow)
        return boxes, flow

    @staticmethod
    def _stack_samples(samples):
        """batch of sample dicts (each with batch size 1) -> one sample dict: tensors are concatenated along the batch axis, lists
        (per-sample clouds of different lengths) are chained, anything else is taken from the first sample"""
        first = samples[0]
        if torch.is_tensor(first):
            return torch.cat(list(samples), dim=0) if first.dim() > 0 else first
        if isinstance(first, dict):
            return {k: LisoLoopTrainer._stack_samples([s_[k] for s_ in samples]) for k in first}
        if isinstance(first, (list, tuple)):
            return type(first)(x for s_ in samples for x in s_)
        return first

    def _pad_loss_cloud(self, sample):
        """the sample with the point axis of `pcl_ta` (NaN rows, pcl_is_valid False, pillar_coors -1: the dataset's own collate padding,
        torch_dataset_commons.py:380-431) and of `pcl_full_w_ground_ta` (NaN rows at the end: no box contains them) grown to the next
        multiple of `infer_point_bucket`.  A new dict: the caller's sample is not modified."""
        pa = sample["pcl_ta"]
        bk = self.infer_point_bucket
        n = pa["pcl"].shape[1]
        pad = (-n) % bk if bk > 1 else 0
        full = sample.get("pcl_full_w_ground_ta")
        fpad = (-full.shape[1]) % bk if (bk > 1 and torch.is_tensor(full) and full.dim() == 3) else 0
        if pad == 0 and fpad == 0:
            return sample
        F = torch.nn.functional
        out = dict(sample)
        if pad:
            out["pcl_ta"] = {**pa, "pcl": F.pad(pa["pcl"], (0, 0, 0, pad), value=float("nan")),
                             "pcl_is_valid": F.pad(pa["pcl_is_valid"], (0, pad), value=False),
                             "pillar_coors": F.pad(pa["pillar_coors"], (0, 0, 0, pad), value=-1)}
        if fpad:
            out["pcl_full_w_ground_ta"] = F.pad(full, (0, 0, 0, fpad), value=float("nan"))
        return out

    def _view(self, sample):
        """bucket-padded view of a sample, built once per sample object and kept on the trainer (bounded, oldest first out).  The
        three pipeline stages run on different streams: the padding launches are followed by an event, a stage that meets the view on
        another stream waits for it and marks the padded tensors as in use there (they must not be recycled under that stream when
        the entry leaves the cache)."""
        cuda = self.device.type == "cuda"
        cur = torch.cuda.current_stream(self.device) if cuda else None
        hit = self._views.get(id(sample))
        if hit is not None and hit[0] is sample:
            self._views.move_to_end(id(sample))
            _, v, ev, made_on, seen = hit
            if cuda and ev is not None and cur.cuda_stream != made_on and cur.cuda_stream not in seen:
                cur.wait_event(ev)
                for t in self._padded_tensors(sample, v):
                    t.record_stream(cur)
                seen.add(cur.cuda_stream)
            return v
        v = self._pad_loss_cloud(sample)
        ev = None
        if cuda and v is not sample:
            ev = torch.cuda.Event()
            ev.record(cur)
        self._views[id(sample)] = (sample, v, ev, cur.cuda_stream if cuda else None, set())  # (the strong reference pins id(sample))
        while len(self._views) > 64:
            self._views.popitem(last=False)
        return v

    @staticmethod
    def _padded_tensors(sample, view):
        """the tensors `_pad_loss_cloud` created for `view` (those that are not the sample's own)"""
        out = []
        if view["pcl_ta"] is not sample["pcl_ta"]:
            out += [view["pcl_ta"][k] for k in ("pcl", "pcl_is_valid", "pillar_coors")]
        if view.get("pcl_full_w_ground_ta") is not sample.get("pcl_full_w_ground_ta"):
            out.append(view["pcl_full_w_ground_ta"])
        return out

    def _infer_view(self, sample):
        """what the frozen SLIM inference reads from a (padded) sample: the loss cloud, the odometry and the network-input clouds --
        samples whose OTHER tensors differ in shape (the full cloud with ground) still stack into one inference batch"""
        v = self._view(sample)
        key = "pcl_full_w_ground_ta" if self.cfg.data.use_ground_for_network else "pcl_full_no_ground_ta"
        return {"pcl_ta": v["pcl_ta"], "gt": {"odom_ta_tb": v["gt"]["odom_ta_tb"]}, key: list(sample[key])}

    def _infer_flow(self, sample_t0, sample_t1):
        """frozen SLIM inference of one sample pair (any batch size): eager pillar encoder + one hipGraph replay per input signature.
        Eager and replayed calls see the same (bucket-padded) clouds: their results are bit-identical.  -> flow [B, N, 3] of the
        sample's own rows."""
        n_true = sample_t0["pcl_ta"]["pcl"].shape[1]
        flow = self._infer_flow_padded(self._infer_view(sample_t0), self._infer_view(sample_t1))
        return flow if flow.shape[1] == n_true else flow[:, :n_true].contiguous()

    @staticmethod
    def _graph_inputs(sample_t0, sample_t1):
        """what the captured inference reads from the samples (the network-input clouds go through the eager pillar encoder)"""
        return ({"pcl_ta": sample_t0["pcl_ta"], "gt": {"odom_ta_tb": sample_t0["gt"]["odom_ta_tb"]}},
                {"pcl_ta": sample_t1["pcl_ta"], "gt": {"odom_ta_tb": sample_t1["gt"]["odom_ta_tb"]}})

    def _infer_flow_padded(self, sample_t0, sample_t1):
        if not self._graph_infer:
            return self.slim.infer_point_flow_t0_t1(sample_t0, sample_t1)
        shapes = []
        # the key holds only what the graph consumes: the padded loss clouds + odometry (the raw clouds' lengths do not enter)
        SlimTrainer._map_tensors(self._graph_inputs(sample_t0, sample_t1), lambda t: shapes.append((tuple(t.shape), t.dtype)) or t)
        sig = tuple(shapes)
        dev = self.device
        from liso_amd.slim.model.slim import get_network_input_pcls

        raft = self.slim.raft_network
        pcls = (get_network_input_pcls(self.cfg, sample_t0, "ta", to_device=dev), get_network_input_pcls(self.cfg, sample_t1, "ta", to_device=dev))
        st = self._infer_graphs.get(sig)
        with torch.no_grad():  # pillar encoder eagerly (its rocPRIM sort memsets: liso_amd/utils/graph_safety.py) ...
            if st is None:
                canv = raft.encode_pillars(*pcls)
            else:  # ... straight into the graph's input buffers: no copy, no concatenation of the two sweeps
                raft.encode_pillars(*pcls, out=st["rows"])
        with torch.no_grad():  # a device scan (torch.cumsum): eagerly, its memset nodes do not survive in a graph (graph_safety.py)
            thr = self.slim.moving_dynamicness_threshold.value()
        if st is not None:
            self._infer_graphs.move_to_end(sig)
        if st is None:
            if len(self._infer_graphs) >= max(self.max_infer_graphs, 1):
                torch.cuda.synchronize(dev)  # (replays of the graph that goes may still be in flight on the inference stream)
            while len(self._infer_graphs) >= max(self.max_infer_graphs, 1):  # least recently used graph + its buffers go
                _, old = self._infer_graphs.popitem(last=False)
                if self._infer_graph is old.get("graph"):
                    self._infer_graph = None
                old.clear()
            st = self._infer_graphs[sig] = {}
            st["in"] = SlimTrainer._map_tensors(self._graph_inputs(sample_t0, sample_t1), lambda t: t.to(dev).clone())
            B_ = canv[0].shape[0]
            rows = torch.cat([canv[0], canv[2]], dim=0).permute(0, 2, 3, 1).contiguous()  # [2B, gx, gy, 64]
            occ = torch.cat([canv[1], canv[3]], dim=0).contiguous()
            st["rows"] = (rows, occ)
            st["canv"] = (rows[:B_].permute(0, 3, 1, 2), occ[:B_], rows[B_:].permute(0, 3, 1, 2), occ[B_:], rows.permute(0, 3, 1, 2), occ)
            st["thr"] = thr.clone()
            s0, s1 = st["in"]
            side = self._flow_stream  # (HIP maps streams onto 4 hardware queues: capture on a pipeline stream, no extra one)
            side.wait_stream(torch.cuda.current_stream(dev))
            # In the pipeline the captured inference leaves compute units to the other streams (`infer_cus`, LISO_INFER_CUS=n; 0 = all;
            # mfma_conv.roles_cus: its 3x3 convolutions are persistent blocks, one per CU, and a 60-us launch that holds all 256 makes
            # every kernel of the detector step -- the critical path, ~250 small dependent launches -- wait for it to drain).  Measured:
            # round 5, one box: all CUs 4.23-4.26 ms per step; 128: 4.16-4.17; 144: 4.26; 112: 4.35; 96: 4.34; 64: 4.78 (then stage A is
            # the slowest stage); round 6, one box, 40 steps, twice each: all 4.38 / 4.40; 128: 4.25 / 4.27; 160: 4.39 / 4.40; 96: 4.33 /
            # 4.34 (scripts/infer_cus_sweep.sh).  Default since round 6: half the chip; bench.py's `roofline` prices the capped launches
            # against the peak of the CUs they may use.
            from liso_amd.utils import mfma_conv as MC

            cus = self.infer_cus
            with torch.cuda.stream(side), torch.no_grad(), MC.roles_cus(cus):
                for _ in range(2):
                    self.slim.infer_point_flow_t0_t1(s0, s1, canvases=st["canv"], dynamicness_threshold=st["thr"])
            torch.cuda.current_stream(dev).wait_stream(side)
            st["graph"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st["graph"], stream=side), torch.no_grad(), MC.roles_cus(cus):
                st["flow"] = self.slim.infer_point_flow_t0_t1(s0, s1, canvases=st["canv"], dynamicness_threshold=st["thr"])
            self._infer_graph = st["graph"]  # (the most recently captured one: scripts / bench introspection)
        else:
            SlimTrainer._copy_tensors(st["in"], self._graph_inputs(sample_t0, sample_t1))
            st["thr"].copy_(thr, non_blocking=True)
        st["graph"].replay()
        return st["flow"]

    def eager_pass(self, sample_t0, sample_t1, also=()):
        return self.eager_pass_batch([(sample_t0, sample_t1)], also)

    def eager_pass_batch(self, pairs, also=()):
        """`_eager_pass_batch` with the convolution plans the pipelined steps use (the kernels bench.py times are the replayed ones)"""
        if self.device.type != "cuda":
            return self._eager_pass_batch(pairs, also)
        from liso_amd.utils import mfma_conv as MC

        with MC.shared_gpu():
            return self._eager_pass_batch(pairs, also)

    def _eager_pass_batch(self, pairs, also=()):
        """the whole iteration on a batch of pairs with eager launches and no parameter update (per-kernel event timing in bench.py).
        `also`: further pairs that join the SLIM inference batch, as in the pipeline's stage A (the kernels are then timed at the
        batch they run at; their launches count len(pairs) / (len(pairs) + len(also)) towards this step)"""
        from liso_amd import _lib as L

        g, self._graph_infer = self._graph_infer, False
        try:
            allp = [*pairs, *also]
            with torch.no_grad():
                L.TIMER.weight = len(pairs) / len(allp)  # (these launches serve len(allp) pairs, the step consumes len(pairs))
                from liso_amd.utils import mfma_conv as MC

                try:  # (the plans of the captured inference: the same compute-unit cap as in _infer_flow_padded)
                    with MC.roles_cus(self.infer_cus):
                        flow = self._infer_flow_padded(self._stack_infer_views([p_[0] for p_ in allp]), self._stack_infer_views([p_[1] for p_ in allp]))
                finally:
                    L.TIMER.weight = 1.0
                b = flow.shape[0] // len(allp)
                per = [self._targets_from_flow(p_[0], flow[k * b:(k + 1) * b].contiguous())[0] for k, p_ in enumerate(pairs)]
        finally:
            self._graph_infer = g
        targets = per[0] if len(per) == 1 else {k: torch.cat([t[k] for t in per], dim=0) for k in per[0]}
        return self.detector.eager_pass([c for p_ in pairs for c in p_[0]["pcl_full_no_ground_ta"]], targets)

    def _mine_from_graph(self, sample_t0, flow, side):
        """stage B of one sweep pair -- flow clustering, z-fit, filters, Kabsch heading, NMS, target maps with `box_capacity` fixed
        slots -- replayed from a hipGraph on stream `side` (the caller's current stream): ~45 nodes, no scan / sort library call and
        therefore no memset node (liso_amd/utils/graph_safety.py), no host read.  The inputs are the sample's BUCKET-PADDED view
        (`_view`: loss cloud, validity, pillar coordinates, full cloud, all grown to multiples of `infer_point_bucket` like the
        inference graph's) and the padded flow, so sweeps of different point counts share a signature; they are copied into the
        captured inputs by one launch, inv(odom) - I is computed INSIDE the graph (liso_odom_inverse_minus_eye_f64), and all results
        come out of ONE packed buffer that is cloned behind the replay (the next replay overwrites the captured one).  Beyond
        `mine_capture_budget` captures per `mine_capture_window` calls a new signature runs the eager fixed-slot path instead of evicting a graph.
        -> (targets dict, boxes Shape, max cluster count int64 [1])"""
        from liso_amd.kabsch.shape_utils import Shape

        dev = self.device
        v = self._view(sample_t0)
        pa = v["pcl_ta"]
        gt = v[self.cfg.data.odom_source]
        flow = self._pad_flow(flow, pa["pcl"].shape[1])
        ins = {"pcl": pa["pcl"], "valid": pa["pcl_is_valid"], "coors": pa["pillar_coors"], "full": v["pcl_full_w_ground_ta"],
               "flow": flow, "dt": v["src_trgt_time_delta_s"], "odom": gt["odom_ta_tb"]}
        ins = {k: (t if t.device == dev else t.to(dev, non_blocking=True)) for k, t in ins.items()}  # (host-resident loader outputs)
        # (one captured graph -- static inputs, intermediates, packed output -- per mining stream: replays on different streams run
        # concurrently and must not share buffers)
        sig = (side.cuda_stream,) + tuple((k, tuple(t.shape), t.dtype) for k, t in ins.items())
        st = self._mine_graphs.get(sig)
        # the budget is a RATE -- at most `mine_capture_budget` captures per `mine_capture_window` mining calls -- so a signature that
        # was evicted from the LRU can be captured again later; a lifetime budget (round 4) left every non-resident signature on the
        # eager path for the rest of a long run
        self._mine_calls += 1
        while self._mine_capture_calls and self._mine_capture_calls[0] <= self._mine_calls - self.mine_capture_window:
            self._mine_capture_calls.pop(0)
        if st is None and len(self._mine_capture_calls) >= self.mine_capture_budget:
            # too many different signatures (bucket too fine for this data): same kernels, launched eagerly, nothing evicted
            self.mine_eager_fallbacks += 1
            if not self._mine_fallback_warned:
                self._mine_fallback_warned = True
                import warnings
                warnings.warn(f"LisoLoopTrainer: {self.mine_capture_budget} box-mining graph captures within {self.mine_capture_window} "
                              "calls -- new point-count signatures run stage B eagerly until the rate drops (raise "
                              "infer_point_bucket or max_mine_graphs)")
            targets, boxes = self._targets_from_flow(sample_t0, flow, capacity=self.box_capacity)
            return targets, boxes, self.cluster_detector.last_num_labels.max().reshape(1)
        if st is not None:
            self._mine_graphs.move_to_end(sig)
            if st["in_flat"] is not None:  # every input into the captured buffers with ONE launch (a byte-wise concatenation)
                parts = [ins[k].reshape(-1).view(torch.uint8) for k in st["order"]]
                assert sum(p_.numel() for p_ in parts) == st["in_flat"].numel()  # (cat(out=) would silently resize the static buffer)
                torch.cat(parts, out=st["in_flat"])
            else:
                for k, t in ins.items():
                    st["in"][k].copy_(t, non_blocking=True)
        else:
            if len(self._mine_graphs) >= max(self.max_mine_graphs, 1):
                torch.cuda.synchronize(dev)  # (replays of the graph that goes may still be in flight on a mining stream)
            while len(self._mine_graphs) >= max(self.max_mine_graphs, 1):
                self._mine_graphs.popitem(last=False)[1].clear()
            self.mine_captures += 1
            self._mine_capture_calls.append(self._mine_calls)
            st = self._mine_graphs[sig] = _packed_inputs(ins)
            si = st["in"]
            sample = {"pcl_ta": {"pcl": si["pcl"], "pcl_is_valid": si["valid"], "pillar_coors": si["coors"]},
                      "pcl_full_w_ground_ta": si["full"], "src_trgt_time_delta_s": si["dt"],
                      self.cfg.data.odom_source: {"odom_ta_tb": si["odom"]}}

            def body():
                targets, boxes = self._targets_from_flow(sample, si["flow"], capacity=self.box_capacity, padded=True)
                named = [("t_" + k, t) for k, t in targets.items()] + \
                    [("b_" + k, t) for k, t in boxes.__dict__.items() if torch.is_tensor(t)] + \
                    [("n_clusters", self.cluster_detector.last_num_labels.max().reshape(1))]
                return _pack_tensors(named)

            with torch.no_grad():
                for _ in range(2):  # warm-up on the capture stream: lazy initialisations, allocator pools
                    body()
                st["graph"] = torch.cuda.CUDAGraph()
                with torch.cuda.graph(st["graph"], stream=side):
                    st["flat"], st["layout"] = body()
        st["graph"].replay()
        out = _unpack_tensors(st["flat"].clone(), st["layout"])
        targets = {k[2:]: t for k, t in out.items() if k.startswith("t_")}
        boxes = Shape(**{k[2:]: t for k, t in out.items() if k.startswith("b_")})
        return targets, boxes, out["n_clusters"]

    @staticmethod
    def _pad_flow(flow, n):
        """point flow [B, n', 3] -> [B, n, 3] (zero rows for the padding points, which are invalid everywhere downstream)"""
        if flow.shape[1] == n:
            return flow
        if flow.shape[1] > n:  # inferred in a batch with a larger point-count bucket: drop the extra padding rows
            return flow[:, :n].contiguous()
        return torch.nn.functional.pad(flow, (0, 0, 0, n - flow.shape[1]))

    def _targets_from_flow(self, sample_t0, flow, capacity=None, odom_minus_eye=None, padded=False):
        """flow clustering -> NMS -> CenterPoint target maps.  Reference-shaped call: two box-count reads size the padded Shape.
        `capacity`: fixed number of box slots and no device->host read at all (FlowClusterDetector.forward); the caller compares
        `self.cluster_detector.last_num_labels` with the capacity later.  Every path (one-stream loop, pipeline stage B eager or
        replayed) works on the sample's bucket-padded view (`_view`) and the flow padded to it: identical inputs, identical boxes;
        `padded`: `sample_t0` already is such a view (the mining graph's static inputs)."""
        from liso_amd.datasets.targets import render_center_targets
        from liso_amd.utils.nms_iou import perform_nms_on_shapes_padded

        with torch.no_grad():
            if not padded:
                sample_t0 = self._view(sample_t0)
            flow = self._pad_flow(flow, sample_t0["pcl_ta"]["pcl"].shape[1])
            sample = dict(sample_t0)
            sample[self.cfg.data.flow_source] = {**sample_t0.get(self.cfg.data.flow_source, {}), "flow_ta_tb": flow}
            boxes = self.cluster_detector(sample, global_step=1, capacity=capacity, odom_minus_eye=odom_minus_eye)
            B = boxes.shape[0]
            if boxes.shape[1] > 0:
                # confidence order, pre-NMS cut, rotated NMS, post-NMS selection, padding values: 3 small launches around the NMS
                # kernels; the fp32 box arrays for the target renderer come out of the last one
                boxes, (pos, dims, rot, valid) = perform_nms_on_shapes_padded(
                    boxes, max_num_boxes=self.post_nms, overlap_threshold=self.nms_iou, pre_nms_max_num_boxes=self.pre_nms,
                    return_target_arrays=True)
                rot = rot[..., None]
            else:  # nothing moved: an all-background target (one padded slot)
                z = torch.zeros((B, 1, 3), device=self.device)
                pos, dims, rot, valid = z, z + 1.0, z[..., :1], torch.zeros((B, 1), dtype=torch.bool, device=self.device)
            out = tuple(int(g) // 4 for g in self.cfg.data.img_grid_size)
            return render_center_targets(pos, dims, rot, valid, out, tuple(self.cfg.data.bev_range_m)), boxes

    # ---- three-stage software pipeline (overlap=True) ------------------------------------------------------------------
    # stage A: frozen SLIM inference of pair i+2 (no host synchronisation)          -> stream _flow_stream
    # stage B: flow clustering + NMS + target maps of pair i+1 (two box-count reads) -> stream _mine_stream
    # stage C: detector train step on pair i                                         -> the caller's stream
    # A and B depend on the sweeps and the frozen SLIM weights only, never on the detector: every pair gets exactly the
    # boxes, targets and parameter update of the one-stream loop (tests/test_gpu_liso_loop.py), the three stages -- each far
    # too small to fill 256 CUs at batch 1 -- share the GPU instead of taking turns, and the host never waits for stage A.
    def _stage_a(self, *pairs):
        """SLIM inference of one or several pairs in ONE batch (same shapes): every convolution of the replay then works on that many
        times the pixels -- at batch 1 they launch 100-200 blocks on 256 CUs"""
        side = self._flow_stream
        if self._inputs_ready is not None:
            side.wait_event(self._inputs_ready)
        if self._main_used_static is not None:  # the static inference buffers were last used on the caller's stream
            side.wait_event(self._main_used_static)
            self._main_used_static = None
        with torch.cuda.stream(side), torch.no_grad():
            # (the flows stay bucket-padded: stage B reads them next to the padded view of the sample)
            if len(pairs) == 1:
                flows = [self._infer_flow_padded(self._infer_view(pairs[0][0]), self._infer_view(pairs[0][1])).clone()]  # (the next replay overwrites the static output)
            else:
                # pairs from different point-count buckets share the batch at the largest bucket among them (more padding rows,
                # ignored everywhere): the inference graph's signature is (batch size, largest bucket), not one per combination
                flow = self._infer_flow_padded(self._stack_infer_views([p_[0] for p_ in pairs]), self._stack_infer_views([p_[1] for p_ in pairs]))
                b = flow.shape[0] // len(pairs)
                flows = [flow[k * b:(k + 1) * b].clone() for k in range(len(pairs))]
            done = torch.cuda.Event()
            done.record(side)
        for p_, fl in zip(pairs, flows):
            self._flows.append(_Prefetched(pair=p_, flow=fl, done=done))

    def _same_shapes(self, pa, pb):
        """can the two pairs share one inference batch?  What the captured inference consumes -- the loss clouds and the odometry --
        must agree in everything but the (bucket-padded) point count, which `_stage_a` levels to the largest bucket of the batch."""
        def sig(pair):
            out = []
            SlimTrainer._map_tensors(self._graph_inputs(self._infer_view(pair[0]), self._infer_view(pair[1])),
                                     lambda t: out.append((tuple(t.shape[:1]) + tuple(t.shape[2:]), t.dtype)) or t)
            return out
        return sig(pa) == sig(pb)

    def _stack_infer_views(self, samples):
        """the inference views of several samples as one batch, levelled to the largest point-count bucket among them"""
        views = [self._infer_view(s_) for s_ in samples]
        n = max(v["pcl_ta"]["pcl"].shape[1] for v in views)
        return self._stack_samples([self._grow_view(v, n) for v in views])

    @staticmethod
    def _grow_view(view, n):
        """an inference view with its loss cloud padded further, to `n` rows"""
        pa = view["pcl_ta"]
        pad = n - pa["pcl"].shape[1]
        if pad == 0:
            return view
        F = torch.nn.functional
        return {**view, "pcl_ta": {**pa, "pcl": F.pad(pa["pcl"], (0, 0, 0, pad), value=float("nan")),
                                   "pcl_is_valid": F.pad(pa["pcl_is_valid"], (0, pad), value=False),
                                   "pillar_coors": F.pad(pa["pillar_coors"], (0, 0, 0, pad), value=-1)}}

    @staticmethod
    def _take(store, pair):
        """pop the entry prefetched for exactly these sample objects"""
        for k, e in enumerate(store):
            if e.is_for(pair):
                return store.pop(k)
        return None

    def _stage_b(self, pair):
        f = self._take(self._flows, pair)
        if f is None:
            self._stage_a(pair)
            f = self._take(self._flows, pair)
        side = self._mine_streams[self._mine_turn % len(self._mine_streams)]
        self._mine_turn += 1
        side.wait_event(f.done)
        if self._inputs_ready is not None:
            side.wait_event(self._inputs_ready)
        with torch.cuda.stream(side):
            f.flow.record_stream(side)
            # fixed number of box slots: the host only enqueues (no box-count reads).  The cluster count goes to pinned memory
            # behind the work; step() looks at it when it takes the result (long after this stream got there) and redoes the
            # pair with the reference-shaped call in the -- so far never seen -- case of more clusters than slots.
            count = torch.empty(1, dtype=torch.int64, pin_memory=True)
            if self._graph_mine:
                targets, boxes, n_clusters = self._mine_from_graph(pair[0], f.flow, side)
                count.copy_(n_clusters, non_blocking=True)
            else:
                targets, boxes = self._targets_from_flow(pair[0], f.flow, capacity=self.box_capacity)
                count.copy_(self.cluster_detector.last_num_labels.max().reshape(1), non_blocking=True)
            done = torch.cuda.Event()
            done.record(side)
        self._mined.append(_Prefetched(pair=pair, flow=f.flow, done=done, targets=targets, boxes=boxes, cluster_count=count))

    def _take_mined(self, pair, cur):
        """targets + boxes of `pair` from stage B (made visible to stream `cur`), or None if the pair was not prefetched"""
        m = self._take(self._mined, pair)
        if m is None:
            return None
        if os.environ.get("LISO_NO_COUNT_WAIT") == "1":  # (experiment: what does the host's wait for the cluster count cost?  scripts/loop_floor.py)
            cur.wait_event(m.done)
            for t in list(m.targets.values()) + [v for v in m.boxes.__dict__.values() if torch.is_tensor(v)]:
                t.record_stream(cur)
            return m.targets, m.boxes
        m.done.synchronize()  # (stage B of this pair was enqueued one or two steps ago)
        cur.wait_event(m.done)
        if int(m.cluster_count[0]) > self.box_capacity:  # more clusters than slots: the exact, reference-shaped call
            m.flow.record_stream(cur)
            self.capacity_overflows += 1
            return self._targets_from_flow(pair[0], m.flow)
        for t in list(m.targets.values()) + [v for v in m.boxes.__dict__.values() if torch.is_tensor(v)]:
            t.record_stream(cur)  # (allocated on the mining stream, consumed here)
        return m.targets, m.boxes

    def step(self, sample_t0, sample_t1, upcoming=()):
        """one iteration on (sample_t0, sample_t1).  With `overlap`, `upcoming` = the pairs of the following calls (up to
        infer_batch + flow_ahead + 1): stage A (SLIM inference) runs ahead on batches of them, stage B (box mining) up to two pairs ahead, each on
        its own stream.
        Prefetched results are matched by object identity; anything announced but not requested next is dropped."""
        return self.step_batch([(sample_t0, sample_t1)], upcoming)

    def step_batch(self, pairs, upcoming=(), inputs_ready=None):
        """`_step_batch` with the convolution plans of a shared GPU (the three pipeline stages run next to each other)"""
        if self.device.type != "cuda":
            return self._step_batch(pairs, upcoming)
        from liso_amd.utils import mfma_conv as MC

        with MC.shared_gpu():
            return self._step_batch(pairs, upcoming, inputs_ready)

    def _step_batch(self, pairs, upcoming=(), inputs_ready=None):
        """one iteration on a BATCH of sweep pairs (each a (sample_t0, sample_t1) with batch size 1): boxes are mined per pair, the
        detector takes ONE train step on the batch of len(pairs) clouds and target maps -- the reference's `batch_size` (2 in
        liso_config.yml:121, 4 in the README commands :637-639), BatchNorm statistics over that batch like the reference's.
        `upcoming`: the pairs of the following calls in order (flat list); the pipeline keeps stage B two batches ahead."""
        cuda = self.device.type == "cuda"
        cur = torch.cuda.current_stream(self.device) if cuda else None
        if cuda:
            # `inputs_ready`: an event the CALLER recorded on the stream that produced / uploaded the announced samples (a DataLoader's
            # copy stream: bench.py --loader).  Every stream of the pipeline that reads them waits for it.  Without it the samples
            # must be complete when they are announced.  (An event recorded HERE on the caller's stream -- the first form of this,
            # early in round 6 -- sits behind the previous step's whole detector work in that stream: the side stages of step i then
            # start only when detector step i - 1 has finished, the host blocks on their results ~2 ms per step and the detector's
            # stream runs dry while the host enqueues the next step: 4.26 instead of 4.10 ms per step, 3.16 instead of 2.72 with both
            # side stages cached; scripts/loop_floor.py.)
            self._inputs_ready = inputs_ready
            if inputs_ready is not None:
                cur.wait_event(inputs_ready)
        mined = []
        for pair in pairs:
            sample_t0, sample_t1 = pair
            got = self._take_mined(pair, cur) if cuda else None
            if got is None:
                f = self._take(self._flows, pair)
                if f is not None:
                    cur.wait_event(f.done)
                    f.flow.record_stream(cur)
                    flow = f.flow
                else:
                    if self.overlap and cuda:
                        # not prefetched (first call, wrong or missing announcement): the inference runs HERE, on the caller's stream,
                        # with the same static graph buffers, packed weight panels and decoder caches stage A uses on its stream
                        cur.wait_stream(self._flow_stream)
                    with torch.no_grad():
                        flow = self._infer_flow_padded(self._infer_view(sample_t0), self._infer_view(sample_t1))
                got = self._targets_from_flow(sample_t0, flow)
                if self.overlap and f is None:  # the static inference buffers were used on this stream up to here
                    self._main_used_static = torch.cuda.Event()
                    self._main_used_static.record(cur)
            mined.append(got)
        self.last_boxes = mined[-1][1]
        self.last_boxes_batch = [m[1] for m in mined]
        nb = len(pairs)
        if nb == 1:
            targets = mined[0][0]
        else:  # (the graph path copies each sample's maps into its slice of the captured inputs: no concatenation launch)
            targets = _BatchedTargets([m[0] for m in mined])
        pcls = [c for p_ in pairs for c in p_[0]["pcl_full_no_ground_ta"]]
        prep = None
        if self.overlap and cuda and self._pillar_prep is not None:
            key, got_prep, done = self._pillar_prep
            self._pillar_prep = None
            if len(key) == len(pcls) and all(a is b_ for a, b_ in zip(key, pcls)):  # (matched by object identity like every prefetch)
                cur.wait_event(done)
                cat, _, parts = got_prep
                for t in (cat, *parts):
                    t.record_stream(cur)  # (allocated on the mining stream, consumed here and by the encoder's backward)
                prep = got_prep
        loss = self.detector.step(pcls, targets, prep=prep) if prep is not None else self.detector.step(pcls, targets)
        if self.overlap and cuda and len(upcoming) >= len(pairs) and self._prep_ahead and self.detector.use_graph:
            # the weight-independent half of the NEXT step's pillar encoder (voxelisation + decorated rows: ~14 of its ~20 eager
            # launches) now, on the mining stream, next to the detector step that was just enqueued: the detector's stream is the
            # pipeline's critical path, and these launches sat at the head of every step of it
            nxt = [c for p_ in upcoming[:len(pairs)] for c in p_[0]["pcl_full_no_ground_ta"]]
            side = self._mine_stream
            if self._inputs_ready is not None:
                side.wait_event(self._inputs_ready)
            with torch.cuda.stream(side):
                got_prep = self.detector.net.model.pfn.prepare(nxt)
                done = torch.cuda.Event()
                done.record(side)
            self._pillar_prep = (nxt, got_prep, done)
        if self.overlap and len(upcoming) > 0:
            ahead = nb * (1 + len(self._mine_streams))  # stage B runs this many pairs ahead (one batch more than chains in flight)
            fa = nb * self.flow_ahead
            up = list(upcoming[:ahead + fa + self.infer_batch - 1])
            has = lambda store, p: any(e.is_for(p) for e in store)  # noqa: E731
            self._mined = [e for e in self._mined if any(e.is_for(q) for q in up)]
            self._flows = [e for e in self._flows if any(e.is_for(q) for q in up)]
            # stage A first (the GPU works on it while the host walks through stage B).  It runs when a pair that stage B needs now
            # (one of the next two batches) has no flow yet, and then takes every announced pair without a flow -- up to `infer_batch`
            # of the same shape -- in one batch: with k pairs announced it runs every k-1 pairs on k-1 pairs.
            missing = [p_ for k, p_ in enumerate(up) if not has(self._flows, p_) and not has(self._mined, p_)
                       and not any(p_[0] is q[0] and p_[1] is q[1] for q in up[:k])]
            if missing and any(p_ is q for p_ in missing for q in up[:ahead + fa]):
                while missing:
                    n = 1
                    while n < min(len(missing), self.infer_batch) and self._same_shapes(missing[0], missing[n]):
                        n += 1
                    self._stage_a(*missing[:n])
                    missing = missing[n:]
            for p_ in up[:ahead]:  # by the time a result is taken, its stream got there long ago
                if not has(self._mined, p_):
                    self._stage_b(p_)
        return loss
