"""Synthetic KITTI-shaped LiDAR scenes on the device (there is no dataset access).

64 rings x 1875 azimuth steps = 120 000 rays from a sensor 1.73 m above a ground plane, cast against ~30 car-sized
boxes and four walls; a second sweep is the same scene after an SE(2) ego motion (<= 1.5 m, <= 3 deg) and per-object
motion (<= 2 m) with 2 cm range noise.  Returns points [N,4] = (x, y, z, intensity), the boxes, and the true flow.
"""
import math

import torch

from liso_amd.datasets.torch_dataset_commons import voxelize_sample


def _rays(device, n_rings=64, n_az=1875):
    elev = torch.deg2rad(torch.linspace(-24.8, 2.0, n_rings, device=device))
    az = torch.arange(n_az, device=device) * (2 * math.pi / n_az)
    ce, se = torch.cos(elev)[:, None], torch.sin(elev)[:, None]
    d = torch.stack([ce * torch.cos(az)[None], ce * torch.sin(az)[None], se.expand(-1, n_az)], dim=-1)
    return d.reshape(-1, 3)


def make_scene(seed, device, n_boxes=30, extent=40.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    pos = (torch.rand(n_boxes, 2, generator=g) * 2 - 1) * extent
    pos = pos + torch.sign(pos) * 3.0  # keep the sensor clear
    dims = torch.stack([torch.rand(n_boxes, generator=g) * 1.5 + 3.5, torch.rand(n_boxes, generator=g) * 0.5 + 1.6,
                        torch.rand(n_boxes, generator=g) * 0.4 + 1.4], dim=-1)
    yaw = (torch.rand(n_boxes, generator=g) * 2 - 1) * math.pi
    speed = torch.rand(n_boxes, generator=g) * 2.0 * (torch.rand(n_boxes, generator=g) > 0.5)  # half of them move
    z = -1.73 + dims[:, 2] / 2
    boxes = torch.cat([pos, z[:, None], dims, yaw[:, None]], dim=-1)
    ego = torch.tensor([torch.rand(1, generator=g).item() * 1.5, (torch.rand(1, generator=g).item() - 0.5) * 0.3,
                        math.radians((torch.rand(1, generator=g).item() - 0.5) * 6.0)])
    return boxes.to(device), speed.to(device), ego.to(device)


def render(boxes, device, seed, n_points=120000, wall=48.0, noise=0.02):
    """cast the 120k rays against ground, walls and boxes (all in the sensor frame); pad/trim to exactly n_points."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = _rays(device)
    R = d.shape[0]
    t = torch.full((R,), float("inf"), device=device)
    t_ground = torch.where(d[:, 2] < -1e-4, -1.73 / d[:, 2], torch.full_like(t, float("inf")))
    t = torch.minimum(t, t_ground)
    for ax in (0, 1):  # four walls, 6 m tall
        tw = wall / d[:, ax].abs().clamp(min=1e-6)
        zhit = tw * d[:, 2]
        t = torch.minimum(t, torch.where(zhit < 4.3, tw, torch.full_like(t, float("inf"))))
    # oriented boxes: slab test in the box frame
    c, s = torch.cos(boxes[:, 6]), torch.sin(boxes[:, 6])
    ox = -(boxes[:, 0] * c + boxes[:, 1] * s)
    oy = -(-boxes[:, 0] * s + boxes[:, 1] * c)
    oz = -boxes[:, 2]
    dx = d[:, None, 0] * c[None] + d[:, None, 1] * s[None]
    dy = -d[:, None, 0] * s[None] + d[:, None, 1] * c[None]
    dz = d[:, None, 2].expand(-1, boxes.shape[0])
    tmin = torch.zeros(R, boxes.shape[0], device=device)
    tmax = torch.full((R, boxes.shape[0]), float("inf"), device=device)
    for o, dd, half in ((ox, dx, boxes[:, 3] / 2), (oy, dy, boxes[:, 4] / 2), (oz, dz, boxes[:, 5] / 2)):
        inv = 1.0 / torch.where(dd.abs() < 1e-9, torch.full_like(dd, 1e-9), dd)
        t1, t2 = (-half[None] - o[None]) * inv, (half[None] - o[None]) * inv
        tmin = torch.maximum(tmin, torch.minimum(t1, t2))
        tmax = torch.minimum(tmax, torch.maximum(t1, t2))
    tb = torch.where(tmax >= tmin, tmin, torch.full_like(tmin, float("inf")))
    tb, which = tb.min(dim=1)
    on_box = tb < t
    t = torch.minimum(t, tb)
    ok = torch.isfinite(t) & (t < 90.0) & (t > 1.0)
    t = t + torch.randn(R, generator=g).to(device) * noise
    pts = d * t[:, None]
    inten = torch.rand(R, generator=g).to(device)
    cloud = torch.cat([pts, inten[:, None]], dim=-1)[ok]
    obj = torch.where(on_box, which, torch.full_like(which, -1))[ok]
    n = cloud.shape[0]
    if n >= n_points:
        cloud, obj = cloud[:n_points], obj[:n_points]
    else:  # top up with jittered duplicates so that the workload is exactly n_points
        idx = torch.randint(0, n, (n_points - n,), generator=g).to(device)
        extra = cloud[idx] + torch.randn(n_points - n, 4, generator=g).to(device) * torch.tensor([0.03, 0.03, 0.01, 0.0], device=device)
        cloud, obj = torch.cat([cloud, extra]), torch.cat([obj, obj[idx]])
    return cloud.contiguous(), obj


def move_scene(boxes, speed, ego):
    """boxes at t1 expressed in the t1 sensor frame: objects advance along their heading, then inverse ego motion."""
    b = boxes.clone()
    b[:, 0] += speed * torch.cos(b[:, 6])
    b[:, 1] += speed * torch.sin(b[:, 6])
    c, s = torch.cos(-ego[2]), torch.sin(-ego[2])
    x, y = b[:, 0] - ego[0], b[:, 1] - ego[1]
    b[:, 0], b[:, 1] = x * c - y * s, x * s + y * c
    b[:, 6] -= ego[2]
    return b


def detector_batch(seed, batch, device, n_points=120000, grid=512, bev_range_m=100.0, max_boxes=15):
    """B clouds + CenterPoint targets on the G/4 grid (<= 15 boxes per sample, liso_config.yml:60)."""
    from liso_amd.datasets.targets import render_center_targets

    pcls, P, D, Rt, V = [], [], [], [], []
    for b in range(batch):
        boxes, _, _ = make_scene(seed * 1000 + b, device)
        cloud, _ = render(boxes, device, seed * 1000 + b, n_points)
        pcls.append(cloud)
        near = torch.argsort(boxes[:, :2].norm(dim=-1))[:max_boxes]
        bx = boxes[near]
        P.append(bx[:, 0:3]); D.append(bx[:, 3:6]); Rt.append(bx[:, 6:7]); V.append(torch.ones(len(bx), dtype=torch.bool, device=device))
    out = grid // 4
    t = render_center_targets(torch.stack(P), torch.stack(D), torch.stack(Rt), torch.stack(V), (out, out),
                              (bev_range_m, bev_range_m))
    return pcls, t


def _se2(dx, dy, dth, device):
    T = torch.eye(4, dtype=torch.float64, device=device)
    c, s = math.cos(dth), math.sin(dth)
    T[0, 0], T[0, 1], T[1, 0], T[1, 1], T[0, 3], T[1, 3] = c, -s, s, c, dx, dy
    return T


def _loss_cloud(cloud, n_points, grid, bev_range_m, gen):
    """ground removed by a z threshold, in BEV range, exactly n_points rows"""
    half = bev_range_m / 2
    keep = (cloud[:, 2] > -1.45) & (cloud[:, :2].abs().amax(dim=1) < half - 1e-3)
    c = cloud[keep]
    n = c.shape[0]
    if n >= n_points:
        c = c[torch.randperm(n, generator=gen)[:n_points].to(c.device)]
    else:
        idx = torch.randint(0, n, (n_points - n,), generator=gen).to(c.device)
        c = torch.cat([c, c[idx] + torch.randn(n_points - n, 4, generator=gen).to(c.device) * torch.tensor([0.02, 0.02, 0.01, 0.0], device=c.device)])
        c[:, :2] = c[:, :2].clamp(-half + 1e-3, half - 1e-3)
    return c.contiguous()


def slim_pair(seed, device, n_points=120000, grid=512, bev_range_m=100.0):
    """Two consecutive KITTI-shaped sweeps for the SLIM step in the reference's sample layout
    (liso/datasets/torch_dataset_commons.py:380-431,975-987): `pcl_full_no_ground_ta` (network input, list of [N,4]),
    `pcl_ta` = {pcl [1,N,4], pcl_is_valid [1,N], pillar_coors [1,N,2] int32 by truncation (analyse_boxes.py:11-17)},
    `gt.odom_ta_tb` [1,4,4] fp64.  Returns (sample_t0, sample_t1); sample_t1 is the time-reversed view."""
    gen = torch.Generator(device="cpu").manual_seed(seed + 77)
    boxes0, speed, ego = make_scene(seed, device, n_boxes=30)
    boxes1 = move_scene(boxes0, speed, ego)
    # denser ray fan than the 64x1875 detector input so that 120k non-ground points remain (rendered twice, merged)
    clouds, with_ground = [], []
    for bx, sd in ((boxes0, seed), (boxes1, seed + 1)):
        a, _ = render(bx, device, sd, n_points=120000)
        b, _ = render(bx, device, sd + 1000, n_points=120000, noise=0.03)
        b[:, :2] += 0.011  # decorrelate the two fans
        clouds.append(_loss_cloud(torch.cat([a, b]), n_points, grid, bev_range_m, gen))
        with_ground.append(a)
    T01 = _se2(float(ego[0]), float(ego[1]), float(ego[2]), device)

    def sample(cloud, full, odom):
        coors, _ = voxelize_sample(cloud, (bev_range_m, bev_range_m), (grid, grid))  # the dataset's convention (B4)
        return {"pcl_full_no_ground_ta": [cloud], "pcl_full_w_ground_ta": full[None],
                "pcl_ta": {"pcl": cloud[None], "pcl_is_valid": torch.ones(1, cloud.shape[0], dtype=torch.bool, device=device),
                           "pillar_coors": coors[None]},
                "gt": {"odom_ta_tb": odom[None]}, "src_trgt_time_delta_s": torch.full((1,), 0.1, device=device)}

    return sample(clouds[0], with_ground[0], T01), sample(clouds[1], with_ground[1], torch.linalg.inv(T01))


def cluster_sample(seed, device, batch=1, n_points=120000, grid=512, bev_range_m=100.0, time_delta_s=0.1):
    """A batch for FlowClusterDetector.forward in the reference's sample layout (flow_cluster_detector.py:94-103):
    `pcl_ta` = {pcl [B,N,4] without ground, pcl_is_valid, pillar_coors}, `pcl_full_w_ground_ta` [B,M,4],
    `gt` = {flow_ta_tb [B,N,3] true per-point flow, odom_ta_tb [B,4,4] fp64}, `src_trgt_time_delta_s` [B].
    Also returns the scene boxes at t0 and their per-sweep displacement for checking the mined boxes."""
    gen = torch.Generator(device="cpu").manual_seed(seed + 177)
    half = bev_range_m / 2
    pcls, flows, odoms, full, scenes = [], [], [], [], []
    for b in range(batch):
        boxes, speed, ego = make_scene(seed * 100 + b, device, n_boxes=30)
        cloud, obj = render(boxes, device, seed * 100 + b, n_points=120000)
        T01 = _se2(float(ego[0]), float(ego[1]), float(ego[2]), device)
        disp = torch.zeros(cloud.shape[0], 3, device=device)
        on = obj >= 0
        k = obj.clamp(min=0)
        disp[:, 0] = torch.where(on, speed[k] * torch.cos(boxes[k, 6]), torch.zeros_like(disp[:, 0]))
        disp[:, 1] = torch.where(on, speed[k] * torch.sin(boxes[k, 6]), torch.zeros_like(disp[:, 1]))
        moved = torch.cat([cloud[:, :3] + disp, torch.ones_like(cloud[:, :1])], dim=-1).double()
        flow = ((torch.linalg.inv(T01) @ moved.T).T[:, :3] - cloud[:, :3].double()).float()
        keep = (cloud[:, 2] > -1.45) & (cloud[:, :2].abs().amax(dim=1) < half - 1e-3)
        c, f = cloud[keep], flow[keep]
        n = c.shape[0]
        idx = torch.randperm(n, generator=gen)[:n_points].to(device) if n >= n_points else \
            torch.cat([torch.arange(n), torch.randint(0, n, (n_points - n,), generator=gen)]).to(device)
        pcls.append(c[idx].contiguous()), flows.append(f[idx].contiguous()), odoms.append(T01), full.append(cloud)
        scenes.append((boxes, speed))
    pcl = torch.stack(pcls)
    coors = torch.stack([voxelize_sample(p, (bev_range_m, bev_range_m), (grid, grid))[0] for p in pcl])
    sample = {"pcl_ta": {"pcl": pcl, "pcl_is_valid": torch.ones(pcl.shape[:2], dtype=torch.bool, device=device),
                         "pillar_coors": coors},
              "pcl_full_w_ground_ta": torch.stack(full),
              "gt": {"flow_ta_tb": torch.stack(flows), "odom_ta_tb": torch.stack(odoms)},
              "src_trgt_time_delta_s": torch.full((batch,), time_delta_s, device=device)}
    return sample, scenes
