"""mirror of liso/tracker/tracking_helpers.py:30-44"""
import torch


def aggregate_odometry_to_world_poses(sensor_odometry_ti_tii, w_T_st0_start_pose=None):
    """[T] odometries sensor(t_i) <- sensor(t_i+1), fp64 [4,4] -> world_T_sensor of the T + 1 frames, fp64 [T+1,4,4] (frame 0 = start pose)"""
    first = sensor_odometry_ti_tii[0]
    poses = [torch.eye(4, dtype=torch.float64, device=first.device) if w_T_st0_start_pose is None else w_T_st0_start_pose]
    for step in sensor_odometry_ti_tii:
        assert poses[-1].dtype == torch.float64 and step.dtype == torch.float64
        poses.append(poses[-1] @ step)
    return torch.stack(poses, dim=0)
