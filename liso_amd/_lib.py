"""ctypes loader for liso_amd/libliso_hip.so (C ABI: include/*.h).

The HIP library is the product.  There is no CPU fallback: if the shared object is
missing, or a device op is called without a GPU, we raise.  PyTorch is only used
for device memory and streams (``tensor.data_ptr()``, ``torch.cuda.current_stream()``).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libliso_hip.so")

_lib = None


class LisoHipError(RuntimeError):
    pass


_ERRORS = {-1: "LISO_EINVAL (bad pointer/size)", -2: "LISO_EWORKSPACE (workspace too small)",
           -3: "LISO_ELAUNCH (HIP launch failed)"}


def _preload_torch_hip_runtime():
    # torch bundles its own libamdhip64 (SONAME libamdhip64.so.7).  Our library needs the same SONAME; make
    # sure torch's copy is the one already mapped so that both share one HIP runtime (one context, one
    # allocator view of device pointers and streams).
    tl = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(tl):
        ctypes.CDLL(tl, mode=ctypes.RTLD_GLOBAL)


def lib():
    """Return the loaded library; raise loudly if it was never built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LisoHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C liso_amd/csrc`.  liso_amd has no CPU fallback for device ops.")
        _preload_torch_hip_runtime()
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def check(code, what):
    if code != 0:
        raise LisoHipError(f"{what} failed: {_ERRORS.get(code, code)}")


def stream_ptr(device=None):
    """Current PyTorch HIP stream as a void* for the C ABI."""
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def multi_copy(pairs):
    """dst.copy_(src) for every (dst, src) of `pairs`, the device-resident, contiguous, same-dtype ones of one device as ONE launch
    (include/liso_optim.h: liso_multi_copy) on the current stream; everything else (host sources, dtype changes, strided views) by
    `copy_(non_blocking=True)`."""
    import torch

    fast = []
    for d, s_ in pairs:
        if (d.is_cuda and s_.is_cuda and d.device == s_.device and d.dtype == s_.dtype and d.shape == s_.shape and d.is_contiguous()
                and s_.is_contiguous() and (not fast or fast[0][0].device == d.device)):
            if d.numel() and d.data_ptr() != s_.data_ptr():
                fast.append((d, s_))
        else:
            d.copy_(s_, non_blocking=True)
    if len(fast) == 1:
        fast[0][0].copy_(fast[0][1], non_blocking=True)
    elif fast:
        n = len(fast)
        dst = (ctypes.c_void_p * n)(*[d.data_ptr() for d, _ in fast])
        src = (ctypes.c_void_p * n)(*[s_.data_ptr() for _, s_ in fast])
        nb = (ctypes.c_size_t * n)(*[d.numel() * d.element_size() for d, _ in fast])
        with torch.cuda.device(fast[0][0].device):
            check(lib().liso_multi_copy(n, dst, src, nb, stream_ptr()), "multi_copy")


def copy_blocks(jobs):
    """`jobs`: (dst, src) pairs of fp32 tensors that are 2-D "rows" views -- dst[i] <- src[i] where both index the same logical block
    [rows, cols...] with the trailing dimensions contiguous and one stride between rows (a filter block inside a merged filter, a
    column range of a matrix, a whole contiguous tensor) -- as ONE launch (include/liso_optim.h: liso_multi_copy_rows)."""
    import torch

    n = len(jobs)
    if n == 0:
        return
    dst = (ctypes.c_void_p * n)()
    src = (ctypes.c_void_p * n)()
    rows = (ctypes.c_uint * n)()
    rb = (ctypes.c_uint * n)()
    ds = (ctypes.c_size_t * n)()
    ss = (ctypes.c_size_t * n)()

    def rows_view(t):
        # -> (rows, elements per row, row stride in elements): dim 0 = rows, the rest one contiguous run
        assert t.dtype == torch.float32 and t.is_cuda
        if t.dim() == 1:
            assert t.stride(0) == 1 or t.numel() <= 1
            return 1, t.numel(), t.numel()
        inner = 1
        for k in range(t.dim() - 1, 0, -1):
            assert t.shape[k] == 1 or t.stride(k) == inner, (tuple(t.shape), t.stride())
            inner *= t.shape[k]
        return t.shape[0], inner, (t.stride(0) if t.shape[0] > 1 else inner)

    for i, (d, s_) in enumerate(jobs):
        assert tuple(d.shape) == tuple(s_.shape), (tuple(d.shape), tuple(s_.shape))
        r, c, dstr = rows_view(d)
        r2, c2, sstr = rows_view(s_)
        assert (r, c) == (r2, c2)
        dst[i], src[i], rows[i], rb[i], ds[i], ss[i] = d.data_ptr(), s_.data_ptr(), r, 4 * c, 4 * dstr, 4 * sstr
    with torch.cuda.device(jobs[0][0].device):
        check(lib().liso_multi_copy_rows(n, dst, src, rows, rb, ds, ss, stream_ptr()), "multi_copy_rows")


def require_cuda(*tensors):
    for t in tensors:
        if not t.is_cuda:
            raise LisoHipError("device op called with a CPU tensor; liso_amd has no CPU fallback")


_vp, _i, _f, _sz, _lg = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_long

# symbol -> (restype, argtypes); mirrors include/*.h exactly (tests/test_abi.py parses the headers and checks)
SIGNATURES = {
    # include/liso_iou3d.h
    "liso_iou3d_overlap_bev_f32": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "liso_iou3d_iou_bev_f32": (_i, [_vp, _i, _vp, _i, _vp, _vp]),
    "liso_iou3d_nms_workspace_bytes": (_sz, [_i]),
    "liso_iou3d_nms_f32": (_i, [_vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "liso_iou3d_nms_normal_f32": (_i, [_vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "liso_iou3d_iou_bev_cpu_f32": (_i, [_vp, _i, _vp, _i, _vp]),
    # include/liso_pillars.h
    "liso_pillars_voxelize_workspace_bytes": (_sz, [_vp, _i, _i]),
    "liso_pillars_voxelize_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_pfn_partials_bytes": (_sz, []),
    "liso_pfn_decorate_workspace_bytes": (_sz, [_i, _i]),
    "liso_pfn_decorate_f32": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_pfn_bn_prepare_f32": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _i, _vp, _vp, _vp, _vp]),
    "liso_pfn_forward_scatter": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "liso_pfn_backward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    # include/liso_kabsch.h
    "liso_kabsch_workspace_bytes": (_sz, [_vp]),
    "liso_kabsch_trafos_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_kabsch_trafos_counted_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_symm_ortho_fwd_f64": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "liso_symm_ortho_bwd_f64": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "liso_weighted_moments_workspace_bytes": (_sz, [_i]),
    "liso_weighted_moments_fwd_f32": (_i, [_vp, _vp, _vp, _i, ctypes.c_long, _vp, _vp, _sz, _vp]),
    "liso_weighted_moments_bwd_f32": (_i, [_vp, _vp, _vp, _i, ctypes.c_long, _vp, _vp, _vp, _vp, _vp]),
    # include/liso_flow_cluster.h
    "liso_bev_dynamic_flow_workspace_bytes": (_sz, [_i, _i, _i]),
    "liso_bev_dynamic_flow_f32": (_i, [_vp, _i, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "liso_odom_inverse_minus_eye_f64": (_i, [_vp, _i, _vp, _vp]),
    "liso_fit_box_z_workspace_bytes": (_sz, [_i, _i]),
    "liso_fit_box_z_f32": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_dbscan_components": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_dbscan_labels": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_region_props": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "liso_region_props_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "liso_region_props_ws": (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    # include/liso_slim.h
    "liso_corr_lookup_fwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_corr_lookup_fwd_tiled_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_corr_lookup_bwd_dvol_f32": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "liso_corr_pyramid_fwd_f32": (_i, [_vp, _vp, _vp, _vp]),
    "liso_corr_pyramid_bwd_f32": (_i, [_vp, _vp, _vp, _vp]),
    "liso_corr_bwd_features_workspace_bytes": (_sz, [_vp]),
    "liso_corr_bwd_features_f32": (_i, [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_rows_combine_f32": (_i, [_lg, _i, _vp, _lg, _vp, _lg, _vp, _lg, _vp, _lg, _vp, _lg, _i, _vp]),
    "liso_gru_out_rows_train_f32": (_i, [_lg, _i, _vp, _lg, _vp, _vp, _lg, _vp, _lg, _vp]),
    "liso_gru_out_rows_bwd_f32": (_i, [_lg, _i, _vp, _lg, _vp, _vp, _lg, _vp, _lg, _vp, _lg, _vp, _vp, _vp]),
    "liso_gru_in_rows_bwd_f32": (_i, [_lg, _i, _vp, _lg, _vp, _lg, _vp, _vp, _vp, _lg, _vp, _lg, _vp, _vp]),
    "liso_raft_state_step_train_f32": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_raft_pack_output_grads_f32": (_i, [_lg, _i, _vp, _vp, _vp, _vp]),
    "liso_nearest_point_loss_fwd_f32": (_i, [_vp] * 8),
    "liso_nearest_point_loss_bwd_f32": (_i, [_vp] * 9),
    # include/liso_slim_decode.h
    "liso_slim_decode_weights_fwd": (_i, [_vp] * 6 + [_i] + [_vp] * 4),
    "liso_slim_decode_weights_bwd": (_i, [_vp] * 9),
    "liso_slim_decode_points_fwd": (_i, [_vp] * 9),
    "liso_slim_decode_points_bwd": (_i, [_vp] * 11),
    "liso_slim_loss_workspace_bytes": (_sz, []),
    "liso_slim_static_points_loss_fwd": (_i, [_i, ctypes.c_long, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_slim_static_points_loss_bwd": (_i, [_i, ctypes.c_long, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_slim_knn_queries": (_i, [_i, _i, ctypes.c_long, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "liso_slim_nearest_point_loss_fwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_slim_nearest_point_loss_bwd": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_bev_gather_fwd_f32": (_i, [_vp, _vp, ctypes.c_long, _i, ctypes.c_float, _vp, _vp]),
    "liso_bev_gather_bwd_f32": (_i, [_vp, _vp, _vp, _vp, ctypes.c_long, _i, _vp, _vp, _vp]),
    "liso_gru_in_fwd_f32": (_i, [_vp, _vp, _vp, ctypes.c_long, _vp, _vp, _vp, _vp, _vp]),
    "liso_gru_in_bwd_f32": (_i, [_vp, _vp, ctypes.c_long, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_long, _vp, _vp]),
    "liso_gru_out_fwd_f32": (_i, [ctypes.c_long, _vp, _vp, _vp, _vp, _vp]),
    "liso_gru_out_bwd_f32": (_i, [ctypes.c_long, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_gru_in_rows_f32": (_i, [ctypes.c_long, _i, _vp, ctypes.c_long, _vp, ctypes.c_long, _vp, _vp, ctypes.c_long, _vp]),
    "liso_gru_out_rows_f32": (_i, [ctypes.c_long, _i, _vp, ctypes.c_long, _vp, _vp, ctypes.c_long, _vp]),
    "liso_multi_copy": (_i, [_i, _vp, _vp, _vp, _vp]),
    "liso_bev_lin_index": (_i, [_vp, _i, _vp, _i, ctypes.c_long, _i, _i, _vp, _vp]),
    "liso_channel_extrema_workspace_bytes": (_sz, []),
    "liso_channel_extrema_f32": (_i, [_vp, ctypes.c_long, _i, _i, _vp, _vp, _sz, _vp]),
    "liso_bev_plan_tile_lin": (_i, [_vp, _i, ctypes.c_long, _i, _i, _i, _i, _vp, _vp]),
    "liso_bev_plan_rank": (_i, [_vp, _vp, ctypes.c_long, _vp, _vp, _vp]),
    "liso_bev_plan_expand": (_i, [_vp, _vp, _vp, _i, ctypes.c_long, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "liso_raft_state_step_f32": (_i, [_i, _i, _vp, ctypes.c_long, _vp, _vp, _vp, _vp]),
    "liso_raft_upsample_scratch_bytes": (_sz, [_vp]),
    "liso_raft_upsample_outputs_fwd_f32": (_i, [_vp] * 5),
    "liso_raft_upsample_outputs_bwd_f32": (_i, [_vp, _vp, _vp, _sz, _vp, _vp, _vp]),
    # include/liso_detector.h
    "liso_centerloss_workspace_bytes": (_sz, [_vp]),
    "liso_centerloss_fwd_f32": (_i, [_vp] * 17 + [_sz, _vp]),
    "liso_centerloss_bwd_f32": (_i, [_vp] * 21),
    "liso_render_center_targets_f32": (_i, [_vp] * 12),
    "liso_conv_pack_weights_batched": (_i, [_vp, _i, _vp]),
    "liso_conv_in_finalize": (_i, [_vp, _i, _i, _i, _i, ctypes.c_long, _vp, _vp, _f, _vp, _vp]),
    "liso_residual_affine_relu_f32": (_i, [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _i, _i, _vp, _i, ctypes.c_long, _i, _vp]),
    # include/liso_box_mining.h
    "liso_scan_workspace_bytes": (_sz, [_i, ctypes.c_long]),
    "liso_scan_inclusive_i32": (_i, [_vp, _i, ctypes.c_long, _vp, _vp, _sz, _vp]),
    "liso_mine_boxes_from_regions": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_mine_filter_compact": (_i, [_vp] * 21),
    "liso_mine_box_motion": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "liso_mine_nms_workspace_bytes": (_sz, [_i, _i]),
    "liso_mine_nms_prepare": (_i, [_i, _i, _i] + [_vp] * 11 + [_sz, _vp]),
    "liso_mine_nms_finish": (_i, [_i, _i, _i] + [_vp] * 16),
    # include/liso_optim.h
    "liso_adamw_step_f32": (_i, [_vp, _vp, _vp, _vp, _sz] + [ctypes.c_double] * 5 + [ctypes.c_long, _vp]),
    "liso_adamw_step_scaled_f32": (_i, [_vp, _vp, _vp, _vp, _sz] + [ctypes.c_double] * 6 + [ctypes.c_long, _vp]),
    "liso_rmsprop_step_f32": (_i, [_vp, _vp, _vp, _sz] + [ctypes.c_double] * 4 + [_vp]),
    "liso_multi_copy_rows": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_gather_f32": (_i, [_i, _vp, _vp, _vp, _vp]),
    # include/liso_bn.h
    "liso_bn_workspace_bytes": (_sz, [_i]),
    "liso_bn_relu_fwd": (_i, [_vp, _i, ctypes.c_long, _i, _vp, _vp, _vp, _vp, _f, _f, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "liso_in_workspace_bytes": (_sz, [_i, _i]),
    "liso_in_relu_fwd": (_i, [_vp, _i, _i, ctypes.c_long, _i, _vp, _vp, _f, _i, _vp, _vp, _vp, _sz, _vp]),
    "liso_in_relu_bwd": (_i, [_vp, _vp, _i, _i, ctypes.c_long, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_in_relu_bwd_sum": (_i, [_vp, _vp, _i, _i, ctypes.c_long, _i, _vp, _vp, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_bn_relu_bwd": (_i, [_vp, _vp, _i, ctypes.c_long, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_bn_relu_bwd_strided": (_i, [_vp, ctypes.c_long, _vp, ctypes.c_long, _i, ctypes.c_long, _i, _vp, _vp, _i, _i, _vp, ctypes.c_long, _vp, _vp, _vp, _sz, _vp]),
    "liso_bn_relu_bwd_ticket": (_i, [_vp, _vp, _i, ctypes.c_long, _i, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "liso_knn_workspace_bytes": (_sz, [_vp, _i]),
    "liso_knn_build_f32": (_i, [_vp, _vp, _i, _i, _vp, _sz, _vp]),
    "liso_knn_sorted_ids": (_i, [_vp, _vp, _i, _vp, _vp]),
    "liso_knn_query_f32": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp]),
    # include/liso_conv.h
    "liso_conv_packed_bytes": (_sz, [_i, _i, _i, _i]),
    "liso_conv_pack_weights": (_i, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "liso_conv_stats_rows": (_i, [_vp]),
    "liso_conv_kernel_kind": (_i, [_vp]),
    "liso_conv_forward": (_i, [_vp] * 10),
    "liso_conv_set_option": (_i, [_i, _i]),
    "liso_conv_forward_sparse": (_i, [_vp] * 11),
    "liso_conv_plan_info": (_i, [_vp, _vp]),
    "liso_conv_wgrad_workspace_bytes": (_sz, [_vp]),
    "liso_conv_wgrad": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "liso_conv_wgrad_sparse_workspace_bytes": (_sz, [_vp]),
    "liso_conv_wgrad_sparse_f32": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _sz, _vp]),
    "liso_conv_wgrad_smallci_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "liso_conv_wgrad_smallci_f32": (_i, [_vp, ctypes.c_long, _vp, ctypes.c_long, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "liso_sparse_conv_stat_groups": (_i, [_i, _i, _i]),
    "liso_sparse_conv_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    "liso_sparse_conv_forward": (_i, [_vp, ctypes.c_long, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_sparse_conv_dgrad": (_i, [_vp, ctypes.c_long, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp, ctypes.c_long, _vp, _vp, _sz, _i, _vp]),
    "liso_sparse_stem_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "liso_sparse_stem_forward_f32": (_i, [_vp, ctypes.c_long, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "liso_conv_bn_finalize": (_i, [_vp, _i, _i, _i, ctypes.c_long, _vp, _vp, _vp, _vp, _vp, _f, _f, _vp, _vp]),
    # include/liso_tracking.h
    "liso_points_in_boxes_workspace_bytes": (_sz, [_vp]),
    "liso_points_in_boxes_f32": (_i, [_vp] * 9 + [_sz, _vp]),
    "liso_match_greedy_f32": (_i, [_vp, ctypes.c_long, ctypes.c_long, _i, _i, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_fit_boxes_closeness_workspace_bytes": (_sz, [ctypes.c_long, _i]),
    "liso_fit_boxes_closeness_f32": (_i, [_vp, ctypes.c_long, _i, _vp, _vp, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "liso_bike_rollout_fwd_f32": (_i, [_i, _i, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp]),
    "liso_bike_rollout_bwd_f32": (_i, [_i, _i, _vp, _vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "liso_smooth_tracks_jerk_f32": (_i, [_vp, _vp, _i, _i, _i, _f, _f, _vp, _vp]),
    # include/liso_augment.h
    "liso_bev_free_mask_workspace_bytes": (_sz, [_i, _i]),
    "liso_bev_free_mask": (_i, [_vp, ctypes.c_long, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "liso_bev_select_free_cells": (_i, [_vp, _vp, _i, _i, _vp, _i, _vp, _vp]),
    "liso_snippet_paste": (_i, [_vp, ctypes.c_long, _vp, _vp, _vp, _vp, ctypes.c_double, ctypes.c_double, _i, _vp, _vp, _vp, _vp]),
}


CONV_MAX_TAPS, CONV_MAX_CLASSES, CONV_BF16, CONV_F32X3, CONV_F32 = 49, 4, 0, 1, 2
CONV_OPT_SHARED_GPU = 1
CONV_OPT_ROLES_CUS = 2


class ConvDesc(ctypes.Structure):
    """mirror of liso_conv_desc (include/liso_conv.h)"""
    _fields_ = [("batch", _i), ("hi", _i), ("wi", _i), ("ci", _i), ("x_pix_stride", _i), ("ho", _i), ("wo", _i), ("co", _i),
                ("y_pix_stride", _i), ("y_ch_off", _i), ("hv", _i), ("wv", _i), ("isy", _i), ("isx", _i), ("osy", _i), ("osx", _i),
                ("n_classes", _i), ("class_tap_begin", _i * (CONV_MAX_CLASSES + 1)), ("class_ooy", _i * CONV_MAX_CLASSES),
                ("class_oox", _i * CONV_MAX_CLASSES), ("n_taps", _i), ("tap_dy", _i * CONV_MAX_TAPS), ("tap_dx", _i * CONV_MAX_TAPS),
                ("tap_w", _i * CONV_MAX_TAPS), ("w_taps", _i), ("mode", _i), ("out_f32", _i), ("in_relu", _i), ("out_relu", _i),
                ("in_affine_batch_stride", _i), ("wgrad_co", _i)]


class ConvPackJob(ctypes.Structure):
    """mirror of liso_conv_pack_job (include/liso_conv.h)"""
    _fields_ = [("src", _vp), ("dst", _vp), ("d0", _i), ("d1", _i), ("kh", _i), ("kw", _i), ("transposed", _i), ("for_dgrad", _i),
                ("mode", _i)]


class KnnGrid(ctypes.Structure):
    """mirror of liso_knn_grid (include/liso_slim.h)"""
    _fields_ = [("x_min", _f), ("y_min", _f), ("cell", _f), ("nx", _i), ("ny", _i), ("z_min", _f), ("z_cell", _f), ("nz", _i)]


class CenterLossCfg(ctypes.Structure):
    """mirror of liso_centerloss_cfg (include/liso_detector.h)"""
    _fields_ = [("batch", _i), ("h", _i), ("w", _i), ("res_x", _f), ("res_y", _f), ("z_min", _f), ("z_max", _f),
                ("sup_weight", _f), ("rot_reg_weight", _f)]


class NpLossCfg(ctypes.Structure):
    """mirror of liso_nploss_cfg (include/liso_slim.h)"""
    _fields_ = [("batch", _i), ("n", ctypes.c_long), ("n_b", ctypes.c_long), ("ext", _f * 4), ("fov_mode", _i), ("delta", _f)]


class SlimDecodeCfg(ctypes.Structure):
    """mirror of liso_slim_decode_cfg (include/liso_slim_decode.h)"""
    _fields_ = [("samples", _i), ("n", ctypes.c_long), ("h", _i), ("w", _i), ("logit_mode", _i * 4), ("static_flow_zero", _i),
                ("dynamic_flow_zero", _i), ("overwrite_flow", _i), ("overwrite_logits", _i), ("non_rigid", _i), ("use_static_aggr", _i),
                ("dyn_grad_scale", _f), ("ext_lo", ctypes.c_double * 2), ("ext_span", ctypes.c_double * 2)]


class SlimDecodeOut(ctypes.Structure):
    """mirror of liso_slim_decode_out: 11 float pointers + the flags pointer (None = not wanted)"""
    FIELDS = ("dis_logit", "dis", "logits", "probs", "staticness", "dynamicness", "groundness", "dyn_flow", "stat_flow", "agg_flow", "saf_flow")
    _fields_ = [(k, _vp) for k in FIELDS] + [("flags", _vp)]


class SlimNpLossCfg(ctypes.Structure):
    """mirror of liso_slim_nploss_cfg"""
    _fields_ = [("samples", _i), ("clouds", _i), ("n", ctypes.c_long), ("n_b", ctypes.c_long), ("ext", _f * 4), ("fov_mode", _i), ("delta", _f)]


class BoxPtsCfg(ctypes.Structure):
    """mirror of liso_boxpts_cfg (include/liso_tracking.h)"""
    _fields_ = [("batch", _i), ("n", ctypes.c_long), ("k", _i), ("point_stride", _i), ("precision", _i), ("dims_bloat", _f)]


class GruCfg(ctypes.Structure):
    """mirror of liso_gru_cfg (include/liso_slim.h)"""
    _fields_ = [("batch", _i), ("ch", _i), ("cx", _i), ("hw", ctypes.c_long)]


class UpsampleCfg(ctypes.Structure):
    """mirror of liso_upsample_cfg (include/liso_slim.h)"""
    _fields_ = [("n_it", _i), ("batch2", _i), ("dirs", _i), ("h", _i), ("w", _i), ("factor", _i), ("flow_scale", _f)]


class MineFilterCfg(ctypes.Structure):
    """mirror of liso_mine_filter_cfg (include/liso_box_mining.h)"""
    _fields_ = [("batch", _i), ("k", _i), ("min_points", _i), ("aspect_ratio_max", ctypes.c_double), ("max_box_len_m", ctypes.c_double),
                ("min_box_area_m2", ctypes.c_double), ("min_box_volume_m3", ctypes.c_double), ("park_invalid", _i)]


class TargetsCfg(ctypes.Structure):
    """mirror of liso_targets_cfg (include/liso_detector.h)"""
    _fields_ = [("batch", _i), ("n_boxes", _i), ("h", _i), ("w", _i), ("range_x", _f), ("range_y", _f)]


class DbscanCfg(ctypes.Structure):
    """mirror of liso_dbscan_cfg (include/liso_flow_cluster.h)"""
    _fields_ = [("batch", _i), ("gx", _i), ("gy", _i), ("window", _i), ("min_samples", _i), ("eps", _f), ("flow_weight", _f)]


class CorrCfg(ctypes.Structure):
    """mirror of liso_corr_cfg (include/liso_slim.h)"""
    _fields_ = [("batch", _i), ("h", _i), ("w", _i), ("dim", _i), ("levels", _i), ("radius", _i)]


class KabschCfg(ctypes.Structure):
    """mirror of liso_kabsch_cfg (include/liso_kabsch.h)"""
    _fields_ = [("batch", _i), ("n_points", _i), ("n_slots", _i), ("point_stride", _i), ("flow_stride", _i),
                ("slope", _f), ("scale_fg", _f), ("scale_bg", _f), ("softness", _i)]


class PillarCfg(ctypes.Structure):
    """mirror of liso_pillar_cfg (include/liso_pillars.h)"""
    _fields_ = [("x_min", _f), ("y_min", _f), ("z_min", _f), ("vx", _f), ("vy", _f), ("vz", _f), ("gx", _i),
                ("gy", _i), ("max_points", _i), ("max_voxels", _i), ("n_channels", _i)]


class _Everything:
    def __contains__(self, name):
        return True


class KernelTimer:
    """Optional HIP-event timing of individual C-ABI launches on the current PyTorch stream (used by bench.py for the
    `roofline` object).  Disabled unless bench.py turns it on; costs two event records per timed launch."""

    def __init__(self):
        self.enabled = set()
        self.events = {}
        self.units = {}
        self.bytes = {}
        self.weights = {}
        self.weight = 1.0

    def enable(self, name):
        self.enabled.add(name)
        self.events.setdefault(name, [])

    def enable_all(self):
        """time every `launch()` whatever its name (bench.py: find the kernel family with the largest share of a step)"""
        self.enabled = _Everything()

    def disable_all(self):
        self.enabled = set()

    def launch(self, name, fn, units=None, nbytes=None):
        """`units`: how many work items (queries, points, flops ...) this launch processes, for per-launch algorithmic bytes;
        `nbytes`: algorithmic HBM bytes of a launch whose `units` are flops (the partner of the PMC traffic figure)"""
        if name not in self.enabled:
            return fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        r = fn()
        b.record()
        self.events.setdefault(name, []).append((a, b))
        self.weights.setdefault(name, []).append(float(self.weight))
        if units is not None:
            self.units.setdefault(name, []).append(units)
        if nbytes is not None:
            self.bytes.setdefault(name, []).append(nbytes)
        return r

    def mean_units(self, name):
        u = self.units.get(name, [])
        return sum(u) / len(u) if u else None

    def durations_ms(self, name):
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in self.events.get(name, [])]

    def weighted_total_ms(self, name):
        """sum of the launch durations x the weight that was set when each was recorded (`weight` = share of the launch that belongs
        to one unit of work: 1 / batch for launches that serve a batch of units)"""
        return sum(d * w for d, w in zip(self.durations_ms(name), self.weights.get(name, [])))

    def reset(self):
        for k in self.events:
            self.events[k] = []
        self.units = {}
        self.bytes = {}
        self.weights = {}


TIMER = KernelTimer()


def _declare(l):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(l, name)  # AttributeError here == header/library mismatch, fail loudly
        fn.restype = res
        fn.argtypes = args
