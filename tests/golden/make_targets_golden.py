"""Generates tests/golden/targets_reference.npz from the reference's own python:
  liso.datasets.torch_dataset_commons.draw_heat_regression_maps  (CenterPoint target maps, :190-339)
which renders the per-box gaussians with liso.kabsch.kabsch_mask.batched_render_gaussian_kabsch_mask (:56-116).
The module's unrelated third-party imports that are absent from this image are stubbed with empty modules (no
arithmetic).  Run in the build container only:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_targets_golden.py
"""
import os
import sys
import types

import numpy as np

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")


class _Anything(types.ModuleType):
    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        return _Anything(self.__name__ + "." + k)

    def __call__(self, *a, **k):
        return lambda f: f


def import_with_stubs(import_fn, max_rounds=60):
    """run `import_fn`; whenever it fails with ModuleNotFoundError, install an empty stub for the missing module (names
    only, never numbers) and retry"""
    for _ in range(max_rounds):
        try:
            return import_fn()
        except ModuleNotFoundError as e:
            name = e.name
            parts = name.split(".")
            for i in range(1, len(parts) + 1):
                sub = ".".join(parts[:i])
                if sub not in sys.modules:
                    m = _Anything(sub)
                    m.__path__ = []
                    sys.modules[sub] = m
    raise RuntimeError("too many missing modules")


class _Cfg(dict):
    __getattr__ = dict.__getitem__


def cfg(d):
    return _Cfg({k: cfg(v) if isinstance(v, dict) else v for k, v in d.items()})


def main():
    def _imp():
        from liso.datasets.torch_dataset_commons import draw_heat_regression_maps
        from liso.kabsch.shape_utils import Shape
        return draw_heat_regression_maps, Shape

    draw_heat_regression_maps, Shape = import_with_stubs(_imp)

    g = np.random.default_rng(0)
    out = {}
    box_cfg = cfg({"dimensions_representation": {"method": "predict_abs_size"}, "rotation_representation": {"method": "vector"},
                   "position_representation": {"method": "local_relative_offset"}, "activations": {"dims": "softplus"}})
    for tag, (K, G, R) in {"a": (7, 32, 40.0), "b": (15, 128, 100.0), "c": (1, 16, 20.0)}.items():
        pos = np.concatenate([g.uniform(-0.45 * R, 0.45 * R, (K, 2)), g.uniform(-1.5, -0.5, (K, 1))], -1)
        if tag == "a":
            pos[1, :2] = pos[0, :2] + 0.8  # two overlapping boxes: exercises the hottest-object selection
        dims = np.stack([g.uniform(3.0, 5.0, K), g.uniform(1.5, 2.2, K), g.uniform(1.4, 1.8, K)], -1)
        rot = g.uniform(-np.pi, np.pi, (K, 1))
        boxes = Shape(pos=pos, dims=dims, rot=rot, probs=np.ones((K, 1)))
        maps = draw_heat_regression_maps(boxes, np.array([G, G]), np.array([R, R]), box_cfg)
        for k, v in (("pos", pos), ("dims", dims), ("rot", rot)):
            out[f"{tag}_box_{k}"] = v.astype(np.float64)
        out[f"{tag}_grid_range"] = np.array([G, R])
        for k in ("probs", "dims", "pos", "rot", "center_bool_mask"):
            out[f"{tag}_{k}"] = maps[k]
    np.savez_compressed(os.path.join(HERE, "targets_reference.npz"), **out)
    print({k: v.shape for k, v in out.items() if k.startswith("a_")})


if __name__ == "__main__":
    main()
