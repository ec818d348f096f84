"""liso_amd -- MI355X-native (gfx950) implementation of LISO's data-parallel hot path.

Module tree mirrors the reference's (`liso.*`, `iou3d_nms.*`) for the rows of SURVEY.md section 8 only.
Kernels: liso_amd/csrc/*.hip behind the C ABI of include/*.h (libliso_hip.so, loaded by liso_amd._lib).
"""
__version__ = "0.1.0"


def install_as(name="liso", iou3d_name="iou3d_nms"):
    """Make this package importable under the reference's names: after `liso_amd.install_as()`, `import liso.utils.nms_iou`,
    `from liso.networks.flow_cluster_detector.flow_cluster_detector import FlowClusterDetector`, `import iou3d_nms_cuda`,
    `from iou3d_nms.iou3d_nms_utils import nms_gpu` ... resolve to the MI355X implementations, so a training loop written against
    the reference runs unchanged.  Only the hot-path modules exist here: importing anything else from `liso.*`
    raises ModuleNotFoundError with that explanation.  Refuses to shadow a real `liso` package that is already imported."""
    import importlib
    import importlib.abc
    import importlib.machinery
    import sys

    if name in sys.modules and getattr(sys.modules[name], "__liso_amd_alias__", None) is None and sys.modules[name] is not sys.modules[__name__]:
        raise RuntimeError(f"a different package named {name!r} is already imported; refusing to shadow it")
    me = sys.modules[__name__]
    me.__liso_amd_alias__ = True

    class _Alias(importlib.abc.MetaPathFinder, importlib.abc.Loader):
        """`<name>.x.y` -> the module object of `liso_amd.x.y` (one object under both names: isinstance / state_dict keys agree)"""

        def find_spec(self, fullname, path=None, target=None):
            if fullname == name or fullname.startswith(name + "."):
                return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
            return None

        def create_module(self, spec):
            real = __name__ + spec.name[len(name):]
            try:
                return importlib.import_module(real)
            except ModuleNotFoundError as e:
                if e.name == real:
                    raise ModuleNotFoundError(f"{spec.name}: not part of the MI355X hot path (liso_amd mirrors SURVEY.md section 8 only)",
                                              name=spec.name) from None
                raise

        def exec_module(self, module):
            pass

    if not any(type(f).__name__ == "_Alias" and getattr(f, "_liso_name", None) == name for f in sys.meta_path):
        finder = _Alias()
        finder._liso_name = name
        sys.meta_path.insert(0, finder)
    sys.modules[name] = me
    # the reference's top-level extension module and its python wrapper package
    sys.modules.setdefault("iou3d_nms_cuda", importlib.import_module(__name__ + ".iou3d_nms_cuda"))
    sys.modules.setdefault(iou3d_name, importlib.import_module(__name__ + ".iou3d_nms"))
    sys.modules.setdefault(iou3d_name + ".iou3d_nms_utils", importlib.import_module(__name__ + ".iou3d_nms.iou3d_nms_utils"))
    return me
