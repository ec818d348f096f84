"""liso_amd -- MI355X-native (gfx950) implementation of LISO's data-parallel hot path.

Module tree mirrors the reference's (`liso.*`, `iou3d_nms.*`) for the rows of SURVEY.md section 8 only.
Kernels: liso_amd/csrc/*.hip behind the C ABI of include/*.h (libliso_hip.so, loaded by liso_amd._lib).
"""
__version__ = "0.1.0"
