"""liso_amd.install_as(): the reference's import paths resolve to this package's modules (one module object under both names)."""
import subprocess
import sys
import os

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

CODE = r'''
import sys
import liso_amd
liso_amd.install_as("liso")
import liso
import liso.utils.nms_iou as a
import liso_amd.utils.nms_iou as b
assert a is b and liso is liso_amd
from liso.kabsch.shape_utils import Shape
from liso_amd.kabsch.shape_utils import Shape as Shape2
assert Shape is Shape2
from liso.networks.flow_cluster_detector.flow_cluster_detector import FlowClusterDetector
from liso.slim.model.slim import SLIM
from liso.networks.centerpoint.rpn import RPN
import liso.torch_symm_ortho, liso.weighted_pc_alignment
from iou3d_nms.iou3d_nms_utils import nms_gpu, boxes_iou_bev
import iou3d_nms_cuda
assert iou3d_nms_cuda is sys.modules["liso_amd.iou3d_nms_cuda"]
try:
    import liso.visu.utils
except ModuleNotFoundError as e:
    assert "hot path" in str(e), e
else:
    raise AssertionError("liso.visu must not resolve")
liso_amd.install_as("liso")  # idempotent
print("ALIAS_OK")
'''


def test_install_as_resolves_reference_import_paths():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0 and "ALIAS_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_install_as_refuses_to_shadow_another_package(tmp_path):
    (tmp_path / "liso").mkdir()
    (tmp_path / "liso" / "__init__.py").write_text("X = 1\n")
    code = "import liso, liso_amd\ntry:\n    liso_amd.install_as('liso')\nexcept RuntimeError as e:\n    print('REFUSED')\n"
    env = dict(os.environ, PYTHONPATH=str(tmp_path) + os.pathsep + ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path), timeout=300)
    assert "REFUSED" in r.stdout, (r.stdout, r.stderr[-2000:])
