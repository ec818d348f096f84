"""alias of liso/weighted_pc_alignment (stand-alone duplicate in the reference, weighted_pc_alignment.py:54-141)"""
from liso_amd.slim.slim_loss.weighted_pc_alignment import EPSILON, weighted_pc_alignment  # noqa: F401
