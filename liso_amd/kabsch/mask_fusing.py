"""mirror of liso/kabsch/mask_fusing.py:4-6"""
import torch


def fuse_masks_screen(masks, *, dim, keepdim=False):
    return 1.0 - torch.prod(1.0 - masks, dim=dim, keepdim=keepdim)
