import numpy as np
import torch

from .._lib import OdetError


def current_device():
    if not torch.cuda.is_available():
        raise OdetError('no HIP device visible: tf_eager_object_detection_amd has no CPU path')
    return torch.device('cuda', torch.cuda.current_device())


def to_gpu_f32(x, like=None):
    """numpy / list / torch (any device) -> contiguous float32 GPU tensor."""
    if isinstance(x, torch.Tensor):
        if x.is_cuda:
            return x.float().contiguous() if x.dtype != torch.float32 else x.contiguous()
        dev = like.device if like is not None else current_device()
        return x.to(dev, torch.float32).contiguous()
    dev = like.device if like is not None else current_device()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32)).to(dev)
