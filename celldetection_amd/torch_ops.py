"""PyTorch custom-op registration (``torch.library``) of the HIP kernels behind the C ABI.

``BASELINE.json.north_star``: the kernels are "called from Python via PyTorch-ROCm custom ops (thin C-ABI)".  The ops of
``celldetection_amd.ops`` are registered in the ``cpn_hip`` namespace so that they are addressable as
``torch.ops.cpn_hip.<name>`` (dispatcher-visible, GPU only -- there is no CPU kernel, a CPU tensor raises), and
``install_torchvision_nms()`` provides ``torch.ops.torchvision.nms`` on ROCm systems without torchvision, which is the
operator the reference calls directly (celldetection/ops/cpn.py:181,211,216,223; celldetection_scripts/cpn_inference.py:407,
426; celldetection/models/lightning_cpn.py:167): with it those call sites run on the HIP NMS without being edited.
"""
import torch
from torch import Tensor

from . import ops as _ops

__all__ = ['install_torchvision_nms', 'NAMESPACE']

NAMESPACE = 'cpn_hip'
_lib_keepalive = []


@torch.library.custom_op(f'{NAMESPACE}::nms', mutates_args=(), device_types='cuda')
def nms(boxes: Tensor, scores: Tensor, iou_threshold: float) -> Tensor:
    """torchvision.ops.nms semantics on the HIP kernels (dense bit mask or spatially binned, by size)."""
    return _ops.nms(boxes, scores, iou_threshold)


@nms.register_fake
def _(boxes, scores, iou_threshold):
    return boxes.new_empty((torch.library.get_ctx().new_dynamic_size(),), dtype=torch.int64)


@torch.library.custom_op(f'{NAMESPACE}::fouriers2contours', mutates_args=(), device_types='cuda')
def fouriers2contours(fourier: Tensor, locations: Tensor, samples: int) -> Tensor:
    """celldetection/ops/cpn.py:44-95 (default sampling): [..., order, 4], [..., 2] -> [..., samples, 2]."""
    return _ops.fouriers2contours(fourier, locations, samples)[0]


@fouriers2contours.register_fake
def _(fourier, locations, samples):
    return fourier.new_empty(fourier.shape[:-2] + (samples, 2), dtype=torch.float32)


@torch.library.custom_op(f'{NAMESPACE}::local_refinement', mutates_args=(), device_types='cuda')
def local_refinement(contours: Tensor, refinement: Tensor, num_loops: int, b: Tensor, num_buckets: int) -> Tensor:
    """celldetection/models/cpn.py:63-85."""
    return _ops.local_refinement(contours, refinement, num_loops, b, num_buckets=num_buckets)


@local_refinement.register_fake
def _(contours, refinement, num_loops, b, num_buckets):
    return torch.empty_like(contours, dtype=torch.float32)


@torch.library.custom_op(f'{NAMESPACE}::remove_border_contours', mutates_args=(), device_types='cuda')
def remove_border_contours(contours: Tensor, height: int, width: int, padding: float, sides: int, offset_x: float,
                           offset_y: float) -> Tensor:
    """celldetection/ops/cpn.py:258-290; sides: bit0 top, bit1 right, bit2 bottom, bit3 left."""
    return _ops.remove_border_contours(contours, (height, width), padding, top=bool(sides & 1), right=bool(sides & 2),
                                       bottom=bool(sides & 4), left=bool(sides & 8), offsets=(offset_x, offset_y))


@remove_border_contours.register_fake
def _(contours, height, width, padding, sides, offset_x, offset_y):
    return contours.new_empty((contours.shape[0],), dtype=torch.bool)


@torch.library.custom_op(f'{NAMESPACE}::box_votes', mutates_args=(), device_types='cuda')
def box_votes(boxes: Tensor, thresh: float) -> Tensor:
    """get_iou_voting, celldetection/ops/boxes.py:52-58."""
    from . import _lib
    bx = boxes.contiguous().float()
    votes = torch.empty((bx.shape[0],), dtype=torch.float32, device=bx.device)
    _lib.check(_lib.load().cpn_box_votes(_lib.ptr(bx), int(bx.shape[0]), float(thresh), _lib.ptr(votes),
                                         _lib.stream_ptr()), 'box_votes')
    return votes


@box_votes.register_fake
def _(boxes, thresh):
    return boxes.new_empty((boxes.shape[0],), dtype=torch.float32)


def install_torchvision_nms(force: bool = False) -> bool:
    """Defines ``torchvision::nms`` (schema of torchvision's operator) with the HIP implementation for GPU tensors when
    torchvision is not installed, so that the reference's ``torch.ops.torchvision.nms(...)`` call sites dispatch to
    libcpn_hip.so unchanged.  With torchvision present nothing is touched unless ``force`` (then the CUDA/HIP kernel of
    the existing operator is overridden).  Returns True when the HIP implementation is active."""
    try:
        has = hasattr(torch.ops.torchvision, 'nms') and torch.ops.torchvision.nms is not None
        if has:
            torch.ops.torchvision.nms.default  # resolves only when the operator really exists
    except (AttributeError, RuntimeError):
        has = False
    impl = lambda dets, scores, iou_threshold: _ops.nms(dets, scores, float(iou_threshold))
    if not has:
        lib = torch.library.Library('torchvision', 'DEF')
        lib.define('nms(Tensor dets, Tensor scores, float iou_threshold) -> Tensor')
        lib.impl('nms', impl, 'CUDA')
        _lib_keepalive.append(lib)
        return True
    if force:
        lib = torch.library.Library('torchvision', 'IMPL')
        lib.impl('nms', impl, 'CUDA', allow_override=True) if 'allow_override' in lib.impl.__code__.co_varnames \
            else lib.impl('nms', impl, 'CUDA')
        _lib_keepalive.append(lib)
        return True
    return False
