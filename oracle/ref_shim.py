"""TEST INFRASTRUCTURE ONLY -- build-container helper, never shipped, never imported by the product.

Makes the read-only Python reference at /root/reference importable in the build container, which lacks
torchvision / pytorch_lightning / cv2 / h5py / ....  Used exclusively by
``tests/golden/make_golden.py`` to emit golden input/output vectors; the reference itself never travels
to the GPU box and nothing in ``celldetection_amd`` imports this file.

Two kinds of stand-ins are installed into ``sys.modules`` *before* ``import celldetection``:

* permissive empty modules for packages whose symbols are only *named* on the CPN inference path
  (cv2, h5py, pynvml, skimage, albumentations, timm, ...);
* small *functional* stand-ins, written from the public documented behaviour of the third-party symbol,
  for the few third-party functions with arithmetic or structural meaning on the path
  (torchvision ``nms``/``IntermediateLayerGetter``/``FeaturePyramidNetwork.forward``/resnet block forwards/
  ``transforms.Normalize``; pytorch_lightning ``HyperparametersMixin``).

Vectors that flow through the functional stand-ins (NMS keep order, FPN top-down add, residual block
forward) are therefore pinned by THIS restatement of third-party behaviour, not by the reference repository;
DESIGN.md lists them as "third-party, unpinned by reference tests".
"""
import importlib.abc
import importlib.machinery
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = '/root/reference'

_STUB_ROOTS = ('torchvision', 'pytorch_lightning', 'lightning_fabric', 'cv2', 'h5py', 'pynvml', 'skimage',
               'albumentations', 'tifffile', 'seaborn', 'timm', 'segmentation_models_pytorch', 'imageio',
               'mpi4py', 'tensorboard', 'pytiff', 'matplotlib', 'PIL', 'pandas_stub_never')


class _AnyMeta(type):
    """Metaclass: unknown class-level attributes of a stub class are again stub classes."""

    def __getattr__(cls, item):
        if item.startswith('__') and item.endswith('__'):
            raise AttributeError(item)
        return _AnyMeta(item, (_Anything,), {})

    def __iter__(cls):
        return iter(())


class _Anything(metaclass=_AnyMeta):
    """Class returned for any attribute of a permissive stub module (usable as base class / decorator)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return _Anything()

    def __getattr__(self, item):
        if item.startswith('__') and item.endswith('__'):
            raise AttributeError(item)
        return _Anything()

    def __iter__(self):
        return iter(())


class _StubModule(types.ModuleType):
    __path__ = []
    __all__ = []

    def __getattr__(self, item):
        if item.startswith('__') and item.endswith('__'):
            raise AttributeError(item)
        cls = _AnyMeta(item, (_Anything,), {'__module__': self.__name__})
        setattr(self, item, cls)
        return cls


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split('.')[0]
        if root in _STUB_ROOTS and fullname not in sys.modules:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _StubModule(spec.name)

    def exec_module(self, module):
        pass


# ----------------------------------------------------------------------------------------------------------------
# functional stand-ins (own restatement of documented third-party behaviour)
# ----------------------------------------------------------------------------------------------------------------

class HyperparametersMixin:
    """pytorch_lightning.core.mixins.HyperparametersMixin: records ctor kwargs in ``hparams``."""

    def save_hyperparameters(self, *args, ignore=None, frame=None, logger=True):
        import inspect
        if not hasattr(self, '_hparams'):
            self._hparams = {}
            self._hparams_initial = {}
        self._log_hyperparams = False
        fr = inspect.currentframe().f_back
        try:
            info = inspect.getargvalues(fr)
            loc = info.locals
            names = [a for a in info.args if a != 'self']
            hp = {n: loc[n] for n in names if n in loc}
            if info.keywords and info.keywords in loc:
                hp.update(loc[info.keywords])
            if ignore:
                ignore = [ignore] if isinstance(ignore, str) else ignore
                hp = {k: v for k, v in hp.items() if k not in ignore}
            hp.pop('__class__', None)
            self._hparams.update(hp)
            self._hparams_initial.update(hp)
        finally:
            del fr

    def _set_hparams(self, hp):
        self._hparams = dict(hp)

    @property
    def hparams(self):
        if not hasattr(self, '_hparams'):
            self._hparams = {}
            self._hparams_initial = {}
        return self._hparams

    @property
    def hparams_initial(self):
        if not hasattr(self, '_hparams_initial'):
            self._hparams_initial = {}
        return self._hparams_initial


class LightningModule(nn.Module, HyperparametersMixin):
    """pytorch_lightning.LightningModule as far as ``LitCpn.forward_tiled`` / ``LitBase.forward`` touch it: an nn.Module that
    records hyper-parameters and knows the device of its parameters."""

    @property
    def device(self):
        try:
            return next(self.parameters()).device
        except StopIteration:
            return torch.device('cpu')

    global_rank = 0


class IntermediateLayerGetter(nn.ModuleDict):
    """torchvision.models._utils.IntermediateLayerGetter: run children in order, collect named outputs."""

    def __init__(self, model, return_layers):
        orig = dict(return_layers)
        return_layers = {str(k): str(v) for k, v in return_layers.items()}
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            if name in return_layers:
                del return_layers[name]
            if not return_layers:
                break
        super().__init__(layers)
        self.return_layers = {str(k): str(v) for k, v in orig.items()}

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


class ExtraFPNBlock(nn.Module):
    pass


class TvFeaturePyramidNetwork(nn.Module):
    """torchvision.ops.feature_pyramid_network.FeaturePyramidNetwork (forward: top-down nearest + add)."""

    def __init__(self, in_channels_list, out_channels, extra_blocks=None, norm_layer=None):
        super().__init__()
        self.inner_blocks = nn.ModuleList()
        self.layer_blocks = nn.ModuleList()
        self.extra_blocks = extra_blocks

    def get_result_from_inner_blocks(self, x, idx):
        n = len(self.inner_blocks)
        if idx < 0:
            idx += n
        return self.inner_blocks[idx](x)

    def get_result_from_layer_blocks(self, x, idx):
        n = len(self.layer_blocks)
        if idx < 0:
            idx += n
        return self.layer_blocks[idx](x)

    def forward(self, x):
        names = list(x.keys())
        x = list(x.values())
        last_inner = self.get_result_from_inner_blocks(x[-1], -1)
        results = [self.get_result_from_layer_blocks(last_inner, -1)]
        for idx in range(len(x) - 2, -1, -1):
            inner_lateral = self.get_result_from_inner_blocks(x[idx], idx)
            feat_shape = inner_lateral.shape[-2:]
            inner_top_down = F.interpolate(last_inner, size=feat_shape, mode='nearest')
            last_inner = inner_lateral + inner_top_down
            results.insert(0, self.get_result_from_layer_blocks(last_inner, idx))
        if self.extra_blocks is not None:
            results, names = self.extra_blocks(results, x, names)
        return OrderedDict([(k, v) for k, v in zip(names, results)])


class TvBackboneWithFPN(nn.Module):
    pass


class TvBasicBlock:
    expansion = 1

    def forward(self, x):
        identity = x
        out = self.conv1(x)
        out = self.bn1(out)
        out = self.relu(out)
        out = self.conv2(out)
        out = self.bn2(out)
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        out = self.relu(out)
        return out


class TvBottleneck:
    expansion = 4

    def forward(self, x):
        identity = x
        out = self.conv1(x)
        out = self.bn1(out)
        out = self.relu(out)
        out = self.conv2(out)
        out = self.bn2(out)
        out = self.relu(out)
        out = self.conv3(out)
        out = self.bn3(out)
        if self.downsample is not None:
            identity = self.downsample(x)
        out += identity
        out = self.relu(out)
        return out


class TvNormalize(nn.Module):
    def __init__(self, mean, std, inplace=False):
        super().__init__()
        self.mean, self.std = mean, std

    def forward(self, x):
        mean = torch.as_tensor(self.mean, dtype=x.dtype, device=x.device)
        std = torch.as_tensor(self.std, dtype=x.dtype, device=x.device)
        if mean.ndim == 1:
            mean = mean.view(-1, 1, 1)
        if std.ndim == 1:
            std = std.view(-1, 1, 1)
        return (x.clone() - mean) / std


class TvCompose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


def _upcast(t):
    if t.is_floating_point():
        return t if t.dtype in (torch.float32, torch.float64) else t.float()
    return t if t.dtype in (torch.int32, torch.int64) else t.int()


def box_area(boxes):
    boxes = _upcast(boxes)
    return (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])


def box_iou(boxes1, boxes2):
    area1 = box_area(boxes1)
    area2 = box_area(boxes2)
    lt = torch.max(boxes1[:, None, :2], boxes2[:, :2])
    rb = torch.min(boxes1[:, None, 2:], boxes2[:, 2:])
    wh = _upcast(rb - lt).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    union = area1[:, None] + area2 - inter
    return inter / union


def nms_cpu(dets, scores, iou_threshold):
    """Restatement of torchvision's public CPU NMS kernel (csrc/ops/cpu/nms_kernel.cpp semantics):
    areas=(x2-x1)*(y2-y1); order = scores.sort(descending, stable); greedy; suppress j when
    inter/(area_i+area_j-inter) > thr (NaN compares false); returns kept original indices in that order."""
    import numpy as np
    if dets.numel() == 0:
        return torch.empty((0,), dtype=torch.long)
    d = dets.detach().cpu().float().numpy()
    s = scores.detach().cpu().float()
    order = torch.sort(s, stable=True, descending=True).indices.numpy()
    x1, y1, x2, y2 = d[:, 0], d[:, 1], d[:, 2], d[:, 3]
    areas = (x2 - x1) * (y2 - y1)
    n = d.shape[0]
    suppressed = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(iou_threshold)
    with np.errstate(invalid='ignore', divide='ignore'):
        for _i in range(n):
            i = order[_i]
            if suppressed[i]:
                continue
            keep.append(i)
            rest = order[_i + 1:]
            xx1 = np.maximum(x1[i], x1[rest])
            yy1 = np.maximum(y1[i], y1[rest])
            xx2 = np.minimum(x2[i], x2[rest])
            yy2 = np.minimum(y2[i], y2[rest])
            w = np.maximum(np.float32(0), xx2 - xx1)
            h = np.maximum(np.float32(0), yy2 - yy1)
            inter = w * h
            ovr = inter / (areas[i] + areas[rest] - inter)
            suppressed[rest[ovr > thr]] = True
    return torch.as_tensor(np.asarray(keep, dtype='int64'))


def remove_small_boxes(boxes, min_size):
    ws, hs = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
    keep = (ws >= min_size) & (hs >= min_size)
    return torch.where(keep)[0]


def cv2_drawContours(image, contours, contourIdx, color, thickness=1, lineType=8, hierarchy=None, maxLevel=None,
                     offset=(0, 0)):
    """Functional stand-in for ``cv2.drawContours(..., thickness=-1, offset=...)`` as ``render_contour`` calls it
    (celldetection/data/cpn.py:252-254): fills ONE integer polygon into ``image`` in place and returns it.  The fill rule is
    ``labels_oracle.fill_polygon`` -- a restatement of OpenCV's published scan-line fill that could NOT be checked against
    cv2 here (third-party, unpinned, exactly like ``nms_cpu`` above).  What this stand-in makes importable -- and therefore
    pins -- is the reference's own loop around it (``contours2labels``, data/cpn.py:292-358)."""
    from labels_oracle import fill_polygon
    if thickness >= 0:
        raise NotImplementedError('stand-in: filled contours only (thickness=-1)')
    pts = np.asarray(contours[contourIdx]).reshape(-1, 2).astype(np.int64) + np.asarray(offset, np.int64)
    h, w = image.shape[:2]
    image[fill_polygon(pts, 0, 0, w, h)] = color
    return image


# --- third-party stand-ins of the slide preprocessing (celldetection/data/misc.py:156-161, celldetection_scripts/
# cpn_inference.py:196-222).  Written from the published behaviour of skimage / OpenCV / albumentations 1.x; none of them is in
# the image, so -- like ``nms_cpu`` and ``cv2_drawContours`` -- their arithmetic is third-party and UNPINNED.  What importing the
# reference's own ``normalize_percentile`` / ``preprocess`` through them pins is the reference's code around them: the percentile
# pair, np.percentile's interpolation, the clip / rescale expression, which step runs when and in which order.
CV2_COLOR_RGB2GRAY, CV2_COLOR_RGBA2GRAY, CV2_COLOR_GRAY2RGB = 7, 11, 8  # (OpenCV's enum values; only compared, never computed on)


def skimage_img_as_ubyte(image, force_copy=False):
    """skimage.util.img_as_ubyte for the float images ``normalize_percentile`` hands it: rint(image * 255) clipped to 0..255,
    computed in the float type of the input (float64 here)."""
    image = np.asarray(image)
    if image.dtype == np.uint8:
        return image.copy() if force_copy else image
    if image.dtype.kind != 'f':
        raise NotImplementedError('stand-in: float (and uint8) images only')
    if image.size and (image.min() < -1. or image.max() > 1.):
        raise ValueError('Images of type float must be between -1 and 1.')
    out = np.multiply(image, 255, dtype=image.dtype if image.dtype.itemsize >= 4 else np.float32)
    np.rint(out, out=out)
    np.clip(out, 0, 255, out=out)
    return out.astype(np.uint8)


def cv2_cvtColor(src, code):
    """cv2.cvtColor for the three 8-bit conversions of the script: OpenCV's fixed-point luma (14 fractional bits) and the
    channel replication of GRAY2RGB.  Other depths raise, as cv2 does for the float64 array of the script's 2-channel branch."""
    src = np.asarray(src)
    if src.dtype != np.uint8:
        raise TypeError(f'stand-in cv2.cvtColor: unsupported depth {src.dtype} (cv2: "Unsupported depth of input image")')
    if code in (CV2_COLOR_RGB2GRAY, CV2_COLOR_RGBA2GRAY):
        assert src.ndim == 3 and src.shape[-1] == (3 if code == CV2_COLOR_RGB2GRAY else 4)
        x = src.astype(np.int64)
        return ((x[..., 0] * 4899 + x[..., 1] * 9617 + x[..., 2] * 1868 + (1 << 13)) >> 14).astype(np.uint8)
    if code == CV2_COLOR_GRAY2RGB:
        assert src.ndim == 2
        return np.repeat(src[..., None], 3, -1)
    raise NotImplementedError(f'stand-in cv2.cvtColor: code {code}')


def cv2_LUT(src, lut):
    return np.asarray(lut)[np.asarray(src)]


def alb_gamma_transform(img, gamma):
    """albumentations.augmentations.functional.gamma_transform (1.x): uint8 -> 256-entry table, floats -> power."""
    if img.dtype == np.uint8:
        table = (np.arange(0, 256.0 / 255, 1.0 / 255) ** gamma) * 255
        return cv2_LUT(img, table.astype(np.uint8))
    return np.power(img, gamma)


def alb_brightness_contrast_adjust(img, alpha=1, beta=0, beta_by_max=False):
    """albumentations.augmentations.functional.brightness_contrast_adjust (1.x), uint8 branch."""
    if img.dtype != np.uint8:
        raise NotImplementedError('stand-in: uint8 images only')
    lut = np.arange(0, 256).astype('float32')
    if alpha != 1:
        lut *= alpha
    if beta != 0:
        lut += beta * 255 if beta_by_max else (alpha * beta) * np.mean(img)
    return cv2_LUT(img, np.clip(lut, 0, 255).astype(np.uint8))


_INSTALLED = False


def install():
    """Install stubs + functional stand-ins and put the reference on sys.path. Idempotent."""
    global _INSTALLED
    if _INSTALLED:
        return
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    finder = _StubFinder()
    sys.meta_path.insert(0, finder)

    def mod(name):
        m = _StubModule(name)
        sys.modules[name] = m
        parent, _, child = name.rpartition('.')
        if parent and parent in sys.modules:
            setattr(sys.modules[parent], child, m)
        return m

    # pytorch_lightning
    pl = mod('pytorch_lightning')
    mod('pytorch_lightning.core')
    mix = mod('pytorch_lightning.core.mixins')
    mix.HyperparametersMixin = HyperparametersMixin
    pl.LightningModule = LightningModule
    pl.Callback = object
    lf = mod('lightning_fabric')
    mod('lightning_fabric.utilities')
    rz = mod('lightning_fabric.utilities.rank_zero')
    rz.rank_zero_only = lambda f: f

    # torchvision
    mod('torchvision')
    mod('torchvision.models')
    u = mod('torchvision.models._utils')
    u.IntermediateLayerGetter = IntermediateLayerGetter
    r = mod('torchvision.models.resnet')
    r.BasicBlock = TvBasicBlock
    r.Bottleneck = TvBottleneck
    mod('torchvision.models.detection')
    bu = mod('torchvision.models.detection.backbone_utils')
    bu.BackboneWithFPN = TvBackboneWithFPN
    mod('torchvision.ops')
    fp = mod('torchvision.ops.feature_pyramid_network')
    fp.FeaturePyramidNetwork = TvFeaturePyramidNetwork
    fp.ExtraFPNBlock = ExtraFPNBlock
    bx = mod('torchvision.ops.boxes')
    bx.box_area, bx.box_iou, bx._upcast, bx.nms, bx.remove_small_boxes = \
        box_area, box_iou, _upcast, nms_cpu, remove_small_boxes
    sys.modules['torchvision.ops'].nms = nms_cpu
    sys.modules['torchvision.ops'].boxes = bx
    tr = mod('torchvision.transforms')
    tr.Normalize = TvNormalize
    tr.Compose = TvCompose

    # cv2: everything a stub except the one call on the label path
    cv = mod('cv2')
    cv.drawContours = cv2_drawContours
    cv.cvtColor, cv.LUT = cv2_cvtColor, cv2_LUT
    cv.COLOR_RGB2GRAY, cv.COLOR_RGBA2GRAY, cv.COLOR_GRAY2RGB = CV2_COLOR_RGB2GRAY, CV2_COLOR_RGBA2GRAY, CV2_COLOR_GRAY2RGB

    # skimage / albumentations: the calls of the slide preprocessing
    sk = mod('skimage')
    sk.img_as_ubyte = skimage_img_as_ubyte
    mod('albumentations')
    mod('albumentations.augmentations')
    af = mod('albumentations.augmentations.functional')
    af.gamma_transform, af.brightness_contrast_adjust = alb_gamma_transform, alb_brightness_contrast_adjust

    # dispatcher op torch.ops.torchvision.nms
    try:
        lib = torch.library.Library('torchvision', 'DEF')
        lib.define('nms(Tensor dets, Tensor scores, float iou_threshold) -> Tensor')
        lib.impl('nms', nms_cpu, 'CPU')
        install._lib = lib
    except Exception as e:  # already defined
        print('torchvision::nms define failed:', e)
    _INSTALLED = True


def import_reference():
    install()
    import celldetection as cd
    return cd
