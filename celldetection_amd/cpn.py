"""Drop-in ``cd.models.CPN`` inference modules running on the MI355X HIP engine.

Mirrors the reference's public surface for the inference path (celldetection/models/cpn.py:287-439,561-734):
constructor signatures of the ``Cpn<Backbone>`` classes, mutable attributes (``score_thresh``, ``nms_thresh``,
``samples``, ``order``, ``refinement_iterations`` ... honoured at run time), ``state_dict()`` key names/shapes
(reference checkpoints load unchanged), ``hparams`` and the ``forward(inputs, targets=None, nms=True, **kwargs)``
output ``OrderedDict`` of per-image lists.  Training (``compute_loss``, cpn.py:441-559) is out of scope.

The conv stack runs through the native graph executor of libcpn_hip.so (bf16 NHWC, MFMA), the post-processing
through the fused decode / NMS kernels; PyTorch only provides device memory, streams and indexing glue.
"""
import os
import warnings
from collections import OrderedDict, deque
from ctypes import c_void_p

import torch
import torch.nn as nn

from . import _lib, graph, ops

__all__ = ['CPN']


class _Container(nn.Module):
    """Anonymous node of the parameter tree (mirrors the reference's module hierarchy for state_dict keys)."""

    def forward(self, *a, **k):
        raise RuntimeError('parameter container; the compute graph runs in the HIP engine')


def _register(root: nn.Module, key: str, shape, kind: str):
    parts = key.split('.')
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Container())
        m = m._modules[p]
    leaf = parts[-1]
    if kind == 'param':
        t = torch.empty(shape)
        if len(shape) >= 2:
            nn.init.kaiming_uniform_(t, a=1)
        elif leaf == 'weight':
            t.fill_(1.)
        else:
            t.zero_()
        m.register_parameter(leaf, nn.Parameter(t, requires_grad=False))
    elif kind == 'long':
        m.register_buffer(leaf, torch.zeros(shape, dtype=torch.long))
    else:
        if leaf == 'running_var':
            t = torch.ones(shape)
        elif key == 'order_weights':
            x = torch.arange(shape[0]).float()
            spread = max(shape[0] - 1, 1)
            t = (1 + 4 * (1 - (x / spread).clamp(0., 1.)) ** 2)[:, None]  # celldetection/ops/cpn.py:230-235
        else:
            t = torch.zeros(shape)
        m.register_buffer(leaf, t)


class _Engine:
    """Packed weights + native plan for one device."""

    def __init__(self, plan: graph.Plan, state_dict, device, precision: str = 'bf16', act_scales=None):
        lib = _lib.load()
        self.device = device
        self.precision = precision
        if precision == 'fp8':
            self.tens, self.ops_desc, self.wblob, self.bblob, _, _ = graph.pack(plan, state_dict, device, 'fp8',
                                                                               act_scales=act_scales)
        else:
            self.tens, self.ops_desc, self.wblob, self.bblob = graph.pack(plan, state_dict, device, precision)
        handle = c_void_p()
        code = dict(bf16=_lib.PRECISION_BF16, fp32=_lib.PRECISION_F32, fp8=_lib.PRECISION_FP8)[precision]
        _lib.check(lib.cpn_plan_create(handle, self.tens, len(self.tens), self.ops_desc, len(self.ops_desc),
                                       _lib.ptr(self.wblob), self.wblob.numel() * self.wblob.element_size(),
                                       _lib.ptr(self.bblob), self.bblob.numel(), code), 'plan_create')
        self.handle = handle
        self.plan = plan
        self._ws = None
        self.sparse_requested = False
        self.sparse = plan.meta.get('sparse_heads') if precision == 'bf16' else None  # dict(ops=(i, j), src=tensor id)
        self._ws_ring, self._ws_turn = [None, None], 0
        self.last_sparse = None
        self._graphs, self._graph_seen, self._graph_broken = {}, {}, False  # hipGraph slots per run shape (see run())

    def __del__(self):
        handle, self.handle = getattr(self, 'handle', None), None
        if handle:
            try:
                _lib.load().cpn_plan_destroy(handle)
            except (RuntimeError, OSError, AttributeError, TypeError):  # interpreter shutdown: library already unloaded
                pass

    def workspace(self, n, h, w):
        need = int(_lib.load().cpn_plan_workspace_bytes(self.handle, n, h, w))
        if need < 0:
            _lib.check(need, 'plan_workspace_bytes')
        if self.sparse:  # the head source of run i is read by the post-processing of run i while run i+1 may already be
            self._ws_turn ^= 1  # enqueued (forward_pipelined): two arenas in turn
            ws = self._ws_ring[self._ws_turn]
            if ws is None or ws.numel() < need:
                ws = self._ws_ring[self._ws_turn] = torch.empty(need, dtype=torch.uint8, device=self.device)
            return ws, need
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    def output_size(self, h, w, out_index):
        """(h, w) of the external output ``out_index`` (``_lib.OUT_*``) for an ``h`` x ``w`` input."""
        from ctypes import c_int32
        oh, ow = c_int32(0), c_int32(0)
        _lib.check(_lib.load().cpn_plan_output_dims(self.handle, h, w, out_index, oh, ow), 'plan_output_dims')
        return int(oh.value), int(ow.value)

    def max_batch(self, n, h, w):
        """Images per graph run: the kernels address activation tensors through 2^31-byte buffer descriptors / 32-bit
        element offsets, so a batch whose largest tensor would exceed that is split (e.g. 8 x 256 ch x 1024^2 in front of
        an FPN refinement head)."""
        per_image = int(_lib.load().cpn_plan_max_tensor_elements(self.handle, h, w))
        if per_image <= 0:
            _lib.check(per_image or _lib.E_INVALID, 'plan_max_tensor_elements')
        limit = (2 ** 31 - 1) // (2 if self.precision == 'bf16' else 1)  # sources: 2^31 bytes (fp32 path: elements)
        cap = max(1, min(n, limit // per_image))
        runs = -(-n // cap)       # balanced sub-batches (8 tiles with room for 7 run as 4 + 4, not 7 + 1)
        return -(-n // runs)

    def executed_flops(self, n, h, w):
        """2*MAC FLOPs the MFMA loops execute for a batch of n (the sum over the sub-batches the engine splits it into)."""
        lib, nb = _lib.load(), self.max_batch(n, h, w)
        total = (n // nb) * float(lib.cpn_plan_executed_flops(self.handle, nb, h, w))
        if n % nb:
            total += float(lib.cpn_plan_executed_flops(self.handle, n % nb, h, w))
        return total

    def profile(self, x: torch.Tensor, order_total: int, refinement: bool):
        """Per-op timing of one conv-graph execution: list of dicts(op, name, ms, gflop_executed)."""
        from ctypes import c_double, c_float
        lib = _lib.load()
        nops = lib.cpn_plan_num_ops(self.handle)
        ms, fl = (c_float * nops)(), (c_double * nops)()
        self.run(x, order_total, refinement, _timed=(ms, fl))
        out = []
        from ctypes import c_int32, c_int64
        n, _, h, w = x.shape
        for i, op in enumerate(self.plan.ops):
            oh, ow = c_int32(0), c_int32(0)
            if op.get('dst') is not None:  # output size of the op (tile selection of the conv kernel depends on it)
                off, cs = c_int64(0), c_int32(0)
                if lib.cpn_plan_tensor_info(self.handle, n, h, w, int(op['dst']), off, oh, ow, cs) != 0:
                    oh, ow = c_int32(0), c_int32(0)
            elif op.get('out_index') is not None:
                lib.cpn_plan_output_dims(self.handle, h, w, int(op['out_index']), oh, ow)
            out.append(dict(index=i, op=op['op'], name=op.get('w', ''), ms=float(ms[i]), gflop=float(fl[i]) / 1e9,
                            k=op.get('k'), cin=op.get('cin'), cout=op.get('cout'), groups=op.get('groups'),
                            stride=op.get('stride'), out_h=int(oh.value), out_w=int(ow.value)))
        return out

    def activation_absmax(self, x: torch.Tensor, order_total: int, refinement: bool):
        """Calibration run (bf16 plans): max |value| of every activation tensor of the graph for the batch ``x``."""
        absmax = torch.zeros(len(self.tens), dtype=torch.float32, device=self.device)
        self.run(x, order_total, refinement, _absmax=absmax)
        return absmax.cpu()

    # ---- hipGraph replay of the conv graph -------------------------------------------------------------------------
    # One conv-graph execution is ~120 dependent launches, most of them tens of microseconds long: replaying them as ONE
    # hipGraph removes the host-side launch gaps (profiles/r03_graph_probe.txt: 29.68 -> 29.29 ms for the BASELINE
    # configs[2] batch).  A shape is captured the second time in a row it is seen, into GRAPH_SLOTS instances that are used in turn:
    # each owns its input copy, arena, head maps and range flag, so the post-processing of run i (second stream of
    # forward_pipelined) can still read slot i while run i+1 replays another slot.  CPN_HIP_GRAPH=0 disables it.
    GRAPH_SLOTS = 3
    GRAPH_SHAPES = 4  # captured shapes kept (least recently used evicted)

    def _alloc_outputs(self, n, h, w, order_total, refinement, gated):
        f32 = dict(dtype=torch.float32, device=self.device)
        meta = self.plan.meta
        # head grids: any H x W (sizes propagated by the executor; heads may read different features / use a stride)
        scores = torch.empty((n, meta.get('score_channels', 1)) + self.output_size(h, w, _lib.OUT_SCORES), **f32)
        locations = None if gated else torch.empty((n, 2) + self.output_size(h, w, _lib.OUT_LOCATIONS), **f32)
        fourier = None if gated else torch.empty((n, 4 * order_total) + self.output_size(h, w, _lib.OUT_FOURIER), **f32)
        ref = torch.empty((n, 2 * meta.get('refinement_buckets', 1)) + self.output_size(h, w, _lib.OUT_REFINEMENT),
                          **f32) if refinement else None
        unc = torch.empty((n, 4) + self.output_size(h, w, _lib.OUT_UNCERTAINTY), **f32) \
            if meta.get('uncertainty_head') else None
        return scores, locations, fourier, ref, unc

    def _launch(self, x, dt, h, w, ws, need, outputs, flag, nb, _timed=None, _absmax=None):
        lib = _lib.load()
        n = x.shape[0]
        for i0 in range(0, n, nb):
            m = min(nb, n - i0)
            xi = x[i0:i0 + m]
            outs = (c_void_p * _lib.NUM_OUTPUTS)(*[0 if t is None else t[i0:i0 + m].data_ptr() for t in outputs])
            if _absmax is not None:
                _lib.check(lib.cpn_plan_run_stats(self.handle, _lib.ptr(xi), dt, m, h, w, _lib.ptr(ws), need, outs,
                                                  _lib.ptr(flag), _lib.ptr(_absmax), _lib.stream_ptr()),
                           'plan_run_stats')
            elif _timed is not None:
                _lib.check(lib.cpn_plan_run_timed(self.handle, _lib.ptr(xi), dt, m, h, w, _lib.ptr(ws), need, outs,
                                                  _lib.ptr(flag), _lib.stream_ptr(), _timed[0], _timed[1]),
                           'plan_run_timed')
            else:
                _lib.check(lib.cpn_plan_run(self.handle, _lib.ptr(xi), dt, m, h, w, _lib.ptr(ws), need, outs,
                                            _lib.ptr(flag), _lib.stream_ptr()), 'plan_run')

    def _graph_slot(self, key, x, dt, order_total, refinement, gated):
        """The hipGraph instance to replay for this run, or None (shape seen for the first time / graphs disabled)."""
        if os.environ.get('CPN_HIP_GRAPH', '1') == '0' or self._graph_broken:
            return None
        st = self._graphs.get(key)
        last, self._graph_last_key = getattr(self, '_graph_last_key', None), key
        if st is None:
            # capture a shape only when it is seen twice IN A ROW (a steady loop).  A shape that merely recurs -- the ragged last
            # batch of every slide of a tile loop -- would pay an eager run + a capture + three arenas for one replay per slide
            # (round 3: a 4096^2 slide of 121 = 7 x 16 + 9 tiles captured its 9-tile batch inside the timed loop)
            self._graph_seen[key] = self._graph_seen.get(key, 0) + 1 if last == key else 1
            if self._graph_seen[key] < 2:
                return None
            while len(self._graphs) >= self.GRAPH_SHAPES:  # evict the least recently used shape (frees its arenas)
                self._graphs.pop(next(iter(self._graphs)))
            st = self._graphs[key] = dict(slots=[], turn=0)
        else:
            self._graphs[key] = self._graphs.pop(key)  # most recently used last
        if len(st['slots']) < self.GRAPH_SLOTS:
            n, _, h, w = x.shape
            cur = torch.cuda.current_stream(self.device)
            side = torch.cuda.Stream(self.device)
            side.wait_stream(cur)
            slot = None
            try:
                # a slot owns its input copy, arena and head maps: on a memory-tight GPU this allocation may fail -- like a
                # failed capture that only costs the optimisation (eager launches, the shape's slots released)
                need = int(_lib.load().cpn_plan_workspace_bytes(self.handle, n, h, w))
                slot = dict(x=torch.empty_like(x), ws=torch.empty(max(need, 1), dtype=torch.uint8, device=self.device),
                            need=need, outputs=self._alloc_outputs(n, h, w, order_total, refinement, gated),
                            flag=torch.zeros(1, dtype=torch.int32, device=self.device))
                with torch.cuda.stream(side):
                    slot['x'].copy_(x)
                    # eager run on the capture stream first: per-device function attributes (dynamic LDS limit) are set on
                    # the first launch of every kernel instantiation, which must not happen inside a capture
                    self._launch(slot['x'], dt, h, w, slot['ws'], need, slot['outputs'], slot['flag'], n)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, stream=side, capture_error_mode='relaxed'):
                        slot['flag'].zero_()
                        self._launch(slot['x'], dt, h, w, slot['ws'], need, slot['outputs'], slot['flag'], n)
                slot['graph'] = g
            except Exception as e:  # capture is an optimisation: fall back to eager launches, loudly
                warnings.warn(f'hipGraph slot of the conv graph not available ({type(e).__name__}: {e}); using eager launches',
                              RuntimeWarning)
                self._graph_broken = True  # (no retry per forward: a model that ran eagerly keeps running eagerly)
                slot = None
                self._graphs.clear()       # releases the arenas / head maps of every captured shape
                if isinstance(e, torch.cuda.OutOfMemoryError):
                    torch.cuda.empty_cache()
                return None
            finally:
                cur.wait_stream(side)
            st['slots'].append(slot)
            st['turn'] = len(st['slots']) % self.GRAPH_SLOTS
            return slot
        slot = st['slots'][st['turn']]
        st['turn'] = (st['turn'] + 1) % self.GRAPH_SLOTS
        return slot

    def run(self, x: torch.Tensor, order_total: int, refinement: bool, _timed=None, _absmax=None, static_ok=False):
        """x: [N,C,H,W] float32 in [0,1] (or uint8) on self.device -> (scores, locations, refinement, fourier, flag).
        ``static_ok``: the caller consumes the outputs before this engine runs GRAPH_SLOTS more times (the forward paths
        do); otherwise outputs that live in a hipGraph slot are cloned."""
        lib = _lib.load()
        n, _, h, w = x.shape
        if x.dtype == torch.uint8:
            dt = 1
        else:
            dt = 0
            if x.dtype != torch.float32:
                x = x.float()
        x = x.contiguous()
        gated = bool(self.sparse) and _timed is None and _absmax is None
        if self.sparse and not gated:
            raise NotImplementedError('per-op profiling / calibration runs need a plan without score-gated heads')
        nb = self.max_batch(n, h, w)
        if _timed is not None and nb < n:
            raise ValueError('per-op profiling needs a batch whose tensors stay below 2^31 elements')
        if gated and nb < n:  # (CPN.core_forward routes such batches to the dense plan)
            raise NotImplementedError('score-gated heads need the whole batch in one graph run (tensors below 2^31 bytes)')
        cin = self.plan.meta.get('in_channels')
        if cin is not None and x.shape[1] != cin:  # (a slot's input copy would silently broadcast a 1-channel batch)
            raise ValueError(f'inputs have {x.shape[1]} channels, the model was built for {cin}')
        if _timed is None and _absmax is None and nb < n and self.precision != 'fp32' and not gated:
            # a batch the engine has to split (2^31-byte tensors): every balanced sub-batch is a run of its own, so that it replays
            # its hipGraph like any other shape (round 6: configs[4]'s 8 x 1024^2 ran its 4 + 4 tiles through ~125 eager launches
            # each); the head maps of the parts are concatenated (copies out of the graph slots: always safe to keep)
            parts, unc = [], []
            keep = -(-n // nb) <= self.GRAPH_SLOTS  # (more parts than slots: a part's maps must be copied out before its slot returns)
            for i0 in range(0, n, nb):
                parts.append(self.run(x[i0:i0 + nb], order_total, refinement, static_ok=keep))
                unc.append(self.last_uncertainty)
            self.last_uncertainty = None if unc[0] is None else torch.cat(unc)
            self.last_sparse = None
            cat = lambda i: None if parts[0][i] is None else torch.cat([p_[i] for p_ in parts])
            flag = parts[0][4].clone()
            for p_ in parts[1:]:
                flag = torch.maximum(flag, p_[4])
            return cat(0), cat(1), cat(2), cat(3), flag
        slot = None
        if _timed is None and _absmax is None and nb == n and self.precision != 'fp32':
            # kernel-selection switches that are read per launch are frozen into a captured graph: part of the key
            env = tuple(os.environ.get(k) for k in ('CPN_RW', 'CPN_PWR', 'CPN_PAIR_CPS', 'CPN_TH64', 'CPN_BLPHASE', 'CPN_PAIR',
                                                      'CPN_S1F', 'CPN_BRF', 'CPN_S1Q', 'CPN_BRIDGE'))
            slot = self._graph_slot((n, x.shape[1], h, w, dt, order_total, bool(refinement), env), x, dt, order_total,
                                    refinement, gated)
        if slot is not None and 'graph' in slot:
            slot['x'].copy_(x, non_blocking=True)
            slot['graph'].replay()
            scores, locations, fourier, ref, self.last_uncertainty = slot['outputs']
            flag, ws = slot['flag'], slot['ws']
            if not static_ok:
                scores, locations, fourier, ref, self.last_uncertainty, flag = (
                    None if t is None else t.clone() for t in (scores, locations, fourier, ref, self.last_uncertainty, flag))
        else:
            scores, locations, fourier, ref, self.last_uncertainty = self._alloc_outputs(n, h, w, order_total, refinement, gated)
            flag = torch.zeros(1, dtype=torch.int32, device=self.device)
            ws, need = self.workspace(nb, h, w)
            self._launch(x, dt, h, w, ws, need, (scores, locations, fourier, ref, self.last_uncertainty), flag, nb,
                         _timed=_timed, _absmax=_absmax)
        if ref is not None and tuple(ref.shape[2:]) != (h, w):  # strided refinement head: `_equal_size(.., inputs)`, cpn.py:279
            ref = _equal_size(ref, x, 'bicubic' if any(o.get('mode') == 'bicubic' for o in self.plan.ops) or
                              self.plan.meta.get('refinement_interpolation') == 'bicubic' else 'bilinear')
        self.last_sparse = None
        if gated:  # where the heads' source tensor lives in this run's arena (it stays intact until the arena's next turn)
            from ctypes import c_int32, c_int64
            off, th, tw, cs = c_int64(0), c_int32(0), c_int32(0), c_int32(0)
            _lib.check(lib.cpn_plan_tensor_info(self.handle, n, h, w, int(self.sparse['src']), off, th, tw, cs),
                       'plan_tensor_info')
            ia, ib = self.sparse['ops']
            self.last_sparse = dict(ws=ws, features_ptr=ws.data_ptr() + int(off.value), grid=(n, int(th.value), int(tw.value)),
                                    channel_stride=int(cs.value), op_a=self.ops_desc[ia], op_b=self.ops_desc[ib],
                                    weights=self.wblob, bias=self.bblob, order_total=order_total)
        return scores, locations, ref, fourier, flag


def _equal_size(x, reference, mode='bilinear'):
    """celldetection/models/cpn.py:109-115: bilinear (| bicubic) resize (align_corners=False) of an fp32 NCHW map to the spatial
    size of ``reference`` (own HIP kernel with torch CPU's arithmetic: the thresholded result must not depend on which
    backend resized the mask)."""
    if reference.shape[2:] != x.shape[2:]:
        x = x.contiguous().float()
        out = torch.empty(x.shape[:2] + tuple(reference.shape[2:]), dtype=torch.float32, device=x.device)
        _lib.check(_lib.load().cpn_resize_f32(_lib.ptr(x), _lib.ptr(out), x.shape[0] * x.shape[1], x.shape[2], x.shape[3],
                                              out.shape[2], out.shape[3], 1 if mode == 'bicubic' else 0, _lib.stream_ptr()),
                   'resize_f32')
        x = out
    return x


class CPN(nn.Module):
    """Contour Proposal Network (inference) on the HIP engine; ``backbone`` is the reference backbone class name,
    e.g. ``'ResNeXt101UNet'``, ``'U22'``, ``'ResNet18FPN'`` (celldetection/models/cpn.py:287-439)."""

    def __init__(self, backbone: str, in_channels: int, order: int = 5, nms_thresh: float = .2,
                 score_thresh: float = .9, certainty_thresh: float = None, samples: int = 32, classes: int = 2,
                 refinement: bool = True, refinement_iterations: int = 4, refinement_margin: float = 3.,
                 refinement_buckets: int = 1, uncertainty_head=False, uncertainty_nms=False, order_weights=True,
                 backbone_kwargs: dict = None, **kwargs):
        super().__init__()
        unsupported = {k: v for k, v in kwargs.items() if (k in ('contour_head_stride', 'refinement_head_stride')
                                                           and v not in (None, 1, 2, 4, 8))
                       }
        if unsupported:
            raise NotImplementedError(f'Unsupported CPN options on the HIP path: {unsupported}')
        features = {name: kwargs[key] for name, key in (('score', 'score_features'), ('location', 'location_features'),
                                                        ('contour', 'contour_features'), ('uncertainty', 'uncertainty_features'),
                                                        ('refinement', 'refinement_features')) if kwargs.get(key) is not None}
        kernel_sizes = {k[len('kernel_size_'):]: int(v) for k, v in kwargs.items() if k.startswith('kernel_size_')}
        # hidden activation of the ReadOut heads: `head_activation_<head>`, else `head_activation`, else ReLU (cpn.py:183-233)
        head_act = {h: graph.head_activation_name(kwargs.get(f'head_activation_{h}', kwargs.get('head_activation', 'relu')))
                    for h in ('score', 'location', 'fourier', 'uncertainty', 'refinement')}
        self.order = order
        self.nms_thresh = nms_thresh
        self.samples = samples
        self.score_thresh = score_thresh
        self.score_channels = 1 if classes in (1, 2) else classes
        self.refinement = refinement
        self.refinement_iterations = refinement_iterations
        self.refinement_margin = refinement_margin
        self.functional = False
        self.full_detail = False
        self.score_target_dtype = None
        self.certainty_thresh = certainty_thresh
        self.uncertainty_nms = uncertainty_nms
        self._backbone_name = backbone
        self._plan_kwargs = dict(backbone=backbone, in_channels=in_channels, order=order,
                                 score_channels=self.score_channels, refinement=refinement,
                                 refinement_margin=refinement_margin, refinement_buckets=refinement_buckets,
                                 order_weights=bool(order_weights), backbone_kwargs=backbone_kwargs,
                                 uncertainty_head=bool(uncertainty_head),
                                 contour_head_channels=kwargs.get('contour_head_channels'),
                                 refinement_head_channels=kwargs.get('refinement_head_channels'),
                                 kernel_sizes=kernel_sizes, features=features or None,
                                 contour_head_stride=int(kwargs.get('contour_head_stride') or 1),
                                 refinement_head_stride=int(kwargs.get('refinement_head_stride') or 1),
                                 head_activations=head_act if any(v != 'relu' for v in head_act.values()) else None,
                                 refinement_full_res=bool(kwargs.get('refinement_full_res', True)),
                                 fuse_kwargs=kwargs.get('fuse_kwargs') or None,
                                 refinement_interpolation='bicubic' if kwargs.get('refinement_interpolation') == 'bicubic'
                                 else 'bilinear')
        # `_equal_size(.., mode=refinement_interpolation, align_corners=False)` (cpn.py:109-115,277-279): torch accepts
        # align_corners for the interpolating modes only, so in the reference every other mode ('nearest', 'area', ...) raises
        # as soon as a resize is needed; reproduced in `_Engine.run` / `core_forward` (bicubic: not built)
        self.refinement_interpolation = kwargs.get('refinement_interpolation', 'bilinear')
        # 'bf16' (MFMA performance path) | 'fp32' (verification path, ~100x slower) | 'fp8' (e4m3 activations and
        # weights on the K=64 scaled MFMA, 2x the bf16 rate; static activation scales from ``calibrate_fp8`` or, if
        # that was not called, from the first batch that is forwarded)
        self.precision = 'bf16'
        # score-gated location / Fourier heads (bf16 only; csrc/sparse_heads.hip): the two heads are evaluated at the
        # proposal pixels only -- CPN.forward reads nothing else of their maps (cpn.py:613-637); outputs are identical.
        # Validated bit-identical on the MI355X (tests/test_gpu_sparse_heads.py).  A run-time switch:
        #   'auto' (default)  forward() / forward_pipelined() and everything built on them (tile loops) gate the two heads
        #                     wherever the plan qualifies, with the dense convs as fallback above the measured break-even
        #                     density (ops.SPARSE_HEADS_MAX_DENSITY) and for batches the engine must split; the public
        #                     core_forward() / engine() -- CPNCore.forward's dense maps, per-op profiles -- stay dense
        #   True              additionally core_forward() / engine() use the gated plan (locations / fourier maps are None)
        #   False             the reference's dense graph everywhere
        self.sparse_heads = 'auto'
        self._fp8_scales = None
        # sub-pixel decompositions (bf16 plans), taken wherever the resize is an exact x2; ONE run-time switch for both:
        # * UNet decoder convs over nearest-upsampled maps (graph._two_conv_norm_relu): 4/9 of the MACs on the upsampled channels
        # * the refinement head over the bilinear-resized feature map of the FPN models (graph._readout): 25 instead of 49 taps
        self.subpixel = True
        # bf16: fused ReadOut tails + fused bilinear head source + sub-pixel decoder convs + stem kernel + fused bottleneck
        # heads (conv1 -> grouped conv2 of the ResNeXt blocks, csrc/conv_pair.hip; the executor picks per input size)
        self._plan = graph.build_plan(**self._plan_kwargs, subpixel=True, stem_fast=True, fuse_blocks=True, bilinear_phases=True)
        self._alt_plans = {}
        for key, shape, kind in self._plan.entries:
            _register(self, key, shape, kind)
        self.core.order = order
        self.core.refinement_buckets = refinement_buckets
        self._engine = None
        self._engine_dense = None  # dense-plan fallback of a model with score-gated heads (batches the engine must split)
        self._hparams = {}
        self.max_imsize = None
        self.eval()

    # ---- reference-compatible plumbing -------------------------------------------------------------------------
    @property
    def hparams(self):
        return self._hparams

    def _set_hparams(self, hp):
        self._hparams = dict(hp)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._engine = self._engine_dense = None  # weights change -> repack on next forward
        self._fp8_scales = None
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def repack(self):
        """Call after modifying parameters in place (the fp8 activation scales are stale then, too)."""
        self._engine = self._engine_dense = None
        self._fp8_scales = None

    def _apply(self, fn, *a, **k):
        self._engine = self._engine_dense = None
        return super()._apply(fn, *a, **k)

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError('celldetection_amd.CPN is an inference engine; training is out of scope.')
        return super().train(False)

    def plan_for(self, precision: str, gate: bool = None) -> graph.Plan:
        """Layer plan per precision: bf16 fuses the ReadOut tails and the bilinear resize in front of the refinement head;
        fp8 keeps that resize as its own op (on e4m3 codes; also the plan its bf16 calibration run uses, so that the
        tensor ids of the activation scales match); fp32 (verification) fuses nothing."""
        if precision == 'bf16':
            # score-gated heads: same entries / weights, the two head convs are deferred ops; sub-pixel decoder convs:
            # same entries, the first conv of every UNet decoder level additionally carries its decomposition
            key = (self._gate_requested(False) if gate is None else bool(gate), bool(self.subpixel))
            if key == (False, True):
                return self._plan
            if key not in self._alt_plans:
                self._alt_plans[key] = graph.build_plan(**self._plan_kwargs, sparse_heads=key[0], subpixel=key[1],
                                                        stem_fast=True, fuse_blocks=True, bilinear_phases=key[1])
            return self._alt_plans[key]
        key = (precision, bool(self.subpixel)) if precision == 'fp8' else precision
        if key not in self._alt_plans:
            # fp8: the resize stays its own op (the refinement head over it carries the bilinear phase decomposition: same
            # tensors either way) and the ResNet stem takes its bf16 fast path (e4m3 output); fp32: nothing fused
            # ... and the UNet decoders carry their sub-pixel triples (partial sums as bf16; the bridge keeps its stated conv)
            extra = dict(fuse_bilinear=False, stem_fast=True, bilinear_phases=bool(self.subpixel),
                         subpixel='triples' if self.subpixel else False) if precision == 'fp8' else \
                dict(fuse_readout=False, fuse_bilinear=False)
            self._alt_plans[key] = graph.build_plan(**self._plan_kwargs, **extra)
        return self._alt_plans[key]

    def _gate_requested(self, forward_path: bool) -> bool:
        sh = self.sparse_heads
        if isinstance(sh, str):
            if sh != 'auto':
                raise ValueError("sparse_heads must be True, False or 'auto'")
            return forward_path
        return bool(sh)

    def engine(self, device=None, calibration_input=None, _forward_path: bool = False) -> _Engine:
        device = torch.device(device) if device is not None else self.order_weights_device()
        if device.type != 'cuda':
            raise RuntimeError('celldetection_amd runs on the MI355X (HIP) only: move the model and inputs to a GPU. '
                               'There is no CPU fallback in the product path.')
        if self.precision not in ('bf16', 'fp32', 'fp8'):
            raise ValueError("precision must be 'bf16', 'fp32' or 'fp8'")
        gate = self._gate_requested(_forward_path)
        if self.precision == 'bf16' and not gate and self.sparse_heads == 'auto':
            return self._dense_engine(device)  # ('auto': the dense engine lives next to the gated one)
        if self.precision == 'bf16' and gate and self.sparse_heads == 'auto' and \
                self.plan_for('bf16', True).meta.get('sparse_heads') is None:
            # ('auto' on a plan whose heads do not qualify for the gate: ONE engine serves both paths -- a second set of
            # packed weights and hipGraph slots would buy nothing)
            return self._dense_engine(device)
        # the packed plan depends on `subpixel` in bf16 (sub-pixel triples) AND in fp8 (bilinear phase head): both keys carry it
        sparse = (gate, bool(self.subpixel)) if self.precision == 'bf16' else \
            ((None, bool(self.subpixel)) if self.precision == 'fp8' else None)
        if self._engine is None or self._engine.device != device or self._engine.precision != self.precision or \
                self._engine.sparse_requested != sparse:
            if self.precision == 'fp8':
                if self._fp8_scales is not None and getattr(self, '_fp8_scales_key', None) != self._fp8_plan_key():
                    # the scales are indexed by the tensor ids of the plan they were calibrated with; `subpixel` changes that plan
                    # (one more tensor per decoder level): scales of another plan must never be applied silently
                    warnings.warn("precision 'fp8': `subpixel` changed after calibrate_fp8(); the activation scales belong to "
                                  'another plan and are dropped', RuntimeWarning, stacklevel=3)
                    self._fp8_scales = None
                if self._fp8_scales is None:
                    if calibration_input is None:
                        raise RuntimeError("precision 'fp8' needs activation scales: call calibrate_fp8(batch) first")
                    warnings.warn("precision 'fp8': calibrating the static activation scales on the first forwarded "
                                  'batch; activations of later batches that exceed its range saturate at 448 * scale. '
                                  'Call calibrate_fp8() on representative tiles instead.', RuntimeWarning, stacklevel=3)
                    self.calibrate_fp8(calibration_input)
                self._engine = _Engine(self.plan_for('fp8'), self.state_dict(), device, 'fp8', act_scales=self._fp8_scales)
            else:
                self._engine = _Engine(self.plan_for(self.precision, gate), self.state_dict(), device, self.precision)
            self._engine.sparse_requested = sparse
        return self._engine

    def _dense_engine(self, device) -> _Engine:
        """bf16 engine of the plan WITHOUT score-gated heads: the public core_forward() / engine() of an 'auto' model and the
        fallback for batches the engine must split (the gathered heads read the heads' source of the whole batch)."""
        key = bool(self.subpixel)
        if self._engine_dense is None or self._engine_dense.device != device or self._engine_dense.sparse_requested != key:
            self._engine_dense = _Engine(self.plan_for('bf16', False), self.state_dict(), device, 'bf16')
            self._engine_dense.sparse_requested = key
        return self._engine_dense

    @torch.no_grad()
    def calibrate_fp8(self, inputs: torch.Tensor):
        """Static e4m3 activation scales (one per tensor of the conv graph) from ONE bf16 run on ``inputs``:
        scale = max|x| / 448.  Weight scales are per output channel and need no data."""
        plan = self.plan_for('fp8')
        eng = _Engine(plan, self.state_dict(), inputs.device, 'bf16')
        absmax = eng.activation_absmax(inputs, self.core.order, self.refinement)
        self._fp8_scales = [max(float(v), 1e-12) / 448. for v in absmax.tolist()]
        self._fp8_scales_key = self._fp8_plan_key()
        for op in plan.ops:  # max-pool / bilinear kernels work on the codes: output scale == input scale
            if op['op'] in ('maxpool', 'bilinear'):
                self._fp8_scales[op['dst']] = self._fp8_scales[op['src0']]
            elif op['op'] in ('input', 'input_stem'):  # inputs lie in [0, 1] (asserted per forward): a fixed scale, whichever
                self._fp8_scales[op['dst']] = 1. / 448.  # of the stem alternatives ran during calibration
        if self._engine is not None and self._engine.precision == 'fp8':
            self._engine = None
        return self._fp8_scales

    def _fp8_plan_key(self):
        """What the tensor ids of the fp8 plan (= the indices of ``_fp8_scales``) depend on."""
        return bool(self.subpixel), len(self.plan_for('fp8').tensors)

    def order_weights_device(self):
        return next(self.parameters()).device

    # ---- forward --------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def core_forward(self, inputs: torch.Tensor, _static_ok: bool = False, _forward_path: bool = False):
        """CPNCore.forward (cpn.py:238-283) -> (scores(sigmoid applied), locations, refinement, fourier).
        (``_static_ok`` / ``_forward_path``: internal -- the forward paths consume the maps before the engine re-uses their
        hipGraph slot, and are the callers for which ``sparse_heads = 'auto'`` gates the location / Fourier heads.)"""
        eng = self.engine(inputs.device, calibration_input=inputs, _forward_path=_forward_path)
        if self.refinement and getattr(self, 'refinement_interpolation', 'bilinear') != 'bilinear':
            self._check_refinement_interpolation(eng, inputs)
        if eng.sparse and eng.max_batch(inputs.shape[0], *inputs.shape[-2:]) < inputs.shape[0]:
            # the engine has to split this batch (2^31-byte tensors), but the gathered heads read the heads' source of the
            # WHOLE batch after the run: such batches take the dense plan (same outputs)
            eng = self._dense_engine(eng.device)
        scores, locations, refinement, fourier, flag = eng.run(inputs, self.core.order, self.refinement,
                                                               static_ok=_static_ok)
        self._last_flag = flag
        self._last_uncertainty = eng.last_uncertainty  # fifth CPNCore output (cpn.py:283), [N,4,h,w] or None
        self._last_sparse = eng.last_sparse  # score-gated heads: locations / fourier are None, evaluated in postprocess
        return scores, locations, refinement, fourier

    def _check_refinement_interpolation(self, eng, inputs):
        """``refinement_interpolation`` other than 'bilinear': the reference calls F.interpolate(.., mode=mode, align_corners=False)
        whenever the refinement feature or the head's output differs from the input size (cpn.py:109-115,277-279), and torch
        rejects align_corners for every non-interpolating mode -- the same ValueError is raised here in that case; when no resize
        is needed the mode is never used (there and here)."""
        from ctypes import c_int32, c_int64
        n, _, h, w = inputs.shape
        sizes = [eng.output_size(h, w, _lib.OUT_REFINEMENT)]
        for op in eng.plan.ops:  # the resize in front of the head: its own op, or fused into the head conv's loader
            if op['op'] == 'bilinear' or (op['op'] == 'conv' and op.get('up0') == 'bilinear'):
                off, th, tw, cs = c_int64(0), c_int32(0), c_int32(0), c_int32(0)
                _lib.check(_lib.load().cpn_plan_tensor_info(eng.handle, n, h, w, int(op['src0']), off, th, tw, cs), 'plan_tensor_info')
                sizes.append((int(th.value), int(tw.value)))
        if self.refinement_interpolation != 'bicubic' and any(tuple(sz) != (h, w) for sz in sizes):
            raise ValueError('align_corners option can only be set with the interpolating modes: linear | bilinear | bicubic | '
                             'trilinear')

    # ---- batches the engine has to split (2^31-byte tensors, e.g. 8 x 3 x 1024^2 on the 256-channel FPN maps) on the forward paths:
    # the score-gated heads read the heads' source of ONE graph run, so such a batch is forwarded as balanced sub-batches, each
    # a complete forward of its own (gated conv graph -> post-processing), and the per-image results are put back together.
    # Every output of CPN.forward is per image (decode, refinement, NMS: cpn.py:616-734), so this equals the one-batch result;
    # up to round 5 these batches fell back to the dense plan (twice the head FLOPs).
    def _gated_sub_batch(self, inputs):
        """Sub-batch size if ``inputs`` must be split to keep the score-gated heads, else None."""
        if self.precision != 'bf16' or not self._gate_requested(True) or inputs.shape[0] < 2:
            return None
        eng = self.engine(inputs.device, _forward_path=True)
        if not eng.sparse:
            return None
        nb = eng.max_batch(inputs.shape[0], *inputs.shape[-2:])
        return nb if nb < inputs.shape[0] else None

    @staticmethod
    def _slice_batch_kwargs(kwargs, i0, i1, n):
        """Per-image keyword tensors (``offsets`` [N,2], score bounds [N,1,H,W], the loader's batch keys) follow their images."""
        return {k: (v[i0:i1] if isinstance(v, torch.Tensor) and v.ndim > 0 and v.shape[0] == n else v) for k, v in kwargs.items()}

    @staticmethod
    def _merge_outputs(parts, sizes):
        """Results of consecutive sub-batches (``sizes`` images each) as the result of the whole batch."""
        if isinstance(parts[0], tuple):  # flat_output: (dict of flat tensors incl. the image index 'b', per-image counts)
            flat, counts, i0 = {}, [], 0
            for (f, c), sz in zip(parts, sizes):
                for k, v in f.items():
                    flat.setdefault(k, []).append(v + i0 if k == 'b' else v)
                counts += list(c)
                i0 += sz
            return {k: torch.cat(v) for k, v in flat.items()}, counts
        out = OrderedDict()
        for k in parts[0]:
            out[k] = None if parts[0][k] is None else [t for p_ in parts for t in p_[k]]
        return out

    @torch.no_grad()
    def forward(self, inputs, targets=None, nms=True, **kwargs):

        if targets is not None:
            raise NotImplementedError('Loss computation / training is out of scope of the HIP inference engine.')
        if not inputs.is_cuda:
            raise RuntimeError('celldetection_amd.CPN.forward needs GPU inputs (no CPU fallback).')
        nb = self._gated_sub_batch(inputs)
        if nb is not None:
            n = inputs.shape[0]
            cuts = list(range(0, n, nb)) + [n]
            parts = [self.forward(inputs[a:b], nms=nms, **self._slice_batch_kwargs(kwargs, a, b, n)) for a, b in zip(cuts, cuts[1:])]
            return self._merge_outputs(parts, [b - a for a, b in zip(cuts, cuts[1:])])
        original_size = tuple(inputs.shape[-2:])
        scores, locations, refinement, fourier = self.core_forward(inputs, _static_ok=True, _forward_path=True)
        return self.postprocess(scores, locations, refinement, fourier, original_size, nms=nms, flag=self._last_flag,
                                uncertainty=self._last_uncertainty, sparse=self._last_sparse, **kwargs)

    @torch.no_grad()
    def forward_pipelined(self, batches, nms=True, _events=None, **kwargs):
        """Throughput mode for streams of batches (the tile loop of celldetection_scripts/cpn_inference.py:357-388):
        generator that yields ``forward(x, **kw)`` for every item of ``batches`` (``x`` or ``(x, kw_dict)``), in
        order.  The conv graph of batch i+1 is enqueued on one HIP stream before the post-processing of batch i
        (compaction -> decode -> NMS, incl. its host read-back of the proposal counts) runs on a second stream, so
        the host latency and the small post-processing kernels hide behind the MFMA work.  Results are identical to
        calling ``forward`` per batch."""
        groups = deque()  # [sub-batch sizes still to come, results so far, sizes] of the caller's items, oldest first

        def expanded():  # the caller's items, those the gated engine must split as consecutive sub-batch items
            for item in batches:
                x, kw = item if isinstance(item, (tuple, list)) else (item, {})
                nb = self._gated_sub_batch(x) if x.is_cuda else None
                if nb is None:
                    groups.append([1, [], None])
                    yield x, kw
                else:
                    n = x.shape[0]
                    cuts = list(range(0, n, nb)) + [n]
                    groups.append([len(cuts) - 1, [], [b - a for a, b in zip(cuts, cuts[1:])]])
                    for a, b in zip(cuts, cuts[1:]):
                        yield x[a:b], self._slice_batch_kwargs(dict(kwargs, **kw), a, b, n)

        for out in self._forward_pipelined(expanded(), nms, _events, kwargs):
            g = groups[0]
            g[1].append(out)
            g[0] -= 1
            if g[0] == 0:
                groups.popleft()
                yield g[1][0] if g[2] is None else self._merge_outputs(g[1], g[2])

    @torch.no_grad()
    def _forward_pipelined(self, batches, nms, _events, kwargs):
        dev = self.order_weights_device()
        s_conv, s_post = self._streams(dev)
        caller = torch.cuda.current_stream(dev)
        pending = None
        post_done = [None]  # event after the most recent post-processing (score-gated heads: arena hand-back)

        def finish(item):
            maps, unc, flag, size, kw, ev, sparse = item
            with torch.cuda.stream(s_post):
                s_post.wait_event(ev)
                out = self.postprocess(*maps, size, nms=nms, flag=flag, uncertainty=unc, sparse=sparse,
                                       **dict(kwargs, **kw))
                done = torch.cuda.Event()
                done.record(s_post)
            post_done[0] = done
            caller.wait_event(done)  # the caller's stream may consume the outputs
            if isinstance(out, tuple):  # flat_output: (dict of flat tensors, per-image counts)
                for t in out[0].values():
                    t.record_stream(caller)
            else:
                for v in out.values():
                    for t in (v or ()):
                        t.record_stream(caller)
            return out

        for item in batches:
            x, kw = item if isinstance(item, (tuple, list)) else (item, {})
            if not x.is_cuda:
                raise RuntimeError('celldetection_amd.CPN.forward needs GPU inputs (no CPU fallback).')
            s_conv.wait_stream(caller)  # x was produced on the caller's stream
            if self._gate_requested(True) and post_done[0] is not None:
                # score-gated heads: this run re-uses the arena of the run before the previous one, whose head source
                # was read by that batch's post-processing (finished before post_done[0] was recorded)
                s_conv.wait_event(post_done[0])
            with torch.cuda.stream(s_conv):
                if _events is not None:  # (start, end) HIP events around the conv graph, on its launch stream
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(s_conv)
                maps = self.core_forward(x, _static_ok=True, _forward_path=True)
                ev = torch.cuda.Event(enable_timing=_events is not None)
                ev.record(s_conv)
                if _events is not None:
                    _events.append((e0, ev))
                for t in maps + (self._last_uncertainty, self._last_flag):
                    if t is not None:
                        t.record_stream(s_post)
                if self._last_sparse is not None:  # the arena holding the heads' source is read on the post stream
                    self._last_sparse['ws'].record_stream(s_post)
                x.record_stream(s_conv)
            cur = (maps, self._last_uncertainty, self._last_flag, tuple(x.shape[-2:]), kw, ev, self._last_sparse)
            if pending is not None:
                yield finish(pending)
            pending = cur
        if pending is not None:
            yield finish(pending)

    def _streams(self, dev):
        st = getattr(self, '_pipe_streams', None)
        if st is None or st[0].device != dev:
            st = self._pipe_streams = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
        return st

    def forward_tiled(self, inputs, crop_size=1024, stride=512, **kwargs):
        """In-model tiling (celldetection/models/lightning_cpn.py:88-177); see ``inference.forward_tiled``."""
        from . import inference
        return inference.forward_tiled(self, inputs, crop_size=crop_size, stride=stride, **kwargs)

    @torch.no_grad()
    def postprocess(self, scores, locations, refinement, fourier, original_size, nms=True, flag=None,
                    uncertainty=None, flat_output=False, sparse=None, **kwargs):
        """CPN.forward after the core (cpn.py:575-734) on given head maps: ``scores`` [N,1,h,w] probabilities (sigmoid
        already applied; binary CPNs) or [N,classes,h,w] raw logits (multi-class, cpn.py:583-585); locations [N,2,h,w];
        refinement [N,2*buckets,H,W] or None; fourier [N,4*O,h,w]; uncertainty [N,4,h,w] or None (cpn.py:209-221).
        ``flat_output``: return ``(dict of flat [K, ...] tensors incl. 'b' = image index int32 [K], per-image counts)``
        instead of per-image lists (the slide loop filters all tiles of a batch at once)."""
        n = scores.shape[0]
        lb, ub = kwargs.get('scores_lower_bound'), kwargs.get('scores_upper_bound')
        ub = None if ub is None else _equal_size(ub.to(scores), scores)  # cpn.py:118-123
        lb = None if lb is None else _equal_size(lb.to(scores), scores)
        class_map = None
        if scores.shape[1] == 1:
            if ub is not None:
                scores = torch.minimum(scores, ub)
            if lb is not None:
                scores = torch.maximum(scores, lb)
            select_map, thresh = scores, self.score_thresh
        else:  # softmax + bounds + argmax; proposals are the pixels whose class is > 0 (cpn.py:583-585,614)
            scores, class_map, select_map, _ = ops.class_scores(scores, lb, ub)
            thresh = .5
        if self.certainty_thresh is not None and uncertainty is not None:  # cpn.py:617-618
            select_map = ops.certainty_mask(select_map, uncertainty, self.certainty_thresh)
            if class_map is not None:
                thresh = .5  # foreground map is 1 / 0 / -1
        order = min(self.order, self.core.order)  # cpn.py:597-598
        indices, counts, flag_v = ops.compact_scores(select_map, thresh, extra_flag=flag)
        if flag_v:
            raise AssertionError('Inputs should be in interval (0.0, 1.0)')  # models/commons.py:696-697
        if self.functional and int(indices.shape[0]) > 0:
            # cpn.py:591-592 views the Fourier map as [n, c / 2, 2, h, w]; fouriers2contours then indexes columns (1, 3) of a
            # last dimension of size 2 (ops/cpn.py:93-94): with at least one proposal the reference's forward raises this
            # IndexError, without proposals the empty index passes and `fourier` comes back as [0, c / 2, 2] (both recorded from
            # the imported reference: tests/golden/reference_behaviours.json)
            raise IndexError('index 3 is out of bounds for dimension 0 with size 2')
        iters = self.refinement_iterations if (self.refinement and refinement is not None) else 0
        gathered = locations is None
        if gathered:  # score-gated heads: location / Fourier head values of the proposals only (ops.sparse_heads)
            if sparse is None:
                raise ValueError('postprocess: locations / fourier maps are missing and no score-gated head context given')
            if tuple(sparse['grid'][1:]) != tuple(scores.shape[-2:]):
                raise RuntimeError('score-gated heads: head grid and score grid differ')
            src = (sparse['features_ptr'], sparse['channel_stride'], sparse['grid'])
            if int(indices.shape[0]) > ops.SPARSE_HEADS_MAX_DENSITY * scores.shape[0] * scores.shape[-2] * scores.shape[-1]:
                # most pixels are proposals: the dense convs are cheaper (and produce the same values)
                locations = ops.dense_head(sparse['op_a'], *src, sparse['weights'], sparse['bias'])
                fourier = ops.dense_head(sparse['op_b'], *src, sparse['weights'], sparse['bias'])
                gathered = False
            else:
                locations, fourier = ops.sparse_heads(sparse['op_a'], sparse['op_b'], *src, indices, sparse['weights'],
                                                      sparse['bias'])
        flat = ops.decode_proposals(indices, scores, locations, fourier, refinement if iters > 0 else None,
                                    size=original_size, order=order, samples=self.samples, iterations=iters,
                                    offsets=kwargs.get('offsets'), num_buckets=self.core.refinement_buckets,
                                    gathered=gathered)
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        if class_map is None:
            flat['classes'] = torch.ones((offs[-1],), dtype=torch.int64, device=scores.device)
        else:
            flat['classes'] = class_map.reshape(-1)[indices.long()].to(torch.int64)
        if self.functional:  # (no proposals: see above)
            flat['fourier'] = flat['fourier'].reshape(0, 2 * flat['fourier'].shape[1], 2)
        keys = ['contours', 'boxes', 'scores', 'classes', 'locations', 'fourier', 'contour_proposals']
        nms_weights = flat['scores']
        if uncertainty is not None:  # cpn.py:634-636,723-726
            flat['box_uncertainties'] = ops.gather_channels(uncertainty, indices)
            keys.append('box_uncertainties')
            if self.uncertainty_nms:
                nms_weights = flat['scores'] * (1. - flat['box_uncertainties'].mean(1))
        if nms and max(counts + [0]) <= ops.NMS_BATCH_SIZE and offs[-1] > 0:
            # one segmented NMS over all images + ONE gather per output key (instead of N x 7 small index kernels)
            keep, kc = ops._nms_segments(flat['boxes'], nms_weights, offs, self.nms_thresh)
            sel = torch.cat([keep[offs[i]:offs[i] + kc[i]] for i in range(n)])
            flat = {k: flat[k].index_select(0, sel) for k in keys + ['b']}  # cpn.py:53-60
            offs = [0]
            for c in kc:
                offs.append(offs[-1] + c)
            nms = False
        if flat_output and not nms:
            return {k: flat[k] for k in keys + ['b']}, [offs[i + 1] - offs[i] for i in range(n)]
        outputs = OrderedDict((k, [flat[k][offs[i]:offs[i + 1]] for i in range(n)]) for k in keys)  # cpn.py:42-50
        if 'box_uncertainties' not in outputs:
            outputs['box_uncertainties'] = None
        if nms:  # > NMS_BATCH_SIZE proposals in one image: the reference's chunked procedure (ops/cpn.py:212-224)
            weights = [nms_weights[offs[i]:offs[i + 1]] for i in range(n)]
            keep = ops.batched_box_nmsi(outputs['boxes'], weights, self.nms_thresh)
            for k in keys:
                outputs[k] = [v[kp] for v, kp in zip(outputs[k], keep)]
        if flat_output:
            cnt = [int(v.shape[0]) for v in outputs['scores']]
            out = {k: torch.cat(outputs[k]) for k in keys}
            out['b'] = torch.repeat_interleave(torch.arange(n, dtype=torch.int32, device=scores.device),
                                               torch.tensor(cnt, device=scores.device))
            return out, cnt
        return outputs


class Inference:
    """``cd.models.Inference`` (celldetection/models/inference.py:8-26): array in -> numpy dict out.  ``amp`` is accepted
    for signature compatibility; the HIP engine's precision is a model attribute (``model.precision``)."""

    def __init__(self, model, device=None, amp=False, transforms=None):
        self.transforms = transforms
        self.device = device or 'cuda'
        self.model = model.to(self.device)
        self.model.eval()
        self.model.requires_grad_(False)
        self.use_amp = amp

    def __call__(self, inputs):
        if self.transforms is not None:
            inputs = self.transforms(inputs)
        inputs = torch.as_tensor(inputs, device=self.device, dtype=torch.float32)
        while inputs.ndim < 4:
            inputs = inputs[None]
        out = self.model(inputs)
        to_np = lambda v: v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v
        return OrderedDict((k, None if v is None else [to_np(t) for t in v]) for k, v in out.items())


__all__.append('Inference')


def _make(backbone):
    def __init__(self, in_channels: int, order: int = 5, nms_thresh: float = .2, score_thresh: float = .9,
                 samples: int = 32, classes: int = 2, refinement: bool = True, refinement_iterations: int = 4,
                 refinement_margin: float = 3., refinement_buckets: int = 1, backbone_kwargs: dict = None, **kwargs):
        hp = dict(in_channels=in_channels, order=order, nms_thresh=nms_thresh, score_thresh=score_thresh,
                  samples=samples, classes=classes, refinement=refinement, refinement_iterations=refinement_iterations,
                  refinement_margin=refinement_margin, refinement_buckets=refinement_buckets,
                  backbone_kwargs=backbone_kwargs, **kwargs)
        CPN.__init__(self, backbone, in_channels, order=order, nms_thresh=nms_thresh, score_thresh=score_thresh,
                     samples=samples, classes=classes, refinement=refinement,
                     refinement_iterations=refinement_iterations, refinement_margin=refinement_margin,
                     refinement_buckets=refinement_buckets, backbone_kwargs=backbone_kwargs, **kwargs)
        self._set_hparams(hp)

    cls = type(f'Cpn{backbone}', (CPN,), {'__init__': __init__, '__doc__': (
        f'Contour Proposal Network with a {backbone} backbone (celldetection/models/cpn.py Cpn{backbone}); '
        f'inference on the MI355X HIP engine.')})
    return cls


for _bb in graph.BACKBONES:
    _c = _make(_bb)
    globals()[_c.__name__] = _c
    __all__.append(_c.__name__)
