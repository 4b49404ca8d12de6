"""``Predictor`` / ``tiled_apply`` -- drop-in for ``elektronn3.inference.Predictor`` (elektronn3/inference/inference.py).

Same constructor arguments, option handling, tile order and coordinates, shape padding and return value (a CPU
tensor) as the reference (inference.py:45-199, 368-687).  What is different is where the data lives:

* the (padded) input volume and the output volume are DEVICE-resident (288 GB of HBM holds cfg 5's 8 GiB volume,
  its 11 GB zero-padded copy and the 21 GB output at once).  The reference stages every tile through host memory
  with synchronous H2D/D2H copies per tile (inference.py:189-197) and pads with a float64 ``np.zeros``
  (inference.py:660); here there is ONE H2D copy of the input and ONE D2H copy of the result.
* tiles are cut as strided views of the padded device volume; for an :class:`elektronn3_amd.unet.UNet` the model
  call is the eval-mode native forward (BatchNorm folded into the conv epilogues) with the ``Softmax(1)`` of
  ``nn.Sequential(model, nn.Softmax(1))`` (inference.py:443-444) fused into the last kernel.
* tiles are independent (each carries its own halo), so with ``torch.distributed`` initialised (one process per GPU of one
  node) and ``tile_parallel=True`` the rows of tiles are split over the ranks with no data-path collective; every rank
  writes its rows into one output buffer in shared host memory.

Any other ``nn.Module`` is accepted too and is simply called like the reference calls it.
"""
import copy
import itertools
import logging
import os
import time
from collections import OrderedDict
from pathlib import Path
from typing import Callable, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import nn

logger = logging.getLogger('elektronn3log')

Transform = Callable[[np.ndarray, Optional[np.ndarray]], Tuple[np.ndarray, Optional[np.ndarray]]]


_COPY_POOL = None
_LIBC = None


def _advise_hugepages(t):
    """madvise(MADV_HUGEPAGE) on a freshly allocated (not yet touched) host tensor: the result volume of the pipelined predict() is faulted in by the download worker
    while it copies (17 GB of the cfg-5 result = 4 M page faults of 4 KB); with transparent huge pages in `madvise` mode (the usual default) this makes them 2 MB
    faults.  Best effort: no effect where the kernel does not offer it.  E3_PREDICTOR_NO_HUGEPAGES=1: A/B switch."""
    global _LIBC
    if os.environ.get('E3_PREDICTOR_NO_HUGEPAGES') is not None or t.numel() * t.element_size() < (64 << 20):
        return
    try:
        import ctypes
        if _LIBC is None:
            _LIBC = ctypes.CDLL(None, use_errno=True)
            _LIBC.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
        lo = (t.data_ptr() + (2 << 20) - 1) & ~((2 << 20) - 1)
        hi = (t.data_ptr() + t.numel() * t.element_size()) & ~((2 << 20) - 1)
        if hi > lo:
            _LIBC.madvise(lo, hi - lo, 14)          # MADV_HUGEPAGE
    except Exception:  # noqa: BLE001
        pass


def _host_copy(dst, src):
    """dst.copy_(src) for two host tensors (N, C, z, ...), large ones split along z over a few threads: a single memcpy stream (~5 GB/s with the
    page faults of a fresh output buffer) is less than the 16-bit Predictor moves per second."""
    global _COPY_POOL
    nz = src.shape[2] if src.dim() >= 3 else 1
    if src.numel() * src.element_size() < (64 << 20) or nz < 4:
        dst.copy_(src)
        return
    if _COPY_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _COPY_POOL = ThreadPoolExecutor(max_workers=4, thread_name_prefix='e3copy')
    step = -(-nz // 4)
    futs = [_COPY_POOL.submit(lambda a=a: dst[:, :, a:a + step].copy_(src[:, :, a:a + step])) for a in range(0, nz, step)]
    for f in futs:
        f.result()


class _PinnedRing:
    """Two page-locked staging buffers per direction for the Predictor's host <-> device pipeline: a slab of a pageable host tensor is copied
    into a pinned slot by the CPU and from there to the device by the copy engines (and the other way round), so the transfers are asynchronous
    DMA instead of the runtime's pageable path (which stages through its own bounce buffer and moves the data with copy KERNELS on the compute
    units, beside the model's kernels).  One ring per Predictor and direction; used by one thread at a time."""

    def __init__(self, slot_bytes=256 << 20, device=None):
        self.slots = [torch.empty(slot_bytes, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        # device-side twins of the slots: the gather / scatter between a strided view of the volume and a contiguous slab goes through THESE instead of
        # the temporaries torch.Tensor.copy_ allocates for a non-contiguous side -- a first-time device allocation in the middle of predict() stalls the
        # running rows of tiles by 50 - 150 ms (profiles/r06_predictor_modes.md), these are allocated with the ring
        self.dev_slots = [torch.empty(slot_bytes, dtype=torch.uint8, device=device) for _ in range(2)] if device is not None else None
        self.events = [None, None]
        self.pending = [None, None]          # d2h: (host view, slot view) still to be copied out
        self.i = 0

    def _slot(self, shape, dtype):
        k = self.i
        self.i ^= 1
        if self.events[k] is not None:
            self.events[k].synchronize()
            self.events[k] = None
        if self.pending[k] is not None:
            host, view = self.pending[k]
            _host_copy(host, view)
            self.pending[k] = None
        n = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        dev_view = self.dev_slots[k][:n].view(dtype).view(*shape) if self.dev_slots is not None else None
        return k, self.slots[k][:n].view(dtype).view(*shape), dev_view

    def _planes(self, t):
        per_plane = max(1, t[:, :, :1].numel() * t.element_size())
        return max(1, self.slots[0].numel() // per_plane)

    def fits(self, t):
        """a z plane of t fits into a slot (otherwise the caller takes the runtime's pageable copy)"""
        return t[:, :, :1].numel() * t.element_size() <= self.slots[0].numel()

    def h2d(self, dst, src, stream, convert_on_device):
        """pageable host view src (N, C, z, y, x) -> device view dst, slabs of z planes through the pinned slots"""
        step = self._planes(src)
        for z in range(0, src.shape[2], step):
            part = src[:, :, z:z + step]
            k, view, dev_view = self._slot(part.shape, part.dtype)
            _host_copy(view, part)
            d = dst[:, :, z:z + step]
            if dev_view is not None:              # contiguous DMA into the slot's device twin, then the scatter (and a dtype conversion) as one device copy
                dev_view.copy_(view, non_blocking=True)
                d.copy_(dev_view)
            elif part.dtype == d.dtype or not convert_on_device:
                d.copy_(view, non_blocking=True)
            else:                                 # (another dtype than the model's: converted on the device, not by the host thread)
                stage = torch.empty(part.shape, dtype=part.dtype, device=d.device)
                stage.copy_(view, non_blocking=True)
                d.copy_(stage)
            self.events[k] = torch.cuda.Event()
            self.events[k].record(stream)

    def d2h(self, dst, src, stream):
        """device view src -> pageable host view dst"""
        step = self._planes(src)
        for z in range(0, src.shape[2], step):
            part = src[:, :, z:z + step]
            k, view, dev_view = self._slot(part.shape, part.dtype)
            if dev_view is not None:              # gather into the slot's device twin, contiguous DMA from there
                dev_view.copy_(part)
                view.copy_(dev_view, non_blocking=True)
            else:
                view.copy_(part, non_blocking=True)
            self.events[k] = torch.cuda.Event()
            self.events[k].record(stream)
            self.pending[k] = (dst[:, :, z:z + step], view)

    def reset(self):
        """Forget events and pending copies (a predict() that raised midway must not leave them to the next one) -- after the copy engines are done with the
        page-locked slots: a DMA still in flight would otherwise race the next call's staging of the same slot."""
        for ev in self.events:
            if ev is not None:
                ev.synchronize()
        self.events = [None, None]
        self.pending = [None, None]
        self.i = 0

    def flush(self):
        for k in (self.i, self.i ^ 1):
            if self.events[k] is not None:
                self.events[k].synchronize()
                self.events[k] = None
            if self.pending[k] is not None:
                host, view = self.pending[k]
                _host_copy(host, view)
                self.pending[k] = None


_RINGS = {}
_RING_LOCKS = {}
_RINGS_GUARD = __import__('threading').Lock()
_RING_SLOT_BYTES = 256 << 20


def _rings_for(device, slot_bytes=_RING_SLOT_BYTES):
    """(upload ring, download ring) of a device: allocated on the FIRST pipelined predict() of the process (not when a Predictor is constructed:
    device-resident or small inputs never take the pipelined path), slots sized for the slabs that path moves (capped at 256 MB each, 1 GB
    of page-locked memory at most).  None with E3_PREDICTOR_NO_PINNED=1, or when the page-locked allocation fails (memlock limit of a
    container, small host): the caller then takes the runtime's pageable copies."""
    if os.environ.get('E3_PREDICTOR_NO_PINNED') is not None:
        return None
    key = torch.device(device).index or 0
    slot_bytes = int(max(1 << 20, min(_RING_SLOT_BYTES, slot_bytes)))
    with _RINGS_GUARD:
        have = _RINGS.get(key)
        if have is False:                       # an earlier allocation failed: do not retry per call
            return None
        if have is not None and have[0].slots[0].numel() >= slot_bytes:
            return have
        try:
            _RINGS[key] = (_PinnedRing(slot_bytes, torch.device(device)), _PinnedRing(slot_bytes, torch.device(device)))
        except RuntimeError as e:               # hipHostMalloc failed
            logger.warning(f'Predictor: no page-locked staging buffers ({e}); using pageable host copies')
            _RINGS[key] = False
            return None
        return _RINGS[key]


def _ring_lock(device):
    """The staging rings of a device serve one pipelined predict() at a time (Predictors on several threads take turns)."""
    with _RINGS_GUARD:
        return _RING_LOCKS.setdefault(torch.device(device).index or 0, __import__('threading').Lock())


class _SharedHostTensor:
    """A host tensor backed by a file in /dev/shm that the ranks of one node map together (np.memmap): the tile-parallel
    Predictor's output buffer.  The creator unlinks the name once every rank has finished writing; the mappings stay valid
    for as long as the tensors live."""

    def __init__(self, name, shape, dtype, create):
        import tempfile
        nbytes = max(int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size(), 1)
        self.owner = create
        if create:
            shm_dir = '/dev/shm' if os.path.isdir('/dev/shm') else tempfile.gettempdir()
            fd, name = tempfile.mkstemp(prefix='e3pred_', dir=shm_dir)
            os.close(fd)
        self.name = name
        buf = np.memmap(name, dtype=np.uint8, mode='w+' if create else 'r+', shape=(nbytes,))
        self.tensor = torch.from_numpy(buf).view(dtype).view(*shape)     # (the tensor keeps the mapping alive)

    @classmethod
    def create(cls, shape, dtype):
        return cls(None, shape, dtype, True)

    @classmethod
    def open(cls, name, shape, dtype):
        return cls(name, shape, dtype, False)

    def unlink_if_owner(self):
        if self.owner:
            try:
                os.unlink(self.name)
            except FileNotFoundError:
                pass


_ROI = os.environ.get('E3_PREDICTOR_NO_ROI') is None      # (A/B switch: whole tiles instead of the needed region)
_NO_CLIP = os.environ.get('E3_PREDICTOR_NO_CLIP') is not None      # (A/B switch: edge tiles compute their whole kept crop, also the part beyond the volume's end)


def _extend_nc(spatial_slice):
    return (slice(None), slice(None)) + tuple(spatial_slice)


def tile_plan(out_spatial, tile_shape, overlap_shape):
    """Visiting order and coordinates of the reference's tile loop (inference.py:153-189): C-order product over tile
    indices; input slab [tile*pos, tile*(pos+1) + 2*overlap) in padded coordinates, output slab [tile*pos, tile*(pos+1))."""
    tile_shape, overlap_shape = np.asarray(tile_shape), np.asarray(overlap_shape)
    tiles = np.ceil(np.asarray(out_spatial) / tile_shape).astype(int)
    plan = []
    for pos in itertools.product(*[range(int(t)) for t in tiles]):
        pos = np.array(pos)
        lo, hi = tile_shape * pos, tile_shape * (pos + 1)
        plan.append((tuple(int(v) for v in lo), tuple(int(v) for v in hi + 2 * overlap_shape),
                     tuple(int(v) for v in lo), tuple(int(v) for v in hi)))
    return plan


def tiled_apply(func, inp, tile_shape, overlap_shape, offset, out_shape, verbose=False, device=None,
                tile_indices=None, out=None):
    """Splits ``inp`` into overlapping tiles, applies ``func(tile, crop_slice)`` to each and assembles the central
    regions (inference.py:45-199).  ``inp`` may live on any device; padded input and output are allocated on
    ``device`` (default: ``inp``'s device).  ``tile_indices`` restricts the loop to a subset (tile-parallel ranks)."""
    if not (inp.dim() - 2 == len(tile_shape) == len(overlap_shape)):
        raise ValueError(f'ndims of tile shape ({len(tile_shape)}) and overlap shape ({len(overlap_shape)}) don\'t match input '
                         f'shape ndim - 2({inp.dim() - 2}).')
    if not np.all(np.mod(out_shape[2:], tile_shape) == 0):
        raise ValueError(f'spatial out shape[2:] {tuple(out_shape[2:])} has to be divisible by tile_shape {tile_shape}.')
    device = torch.device(device) if device is not None else inp.device
    if offset is not None:
        offset = np.array(offset)
    inp_shape = np.array(inp.shape)
    out_shape = np.array(out_shape)
    tile_shape = np.array(tile_shape)
    overlap_shape = np.array(overlap_shape)
    final_crop_slice = None
    if not np.array_equal(out_shape[2:], inp_shape[2:]):   # input is already padded (valid-conv networks)
        inp_padded = inp.to(device)
    else:
        padded_shape = inp_shape + np.array((0, 0, *overlap_shape * 2))
        logger.info(f'additional input padding to {padded_shape}')
        inp_padded = torch.zeros(tuple(int(v) for v in padded_shape), dtype=inp.dtype, device=device)
        inp_padded[_extend_nc([slice(int(l), int(h)) for l, h in zip(overlap_shape, padded_shape[2:] - overlap_shape)])] = \
            inp.to(device, non_blocking=True)
        final_crop_slice = _extend_nc([slice(int(l), int(h)) for l, h in zip(overlap_shape, tile_shape + overlap_shape)])
    if offset is not None:   # no cropping necessary for valid conv
        final_crop_slice = None
    del inp
    plan = tile_plan(out_shape[2:], tile_shape, overlap_shape)
    it = range(len(plan)) if tile_indices is None else tile_indices
    if verbose:
        try:
            from tqdm import tqdm
            it = tqdm(it, 'Predicting', total=len(plan) if tile_indices is None else len(tile_indices), dynamic_ncols=True)
        except ImportError:
            pass
    for ti in it:
        ilo, ihi, olo, ohi = plan[ti]
        assert all(h <= s for h, s in zip(ihi, inp_padded.shape[2:])), ihi
        inp_tile = inp_padded[_extend_nc([slice(l, h) for l, h in zip(ilo, ihi)])].contiguous()
        out_tile = func(inp_tile, final_crop_slice)
        if out is None:
            out = torch.zeros([int(v) for v in out_shape], dtype=out_tile.dtype, device=device)
        out[_extend_nc([slice(l, h) for l, h in zip(olo, ohi)])] = out_tile
    if out is None:
        out = torch.zeros([int(v) for v in out_shape], dtype=inp_padded.dtype, device=device)
    return out


class Argmax(nn.Module):
    def __init__(self, dim=1, unsqueeze=True):
        super().__init__()
        self.dim, self.unsqueeze = dim, unsqueeze

    def forward(self, x):
        a = torch.argmax(x, self.dim)
        return a.unsqueeze(1) if self.unsqueeze else a


class FlipAugment:
    def __init__(self, dims):
        self.dims = tuple(int(d) + 2 for d in dims)   # skip (N, C)

    def forward(self, inp):
        return torch.flip(inp, dims=self.dims)

    def backward(self, inp):
        return self.forward(inp)


DEFAULT_AUGMENTATIONS_3D = [FlipAugment(d) for d in [(0,), (1,), (0, 1), (2,), (0, 2), (1, 2), (0, 1, 2)]]
DEFAULT_AUGMENTATIONS_2D = DEFAULT_AUGMENTATIONS_3D[:3]


def calculate_offset(model, tile_shape=None):
    """Half the spatial shape difference between a model's input and output (elektronn3/data/utils.py:63-78)."""
    with torch.no_grad():
        param = next(model.parameters())
        shapes = [tuple(tile_shape)] if tile_shape else [(90, 90, 90), (186, 186)]
        last = None
        for sh in shapes:
            try:
                ex = torch.randn(1, param.size()[1], *sh, device=param.device, dtype=param.dtype)
                out = model.eval().forward(ex)
                return np.subtract(ex.shape[2:], out.shape[2:]) // 2
            except (RuntimeError, ValueError) as e:      # (wrong rank for this model: try the next probe shape)
                last = e
        raise last


def set_state_dict(model, state_dict):
    """Also accepts state dicts saved from ``nn.DataParallel`` wrappers (inference.py:698-710)."""
    try:
        model.load_state_dict(state_dict)
    except RuntimeError:
        model.load_state_dict(OrderedDict((k.replace('module.', ''), v) for k, v in state_dict.items()))


def _open_model(model, device):
    """A module as given, or loaded from a TorchScript archive (.pts) / a pickled module (.pt) -- the three forms
    ``Predictor(model=...)`` accepts (inference.py:421-433)."""
    if isinstance(model, Path):
        model = str(model)
    if not isinstance(model, str):
        return model
    if not os.path.isfile(model):
        raise ValueError(f'Model path {model} not found.')
    loaders = {'.pts': lambda f: torch.jit.load(f, map_location=device),
               '.pt': lambda f: torch.load(f, map_location=device, weights_only=False)}
    ext = os.path.splitext(model)[1]
    if ext not in loaders:
        raise ValueError(f'{model} has an unkown file extension. Supported are .pt and .pts')     # (sic: the reference's message)
    return loaders[ext](model)


def _resolve_state_dict(src):
    """None, a state_dict, or a path to a checkpoint that is one or contains one under 'model_state_dict' (inference.py:434-441)."""
    if src is None or isinstance(src, dict):
        return src
    if isinstance(src, str):
        loaded = torch.load(src)
        return loaded.get('model_state_dict', loaded)
    raise ValueError('"state_dict_src" has to be either a path to a .pth file (str), a state_dict object (dict) or None.')


class _OutputStages:
    """What happens to the network's logits before a tile is stored: softmax, test-time-augmentation mean, threshold + argmax.
    The reference expresses this by wrapping the model in nn.Sequential layers (inference.py:443-456); for the native UNet the softmax
    is a flag of the last kernel and the remaining stages run on its output."""

    def __init__(self, apply_softmax, augmentations, apply_argmax, threshold):
        if not apply_softmax and augmentations is not None:
            raise ValueError('When augmentations are enabled, apply_softmax cannot be False.')
        self.softmax = bool(apply_softmax or augmentations is not None)
        self.wants_argmax = bool(apply_argmax or threshold is not None)
        self.threshold = threshold
        # with test-time augmentation the arg-max is taken of the MEAN of the augmented predictions, i.e. after the model
        self.argmax_after_tta = self.wants_argmax and augmentations is not None

    def layers_after_softmax(self):
        if not self.wants_argmax or self.argmax_after_tta:
            return []
        stages = [Argmax(dim=1, unsqueeze=True)]
        if self.threshold:
            stages.insert(0, nn.Threshold(self.threshold, 0))
        return stages


class _Tiling:
    """tile_shape / overlap_shape / offset / out_shape as one consistent set (inference.py:460-494): either all unset (whole input in
    one call), or a tile grid with a zero-padded overlap ('same' networks) or an offset (networks whose output is smaller)."""

    def __init__(self, tile_shape, overlap_shape, offset, out_shape, estimate_offset):
        given = lambda a: a is not None and np.any(a)
        if given(overlap_shape) and given(offset):
            raise ValueError(f'overlap_shape={overlap_shape} and offet={offset} are both specified, but this is not supported.\n'
                             'Either specify overlap_shape (if the spatial shape of inputs and outputs are the same)\n'
                             'or offset (if the output is smaller).')
        self.enabled = bool(given(tile_shape))
        if not self.enabled:
            assert not (given(out_shape) or given(overlap_shape) or given(offset)), \
                'If tile_shape is not set, out_shape, overlap_shape and offset should not be set either.'
        else:
            assert given(out_shape), 'If tile_shape is set, out_shape is required to be set, too.'
            if offset is None:
                offset = estimate_offset(len(tile_shape))
            if np.count_nonzero(offset) == 0:
                offset = None
            else:          # the halo a tile needs IS the offset, and the output volume shrinks by it on every side
                offset = np.array(offset)
                overlap_shape = offset
                nsp = len(offset)
                out_shape = np.array([*out_shape[:-nsp], *(out_shape[-nsp:] - 2 * offset)])
                logger.info(f'Adjusted out_shape: {out_shape}')
        as_array = lambda a: np.array(a) if a is not None else None
        self.offset = offset
        self.tile_shape, self.overlap_shape, self.out_shape = as_array(tile_shape), as_array(overlap_shape), as_array(out_shape)


class Predictor:
    """Tiled sliding-window inference with the reference's interface (inference.py:368-388)."""

    def __init__(
            self,
            model: Union[nn.Module, str, Path],
            state_dict_src: Optional[Union[str, dict]] = None,
            device: Optional[Union[torch.device, str]] = None,
            batch_size: Optional[int] = None,
            tile_shape: Optional[Tuple[int, ...]] = None,
            overlap_shape: Optional[Tuple[int, ...]] = None,
            offset: Optional[Tuple[int, ...]] = None,
            out_shape: Optional[Tuple[int, ...]] = None,
            out_dtype: Optional[torch.dtype] = None,
            float16: bool = False,
            apply_softmax: bool = True,
            transform: Optional[Transform] = None,
            augmentations: Union[int, Optional[Sequence]] = None,
            strict_shapes: bool = False,
            apply_argmax: bool = False,
            argmax_with_threshold: Optional[float] = None,
            verbose: bool = False,
            report_inp_stats: bool = False,
            tile_parallel: bool = False,
    ):
        from .unet import UNet
        self.device = torch.device(device) if device is not None else torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.batch_size, self.out_dtype, self.float16 = batch_size, out_dtype, float16
        self.dtype = torch.float16 if float16 else torch.float32
        self.transform, self.strict_shapes, self.verbose = transform, strict_shapes, verbose
        self.apply_argmax, self.argmax_with_threshold = apply_argmax, argmax_with_threshold
        self.report_inp_stats, self.tile_parallel = report_inp_stats, tile_parallel
        self.augmentations = DEFAULT_AUGMENTATIONS_3D[:augmentations] if isinstance(augmentations, int) else augmentations
        self._warn_about_shapes = True

        # ---- the network: a module (possibly from a file), optionally other weights, optionally cast to half precision
        net = _open_model(model, self.device)
        if float16 and isinstance(net, nn.Module) and not isinstance(model, (str, Path)) and next(net.parameters()).dtype != torch.float16:
            net = copy.deepcopy(net)       # .half() is in-place: the caller's fp32 module stays intact (inference.py:402-407)
        weights = _resolve_state_dict(state_dict_src)
        if weights is not None:
            set_state_dict(net, weights)
        # a module that already lives in bfloat16 (model.to(torch.bfloat16), BASELINE cfg 3's storage type) gets bfloat16 tiles: the
        # reference would feed it fp32 tiles and fail in the first conv; here it selects the native bf16 kernels (the bf16 counterpart of
        # float16=True).  Outputs default to bfloat16 like float16=True defaults to float16; pass out_dtype=torch.float32 for fp32 volumes.
        if not float16 and isinstance(net, nn.Module):
            p0 = next(net.parameters(), None)
            if p0 is not None and p0.dtype in (torch.bfloat16, torch.float16):      # (a module already in half precision: same as float16=True)
                self.dtype = p0.dtype

        # ---- output stages.  Native UNet: softmax inside the last kernel, the rest as a small post-module; any other module is wrapped
        # the way the reference wraps it
        stages = _OutputStages(apply_softmax, self.augmentations, apply_argmax, argmax_with_threshold)
        self._native = isinstance(net, UNet)
        self._softmax = stages.softmax
        self.apply_argmax_after_tta = stages.argmax_after_tta
        tail = stages.layers_after_softmax()
        self._post = nn.Sequential(*tail) if (self._native and tail) else None
        if self._native:
            self.model = net
        else:
            wrapped = [net] + ([nn.Softmax(1)] if stages.softmax else []) + tail
            self.model = net if len(wrapped) == 1 else nn.Sequential(*wrapped)
        if stages.wants_argmax and self.out_dtype is None:
            self.out_dtype = torch.uint8
        if float16:
            self.model.half()
        self.model.eval()                  # (side effect of the reference, inference.py:458: the caller's module leaves train mode)
        if isinstance(self.model, nn.Module) and self.device.type == 'cuda':
            self.model.to(self.device)

        # ---- tiling geometry
        def estimate_offset(ndim):
            if self._native and getattr(net, 'conv_mode', 'same') == 'same':
                return np.zeros(ndim, dtype=np.int64)           # 'same' convolutions: known without a probe forward
            logger.warning('Predictor: offset=None -> Estimating offset from forward pass.')
            return calculate_offset(self.model)

        geo = _Tiling(tile_shape, overlap_shape, offset, out_shape, estimate_offset)
        self.enable_tiling = geo.enabled
        self.offset, self.tile_shape, self.overlap_shape, self.out_shape = geo.offset, geo.tile_shape, geo.overlap_shape, geo.out_shape

    # ------------------------------------------------------------------ per-tile model call (inference.py:496-525)
    def _call_model(self, dinp, crop_slice=None, keep=None):
        """``crop_slice`` (the central crop that follows): the native UNet is told that only those voxels are wanted and skips what they
        do not depend on (``UNet.forward_roi``; E3_PREDICTOR_NO_ROI=1: A/B switch, whole tiles)."""
        if self._native:
            roi = None
            if crop_slice is not None and dinp.dim() == 5 and _ROI and hasattr(self.model, 'forward_roi'):
                sp = tuple(dinp.shape[2:])
                roi = tuple(sl.indices(n)[:2] for sl, n in zip(crop_slice[-3:], sp))
                if keep is not None:      # an edge tile of the pipelined run: only the first `keep` voxels of the crop lie inside the volume
                    roi = tuple((lo, min(hi, lo + int(kp))) for (lo, hi), kp in zip(roi, keep))
            if roi is not None:
                y = self.model.forward_roi(dinp, roi, softmax=self._softmax)
            else:
                y = self.model.forward_softmax(dinp) if self._softmax else self.model(dinp)
            return self._post(y) if self._post is not None else y
        return self.model(dinp)

    @torch.no_grad()
    def _predict(self, dinp, crop_slice=None, keep=None):
        """One tile: model call (+ test-time augmentation mean, + arg-max) and the central crop (inference.py:496-525)."""
        dinp = dinp.to(self.device, dtype=self.dtype)
        crop = (lambda t: t[crop_slice]) if crop_slice is not None else (lambda t: t)
        dout = crop(self._call_model(dinp, crop_slice, keep))
        if self.augmentations is not None:       # mean over the identity and every flip (prediction of the flipped tile, flipped back)
            votes = [dout] + [crop(aug.backward(self._call_model(aug.forward(dinp)))) for aug in self.augmentations]
            dout = torch.stack(votes).mean(dim=0)
        if self.apply_argmax_after_tta:
            if self.argmax_with_threshold:
                dout[dout <= self.argmax_with_threshold] = 0
            dout = dout.argmax(dim=1)
        return dout.to(self.out_dtype)

    def _frozen(self):
        """Nobody changes the weights during a predict() call: the native model packs them for the first tile only (UNet.frozen_weights)."""
        import contextlib
        return self.model.frozen_weights() if (self._native and hasattr(self.model, 'frozen_weights')) else contextlib.nullcontext()

    def _tiled_predict(self, inp, out_shape=None):
        if not self.enable_tiling:
            return self._predict(inp)
        if self.out_shape is None:
            raise ValueError('If you use tiling, you also need to supply out_shape.')
        out_shape = (inp.shape[0], *out_shape)
        tile_indices, world, rank = None, 1, 0
        if self.tile_parallel and torch.distributed.is_available() and torch.distributed.is_initialized():
            world, rank = torch.distributed.get_world_size(), torch.distributed.get_rank()
            ntiles = int(np.prod(np.ceil(np.asarray(out_shape[2:]) / self.tile_shape)))
            tile_indices = list(range(rank, ntiles, world))     # static round-robin shard: independent units, no exchange
        out = tiled_apply(self._predict, inp=inp, tile_shape=self.tile_shape, overlap_shape=self.overlap_shape,
                          offset=self.offset, out_shape=out_shape, verbose=self.verbose, device=self.device,
                          tile_indices=tile_indices)
        if world > 1:   # rank 0 collects the disjoint slabs (zeros elsewhere => a sum assembles them exactly)
            red = out if out.dtype in (torch.float32, torch.float16, torch.int32, torch.int64) else out.to(torch.int32)
            torch.distributed.reduce(red, dst=0, op=torch.distributed.ReduceOp.SUM)
            out = red.to(out.dtype)
        return out

    def _splitbatch_predict(self, inp, num_batches, out_shape=None):
        if self.out_shape is None:
            raise ValueError('If you define a batch_size, you also need to supply out_shape.')
        outs = []
        for k in range(num_batches):
            outs.append(self._tiled_predict(inp[self.batch_size * k:self.batch_size * (k + 1)], out_shape=out_shape))
        return torch.cat(outs, 0)

    # ------------------------------------------------------------------ host <-> device pipeline
    def _pipeline_applicable(self, inp):
        return (self.enable_tiling and self.device.type == 'cuda' and isinstance(inp, torch.Tensor) and not inp.is_cuda
                and self.offset is None and self.out_shape is not None and self.tile_shape is not None
                and self.overlap_shape is not None and len(self.tile_shape) == 3 and inp.dim() == 5
                and tuple(inp.shape[2:]) == tuple(int(v) for v in self.out_shape[1:])
                and (self.batch_size is None or self.batch_size >= inp.shape[0])
                and os.environ.get('E3_PREDICTOR_NO_PIPELINE') is None)

    def _dist(self):
        """(world, rank) of the tile-parallel run, (1, 0) otherwise."""
        if self.tile_parallel and torch.distributed.is_available() and torch.distributed.is_initialized():
            return torch.distributed.get_world_size(), torch.distributed.get_rank()
        return 1, 0

    def _ring_bytes(self, inp):
        """Bytes of the largest slab the pipelined path moves in one piece for this input (a z row of tiles incl. its halo going up; a z row of
        output going down, at most 4 bytes x 16 channels per voxel) -- the page-locked slots need not be larger (cap: 256 MB)."""
        N, Cin = int(inp.shape[0]), int(inp.shape[1])
        z = int(self.tile_shape[0] + 2 * self.overlap_shape[0])
        plane = int(np.prod(self.out_shape[2:]))
        up = N * Cin * z * plane * inp.element_size()
        down = N * 16 * int(self.tile_shape[0]) * plane * 4
        return min(_RING_SLOT_BYTES, max(up, down))

    def prepare(self, inp_like=None):
        """Optional: allocate what the host <-> device pipeline of predict() needs once per process (the page-locked staging slots, up to 1 GB)
        before the first call instead of inside it.  `inp_like`: a tensor / array of the shape and dtype predict() will get (default: the
        largest slots)."""
        if self.enable_tiling and self.device.type == 'cuda' and self.out_shape is not None:
            nbytes = _RING_SLOT_BYTES
            if inp_like is not None:
                t = torch.as_tensor(inp_like)
                if t.dim() == 5:
                    nbytes = self._ring_bytes(t)
            with _ring_lock(self.device):
                _rings_for(self.device, nbytes)
        return self

    @torch.no_grad()
    def _pipelined_predict(self, inp):
        """Same result as the plain path (pad to a multiple of the tile shape, zero halo, tiles in the reference's C order,
        central crops assembled), but the volume streams through the GPU: the input is uploaded in z slabs just ahead of the
        tile rows that need them, and every finished row of output tiles goes back to the host (cropped to out_shape) on a
        side stream while the next rows are computed.  Host copies are done by two worker threads (pageable memory copies
        block their calling thread, not the GPU); the compute stream only waits on events.

        Tile-parallel (``tile_parallel=True`` under torch.distributed, one process per GPU of ONE node): the
        (z, y) rows of tiles are split into contiguous shares, one per rank; a rank uploads only the z slabs its rows need and
        writes its finished rows straight into ONE output buffer in POSIX shared memory that all ranks map -- independent
        units, no data-path collective, eight PCIe links used in parallel.  Every rank returns that (complete) tensor."""
        from concurrent.futures import ThreadPoolExecutor
        import time
        dev = self.device
        t_start = time.perf_counter()
        cpu_start = time.thread_time()
        ev_first, ev_last = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        world, rank = self._dist()
        N, Cin = int(inp.shape[0]), int(inp.shape[1])
        real = np.array(self.out_shape[1:], dtype=np.int64)
        tile, ov = self.tile_shape.astype(np.int64), self.overlap_shape.astype(np.int64)
        if np.any(real % tile) and self.strict_shapes:
            raise ValueError('Make sure that out_shape is divisible by tile_shape or relax this constraint by setting '
                             'strict_shapes=False.')
        padded = (np.ceil(real / tile) * tile).astype(np.int64)           # spatial shape the tile loop works on
        ntz, nty, ntx = (int(v) for v in padded // tile)
        if self.out_dtype is None:
            self.out_dtype = torch.uint8 if self.argmax_with_threshold is not None else self.dtype      # (inp is cast to the compute dtype first, inference.py:606-614)
        t_alloc = time.perf_counter()
        inp_padded = torch.zeros((N, Cin, *(int(v) for v in padded + 2 * ov)), dtype=self.dtype, device=dev)
        alloc_s = {'input': time.perf_counter() - t_alloc, 'output': 0.0}       # host time of the two big device allocations (last_timing)
        crop = _extend_nc([slice(int(l), int(h)) for l, h in zip(ov, tile + ov)])
        plan = tile_plan(padded, tile, ov)
        main = torch.cuda.current_stream(dev)
        up_stream, down_stream = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        up_stream.wait_stream(main); down_stream.wait_stream(main)       # (the zero fill of inp_padded precedes every upload)
        # this rank's share of the (z row k, y row j) rows of tiles, in the reference's visiting order
        rows = [(k, j) for k in range(ntz) for j in range(nty)]
        per = -(-len(rows) // world)                                      # contiguous shares; rank 0 always has work
        mine = rows[min(len(rows), per * rank):min(len(rows), per * (rank + 1))]
        zrows = sorted({k for k, _ in mine})
        # z planes of the ORIGINAL input that tile row k needs: padded coords [tile*k, tile*(k+1) + 2 ov) = original - ov
        need_lo = {k: int(max(0, tile[0] * k - ov[0])) for k in zrows}
        need_hi = {k: int(min(real[0], tile[0] * (k + 1) + ov[0])) for k in zrows}
        up_events = {k: torch.cuda.Event() for k in zrows}
        uploaded = [None]                                                 # z plane up to which the input is on the device
        # pinned staging rings (one per direction and device, allocated on first use -- or ahead of time by prepare()): E3_PREDICTOR_NO_PINNED=1
        # = the runtime's pageable copies
        rings = _rings_for(dev, self._ring_bytes(inp))

        def put(dst, src):
            """host slab -> device view.  A volume in another dtype than the model's (fp32 volume, bf16 / float16 model) travels as it is and is
            converted on the device: converting 2 G voxels on the host first made the upload thread the bottleneck of the 16-bit Predictor."""
            if rings is not None and rings[0].fits(src):
                rings[0].h2d(dst, src, up_stream, True)
            elif src.dtype == dst.dtype:
                dst.copy_(src)
            else:
                stage = torch.empty(src.shape, dtype=src.dtype, device=dev)
                stage.copy_(src)
                dst.copy_(stage)

        busy = {'up': 0.0, 'down': 0.0}                                   # wall time the two copy workers spent in their jobs (last_timing)

        def upload(k):
            t_job = time.perf_counter()
            a = need_lo[k] if uploaded[0] is None else max(uploaded[0], need_lo[k])
            b = need_hi[k]
            with torch.cuda.stream(up_stream):
                if b > a:
                    dst = inp_padded[:, :, int(ov[0]) + a:int(ov[0]) + b, int(ov[1]):int(ov[1] + real[1]), int(ov[2]):int(ov[2] + real[2])]
                    put(dst, inp[:, :, a:b])
                uploaded[0] = b if uploaded[0] is None else max(uploaded[0], b)
                up_events[k].record(up_stream)
            busy['up'] += time.perf_counter() - t_job

        # the FIRST z slab goes up in y pieces, one per row of tiles, so that the first row starts after 1 / nty of the slab has arrived (the
        # cfg-5 volume: 0.24 GB instead of 2.1 GB of pageable host memory in front of the first tile); later slabs travel while tiles compute
        k_first = zrows[0] if zrows else None
        piece_events = [torch.cuda.Event() for _ in range(nty)]

        def upload_piece(j):
            t_job = time.perf_counter()
            a, b = need_lo[k_first], need_hi[k_first]
            y0 = 0 if j == 0 else int(min(real[1], tile[1] * j + ov[1]))
            y1 = int(real[1]) if j == nty - 1 else int(min(real[1], tile[1] * (j + 1) + ov[1]))
            with torch.cuda.stream(up_stream):
                if b > a and y1 > y0:
                    dst = inp_padded[:, :, int(ov[0]) + a:int(ov[0]) + b, int(ov[1]) + y0:int(ov[1]) + y1, int(ov[2]):int(ov[2] + real[2])]
                    put(dst, inp[:, :, a:b, y0:y1])
                piece_events[j].record(up_stream)
                if j == nty - 1:
                    uploaded[0] = b
                    up_events[k_first].record(up_stream)
            busy['up'] += time.perf_counter() - t_job

        state = {'host_out': None, 'out_dev': None, 'shm': None}
        fill_events = []                                                  # around the zero fill of the output buffer on the compute stream (last_timing)
        downs = []
        call_s = []                                                       # wall time of every in-place tile call on the issuing thread
        # no tile copy / crop copy around the native fp32 model (E3_PREDICTOR_NO_INPLACE=1: A/B switch)
        in_place = (self._native and _ROI and self._post is None and self.augmentations is None and not self.apply_argmax_after_tta
                    and self.dtype == torch.float32 and self.out_dtype == torch.float32 and hasattr(self.model, 'forward_tile')
                    and os.environ.get('E3_PREDICTOR_NO_INPLACE') is None)

        def make_outputs(out_tile, meta=None):
            """Output buffers, once the first tile tells channel layout and dtype (world > 1: every rank calls this exactly once,
            whether it has tiles or not -- the shared-memory name travels in one broadcast_object_list)."""
            if out_tile is not None:
                meta = (tuple(int(v) for v in out_tile.shape[1:-3]), out_tile.dtype)
            if world > 1:
                box = [None]
                if rank == 0:
                    assert meta is not None
                    shape = (N, *meta[0], *(int(v) for v in real))
                    state['shm'] = _SharedHostTensor.create(shape, meta[1])
                    box = [(state['shm'].name, shape, meta[1])]
                torch.distributed.broadcast_object_list(box, src=0)
                if rank != 0:
                    state['shm'] = _SharedHostTensor.open(*box[0])
                    meta = (tuple(box[0][1][1:-3]), box[0][2])
                state['host_out'] = state['shm'].tensor
            else:
                state['host_out'] = torch.empty((N, *meta[0], *(int(v) for v in real)), dtype=meta[1])
                _advise_hugepages(state['host_out'])
            t_out = time.perf_counter()
            fill_events.extend([torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)])
            fill_events[0].record(main)
            state['out_dev'] = torch.zeros((N, *meta[0], *(int(v) for v in padded)), dtype=meta[1], device=dev)
            fill_events[1].record(main)
            alloc_s['output'] = time.perf_counter() - t_out

        def download(k, j0, j1, ev):
            t_job = time.perf_counter()
            with torch.cuda.stream(down_stream):
                down_stream.wait_event(ev)
                z0, z1 = int(tile[0] * k), int(min(tile[0] * (k + 1), real[0]))
                y0, y1 = int(tile[1] * j0), int(min(tile[1] * j1, real[1]))
                if z1 > z0 and y1 > y0:
                    if rings is not None and rings[1].fits(state['out_dev'][:, :, z0:z1, y0:y1, :int(real[2])]):
                        rings[1].d2h(state['host_out'][:, :, z0:z1, y0:y1], state['out_dev'][:, :, z0:z1, y0:y1, :int(real[2])], down_stream)
                    else:
                        state['host_out'][:, :, z0:z1, y0:y1].copy_(state['out_dev'][:, :, z0:z1, y0:y1, :int(real[2])])
            busy['down'] += time.perf_counter() - t_job

        row_events = []                                                   # (first, last) event of every row of tiles on the compute stream, behind its waits
        lead = {}                                                         # host time in front of the first tile (last_timing)
        with ThreadPoolExecutor(max_workers=1) as up_pool, ThreadPoolExecutor(max_workers=1) as down_pool:
            pieces = [up_pool.submit(upload_piece, j) for j in range(nty)] if zrows else []
            ups = {k: up_pool.submit(upload, k) for k in zrows[1:]}
            # the output buffers NOW, while the first piece of the input travels -- allocated inside the first row of tiles (behind the stream's wait for the upload,
            # beside the upload worker's copies) they stalled the compute stream by 0.49 s in every third process (profiles/r06_predictor_modes.md).  Layout and
            # dtype of a native module's result are known without a first tile: (out_channels,) and out_dtype, unless an argmax / threshold stage follows
            lead['to_uploads_submitted_s'] = time.perf_counter() - t_start
            early_meta = None
            if self._native and self._post is None and not self.apply_argmax_after_tta and hasattr(self.model, 'out_channels') and (in_place or world == 1):
                early_meta = ((int(self.model.out_channels),), torch.float32 if in_place else self.out_dtype)
                make_outputs(None, early_meta)
            elif not mine:
                make_outputs(None)
            for i, (k, j) in enumerate(mine):
                if k == k_first:
                    t_wait = time.perf_counter()
                    pieces[j].result()                # (the copy has been issued; the stream-side wait is the event)
                    main.wait_event(piece_events[j])
                    if i == 0:
                        lead['first_piece_wait_s'] = time.perf_counter() - t_wait
                        lead['to_first_tile_s'] = time.perf_counter() - t_start
                if i == 0 or mine[i - 1][0] != k:
                    if k != k_first:
                        ups[k].result()
                        main.wait_event(up_events[k])
                    j_first = j
                    if i == 0:
                        ev_first.record(main)
                row_events.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                row_events[-1][0].record(main)
                for ti in range((k * nty + j) * ntx, (k * nty + j + 1) * ntx):
                    ilo, ihi, olo, ohi = plan[ti]
                    if in_place:          # the tile is read where it lies and its kept region written where it belongs (UNet.forward_tile)
                        try:
                            if state['out_dev'] is None:
                                make_outputs(None, ((int(self.model.out_channels),), torch.float32))
                            # kept region of the tile, clipped to the REAL volume: the last tile of an axis hangs over the volume's end (the padded volume is a
                            # multiple of the tile shape), what it would write there is never downloaded -- the needed-region forward does not compute it
                            # (the cfg-5 volume: 512 = 5.33 x 96, 2048 = 10.67 x 192: 16 % of the kept voxels of all tiles lie outside; E3_PREDICTOR_NO_CLIP=1: A/B)
                            keep = [int(t) if _NO_CLIP else max(1, min(int(t), int(r) - int(lo_))) for t, r, lo_ in zip(tile, real, olo)]
                            t_call = time.perf_counter()
                            self.model.forward_tile(inp_padded, ilo, [h - l for l, h in zip(ilo, ihi)], state['out_dev'], olo,
                                                    [(int(o), int(o) + kp) for o, kp in zip(ov, keep)], softmax=self._softmax)
                            call_s.append(time.perf_counter() - t_call)
                            continue
                        except NotImplementedError:
                            in_place = False
                    inp_tile = inp_padded[_extend_nc([slice(l, h) for l, h in zip(ilo, ihi)])].contiguous()
                    out_tile = self._predict(inp_tile, crop, None if _NO_CLIP else [max(1, min(int(t), int(r) - int(lo_))) for t, r, lo_ in zip(tile, real, olo)])
                    if state['out_dev'] is None or (early_meta is not None and (tuple(int(v) for v in out_tile.shape[1:-3]), out_tile.dtype) != early_meta):
                        early_meta = None
                        make_outputs(out_tile)                            # (the first tile tells; an early guess that did not hold is replaced)
                    state['out_dev'][_extend_nc([slice(l, h) for l, h in zip(olo, ohi)])] = out_tile
                row_events[-1][1].record(main)
                last_of_zrow = i + 1 == len(mine) or mine[i + 1][0] != k
                # one rank: whole z rows go back (contiguous in host memory) -- except the LAST z row, which goes back tile row by tile row:
                # what is still to be downloaded when the last tile finishes is then one row (0.3 GB) instead of a z row (3.2 GB of the
                # cfg-5 volume, ~0.25 s of exposed PCIe time)
                per_row = world > 1 or k == zrows[-1]
                if per_row or last_of_zrow:
                    ev = torch.cuda.Event(); ev.record(main)
                    downs.append(down_pool.submit(download, k, j if per_row else j_first, j + 1, ev))
            ev_last.record(main)
            t_issued = time.perf_counter()
            cpu_issued = time.thread_time()
            for d in downs:
                d.result()
            if rings is not None:
                down_pool.submit(rings[1].flush).result()
        torch.cuda.synchronize(dev)
        # where the wall time went (bench.py reports it): the compute stream's span from the first tile to the last one (it includes waits for
        # uploads), the host time to issue the tile loop, and the wall time around both
        # (issue_s is wall time of the issuing thread INCLUDING the time it is blocked -- on the bounded launch queue of a busy GPU, on the upload workers;
        # issue_cpu_s is that thread's CPU time, which still contains the runtime's spinning on a full queue; tile_call_s is what ONE tile costs the host when
        # nothing blocks it: the lower quartile of the tile calls' wall times -- the first dozens of tiles are issued into an empty queue)
        self.last_timing = {'wall_s': time.perf_counter() - t_start, 'issue_s': t_issued - t_start, 'issue_cpu_s': cpu_issued - cpu_start,
                            'compute_stream_s': (ev_first.elapsed_time(ev_last) / 1e3) if mine else 0.0, 'tiles': len(mine) * ntx,
                            # rows_s: the compute stream's time inside the rows of tiles, i.e. compute_stream_s without its waits for uploads;
                            # upload_worker_s / download_worker_s: wall time the two copy workers spent in their jobs (host copies + issuing the DMA)
                            'rows_s': sum(a.elapsed_time(b) for a, b in row_events) / 1e3, 'upload_worker_s': busy['up'], 'download_worker_s': busy['down'],
                            'alloc_input_s': alloc_s['input'], 'alloc_output_s': alloc_s['output'],
                            'fill_output_s': (fill_events[0].elapsed_time(fill_events[1]) / 1e3) if fill_events else 0.0, **lead}
        if row_events:      # (per row of tiles, in ms: a uniformly slower GPU shifts all three, interference shows in the maximum only)
            rows_ms = [a.elapsed_time(b) for a, b in row_events]
            per_row = sorted(rows_ms)
            self.last_timing['row_ms_min_median_max'] = [round(per_row[0], 2), round(per_row[len(per_row) // 2], 2), round(per_row[-1], 2)]
            self.last_timing['slowest_rows'] = [[i, round(rows_ms[i], 2)] for i in sorted(range(len(rows_ms)), key=lambda i: -rows_ms[i])[:3]]      # (row index in issue order, ms)
        if call_s:
            srt = sorted(call_s)
            self.last_timing.update(tile_call_s=srt[len(srt) // 4], tile_call_min_s=srt[0], tile_call_median_s=srt[len(srt) // 2])
        if world > 1:
            torch.distributed.barrier()               # every rank's rows are in the shared buffer
            state['shm'].unlink_if_owner()            # the mapping stays valid; the name disappears
        return state['host_out']

    # ------------------------------------------------------------------ public API (inference.py:569-642)
    def predict(self, inp):
        """``inp``: ndarray or tensor (N, C, *spatial) -> CPU tensor (inference.py:569-642)."""
        with self._frozen():
            return self._predict_volume(inp)

    def _predict_volume(self, inp):
        t_start = time.time()
        inp = torch.as_tensor(self._transformed(inp))
        if self._pipeline_applicable(inp):                       # host volume streamed through the GPU in rows of tiles
            with _ring_lock(self.device):
                try:
                    out = self._pipelined_predict(inp)
                finally:                                     # (a call that raised midway must not leave its events / pending copies to the next one)
                    have = _RINGS.get(self.device.index or 0)
                    if have:
                        for r in have:
                            r.reset()
            self._report(t_start, out.numel())
            return out
        # device-resident path: (padded) input and output live in HBM, one upload and one download
        if self.enable_tiling:
            dev_inp, work_out_shape, keep = self._ensure_matching_shapes(inp)
        else:
            dev_inp, work_out_shape, keep = inp.to(self.device, dtype=self.dtype).contiguous(), self.out_shape, None
        n_samples, spatial = dev_inp.shape[0], np.array(dev_inp.shape[2:])
        if self.out_dtype is None:
            self.out_dtype = torch.uint8 if self.argmax_with_threshold is not None else dev_inp.dtype      # (the compute dtype)
        if work_out_shape is not None and work_out_shape[0] > 255 and self.out_dtype == torch.uint8:
            raise ValueError(f'C = out_shape[0] = {work_out_shape[0]}, but out_dtype torch.uint8 can only hold values up to 255.')
        # unset geometry means "the whole input in one piece"
        self.tile_shape = spatial if self.tile_shape is None else self.tile_shape
        self.overlap_shape = np.zeros_like(spatial) if self.overlap_shape is None else self.overlap_shape
        self.batch_size = n_samples if self.batch_size is None else self.batch_size
        n_chunks = -(-n_samples // self.batch_size)
        out = self._tiled_predict(dev_inp, work_out_shape) if n_chunks == 1 else self._splitbatch_predict(dev_inp, n_chunks, work_out_shape)
        if self.device.type == 'cuda':
            torch.cuda.synchronize(self.device)
        out = (out if keep is None else out[keep]).cpu()
        counted = out.numel()
        if work_out_shape is not None and np.array_equal(work_out_shape[2:], dev_inp.shape[2:]):      # (the reference's voxel count, :637-640)
            counted = np.prod([*out.shape[:-3], *(out.shape[-3:] - 2 * self.overlap_shape)])
        self._report(t_start, counted)
        return out

    def _transformed(self, inp):
        if self.transform is None:
            return inp
        arr = inp.numpy() if isinstance(inp, torch.Tensor) else inp
        res = np.empty_like(arr)
        for n in range(arr.shape[0]):
            res[n], _ = self.transform(arr[n], None)
        return res

    def _report(self, t_start, voxels):
        if self.verbose:
            dtime = time.time() - t_start
            print(f'Inference speed: {voxels / dtime / 1e6:.2f} MVox/s, time: {dtime:.2f}.')

    def _ensure_matching_shapes(self, inp):
        """(device input, out_shape to work on, slice that undoes the padding or None).  Where out_shape is not a multiple of tile_shape
        the input is zero-padded on the device, in the compute dtype, up to the next multiple (the reference pads on the host through a
        float64 array, inference.py:645-687)."""
        inp = inp.to(self.device, dtype=self.dtype)
        ragged = self.out_shape is not None and np.any(self.out_shape[1:] % self.tile_shape)
        if not ragged:
            return inp.contiguous(), self.out_shape, None
        if self.strict_shapes:
            raise ValueError('Make sure that out_shape is divisible by tile_shape or relax this constraint by setting '
                             'strict_shapes=False.')
        full = np.array(self.out_shape)
        full[1:] = -(-self.out_shape[1:] // self.tile_shape) * self.tile_shape
        halo = 0 if self.offset is None else 2 * np.array(self.offset)
        grown = torch.zeros((*inp.shape[:2], *(int(v) for v in full[1:] + halo)), dtype=self.dtype, device=self.device)
        grown[_extend_nc([slice(0, d) for d in inp.shape[2:]])] = inp
        if self._warn_about_shapes:
            logger.info(f'Adapting out_shape {tuple(self.out_shape[1:])} to tile_shape {tuple(self.tile_shape)} by padding '
                        f'out_shape to {tuple(full[1:])}.\nSuboptimal shapes will reduce execution speed.')
            self._warn_about_shapes = False
        return grown, full, _extend_nc([slice(0, int(d)) for d in self.out_shape[1:]])

    def predict_proba(self, inp):
        logger.warning('Predictor.predict_proba(inp) is deprecated. Please use Predictor.predict(inp) instead.')
        return self.predict(inp)
