"""``UNet`` -- drop-in for ``elektronn3.models.unet.UNet`` whose forward/backward run in libe3unet (HIP, gfx950).

Mirrors the reference's constructor signature, attribute names, sub-module tree and ``state_dict`` keys
(elektronn3/models/unet.py:755-883), so ``Trainer``, ``Predictor``, ``torch.save(model)``, ``load_state_dict`` of a
reference checkpoint, ``SWA.bn_update`` (which walks ``_BatchNorm`` modules and mutates ``momentum``,
training/swa.py:317-342) and parameter/gradient histograms keep working.  The sub-modules are ordinary
``nn.Conv3d`` / ``nn.BatchNorm3d`` / ``nn.ConvTranspose3d`` objects that only HOLD parameters and buffers: their
``forward`` is never called.  ``UNet.forward`` makes ONE native call per pass (``e3_unet_forward`` /
``e3_unet_backward``) on torch's current HIP stream; parameters, buffers and ``momentum`` are read from the module
at call time (nothing is cached across calls except scratch memory).

There is no CPU path: a CPU input raises ``RuntimeError``.
"""
import ctypes
import weakref
import threading
from typing import List, Sequence, Union

import torch
from torch import nn

from . import _lib
from ._lib import E3_BWD_CU_RESERVE, E3_BWD_FROZEN_BN, E3_FWD_FROZEN_BN, E3_FWD_REUSE_PACKED, E3_FWD_SOFTMAX, E3_FWD_TRAINING, UNetCfg, c_size_t, c_void_p, check, ptr

_plans = {}
_plans_lock = threading.Lock()
_scratch = {}
_NO_LOSS_BWD = bool(int(__import__('os').environ.get('E3_NO_LOSS_BWD', '0')))   # A/B switch: forward_with_loss' backward through e3_ce_dice_bwd + e3_unet_backward2
_NO_BF16 = bool(int(__import__('os').environ.get('E3_NO_BF16', '0')))      # A/B switch: low-precision modules on the fp32 kernels


class _Plan:
    """Native plan + the parameter table order (shared by all modules with the same configuration)."""

    def __init__(self, key):
        lib = _lib.load()
        cfg = UNetCfg(*key)
        handle = c_void_p()
        check(lib.e3_unet_plan_create(ctypes.byref(cfg), ctypes.byref(handle)))
        self.handle = handle
        self.names, self.kinds = [], []
        for i in range(lib.e3_unet_param_count(handle)):
            buf = ctypes.create_string_buffer(160)
            numel, kind = ctypes.c_int64(), ctypes.c_int()
            check(lib.e3_unet_param_info(handle, i, buf, 160, ctypes.byref(numel), ctypes.byref(kind)))
            self.names.append(buf.value.decode())
            self.kinds.append(kind.value)
        # BatchNorm layers in table order (momenta are passed in this order): the blocks' norms and, with attention=True, the BatchNorm
        # behind each GridAttention's output transform ('up_convs.i.attention.w.1')
        self.bn_names = [n[:-len('.weight')] for n in self.names if ('.norm' in n or '.attention.w.1.' in n) and n.endswith('.weight')]
        self.n_bn = lib.e3_unet_bn_count(handle)
        self.out_channels = int(key[1])
        assert self.n_bn == len(self.bn_names)
        self.conv_names = []
        for i in range(lib.e3_unet_conv_count(handle)):
            buf = ctypes.create_string_buffer(160)
            ci, co, taps, lvl = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            check(lib.e3_unet_conv_info(handle, i, buf, 160, ctypes.byref(ci), ctypes.byref(co), ctypes.byref(taps), ctypes.byref(lvl)))
            self.conv_names.append((buf.value.decode(), ci.value, co.value, taps.value, lvl.value))

    def out_dims(self, D, H, W):
        do, ho, wo = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(_lib.load().e3_unet_out_dims(self.handle, D, H, W, ctypes.byref(do), ctypes.byref(ho), ctypes.byref(wo)))
        return do.value, ho.value, wo.value

    def sizes(self, N, D, H, W, training, bf16=None):
        """bf16: None / False = fp32 path; torch.bfloat16 (or True) / torch.float16 = the native 16-bit path of that type."""
        saved, scratch = c_size_t(), c_size_t()
        lib = _lib.load()
        fn = lib.e3_unet_sizes_f16 if bf16 is torch.float16 else (lib.e3_unet_sizes_bf16 if bf16 else lib.e3_unet_sizes)
        check(fn(self.handle, N, D, H, W, int(training), ctypes.byref(saved), ctypes.byref(scratch)))
        return saved.value, scratch.value

    def bf16_supported(self):
        """True when the native bf16 kernels cover this configuration (csrc/unet_bf16.cpp); otherwise low-precision modules compute in
        fp32 on up-cast copies."""
        return bool(_lib.load().e3_unet_bf16_supported(self.handle))


def _get_plan(key):
    with _plans_lock:
        p = _plans.get(key)
        if p is None:
            p = _plans[key] = _Plan(key)
        return p


import itertools as _itertools
_scope_ids = _itertools.count(1)
_packed = {}        # scratch key -> token of the inference forward whose packed weights lie in that buffer (frozen_weights scopes)
_NO_PACK_REUSE = __import__('os').environ.get('E3_NO_PACK_REUSE') is not None      # A/B switch: every inference forward packs its weights again


def _get_scratch(device, nbytes, token=None):
    """Per-(device, stream) scratch buffer, grown on demand.  Safe to share between consecutive calls on one stream
    (everything is stream-ordered).  ``token`` (inference forwards inside a ``UNet.frozen_weights()`` scope): returns ``(buffer, reuse, key)`` --
    ``reuse`` says that the previous SUCCESSFUL call on this buffer carried the same token, i.e. its packed weights are still in place
    (E3_FWD_REUSE_PACKED).  The buffer's token is cleared here in every case; the caller records it with ``_packed_ok(key, token)`` once its
    native call has returned without an error (a call that raises half-way -- out of memory for the output, a refused argument -- must not
    leave a promise behind that the next call of the same shape would trust)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(device).cuda_stream)
    buf = _scratch.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _scratch.pop(key, None)
        _packed.pop(key, None)
        buf = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
        _scratch[key] = buf
    if token is None:
        _packed.pop(key, None)
        return buf
    reuse = _packed.pop(key, None) == token and not _NO_PACK_REUSE
    return buf, reuse, key


def _packed_ok(key, token):
    """The native inference forward that packed (or reused) the weights in scratch buffer `key` has succeeded."""
    if key in _scratch:
        _packed[key] = token


def _frozen_token(module, plan, dims, tens):
    """Token of an inference forward inside ``module.frozen_weights()`` (None outside a scope): scope, plan, shape and the parameter table's addresses."""
    scope = module.__dict__.get('_frozen_scope') if module is not None else None      # (the registered operator of the scripted module has no module at hand)
    if scope is None:
        return None
    return (scope, plan.handle.value if hasattr(plan.handle, 'value') else int(plan.handle), tuple(dims), tuple(t.data_ptr() for t in tens))


def _alloc_saved(device, nbytes):
    """The `saved` arena of one training forward (owned by its autograd node).  A function of its own so that tests can hand the library
    canary-framed memory (tests/test_unet_gpu.py::test_arenas_are_not_overrun)."""
    return torch.empty(int(nbytes), dtype=torch.uint8, device=device)


def release_scratch():
    """Drop the cached scratch buffers (they are re-created on the next call)."""
    _scratch.clear()
    _packed.clear()


class _TableRef:
    """The parameter table a pending backward will read: `shared` while it is the module's fp32 cache itself (see _fp32_table)."""
    __slots__ = ('tens', 'shared', '__weakref__')

    def __init__(self, tens, shared):
        self.tens, self.shared = tens, shared


def _fp32_table(module, tens):
    """fp32 copies of a low-precision module's table tensors in ONE persistent flat buffer, refreshed with a single multi-tensor copy
    (121 separate `.float()` launches would cost more than the forward's convolutions)."""
    cache = module.__dict__.get('_fp32_cache')
    key = tuple((t.data_ptr(), t.numel(), t.dtype) for t in tens)
    if cache is None or cache[0] != key:
        sizes = [(t.numel() + 63) // 64 * 64 for t in tens]        # 256-byte aligned slots
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=tens[0].device)
        views, off = [], 0
        for t, n in zip(tens, sizes):
            views.append(flat[off:off + t.numel()].view(t.shape))
            off += n
        cache = (key, flat, views, [])
        module.__dict__['_fp32_cache'] = cache
    # copy on write: a forward whose backward is still pending reads this cache in its backward -- it gets its own copy of the values BEFORE they are
    # overwritten (an optimizer step or an SWA swap between two pending graphs would otherwise be read silently where torch raises a version error);
    # the usual step (forward, backward, forward, ...) never pays for the copy
    pending = cache[3]
    while pending:
        holder = pending.pop()()
        if holder is not None and holder.shared:
            own_flat = cache[1].clone()
            off, own = 0, []
            for t in cache[2]:
                own.append(own_flat[off:off + t.numel()].view(t.shape)); off += (t.numel() + 63) // 64 * 64
            holder.tens, holder.shared = own, False
    torch._foreach_copy_(cache[2], list(tens))
    return cache[2]


def _native_forward(module, plan, x, tens, softmax, want16, x_needs_grad, training, momenta, frozen=False, loss=None, roi=None):
    """One call of e3_unet_forward / e3_unet_forward_bf16 / e3_unet_forward_f16.  `tens`: fp32 table tensors (contiguous); `want16`: None, or
    the 16-bit type to compute in (torch.bfloat16 / torch.float16).  Returns (y fp32, saved buffer or None, the input as handed to the
    library, the 16-bit type of the native path taken or None)."""
    lib = _lib.load()
    dev = x.device
    N, Cin, D, H, W = x.shape
    b16 = want16 if (want16 and not x_needs_grad and not frozen and plan.bf16_supported() and not _NO_BF16) else None
    xin = x.detach().to(b16 if b16 is not None else torch.float32).contiguous()
    saved_bytes, scratch_bytes = plan.sizes(N, D, H, W, training, bf16=b16)
    saved = _alloc_saved(dev, max(saved_bytes, 256)) if training else None
    token = _frozen_token(module, plan, (N, D, H, W), tens) if (not training and b16 is None and loss is None) else None
    if token is not None:
        scratch, reuse, pkey = _get_scratch(dev, max(scratch_bytes, 256), token)
    else:
        scratch, reuse, pkey = _get_scratch(dev, max(scratch_bytes, 256)), False, None
    Do, Ho, Wo = plan.out_dims(D, H, W)      # == (D, H, W) unless conv_mode='valid'
    y = torch.empty((N, plan.out_channels, Do, Ho, Wo), dtype=torch.float32, device=dev)
    ptrs = (c_void_p * len(tens))(*[t.data_ptr() for t in tens])
    cmom = (ctypes.c_float * len(momenta))(*momenta) if training else None
    flags = (E3_FWD_TRAINING if training else 0) | (E3_FWD_SOFTMAX if softmax else 0) | (E3_FWD_FROZEN_BN if frozen else 0) | (E3_FWD_REUSE_PACKED if reuse else 0)
    fwd = lib.e3_unet_forward_f16 if b16 is torch.float16 else (lib.e3_unet_forward_bf16 if b16 is not None else lib.e3_unet_forward)
    with torch.cuda.device(dev):
        args = (plan.handle, _lib.stream_ptr(dev), c_void_p(xin.data_ptr()), N, D, H, W, ptrs, cmom,
                c_void_p(y.data_ptr()), c_void_p(saved.data_ptr()) if saved is not None else None,
                c_size_t(saved.numel() if saved is not None else 0),
                c_void_p(scratch.data_ptr()), c_size_t(scratch.numel()), flags)
        if loss is not None and b16 is None:      # criterion inside the head (fp32 path): loss = dict(target, weight, ce, dice, eps, smooth) -> + ws, out
            nbytes = lib.e3_ce_dice_workspace_bytes(plan.out_channels)
            loss['ws'] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            loss['out'] = torch.empty((), dtype=torch.float32, device=dev)
            w = loss['weight']
            reduce_sums = loss.get('reduce_sums')      # sharded minibatch (CombinedCEDiceLoss(global_batch=True)): the sums travel between the ranks first
            sums = torch.empty(2 + 3 * plan.out_channels, dtype=torch.float64, device=dev) if reduce_sums is not None else None
            la = _lib.CEDiceArgs(loss['target'].data_ptr(), w.data_ptr() if w is not None else None, loss['ce'], loss['dice'], loss['eps'], loss['smooth'],
                                 loss['ws'].data_ptr(), nbytes, loss['out'].data_ptr(), sums.data_ptr() if sums is not None else None)
            check(lib.e3_unet_forward_loss(*args, ctypes.byref(la)))
            if sums is not None:
                sums = reduce_sums(sums)
                check(lib.e3_ce_dice_from_sums(_lib.stream_ptr(dev), ptr(sums), w.data_ptr() if w is not None else None, plan.out_channels,
                                               loss['ce'], loss['dice'], loss['eps'], loss['smooth'], ptr(loss['ws']), c_size_t(nbytes), ptr(loss['out'])))
            loss['in_head'] = True
        elif roi is not None and not training:      # only the voxels of `roi` are wanted (UNet.forward_roi)
            fwd_roi = lib.e3_unet_forward_roi_f16 if b16 is torch.float16 else (lib.e3_unet_forward_roi_bf16 if b16 is not None else lib.e3_unet_forward_roi)
            check(fwd_roi(plan.handle, _lib.stream_ptr(dev), c_void_p(xin.data_ptr()), N, D, H, W, ptrs, c_void_p(y.data_ptr()),
                                          c_void_p(scratch.data_ptr()), c_size_t(scratch.numel()), flags, (ctypes.c_int * 6)(*roi)))
        else:
            check(fwd(*args))
    if pkey is not None:
        _packed_ok(pkey, token)
    return y, saved, xin, b16


def _native_backward(plan, dy, xin, tens, saved, b16, want_dx, sync=None, frozen=False, loss=None):
    """One call of e3_unet_backward / _bf16: (flat fp32 gradient buffer, its per-table-entry views, dx or None).
    loss = (logits, CEDiceArgs, gout or None): e3_unet_backward_loss -- `dy` is then ignored (the head's kernels form dLoss/dlogits)."""
    lib = _lib.load()
    N, _, D, H, W = xin.shape
    dev = xin.device
    dy32 = dy.detach().to(torch.float32).contiguous() if loss is None else None
    # one flat gradient buffer; every trainable tensor gets a view (table order)
    flat, views = (sync.flat_views(plan, tens) if sync is not None else _flat_views(plan, tens, dev))
    gptrs = (c_void_p * len(tens))(*[(v.data_ptr() if v is not None else None) for v in views])
    ptrs = (c_void_p * len(tens))(*[t.data_ptr() for t in tens])
    dx = torch.empty_like(xin) if want_dx else None
    _, scratch_bytes = plan.sizes(N, D, H, W, True, bf16=b16)
    scratch = _get_scratch(dev, max(scratch_bytes, 256))
    ev, ev_blk = (sync.bucket_event(), sync.bucket_after_down_block) if sync is not None else (None, 0)
    tail = (c_void_p(xin.data_ptr()), N, D, H, W, ptrs, gptrs, c_void_p(dx.data_ptr()) if dx is not None else None,
            c_void_p(saved.data_ptr()), c_size_t(saved.numel()), c_void_p(scratch.data_ptr()), c_size_t(scratch.numel()), ev, ev_blk)
    args = (plan.handle, _lib.stream_ptr(dev), c_void_p(dy32.data_ptr()) if dy32 is not None else None) + tail
    with torch.cuda.device(dev):
        if loss is not None:
            y, la, gout = loss
            if b16 is not None:
                fn = lib.e3_unet_backward_loss_f16 if b16 is torch.float16 else lib.e3_unet_backward_loss_bf16
                check(fn(plan.handle, _lib.stream_ptr(dev), c_void_p(y.data_ptr()), ctypes.byref(la), c_void_p(gout.data_ptr()) if gout is not None else None, *tail))
                if sync is not None:
                    sync.after_backward(plan)
                return flat, views, dx
            reserve = getattr(sync, 'cu_reserve', 0) if (sync is not None and ev is not None) else 0
            check(lib.e3_unet_backward_loss(plan.handle, _lib.stream_ptr(dev), c_void_p(y.data_ptr()), ctypes.byref(la),
                                            c_void_p(gout.data_ptr()) if gout is not None else None, *tail,
                                            (E3_BWD_FROZEN_BN if frozen else 0) | E3_BWD_CU_RESERVE(reserve)))
        elif b16 is torch.float16:
            check(lib.e3_unet_backward_f16(*args))
        elif b16:
            check(lib.e3_unet_backward_bf16(*args))
        else:
            reserve = getattr(sync, 'cu_reserve', 0) if (sync is not None and ev is not None) else 0
            check(lib.e3_unet_backward2(*args, (E3_BWD_FROZEN_BN if frozen else 0) | E3_BWD_CU_RESERVE(reserve)))
    if sync is not None:
        sync.after_backward(plan)
    return flat, views, dx


class _UNetLossFunction(torch.autograd.Function):
    """UNet.forward_with_loss: (logits, loss) = one native forward whose 1x1x1 head also evaluates the criterion (e3_unet_forward_loss);
    the backward seeds e3_unet_backward with dLoss/dlogits from e3_ce_dice_bwd (+ an incoming logits gradient, if the logits are used too)."""

    @staticmethod
    def forward(ctx, module, crit, want16, x, target, *params):
        ctx.set_materialize_grads(False)
        req = dict(crit, target=target.contiguous())
        # (the request travels in `mode` like the needed region does: nothing is parked on the module, so a concurrent or re-entrant plain
        # forward of the same module cannot pick it up)
        y = _UNetFunction.forward(ctx, module, (2, None, req), want16, x, *params)
        lib = _lib.load()
        N, C = y.shape[:2]
        sp = [1] * (5 - y.dim()) + list(y.shape[2:])
        ctx.ce_dims = (C, N, *sp)
        if 'out' not in req:            # (the head did not take the criterion along -- a 16-bit native path: the separate pass over the logits)
            nbytes = lib.e3_ce_dice_workspace_bytes(C)
            req['ws'] = torch.empty(nbytes, dtype=torch.uint8, device=y.device)
            req['out'] = torch.empty((), dtype=torch.float32, device=y.device)
            y32 = y.float().contiguous()
            w = req['weight']
            if req.get('reduce_sums') is not None:      # sharded minibatch: local sums -> sum over ranks -> loss and coefficients from the totals
                sums = torch.empty(2 + 3 * C, dtype=torch.float64, device=y.device)
                check(lib.e3_ce_dice_sums(_lib.stream_ptr(y.device), ptr(y32), ptr(req['target']), ptr(w) if w is not None else None, C, N, *sp,
                                          ptr(req['ws']), c_size_t(nbytes), ptr(sums)))
                sums = req['reduce_sums'](sums)
                check(lib.e3_ce_dice_from_sums(_lib.stream_ptr(y.device), ptr(sums), ptr(w) if w is not None else None, C,
                                               req['ce'], req['dice'], req['eps'], req['smooth'], ptr(req['ws']), c_size_t(nbytes), ptr(req['out'])))
            else:
                check(lib.e3_ce_dice_fwd(_lib.stream_ptr(y.device), ptr(y32), ptr(req['target']), ptr(w) if w is not None else None, C, N, *sp,
                                         req['ce'], req['dice'], req['eps'], req['smooth'], ptr(req['ws']), c_size_t(nbytes), ptr(req['out'])))
        ctx.ce = (req['target'], req['weight'], req['ws'])
        # the criterion went through the head (fp32 path, 2..4 classes, a norm in front of the head): its backward can stay there too
        ctx.ce_in_head = bool(req.get('in_head')) and 2 <= C <= 4 and module.normalization == 'batch' and not _NO_LOSS_BWD
        ctx.ce_b16_ok = 2 <= C <= 4 and not _NO_LOSS_BWD       # (native 16-bit paths: the criterion ran as its own pass over the fp32 logits; its backward still lives in the head's kernels)
        ctx.ce_w = (req['ce'], req['dice'], req['eps'], req['smooth'])
        ctx.ce_scale = float(req.get('grad_scale', 1.0))      # (sharded minibatch with averaged gradients: x world size)
        ctx.save_for_backward(y)
        return y, req['out']

    @staticmethod
    def backward(ctx, dy, dloss):
        y, = ctx.saved_tensors
        target, w, ws = ctx.ce
        C, N, D, H, W = ctx.ce_dims
        dl = None
        if dloss is not None and dy is None and (ctx.ce_in_head or (ctx.b16 is not None and ctx.ce_b16_ok)):
            # e3_unet_backward_loss: dLoss/dlogits is formed inside the head's backward kernels (no dlogits tensor, no e3_ce_dice_bwd pass)
            g = dloss.to(device=y.device, dtype=torch.float32).contiguous()
            if ctx.ce_scale != 1.0:
                g = g * ctx.ce_scale
            la = _lib.CEDiceArgs(target.data_ptr(), w.data_ptr() if w is not None else None, *ctx.ce_w, ws.data_ptr(), ws.numel(), None, None)
            ctx.loss_bwd = (y if y.dtype == torch.float32 else y.float().contiguous(), la, g)
            grads = _UNetFunction.backward(ctx, None)
            return grads[:4] + (None,) + grads[4:]
        if dloss is not None:
            y32 = y.float().contiguous()
            dl = torch.empty_like(y32)
            g = dloss.to(device=y.device, dtype=torch.float32).contiguous() * ctx.ce_scale
            check(_lib.load().e3_ce_dice_bwd(_lib.stream_ptr(y.device), ptr(y32), ptr(target), ptr(w) if w is not None else None, C, N, D, H, W,
                                             ptr(ws), c_size_t(ws.numel()), ptr(g), ptr(dl)))
        if dy is not None:
            dl = dy.float() if dl is None else dl + dy.float()
        if dl is None:
            dl = torch.zeros_like(y, dtype=torch.float32)
        grads = _UNetFunction.backward(ctx, dl)
        return grads[:4] + (None,) + grads[4:]


def _store_attention_maps(module, plan, x, training, saved):
    """up_convs[i].att = the block's attention map (N, 1, D, H, W) of this forward, as the reference's UpConvBlock keeps it for later
    analysis (unet.py:382,394-395; plotted by the Trainer, trainer.py:611-617).  Copied out of the forward's workspaces on its stream."""
    lib = _lib.load()
    N, _, D, H, W = x.shape
    dev = x.device
    scratch = _get_scratch(dev, 256)
    with torch.cuda.device(dev):
        for i, blk in enumerate(module.up_convs):
            do, ho, wo = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
            args = (plan.handle, _lib.stream_ptr(dev), N, D, H, W, int(training), c_void_p(saved.data_ptr()) if saved is not None else None,
                    c_void_p(scratch.data_ptr()), i)
            check(lib.e3_unet_attention_map(*args, None, ctypes.byref(do), ctypes.byref(ho), ctypes.byref(wo)))
            att = torch.empty((N, 1, do.value, ho.value, wo.value), dtype=torch.float32, device=dev)
            check(lib.e3_unet_attention_map(*args, c_void_p(att.data_ptr()), None, None, None))
            blk.att = att.squeeze(2) if module.dim == 2 else att


class _UNetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, mode, want16, x, *params):
        plan = module._plan()
        in_dtype = x.dtype
        training = module.training or module._per_sample_norm()   # instance / group statistics also in eval mode
        # parameters / buffers at call time, in the plan's table order
        tens = module._table(plan, params)
        lowp = None
        if any(t.dtype != torch.float32 for t in tens):
            # model.half() / model.bfloat16() (Predictor(float16=True), BASELINE cfg 3's bf16 storage): the parameter table handed to the
            # library is fp32 (up-cast copies; results are cast back: the output below, gradients in backward, the running statistics
            # right after the call).  bf16 modules on a covered configuration COMPUTE in bf16 (native kernels, csrc/unet_bf16.cpp).
            lowp = tens
            tens = _fp32_table(module, tens)
        tens = [t if t.is_contiguous() else t.contiguous() for t in tens]
        # a module stored entirely in one 16-bit type (model.to(torch.bfloat16) / model.half()) computes natively in that type
        all16 = None
        if lowp is not None:
            kinds = {t.dtype for t in lowp if t.is_floating_point()}
            if len(kinds) == 1 and next(iter(kinds)) in (torch.bfloat16, torch.float16):
                all16 = next(iter(kinds))
        # (grad mode is off inside Function.forward; needs_input_grad tells whether a backward can follow)
        # (needs_input_grad mirrors requires_grad of the inputs whatever the grad mode: `mode` carries torch.is_grad_enabled() of the caller,
        # so that validation under torch.no_grad() takes the inference path instead of saving activations nobody will use)
        roi, loss_req = None, None
        if isinstance(mode, tuple):          # (mode bits, needed region[, criterion request]): UNet.forward_roi / forward_with_loss
            mode, roi, loss_req = (tuple(mode) + (None,))[:3]
        softmax, grad_on = bool(mode & 1), bool(mode & 2)
        need_grad = grad_on and any(ctx.needs_input_grad) and not softmax
        # a module in eval mode that a backward will follow (frozen-BatchNorm fine-tuning, training/recalibration.py:53-73 style uses):
        # the training data flow with the running statistics as constants
        # (attention=True: the gates' own nn.BatchNorm layers are there whatever `normalization` says)
        frozen = need_grad and not training and (module.normalization == 'batch' or bool(getattr(module, 'attention', False)))
        if need_grad and not training and not frozen:
            training = True            # (normalization='none': train and eval mode are the same function; take the flow that saves)
        # nn.RReLU in train mode: one seed per native call from torch's default generator (torch.manual_seed makes runs repeatable); the
        # backward re-arms the same seed, the kernels recompute the slopes
        rr = module._rrelu_interval() if (module.training and not frozen) else None
        ctx.rrelu = None
        if rr is not None:
            ctx.rrelu = (rr[0], rr[1], int(torch.randint(1, 2 ** 31 - 1, (1,)).item()))
        check(_lib.load().e3_unet_set_rrelu(plan.handle, *(ctx.rrelu if ctx.rrelu is not None else (0.0, 0.0, 0))))
        y, saved, xin, b16 = _native_forward(module, plan, x, tens, softmax, want16 if want16 else (all16 if in_dtype == all16 else None),
                                             ctx.needs_input_grad[3], training or frozen,
                                             module._momenta(plan) if (training or frozen) else None, frozen=frozen,
                                             loss=loss_req if (training and not softmax) else None,
                                             roi=roi if not (training or frozen) else None)
        if getattr(module, 'attention', False) and b16 is None:
            _store_attention_maps(module, plan, x, training or frozen, saved)
        ctx.frozen = frozen
        # (autocast / compute_dtype: the result has that dtype whichever kernels computed it, as under torch.autocast with the reference --
        # uncovered configurations, a requested input gradient or E3_NO_BF16 compute in fp32 and round once at the end)
        out_dtype = want16 if want16 else in_dtype
        if training and not frozen and module.training:
            module._bump_num_batches_tracked(plan)
            if lowp is not None:         # running statistics were updated in the fp32 copies
                pairs = [(lo_t, hi_t) for kind, lo_t, hi_t in zip(plan.kinds, lowp, tens) if kind != 0 and lo_t.dtype != torch.float32]
                if pairs:
                    torch._foreach_copy_([a for a, _ in pairs], [b for _, b in pairs])
        ctx.module, ctx.plan, ctx.b16 = module, plan, b16
        ctx.softmax, ctx.eval_mode, ctx.in_dtype = softmax, not (training or frozen), in_dtype
        # (a low-precision module's fp32 table is a cache that the next forward refreshes with the same parameter values:
        # the backward only reads weights and affine parameters from it, never the running statistics)
        ctx.x32, ctx.saved_buf, ctx.tens = (xin, saved, _TableRef(tens, False)) if need_grad else (None, None, None)
        if need_grad and lowp is not None:
            # the fp32 table of a low-precision module is ONE persistent cache that the next forward overwrites in place: this forward registers its
            # table reference, and _fp32_table hands it a private copy if the cache is refreshed while the backward is still pending (copy on write)
            cache = module.__dict__.get('_fp32_cache')
            if cache is not None and len(tens) == len(cache[2]) and all(a is b for a, b in zip(tens, cache[2])):
                ctx.tens.shared = True
                cache[3].append(weakref.ref(ctx.tens))
        return y if out_dtype == torch.float32 else y.to(out_dtype)

    @staticmethod
    def backward(ctx, dy):
        if ctx.saved_buf is None:
            if ctx.eval_mode:
                raise NotImplementedError('backward through an eval-mode (running-statistics) forward is not implemented on the HIP path')
            raise RuntimeError('backward called but the forward did not save activations')
        if ctx.softmax:
            raise NotImplementedError('backward through the fused softmax head is not implemented')
        module, plan = ctx.module, ctx.plan
        check(_lib.load().e3_unet_set_rrelu(plan.handle, *(ctx.rrelu if ctx.rrelu is not None else (0.0, 0.0, 0))))
        flat, views, dx = _native_backward(plan, dy, ctx.x32, ctx.tens.tens, ctx.saved_buf, ctx.b16, ctx.needs_input_grad[3],
                                           getattr(module, '_grad_sync', None), frozen=ctx.frozen, loss=getattr(ctx, 'loss_bwd', None))
        ctx.loss_bwd = None
        ctx.saved_buf = ctx.x32 = ctx.tens = None   # free the activations now
        named = list(module._named_table_params(plan))
        lowp_dtype = next((p.dtype for _, p in named if p.dtype != torch.float32), None)
        if lowp_dtype is not None and all(p.dtype == lowp_dtype for _, p in named):
            # one cast of the whole flat buffer instead of one launch per parameter
            flat_lo = flat.to(lowp_dtype)
            off, lo_views = 0, []
            for v in views:
                lo_views.append(flat_lo[off:off + v.numel()] if v is not None else None)
                off += v.numel() if v is not None else 0
            views = lo_views
        by_name = dict(zip(plan.names, views))
        grads = []
        for name, p in named:
            g = by_name[name].view_as(p)
            grads.append(g if g.dtype == p.dtype else g.to(p.dtype))
        if dx is not None and dx.dtype != ctx.in_dtype:
            dx = dx.to(ctx.in_dtype)
        return (None, None, None, dx, *grads)


def _flat_views(plan, tens, device):
    sizes = [t.numel() if k == 0 else 0 for t, k in zip(tens, plan.kinds)]
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    views, off = [], 0
    for n in sizes:
        views.append(flat[off:off + n] if n else None)
        off += n
    return flat, views


# ---------------------------------------------------------------------------------------------------------------- TorchScript boundary
# `torch.jit.script(model)` is what the reference's training script does by default (examples/train_unet_neurodata.py:110-113, --jit
# onsave; Trainer._save_model scripts and saves again, training/trainer.py:876-881).  A scripted UNet.forward gathers the module's
# tensors in table order and calls ONE registered operator; the operator (and its autograd formula) run the same native entry points as
# the eager path.  Functional by construction: the updated running statistics are RETURNED and copied back by the scripted code.
def _plan_key_from_floats(key: List[float]):
    ints = (0, 1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 13, 14, 15, 16)
    return tuple(int(round(v)) if i in ints else float(v) for i, v in enumerate(key))


@torch.library.custom_op('e3unet::unet_fwd', mutates_args=())
def _op_unet_fwd(x: torch.Tensor, tensors: List[torch.Tensor], key: List[float], momenta: List[float], training: bool,
                 softmax: bool) -> List[torch.Tensor]:
    """-> [logits, saved activations (empty unless training), updated running statistics in table order]"""
    if not x.is_cuda:
        raise RuntimeError('elektronn3_amd.UNet runs only on a ROCm GPU (hand-written HIP kernels); there is no CPU fallback')
    if x.dtype != torch.float32 or any(t.dtype != torch.float32 for t in tensors):
        raise NotImplementedError('the scripted UNet supports fp32 modules (cast the eager module for reduced precision)')
    plan = _get_plan(_plan_key_from_floats(key))
    tens = [t.detach().contiguous() for t in tensors]
    new_bufs = []
    if training:
        for i, kind in enumerate(plan.kinds):
            if kind != 0:
                tens[i] = tens[i].clone()
                new_bufs.append(tens[i])
    check(_lib.load().e3_unet_set_rrelu(plan.handle, 0.0, 0.0, 0))     # (scripted modules: RReLU with its eval-mode slope; see _script_ok)
    y, saved, _, _ = _native_forward(None, plan, x, tens, softmax, False, False, training, list(momenta) if training else None)
    return [y, saved if saved is not None else x.new_empty(0, dtype=torch.uint8)] + new_bufs


@_op_unet_fwd.register_fake
def _(x, tensors, key, momenta, training, softmax):
    plan = _get_plan(_plan_key_from_floats(key))
    N, _, D, H, W = x.shape
    Do, Ho, Wo = plan.out_dims(D, H, W)
    saved_bytes = plan.sizes(N, D, H, W, True)[0] if training else 0
    bufs = [torch.empty_like(t) for t, k in zip(tensors, plan.kinds) if k != 0] if training else []
    return [x.new_empty((N, plan.out_channels, Do, Ho, Wo)), x.new_empty(max(saved_bytes, 256) if training else 0, dtype=torch.uint8)] + bufs


@torch.library.custom_op('e3unet::unet_bwd', mutates_args=())
def _op_unet_bwd(dy: torch.Tensor, x: torch.Tensor, tensors: List[torch.Tensor], saved: torch.Tensor, key: List[float]) -> List[torch.Tensor]:
    """-> gradients of the trainable table entries, in table order"""
    plan = _get_plan(_plan_key_from_floats(key))
    tens = [t.detach().contiguous() for t in tensors]
    check(_lib.load().e3_unet_set_rrelu(plan.handle, 0.0, 0.0, 0))
    _, views, _ = _native_backward(plan, dy, x.detach().contiguous(), tens, saved, False, False)
    return [v.view_as(t).clone() for v, t in zip(views, tens) if v is not None]     # (operator outputs must not alias each other)


@_op_unet_bwd.register_fake
def _(dy, x, tensors, saved, key):
    plan = _get_plan(_plan_key_from_floats(key))
    return [torch.empty_like(t) for t, k in zip(tensors, plan.kinds) if k == 0]


def _unet_fwd_setup(ctx, inputs, output):
    x, tensors, key, momenta, training, softmax = inputs
    if x.requires_grad:       # (silently returning no input gradient would be wrong; the eager module supports it)
        raise NotImplementedError('scripted elektronn3_amd.UNet: a gradient w.r.t. the input is not available through the TorchScript operator '
                                  '(use the eager module)')
    ctx.key, ctx.training, ctx.softmax, ctx.n_mom = key, training, softmax, len(momenta)
    # (the running statistics are overwritten right after the call and never read by the backward: placeholders keep the table's shape)
    kinds = _get_plan(_plan_key_from_floats(key)).kinds
    ctx.save_for_backward(x, output[1], *[t if k == 0 else torch.empty_like(t) for t, k in zip(tensors, kinds)])


def _unet_fwd_backward(ctx, grads):
    if not ctx.training:
        raise NotImplementedError('backward through an eval-mode (running-statistics) forward is not implemented on the HIP path')
    if ctx.softmax:
        raise NotImplementedError('backward through the fused softmax head is not implemented')
    x, saved, *tensors = ctx.saved_tensors
    plan = _get_plan(_plan_key_from_floats(ctx.key))
    g = iter(torch.ops.e3unet.unet_bwd(grads[0], x, tensors, saved, ctx.key))
    # (one entry per operator argument; an EMPTY list argument -- no BatchNorm momenta -- flattens to zero leaves, not to one)
    return None, [next(g) if k == 0 else None for k in plan.kinds], None, (None if ctx.n_mom else []), None, None


_op_unet_fwd.register_autograd(_unet_fwd_backward, setup_context=_unet_fwd_setup)


# layer types per dimensionality: conv, transposed conv, max-pool, batch norm (unet.py:47-105)
_LAYERS = {3: (nn.Conv3d, nn.ConvTranspose3d, nn.MaxPool3d, nn.BatchNorm3d),
           2: (nn.Conv2d, nn.ConvTranspose2d, nn.MaxPool2d, nn.BatchNorm2d)}


def _activation_slope(activation):
    """Negative-side slope of the activations on the HIP path (get_activation, unet.py:183-199): 'relu' 0, 'leaky' 0.1
    (nn.LeakyReLU(negative_slope=0.1)), 'lin' 1 (nn.Identity); module instances of those types are accepted like the reference does
    (it deep-copies them).  None = not on the HIP path."""
    if isinstance(activation, str):
        return {'relu': 0.0, 'leaky': 0.1, 'lin': 1.0, 'silu': 2.0, 'prelu': 3.0,     # (2.0 = ACT_SILU, 3.0 = ACT_PRELU in csrc/common.h)
                'rrelu': (1.0 / 8 + 1.0 / 3) / 2}.get(activation)                     # nn.RReLU() in eval mode: slope (lower + upper) / 2
    if isinstance(activation, nn.RReLU):
        return (float(activation.lower) + float(activation.upper)) / 2 if 0.0 <= activation.lower <= activation.upper <= 1.0 else None
    if isinstance(activation, nn.LeakyReLU):
        return float(activation.negative_slope) if 0.0 <= activation.negative_slope <= 1.0 else None
    if isinstance(activation, nn.ReLU):
        return 0.0
    if isinstance(activation, nn.Identity):
        return 1.0
    if isinstance(activation, nn.SiLU):
        return 2.0
    if isinstance(activation, nn.PReLU) and activation.num_parameters == 1:
        return 3.0
    return None


def _make_activation(activation):
    if isinstance(activation, str):
        return {'relu': nn.ReLU, 'leaky': lambda: nn.LeakyReLU(negative_slope=0.1), 'lin': nn.Identity, 'silu': nn.SiLU,
                'prelu': lambda: nn.PReLU(num_parameters=1), 'rrelu': nn.RReLU}[activation]()
    import copy
    return copy.deepcopy(activation)


def _norm_factory(normalization, BatchNorm, dim, channels):
    """get_normalization (unet.py:77-105) for the modes on the HIP path."""
    if normalization == 'batch':
        return lambda: BatchNorm(channels)
    if normalization == 'instance':
        return lambda: (nn.InstanceNorm3d if dim == 3 else nn.InstanceNorm2d)(channels)   # affine=False, no running statistics
    if normalization.startswith('group'):
        return lambda: nn.GroupNorm(num_groups=_num_groups(normalization), num_channels=channels)
    return nn.Identity


def _num_groups(normalization):
    """'group' -> 8, 'group<G>' -> G (unet.py:81-90)."""
    if normalization == 'group':
        return 8
    if normalization[len('group'):].isdigit():
        return int(normalization[len('group'):])
    raise ValueError(f'normtype "{normalization}" not understood. It should be "group<G>", where <G> is the number of groups.')


class ResizeConv(nn.Module):
    """Parameter container of the reference's ResizeConv (unet.py:411-449): ``nn.Upsample(scale_factor, mode='nearest')`` followed by a
    3x3x3 (planar: 1x3x3; dim=2: 3x3) convolution -- ``up_mode='resizeconv_nearest'``.  state_dict keys: ``conv.weight``, ``conv.bias``."""

    def __init__(self, in_channels, out_channels, planar=False, dim=3, upsampling_mode='nearest', kernel_size=3):
        super().__init__()
        Conv = _LAYERS[dim][0]
        k, p = ((1, 3, 3), (0, 1, 1)) if (planar and dim == 3) else (3, 1)
        if kernel_size == 1:        # conv1(), unet.py:178-180,443-444
            k, p = 1, 0
        self.upsampling_mode = upsampling_mode          # 'nearest', or 'trilinear' / 'bilinear' (upconv2, unet.py:166-170)
        self.scale_factor = (1, 2, 2) if (planar and dim == 3) else 2
        self.dim = dim
        self.upsample = nn.Upsample(scale_factor=self.scale_factor, mode=self.upsampling_mode)
        self.conv = Conv(in_channels, out_channels, kernel_size=k, padding=p)

    def forward(self, x):
        raise RuntimeError('elektronn3_amd sub-modules only hold parameters; call UNet.forward')


class GridAttention(nn.Module):
    """Parameter container of the reference's GridAttention (unet.py:452-541; Attention U-Net, arXiv:1804.03999): ``theta`` (k = s = 2 conv
    of the skip, no bias), ``phi`` (1x1x1 conv of the gating signal), ``psi`` (1x1x1 conv to one channel) and the output transform
    ``w = Sequential(conv 1x1x1, BatchNorm)``.  Same state_dict keys and the same initialisation (unet.py:532-541)."""

    def __init__(self, in_channels, gating_channels, inter_channels=None, dim=3, sub_sample_factor=2):
        super().__init__()
        assert dim in [2, 3]
        if sub_sample_factor not in (2, (2,) * dim, [2] * dim):
            raise NotImplementedError('GridAttention on the HIP path: sub_sample_factor=2 (what UpConvBlock passes, unet.py:377-379)')
        self.dim = dim
        self.sub_sample_factor = (2,) * dim
        self.sub_sample_kernel_size = self.sub_sample_factor
        self.in_channels, self.gating_channels = in_channels, gating_channels
        self.inter_channels = inter_channels if inter_channels is not None else max(in_channels // 2, 1)
        if self.inter_channels != in_channels // 2:
            raise NotImplementedError('GridAttention on the HIP path: inter_channels = in_channels // 2 (the default, unet.py:475-478)')
        Conv, Norm = _LAYERS[dim][0], _LAYERS[dim][3]
        self.upsample_mode = 'trilinear' if dim == 3 else 'bilinear'
        self.w = nn.Sequential(Conv(in_channels, in_channels, kernel_size=1), Norm(in_channels))
        self.theta = Conv(in_channels, self.inter_channels, kernel_size=self.sub_sample_kernel_size, stride=self.sub_sample_factor, bias=False)
        self.phi = Conv(gating_channels, self.inter_channels, kernel_size=1, stride=1, padding=0, bias=True)
        self.psi = Conv(self.inter_channels, 1, kernel_size=1, stride=1, bias=True)
        self.init_weights()

    def init_weights(self):
        def weight_init(m):
            name = m.__class__.__name__
            if name.find('Conv') != -1 or name.find('Linear') != -1:
                nn.init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif name.find('BatchNorm') != -1:
                nn.init.normal_(m.weight.data, 1.0, 0.02)
                nn.init.constant_(m.bias.data, 0.0)
        self.apply(weight_init)

    def forward(self, x, g):
        raise RuntimeError('elektronn3_amd sub-modules only hold parameters; call UNet.forward')


class DummyAttention(nn.Module):
    """attention=False (unet.py:544-546): the skip passes through."""

    def forward(self, x, g):
        return x, None


class DownConv(nn.Module):
    """Parameter container for one encoder block (two convs, two norms, max-pool) -- reference: unet.py:202-253."""

    def __init__(self, in_channels, out_channels, pooling=True, planar=False, dim=3, normalization='batch', full_norm=True, activation='relu',
                 conv_mode='same'):
        super().__init__()
        self.in_channels, self.out_channels, self.pooling, self.planar = in_channels, out_channels, pooling, planar
        self.dim = dim
        self.normalization = normalization
        Conv, Pool, Norm = _LAYERS[dim][0], _LAYERS[dim][2], _LAYERS[dim][3]
        k, p = ((1, 3, 3), (0, 1, 1)) if (planar and dim == 3) else (3, 1)
        if 'same' not in conv_mode:      # padding = 1 if 'same' in conv_mode else 0 (unet.py:217)
            p = (0, 0, 0) if (planar and dim == 3) else 0
        self.conv1 = Conv(in_channels, out_channels, kernel_size=k, padding=p)
        self.conv2 = Conv(out_channels, out_channels, kernel_size=k, padding=p)
        if pooling:
            self.pool = Pool(kernel_size=(1, 2, 2) if (planar and dim == 3) else 2, ceil_mode=True)
        else:
            self.pool = nn.Identity()
        self.act1, self.act2 = _make_activation(activation), _make_activation(activation)
        norm = _norm_factory(normalization, Norm, dim, out_channels)                          # get_normalization, unet.py:77-105
        self.norm0 = norm() if full_norm else nn.Identity()                                   # unet.py:238-242
        self.norm1 = norm()

    def forward(self, x):
        raise RuntimeError('elektronn3_amd sub-modules only hold parameters; call UNet.forward')


class UpConv(nn.Module):
    """Parameter container for one decoder block (transposed conv, two convs, three norms) -- reference: unet.py:328-408."""

    def __init__(self, in_channels, out_channels, planar=False, dim=3, normalization='batch', full_norm=True, merge_mode='concat', activation='relu',
                 up_mode='transpose', conv_mode='same', attention=False):
        super().__init__()
        self.in_channels, self.out_channels, self.planar = in_channels, out_channels, planar
        self.merge_mode = merge_mode
        self.dim = dim
        self.normalization = normalization
        Conv, ConvT, Norm = _LAYERS[dim][0], _LAYERS[dim][1], _LAYERS[dim][3]
        ks = (1, 2, 2) if (planar and dim == 3) else 2
        k, p = ((1, 3, 3), (0, 1, 1)) if (planar and dim == 3) else (3, 1)
        if 'same' not in conv_mode:      # (unet.py:347; the ResizeConv's own conv keeps padding 1)
            p = (0, 0, 0) if (planar and dim == 3) else 0
        self.up_mode = up_mode
        if up_mode == 'transpose':
            self.upconv = ConvT(in_channels, out_channels, kernel_size=ks, stride=ks)
        else:       # 'resizeconv_nearest' / 'resizeconv_linear' (upconv2, unet.py:152-176)
            mode = 'nearest' if 'nearest' in up_mode else ('trilinear' if dim == 3 else 'bilinear')
            self.upconv = ResizeConv(in_channels, out_channels, planar=planar, dim=dim, upsampling_mode=mode,
                                     kernel_size=1 if up_mode.endswith('1') else 3)
        self.conv1 = Conv((2 if merge_mode == 'concat' else 1) * out_channels, out_channels, kernel_size=k, padding=p)   # unet.py:352-360
        self.conv2 = Conv(out_channels, out_channels, kernel_size=k, padding=p)
        self.act0, self.act1, self.act2 = (_make_activation(activation) for _ in range(3))
        norm = _norm_factory(normalization, Norm, dim, out_channels)
        self.norm0 = norm() if full_norm else nn.Identity()                                   # unet.py:369-375
        self.norm1 = norm() if full_norm else nn.Identity()
        self.norm2 = norm()
        # unet.py:376-382: the gate sees the skip (in_channels // 2 channels) and the block's input as the gating signal
        self.attention = GridAttention(in_channels=in_channels // 2, gating_channels=in_channels, dim=dim) if attention else DummyAttention()
        self.att = None   # Trainer reads model.up_convs[i].att (trainer.py:611-617); None without attention

    def forward(self, enc, dec):
        raise RuntimeError('elektronn3_amd sub-modules only hold parameters; call UNet.forward')


class UNet(nn.Module):
    """3D U-Net with the reference's interface (elektronn3/models/unet.py:755-771), executed by hand-written HIP
    kernels.  Options of the reference that are not yet on the HIP path raise ``NotImplementedError`` at
    construction: ``up_mode='upsample'`` (the reference's ``upconv2`` has no branch for it either),
    ``conv_mode`` other than ``'same'`` / ``'valid'``.  ``attention=True`` puts a ``GridAttention`` gate on every decoder block (fp32 kernels;
    not with per-sample norms); ``up_convs[i].att`` holds the attention map of the last forward.  ``activation='rrelu'``: eval mode uses the fixed slope (1/8 + 1/3)/2 of
    ``nn.RReLU``; a train-mode forward draws a slope per negative element from U(1/8, 1/3) inside the kernels (a hash of a per-call seed taken
    from torch's generator, the unit and the element index; the backward recomputes it).  ``dim=2`` (Conv2d/BatchNorm2d/... parameters, 4D input) runs on the planar kernels: a 2D U-Net is
    the 3D one with every block planar and a depth of 1."""

    _script_key: List[float]

    def __init__(
            self,
            in_channels: int = 1,
            out_channels: int = 2,
            n_blocks: int = 3,
            start_filts: int = 32,
            up_mode: str = 'transpose',
            merge_mode: str = 'concat',
            planar_blocks: Sequence = (),
            batch_norm: str = 'unset',
            attention: bool = False,
            activation: Union[str, nn.Module] = 'relu',
            normalization: str = 'batch',
            full_norm: bool = True,
            dim: int = 3,
            conv_mode: str = 'same',
    ):
        super().__init__()
        self._setup(in_channels, out_channels, n_blocks, start_filts, up_mode, merge_mode, planar_blocks, batch_norm, attention, activation,
                    normalization, full_norm, dim, conv_mode)

    def _setup(self, in_channels, out_channels, n_blocks, start_filts, up_mode, merge_mode, planar_blocks, batch_norm, attention, activation,
               normalization, full_norm, dim, conv_mode, res_blocks=None):
        """Argument checks and module tree; res_blocks = (enc_res_blocks, dec_res_blocks) builds the ResUNet's blocks (elektronn3_amd/resunet.py)."""
        # -- the reference's argument validation (unet.py:774-824), same exception types
        if n_blocks < 1:
            raise ValueError('n_blocks must be > 1.')
        if dim not in {2, 3}:
            raise ValueError('dim has to be 2 or 3')
        if dim == 2 and planar_blocks != ():
            raise ValueError('If dim=2, you can\'t use planar_blocks since everything will be planar (2-dimensional) anyways.\n'
                             'Either set dim=3 or set planar_blocks=().')
        if up_mode not in ('transpose', 'upsample', 'resizeconv_nearest', 'resizeconv_linear', 'resizeconv_nearest1', 'resizeconv_linear1'):
            raise ValueError(f'"{up_mode}" is not a valid mode for upsampling')
        if merge_mode not in ('concat', 'add'):
            raise ValueError(f'"{merge_mode}" is not a valid mode for merging up and down paths. Only "concat" and "add" are allowed.')
        if 'resizeconv' in up_mode and merge_mode == 'add':
            raise ValueError('up_mode "resizeconv" is incompatible with merge_mode "add" at the moment')
        if len(planar_blocks) > n_blocks:
            raise ValueError('planar_blocks can\'t be longer than n_blocks.')
        if planar_blocks and (max(planar_blocks) >= n_blocks or min(planar_blocks) < 0):
            raise ValueError('planar_blocks has invalid value range. All values have to be block indices, meaning integers '
                             'between 0 and (n_blocks - 1).')
        if batch_norm != 'unset':
            raise RuntimeError('The `batch_norm` option has been replaced with the more general `normalization` option.\n'
                               'If you still want to use batch normalization, set `normalization=batch` instead.')
        # -- what the HIP path implements this round
        unsupported = []
        if up_mode == 'upsample': unsupported.append("up_mode='upsample' (upconv2 has no branch for it in the reference either)")
        if isinstance(activation, str) and activation not in ('relu', 'leaky', 'prelu', 'rrelu', 'silu', 'lin'):
            raise ValueError(f'unknown activation {activation!r}')
        if _activation_slope(activation) is None: unsupported.append(f'activation={activation!r}')
        if normalization is None:
            normalization = 'none'
        if not (normalization in ('none', 'batch', 'instance') or (isinstance(normalization, str) and normalization.startswith('group'))):
            raise ValueError(f'Unknown normalization type "{normalization}".\nValid choices are "batch", "instance", "group" or "group<G>",'
                             'where <G> is the number of groups.')      # get_normalization, unet.py:106-111
        if normalization.startswith('group'):
            if start_filts % _num_groups(normalization) != 0:
                raise ValueError('num_channels must be divisible by num_groups')      # (torch.nn.GroupNorm's own check)
        elif normalization not in ('batch', 'none', 'instance'): unsupported.append(f'normalization={normalization!r}')
        if conv_mode not in ('same', 'valid'): unsupported.append(f'conv_mode={conv_mode!r}')
        if attention and (normalization == 'instance' or normalization.startswith('group')):
            # (the gate's own nn.BatchNorm needs the whole batch; these norms run one native call per sample)
            unsupported.append(f'attention=True with normalization={normalization!r}')
        if start_filts % 8 != 0: unsupported.append(f'start_filts={start_filts} (must be a multiple of 8)')
        if not (1 <= out_channels <= 16): unsupported.append(f'out_channels={out_channels} (1..16)')
        if not (in_channels < 8 or in_channels % 8 == 0): unsupported.append(f'in_channels={in_channels}')
        if res_blocks is not None:
            if dim != 3: unsupported.append('ResUNet with dim=2 (the reference builds its ConvBlocks with Conv3d whatever dim says, resunet.py:289-299)')
            if conv_mode != 'same' and (res_blocks[0] or res_blocks[1]):
                unsupported.append("residual blocks with conv_mode='valid' (the shortcut and the conv output differ in size)")
            if not all(isinstance(r, int) and 0 <= r <= 8 for r in res_blocks): unsupported.append(f'res_blocks={res_blocks} (0..8)')
        if unsupported:
            raise NotImplementedError('not implemented on the MI355X HIP path yet: ' + ', '.join(unsupported))

        self.out_channels = out_channels
        self.in_channels = in_channels
        self.start_filts = start_filts
        self.n_blocks = n_blocks
        self.normalization = normalization
        self.full_norm = full_norm
        self.attention = attention
        self.conv_mode = conv_mode
        self.activation = activation
        self.dim = dim
        self.up_mode = up_mode
        self.merge_mode = merge_mode
        self.planar_blocks = tuple(planar_blocks)

        self.down_convs = nn.ModuleList()
        self.up_convs = nn.ModuleList()
        outs = in_channels
        if res_blocks is not None:
            self.enc_res_blocks, self.dec_res_blocks = res_blocks
            outs = self._build_blocks()
        for i in range(n_blocks if res_blocks is None else 0):
            ins = in_channels if i == 0 else outs
            outs = start_filts * (2 ** i)
            self.down_convs.append(DownConv(ins, outs, pooling=i < n_blocks - 1, planar=i in self.planar_blocks, dim=dim,
                                            normalization=normalization, full_norm=full_norm, activation=activation, conv_mode=conv_mode))
        for i in range(n_blocks - 1 if res_blocks is None else 0):
            ins = outs
            outs = ins // 2
            self.up_convs.append(UpConv(ins, outs, planar=(n_blocks - 2 - i) in self.planar_blocks, dim=dim,
                                        normalization=normalization, full_norm=full_norm, merge_mode=merge_mode, activation=activation, up_mode=up_mode, conv_mode=conv_mode,
                                        attention=attention))
        self.conv_final = _LAYERS[dim][0](outs, out_channels, kernel_size=1)
        self.apply(self.weight_init)
        self._script_key = [float(v) for v in self._plan_key()]      # (read by the scripted forward)
        self._script_ok = (normalization != 'instance' and dim == 3 and self._rrelu_interval() is None and not attention)     # (train-mode RReLU needs a per-call seed)

    @staticmethod
    def weight_init(m):
        """Xavier-normal weights, zero biases for every (transposed) conv -- same scheme as unet.py:885-892."""
        if isinstance(m, GridAttention):      # (as the reference: only the container is skipped, `apply` still reaches its convs)
            return
        if isinstance(m, (nn.Conv3d, nn.ConvTranspose3d, nn.Conv2d, nn.ConvTranspose2d)):   # get_conv/get_convtranspose of the model's dim
            nn.init.xavier_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def __getstate__(self):
        # torch.save(model) / copy.deepcopy(model) (trainer.py:874, inference.py:402-407): never pickle runtime
        # attachments (streams, events, process groups of a GradSync)
        state = self.__dict__.copy()
        state.pop('_grad_sync', None)
        state.pop('_inst_consts', None)
        state.pop('_fp32_cache', None)
        return state

    # ------------------------------------------------------------------ native plumbing
    def _plan_key(self):
        mask = 0
        for b in (range(self.n_blocks) if self.dim == 2 else self.planar_blocks):   # dim=2: every block is planar, depth 1
            mask |= 1 << int(b)
        eps = next((float(m.eps) for m in self.modules() if isinstance(m, (nn.modules.batchnorm._BatchNorm, nn.GroupNorm))), 1e-5)
        return (self.in_channels, self.out_channels, self.n_blocks, self.start_filts, mask, 2 if self.normalization.startswith('group') else (1 if self.normalization in ('batch', 'instance') else 0), eps,
                1 if getattr(self, 'full_norm', True) else 0, 1 if self.merge_mode == 'add' else 0,
                _num_groups(self.normalization) if self.normalization.startswith('group') else 0,
                {'transpose': 0, 'resizeconv_nearest': 1, 'resizeconv_linear': 2, 'resizeconv_nearest1': 3, 'resizeconv_linear1': 4}[self.up_mode],
                1 if self.conv_mode == 'valid' else 0, float(_activation_slope(self.activation)),
                (2 if self.dim == 2 else 1) if getattr(self, 'attention', False) else 0) + self._variant_key()

    def _variant_key(self):
        """(resunet, enc_res_blocks, dec_res_blocks) of e3_unet_cfg."""
        return (0, 0, 0)

    def _plan(self):
        return _get_plan(self._plan_key())

    def _per_sample_norm(self):
        return self.normalization == 'instance' or self.normalization.startswith('group')

    def _named_table_params(self, plan):
        """(name, Parameter) for the trainable entries of the plan table, in table order."""
        inst = self.normalization == 'instance'
        for name, kind in zip(plan.names, plan.kinds):
            if kind == 0 and not (inst and '.norm' in name):    # InstanceNorm has no affine parameters (gamma = 1, beta = 0 constants)
                yield name, self.get_parameter(name)

    def _table(self, plan, params):
        """Tensors for every table slot: trainable ones come from the autograd inputs (same order), buffers from
        the module (running statistics are read -- and updated -- in place at call time)."""
        it = iter(params)
        out = []
        inst = self.normalization == 'instance'
        for name, kind in zip(plan.names, plan.kinds):
            if inst and '.norm' in name:
                # InstanceNorm = BatchNorm over a batch of one sample with gamma = 1, beta = 0 and throw-away running statistics
                dev = params[0].device
                C = self.get_submodule(name.rsplit('.', 1)[0]).num_features
                key = (name.rsplit('.', 1)[1], C, dev)
                consts = self.__dict__.setdefault('_inst_consts', {})
                if key not in consts:
                    fill = 1.0 if key[0] in ('weight', 'running_var') else 0.0
                    consts[key] = torch.full((C,), fill, dtype=torch.float32, device=dev)
                out.append(consts[key])
            else:
                out.append(next(it).detach() if kind == 0 else self.get_buffer(name))
        return out

    def _momenta(self, plan):
        moms = []
        if self._per_sample_norm():
            return [0.0] * plan.n_bn               # (no running statistics: nn.InstanceNorm3d defaults / nn.GroupNorm)
        for bn_name in plan.bn_names:
            bn = self.get_submodule(bn_name)
            if bn.momentum is None:   # cumulative moving average (torch semantics)
                moms.append(1.0 / float(int(bn.num_batches_tracked) + 1))
            else:
                moms.append(float(bn.momentum))
        return moms

    def _bump_num_batches_tracked(self, plan):
        if self._per_sample_norm():
            return
        nbt = [self.get_submodule(n).num_batches_tracked for n in plan.bn_names]
        if nbt:
            torch._foreach_add_(nbt, 1)

    # ------------------------------------------------------------------ public API
    def forward(self, x):
        if torch.jit.is_scripting():
            return self._scripted_forward(x)
        return self._run(x, softmax=False)

    def _scripted_forward(self, x: torch.Tensor) -> torch.Tensor:
        """UNet.forward as TorchScript sees it: the module's tensors in the native parameter-table order (per unit: conv weight, conv
        bias, [norm weight, norm bias, [running_mean, running_var]], [PReLU slope]; units in execution order; conv_final last), ONE call
        of the registered operator e3unet::unet_fwd, the returned running statistics copied back."""
        if not self._script_ok:
            raise RuntimeError('scripted elektronn3_amd.UNet: dim=3 with batch / group / no normalization, activations other than rrelu')
        t: List[torch.Tensor] = []
        bufs: List[torch.Tensor] = []
        counters: List[torch.Tensor] = []
        mom: List[float] = []
        for blk in self.down_convs:
            t.append(blk.conv1.weight)
            b1 = blk.conv1.bias
            assert b1 is not None
            t.append(b1)
            if hasattr(blk.norm0, 'weight'):
                t.append(blk.norm0.weight); t.append(blk.norm0.bias)
                if hasattr(blk.norm0, 'running_mean'):
                    rm, rv, nb, mo = blk.norm0.running_mean, blk.norm0.running_var, blk.norm0.num_batches_tracked, blk.norm0.momentum
                    assert rm is not None and rv is not None and nb is not None and mo is not None
                    t.append(rm); t.append(rv); bufs.append(rm); bufs.append(rv); counters.append(nb); mom.append(mo)
            if hasattr(blk.act1, 'weight'):
                t.append(blk.act1.weight)
            t.append(blk.conv2.weight)
            b2 = blk.conv2.bias
            assert b2 is not None
            t.append(b2)
            if hasattr(blk.norm1, 'weight'):
                t.append(blk.norm1.weight); t.append(blk.norm1.bias)
                if hasattr(blk.norm1, 'running_mean'):
                    rm, rv, nb, mo = blk.norm1.running_mean, blk.norm1.running_var, blk.norm1.num_batches_tracked, blk.norm1.momentum
                    assert rm is not None and rv is not None and nb is not None and mo is not None
                    t.append(rm); t.append(rv); bufs.append(rm); bufs.append(rv); counters.append(nb); mom.append(mo)
            if hasattr(blk.act2, 'weight'):
                t.append(blk.act2.weight)
        for ub in self.up_convs:
            if hasattr(ub.upconv, 'conv'):
                t.append(ub.upconv.conv.weight)
                b0 = ub.upconv.conv.bias
            else:
                t.append(ub.upconv.weight)
                b0 = ub.upconv.bias
            assert b0 is not None
            t.append(b0)
            if hasattr(ub.norm0, 'weight'):
                t.append(ub.norm0.weight); t.append(ub.norm0.bias)
                if hasattr(ub.norm0, 'running_mean'):
                    rm, rv, nb, mo = ub.norm0.running_mean, ub.norm0.running_var, ub.norm0.num_batches_tracked, ub.norm0.momentum
                    assert rm is not None and rv is not None and nb is not None and mo is not None
                    t.append(rm); t.append(rv); bufs.append(rm); bufs.append(rv); counters.append(nb); mom.append(mo)
            if hasattr(ub.act0, 'weight'):
                t.append(ub.act0.weight)
            t.append(ub.conv1.weight)
            b1 = ub.conv1.bias
            assert b1 is not None
            t.append(b1)
            if hasattr(ub.norm1, 'weight'):
                t.append(ub.norm1.weight); t.append(ub.norm1.bias)
                if hasattr(ub.norm1, 'running_mean'):
                    rm, rv, nb, mo = ub.norm1.running_mean, ub.norm1.running_var, ub.norm1.num_batches_tracked, ub.norm1.momentum
                    assert rm is not None and rv is not None and nb is not None and mo is not None
                    t.append(rm); t.append(rv); bufs.append(rm); bufs.append(rv); counters.append(nb); mom.append(mo)
            if hasattr(ub.act1, 'weight'):
                t.append(ub.act1.weight)
            t.append(ub.conv2.weight)
            b2 = ub.conv2.bias
            assert b2 is not None
            t.append(b2)
            if hasattr(ub.norm2, 'weight'):
                t.append(ub.norm2.weight); t.append(ub.norm2.bias)
                if hasattr(ub.norm2, 'running_mean'):
                    rm, rv, nb, mo = ub.norm2.running_mean, ub.norm2.running_var, ub.norm2.num_batches_tracked, ub.norm2.momentum
                    assert rm is not None and rv is not None and nb is not None and mo is not None
                    t.append(rm); t.append(rv); bufs.append(rm); bufs.append(rv); counters.append(nb); mom.append(mo)
            if hasattr(ub.act2, 'weight'):
                t.append(ub.act2.weight)
        t.append(self.conv_final.weight)
        bf = self.conv_final.bias
        assert bf is not None
        t.append(bf)
        # nn.GroupNorm keeps no running statistics: its statistics are per sample in training and eval mode alike
        if self._script_key[5] == 2.0:
            # nn.GroupNorm: statistics per sample in training and eval mode alike, no running statistics: one call per sample
            ys: List[torch.Tensor] = []
            for n in range(x.shape[0]):
                ys.append(torch.ops.e3unet.unet_fwd(x[n:n + 1], t, self._script_key, mom, True, False)[0])
            return torch.cat(ys, 0)
        outs = torch.ops.e3unet.unet_fwd(x, t, self._script_key, mom, self.training, False)
        if self.training:
            for i in range(len(bufs)):
                bufs[i].copy_(outs[2 + i])
            for c in counters:
                c.add_(1)
        return outs[0]

    @torch.jit.unused
    def _rrelu_interval(self):
        """(lower, upper) of nn.RReLU if that is the network's activation (train mode draws a slope per element), else None."""
        a = self.activation
        if a == 'rrelu':
            return (1.0 / 8, 1.0 / 3)
        if isinstance(a, nn.RReLU):
            return (float(a.lower), float(a.upper))
        return None

    def _run(self, x, softmax=False, roi=None):
        if self.dim == 2:
            if not isinstance(x, torch.Tensor) or x.dim() != 4:
                raise ValueError('expected a 4D (N, C, H, W) tensor')
            x = x.unsqueeze(2)
        if not isinstance(x, torch.Tensor) or x.dim() != 5:
            raise ValueError('expected a 5D (N, C, D, H, W) tensor')
        if x.shape[1] != self.in_channels:
            raise ValueError(f'expected {self.in_channels} input channels, got {x.shape[1]}')
        if not x.is_cuda:
            raise RuntimeError('elektronn3_amd.UNet runs only on a ROCm GPU (hand-written HIP kernels); there is no CPU fallback')
        plan = self._plan()
        params = [p for _, p in self._named_table_params(plan)]
        # torch.autocast('cuda', dtype=torch.bfloat16) around the call (the bf16 counterpart of Trainer(mixed_precision=True),
        # trainer.py:519), or module.compute_dtype = torch.bfloat16: bf16 compute with the module's own (fp32 master) parameters
        # ... and the reference's own default, float16 autocast (torch.cuda.amp.autocast(), trainer.py:519): float16 compute (native f16 kernels)
        want16 = None
        if torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in (torch.bfloat16, torch.float16):
            want16 = torch.get_autocast_dtype('cuda')
        elif getattr(self, 'compute_dtype', None) in (torch.bfloat16, torch.float16):
            want16 = self.compute_dtype
        if any(p.device != x.device for p in params):
            raise RuntimeError('input and parameters are on different devices')
        mode = (1 if softmax else 0) | (2 if torch.is_grad_enabled() else 0)
        if roi is not None:
            mode = (mode, roi)
        if self._per_sample_norm():
            # per-sample statistics in training AND eval mode (nn.InstanceNorm3d defaults, nn.GroupNorm): one native call per sample;
            # autograd sums the parameter gradients of the calls
            y = torch.cat([_UNetFunction.apply(self, mode, want16, x[n:n + 1], *params) for n in range(x.shape[0])], 0)
        else:
            y = _UNetFunction.apply(self, mode, want16, x, *params)
        return y.squeeze(2) if self.dim == 2 else y

    @torch.jit.unused
    def forward_with_loss(self, x, target, criterion):
        """``(out, loss)`` with ``out = self(x)`` and ``loss = criterion(out, target)`` -- the two lines of the reference's training step
        (training/trainer.py:520-524) as ONE call, so that the criterion of the training example (``CombinedCEDiceLoss``, i.e.
        ``CombinedLoss([CrossEntropyLoss(w), DiceLoss(apply_softmax=True, w)])``, modules/loss.py:19-49,158-234) is evaluated by the kernel of
        the final 1x1x1 conv while it holds the logits in registers ("loss on device fused with conv_final").  Same values and
        gradients as the two separate calls.  Any other criterion, module configuration or a data-parallel ``global_batch`` criterion (whose sums
        travel between ranks first) takes exactly those two calls."""
        from .loss import CombinedCEDiceLoss
        fusable = (type(criterion) is CombinedCEDiceLoss and self.training and torch.is_grad_enabled() and self.dim == 3 and not self._per_sample_norm()
                   and isinstance(x, torch.Tensor) and x.is_cuda and x.dim() == 5 and target.dtype == torch.int64 and target.is_cuda
                   )
        if fusable:
            # the head kernel reads target[v] for every output voxel: the target must be exactly the logits' grid on the input's device (a target
            # on another grid -- the input size with conv_mode='valid', an (N, 1, D, H, W) tensor -- takes the separate calls, which raise
            # like the reference's CrossEntropyLoss does)
            out_sp = self._plan().out_dims(*(int(v) for v in x.shape[2:]))
            fusable = (x.shape[1] == self.in_channels and target.device == x.device
                       and tuple(target.shape) == (int(x.shape[0]), *out_sp))
        if not fusable:
            out = self(x)
            return out, criterion(out, target)
        plan = self._plan()
        params = [p for _, p in self._named_table_params(plan)]
        want16 = None
        if torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') in (torch.bfloat16, torch.float16):
            want16 = torch.get_autocast_dtype('cuda')
        elif getattr(self, 'compute_dtype', None) in (torch.bfloat16, torch.float16):
            want16 = self.compute_dtype
        w = criterion.weight
        crit = dict(weight=None if w is None else w.to(device=x.device, dtype=torch.float32).contiguous(), ce=criterion.ce_weight, dice=criterion.dice_weight,
                    eps=criterion.eps, smooth=criterion.smooth)
        if criterion.global_batch:       # a minibatch sharded over ranks: the criterion's 2 + 3C sums are exchanged between head and finaliser
            world = criterion._world()
            if world != 1:
                crit['reduce_sums'] = criterion._reduce_sums
                crit['grad_scale'] = float(abs(world)) if criterion.grads_averaged else 1.0
        return _UNetLossFunction.apply(self, crit, want16, x, target, *params)

    @torch.jit.unused
    def forward_softmax(self, x):
        """``softmax(forward(x), dim=1)`` with the softmax fused into the final 1x1x1 conv kernel (used by Predictor)."""
        return self._run(x, softmax=True)

    @torch.jit.unused
    def frozen_weights(self):
        """Context manager: the caller promises not to change parameters or running statistics inside the ``with`` block.  Inference forwards of one
        shape (``self(x)``, ``forward_roi``, ``forward_tile``) then pack / Winograd-transform / fold the weights ONCE and later calls take them as
        they lie in the scratch buffer (E3_FWD_REUSE_PACKED) -- the tile loop of ``inference.Predictor`` packed 22 MB of weights 726 times.  A call of
        another shape, of another module on the same stream, or any training call simply packs again; nesting is allowed."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            outer = self.__dict__.get('_frozen_scope')
            self.__dict__['_frozen_scope'] = outer if outer is not None else next(_scope_ids)
            try:
                yield self
            finally:
                if outer is None:
                    self.__dict__.pop('_frozen_scope', None)
        return scope()

    @torch.jit.unused
    def forward_roi(self, x, roi, softmax=False):
        """Inference forward of which only the output voxels ``roi = ((d0, d1), (h0, h1), (w0, w1))`` will be used (the tile loop of
        ``inference.Predictor`` keeps the central crop of every tile, inference.py:496-525): the result has the full shape and equals
        ``self(x)`` (``forward_softmax(x)``) INSIDE the region; outside it the values are unspecified.  The decoder's 3x3x3 convs then skip
        the bricks the region does not depend on (e3_unet_forward_roi).  Training mode, enabled grad or 2D modules: the plain forward."""
        if self.training or torch.is_grad_enabled() or self.dim != 3 or self._per_sample_norm():
            return self._run(x, softmax=softmax)
        (d0, d1), (h0, h1), (w0, w1) = roi
        return self._run(x, softmax=softmax, roi=(int(d0), int(h0), int(w0), int(d1), int(h1), int(w1)))

    @torch.jit.unused
    def forward_tile(self, vol, in_lo, tile_shape, out, out_lo, roi, softmax=False):
        """One tile of a tiled inference run WITHOUT the tile copy in front of the model and the crop copy behind it (the tile loop of
        ``inference.tiled_apply``, inference.py:153-199): the tile ``vol[:, :, in_lo : in_lo + tile_shape]`` of the padded device volume is read in
        place (zero padding at the tile's faces, exactly like a contiguous copy of it), and the output voxels ``roi = ((d0, d1), (h0, h1), (w0, w1))``
        of the tile go to ``out[:, :, out_lo : out_lo + (d1 - d0, ...)]`` (e3_unet_forward_tile).  fp32 module / tensors, one input channel,
        'same' convolutions, eval mode; anything else raises NotImplementedError (callers copy the tile and use forward_roi)."""
        if (self.training or torch.is_grad_enabled() or self.dim != 3 or self._per_sample_norm() or self.in_channels != 1 or self.conv_mode != 'same'
                or getattr(self, 'attention', False) or vol.dtype != torch.float32 or out.dtype != torch.float32 or not vol.is_cuda
                or vol.stride(-1) != 1 or out.stride(-1) != 1 or vol.dim() != 5 or out.dim() != 5):
            raise NotImplementedError('forward_tile: configuration outside the in-place tile path')
        plan = self._plan()
        params = [p for _, p in self._named_table_params(plan)]
        tens = self._table(plan, params)
        if any(t.dtype != torch.float32 for t in tens):
            raise NotImplementedError('forward_tile: fp32 modules only')
        tens = [t if t.is_contiguous() else t.contiguous() for t in tens]
        lib = _lib.load()
        dev = vol.device
        N = int(vol.shape[0])
        D, H, W = (int(v) for v in tile_shape)
        (d0, d1), (h0, h1), (w0, w1) = roi
        # the library sees raw pointers and strides: everything it will touch is checked here
        if out.device != dev or any(t.device != dev for t in tens):
            raise ValueError('forward_tile: volume, output and parameters must be on one device')
        if int(vol.shape[1]) != 1 or int(out.shape[0]) != N or int(out.shape[1]) != self.out_channels:
            raise ValueError('forward_tile: vol must be (N, 1, ...), out (N, out_channels, ...)')
        for ax, (lo, n, size) in enumerate(zip(in_lo, (D, H, W), vol.shape[2:])):
            if n <= 0 or int(lo) < 0 or int(lo) + n > int(size):
                raise ValueError(f'forward_tile: tile [{int(lo)}, {int(lo) + n}) exceeds the volume (axis {ax}, size {int(size)})')
        for ax, ((r0, r1), n, lo, size) in enumerate(zip(roi, (D, H, W), out_lo, out.shape[2:])):
            if not (0 <= int(r0) < int(r1) <= n):
                raise ValueError(f'forward_tile: region [{int(r0)}, {int(r1)}) is not inside the tile (axis {ax}, size {n})')
            if int(lo) < 0 or int(lo) + int(r1) - int(r0) > int(size):
                raise ValueError(f'forward_tile: region written at {int(lo)} exceeds the output (axis {ax}, size {int(size)})')
        _, scratch_bytes = plan.sizes(N, D, H, W, False, bf16=None)
        token = _frozen_token(self, plan, (N, D, H, W), tens)
        if token is not None:
            scratch, reuse, pkey = _get_scratch(dev, max(scratch_bytes, 256), token)
        else:
            scratch, reuse, pkey = _get_scratch(dev, max(scratch_bytes, 256)), False, None
        view = _lib.TileView()
        view.x = vol.data_ptr() + 4 * (int(in_lo[0]) * vol.stride(2) + int(in_lo[1]) * vol.stride(3) + int(in_lo[2]))
        view.x_stride[:] = [vol.stride(0), vol.stride(2), vol.stride(3)]
        view.y = out.data_ptr() + 4 * (int(out_lo[0]) * out.stride(2) + int(out_lo[1]) * out.stride(3) + int(out_lo[2]))
        view.y_stride[:] = [out.stride(0), out.stride(1), out.stride(2), out.stride(3)]
        ptrs = (c_void_p * len(tens))(*[t.data_ptr() for t in tens])
        with torch.cuda.device(dev):
            check(lib.e3_unet_set_rrelu(plan.handle, 0.0, 0.0, 0))
            check(lib.e3_unet_forward_tile(plan.handle, _lib.stream_ptr(dev), ctypes.byref(view), N, D, H, W, ptrs, c_void_p(scratch.data_ptr()),
                                           c_size_t(scratch.numel()), (E3_FWD_SOFTMAX if softmax else 0) | (E3_FWD_REUSE_PACKED if reuse else 0),
                                           (ctypes.c_int * 6)(int(d0), int(h0), int(w0), int(d1), int(h1), int(w1))))
        if pkey is not None:
            _packed_ok(pkey, token)

    @torch.jit.unused
    def forward_gradcp(self, x):
        """Same as :meth:`forward` (unet.py:918-935 trades recompute for memory; with 288 GB of HBM nothing needs to be
        recomputed, so checkpointing is a no-op here)."""
        return self.forward(x)

    # profiling hook used by bench.py (roofline of a single layer measured with HIP events inside the step)
    def conv_layers(self):
        return list(self._plan().conv_names)

    def profile_select(self, layer, which=0):
        check(_lib.load().e3_unet_profile_select(self._plan().handle, int(layer), int(which)))

    def profile_read(self):
        ms, n = ctypes.c_double(), ctypes.c_int()
        check(_lib.load().e3_unet_profile_read(self._plan().handle, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value
