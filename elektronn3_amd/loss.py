"""Criterion of the reference's training example, on device in two kernels.

``CombinedCEDiceLoss(class_weights)`` computes what the reference assembles as
``CombinedLoss([CrossEntropyLoss(weight=w), DiceLoss(apply_softmax=True, weight=w)], weight=(0.5, 0.5))``
(elektronn3/modules/loss.py:19-49, 158-234; examples/train_unet_neurodata.py:294-296) -- same arguments
``(output, target)``, same scalar, same gradient -- with one pass over the logits for the forward and one for the
backward (libe3unet ``e3_ce_dice_fwd/bwd``) instead of ~20 ATen kernels.  There is no CPU fallback.

Data-parallel training (one process per GPU): the reference computes ONE loss over the minibatch that ``nn.DataParallel`` gathers on
GPU 0 (training/trainer.py:520-524), and neither the class-weighted CE (normalised by the batch's sum of target weights) nor Dice (a
ratio of batch-wide sums, modules/loss.py:181-186) is the mean of per-shard losses.  ``CombinedCEDiceLoss(..., global_batch=True)``
reproduces the reference exactly: every rank reduces its shard to 2 + 3C sums, ONE all-reduce of those <= 26 doubles makes them
batch-wide, and loss and gradient are taken from the totals (``e3_ce_dice_sums`` / ``e3_ce_dice_from_sums``).
"""
import torch

from . import _lib
from ._lib import c_size_t, check, ptr, stream_ptr


class _CEDice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, ce_w, dice_w, eps, smooth, reduce_sums=None, grad_scale=1.0):
        if not (logits.is_cuda and logits.dtype == torch.float32):
            raise ValueError('CombinedCEDiceLoss: logits must be a float32 CUDA tensor (the HIP path has no CPU fallback)')
        if target.dtype != torch.int64 or target.shape != (logits.shape[0],) + tuple(logits.shape[2:]):
            raise ValueError(f'CombinedCEDiceLoss: target must be int64 class indices of shape (N, *spatial), got {tuple(target.shape)} {target.dtype}')
        L = _lib.load()
        logits = logits.contiguous(); target = target.contiguous()
        N, C = logits.shape[:2]
        sp = list(logits.shape[2:])
        while len(sp) < 3:
            sp.insert(0, 1)
        D, H, W = sp
        nbytes = L.e3_ce_dice_workspace_bytes(C)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=logits.device)
        out = torch.empty((), dtype=torch.float32, device=logits.device)
        w = None if weight is None else weight.to(device=logits.device, dtype=torch.float32).contiguous()
        wp = ptr(w) if w is not None else None
        if reduce_sums is None:
            check(L.e3_ce_dice_fwd(stream_ptr(logits.device), ptr(logits), ptr(target), wp, C, N, D, H, W,
                                   ce_w, dice_w, eps, smooth, ptr(ws), c_size_t(nbytes), ptr(out)))
        else:        # sharded minibatch: local sums -> sum over ranks -> loss and backward coefficients from the totals
            sums = torch.empty(2 + 3 * C, dtype=torch.float64, device=logits.device)
            check(L.e3_ce_dice_sums(stream_ptr(logits.device), ptr(logits), ptr(target), wp, C, N, D, H, W, ptr(ws), c_size_t(nbytes), ptr(sums)))
            sums = reduce_sums(sums)
            check(L.e3_ce_dice_from_sums(stream_ptr(logits.device), ptr(sums), wp, C, ce_w, dice_w, eps, smooth, ptr(ws), c_size_t(nbytes), ptr(out)))
        ctx.grad_scale = float(grad_scale)
        ctx.save_for_backward(logits, target, ws)
        ctx.w = w
        ctx.dims = (C, N, D, H, W, nbytes)
        return out

    @staticmethod
    def backward(ctx, gout):
        logits, target, ws = ctx.saved_tensors
        C, N, D, H, W, nbytes = ctx.dims
        L = _lib.load()
        g = gout.to(device=logits.device, dtype=torch.float32).contiguous()
        if ctx.grad_scale != 1.0:
            g = g * ctx.grad_scale
        dl = torch.empty_like(logits)
        w = ctx.w
        check(L.e3_ce_dice_bwd(stream_ptr(logits.device), ptr(logits), ptr(target), ptr(w) if w is not None else None, C, N, D, H, W,
                               ptr(ws), c_size_t(nbytes), ptr(g), ptr(dl)))
        return dl, None, None, None, None, None, None, None, None


class CombinedCEDiceLoss(torch.nn.Module):
    """``ce_weight * CrossEntropyLoss(weight)(out, tgt) + dice_weight * DiceLoss(apply_softmax=True, weight, smooth)(out, tgt)``.

    Args mirror the reference modules: ``weight`` is the class-weight tensor both criteria receive in the example,
    ``smooth`` is DiceLoss's smoothing term, ``eps`` the constant of ``dice_loss`` (1e-4).

    ``global_batch=True`` (with an initialised ``torch.distributed`` group of more than one rank): ``output``/``target`` are this rank's
    shard of the minibatch; the value returned is the loss of the WHOLE minibatch (identical on all ranks) and the gradient is
    d(that loss)/d(local logits).  Summing the resulting parameter gradients over ranks gives the reference's gradient;
    ``grads_averaged=True`` (the default, matching ``GradSync(average=True)``) pre-multiplies by the world size so that the AVERAGE does.
    """

    def __init__(self, weight=None, ce_weight=0.5, dice_weight=0.5, smooth=0.0, eps=1e-4, global_batch=False, process_group=None,
                 grads_averaged=True):
        super().__init__()
        self.global_batch, self.process_group, self.grads_averaged = bool(global_batch), process_group, bool(grads_averaged)
        if weight is not None:
            self.register_buffer('weight', torch.as_tensor(weight, dtype=torch.float32))
        else:
            self.weight = None
        self.ce_weight, self.dice_weight, self.smooth, self.eps = float(ce_weight), float(dice_weight), float(smooth), float(eps)

    # -- the two hooks of the sharded mode (overridden by the single-process tests) ------------------------------------------------
    def _world(self):
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return 1
        world = dist.get_world_size(self.process_group)
        return world if world > 1 or not int(__import__('os').environ.get('E3_FORCE_GRADSYNC', '0')) else -1   # -1: forced path, world 1

    def _reduce_sums(self, sums):
        import torch.distributed as dist
        dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=self.process_group)
        return sums

    def forward(self, output, target):
        if output.dtype in (torch.bfloat16, torch.float16):
            # logits of a low-precision module (model.bfloat16(), autocast): the criterion's sums are fp32 either way (ATen's
            # cross_entropy accumulates bf16 inputs in fp32 too); autograd casts the logits gradient back
            output = output.float()
        reduce_sums, scale = None, 1.0
        if self.global_batch:
            world = self._world()
            if world != 1:
                reduce_sums, scale = self._reduce_sums, (float(abs(world)) if self.grads_averaged else 1.0)
        return _CEDice.apply(output, target, self.weight, self.ce_weight, self.dice_weight, self.eps, self.smooth, reduce_sums, scale)
