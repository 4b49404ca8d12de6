"""Criterion of the reference's training example, on device in two kernels.

``CombinedCEDiceLoss(class_weights)`` computes what the reference assembles as
``CombinedLoss([CrossEntropyLoss(weight=w), DiceLoss(apply_softmax=True, weight=w)], weight=(0.5, 0.5))``
(elektronn3/modules/loss.py:19-49, 158-234; examples/train_unet_neurodata.py:294-296) -- same arguments
``(output, target)``, same scalar, same gradient -- with one pass over the logits for the forward and one for the
backward (libe3unet ``e3_ce_dice_fwd/bwd``) instead of ~20 ATen kernels.  There is no CPU fallback.
"""
import torch

from . import _lib
from ._lib import c_size_t, check, ptr, stream_ptr


class _CEDice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, ce_w, dice_w, eps, smooth):
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
        check(L.e3_ce_dice_fwd(stream_ptr(logits.device), ptr(logits), ptr(target), ptr(w) if w is not None else None, C, N, D, H, W,
                               ce_w, dice_w, eps, smooth, ptr(ws), c_size_t(nbytes), ptr(out)))
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
        dl = torch.empty_like(logits)
        w = ctx.w
        check(L.e3_ce_dice_bwd(stream_ptr(logits.device), ptr(logits), ptr(target), ptr(w) if w is not None else None, C, N, D, H, W,
                               ptr(ws), c_size_t(nbytes), ptr(g), ptr(dl)))
        return dl, None, None, None, None, None, None


class CombinedCEDiceLoss(torch.nn.Module):
    """``ce_weight * CrossEntropyLoss(weight)(out, tgt) + dice_weight * DiceLoss(apply_softmax=True, weight, smooth)(out, tgt)``.

    Args mirror the reference modules: ``weight`` is the class-weight tensor both criteria receive in the example,
    ``smooth`` is DiceLoss's smoothing term, ``eps`` the constant of ``dice_loss`` (1e-4).
    """

    def __init__(self, weight=None, ce_weight=0.5, dice_weight=0.5, smooth=0.0, eps=1e-4):
        super().__init__()
        if weight is not None:
            self.register_buffer('weight', torch.as_tensor(weight, dtype=torch.float32))
        else:
            self.weight = None
        self.ce_weight, self.dice_weight, self.smooth, self.eps = float(ce_weight), float(dice_weight), float(smooth), float(eps)

    def forward(self, output, target):
        return _CEDice.apply(output, target, self.weight, self.ce_weight, self.dice_weight, self.eps, self.smooth)
