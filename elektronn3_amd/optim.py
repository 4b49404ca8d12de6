"""AdamW as ONE HIP launch over all parameter tensors.

Drop-in for the optimizer the reference example builds, ``optim.AdamW(model.parameters(), lr=1e-3, weight_decay=0.5e-4)``
(examples/train_unet_neurodata.py:257-262), including what its callers do with it:

* it IS a ``torch.optim.Optimizer`` (``param_groups`` with live ``lr`` -> ``CyclicLR``/``StepLR`` work; ``zero_grad``;
  the reference's ``SWA(optimizer)`` wrapper, training/swa.py:100-225, only touches ``param_groups``, ``state`` and ``step()``);
* ``state_dict()`` has torch.optim.AdamW's layout (per parameter ``step``, ``exp_avg``, ``exp_avg_sq``) so Trainer checkpoints
  (training/trainer.py:864-880) interchange with the stock optimizer;
* ``torch.amp.GradScaler.step(optimizer)`` (trainer.py:539-542): ``_step_supports_amp_scaling`` -- the scaler hands over
  ``grad_scale``/``found_inf`` device scalars; un-scaling and the skip-on-inf decision happen inside the kernel, no host sync.

The moments live in two flat buffers per group (tensor slices padded to the kernel's chunk size); the parameters keep
their own allocations.  There is no CPU path: CPU parameters raise.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


class AdamW(torch.optim.Optimizer):
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, amsgrad=False, *, maximize=False):
        if not 0.0 <= lr:
            raise ValueError(f'Invalid learning rate: {lr}')
        if not 0.0 <= eps:
            raise ValueError(f'Invalid epsilon value: {eps}')
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f'Invalid beta parameter at index 0: {betas[0]}')
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f'Invalid beta parameter at index 1: {betas[1]}')
        if not 0.0 <= weight_decay:
            raise ValueError(f'Invalid weight_decay value: {weight_decay}')
        if amsgrad or maximize:
            raise NotImplementedError('amsgrad / maximize are not implemented on the HIP path')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False))
        self._flat = {}   # group index -> dict(m, v, step, coef, numels, offsets, tables)

    # ------------------------------------------------------------------ flat state
    def _group_state(self, gi, group):
        fs = self._flat.get(gi)
        params = group['params']
        key = tuple((p.data_ptr(), p.numel()) for p in params)
        if fs is not None and fs['key'] == key:
            return fs
        if not params:
            raise ValueError('empty parameter group')
        dev = params[0].device
        for p in params:
            if p.device.type != 'cuda':
                raise RuntimeError('elektronn3_amd.optim.AdamW runs on the GPU only (there is no CPU path)')
            if p.dtype != torch.float32 or p.device != dev or not p.is_contiguous():
                raise NotImplementedError('AdamW on the HIP path needs contiguous fp32 parameters on one device')
        n = len(params)
        numels = (ctypes.c_int64 * n)(*[p.numel() for p in params])
        lib = _lib.load()
        total = lib.e3_adamw_state_floats(n, numels)
        offs = [lib.e3_adamw_state_offset(n, numels, i) for i in range(n)]
        m = torch.zeros(total, dtype=torch.float32, device=dev)
        v = torch.zeros(total, dtype=torch.float32, device=dev)
        step = torch.zeros(1, dtype=torch.float32, device=dev)
        # adopt per-parameter state that is already there (load_state_dict, or a parameter list that changed)
        for p, o in zip(params, offs):
            st = self.state.get(p)
            if st:
                m[o:o + p.numel()].copy_(st['exp_avg'].reshape(-1))
                v[o:o + p.numel()].copy_(st['exp_avg_sq'].reshape(-1))
                step.fill_(float(st['step']))
        for p, o in zip(params, offs):   # torch.optim.AdamW-shaped per-parameter state: views into the flat buffers
            self.state[p] = {'step': step[0], 'exp_avg': m[o:o + p.numel()].view_as(p), 'exp_avg_sq': v[o:o + p.numel()].view_as(p)}
        fs = dict(key=key, m=m, v=v, step=step, coef=torch.zeros(8, dtype=torch.float32, device=dev), numels=numels,
                  pp=(ctypes.c_void_p * n)(*[p.data_ptr() for p in params]), gp=(ctypes.c_void_p * n)(), n=n, dev=dev)
        self._flat[gi] = fs
        return fs

    def state_dict(self):
        sd = super().state_dict()
        # the live 'step' entries are views of ONE device counter per group; a stock torch.optim.AdamW that loads this dict
        # increments every parameter's step tensor on its own, so hand out independent CPU scalars (torch's default layout)
        for st in sd['state'].values():
            if 'step' in st:
                st['step'] = torch.tensor(float(st['step']), dtype=torch.float32)
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._flat = {}                  # rebuilt (adopting the loaded per-parameter tensors) at the next step

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale = getattr(self, 'grad_scale', None)
        found_inf = getattr(self, 'found_inf', None)
        lib = _lib.load()
        for gi, group in enumerate(self.param_groups):
            fs = self._group_state(gi, group)
            any_grad = False
            for i, p in enumerate(group['params']):
                g = p.grad
                if g is None:
                    fs['gp'][i] = None
                    continue
                if g.is_sparse:
                    raise RuntimeError('AdamW does not support sparse gradients')
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous(); p.grad = g
                fs['gp'][i] = g.data_ptr()
                any_grad = True
            if not any_grad:
                continue
            b1, b2 = group['betas']
            with torch.cuda.device(fs['dev']):
                check(lib.e3_adamw_step(stream_ptr(fs['dev']), fs['n'], fs['pp'], fs['gp'], fs['numels'], ptr(fs['m']), ptr(fs['v']),
                                        ptr(fs['step']), ptr(fs['coef']), float(group['lr']), float(b1), float(b2), float(group['eps']),
                                        float(group['weight_decay']),
                                        ptr(grad_scale.float()) if grad_scale is not None else None,
                                        ptr(found_inf.float()) if found_inf is not None else None))
        return loss


__all__ = ['AdamW']
