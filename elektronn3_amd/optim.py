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
            if p.dtype not in (torch.float32, torch.bfloat16) or p.dtype != params[0].dtype or p.device != dev or not p.is_contiguous():
                raise NotImplementedError('AdamW on the HIP path needs contiguous parameters of one dtype (fp32 or bf16) on one device')
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
                m[o:o + p.numel()].copy_(st['exp_avg'].reshape(-1).float())
                v[o:o + p.numel()].copy_(st['exp_avg_sq'].reshape(-1).float())
                step.fill_(float(st['step']))
        for p, o in zip(params, offs):   # torch.optim.AdamW-shaped per-parameter state: views into the flat buffers
            self.state[p] = {'step': step[0], 'exp_avg': m[o:o + p.numel()].view_as(p), 'exp_avg_sq': v[o:o + p.numel()].view_as(p)}
        fs = dict(key=key, m=m, v=v, step=step, coef=torch.zeros(8, dtype=torch.float32, device=dev), numels=numels, bf16=params[0].dtype == torch.bfloat16,
                  pp=(ctypes.c_void_p * n)(*[p.data_ptr() for p in params]), gp=(ctypes.c_void_p * n)(), n=n, dev=dev)
        self._flat[gi] = fs
        return fs

    def state_dict(self):
        sd = super().state_dict()
        # the live 'step' entries are views of ONE device counter per group; a stock torch.optim.AdamW that loads this dict
        # increments every parameter's step tensor on its own, so hand out independent CPU scalars (torch's default layout).
        # super().state_dict() returns the LIVE per-parameter dicts: edit shallow copies, never self.state itself (a second
        # state_dict() call must read the device counter again, not the scalar of the first one)
        sd['state'] = {k: dict(v) for k, v in sd['state'].items()}
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
                if g.dtype != p.dtype or not g.is_contiguous():
                    g = g.to(p.dtype).contiguous(); p.grad = g
                fs['gp'][i] = g.data_ptr()
                any_grad = True
            if not any_grad:
                continue
            b1, b2 = group['betas']
            with torch.cuda.device(fs['dev']):
                step_fn = lib.e3_adamw_step_bf16 if fs['bf16'] else lib.e3_adamw_step      # bf16 parameters: fp32 moments, one rounding per step
                check(step_fn(stream_ptr(fs['dev']), fs['n'], fs['pp'], fs['gp'], fs['numels'], ptr(fs['m']), ptr(fs['v']),
                                        ptr(fs['step']), ptr(fs['coef']), float(group['lr']), float(b1), float(b2), float(group['eps']),
                                        float(group['weight_decay']),
                                        ptr(grad_scale.float()) if grad_scale is not None else None,
                                        ptr(found_inf.float()) if found_inf is not None else None))
        return loss


__all__ = ['AdamW']


class SWA(torch.optim.Optimizer):
    """Stochastic weight averaging around any optimizer, with the interface of the reference's ``SWA`` wrapper
    (elektronn3/training/swa.py:11-345, vendored torchcontrib; used by ``Trainer`` as ``SWA(optimizer)`` +
    ``update_swa()`` / ``swap_swa_sgd()`` / ``bn_update``): automatic mode (``swa_start``, ``swa_freq``, optional ``swa_lr``) or manual
    mode (``update_swa`` / ``update_swa_group``), ``swap_swa_sgd``, ``state_dict`` with ``opt_state`` / ``swa_state`` / ``param_groups``.

    The running averages live in ``state[p]['swa_buffer']`` like the reference's, but one group is averaged (or swapped) by ONE HIP
    launch over all of its tensors (libe3unet ``e3_swa_update`` / ``e3_swa_swap``) instead of 3-4 ATen kernels per tensor; the
    arithmetic is the reference's (two rounded fp32 operations), so the averages are bit-identical.  GPU only."""

    def __init__(self, optimizer, swa_start=None, swa_freq=None, swa_lr=None):
        import warnings
        from collections import defaultdict
        given = [v is not None for v in (swa_start, swa_freq)]
        self._auto_mode = all(given)
        if any(given) and not all(given):
            warnings.warn('Some of swa_start, swa_freq is None, ignoring other')
        if self._auto_mode:
            if not isinstance(swa_start, int) or not isinstance(swa_freq, int):
                warnings.warn('Casting swa_start, swa_freq to int')
                swa_start, swa_freq = int(swa_start), int(swa_freq)
            if swa_start < 0:
                raise ValueError(f'Invalid swa_start: {swa_start}')
            if swa_freq < 1:
                raise ValueError(f'Invalid swa_freq: {swa_freq}')
        else:
            if swa_lr is not None:
                warnings.warn('Some of swa_start, swa_freq is None, ignoring swa_lr')
            swa_start = swa_freq = swa_lr = None
        if swa_lr is not None and swa_lr < 0:
            raise ValueError(f'Invalid SWA learning rate: {swa_lr}')
        self.swa_start, self.swa_freq, self.swa_lr = swa_start, swa_freq, swa_lr
        self.optimizer = optimizer
        self.defaults = optimizer.defaults
        self.param_groups = optimizer.param_groups       # shared: LR schedulers acting on either see the same groups
        self.state = defaultdict(dict)
        self.opt_state = optimizer.state
        for group in self.param_groups:
            group['n_avg'] = 0
            group['step_counter'] = 0

    # ------------------------------------------------------------------ one launch over a list of (parameter, buffer) pairs
    @staticmethod
    def _launch(pairs, n_avg=None):
        if not pairs:
            return
        dev = pairs[0][0].device
        for p, b in pairs:
            if p.device.type != 'cuda':
                raise RuntimeError('elektronn3_amd.optim.SWA runs on the GPU only (there is no CPU path)')
            if p.dtype != torch.float32 or p.device != dev or not p.data.is_contiguous() or not b.is_contiguous():
                raise NotImplementedError('SWA on the HIP path needs contiguous fp32 parameters on one device')
        n = len(pairs)
        pp = (ctypes.c_void_p * n)(*[p.data.data_ptr() for p, _ in pairs])
        bp = (ctypes.c_void_p * n)(*[b.data_ptr() for _, b in pairs])
        numels = (ctypes.c_int64 * n)(*[p.numel() for p, _ in pairs])
        lib = _lib.load()
        with torch.cuda.device(dev):
            if n_avg is None:
                check(lib.e3_swa_swap(stream_ptr(dev), n, pp, bp, numels))
            else:
                check(lib.e3_swa_update(stream_ptr(dev), n, pp, bp, numels, int(n_avg)))

    @torch.no_grad()
    def update_swa_group(self, group):
        """Folds the group's current parameters into their running averages (swa.py:145-176)."""
        pairs = []
        for p in group['params']:
            st = self.state[p]
            if 'swa_buffer' not in st:
                st['swa_buffer'] = torch.zeros_like(p.data)
            pairs.append((p, st['swa_buffer']))
        self._launch(pairs, group['n_avg'])
        group['n_avg'] += 1

    def update_swa(self):
        for group in self.param_groups:
            self.update_swa_group(group)

    @torch.no_grad()
    def swap_swa_sgd(self):
        """Exchanges the optimized variables and their running averages (swa.py:184-202); call it again to continue training."""
        import warnings
        for group in self.param_groups:
            pairs = []
            for p in group['params']:
                st = self.state[p]
                if 'swa_buffer' not in st:
                    warnings.warn(f"SWA wasn't applied to param {p}; skipping it")
                    continue
                pairs.append((p, st['swa_buffer']))
            self._launch(pairs)

    def step(self, closure=None):
        if self.swa_lr is not None:
            for group in self.param_groups:
                if group['step_counter'] >= self.swa_start:
                    group['lr'] = self.swa_lr
        loss = self.optimizer.step(closure)
        for group in self.param_groups:
            group['step_counter'] += 1
            if self._auto_mode and group['step_counter'] > self.swa_start and group['step_counter'] % self.swa_freq == 0:
                self.update_swa_group(group)
        return loss

    def zero_grad(self, set_to_none=True):
        self.optimizer.zero_grad(set_to_none=set_to_none)

    def add_param_group(self, param_group):
        param_group['n_avg'] = 0
        param_group['step_counter'] = 0
        self.optimizer.add_param_group(param_group)

    # ------------------------------------------------------------------ checkpoints (layout of swa.py:221-258)
    def state_dict(self):
        inner = self.optimizer.state_dict()
        order = [p for g in self.param_groups for p in g['params']]
        swa_state = {i: dict(self.state[p]) for i, p in enumerate(order) if p in self.state and self.state[p]}
        return {'opt_state': inner['state'], 'swa_state': swa_state, 'param_groups': inner['param_groups']}

    def load_state_dict(self, state_dict):
        self.optimizer.load_state_dict({'state': state_dict['opt_state'], 'param_groups': state_dict['param_groups']})
        self.param_groups = self.optimizer.param_groups
        self.opt_state = self.optimizer.state
        order = [p for g in self.param_groups for p in g['params']]
        self.state.clear()
        for i, st in state_dict['swa_state'].items():
            p = order[int(i)]
            self.state[p] = {k: (v.to(device=p.device, dtype=p.dtype).clone() if torch.is_tensor(v) else v) for k, v in st.items()}

    # ------------------------------------------------------------------ BatchNorm statistics of the averaged weights
    @staticmethod
    def bn_update(loader, model, device=None):
        """One pass over ``loader`` that re-estimates the running statistics of every ``_BatchNorm`` module as the cumulative average over
        the batches (swa.py:262-306): statistics reset, momentum of batch k = b_k / (n_seen + b_k), train-mode forwards, momenta and
        the training flag restored.  Batches may be tensors, (tensor, ...) sequences or the Trainer's dicts (key ``'inp'``)."""
        bns = [m for m in model.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        if not bns:
            return
        was_training = model.training
        model.train()
        saved = [m.momentum for m in bns]
        for m in bns:
            m.running_mean = torch.zeros_like(m.running_mean)
            m.running_var = torch.ones_like(m.running_var)
        seen = 0
        with torch.no_grad():
            for batch in loader:
                if isinstance(batch, (list, tuple)):
                    batch = batch[0]
                elif isinstance(batch, dict):
                    batch = batch['inp']
                b = batch.size(0)
                for m in bns:
                    m.momentum = b / float(seen + b)
                if device is not None:
                    batch = batch.to(device)
                model(batch)
                seen += b
        for m, mom in zip(bns, saved):
            m.momentum = mom
        model.train(was_training)
