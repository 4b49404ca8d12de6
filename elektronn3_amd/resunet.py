"""``elektronn3.models.resunet.UNet`` on the MI355X HIP path.

The reference's second U-Net file (``elektronn3/models/resunet.py``) is ``models.unet.UNet`` with its blocks rebuilt from ``ConvBlock``s
(conv1-norm1-act1-conv2-[+ shortcut]-norm2-act2, resunet.py:212-262): every encoder / decoder block is a ``Sequential`` of
``max(1, res_blocks)`` ConvBlocks, and ``res_blocks >= 1`` turns on the residual shortcuts -- identity where the channel counts agree, a
1x1x1 "projection" conv otherwise, none from the input image (resunet.py:264-312,386-457,888-934).  Same constructor, same state_dict
keys (``down_convs.i.convs.k.conv1.weight`` ..., ``up_convs.i.convs.k.proj.weight``), same call protocol as :class:`elektronn3_amd.unet.UNet`,
whose native executor runs it: a residual unit's conv writes its accumulations next to the shortcut, and the BatchNorm statistics pass sums the
two (csrc/unet_plan.cpp).  fp32 kernels; bf16 modules compute in fp32 on up-cast copies.
"""
from typing import List, Sequence, Union

import torch

from torch import nn

from . import unet as _unet
from .unet import DummyAttention, GridAttention, ResizeConv, _LAYERS, _make_activation, _norm_factory

__all__ = ['UNet', 'ConvBlock', 'DownBlock', 'UpBlock']


class ConvBlock(nn.Module):
    """Parameter container of the reference's ConvBlock (resunet.py:212-262)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, planar=False, activation='relu', normalization=None, dim=3, conv_mode='same',
                 residual=False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.normalization, self.conv_mode, self.activation, self.residual, self.dim = normalization, conv_mode, activation, residual, dim
        Conv, Norm = _LAYERS[dim][0], _LAYERS[dim][3]
        pad = 0 if (conv_mode == 'valid' or kernel_size == 1) else 1            # get_padding, resunet.py:203-209
        k, p = ((1, kernel_size, kernel_size), (0, pad, pad)) if (planar and dim == 3) else (kernel_size, pad)
        norm = _norm_factory(normalization or 'none', Norm, dim, out_channels)
        self.conv1 = Conv(in_channels, out_channels, kernel_size=k, padding=p)
        self.norm1 = norm()
        self.act1 = _make_activation(activation)
        self.conv2 = Conv(out_channels, out_channels, kernel_size=k, padding=p)
        self.norm2 = norm()
        self.act2 = _make_activation(activation)
        # "projection" to match the channel counts for the residual addition (resunet.py:247-251)
        self.proj = Conv(in_channels, out_channels, kernel_size=1) if (residual and in_channels != out_channels) else nn.Identity()

    def forward(self, inp):
        raise RuntimeError('elektronn3_amd sub-modules only hold parameters; call UNet.forward')


class DownBlock(nn.Module):
    """Parameter container of the reference's DownBlock (resunet.py:264-312)."""

    def __init__(self, in_channels, out_channels, pooling=True, planar=False, activation='relu', normalization=None, dim=3, conv_mode='same',
                 res_blocks=0, skip_first_residual=False):
        super().__init__()
        self.in_channels, self.out_channels, self.pooling, self.normalization, self.res_blocks, self.dim = \
            in_channels, out_channels, pooling, normalization, res_blocks, dim
        on = res_blocks >= 1
        convs = [ConvBlock(in_channels, out_channels, planar=planar, activation=activation, normalization=normalization, conv_mode=conv_mode,
                           residual=on and not skip_first_residual)]
        for _ in range(res_blocks - 1):
            convs.append(ConvBlock(out_channels, out_channels, planar=planar, activation=activation, normalization=normalization,
                                   conv_mode=conv_mode, residual=on))
        self.convs = nn.ModuleList(convs)      # (the reference holds them in a Sequential: same state_dict keys; a ModuleList keeps TorchScript from compiling a chained forward)
        self.pool = _LAYERS[dim][2](kernel_size=(1, 2, 2) if planar else 2, ceil_mode=True) if pooling else nn.Identity()

    def forward(self, x):
        raise RuntimeError('elektronn3_amd sub-modules only hold parameters; call UNet.forward')


class UpBlock(nn.Module):
    """Parameter container of the reference's UpBlock (resunet.py:386-457)."""

    def __init__(self, in_channels, out_channels, merge_mode='concat', up_mode='transpose', planar=False, activation='relu', normalization=None,
                 full_norm=True, dim=3, conv_mode='same', attention=False, res_blocks=0):
        super().__init__()
        self.in_channels, self.out_channels, self.merge_mode, self.up_mode = in_channels, out_channels, merge_mode, up_mode
        self.normalization, self.res_blocks, self.dim = normalization, res_blocks, dim
        ConvT, Norm = _LAYERS[dim][1], _LAYERS[dim][3]
        ks = (1, 2, 2) if (planar and dim == 3) else 2
        if up_mode == 'transpose':
            self.upconv = ConvT(in_channels, out_channels, kernel_size=ks, stride=ks)
        else:
            mode = 'nearest' if 'nearest' in up_mode else ('trilinear' if dim == 3 else 'bilinear')
            self.upconv = ResizeConv(in_channels, out_channels, planar=planar, dim=dim, upsampling_mode=mode, kernel_size=1 if up_mode.endswith('1') else 3)
        self.act0 = _make_activation(activation)
        self.norm0 = _norm_factory(normalization or 'none', Norm, dim, out_channels)()      # always built (resunet.py:411)
        self.attention = GridAttention(in_channels=in_channels // 2, gating_channels=in_channels, dim=dim) if attention else DummyAttention()
        self.att = None
        on = res_blocks >= 1
        first_in = 2 * out_channels if merge_mode == 'concat' else out_channels
        convs = [ConvBlock(first_in, out_channels, planar=planar, activation=activation, normalization=normalization, conv_mode=conv_mode, residual=on)]
        for _ in range(res_blocks - 1):
            convs.append(ConvBlock(out_channels, out_channels, planar=planar, activation=activation, normalization=normalization,
                                   conv_mode=conv_mode, residual=on))
        self.convs = nn.ModuleList(convs)      # (the reference holds them in a Sequential: same state_dict keys; a ModuleList keeps TorchScript from compiling a chained forward)

    def forward(self, enc, dec):
        raise RuntimeError('elektronn3_amd sub-modules only hold parameters; call UNet.forward')


class UNet(_unet.UNet):
    """Drop-in for ``elektronn3.models.resunet.UNet`` (resunet.py:598-934): the reference's constructor, ``enc_res_blocks`` / ``dec_res_blocks``
    included.  ``full_norm`` is accepted and -- as in the reference, whose blocks always build every norm -- has no effect."""

    def __init__(
            self,
            in_channels: int = 1,
            out_channels: int = 2,
            n_blocks: int = 3,
            start_filts: int = 32,
            up_mode: str = 'transpose',
            merge_mode: str = 'concat',
            enc_res_blocks: int = 0,
            dec_res_blocks: int = 0,
            planar_blocks: Sequence = (),
            batch_norm: str = 'unset',
            attention: bool = False,
            activation: Union[str, nn.Module] = 'relu',
            normalization: str = 'batch',
            full_norm: bool = True,
            dim: int = 3,
            conv_mode: str = 'same',
    ):
        nn.Module.__init__(self)
        self._setup(in_channels, out_channels, n_blocks, start_filts, up_mode, merge_mode, planar_blocks, batch_norm, attention, activation,
                    normalization, True, dim, conv_mode, res_blocks=(enc_res_blocks, dec_res_blocks))

    def _build_blocks(self):
        outs = self.in_channels
        for i in range(self.n_blocks):
            ins = self.in_channels if i == 0 else outs
            outs = self.start_filts * (2 ** i)
            self.down_convs.append(DownBlock(ins, outs, pooling=i < self.n_blocks - 1, planar=i in self.planar_blocks, activation=self.activation,
                                             normalization=self.normalization, dim=self.dim, conv_mode=self.conv_mode,
                                             res_blocks=self.enc_res_blocks, skip_first_residual=(i == 0)))
        for i in range(self.n_blocks - 1):
            ins = outs
            outs = ins // 2
            self.up_convs.append(UpBlock(ins, outs, up_mode=self.up_mode, merge_mode=self.merge_mode, planar=(self.n_blocks - 2 - i) in self.planar_blocks,
                                         activation=self.activation, normalization=self.normalization, attention=self.attention, dim=self.dim,
                                         conv_mode=self.conv_mode, res_blocks=self.dec_res_blocks))
        return outs

    def _variant_key(self):
        return (1, int(self.enc_res_blocks), int(self.dec_res_blocks))

    def _scripted_forward(self, x: torch.Tensor) -> torch.Tensor:
        """``forward`` as TorchScript sees it (the reference's resunet is scriptable and ``Trainer._save_model`` scripts the model when
        ``save_jit='script'``, trainer.py:871-887): the tensors in the native parameter-table order -- per ConvBlock: conv1, norm1, [act1],
        conv2, norm2, [act2], [proj]; per decoder block the up-convolution with norm0 / [act0] first -- and ONE call of e3unet::unet_fwd."""
        if not self._script_ok:
            raise RuntimeError('scripted elektronn3_amd.resunet.UNet: batch / group / no normalization, no attention, activations other than rrelu')
        t: List[torch.Tensor] = []
        bufs: List[torch.Tensor] = []
        counters: List[torch.Tensor] = []
        mom: List[float] = []
        for blk in self.down_convs:
            for cb in blk.convs:
                t.append(cb.conv1.weight)
                b1 = cb.conv1.bias
                assert b1 is not None
                t.append(b1)
                if hasattr(cb.norm1, 'weight'):
                    t.append(cb.norm1.weight); t.append(cb.norm1.bias)
                    if hasattr(cb.norm1, 'running_mean'):
                        rm, rv, nb, mo = cb.norm1.running_mean, cb.norm1.running_var, cb.norm1.num_batches_tracked, cb.norm1.momentum
                        assert rm is not None and rv is not None and nb is not None and mo is not None
                        t.append(rm); t.append(rv); bufs.append(rm); bufs.append(rv); counters.append(nb); mom.append(mo)
                if hasattr(cb.act1, 'weight'):
                    t.append(cb.act1.weight)
                t.append(cb.conv2.weight)
                b2 = cb.conv2.bias
                assert b2 is not None
                t.append(b2)
                if hasattr(cb.norm2, 'weight'):
                    t.append(cb.norm2.weight); t.append(cb.norm2.bias)
                    if hasattr(cb.norm2, 'running_mean'):
                        rm, rv, nb, mo = cb.norm2.running_mean, cb.norm2.running_var, cb.norm2.num_batches_tracked, cb.norm2.momentum
                        assert rm is not None and rv is not None and nb is not None and mo is not None
                        t.append(rm); t.append(rv); bufs.append(rm); bufs.append(rv); counters.append(nb); mom.append(mo)
                if hasattr(cb.act2, 'weight'):
                    t.append(cb.act2.weight)
                if hasattr(cb.proj, 'weight'):
                    t.append(cb.proj.weight)
                    bp = cb.proj.bias
                    assert bp is not None
                    t.append(bp)
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
            for cb in ub.convs:
                t.append(cb.conv1.weight)
                b1 = cb.conv1.bias
                assert b1 is not None
                t.append(b1)
                if hasattr(cb.norm1, 'weight'):
                    t.append(cb.norm1.weight); t.append(cb.norm1.bias)
                    if hasattr(cb.norm1, 'running_mean'):
                        rm, rv, nb, mo = cb.norm1.running_mean, cb.norm1.running_var, cb.norm1.num_batches_tracked, cb.norm1.momentum
                        assert rm is not None and rv is not None and nb is not None and mo is not None
                        t.append(rm); t.append(rv); bufs.append(rm); bufs.append(rv); counters.append(nb); mom.append(mo)
                if hasattr(cb.act1, 'weight'):
                    t.append(cb.act1.weight)
                t.append(cb.conv2.weight)
                b2 = cb.conv2.bias
                assert b2 is not None
                t.append(b2)
                if hasattr(cb.norm2, 'weight'):
                    t.append(cb.norm2.weight); t.append(cb.norm2.bias)
                    if hasattr(cb.norm2, 'running_mean'):
                        rm, rv, nb, mo = cb.norm2.running_mean, cb.norm2.running_var, cb.norm2.num_batches_tracked, cb.norm2.momentum
                        assert rm is not None and rv is not None and nb is not None and mo is not None
                        t.append(rm); t.append(rv); bufs.append(rm); bufs.append(rv); counters.append(nb); mom.append(mo)
                if hasattr(cb.act2, 'weight'):
                    t.append(cb.act2.weight)
                if hasattr(cb.proj, 'weight'):
                    t.append(cb.proj.weight)
                    bp = cb.proj.bias
                    assert bp is not None
                    t.append(bp)
        t.append(self.conv_final.weight)
        bf = self.conv_final.bias
        assert bf is not None
        t.append(bf)
        if self._script_key[5] == 2.0:       # nn.GroupNorm: statistics per sample in training and eval mode alike: one call per sample
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
