"""``elektronn3.models.resunet.UNet`` on the MI355X HIP path.

The reference's second U-Net file (``elektronn3/models/resunet.py``) is ``models.unet.UNet`` with its blocks rebuilt from ``ConvBlock``s
(conv1-norm1-act1-conv2-[+ shortcut]-norm2-act2, resunet.py:212-262): every encoder / decoder block is a ``Sequential`` of
``max(1, res_blocks)`` ConvBlocks, and ``res_blocks >= 1`` turns on the residual shortcuts -- identity where the channel counts agree, a
1x1x1 "projection" conv otherwise, none from the input image (resunet.py:264-312,386-457,888-934).  Same constructor, same state_dict
keys (``down_convs.i.convs.k.conv1.weight`` ..., ``up_convs.i.convs.k.proj.weight``), same call protocol as :class:`elektronn3_amd.unet.UNet`,
whose native executor runs it: a residual unit's conv writes its accumulations next to the shortcut, and the BatchNorm statistics pass sums the
two (csrc/unet_plan.cpp).  fp32 kernels; bf16 modules compute in fp32 on up-cast copies.
"""
from typing import Sequence, Union

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
        self.convs = nn.Sequential(*convs)
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
        self.convs = nn.Sequential(*convs)

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
