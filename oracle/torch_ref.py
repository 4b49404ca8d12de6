"""Functional torch restatement of ``UNet.forward`` -- TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/unet_oracle.py
for who may import this).

It issues exactly the ATen ops the reference's ``nn.Module`` graph issues (elektronn3/models/unet.py:244-253,
384-408, 894-916): ``conv3d -> batch_norm -> relu -> conv3d -> batch_norm -> relu -> max_pool3d(ceil_mode=True)``
per encoder block, ``conv_transpose3d -> (autocrop) -> batch_norm -> relu -> cat -> conv3d ...`` per decoder block,
1x1x1 ``conv3d`` head.  Parameters come from a state_dict with the reference's key names, so the same weights drive
this, the C oracle and the HIP path.  Used for

* ``bench.py``'s ``cpu_baseline`` leg: "the reference's CPU PyTorch path timed on the host cores" (north_star) --
  the reference's own Python cannot travel to the GPU box, its ATen call sequence can;
* full-size parity checks on the GPU box (same ops through PyTorch-ROCm/MIOpen, device='cuda').
"""
import torch
import torch.nn.functional as F


def _act(x, sd, name=None):
    """get_activation (unet.py:183-199): ReLU unless sd['__act_slope__'] says LeakyReLU(slope) (1.0 = nn.Identity, 'lin'; 2.0 = nn.SiLU)."""
    if name is not None and name + '.weight' in sd:      # nn.PReLU(num_parameters=1): learnable slope of THIS activation module
        return F.prelu(x, sd[name + '.weight'])
    s = sd.get('__act_slope__', 0.0)
    if s == 2.0:
        return F.silu(x)
    return F.relu(x) if s == 0.0 else F.leaky_relu(x, negative_slope=s)


def _bn(x, sd, name, training, momentum=0.1, eps=1e-5):
    if name + '.weight' not in sd:      # no parameters: nn.Identity (normalization='none' / full_norm=False, unet.py:77-80,238-242,
        if name in sd.get('__instance_norms__', ()):     # 369-375) or nn.InstanceNorm3d (affine=False, instance statistics always)
            return F.instance_norm(x, eps=eps)
        return x
    if name + '.running_mean' not in sd:         # nn.GroupNorm(num_groups, C): affine, no running statistics (unet.py:81-90)
        return F.group_norm(x, sd['__num_groups__'], sd[name + '.weight'], sd[name + '.bias'], eps=eps)
    return F.batch_norm(x, sd[name + '.running_mean'], sd[name + '.running_var'], sd[name + '.weight'], sd[name + '.bias'],
                        training=training, momentum=momentum, eps=eps)


def _conv(x, sd, name):
    w = sd[name + '.weight']
    pad = tuple((k - 1) // 2 for k in w.shape[2:])
    if sd.get('__valid__', False) and (name.endswith('.conv1') or name.endswith('.conv2')):
        pad = tuple(0 for _ in pad)             # conv_mode='valid': padding 0 in the convs of the blocks (unet.py:217,347)
    return (F.conv3d if w.dim() == 5 else F.conv2d)(x, w, sd[name + '.bias'], padding=pad)   # get_conv(dim), unet.py:47-54


def autocrop(from_down, from_up):
    """unet.py:256-325 restated (4D and 5D tensors)."""
    if from_down.shape[2:] == from_up.shape[2:]:
        return from_down, from_up
    ds, us = from_down.shape[2:], from_up.shape[2:]
    upcrop = [u - ((u - d) % 2) for d, u in zip(ds, us)]
    from_up = from_up[(slice(None), slice(None)) + tuple(slice(0, c) for c in upcrop)]
    us = from_up.shape[2:]
    from_down = from_down[(slice(None), slice(None)) + tuple(slice((d - u) // 2, (d + u) // 2) for d, u in zip(ds, us))]
    return from_down, from_up


def grid_attention(sd, p, x, g, training, momentum=0.1, eps=1e-5):
    """GridAttention.forward (unet.py:509-530) restated with ATen ops; p = 'up_convs.i.attention.'.  Returns (W(att * x) through its BatchNorm, att)."""
    nd = x.dim() - 2
    conv = F.conv3d if nd == 3 else F.conv2d
    mode = 'trilinear' if nd == 3 else 'bilinear'
    theta_x = conv(x, sd[p + 'theta.weight'], None, stride=2)
    phi_g = F.interpolate(conv(g, sd[p + 'phi.weight'], sd[p + 'phi.bias']), size=theta_x.shape[2:], mode=mode, align_corners=False)
    f = F.relu(theta_x + phi_g)
    att = torch.sigmoid(conv(f, sd[p + 'psi.weight'], sd[p + 'psi.bias']))
    att = F.interpolate(att, size=x.shape[2:], mode=mode, align_corners=False)
    wy = conv(att.expand_as(x) * x, sd[p + 'w.0.weight'], sd[p + 'w.0.bias'])
    wy = F.batch_norm(wy, sd[p + 'w.1.running_mean'], sd[p + 'w.1.running_var'], sd[p + 'w.1.weight'], sd[p + 'w.1.bias'],
                      training=training, momentum=momentum, eps=eps)
    return wy, att


def instance_norm_names(n_blocks, full_norm=True):
    """Names of the norm layers of UNet(normalization='instance') -- they have no state_dict entries; pass the result as
    sd['__instance_norms__']."""
    names = []
    for i in range(n_blocks):
        names += ([f'down_convs.{i}.norm0'] if full_norm else []) + [f'down_convs.{i}.norm1']
    for i in range(n_blocks - 1):
        names += ([f'up_convs.{i}.norm0', f'up_convs.{i}.norm1'] if full_norm else []) + [f'up_convs.{i}.norm2']
    return tuple(names)


def unet_forward(sd, x, n_blocks, planar_blocks=(), training=True, atts=None):
    """sd: name -> tensor (parameters may require grad; running stats are updated in place when training).  atts: optional list that
    receives the attention map of every decoder block (UpConvBlock.att, unet.py:394-395)."""
    enc = []
    for i in range(n_blocks):
        p = f'down_convs.{i}.'
        y = _act(_bn(_conv(x, sd, p + 'conv1'), sd, p + 'norm0', training), sd, p + 'act1')
        y = _act(_bn(_conv(y, sd, p + 'conv2'), sd, p + 'norm1', training), sd, p + 'act2')
        enc.append(y)
        if i < n_blocks - 1:
            if y.dim() == 4:
                x = F.max_pool2d(y, kernel_size=2, ceil_mode=True)
            else:
                x = F.max_pool3d(y, kernel_size=(1, 2, 2) if i in planar_blocks else 2, ceil_mode=True)
        else:
            x = y
    for i in range(n_blocks - 1):
        p = f'up_convs.{i}.'
        if p + 'upconv.conv.weight' in sd:      # up_mode='resizeconv_nearest': ResizeConv = nn.Upsample(nearest) + conv3 (unet.py:411-449)
            w = sd[p + 'upconv.conv.weight']
            scale = (1, 2, 2) if (w.dim() == 5 and (n_blocks - 2 - i) in planar_blocks) else 2      # planar decoder block: only (H, W) grow
            lin = sd.get('__up_linear__', False)      # 'resizeconv_linear': nn.Upsample(mode='trilinear' | 'bilinear'), align_corners=False
            xu = F.interpolate(x, scale_factor=scale, mode=('trilinear' if w.dim() == 5 else 'bilinear'), align_corners=False) if lin \
                else F.interpolate(x, scale_factor=scale, mode='nearest')
            up = _conv(xu, sd, p + 'upconv.conv')
        else:
            w = sd[p + 'upconv.weight']
            up = (F.conv_transpose3d if w.dim() == 5 else F.conv_transpose2d)(x, w, sd[p + 'upconv.bias'], stride=tuple(w.shape[2:]))
        skip, up = autocrop(enc[-(i + 2)], up)
        if p + 'attention.theta.weight' in sd:      # attention=True: the (cropped) skip is gated by the block's input (unet.py:391-393)
            skip, att = grid_attention(sd, p + 'attention.', skip, x, training)
            if atts is not None:
                atts.append(att)
        up = _act(_bn(up, sd, p + 'norm0', training), sd, p + 'act0')
        cat = sd[p + 'conv1.weight'].shape[1] == 2 * up.shape[1]      # merge_mode 'concat' vs 'add' (unet.py:398-401) shows in conv1's Cin
        y = torch.cat((up, skip), 1) if cat else up + skip
        y = _act(_bn(_conv(y, sd, p + 'conv1'), sd, p + 'norm1', training), sd, p + 'act1')
        x = _act(_bn(_conv(y, sd, p + 'conv2'), sd, p + 'norm2', training), sd, p + 'act2')
    return _conv(x, sd, 'conv_final')


def resunet_forward(sd, x, n_blocks, planar_blocks=(), training=True, enc_res_blocks=0, dec_res_blocks=0, atts=None):
    """elektronn3.models.resunet.UNet.forward (resunet.py:944-967) restated with ATen ops: DownBlock / UpBlock = Sequential of ConvBlocks
    (resunet.py:254-262: conv1-norm1-act1-conv2-[+ proj(inp)]-norm2-act2), max(1, res_blocks) per block; shortcuts when res_blocks >= 1, none
    from the input image (resunet.py:287-299,906).  dim=3 only (the reference's ConvBlocks are Conv3d whatever `dim` says)."""
    def conv_block(p, inp, residual):
        y = _act(_bn(_conv(inp, sd, p + 'conv1'), sd, p + 'norm1', training), sd, p + 'act1')
        y = _conv(y, sd, p + 'conv2')
        if residual:
            y = y + (F.conv3d(inp, sd[p + 'proj.weight'], sd[p + 'proj.bias']) if p + 'proj.weight' in sd else inp)
        return _act(_bn(y, sd, p + 'norm2', training), sd, p + 'act2')

    enc = []
    for i in range(n_blocks):
        y = x
        for c in range(max(1, enc_res_blocks)):
            y = conv_block(f'down_convs.{i}.convs.{c}.', y, enc_res_blocks >= 1 and not (c == 0 and i == 0))
        enc.append(y)
        x = F.max_pool3d(y, kernel_size=(1, 2, 2) if i in planar_blocks else 2, ceil_mode=True) if i < n_blocks - 1 else y
    for i in range(n_blocks - 1):
        p = f'up_convs.{i}.'
        if p + 'upconv.conv.weight' in sd:
            scale = (1, 2, 2) if (n_blocks - 2 - i) in planar_blocks else 2
            xu = F.interpolate(x, scale_factor=scale, mode='trilinear', align_corners=False) if sd.get('__up_linear__', False) \
                else F.interpolate(x, scale_factor=scale, mode='nearest')
            up = _conv(xu, sd, p + 'upconv.conv')
        else:
            w = sd[p + 'upconv.weight']
            up = F.conv_transpose3d(x, w, sd[p + 'upconv.bias'], stride=tuple(w.shape[2:]))
        skip, up = autocrop(enc[-(i + 2)], up)
        if p + 'attention.theta.weight' in sd:
            skip, att = grid_attention(sd, p + 'attention.', skip, x, training)
            if atts is not None:
                atts.append(att)
        up = _act(_bn(up, sd, p + 'norm0', training), sd, p + 'act0')
        cat = sd[p + 'convs.0.conv1.weight'].shape[1] == 2 * up.shape[1]
        y = torch.cat((up, skip), 1) if cat else up + skip
        for c in range(max(1, dec_res_blocks)):
            y = conv_block(p + f'convs.{c}.', y, dec_res_blocks >= 1)
        x = y
    return _conv(x, sd, 'conv_final')


def combined_loss(logits, target, class_weights=(0.2653, 0.7347)):
    """0.5*CrossEntropy(weight) + 0.5*Dice(softmax, weight) -- the example's criterion
    (examples/train_unet_neurodata.py:294-296; modules/loss.py:19-49,165-189), restated with torch ops."""
    cw = torch.tensor(class_weights, dtype=logits.dtype, device=logits.device)
    ce = F.cross_entropy(logits, target, weight=cw)
    probs = torch.softmax(logits, 1)
    onehot = torch.zeros_like(probs).scatter_(1, target.unsqueeze(1), 1)
    dims = (0,) + tuple(range(2, logits.dim()))
    num = 2 * (probs * onehot).sum(dims)
    den = (probs + onehot).sum(dims) + 1e-4
    dice = (cw * (1 - num / den)).mean()
    return 0.5 * ce + 0.5 * dice
