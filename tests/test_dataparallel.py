"""Data-parallel path on CPU with gloo, world_size 2 (the 8-GPU run is the driver's): bucketed all-reduce of the flat
gradient buffer (GradSync), batch sharding, tile-parallel Predictor sharding."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, tmp):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from elektronn3_amd.dataparallel import GradSync, shard_batch
        from elektronn3_amd.unet import UNet
        torch.manual_seed(0)
        model = UNet(1, 2, n_blocks=3, start_filts=8)          # CPU: parameters only, no forward
        sync = GradSync(model, bucket_after_down_block=2)
        plan = model._plan()
        tens = [model.get_parameter(n) if k == 0 else model.get_buffer(n) for n, k in zip(plan.names, plan.kinds)]
        flat, views = sync.flat_views(plan, tens)
        # table order puts the first two encoder blocks in a prefix = bucket B
        n_prefix = sum(p.numel() for n, p in model.named_parameters() if n.startswith(('down_convs.0.', 'down_convs.1.')))
        assert sync._split == n_prefix and 0 < sync._split < flat.numel()
        assert sum(v.numel() for v in views if v is not None) == sum(p.numel() for p in model.parameters()) == flat.numel()
        # every backward gets its own buffer (autograd may keep views of the previous one: per-sample norms, gradient accumulation)
        flat_prev = flat
        flat, views = sync.flat_views(plan, tens)
        assert flat.data_ptr() != flat_prev.data_ptr()
        # rank-dependent "gradients"
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(flat.numel(), generator=g)
        flat.copy_(local)
        assert sync.bucket_event() is None                    # CPU: no HIP event
        sync.after_backward(plan)
        # expected: mean over ranks
        exp = sum(torch.randn(flat.numel(), generator=torch.Generator().manual_seed(100 + r)) for r in range(world)) / world
        assert torch.allclose(flat, exp, atol=1e-6)
        # views alias the flat buffer
        k = next(i for i, v in enumerate(views) if v is not None)
        views[k].zero_()
        assert float(flat[:views[k].numel()].abs().sum()) == 0.0
        # batch sharding
        batch = torch.arange(8).view(8, 1)
        assert shard_batch(batch, rank, world).flatten().tolist() == list(range(rank * 4, rank * 4 + 4))
        # parameter broadcast makes replicas identical
        with torch.no_grad():
            for p in model.parameters():
                p.add_(rank)
        sync.broadcast_parameters(0)
        ref = UNet(1, 2, n_blocks=3, start_filts=8)
        torch.manual_seed(0)
        ref = UNet(1, 2, n_blocks=3, start_filts=8)
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert torch.equal(p, q), n
        # tile-parallel Predictor: generic module on CPU, tiles sharded round-robin, rank 0 assembles
        from elektronn3_amd.inference import Predictor
        conv = torch.nn.Conv3d(1, 2, 3, padding=1)
        torch.manual_seed(5)
        with torch.no_grad():
            conv.weight.copy_(torch.randn_like(conv.weight)); conv.bias.copy_(torch.randn_like(conv.bias))
        vol = torch.randn(1, 1, 8, 16, 16, generator=torch.Generator().manual_seed(9))
        kw = dict(device='cpu', tile_shape=(4, 8, 8), overlap_shape=(2, 2, 2), offset=(0, 0, 0), out_shape=(2, 8, 16, 16), apply_softmax=True)
        y_par = Predictor(conv, tile_parallel=True, **kw).predict(vol)
        y_one = Predictor(conv, tile_parallel=False, **kw).predict(vol)
        if rank == 0:
            assert torch.allclose(y_par, y_one, atol=1e-6)
        # the criterion over a sharded minibatch: per-rank sums + ONE all-reduce = the loss of the gathered batch
        # (what the reference computes on GPU 0), which the mean of the per-rank losses is not
        import oracle.unet_oracle as O
        from oracle.torch_ref import combined_loss
        gen = torch.Generator().manual_seed(3)
        logits = torch.randn(4, 2, 3, 5, 6, generator=gen, dtype=torch.float64) * 2
        target = (torch.rand(4, 3, 5, 6, generator=gen) < torch.tensor([0.1, 0.3, 0.6, 0.9]).view(4, 1, 1, 1)).long()   # unbalanced shards
        logits[:2, 1] += 3 * target[:2]                      # shard 0 predicts well, shard 1 does not
        cw = (0.2653, 0.7347)
        sums = torch.from_numpy(O.ce_dice_sums(shard_batch(logits, rank, world).numpy(), shard_batch(target, rank, world).numpy(), cw))
        local = O.ce_dice_from_sums(sums.numpy(), cw)
        dist.all_reduce(sums)
        glob = O.ce_dice_from_sums(sums.numpy(), cw)
        want = float(combined_loss(logits, target, cw))
        assert abs(glob - want) < 1e-12, (glob, want)
        locs = [None] * world
        dist.all_gather_object(locs, local)
        assert abs(sum(locs) / world - want) > 1e-4           # the difference the sharded mode removes
        # the module's own hooks for that exchange (the device kernels around them are covered by the GPU tests)
        from elektronn3_amd.loss import CombinedCEDiceLoss
        crit = CombinedCEDiceLoss(weight=cw, global_batch=True)
        assert crit._world() == world and crit.grads_averaged
        mine = torch.arange(8, dtype=torch.float64) + 10 * rank
        assert torch.equal(crit._reduce_sums(mine.clone()), sum(torch.arange(8, dtype=torch.float64) + 10 * r for r in range(world)))
        assert CombinedCEDiceLoss(weight=cw)._world() == world and not CombinedCEDiceLoss(weight=cw).global_batch
        open(os.path.join(tmp, f'ok{rank}'), 'w').write('ok')
    finally:
        dist.destroy_process_group()


def test_gradsync_and_tile_parallel_gloo_world2(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f'ok{r}').exists() for r in range(world))


def test_gradsync_configures_its_mode_from_the_environment(monkeypatch):
    """overlap only when RCCL's channel bound fits the CU reserve (the launcher exported NCCL_MAX_NCHANNELS before the process
    group was created); an overlap request without it runs serial and says so; backends without a device kernel need no bound."""
    import warnings
    from elektronn3_amd import dataparallel as dp
    from elektronn3_amd.unet import UNet
    model = UNet(1, 2, n_blocks=2, start_filts=8)
    for k in ('E3_DP_OVERLAP', 'NCCL_MAX_NCHANNELS', 'E3_DP_CU_RESERVE'):
        monkeypatch.delenv(k, raising=False)
    s = dp.GradSync(model)
    assert not s.overlap and s.cu_reserve == 0 and s.mode.startswith('serial') and 'default' in s.mode
    # the bound is in the environment: overlap configures itself
    monkeypatch.setenv('NCCL_MAX_NCHANNELS', '16')
    s = dp.GradSync(model)
    assert s.overlap and s.cu_reserve == 16 and 'NCCL_MAX_NCHANNELS=16' in s.mode
    monkeypatch.setenv('NCCL_MAX_NCHANNELS', '32')         # larger than the default reserve: not a bound
    assert not dp.GradSync(model).overlap
    assert dp.GradSync(model, cu_reserve=32).overlap
    assert not dp.GradSync(model, overlap=False, cu_reserve=32).overlap
    # an RCCL group (faked: no GPU here) without the bound: an explicit request is refused, loudly
    monkeypatch.delenv('NCCL_MAX_NCHANNELS')
    monkeypatch.setattr(dp.dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dp.dist, 'get_backend', lambda group=None: 'nccl')
    monkeypatch.setattr(dp.dist, 'get_world_size', lambda group=None: 2)
    with pytest.warns(RuntimeWarning, match='running SERIAL'):
        s = dp.GradSync(model, overlap=True)
    assert not s.overlap and s.cu_reserve == 0 and 'refused' in s.mode
    monkeypatch.setenv('E3_DP_OVERLAP', '1')
    with pytest.warns(RuntimeWarning, match='NCCL_MAX_NCHANNELS'):
        assert not dp.GradSync(model).overlap
    monkeypatch.setenv('NCCL_MAX_NCHANNELS', '8')
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        s = dp.GradSync(model, overlap=True)
    assert s.overlap and s.cu_reserve == 16 and 'requested' in s.mode
    # gloo: no device kernel of its own, no bound needed
    monkeypatch.delenv('NCCL_MAX_NCHANNELS')
    monkeypatch.setattr(dp.dist, 'get_backend', lambda group=None: 'gloo')
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        assert dp.GradSync(model, overlap=True).overlap
