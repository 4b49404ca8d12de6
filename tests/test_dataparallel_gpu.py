"""Data-parallel training step on real devices: two ranks, the bucketed GradSync all-reduce overlapped with the backward, and the
criterion's exchange of its 2 + 3C sums -- against ONE process that runs both shards and takes the loss of the gathered batch (what the
reference computes behind nn.DataParallel, training/trainer.py:520-524; per-replica BatchNorm statistics as there).

  * backend 'nccl' (= RCCL over xGMI): needs two GPUs, skipped on a one-GPU box;
  * backend 'gloo' with CUDA tensors: the same code path (side stream, bucket event, criterion hooks) with both ranks on GPU 0, so
    that the logic is exercised on the one-GPU test box too.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
CW = (0.2653, 0.7347)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _make(seed=0, bf16=False):
    from elektronn3_amd.unet import UNet
    torch.manual_seed(seed)
    m = UNet(1, 2, n_blocks=3, start_filts=32 if bf16 else 16)        # (the native bf16 kernels need multiples of 32 channels)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if 'norm' in n and n.endswith('weight'):
                p.copy_(1 + 0.2 * torch.randn_like(p))
            elif n.endswith('bias'):
                p.copy_(0.1 * torch.randn_like(p))
    return m.to(torch.bfloat16) if bf16 else m


def _batch(big=False):
    """big: 2 x 32 x 64 x 64 per rank = 1024 Winograd bricks at level 0 -- the persistent conv kernel and the one-round weight-gradient
    kernel run (the small case takes the one-brick-per-workgroup kernels)."""
    g = torch.Generator().manual_seed(1)
    sp = (32, 64, 64) if big else (16, 32, 32)
    x = torch.randn(4, 1, *sp, generator=g)
    t = (torch.rand(4, *sp, generator=g) < torch.tensor([0.1, 0.3, 0.6, 0.9]).view(4, 1, 1, 1)).long()     # unbalanced shards
    return x, t


def _worker(rank, world, port, backend, tmp, bf16=False, mode='overlap', big=False):
    """mode: 'serial' (GradSync's default: one all-reduce behind the backward), 'overlap' (bucket A reduced at the bucket event, 16 CUs
    reserved), 'overlap+spin' (as 'overlap', and a stand-in for RCCL's kernel -- 8 resident workgroups -- in front of the all-reduce)."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dev = torch.device('cuda', rank if backend == 'nccl' else 0)
    torch.cuda.set_device(dev)
    kw = {'device_id': dev} if backend == 'nccl' else {}
    if backend == 'nccl' and mode.startswith('overlap'):
        os.environ['NCCL_MAX_NCHANNELS'] = '16'      # the launcher's part of the contract: RCCL's kernel fits the CU reserve (GradSync refuses to overlap otherwise)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    try:
        from elektronn3_amd.dataparallel import GradSync, shard_batch
        from elektronn3_amd.loss import CombinedCEDiceLoss
        model = _make(bf16=bf16).to(dev).train()
        if mode == 'env':          # GradSync's own defaults: serial, unless the environment says otherwise (E3_DP_OVERLAP / E3_DP_CU_RESERVE: tests/test_switches_gpu.py)
            sync = GradSync(model, bucket_after_down_block=2)
            assert sync.overlap == (os.environ.get('E3_DP_OVERLAP') is not None), sync.mode
        else:
            sync = GradSync(model, bucket_after_down_block=2, overlap=mode != 'serial', cu_reserve=16)
            assert sync.overlap == (mode != 'serial') and sync.cu_reserve == (16 if sync.overlap else 0) and mode.split('+')[0] in sync.mode
        if mode == 'overlap+spin':
            from helpers import spin
            real = sync._allreduce

            def with_foreign_kernel(tns, async_op=False):
                spin(torch.cuda.current_stream(), 8, 300.0)          # resident on the side stream while the backward's tail runs
                return real(tns, async_op=async_op)
            sync._allreduce = with_foreign_kernel
        crit = CombinedCEDiceLoss(weight=CW, global_batch=True).to(dev)
        x, t = _batch(big)
        xr, tr = shard_batch(x, rank, world).to(dev), shard_batch(t, rank, world).to(dev)
        if bf16:
            xr = xr.to(torch.bfloat16)
        for step in range(2):                 # twice: the second backward must not be disturbed by the first one's buffers
            model.zero_grad(set_to_none=True)
            loss = crit(model(xr), tr)
            loss.backward()
            sync.wait()
        torch.cuda.synchronize()
        torch.save({'loss': float(loss), 'grads': {k: p.grad.float().cpu() for k, p in model.named_parameters()},
                    'buffers': {k: b.float().cpu() for k, b in model.named_buffers()}}, os.path.join(tmp, f'dp{rank}.pt'))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _reference(bf16=False, big=False):
    """One process, both shards through the same replica (per-shard BatchNorm statistics), ONE loss over the gathered logits."""
    from elektronn3_amd.loss import CombinedCEDiceLoss
    dev = torch.device('cuda', 0)
    model = _make(bf16=bf16).to(dev).train()
    crit = CombinedCEDiceLoss(weight=CW).to(dev)
    x, t = _batch(big)
    for step in range(2):
        model.zero_grad(set_to_none=True)
        outs = [model(x[r * 2:(r + 1) * 2].to(dev).to(torch.bfloat16 if bf16 else torch.float32)) for r in range(2)]
        loss = crit(torch.cat(outs, 0), t.to(dev))
        loss.backward()
    torch.cuda.synchronize()
    return float(loss), {k: p.grad.float().cpu() for k, p in model.named_parameters()}


@pytest.mark.parametrize('backend,mode', [('gloo', 'overlap'), ('gloo', 'env'), ('nccl', 'overlap'), ('nccl', 'env')])
def test_two_rank_train_step_equals_gathered_batch(backend, mode, tmp_path):
    if backend == 'nccl' and torch.cuda.device_count() < 2:
        pytest.skip('RCCL needs one GPU per rank; this box has one')
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), backend, str(tmp_path), False, mode), nprocs=world, join=True)
    loss_ref, g_ref = _reference()
    res = [torch.load(tmp_path / f'dp{r}.pt') for r in range(world)]
    gscale = max(float(g.norm()) for g in g_ref.values())
    for r in range(world):
        assert abs(res[r]['loss'] - loss_ref) < 1e-5 * max(1.0, abs(loss_ref)), (r, res[r]['loss'], loss_ref)
        for k, g in g_ref.items():
            got = res[r]['grads'][k]
            err = float((got - g).norm()) / max(float(g.norm()), 1e-4 * gscale)
            assert err < 2e-3, (backend, r, k, err)
    # every rank holds the same averaged gradients
    for k in g_ref:
        assert torch.allclose(res[0]['grads'][k], res[1]['grads'][k], rtol=0, atol=1e-6 * gscale), k


def test_two_rank_train_step_bf16_module(tmp_path):
    """BASELINE configs[2]'s combination: a bfloat16 module (native bf16 kernels) under GradSync.  The native backward hands fp32 parameter
    gradients to the buckets, the all-reduce averages them in fp32 and each parameter's .grad is rounded to bf16 once; the one-process
    reference accumulates the two shards' bf16 gradients in bf16, so the comparison is at bf16 resolution."""
    world = 2
    backend = 'nccl' if torch.cuda.device_count() >= 2 else 'gloo'
    mp.spawn(_worker, args=(world, _free_port(), backend, str(tmp_path), True), nprocs=world, join=True)
    loss_ref, g_ref = _reference(bf16=True)
    res = [torch.load(tmp_path / f'dp{r}.pt') for r in range(world)]
    gscale = max(float(g.norm()) for g in g_ref.values())
    for r in range(world):
        assert abs(res[r]['loss'] - loss_ref) < 2e-3 * max(1.0, abs(loss_ref)), (r, res[r]['loss'], loss_ref)
        for k, g in g_ref.items():
            err = float((res[r]['grads'][k] - g).norm()) / max(float(g.norm()), 1e-3 * gscale)
            assert err < 2e-2, (r, k, err)
    for k in g_ref:
        assert torch.equal(res[0]['grads'][k], res[1]['grads'][k]), k        # identical replicas after the all-reduce


@pytest.mark.parametrize('mode', ['serial', 'overlap+spin'])
def test_two_rank_step_beside_a_resident_foreign_kernel(mode, tmp_path):
    """The data-parallel step at a size where the persistent Winograd kernels run, (a) with GradSync's default (serial) mode and (b) with the
    overlapped bucket, 16 compute units reserved after the bucket event (E3_BWD_CU_RESERVE) and a stand-in for RCCL's kernel -- 8 workgroups
    resident on the side stream for 300 us -- in front of the real (gloo) all-reduce: same gradients as the gathered batch, identical
    replicas, and BatchNorm statistics that are bit-identical to the serial run's (the reserve only touches kernels after the event:
    data and weight gradients of the first encoder blocks)."""
    from helpers import spin_lib
    if mode != 'serial' and spin_lib() is None:
        pytest.skip('hipcc is not available: the resident stand-in kernel cannot be built')
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), 'gloo', str(tmp_path), False, mode, True), nprocs=world, join=True)
    res = [torch.load(tmp_path / f'dp{r}.pt') for r in range(world)]
    loss_ref, g_ref = _reference(big=True)
    gscale = max(float(g.norm()) for g in g_ref.values())
    for r in range(world):
        assert abs(res[r]['loss'] - loss_ref) < 1e-5 * max(1.0, abs(loss_ref)), (r, res[r]['loss'], loss_ref)
        for k, g in g_ref.items():
            err = float((res[r]['grads'][k] - g).norm()) / max(float(g.norm()), 1e-4 * gscale)
            assert err < 2e-3, (mode, r, k, err)
    for k in g_ref:
        assert torch.allclose(res[0]['grads'][k], res[1]['grads'][k], rtol=0, atol=1e-6 * gscale), k
    if mode != 'serial':       # against the serial run of the same ranks: statistics bit for bit, gradients to rounding (other split of the weight-gradient sums)
        sub = tmp_path / 'serial'
        sub.mkdir()
        mp.spawn(_worker, args=(world, _free_port(), 'gloo', str(sub), False, 'serial', True), nprocs=world, join=True)
        for r in range(world):
            ser = torch.load(sub / f'dp{r}.pt')
            for k, b in ser['buffers'].items():
                assert torch.equal(b, res[r]['buffers'][k]), (r, k)
            from helpers import is_prebn_bias
            for k, g in ser['grads'].items():
                if is_prebn_bias(k):       # analytically zero (bias in front of a train-mode BatchNorm): the value IS the rounding of the BatchNorm backward's
                    continue               # sums, whose order differs (with a CU reserve the reduce pass is its own kernel, without it part of the data gradient)
                err = float((res[r]['grads'][k] - g).norm()) / max(float(g.norm()), 1e-4 * gscale)
                assert err < 1e-5, (r, k, err)


def test_bench_n_gt_1_branch_dry_run_on_one_gpu():
    """`python bench.py --gpus 2` end to end -- launcher respawn (torch.distributed.run), process group, barriers, the
    MAX-reduced timed region, the global-batch criterion, GradSync, the tile-parallel Predictor leg and rank 0's ONE JSON line -- with the
    test-only `--backend gloo --share-device` (both ranks on cuda:0), so that bench.py's N > 1 branch has run before it meets an 8-GPU node."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'E3_DP_OVERLAP', 'NCCL_MAX_NCHANNELS')}
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--backend', 'gloo', '--share-device',
           '--predictor-volume', 'tiny']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    res = json.loads(lines[0])
    assert res['n_gpus'] == 2 and res['dist_ranks'] == 2 and res['backend'] == 'gloo' and res['rccl_ranks'] == 0 and 'test_only' in res
    assert res['steps'] == 2 and res['warmup'] == 1 and res['scaling'] == 'weak' and res['config']['global_batch'] == 4 and res['config']['parallelism'] == 'dp2'
    assert res['dp_mode'].startswith('serial') and res['ms_per_step'] > 0 and res['value'] > 0
    assert abs(res['value'] - 4 * 64 * 128 * 128 / (res['ms_per_step'] * 1e-3)) < 1e-6 * res['value']
    assert res['roofline']['launches_timed'] == 2 and 0 < res['roofline']['frac'] < 1       # one launch of the profiled layer per timed step
    assert 'cpu_baseline' not in res                       # rank 0 at N = 1 only
    p = res['predictor']
    assert p['value'] and p['n_gpus'] == 2 and p['tiles'] == 4 and p['finite'] and 'tile-parallel over 2 ranks' in p['parallelism']
