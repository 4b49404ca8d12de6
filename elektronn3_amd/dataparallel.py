"""Data-parallel training for :class:`elektronn3_amd.unet.UNet`: one process per GPU, RCCL all-reduce over xGMI.

Replaces ``torch.nn.DataParallel`` as the reference uses it (benchmark/train_benchmark.py:109-110,
elektronn3/models/base.py:48-49): instead of one process that re-broadcasts all parameters, scatters the batch and
reduces gradients onto GPU 0 every iteration, every rank owns a replica and a minibatch shard (samples are the
independent units) and the only exchange is ONE sum of the flat fp32 gradient buffer
(5.6 M floats = 22.4 MB for cfg 2), issued on a side HIP stream -- by default as one collective behind the backward
(``overlap=False``), with ``overlap=True`` in two buckets:

  bucket A  every layer except the first ``bucket_after_down_block`` encoder blocks.  Its gradients are complete
            while the backward is still working through the full-resolution encoder blocks (which hold only ~3 %
            of the parameters but ~30 % of the backward time) -- libe3unet records a HIP event at that point and the
            all-reduce of A overlaps with the rest of the backward.
  bucket B  the remaining (tiny) prefix of the buffer, reduced when the backward has finished.

The compute stream only waits (stream-level, no host sync) for both before autograd hands the gradients on.
BatchNorm statistics stay per rank, as with ``nn.DataParallel`` replicas (no SyncBN in the reference).
Gradients are AVERAGED over ranks (the reference computes one mean loss over the gathered global batch).
"""
import torch
import torch.distributed as dist


class GradSync:
    """``overlap``: False (default) = both buckets are reduced when the backward has finished -- the all-reduce of 22 MB costs ~3-4 % of a
    cfg-2 step on xGMI and nothing else is touched.  True (or ``E3_DP_OVERLAP=1``) = bucket A is reduced on the side stream WHILE the backward
    works through the first encoder blocks; the kernels launched after the bucket's event then leave ``cu_reserve`` compute units (a multiple of
    8, default 16 = two per XCD; ``E3_DP_CU_RESERVE``) to the collective's resident workgroups: the persistent conv kernels occupy every CU they
    get with one 512-register workgroup, and one that found its CU taken would wait a whole round (measured: +55 % on the step with ONE foreign
    wave, tools/probe_foreign_waves.py).  ``NCCL_MAX_NCHANNELS`` must be ``<= cu_reserve`` in the environment before the process group's first collective, so that RCCL's
    kernel has at most that many workgroups.  The constructor does not change the environment; it configures itself from it (see the comment in ``__init__``):
    ``overlap=None`` overlaps iff that bound (or ``E3_DP_OVERLAP``) is exported, an overlap request that RCCL's channel count does not fit runs serial with a warning,
    and so does a device on which no quiet side stream is found.  ``GradSync.mode`` reports the choice and its reason (``bench.py`` prints it as ``dp_mode``)."""

    def __init__(self, model, process_group=None, bucket_after_down_block=2, average=True, overlap=None, cu_reserve=None):
        import os
        if isinstance(model, torch.jit.ScriptModule):
            # (the TorchScript operator's backward has no handle on this object: a scripted replica would silently skip the all-reduce)
            raise TypeError('GradSync needs the eager elektronn3_amd.UNet: script the model for saving (Trainer save_jit), train the eager one')
        self.model = model
        self.group = process_group
        self.average = average
        self.bucket_after_down_block = int(min(bucket_after_down_block, model.n_blocks))
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # world == 1 normally short-circuits; `force` keeps the whole event/side-stream/all-reduce path alive (tests)
        self.force = bool(int(os.environ.get('E3_FORCE_GRADSYNC', '0')))
        # Mode: overlap needs RCCL's kernel bounded to the reserved compute units, and that bound (NCCL_MAX_NCHANNELS) is read when
        # the communicator is created -- the launcher's business, before torch.distributed's first collective (bench.py --dp-overlap exports it in front
        # of init_process_group).  The constructor never touches the environment; it CONFIGURES ITSELF from it:
        #   overlap=None   overlap iff E3_DP_OVERLAP is set or NCCL_MAX_NCHANNELS <= the reserve is already exported; serial otherwise
        #   overlap=True   overlap -- but with an RCCL process group whose channel bound is missing or larger than the reserve: serial + a RuntimeWarning
        #                  (an unbounded collective beside one-512-register-workgroup-per-CU kernels is the slowest form measured, never a safe default)
        #   overlap=False  serial
        # Backends without a device kernel of their own (gloo; the tests' `collective` stand-in) need no bound.  `mode` says what was chosen and why.
        if cu_reserve is None:
            cu_reserve = int(os.environ.get('E3_DP_CU_RESERVE', '16'))
        reserve = max(0, min(128, int(cu_reserve))) // 8 * 8
        ch = os.environ.get('NCCL_MAX_NCHANNELS')
        bounded = ch is not None and ch.isdigit() and 0 < int(ch) <= reserve
        try:
            rccl = dist.is_initialized() and dist.get_backend(process_group) == 'nccl'
        except Exception:      # (no default group yet)
            rccl = False
        requested = (os.environ.get('E3_DP_OVERLAP') is not None or bounded) if overlap is None else bool(overlap)
        self.why = 'requested' if requested else 'default'
        if requested and rccl and reserve and not bounded:
            import warnings
            warnings.warn(f'GradSync: overlap requested with cu_reserve={reserve}, but NCCL_MAX_NCHANNELS={ch!r} does not bound RCCL\'s kernel to the reserved '
                          f'compute units (export NCCL_MAX_NCHANNELS={reserve} before the process group is created): running SERIAL', RuntimeWarning, stacklevel=2)
            requested = False
            self.why = f'overlap refused: NCCL_MAX_NCHANNELS={ch!r} > reserve {reserve}'
        elif requested and overlap is None and bounded:
            self.why = f'NCCL_MAX_NCHANNELS={ch} <= reserve {reserve}'
        self.overlap = requested
        self.cu_reserve = reserve if self.overlap else 0
        self.collective = None        # tests / probes: callable(tensor) run on the side stream in place of the all-reduce
        self._flat = None
        self._views = None
        self._split = 0
        self._comm_stream = None
        self._event = None
        self._works = []
        # plain attribute, not a sub-module/parameter: keeps state_dict and pickling of the model unchanged
        object.__setattr__(model, '_grad_sync', self)

    @property
    def mode(self):
        return (f'overlap (bucket event after down block {self.bucket_after_down_block}, {self.cu_reserve} CUs reserved; {self.why})' if self.overlap
                else f'serial (all-reduce after the backward; {self.why})')

    # -- called from _UNetFunction.backward ------------------------------------------------------------------
    def flat_views(self, plan, tens):
        # A FRESH flat buffer per backward (the caching allocator makes that free): autograd keeps the returned views by reference
        # (InputBuffer of a parameter fed by several Function nodes -- one native call per sample with instance / group norm -- and
        # AccumulateGrad, which may adopt a view as p.grad when gradients are accumulated over several backward passes), so a
        # persistent buffer would be overwritten under them.
        dev = tens[0].device
        sizes = [t.numel() if k == 0 else 0 for t, k in zip(tens, plan.kinds)]
        self._flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        self._views, off, split = [], 0, 0
        prefixes = tuple(f'down_convs.{i}.' for i in range(self.bucket_after_down_block))
        for name, n in zip(plan.names, sizes):
            self._views.append(self._flat[off:off + n] if n else None)
            off += n
            if prefixes and name.startswith(prefixes):
                split = off          # table order == forward order: the first encoder blocks are a prefix
        self._split = split
        return self._flat, self._views

    def _side_stream(self):
        """The quiet side stream of this device (probed once per device and process), or None -- overlap is then switched off for good:
        a side stream whose hardware queue slows every kernel of the compute stream costs more than the hidden collective is worth."""
        if self._comm_stream is None and self.overlap:
            self._comm_stream = quiet_side_stream(self._flat.device)   # (high priority: collectives ahead of queued compute)
            if self._comm_stream is None:
                import warnings
                warnings.warn('GradSync: no quiet side stream on this device (every candidate queue disturbs the compute stream): running SERIAL', RuntimeWarning)
                self.overlap, self.cu_reserve, self.why = False, 0, 'overlap refused: no quiet side stream'
        return self._comm_stream

    def bucket_event(self):
        """Raw hipEvent_t (as c_void_p) that libe3unet records when bucket A is complete; None on CPU and in serial mode (the library then
        launches everything for the whole chip)."""
        if self._flat is None or not self._flat.is_cuda or (self.world == 1 and not self.force) or not self.overlap:
            return None
        if self._side_stream() is None:      # (decided HERE, in front of the backward's launches: without the event the library launches for the whole chip)
            return None
        import ctypes
        if self._event is None:
            self._event = torch.cuda.Event()
            self._event.record()          # forces creation of the underlying hipEvent_t
        return ctypes.c_void_p(self._event.cuda_event)

    def after_backward(self, plan):
        if self.world == 1 and not self.force:
            return
        flat = self._flat
        a, b = flat[self._split:], flat[:self._split]
        if not flat.is_cuda:
            self._allreduce(a)
            self._allreduce(b)
            return
        cur = torch.cuda.current_stream(flat.device)
        if not self.overlap:
            # serial: ONE collective over the whole buffer on the compute stream itself, behind the backward -- no stream of ours is involved
            # (a wait parked on another hardware queue while the backward's ~100 small kernels run can cost ~50 us per kernel when the two
            # queues share a command-processor pipe, see quiet_side_stream)
            w = self._allreduce(flat, async_op=True)
            if w is not None:
                w.wait()
            return
        side = self._side_stream()
        works = []
        with torch.cuda.stream(side):
            side.wait_event(self._event)                   # bucket A's gradients are final
            works.append(self._allreduce(a, async_op=True))
            side.wait_stream(cur)                          # the whole backward has been enqueued on `cur`
            if b.numel():
                works.append(self._allreduce(b, async_op=True))
        for w in works:                                    # stream-level wait: `cur` resumes after the collectives
            if w is not None:
                w.wait()
        if any(w is None for w in works):                  # (non-RCCL backends: the average's division ran on the side stream)
            cur.wait_stream(self._comm_stream)
        flat.record_stream(self._comm_stream)

    # -- helpers -----------------------------------------------------------------------------------------------
    def _allreduce(self, t, async_op=False):
        if t.numel() == 0:
            return None
        if self.collective is not None:
            self.collective(t)
            return None
        backend = dist.get_backend(self.group)
        if self.average and backend == 'nccl':
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        if self.average:
            if w is not None:
                w.wait()
            t.div_(self.world)
        return None if self.average else w

    def wait(self):
        """Kept for symmetry with DDP-style loops: the waits are already stream-ordered in after_backward()."""
        return None

    def broadcast_parameters(self, src=0):
        """One-time replica initialisation (replaces DataParallel's per-iteration broadcast_coalesced)."""
        if self.world == 1:
            return
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            dist.broadcast(t.data, src=src, group=self.group)


_REJECTED_STREAMS = []      # kept alive: a candidate that was found noisy keeps its hardware queue, so the next candidate gets another one


_QUIET_FACTOR = 2.0         # a candidate passes when the burst beside its parked wait takes < 2 x the burst alone (a noisy queue: 4-5 x; tests lower it to force a refusal)
_QUIET_STREAMS = {}         # device index -> the side stream found for it (one probe per device and process, not per GradSync)


def quiet_side_stream(device, priority=-1, tries=8, verbose=False):
    """A side stream whose HARDWARE queue does not disturb the compute stream.

    Measured on MI355X (tools/probe_foreign_waves.py, profiles/r04_dp_probe.md): HIP maps streams onto a few hardware queues in order of first
    use; while a cross-stream wait (hipStreamWaitEvent) is parked on some of them, EVERY kernel of the compute stream takes ~50 us longer
    (the cfg-2 backward: 12 -> 18 ms) -- which ones depends on how many streams the process used before (every fourth or so).  So the
    candidate is tested the way GradSync uses it: a wait parked on it while a burst of tiny kernels runs on the current stream, timed against
    the same burst alone; a noisy candidate is set aside (kept alive) and the next stream is tried.  The verdict is cached per (device, priority):
    one probe per process.  None when no candidate passes (the caller stays serial)."""
    dkey = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), priority)
    if dkey in _QUIET_STREAMS and not verbose:
        return _QUIET_STREAMS[dkey]
    cur = torch.cuda.current_stream(device)
    x = torch.zeros(1 << 20, device=device)      # (kernels of ~1000 workgroups, ~5 us each, like the backward's elementwise passes)
    K = 64

    def burst(side):
        e0, e1, gate = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
        torch.cuda._sleep(6_000_000)                 # ~3 ms on the compute stream: the host gets ahead, so the wait below is parked while the burst runs
        e0.record(cur)
        for _ in range(K):
            x.add_(1.0)
        e1.record(cur)
        gate.record(cur)
        if side is not None:
            side.wait_event(gate)
            with torch.cuda.stream(side):
                x.mul_(1.0)
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1)

    with torch.cuda.device(device):
        burst(None)
        base = min(burst(None) for _ in range(3))
        found = None
        for _ in range(tries):
            cand = torch.cuda.Stream(device=device, priority=priority)
            t = min(burst(cand) for _ in range(2))
            if verbose:
                print(f'quiet_side_stream: candidate {cand.cuda_stream:#x}: burst {t:.3f} ms (alone {base:.3f} ms)')
            if t < _QUIET_FACTOR * base + 0.05:
                found = cand
                break
            _REJECTED_STREAMS.append(cand)
    _QUIET_STREAMS[dkey] = found      # (None: no candidate was quiet)
    return found


def shard_batch(batch, rank, world):
    """Rank r takes samples [r*B/world, (r+1)*B/world) of a global minibatch."""
    n = batch.shape[0]
    if n % world != 0:
        raise ValueError(f'global batch {n} is not divisible by world size {world}')
    per = n // world
    return batch[rank * per:(rank + 1) * per]
