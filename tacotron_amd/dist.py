"""Data-parallel gradient exchange (new work: the reference is single-device).

One process per GPU; each rank runs forward+backward on its own minibatch; ONE exchange step: SUM all-reduce of the
flat fp32 gradient buffer (loss is a plain sum, so SUM -- not mean -- reproduces the N*B-batch gradient exactly), then
every rank applies the same clip + Adam update, so replicas stay identical.  Backend "nccl" is RCCL on ROCm (xGMI);
"gloo" is used by the CPU tests.  The buffer is reduced in a few large buckets issued back-to-back (xGMI rings are
per-link bound, so few large messages beat many small ones): 10 MB buckets make each of the five segments ONE collective."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, force=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).  Returns (rank, world, local_rank).
    force (default: TACO_FORCE_DIST=1): create the process group at world size 1 too, so that the whole distributed path
    (RCCL init, communication stream, segment events, collectives) executes on a single GPU.
    The RCCL process group gets HIGH-PRIORITY streams: its kernels are enqueued while the main stream still holds a
    millisecond of grid-filling GEMM launches, and a normal-priority stream is served behind them (round 3 measured the
    post-net segment's collective finishing 1.46 ms after it became eligible at world size 1, with nothing to move)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if force is None:
        force = os.environ.get('TACO_FORCE_DIST', '0') not in ('', '0')
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            if _comm_priority() < 0:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                kw['pg_options'] = opts
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def _comm_priority():
    """-1 (high) for the communication stream and RCCL's streams, unless TACO_COMM_PRIORITY=0 -- or the process runs with fewer
    than 8 hardware queues: a high-priority stream then displaces one of the step's own streams onto a shared queue, which was
    measured to cost 1.3 ms per step (round 4, GPU_MAX_HW_QUEUES=4), far more than the priority buys.  The variable only counts if it was
    in the environment when the HIP runtime initialised (lib.HW_QUEUES_LATE)."""
    if os.environ.get('TACO_COMM_PRIORITY', '1') in ('', '0'):
        return 0
    from . import lib
    if lib.HW_QUEUES_LATE:
        import warnings
        warnings.warn('tacotron_amd was imported after the HIP runtime had initialised: GPU_MAX_HW_QUEUES=8 did not take effect, the '
                      'communication streams stay at normal priority (import tacotron_amd -- or set the variable -- before the first '
                      'torch.cuda call; INTEGRATION.md 4)')
    return -1 if lib.effective_hw_queues() >= 8 else 0


SEGMENT_NAMES = {4: 'post-net', 3: 'decoder', 2: 'encoder projections+highways+bi-GRU', 1: 'encoder conv bank', 0: 'embedding+encoder pre_net'}


class GradReducer:
    """SUM all-reduce of the flat gradient buffer, overlapped with the backward pass.

    `taco_backward` finalises the buffer in five contiguous segments (`lib.grad_segments`: post-net 7.3 MB, decoder 6.4 MB,
    encoder without its conv bank 4.7 MB, encoder conv bank 8.9 MB, embedding + encoder pre_net 0.5 MB) and records a HIP event
    per segment.
    `reduce_after_backward` enqueues, on a high-priority communication stream, a device-side wait for each event followed by
    that segment's bucketed all-reduce, so the bytes travel while the rest of the backward pass runs and the host never
    blocks.

    No collective ever runs beside the decoder BPTT: that kernel is a persistent launch whose 256 workgroups must all be
    co-resident (one per CU, ~110 KB of LDS each); an RCCL kernel dispatched first would hold CUs that part of every cluster
    needs, and that part's peers would spin until it gets them.  The library therefore announces the post-net segment only
    AFTER the BPTT kernel: segments 4 and 3 (13.7 MB) reduce under the encoder backward (1.5 ms of ordinary kernels),
    segment 2 under the encoder's conv-bank gradients (the last 0.6 ms), segment 1 (the conv bank itself, final behind its
    weight-gradient launch; round 6) under the step's tail (~0.1 ms), and only segment 0 (0.5 MB) is exposed in full.  (Rounds 2-3 had an
    opt-in mode that reduced the post-net segment underneath the older, slower BPTT kernel; it cost more than it hid and was
    removed in round 4.)

    The loss triple rides along (SUM); the decoder error words ride along with MAX -- they are sticky 0/1 flags, and a SUM
    would multiply a set flag by the world size every step until the int32 wraps to 0 (after 32/log2(W) steps)."""

    def __init__(self, bucket_floats=5 * 512 * 1024, group=None, force=False):
        self.bucket = int(bucket_floats)
        self.group = group
        on = dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.active = self.world > 1 or (force and on)
        self.comm = None
        self.comm_priority = _comm_priority()
        self.timing = False          # True: record events around every segment's collectives (bench.py `allreduce` object)
        self._events = []

    def _buckets(self, t, lo, hi):
        for off in range(lo, hi, self.bucket):
            yield t[off:min(hi, off + self.bucket)]

    def _flags(self, works, loss, err):
        if loss is not None:
            works.append(dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        if err is not None:
            works.append(dist.all_reduce(err, op=dist.ReduceOp.MAX, group=self.group, async_op=True))

    def all_reduce(self, grads: torch.Tensor, loss: torch.Tensor | None = None, err: torch.Tensor | None = None):
        """Plain form: everything after backward has finished (also what the CPU tests drive)."""
        if not self.active:
            return
        works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for b in self._buckets(grads, 0, grads.numel())]
        self._flags(works, loss, err)
        for w in works:
            w.wait()

    def reduce_after_backward(self, model):
        """Call right after `model.backward()` has been ENQUEUED (it need not have run yet)."""
        if not self.active:
            return
        grads = model.grads
        if not grads.is_cuda:
            return self.all_reduce(grads, model._loss, model._err)
        from . import lib
        if self.comm is None:
            self.comm = torch.cuda.Stream(priority=self.comm_priority)
        bounds = lib.grad_segments(model.shape)
        nseg = len(bounds) - 1
        works = []
        evs = []
        with torch.cuda.stream(self.comm):
            for seg in range(nseg - 1, -1, -1):         # completion order inside taco_backward
                lib.wait_grad_segment(seg, self.comm)   # device-side: the collectives below start when the segment is final
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self.comm)
                for b in self._buckets(grads, bounds[seg], bounds[seg + 1]):
                    works.append(dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
                if seg == nseg - 1:
                    # the riders travel with the FIRST segment: the loss is final since the forward pass and the error words
                    # since the BPTT kernel, which this segment is announced behind -- nothing small is left for the exposed tail
                    self._flags(works, model._loss, model._err)
                if self.timing:
                    for w in works:
                        w.wait()                        # (comm stream waits for the backend's stream; device side)
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(self.comm)
                    evs.append((seg, e0, e1))
        for w in works:
            w.wait()                                    # the CURRENT stream waits for the collective (no host block with RCCL)
        torch.cuda.current_stream().wait_stream(self.comm)
        if self.timing:
            self._events.append(evs)

    def describe(self, model):
        """Static facts of the exchange for reports: bytes and bucket count per segment, in completion order."""
        from . import lib
        b = lib.grad_segments(model.shape)
        return [{'segment': SEGMENT_NAMES.get(s, str(s)), 'bytes': 4 * (b[s + 1] - b[s]), 'buckets': -(-(b[s + 1] - b[s]) // self.bucket)}
                for s in range(len(b) - 2, -1, -1)]

    def segment_times_us(self):
        """Median microseconds (segment final -> its collectives done) per segment over the recorded steps (timing=True);
        host synchronisation."""
        if not self._events:
            return None
        torch.cuda.synchronize()
        out = {}
        for seg in sorted({s for evs in self._events for s, _, _ in evs}, reverse=True):
            xs = sorted(e0.elapsed_time(e1) * 1e3 for evs in self._events for s, e0, e1 in evs if s == seg)
            out[SEGMENT_NAMES.get(seg, str(seg))] = xs[len(xs) // 2]
        self._events = []
        return out
