"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI (torch.distributed backend "nccl").

Replaces nn.DataParallel (scripts/train.py:93-96).  Clips are independent, BatchNorm statistics stay per replica (as
DataParallel's own per-chunk statistics do), so the only exchange is the gradient mean: the discriminator slab (1 MB)
after its backward, the generator slab in three buckets that become final in backward order -- {out, gru} (22 MB),
{text encoder, speaker path} (up to 30 MB with the word embedding), {audio encoder} (0.3 MB).  Each bucket is a single
contiguous range of the flat gradient slab, all-reduced asynchronously on RCCL's stream while the rest of the backward
keeps the compute stream busy; the optimiser step waits on the outstanding work.  xGMI is point-to-point (7 links per
GPU): few large messages keep every link busy without per-tensor launch overhead.
"""
import os
import warnings

import torch
import torch.distributed as dist

# Hardware queues configure_environment() asks for unless GPU_MAX_HW_QUEUES is already set.  Round 5: 4 (the runtime's default) -- the audio
# encoder's second stream in the forward only pays while both compute streams share a queue (engine._Engine.audio_fork: 8 queues replay the
# two-branch graph at 8.6 ms), and on this software stack RCCL's stream does land on another queue than the compute stream's at 4 (the
# {out, gru} bucket's kernel overlaps the next segment by 27 us, profiles/r5_g_q4_overlap.txt; round 4 had seen it in line and used 8).
# Same box, one rank: plain 4.50 / 4 queues 4.81 / 8 queues 4.88 ms (profiles/r5_f_ddp.txt).  TG_DDP_HW_QUEUES=8 restores round 4's setting.
HW_QUEUES = int(os.environ.get("TG_DDP_HW_QUEUES", "4"))
# RCCL's stream created with high priority: measured WORSE (5.45 against 4.81 ms, profiles/r5_f_ddp.txt: a third queue class beside the two
# compute streams); kept as a switch for other stacks
HIGH_PRIORITY_STREAM = os.environ.get("TG_DDP_PRIO", "0") != "0"


def configure_environment():
    """Call BEFORE the first HIP call of the process (bench.py does; a training script should too).  ROCm maps HIP streams onto
    GPU_MAX_HW_QUEUES hardware queues (default 4) round-robin; with the handful of streams a process creates (capture, warm-up, RCCL's)
    the collective's stream regularly lands on the compute stream's queue, and a gradient bucket's RCCL kernel then runs IN LINE with the
    graph segment behind it instead of beside it.  Measured on one rank (tools/ddp_overlap_probe.sh, profiles/r4_l_ddp_overlap.txt): with 4
    queues every segment started ~10 us after the bucket's kernel had finished; with 8 it starts 35-40 us before.
    An explicitly exported GPU_MAX_HW_QUEUES is honoured (probes set it)."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(HW_QUEUES))
    if os.environ.get("TG_DDP_CAPTURE", "0") != "0":
        # collectives captured inside the iteration's hipGraph: torch's NCCL backend recycles the events of finished work through a cache, and
        # an event last recorded inside a capture may come back for an EAGER collective (bench.py's closing barrier), where the watchdog's
        # hipEventQuery on it aborts the process ("operation not permitted on an event last recorded in a capturing stream", rounds 3-4).
        # Without the cache every work object owns its events.
        os.environ.setdefault("TORCH_NCCL_CUDA_EVENT_CACHE", "0")


def process_group_options():
    """backend options for init_process_group("nccl", ...): with HIGH_PRIORITY_STREAM, RCCL's internal stream is created with high priority --
    the runtime keeps a separate hardware queue per priority level, so the collectives never share the compute stream's queue whatever the
    stream creation order.  None otherwise (or when this torch build has no such option)."""
    if not HIGH_PRIORITY_STREAM:
        return None
    try:
        opts = dist.ProcessGroupNCCL.Options()
        opts.is_high_priority_stream = True
        return opts
    except Exception:
        return None


class GradSync:
    def __init__(self, group=None, chunk_floats=8 * 1024 * 1024):
        assert dist.is_initialized(), "init_process_group first"
        if dist.get_backend(group) == "nccl" and "GPU_MAX_HW_QUEUES" not in os.environ and not HIGH_PRIORITY_STREAM:
            # (configure_environment() sets the variable -- to HW_QUEUES, 4 since round 5, where RCCL's stream was measured on its own queue; what
            # is warned about is a process that never called it and so never made the choice)
            warnings.warn("GPU_MAX_HW_QUEUES is unset: ddp.configure_environment() was not called before the first HIP call, so the hardware-queue "
                          f"count (ddp.HW_QUEUES = {HW_QUEUES}) and the RCCL event-cache setting it applies are the runtime's defaults; RCCL's stream "
                          "may then share a hardware queue with the compute stream and serialise every gradient bucket with the backward it is meant to overlap")
        self.group = group
        self.world = dist.get_world_size(group)
        self.chunk = int(chunk_floats)
        self.pending = []
        self._avg = dist.get_backend(group) == "nccl"

    # ---- primitive: mean all-reduce of one contiguous gradient range, asynchronous
    def _launch(self, flat):
        for s in range(0, flat.numel(), self.chunk):
            piece = flat[s:s + self.chunk]
            if self._avg:
                w = dist.all_reduce(piece, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
                self.pending.append((w, None))
            else:                                   # gloo (CPU tests): SUM then scale
                w = dist.all_reduce(piece, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                self.pending.append((w, piece))

    def wait(self):
        for w, piece in self.pending:
            w.wait()
            if piece is not None:
                piece.div_(self.world)
        self.pending = []

    @staticmethod
    def bucket_range(slab, prefixes):
        """Contiguous [start, end) of the slab covered by the parameters under the given top-level names (cached on the slab: the replay
        loop of a segmented step asks for it between two graph launches; a slab's layout is fixed at construction)."""
        cache = slab.__dict__.setdefault("_bucket_ranges", {})
        key = tuple(prefixes)
        if key not in cache:
            cache[key] = GradSync._bucket_range(slab, prefixes)
        return cache[key]

    @staticmethod
    def _bucket_range(slab, prefixes):
        lo, hi = None, None
        live = [(n, p, o) for n, p, o in zip(slab.names, slab.params, slab.offsets) if n not in slab.frozen]   # frozen: no gradient
        for name, p, off in live:
            if name.split(".", 1)[0] in prefixes:
                lo = off if lo is None else min(lo, off)
                hi = off + p.numel() if hi is None else max(hi, off + p.numel())
        assert lo is not None, prefixes
        for name, p, off in live:                                             # the range must not swallow other groups
            assert not (lo <= off < hi) or name.split(".", 1)[0] in prefixes, (name, prefixes)
        return lo, hi

    # ---- actions issued by the trainer at its synchronisation points
    def run(self, action):
        kind = action[0]
        if kind == "all":                      # whole (trainable) slab, finished before returning control to the optimiser
            self._launch(action[1].grad[:action[1].n_train])
            self.wait()
        elif kind == "bucket":                 # one backward-order bucket, left in flight
            lo, hi = self.bucket_range(action[1], action[2])
            self._launch(action[1].grad[lo:hi])
        elif kind == "bucket_wait":            # the LAST bucket of a backward: launched and awaited in one action (one graph cut instead of two)
            lo, hi = self.bucket_range(action[1], action[2])
            self._launch(action[1].grad[lo:hi])
            self.wait()
        elif kind == "wait":
            self.wait()
        else:
            raise ValueError(kind)


def broadcast_parameters(slabs, src=0, group=None):
    """Make every replica start from rank `src`'s weights (what DataParallel's per-forward replicate guarantees)."""
    for s in slabs:
        dist.broadcast(s.flat, src=src, group=group)
