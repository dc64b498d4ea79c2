"""Data parallelism the MI355X way: one process per GPU, replicas hold identical weights, the flat trainable-gradient
buffer is averaged over xGMI by RCCL all-reduces (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" on CPU) that
are launched bucket by bucket while backward is still running.

Replaces torch.nn.DataParallel at il_modules/base.py:68, il_modules/mrn.py:106,133 (single process: per-iteration
broadcast of ALL parameters -- 1.26 GB for TRBA+MRN-6 -- scatter, gather, reduce on GPU 0).  Frozen experts need no
per-step traffic at all; BatchNorm statistics stay per replica exactly as under DataParallel; parameters AND buffers are
broadcast from rank 0 once, when a learner builds or grows its model.
"""
import os
import sys

import torch
import torch.distributed as dist
import torch.nn as nn


class ReplicaDataParallel(nn.Module):
    """Keeps the reference's `self.model.module` access pattern and the `module.` state_dict key prefix."""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        ops = sys.modules.get(__package__ + ".ops")       # (not imported by the CPU-only users of this module: nothing to count then)
        if ops is not None and torch.is_grad_enabled():
            # parameter uses of this forward are counted in a table of their own (ops.note_param_uses): graphs whose backward never
            # runs under ops.direct_gradients (Fisher passes, grad-enabled validation) leave nothing behind in a later step's counts
            ops.new_use_generation()
        return self.module(*args, **kwargs)


def init_distributed(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world, local_rank).  Single-process runs need no initialisation."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank_ = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("MRN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if os.environ.get("MRN_SHARE_DEVICE"):      # test hook: every rank on cuda:0 (gloo only) -- exercises the N > 1
            local = 0                               # control flow of bench.py / the learners on a one-GPU box
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank_, world_size=world)
    return rank_, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


# MRN_COMM=native: the gradient buckets go through the library's own RCCL entry points (include/mrn_hip.h: mrn_comm_init /
# mrn_allreduce_f32, the path a non-PyTorch host uses) on a dedicated HIP stream; torch.distributed then only carries the 128-byte
# rendezvous id.  Default: torch.distributed's RCCL binding.
NATIVE_COMM = os.environ.get("MRN_COMM") == "native"
_native = {"stream": None, "world": 0}


def init_native_comm(rank_=None, world=None):
    """create the library's RCCL communicator for this process (idempotent); the id travels through torch.distributed when there
    is more than one rank"""
    import ctypes
    from ._lib import call
    if _native["world"]:
        return _native["world"]
    rank_ = rank() if rank_ is None else rank_
    world = world_size() if world is None else world
    n = int(call("mrn_comm_unique_id_bytes"))
    buf = ctypes.create_string_buffer(n)
    if rank_ == 0:
        call("mrn_comm_unique_id", buf)
    if world > 1:
        box = [buf.raw]
        dist.broadcast_object_list(box, src=0)
        buf = ctypes.create_string_buffer(box[0], n)
    call("mrn_comm_init", rank_, world, buf)
    _native["stream"] = torch.cuda.Stream()
    _native["world"] = world
    return world


class _NativeWork:
    """handle of a collective issued on the comm stream: wait() orders the current stream behind it"""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


def native_all_reduce(t, average=True):
    """in place over the library's communicator, asynchronous: the comm stream waits for the producer stream, the returned handle
    makes the consumer stream wait for the collective"""
    from ._lib import call
    st = _native["stream"]
    st.wait_stream(torch.cuda.current_stream())
    call("mrn_allreduce_f32", t.data_ptr(), t.numel(), int(average), st.cuda_stream)
    t.record_stream(st)
    ev = torch.cuda.Event()
    ev.record(st)
    return _NativeWork(ev)


def native_broadcast(t, src=0):
    from ._lib import call
    st = _native["stream"]
    st.wait_stream(torch.cuda.current_stream())
    call("mrn_broadcast_f32", t.data_ptr(), t.numel(), int(src), st.cuda_stream)
    t.record_stream(st)
    torch.cuda.current_stream().wait_stream(st)
    return t


def _avg_inplace(t, async_op=False):
    """all-reduce average of t over the ranks; returns the work handle when async_op"""
    if NATIVE_COMM and t.is_cuda:
        init_native_comm()
        work = native_all_reduce(t, average=True)
        if async_op:
            return work
        work.wait()
        return None
    if dist.get_backend() == "nccl":
        return dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=async_op)
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=async_op)      # gloo has no AVG: the caller divides


def average_gradients(flat_grad):
    """One blocking collective over the flat gradient buffer."""
    if world_size() == 1:
        return flat_grad
    _avg_inplace(flat_grad)
    if not _averages_in_collective(flat_grad):
        flat_grad.div_(world_size())
    return flat_grad


def _averages_in_collective(t):
    return (NATIVE_COMM and t.is_cuda) or dist.get_backend() == "nccl"


def broadcast_parameters(flat_param, src=0, params=None):
    """the optimiser's flat parameter buffer from rank `src`; `params`: the parameters viewing it (their version counters are
    bumped so that repacked-weight caches notice)"""
    if world_size() > 1:
        dist.broadcast(flat_param, src=src)
        if params:
            torch.autograd.graph.increment_version(list(params))
    return flat_param


def broadcast_module(module, src=0):
    """Every parameter and buffer of `module` from rank `src` (frozen experts, BatchNorm running statistics and counters
    included), coalesced into one collective per dtype.  DataParallel does this every iteration (replicate); here it runs when
    a learner builds or grows its model, after which only gradients travel."""
    if world_size() == 1:
        return
    seen, groups = set(), {}
    for t in list(module.parameters()) + list(module.buffers()):
        if t.data_ptr() in seen or t.numel() == 0:       # aliased tensors (fc / Prediction.generator) travel once
            continue
        seen.add(t.data_ptr())
        groups.setdefault(t.dtype, []).append(t.data)
    for dtype, tensors in groups.items():
        flat = torch.cat([t.reshape(-1) for t in tensors])
        dist.broadcast(flat, src=src)
        off = 0
        for t in tensors:
            t.copy_(flat[off:off + t.numel()].view(t.shape))
            off += t.numel()
    torch.autograd.graph.increment_version(list(module.parameters()))      # (.data copies do not bump the parameters' versions)


def global_mean_weight(n_local):
    """DataParallel computes a mean loss over the GATHERED batch; a replica computes it over its own shard and the gradients are
    averaged afterwards.  The two agree exactly when every shard has the same number of contributing items -- not so for
    CrossEntropyLoss(ignore_index=[PAD]), whose valid-target count varies with the label lengths.  Returns the factor
    n_local * world / sum_over_ranks(n_local) that turns this rank's shard mean into its share of the global mean (None: one rank)."""
    if world_size() == 1:
        return None
    n = n_local.detach().to(torch.float32).reshape(1)
    total = n.clone()
    dist.all_reduce(total)                                    # SUM (one scalar per step)
    return n * world_size() / total


def barrier():
    if world_size() > 1:
        dist.barrier()


class BucketedAllReduce:
    """Gradient all-reduce overlapped with backward: the optimiser's flat gradient buffer is cut into contiguous buckets,
    filled from the END of the buffer (parameters are laid out in forward order, backward produces their gradients roughly in
    reverse); a post-accumulate hook on every parameter counts its bucket down and the bucket's all-reduce is launched
    asynchronously as soon as it is complete -- in bucket order on every rank, so the collectives match up.  finish()
    launches whatever is left (parameters that got no gradient this step keep their zeros) and waits.

    Sizes this replaces: loop A 25 MB (SVTR) / 35 MB (CRNN) / 201 MB (TRBA) / 231 MB (DER-6) of fp32
    gradients per step; loop B (router only) 2-11 MB = one bucket.  Ring all-reduce over xGMI is per-link bound (~153 GB/s),
    so 25 MB buckets take ~0.3 ms each and hide behind the remaining backward kernels."""

    def __init__(self, optimizer, bucket_bytes=None):
        self.opt = optimizer
        bucket_bytes = bucket_bytes or int(os.environ.get("MRN_BUCKET_MB", "25")) * (1 << 20)
        cap = max(bucket_bytes // 4, 1)
        n = optimizer.grad.numel()
        ends = [off + (p.numel() + 3) // 4 * 4 for p, off in zip(optimizer.params, optimizer.offsets)]
        # walk the parameters backwards, closing a bucket when it reaches the capacity
        self.buckets = []          # (start, end) element ranges, bucket 0 = the tail of the buffer
        self.bucket_of = [0] * len(optimizer.params)
        hi = n
        members = []
        for i in range(len(optimizer.params) - 1, -1, -1):
            members.append(i)
            lo = optimizer.offsets[i]
            if hi - lo >= cap or i == 0:
                for m in members:
                    self.bucket_of[m] = len(self.buckets)
                self.buckets.append((lo, hi))
                hi, members = lo, []
        assert self.buckets[-1][0] == 0 and ends[-1] <= n
        self.sizes = [0] * len(self.buckets)
        for b in self.bucket_of:
            self.sizes[b] += 1
        self.active = False
        self.index_of = {id(p): i for i, p in enumerate(optimizer.params)}
        self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(i)) for i, p in enumerate(optimizer.params)]
        self.launched_log = []     # bucket indices in launch order of the last step (tests)
        self.record = False        # bench.py telemetry: HIP events around the part of finish() the compute stream has to wait for
        self.exposed = []          # [(event, event)] per step while record is set

    def close(self):
        """detach from the parameters (a learner builds a new optimiser -- and a new reducer -- for every task / step)"""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        self.active = False

    def _make_hook(self, i):
        def hook(_param):
            self._report(i)
        return hook

    def _report(self, i):
        """count parameter i's bucket down ONCE per step, whoever reports it (autograd's hook, the side stream's notify, or both)"""
        if self.active and i not in self.reported:
            self.reported.add(i)
            self.pending[self.bucket_of[i]] -= 1
            self._launch_ready()

    def param_ready(self, param):
        """ops.direct_gradients(notify=...): a parameter whose gradient never passes autograd's accumulation -- it is ADDED into the
        flat gradient on the side stream (functional.side_param_grads, ConvBlockFn.backward) -- has had its last accumulation issued"""
        i = self.index_of.get(id(param))
        if i is not None:
            self._report(i)

    def begin(self):
        self.pending = list(self.sizes)
        self.reported = set()
        self.next = 0
        self.handles = []
        self.launched_log = []
        self.active = True

    def _launch(self, b):
        lo, hi = self.buckets[b]
        side = None
        if self.opt.grad.is_cuda:
            from . import ops
            if ops.side_stream_pending():
                side = ops.side_stream()
        if side is None:
            self.handles.append(_avg_inplace(self.opt.grad[lo:hi], async_op=True))
        else:
            # part of this bucket was accumulated on the side stream: issue the collective from there, behind those kernels AND
            # behind what the main stream has produced so far (the collective is ordered after the stream it is launched from)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self.handles.append(_avg_inplace(self.opt.grad[lo:hi], async_op=True))
        self.launched_log.append(b)

    def _launch_ready(self):
        while self.next < len(self.buckets) and self.pending[self.next] <= 0:
            self._launch(self.next)
            self.next += 1

    def finish(self):
        self.active = False
        e0 = None
        if self.record and self.opt.grad.is_cuda:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        while self.next < len(self.buckets):
            self._launch(self.next)
            self.next += 1
        for h in self.handles:
            h.wait()
        if not _averages_in_collective(self.opt.grad):
            self.opt.grad.div_(world_size())
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.exposed.append((e0, e1))
        self.handles = []

    def bytes_per_step(self):
        return 4 * sum(hi - lo for lo, hi in self.buckets)

    def standalone_ms(self, reps=3):
        """the same buckets all-reduced back to back with nothing else on the GPU (after a synchronize): what one step's gradient
        exchange costs when none of it is hidden behind backward"""
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        keep = self.opt.grad.clone()
        e0.record()
        for _ in range(reps):
            hs = [_avg_inplace(self.opt.grad[lo:hi], async_op=True) for lo, hi in self.buckets]
            for h in hs:
                h.wait()
        e1.record()
        torch.cuda.synchronize()
        self.opt.grad.copy_(keep)
        return e0.elapsed_time(e1) / reps
