"""Multi-GPU data parallelism for independent items.

Every (scalar, point) pair / signature is independent, so the batch is cut into
contiguous shards, one per rank (= one process per GPU); tables and constants
are replicated per device by each rank's own Context.  There is no exchange
inside the computation; the ONLY collective is the final gather of the result
bytes (RCCL all_gather over xGMI with backend "nccl"; "gloo" in the CPU tests).
"""
import numpy as np


def shard_range(n, rank, world):
    """contiguous slice [lo, hi) of an n-item batch owned by `rank`"""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def gather_results(local, n, dist=None):
    """all ranks contribute their shard's result rows (torch tensor, first dim =
    shard length); returns the concatenated n-row tensor on every rank.  Shards
    may differ by one row, so shorter ones are padded for the collective."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    longest = (n + world - 1) // world
    pad = longest - local.shape[0]
    buf = local
    if pad:
        buf = torch.cat([local, torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype,
                                            device=local.device)])
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = []
    for r in range(world):
        lo, hi = shard_range(n, r, world)
        out.append(parts[r][: hi - lo])
    return torch.cat(out)


class ShardedVerifier:
    """ECDSA verify of a global batch across the ranks of a process group.

    Each rank passes the SAME global arrays (or only needs its own slice to be
    valid) and gets the full ok-mask back."""

    def __init__(self, ctx, curve, dist=None, device=None):
        self.ctx = ctx
        self.curve = curve
        self.dist = dist
        self.device = device

    def verify(self, hashes, r, s, pub, msg_bits=0):
        import torch
        n = hashes.shape[0]
        rank = self.dist.get_rank() if self.dist is not None and self.dist.is_initialized() else 0
        world = self.dist.get_world_size() if self.dist is not None and self.dist.is_initialized() else 1
        lo, hi = shard_range(n, rank, world)
        if self.device is not None and self.device.type == "cuda":
            # device-resident global arrays (torch CUDA tensors) are sliced in place; host arrays
            # are copied once.  No host synchronisation: the kernels run on torch's current stream
            # of this device, which the collective is ordered after.
            sl = [x[lo:hi].contiguous() if torch.is_tensor(x) and x.is_cuda
                  else torch.from_numpy(np.array(x[lo:hi], copy=True)).to(self.device, non_blocking=True)
                  for x in (hashes, r, s, pub)]
            ok = torch.zeros(hi - lo, dtype=torch.uint8, device=self.device)
            if hi > lo:
                self.ctx.ecdsa_verify_dev(self.curve, sl[0], sl[1], sl[2], sl[3], ok, msg_bits)
        else:
            if hi > lo:
                ok_np = self.ctx.ecdsa_verify(self.curve, hashes[lo:hi], r[lo:hi], s[lo:hi], pub[lo:hi], msg_bits)
            else:
                ok_np = np.zeros(0, np.uint8)
            ok = torch.from_numpy(np.ascontiguousarray(ok_np))
        return gather_results(ok, n, self.dist)


class ShardedMul:
    """Scalar multiplication of a global batch across the ranks of a process group: rank r
    computes its contiguous slice, the affine results (n x 2B bytes) and the infinity flags are
    gathered on every rank -- SURVEY 8e's point-output gather (N/G x 64 B per rank on the
    256-bit curves).  points = None multiplies the generator (fixed-base comb)."""

    def __init__(self, ctx, curve, dist=None, device=None):
        self.ctx = ctx
        self.curve = curve
        self.dist = dist
        self.device = device

    def mul(self, scalars, points=None):
        import torch
        from . import FIELD_BYTES
        B = FIELD_BYTES[self.curve]
        n = scalars.shape[0]
        on = self.dist is not None and self.dist.is_initialized()
        rank = self.dist.get_rank() if on else 0
        world = self.dist.get_world_size() if on else 1
        lo, hi = shard_range(n, rank, world)
        if self.device is not None and self.device.type == "cuda":
            def dev(x):
                return (x[lo:hi].contiguous() if torch.is_tensor(x) and x.is_cuda
                        else torch.from_numpy(np.array(x[lo:hi], copy=True)).to(self.device, non_blocking=True))
            k = dev(scalars)
            xy = torch.zeros((hi - lo, 2 * B), dtype=torch.uint8, device=self.device)
            inf = torch.zeros(hi - lo, dtype=torch.uint8, device=self.device)
            if hi > lo:
                if points is None:
                    self.ctx.mul_fixed_dev(self.curve, k, xy, inf)
                else:
                    self.ctx.mul_var_dev(self.curve, k, dev(points), xy, inf)
        else:
            if hi > lo:
                if points is None:
                    xy_np, inf_np = self.ctx.mul_fixed(self.curve, scalars[lo:hi])
                else:
                    xy_np, inf_np = self.ctx.mul_var(self.curve, scalars[lo:hi], points[lo:hi])
            else:
                xy_np, inf_np = np.zeros((0, 2 * B), np.uint8), np.zeros(0, np.uint8)
            xy = torch.from_numpy(np.ascontiguousarray(xy_np))
            inf = torch.from_numpy(np.ascontiguousarray(inf_np))
        return gather_results(xy, n, self.dist), gather_results(inf, n, self.dist)


class OverlappedGather:
    """The final gather off the compute stream, for a loop of batches (bench.py's strong-scaling
    steps, a service that verifies block after block): step i's all_gather runs on the
    collective's own stream WHILE step i + 1 computes.  Two result buffers alternate;
    `async_op=True` orders the collective after the work already on the current stream, and
    nothing on that stream waits for it until its buffer comes round again.

        og = OverlappedGather(n, dist, device)            # n = rows of the GLOBAL batch
        for step in ...:
            out = og.begin()                              # this rank's shard rows to write (waits for the gather that last used them)
            ctx.ecdsa_verify_dev(..., out)
            og.submit()                                   # async all_gather of that buffer
        og.drain()
        mask = og.result(b)                               # the n-row gathered result of buffer b (0 / 1)

    gather_device: where the collective runs (the compute device for backend "nccl" = RCCL; CPU
    for "gloo", in which case submit() copies the shard to the host first)."""

    def __init__(self, n, dist, device, gather_device=None, dtype=None):
        import torch
        self.n, self.dist = n, dist
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        self.lo, self.hi = shard_range(n, self.rank, self.world)
        self.longest = (n + self.world - 1) // self.world
        self.gdev = gather_device if gather_device is not None else device
        dtype = dtype or torch.uint8
        self.local = [torch.zeros(self.longest, dtype=dtype, device=device) for _ in range(2)]
        self.full = [torch.zeros(self.world * self.longest, dtype=dtype, device=self.gdev) for _ in range(2)]
        self.pending = [None, None]
        self.turn = 0
        self.cur = None
        # The zero fills above were queued on the stream that is current HERE; the caller's steps
        # may run on other (non-blocking) streams, which do not order themselves after it -- a
        # fill that lands after the first step's kernels wipes that step's shard.  Seen with eight
        # processes sharing one device (3 of 16 runs); one device-wide wait at construction ends it.
        if torch.device(device).type == "cuda":
            torch.cuda.synchronize(device)

    def begin(self):
        b = self.turn
        if self.pending[b] is not None:                   # the gather that last read local[b] / wrote full[b]
            self.pending[b].wait()
            self.pending[b] = None
        self.cur = b
        return self.local[b][: self.hi - self.lo]

    def submit(self):
        b = self.cur
        src = self.local[b] if self.local[b].device == self.full[b].device else self.local[b].to(self.gdev)
        self.pending[b] = self.dist.all_gather_into_tensor(self.full[b], src, async_op=True)
        self.turn ^= 1

    def drain(self):
        for b in (0, 1):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None

    def result(self, b):
        """the n-row gathered result of buffer b: every rank's shard without its padding"""
        import torch
        parts = []
        for r in range(self.world):
            lo, hi = shard_range(self.n, r, self.world)
            parts.append(self.full[b][r * self.longest: r * self.longest + (hi - lo)])
        return torch.cat(parts)
