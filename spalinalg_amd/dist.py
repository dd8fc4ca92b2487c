"""Row-partitioned y = A*x over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" in the CPU tests).  Rows are cut into contiguous ranges with
balanced stored entries; every rank owns its range's CSR arrays and a full
copy of x.  The exchange steps are exactly two:

  broadcast_x : x from rank 0 to everyone          (once per x)
  spmv        : local kernel on the rank's rows, then an all-gather of the y
                slices so every rank ends with the complete y (= the next x of
                an iterative method)

There is no data-path collective inside the local product.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from ._ffi import check


def partition_rows(rowptr: np.ndarray, nparts: int) -> np.ndarray:
    """nnz-balanced contiguous row ranges (library: spal_partition_rows)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
    bounds = np.empty(nparts + 1, dtype=np.uint64)
    check(_ffi.lib().spal_partition_rows(rowptr.ctypes.data_as(_ffi.u64p), C.c_uint64(rowptr.size - 1),
                                         C.c_uint32(nparts), bounds.ctypes.data_as(_ffi.u64p)))
    return bounds.astype(np.int64)


def even_rows(nrows: int, nparts: int) -> np.ndarray:
    """row ranges for a matrix with the same number of entries in every row"""
    return np.array([(nrows * g) // nparts for g in range(nparts + 1)], dtype=np.int64)


class RowPartitionedSpmv:
    """The rank-local piece of a row-partitioned product.

    local_spmv(x_full, out_local) computes this rank's rows; in the product it
    is `DeviceCsr.spmv_torch` of the rank's shard (HIP kernel).  The CPU tests
    inject the oracle here to exercise the partition / collective logic with
    gloo -- the product itself never computes on the host.
    """

    def __init__(self, local_spmv, bounds, rank: int, world: int, dtype, device, group=None):
        import torch
        self.torch = torch
        self.local_spmv = local_spmv
        self.bounds = np.asarray(bounds, dtype=np.int64)
        assert self.bounds.size == world + 1
        self.rank, self.world = rank, world
        self.group = group
        self.nrows = int(self.bounds[-1])
        self.r0, self.r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
        sizes = np.diff(self.bounds)
        self.max_rows = int(sizes.max())
        self.equal = bool(np.all(sizes == sizes[0]))
        self.y_local = torch.empty(self.max_rows, dtype=dtype, device=device)
        self._gather = None if self.equal else torch.empty(world * self.max_rows, dtype=dtype, device=device)

    @classmethod
    def from_shard(cls, shard_dev, bounds, rank, world, device, group=None):
        """shard_dev: DeviceCsr of rows [bounds[rank], bounds[rank+1])."""
        import torch
        tdt = torch.float64 if shard_dev.dtype == np.float64 else torch.float32
        nloc = int(bounds[rank + 1] - bounds[rank])

        def local(x_full, out_local):
            shard_dev.spmv_torch(x_full, out=out_local[:nloc])

        return cls(local, bounds, rank, world, tdt, device, group)

    def broadcast_x(self, x):
        """x: full-length vector on every rank; rank 0's content wins."""
        if self.world > 1:
            self.torch.distributed.broadcast(x, src=0, group=self.group)
        return x

    def local_only(self, x):
        self.local_spmv(x, self.y_local)
        self._y_in = None
        return self.y_local[: self.r1 - self.r0]

    # ---- "x broadcast once ... y slices gathered at the end", without the excess ---------
    # A rank's rows reference only the columns [need_lo, need_hi): for banded shards its own
    # slice plus W/2 either side.  Rank 0 therefore SCATTERS x -- every rank receives the
    # window it reads (equal lengths, clamped into the vector), 1/N of a broadcast's bytes per
    # link -- and the y slices are GATHERED on rank 0 (1/N of an all-gather's bytes per rank).
    def plan_x_windows(self, need_lo: int, need_hi: int):
        """Collective: agree on one window length and every rank's window start."""
        torch, dist = self.torch, self.torch.distributed
        dev = self.y_local.device
        mine = torch.tensor([need_lo, need_hi], dtype=torch.int64, device=dev)
        allneeds = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(allneeds, mine, group=self.group)
        else:
            allneeds = [mine]
        needs = [(int(t[0]), int(t[1])) for t in allneeds]
        self.xw_len = max(hi - lo for lo, hi in needs)
        return needs

    def distribute_x(self, x, ncols: int, needs):
        """x (full length on every rank; rank 0 holds the content): every rank's window
        [start, start + xw_len) is filled from rank 0.  Returns this rank's (start, stop)."""
        dist = self.torch.distributed
        L = min(self.xw_len, ncols)
        starts = [max(0, min(lo, ncols - L)) for lo, _ in needs]
        me = starts[self.rank]
        if self.world > 1:
            if self.rank == 0:
                if getattr(self, "_x_self", None) is None or self._x_self.numel() != L:
                    self._x_self = self.torch.empty(L, dtype=x.dtype, device=x.device)
                dist.scatter(self._x_self, [x[s:s + L] for s in starts], src=0, group=self.group)
            else:
                dist.scatter(x[me:me + L], None, src=0, group=self.group)
        return me, me + L

    def gather_y_root(self, y_full):
        """The slices of the last product (self.y_local) land in y_full on rank 0 only.
        Needs equal slices (else: gather_y)."""
        dist = self.torch.distributed
        if not self.equal:
            raise ValueError("gather_y_root needs equal row slices")
        n = self.r1 - self.r0
        if self.world == 1:
            y_full[:n].copy_(self.y_local[:n])
            return y_full
        if self.rank == 0:
            views = [y_full[int(self.bounds[g]):int(self.bounds[g + 1])] for g in range(self.world)]
            dist.gather(self.y_local[:n], views, dst=0, group=self.group)
        else:
            dist.gather(self.y_local[:n], None, dst=0, group=self.group)
        return y_full

    # ---- halo exchange -------------------------------------
    # For a square matrix used iteratively (y is the next x) a rank does not need
    # the whole vector, only the columns its rows reference: [need_lo, need_hi).
    # For banded matrices that is its own slice plus a halo of W/2 entries on
    # either side, so the per-step exchange shrinks from an all-gather of the
    # whole vector to two small neighbour messages.
    def plan_halo(self, need_lo: int, need_hi: int):
        """need_lo/need_hi: smallest / one past the largest column referenced by
        this rank's rows.  Collective: every rank learns every rank's needs."""
        torch, dist = self.torch, self.torch.distributed
        dev = self.y_local.device
        mine = torch.tensor([need_lo, need_hi], dtype=torch.int64, device=dev)
        allneeds = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(self.world)]
        if self.world > 1:
            dist.all_gather(allneeds, mine, group=self.group)
        else:
            allneeds = [mine]
        needs = [(int(t[0]), int(t[1])) for t in allneeds]
        own = [(int(self.bounds[g]), int(self.bounds[g + 1])) for g in range(self.world)]

        def cut(a, b):  # intersection of two half-open ranges
            lo, hi = max(a[0], b[0]), min(a[1], b[1])
            return (lo, hi) if lo < hi else None

        # what I receive from g: my need inside g's slice; what I send to g: g's need inside mine
        self.halo_recv = [(g, *c) for g in range(self.world) if g != self.rank
                          for c in [cut(needs[self.rank], own[g])] if c]
        self.halo_send = [(g, *c) for g in range(self.world) if g != self.rank
                          for c in [cut(needs[g], own[self.rank])] if c]
        self.halo_bytes = sum(hi - lo for _, lo, hi in self.halo_recv) * self.y_local.element_size()
        return self

    def spmv_halo(self, x, y_vec):
        """y_vec[own rows] = A_local * x, then the entries of y_vec that this
        rank's rows reference as columns ([need_lo, need_hi)) are completed from
        their owners.  x and y_vec are full-length buffers; only the own slice
        and the halo regions hold meaningful data."""
        dist = self.torch.distributed
        self.local_spmv(x, y_vec[self.r0:self.r1])   # the kernel writes the own slice in place
        self._y_in = y_vec                           # (gather_y reads the slice from here)
        if self.world == 1 or not (self.halo_recv or self.halo_send):
            return y_vec
        # the P2P descriptors only depend on the buffer: built once per y_vec, reused every step
        cache = getattr(self, "_halo_ops", None)
        if cache is None or cache[0] != (y_vec.data_ptr(), y_vec.numel()):
            ops = []
            for g, lo, hi in self.halo_send:
                ops.append(dist.P2POp(dist.isend, y_vec[lo:hi], g, group=self.group))
            for g, lo, hi in self.halo_recv:
                ops.append(dist.P2POp(dist.irecv, y_vec[lo:hi], g, group=self.group))
            self._halo_ops = cache = ((y_vec.data_ptr(), y_vec.numel()), ops)
        for w in dist.batch_isend_irecv(cache[1]):
            w.wait()
        return y_vec

    def gather_y(self, y_full):
        """all-gather of the slices computed by the last product (self.y_local)
        into y_full on every rank."""
        dist = self.torch.distributed
        src = getattr(self, "_y_in", None)
        if src is not None:      # the last product was a halo step: its slice lives in the vector itself
            self.y_local[: self.r1 - self.r0].copy_(src[self.r0:self.r1])
            self._y_in = None
        if self.world == 1:
            y_full.copy_(self.y_local[: self.nrows])
            return y_full
        if self.equal:
            dist.all_gather_into_tensor(y_full, self.y_local, group=self.group)
        else:
            dist.all_gather_into_tensor(self._gather, self.y_local, group=self.group)
            for g in range(self.world):
                a, b = int(self.bounds[g]), int(self.bounds[g + 1])
                y_full[a:b].copy_(self._gather[g * self.max_rows: g * self.max_rows + (b - a)])
        return y_full

    def spmv(self, x, y_full):
        """y_full (nrows, on every rank) = A * x."""
        dist = self.torch.distributed
        self.local_spmv(x, self.y_local)
        self._y_in = None
        if self.world == 1:
            y_full.copy_(self.y_local[: self.nrows])
            return y_full
        if self.equal:
            dist.all_gather_into_tensor(y_full, self.y_local, group=self.group)
        else:
            dist.all_gather_into_tensor(self._gather, self.y_local, group=self.group)
            for g in range(self.world):
                a, b = int(self.bounds[g]), int(self.bounds[g + 1])
                y_full[a:b].copy_(self._gather[g * self.max_rows: g * self.max_rows + (b - a)])
        return y_full
