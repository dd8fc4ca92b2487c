"""world_size-2 (and 3) CPU test of the row-partitioned product's host logic:
partition, x broadcast, y all-gather.  The local kernel is replaced by the
oracle HERE (tests only) because there is no GPU in this container; on the GPU
box bench.py runs the same class with the HIP kernel."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, equal, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle
    import spalinalg_amd as sp
    import spal_synth as synth
    from spalinalg_amd.dist import RowPartitionedSpmv, partition_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 3000 if equal else 3001
        if equal:
            rp, ci, va = synth.banded_csr(n, n, 14, 256, 42)
        else:
            rng = np.random.default_rng(1)          # same on every rank
            lens = rng.integers(0, 30, n)
            rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
            ci = np.concatenate([np.sort(rng.choice(n, k, replace=False)) for k in lens]).astype(np.uint64)
            va = rng.uniform(-1, 1, ci.size)
        bounds = partition_rows(rp, world)
        a = sp.CsrMatrix(n, n, rp, ci, va)
        shard = a.row_slice(int(bounds[rank]), int(bounds[rank + 1]))

        def local(x_full, out_local):          # oracle stands in for the HIP kernel (test only)
            y = oracle.csr_spmv(shard.rowptr(), shard.colind(), shard.values(), x_full.numpy())
            out_local[: y.size].copy_(torch.from_numpy(y))

        op = RowPartitionedSpmv(local, bounds, rank, world, torch.float64, "cpu")
        x = torch.from_numpy(synth.vector(n)) if rank == 0 else torch.zeros(n, dtype=torch.float64)
        op.broadcast_x(x)
        y = torch.empty(n, dtype=torch.float64)
        op.spmv(x, y)
        y_ref = oracle.csr_spmv(rp, ci, va, synth.vector(n))
        ok = bool(np.array_equal(y.numpy(), y_ref))
        if equal:
            # scatter of the x windows + gather of the y slices on rank 0 ("end" mode of bench.py)
            lo, hi = int(shard.colind().min()), int(shard.colind().max()) + 1
            needs = op.plan_x_windows(lo, hi)
            x2 = torch.from_numpy(synth.vector(n)) if rank == 0 else torch.full((n,), np.nan, dtype=torch.float64)
            a0, a1 = op.distribute_x(x2, n, needs)
            ok = ok and a0 <= lo and hi <= a1
            ok = ok and bool(np.array_equal(x2.numpy()[a0:a1], synth.vector(n)[a0:a1]))
            xs = torch.nan_to_num(x2, nan=0.0)          # columns outside the window are never read
            op.local_only(xs)
            y2 = torch.full((n,), np.nan, dtype=torch.float64)
            op.gather_y_root(y2)
            if rank == 0:
                ok = ok and bool(np.array_equal(y2.numpy(), y_ref))
        q.put((rank, ok, op.equal, bounds.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,equal", [(2, True), (2, False), (3, False), (3, True), (4, True), (8, False)])
def test_row_partitioned_spmv_gloo(world, equal):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world * 3 + int(equal)
    procs = [ctx.Process(target=_worker, args=(r, world, port, equal, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(o[0] for o in out) == list(range(world))
    assert all(o[1] for o in out), "every rank must end with the complete, identical y"
    assert all(o[2] == equal for o in out)
    assert all(o[3] == out[0][3] for o in out)


def _halo_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle
    import spalinalg_amd as sp
    import spal_synth as synth
    from spalinalg_amd.dist import RowPartitionedSpmv, partition_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n, w = 1500 * max(world, 4), 512      # (a slice stays several bands tall at every world size)
        rp, ci, va = synth.banded_csr(n, n, 14, w, 77)
        bounds = partition_rows(rp, world)
        a = sp.CsrMatrix(n, n, rp, ci, va)
        r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
        shard = a.row_slice(r0, r1)

        def local(x_full, out_local):          # oracle stands in for the HIP kernel (test only)
            y = oracle.csr_spmv(shard.rowptr(), shard.colind(), shard.values(), x_full.numpy())
            out_local[: y.size].copy_(torch.from_numpy(y))

        op = RowPartitionedSpmv(local, bounds, rank, world, torch.float64, "cpu")
        need_lo, need_hi = int(shard.colind().min()), int(shard.colind().max()) + 1
        op.plan_halo(need_lo, need_hi)
        x0 = synth.vector(n)
        # two products in a row, y feeding back as x: only own slice + halo is ever exchanged
        x = torch.from_numpy(x0.copy())
        y = torch.full((n,), float("nan"), dtype=torch.float64)
        op.spmv_halo(x, y)
        z = torch.full((n,), float("nan"), dtype=torch.float64)
        op.spmv_halo(y, z)
        x1 = oracle.csr_spmv(rp, ci, va, x0)
        x2 = oracle.csr_spmv(rp, ci, va, x1)
        ok1 = bool(np.array_equal(y.numpy()[need_lo:need_hi], x1[need_lo:need_hi]))
        ok2 = bool(np.array_equal(z.numpy()[r0:r1], x2[r0:r1]))
        untouched = bool(np.isnan(y.numpy()[:need_lo]).all() and np.isnan(y.numpy()[need_hi:]).all())
        q.put((rank, ok1, ok2, untouched, op.halo_bytes, (r1 - r0) * 8))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_halo_exchange_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, untouched, halo_bytes, own_bytes in out:
        assert ok1, "own slice + halo of the first product must be complete and exact"
        assert ok2, "the second product (fed by the halo-exchanged y) must be exact on the own rows"
        assert untouched, "nothing outside [need_lo, need_hi) is transferred"
        assert halo_bytes <= 2 * 256 * 8 + 64          # +-W/2 entries per side
        assert halo_bytes < own_bytes / 2
