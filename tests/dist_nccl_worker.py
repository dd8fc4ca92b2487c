"""Worker of tests/test_gpu_mg.py::test_row_partitioned_spmv_over_nccl -- run under torch.distributed.run,
one process per GPU, backend nccl (= RCCL over xGMI):

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P tests/dist_nccl_worker.py

Every collective path of spalinalg_amd/dist.py (the host bench.py uses under the driver's launch
contract) against the CPU oracle on every rank; `--backend gloo --same-device` rehearses it on one GPU.
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--same-device", action="store_true")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    import oracle
    import spalinalg_amd as sp
    import spal_synth as synth
    from spalinalg_amd.dist import RowPartitionedSpmv, partition_rows

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if args.same_device else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if args.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)

    for dtype, tdt in ((np.float64, torch.float64), (np.float32, torch.float32)):
        for ragged in (False, True):
            n = 300_000 + 4096 * world
            if ragged:     # unequal slices (nnz-balanced), rows of 1 ... 27 entries
                rp, ci, va = synth.ragged_csr(n, n, 4096, 41, dtype=dtype)
            else:
                rp, ci, va = synth.banded_csr(n, n, 14, 4096, 41, dtype=dtype)
            xh = synth.vector(n, dtype=dtype)
            y_ref = oracle.csr_spmv(rp, ci, va, xh)
            bounds = partition_rows(rp, world)
            r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
            a = sp.CsrMatrix._trusted(n, n, rp, ci, va)
            dev = a.row_slice(r0, r1).device(local)
            op = RowPartitionedSpmv.from_shard(dev, bounds, rank, world, device)
            bits = torch.int64 if dtype == np.float64 else torch.int32
            ref = torch.from_numpy(y_ref).to(device)

            # banded rows: every row streams, bit-identical.  Ragged rows: the few tiles above 1024 entries go to the
            # overflow kernel (tree sums): those rows to rounding (componentwise bound of SURVEY 8d), all others exact.
            tol = 1e-10 if dtype == np.float64 else 1e-4
            bound = torch.from_numpy(oracle.csr_abs_bound(rp, ci, va, xh)).to(device) if ragged else None

            def same(y, want=None, bnd=None):
                want = ref if want is None else want
                if bool(torch.equal(y.view(bits), want.view(bits))):
                    return True
                bad = y.view(bits) != want.view(bits)
                bnd = bound if bnd is None else bnd
                ok = ragged and float(bad.double().mean()) < 0.05 and bool(
                    torch.all((y.double() - want.double()).abs() <= tol * bnd + 1e-300))
                if not ok:
                    idx = bad.nonzero().flatten()
                    print(f"[rank {rank}] {dtype.__name__} ragged={ragged}: {idx.numel()} of {n} rows differ, first {idx[:5].tolist()}; "
                          f"bounds {bounds.tolist()}; got {y[idx[:3]].tolist()} want {want[idx[:3]].tolist()}", flush=True)
                return ok

            # broadcast of x from rank 0 + all-gather of y
            x = torch.from_numpy(xh).to(device) if rank == 0 else torch.zeros(n, dtype=tdt, device=device)
            op.broadcast_x(x)
            y = torch.empty(n, dtype=tdt, device=device)
            op.spmv(x, y)
            assert same(y), f"rank {rank}: broadcast + all-gather differs from the oracle"
            # windows of x scattered from rank 0, local product, gather on rank 0 / all-gather
            lo, hi = int(ci[int(rp[r0]):int(rp[r1])].min()), int(ci[int(rp[r0]):int(rp[r1])].max()) + 1
            needs = op.plan_x_windows(lo, hi)
            probe = x.clone() if rank == 0 else torch.full_like(x, float("nan"))
            a0, a1 = op.distribute_x(probe, n, needs)
            assert a0 <= lo and hi <= a1 and torch.equal(probe[a0:a1], x[a0:a1]), f"rank {rank}: scattered window wrong"
            op.local_only(probe)
            y2 = torch.empty(n, dtype=tdt, device=device)
            op.gather_y(y2)
            assert same(y2), f"rank {rank}: scatter + all-gather differs"
            if op.equal:
                y3 = torch.zeros(n, dtype=tdt, device=device)
                op.gather_y_root(y3)
                assert rank != 0 or same(y3), "gather on rank 0 differs"
            # halo exchange: three steps of x <- A x
            op.plan_halo(lo, hi)
            v, vh = x.clone(), xh
            w = torch.empty_like(v)
            steps = 1 if ragged else 3      # (rounding differences would compound over steps: one step for the ragged case)
            for _ in range(steps):
                op.spmv_halo(v, w)
                v, w = w, v
                vh = oracle.csr_spmv(rp, ci, va, vh)
            yh = torch.empty(n, dtype=tdt, device=device)
            op.gather_y(yh)
            assert same(yh, torch.from_numpy(vh).to(device)), f"rank {rank}: halo steps differ"
            torch.cuda.synchronize()
    dist.barrier()
    # what this rank saw: the communicator's size (RCCL's, not an environment variable) and its device
    seen = torch.zeros(world, dtype=torch.int64, device=device)
    seen[rank] = torch.cuda.current_device() + 1
    dist.all_reduce(seen)
    if rank == 0:
        print(f"[dist worker] backend {dist.get_backend()}, world {dist.get_world_size()}, devices of the ranks "
              f"{[int(v) - 1 for v in seen.tolist()]}", flush=True)
        print("dist worker ok", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
