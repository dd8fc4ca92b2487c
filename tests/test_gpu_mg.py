"""The row-partitioned product from ONE process through the C ABI (spal_mg_*):
x windows scattered from GPU 0, local kernels, y gathered on GPU 0, all-gather, per-step halo
exchange -- against the CPU oracle, bit for bit on the stream path.

Two ways to reach N > 1 shards:
 * `virtual`: the copy transport with a device list that repeats GPU 0 -- several shards, each with its
   own stream, on the one GPU of a gpurun box.  Exercises the partition, the window / halo planner, the
   message lists and the event ordering between streams; runs wherever one GPU is visible.
 * `real`: ngpus distinct GPUs, RCCL (grouped ncclSend / ncclRecv, ncclBroadcast, ncclAllGather) and the
   peer-copy transport.  Collected everywhere, skipped unless that many GPUs are visible: the first
   multi-GPU box runs them as tests before any benchmark does.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

import spalinalg_amd as sp
import spal_synth as synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    try:
        return sp.device_count()
    except Exception:  # noqa: BLE001
        return 0


def need_gpus(n):
    return pytest.mark.skipif(_ngpu() < n, reason=f"needs {n} GPUs, {_ngpu()} visible")


def banded(n, dtype=np.float64, window=4096, seed=17):
    rp, ci, va = synth.banded_csr(n, n, 14, window, seed, dtype=dtype)
    return sp.CsrMatrix(n, n, rp, ci, va), synth.vector(n, dtype=dtype)


def check_all_paths(mg, a, x, oracle):
    """every exchange path of one handle against the oracle (bit-identical: the shards run the stream kernel)"""
    rp, ci, va = a.rowptr(), a.colind(), a.values()
    y_ref = oracle.csr_spmv(rp, ci, va, x)
    bits = np.uint64 if a.dtype == np.float64 else np.uint32
    G = mg.ngpus
    b = mg.partition().astype(np.int64)
    assert b[0] == 0 and b[-1] == a.nrows() and np.all(np.diff(b) > 0)
    lo, hi = mg.windows()
    for g in range(G):       # the window is exactly the span of the shard's columns
        cols = ci[int(rp[b[g]]):int(rp[b[g + 1]])]
        assert (int(lo[g]), int(hi[g])) == (int(cols.min()), int(cols.max()) + 1)
    # host convenience call
    assert np.array_equal(mg.spmv(x).view(bits), y_ref.view(bits))
    # resident: windows scattered, K local products, gathered on GPU 0
    mg.set_x(x)
    mg.scatter_x()
    for _ in range(3):
        mg.spmv_local()
    mg.gather_y()
    assert np.array_equal(mg.y_gathered().view(bits), y_ref.view(bits))
    # resident: plain broadcast + all-gather (the path north_star words)
    mg.set_x(x)
    mg.broadcast_x()
    mg.spmv_resident()
    assert np.array_equal(mg.y_allgathered().view(bits), y_ref.view(bits))
    t = mg.timing()
    assert t["x_distribution"] is not None and t["compute"] > 0 and t["y_collection"] is not None
    eb = mg.exchange_bytes()
    assert eb["y_gather"] == (a.nrows() - int(b[1])) * a.dtype.itemsize
    assert eb["x_scatter"] == sum(int(hi[g] - lo[g]) for g in range(1, G)) * a.dtype.itemsize
    # iterative use: x <- A x three times with the halo exchange, against three oracle products
    mg.set_x(x)
    mg.scatter_x()
    v = x
    for _ in range(3):
        mg.spmv_halo()
        v = oracle.csr_spmv(rp, ci, va, v)
    mg.gather_y()
    assert np.array_equal(mg.y_gathered().view(bits), v.view(bits))
    assert mg.timing()["halo"] is not None
    # ... and a plain product afterwards still works (the vectors swapped roles an odd number of times)
    mg.set_x(x)
    mg.scatter_x()
    mg.spmv_local()
    mg.gather_y()
    assert np.array_equal(mg.y_gathered().view(bits), y_ref.view(bits))


@pytest.mark.parametrize("shards", [2, 3, 4, 8])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_virtual_shards_on_one_gpu(oracle, shards, dtype):
    a, x = banded(120_000 + 77 * shards, dtype)
    mg = sp.MultiGpuCsr(a, shards, devices=[0] * shards)
    assert mg.transport == "copy"
    check_all_paths(mg, a, x, oracle)
    mg.close()


def test_virtual_shards_wide_windows_and_ragged_rows(oracle):
    """windows that cover most of x (the convenience call then broadcasts), rows of 1 ... 27 entries,
    nnz-balanced (unequal) slices"""
    n = 90_000
    rp, ci, va = synth.ragged_csr(n, n, n, 23)       # uniform columns: every shard reads nearly all of x
    a = sp.CsrMatrix(n, n, rp, ci, va)
    x = synth.vector(n)
    mg = sp.MultiGpuCsr(a, 3, devices=[0, 0, 0])
    y = mg.spmv(x)
    ref = oracle.csr_spmv(rp, ci, va, x)
    bound = oracle.csr_abs_bound(rp, ci, va, x)
    assert np.all(np.abs(y - ref) <= 1e-10 * bound + 1e-300)
    sizes = np.diff(mg.partition().astype(np.int64))
    assert sizes.min() != sizes.max() or n % 3 == 0
    mg.close()


def test_non_square_has_no_halo_and_repeats_need_copy_transport(oracle):
    rp, ci, va = synth.banded_csr(50_000, 70_000, 14, 4096, 5)
    a = sp.CsrMatrix(50_000, 70_000, rp, ci, va)
    mg = sp.MultiGpuCsr(a, 2, devices=[0, 0])
    x = synth.vector(70_000)
    assert np.array_equal(mg.spmv(x), oracle.csr_spmv(rp, ci, va, x))
    mg.set_x(x)
    mg.scatter_x()
    with pytest.raises(sp.Panic):
        mg.spmv_halo()
    mg.close()
    with pytest.raises(sp.Panic):
        sp.MultiGpuCsr(a, 2, devices=[0, 0], transport="rccl")   # RCCL refuses duplicate devices


def test_rccl_loads_and_runs_with_one_rank(oracle):
    """what a 1-GPU box can exercise of the RCCL transport: librccl is found and loaded (dlopen), a one-rank
    communicator is created (ncclCommInitAll), ncclBroadcast / ncclAllGather run in their groups; with one rank
    there is nothing to send, so the grouped ncclSend / ncclRecv lists are empty."""
    a, x = banded(200_000)
    mg = sp.MultiGpuCsr(a, 1, transport="rccl")
    assert mg.transport == "rccl"
    y_ref = oracle.csr_spmv(a.rowptr(), a.colind(), a.values(), x)
    assert np.array_equal(mg.spmv(x), y_ref)
    mg.set_x(x)
    mg.broadcast_x()
    mg.spmv_resident()
    assert np.array_equal(mg.y_allgathered(), y_ref)
    mg.set_x(x)
    mg.scatter_x()
    mg.spmv_halo()
    mg.gather_y()
    assert np.array_equal(mg.y_gathered(), y_ref)
    mg.close()


def test_dist_worker_rehearsal_two_ranks_on_one_gpu():
    """tests/dist_nccl_worker.py with two ranks sharing GPU 0 over gloo: every collective path of
    spalinalg_amd/dist.py (scatter of x windows, gather on rank 0, halo steps, ragged unequal slices) with the
    real kernels, on a 1-GPU box; the nccl run of the same worker is test_row_partitioned_spmv_over_nccl."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29711",
                          os.path.join(ROOT, "tests", "dist_nccl_worker.py"), "--backend", "gloo", "--same-device"],
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    assert "dist worker ok" in out.stdout
    seen = [ln for ln in out.stdout.splitlines() if ln.startswith("[dist worker]")]
    print("\n".join(seen))                          # the world size and devices the ranks saw
    assert any("backend gloo, world 2" in ln for ln in seen), out.stdout[-2000:]


def test_dist_worker_over_nccl_with_one_rank():
    """what a 1-GPU box can run of spalinalg_amd/dist.py over the nccl backend (= RCCL): the process group, broadcast,
    all-gather, gather, batched isend / irecv lists (empty with one rank) and the halo steps, against the oracle."""
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
                          "--master-addr", "127.0.0.1", "--master-port", "29712",
                          os.path.join(ROOT, "tests", "dist_nccl_worker.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    assert "dist worker ok" in out.stdout
    seen = [ln for ln in out.stdout.splitlines() if ln.startswith("[dist worker]")]
    print("\n".join(seen))                          # the RCCL world size and devices the ranks saw
    assert any("backend nccl, world 1" in ln for ln in seen), out.stdout[-2000:]


@pytest.mark.parametrize("transport", ["rccl", "copy"])
@pytest.mark.parametrize("ngpus", [pytest.param(2, marks=need_gpus(2)), pytest.param(4, marks=need_gpus(4)),
                                   pytest.param(8, marks=need_gpus(8))])
def test_real_gpus(oracle, ngpus, transport):
    """ngpus distinct GPUs: RCCL and peer copies.  Skipped on a 1-GPU box."""
    for dtype in (np.float64, np.float32):
        a, x = banded(400_000 + 1000 * ngpus, dtype)
        mg = sp.MultiGpuCsr(a, ngpus, transport=transport)
        assert mg.transport == transport
        # (the first box with several GPUs runs this before any benchmark: say what it saw)
        print(f"[test_real_gpus] {transport}: {mg.ngpus} GPUs in the communicator, devices {mg.devices}, "
              f"{np.dtype(dtype).name}, partition {list(mg.partition())}")
        assert mg.ngpus == ngpus and len(set(mg.devices)) == ngpus
        check_all_paths(mg, a, x, oracle)
        mg.close()


@pytest.mark.parametrize("ngpus", [pytest.param(2, marks=need_gpus(2)), pytest.param(4, marks=need_gpus(4)),
                                   pytest.param(8, marks=need_gpus(8))])
def test_row_partitioned_spmv_over_nccl(ngpus):
    """spalinalg_amd/dist.py (one process per GPU, torch.distributed over RCCL): broadcast / scatter of x
    windows / all-gather / gather on rank 0 / halo exchange, every rank checking its results against the
    oracle.  Skipped on a 1-GPU box."""
    port = 29700 + ngpus
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpus}",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "tests", "dist_nccl_worker.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-4000:]
    assert "dist worker ok" in out.stdout
    seen = [ln for ln in out.stdout.splitlines() if ln.startswith("[dist worker]")]
    print("\n".join(seen))                          # the RCCL world size and devices the ranks saw
    assert any(f"world {ngpus}" in ln for ln in seen), out.stdout[-2000:]


def test_eight_shard_rehearsal_at_config3_size_through_the_c_abi():
    """BASELINE config 3 at FULL size (10M x 10M, 140M entries) cut into the EIGHT shards an 8-GPU node
    holds, driven by bench.py --host mg through spal_mg_* with all eight on GPU 0 (copy transport): the partition (eight
    ranges balanced by stored entries), the x windows, the local kernels, the gather and the all-gather at their real sizes --
    one GPU, so NOT a scaling measurement: what it pins is that the N = 8 line parses, carries both end-to-end totals and the
    communication split, and that GPU 0's y carries the oracle's bits."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    out = subprocess.run([sys.executable, bench, "--host", "mg", "--gpus", "8", "--devices", "0,0,0,0,0,0,0,0", "--steps", "5",
                          "--warmup", "2"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    print(lines[0][:1500])
    assert d["config"]["shards"] == 8 and d["config"]["transport"] == "copy" and d["n_gpus"] == 1
    part = d["config"]["partition"]
    assert len(part) == 9 and part[0] == 0 and part[-1] == 10_000_000 and all(b > a for a, b in zip(part, part[1:]))
    assert all(abs((b - a) * 14 - 140_000_000 / 8) <= 0.01 * 140_000_000 / 8 for a, b in zip(part, part[1:])), part   # 14 per row: balanced by entries
    for key in ("end_to_end_windows_ms", "end_to_end_broadcast_allgather_ms", "comm_ms", "compute_only", "halo"):
        assert d.get(key), key
    assert set(d["comm_ms"]) == {"x_distribution", "y_collection"}
    assert d["allgather_equals_gather_bit_for_bit"] is True
    assert d["cpu_baseline"]["gpu_equals_cpu_bit_for_bit"] is True        # GPU 0's gathered y against the oracle, all 10M rows
    # every shard reads its own slice of x +- half the band: 7 windows of ~1.25M + 4096 columns leave GPU 0
    assert d["config"]["exchange_bytes"]["x_scatter"] < 0.2 * 8 * 10_000_000 * 7


def test_dist_host_rehearsal_three_ranks_on_one_gpu_config2():
    """the driver's launch contract (torch.distributed.run, one process per rank) with THREE ranks sharing GPU 0 over gloo at
    config 2's size: the N > 1 line of bench.py parses and carries what the scaling table will be built from.  (The pool allows
    six processes on a card; this test's parent and the launcher count: the 5-rank run at config 3's size is
    tools/rehearse_n8.sh.)"""
    import json
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                          "--master-addr", "127.0.0.1", "--master-port", "29733", os.path.join(ROOT, "bench.py"),
                          "--gpus", "3", "--config", "2", "--steps", "5", "--warmup", "2", "--backend", "gloo", "--same-device",
                          "--no-cpu-baseline"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["value"] > 0
    for key in ("end_to_end_broadcast_allgather_ms", "comm_ms", "compute_only"):
        assert d.get(key) is not None, key
    assert "end_to_end_windows_ms" in d          # (None when the verified dry run of the window scatter sent everyone to the plain collectives)
    part = d["config"]["partition_rows"]
    assert d["config"]["partition"] == "rows/3" and len(part) == 4 and part[0] == 0 and part[-1] == 1_000_000
    assert all(abs((b - a) - 333_333) <= 3_400 for a, b in zip(part, part[1:])), part
