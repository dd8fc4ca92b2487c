"""Development check: the CSC scatter kernel's neighbour hand-off under repetition -- thousands of launches on several
streams, every result compared with the first (a missed hand-off would leave a row without one super-tile's share)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import scipy.sparse as sps
import spalinalg_amd as sp, spal_synth as synth

n = 1_000_000
rp, ci, va = synth.banded_csr(n, n, 14, 4096, synth.matrix_seed(2))
m = sps.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n)).tocsc()
m.sort_indices()
x = synth.vector(n)
y_ref = torch.from_numpy(m @ x).cuda()
dev = sp.CscMatrix(n, n, m.indptr.astype(np.uint64), m.indices.astype(np.uint64), m.data).device()
dev.set_option("kernel", 1)
print(dev.describe()["flush"], flush=True)
xt = torch.from_numpy(x).cuda()
streams = [torch.cuda.Stream() for _ in range(3)]
outs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in streams]
scale = float(y_ref.abs().max())
bad = 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
for i in range(N):
    k = i % 3
    with torch.cuda.stream(streams[k]):
        outs[k].fill_(float("nan"))
        dev.spmv_torch(xt, outs[k])
        err = float((outs[k] - y_ref).abs().max())
    if not (err <= 1e-11 * scale):
        bad += 1
        if bad < 5:
            print("launch", i, "max abs err", err, flush=True)
torch.cuda.synchronize()
print(f"{N} launches, {bad} with a wrong result", flush=True)
