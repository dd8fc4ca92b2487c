"""Development check: the CSC scatter entry point under torch's graph capture, per flush form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spalinalg_amd as sp, spal_synth as synth
import scipy.sparse as sps
n = 100_000
rp, ci, va = synth.banded_csr(n, n, 14, 4096, 33)
m = sps.csr_matrix((va, ci.astype(np.int64), rp.astype(np.int64)), shape=(n, n)).tocsc()
m.sort_indices()
cp, ri, cv = m.indptr.astype(np.uint64), m.indices.astype(np.uint64), m.data
x = synth.vector(n)
y_ref = m @ x
xt = torch.from_numpy(x).cuda()
for flush in (2, 1, 0):
    dev = sp.CscMatrix(n, n, cp, ri, cv).device()
    dev.set_option("kernel", 1)
    dev.set_option("flush", flush)
    print("flush", flush, dev.describe()["flush"], flush=True)
    g = torch.cuda.CUDAGraph()
    yg = torch.zeros(n, dtype=torch.float64, device="cuda")
    cap = torch.cuda.Stream()
    with torch.cuda.stream(cap):
        dev.spmv_torch(xt, yg)
        torch.cuda.synchronize()
        print("  eager max err", float(np.abs(yg.cpu().numpy() - y_ref).max()), flush=True)
        with torch.cuda.graph(g, stream=cap):
            dev.spmv_torch(xt, yg)
    for rep in range(2):
        yg.fill_(float("nan"))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        yh = yg.cpu().numpy()
        print("  replay", rep, "nan count", int(np.isnan(yh).sum()), "max err", float(np.nanmax(np.abs(yh - y_ref))), yh[:4], flush=True)
