#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
cat > /tmp/pl_once.py <<'PY'
import sys, numpy as np, torch
sys.path.insert(0, ".")
import spalinalg_amd as sp, spal_synth as synth
rng = np.random.default_rng(5)
n = 2_000_000
pl = np.minimum((rng.pareto(1.6, n) * 6 + 1).astype(np.int64), 5000)
rows = np.repeat(np.arange(n, dtype=np.int64), pl)
cols = np.clip(rows - 5000 + rng.integers(0, 10000, rows.size), 0, n - 1)
key = np.unique(rows * n + cols)
r2, c2 = key // n, key % n
rp = np.concatenate([[0], np.cumsum(np.bincount(r2, minlength=n))]).astype(np.uint64)
dev = sp.CsrMatrix(n, n, rp, c2.astype(np.uint64), rng.uniform(-1, 1, c2.size)).device()
x = torch.from_numpy(synth.vector(n)).cuda(); y = torch.empty_like(x)
print(dev.describe())
for _ in range(30): dev.spmv_torch(x, out=y)
torch.cuda.synchronize()
dev.autotune(x, y, iters=10)
print("after autotune", dev.describe()["short_part"]["slide"], dev.describe()["short_part"]["autotune_us"])
def t(label):
    for _ in range(5): dev.spmv_torch(x, out=y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): dev.spmv_torch(x, out=y)
    e1.record(); torch.cuda.synchronize()
    d = dev.describe()["short_part"]
    print(label, round(e0.elapsed_time(e1) / 30 * 1e3, 1), "us", d["kernel"], "panel_tiles", d["panel_tiles"], "win", d["lds_window_bytes"], "stream", d["stream_row_fraction"], "lds", d["lds_row_fraction"], flush=True)
t("default")
dev.set_option("cblock", 1); t("cblock=1")
dev.set_option("cblock", -1); dev.set_option("row_split_threshold", 64); t("threshold 64")
dev.set_option("cblock", 1); t("threshold 64, cblock=1")
dev.set_option("cblock", -1); dev.set_option("row_split_threshold", 32); t("threshold 32")
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pl2 -o b -- python3 /tmp/pl_once.py > $O/t25_pl.log 2>&1
grep -v amdgpu $O/t25_pl.log | grep 'us \|after' | cut -c1-300
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/stats_pl2/b_kernel_stats.csv")))[:8]:
    print(r["Name"][:90].ljust(90), r["Calls"], r["AverageNs"])
PY
exit 0
