"""Development tool: config-5 triplets, a few assemblies (no oracle check) -- for profiling variants of the COO kernels."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spalinalg_amd as sp
import spal_synth as synth

cfg = synth.CONFIGS[5]
n = cfg["nrows"]
r, c, v = synth.coo(n, n, cfg["length"], synth.matrix_seed(5), cfg["dup_permille"], cfg["cancel_permille"])
d = sp.CooMatrix.with_triplets(n, n, r, c, v).upload()
for _ in range(8):
    d.assemble_csr().close()
print("ok")
