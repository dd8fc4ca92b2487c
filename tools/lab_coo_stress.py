"""Development check: repeated COO -> CSR assemblies of one handle (the look-back placement under repetition): every
result's arrays are compared with the first assembly's, bit for bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spalinalg_amd as sp, spal_synth as synth

n, length = 2_000_000, 20_000_000
r, c, v = synth.coo(n, n, length, 11, 10, 1)
d = sp.CooMatrix.with_triplets(n, n, r, c, v).upload()
first = None
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
bad = 0
for i in range(N):
    a = d.assemble_csr()
    rp, ci, va = a.download()
    a.close()
    if first is None:
        first = (rp, ci, va.view(np.uint64))
        print("nnz", ci.size, d.describe(), flush=True)
    elif not (np.array_equal(rp, first[0]) and np.array_equal(ci, first[1]) and np.array_equal(va.view(np.uint64), first[2])):
        bad += 1
print(f"{N} assemblies, {bad} differ from the first", flush=True)
