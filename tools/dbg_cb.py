import numpy as np, spalinalg_amd as sp, spal_synth as synth
n=1_000_000
rp,ci,va=synth.banded_csr(n,n,14,n,synth.matrix_seed(2))
dev=sp.CsrMatrix._trusted(n,n,rp,ci,va).device()
print(dev.describe())
