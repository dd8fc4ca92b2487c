import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import spalinalg_amd as sp
import spal_synth as synth
n = 200_000
rp, ci, va = synth.banded_csr(n, n, 14, 4096, 3)
d = sp.CsrMatrix._trusted(n, n, rp, ci, va).device()
print(d.describe())
d.set_option("slide", 0)
print(d.describe())
d.set_option("uniform_rows", 0)
print(d.describe())
