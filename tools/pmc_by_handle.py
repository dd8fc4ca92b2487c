#!/usr/bin/env python3
"""Groups the counter rows of a `rocprofv3 --pmc ... -- python3 tools/lab_place.py pmc` run by handle:
the stream kernel's dispatches come 6 per handle, in handle order.  usage: pmc_by_handle.py <counter_collection.csv>"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if "csr_spmv_stream" in r["Kernel_Name"]]
by_disp = collections.OrderedDict()
for r in rows:
    by_disp.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
disp = list(by_disp.values())
per = 6
for h in range(len(disp) // per):
    grp = disp[h * per + 2:(h + 1) * per]       # skip the two warm-up launches
    names = sorted(grp[0])
    print(f"handle {h}: " + "  ".join(f"{n} {sum(g[n] for g in grp) / len(grp):.4g}" for n in names))
