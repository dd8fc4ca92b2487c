#!/usr/bin/env python3
"""Turns the outputs of `tools/gpu_round.sh prof` (gpurun_out/prof_stats, prof_fetch, prof_write) into the
summaries kept under profiles/: the timed-region text, the --stats csv, the two PMC csvs (development tool)."""
import csv
import re
import shutil
import sys


def main(out_dir):
    rows = list(csv.DictReader(open("gpurun_out/prof_stats/bench_kernel_trace.csv")))
    seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
           for r in rows if "csr_spmv_stream" in r["Kernel_Name"]]
    log = open("gpurun_out/prof_stats.log").read()
    steps = int(re.search(r'"steps": (\d+)', log).group(1))
    warm = int(re.search(r'"warmup": (\d+)', log).group(1))
    ms = float(re.search(r'"ms_per_step": ([0-9.]+)', log).group(1))
    gf = float(re.search(r'"value": ([0-9.]+)', log).group(1))
    frac = float(re.search(r'"frac": ([0-9.]+)', log).group(1))
    kept_pers = '"persistent": 1' in log
    kept = [d for n, d in seq if ("persistent" in n) == kept_pers]
    other = [d for n, d in seq if ("persistent" in n) != kept_pers]
    tail = [d for _, d in seq][-(warm + steps + 3 + steps):]
    timed, after = tail[warm:warm + steps], tail[warm + steps:]
    auto_kept = kept[:len(kept) - len(tail)]

    def mean(v):
        return sum(v) / len(v)
    form = "persistent" if kept_pers else "plain"
    txt = f"""# per-launch durations of the CSR stream kernel from rocprofv3 --kernel-trace (the run that produced
# kernel_stats_config3_banded_f64.csv): python3 bench.py --steps {steps} --warmup {warm} --no-cpu-baseline
# launch order: autotune (plain, persistent, plain+nt_store, persistent+nt_store; 33 launches each, two rounds),
#               then the kept form ({form}): {warm} warm-up, {steps} timed, 3 + {steps} 'kernel alone'
autotune, kept form  ({len(auto_kept)} launches): mean {mean(auto_kept):.2f} us
autotune, other form ({len(other)} launches): mean {mean(other):.2f} us
timed region         ({steps} launches): mean {mean(timed):.2f} us  min {min(timed):.2f}  max {max(timed):.2f}   <- bench.py's HIP events on the same launches: {ms*1e3:.2f} us = {gf:.1f} GFLOP/s = {100*frac:.1f} % of 8 TB/s
kernel alone after it ({len(after)} launches): mean {mean(after):.2f} us
all launches of the kept form ({len(kept)}): mean {mean(kept):.2f} us  (the figure of --stats, which mixes the phases)
"""
    open(f"{out_dir}/kernel_trace_config3_timed_region.txt", "w").write(txt)
    print(txt)
    shutil.copy("gpurun_out/prof_stats/bench_kernel_stats.csv", f"{out_dir}/kernel_stats_config3_banded_f64.csv")
    for name in ("fetch", "write"):
        src = f"gpurun_out/prof_{name}/bench_counter_collection.csv"
        shutil.copy(src, f"{out_dir}/pmc_{name}_size_config3_banded_f64.csv")
        rows = list(csv.DictReader(open(src)))
        for k in ("csr_spmv_stream<", "csr_spmv_stream_persistent<"):
            vals = [float(r["Counter_Value"]) for r in rows if k in r["Kernel_Name"]]
            if vals:
                print(f"{name.upper()}_SIZE {k}...>: {len(vals)} launches, mean {mean(vals):.1f} KiB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r01")
