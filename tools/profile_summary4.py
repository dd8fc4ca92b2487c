#!/usr/bin/env python3
"""Turns the outputs of tools/prof_r4.sh (gpurun_out/prof4/*) into the summaries kept under profiles/r04 and refreshes
profiles/traffic.json (development tool).  FETCH_SIZE is doubled (gfx950 counts 128-byte requests as 64 bytes,
MI355X_MICROARCH.md, HBM section); units are KiB."""
import collections
import csv
import json
import os
import re
import shutil
import sys

P = "gpurun_out/prof4"
OUT = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04"


def short(name):
    m = re.match(r"(?:void )?(?:spal::)?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def bench_json(name):
    for line in reversed(open(f"{P}/{name}.log").read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None


def pmc(name):
    """{kernel: [counter value of every launch]} of a one-counter pass"""
    acc = collections.defaultdict(list)
    path = f"{P}/{name}/b_counter_collection.csv"
    if not os.path.exists(path):
        return acc
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


def mean(v):
    return sum(v) / len(v) if v else 0.0


def main():
    os.makedirs(OUT, exist_ok=True)
    traffic = json.load(open("profiles/traffic.json"))
    src = "profiles/r04/pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled)"
    lines = []
    names = {"3": "kernel_stats_config3_banded_f64.csv", "3f32": "kernel_stats_config3_banded_f32.csv",
             "2": "kernel_stats_config2.csv", "2u": "kernel_stats_config2_uniform.csv", "4": "kernel_stats_config4.csv",
             "5": "kernel_stats_config5.csv", "1": "kernel_stats_config1.csv"}
    for tag, fn in names.items():
        if os.path.exists(f"{P}/stats{tag}/b_kernel_stats.csv"):
            shutil.copy(f"{P}/stats{tag}/b_kernel_stats.csv", f"{OUT}/{fn}")
    # ---- config 3: the timed region's launches against bench.py's own HIP events
    for tag, label in (("3", "f64"), ("3f32", "f32")):
        d = bench_json(f"stats{tag}")
        if not d or not os.path.exists(f"{P}/stats{tag}/b_kernel_trace.csv"):
            continue
        rows = list(csv.DictReader(open(f"{P}/stats{tag}/b_kernel_trace.csv")))
        seq = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows
               if "csr_spmv_" in r["Kernel_Name"]]
        steps, warm = d["steps"], d["warmup"]
        kept = d["roofline"]["kernel"]
        mine = [t for n, t in seq if n == kept]
        tail = mine[-(warm + steps):]          # (at N = 1 the timed region's launches are the process's last of this kernel)
        timed, after = tail[warm:], []
        plan = d["config"]["plan"]
        B = d["roofline"]["algorithmic_bytes_per_launch"]
        txt = (f"# per-launch durations from rocprofv3 --kernel-trace of: python3 bench.py {'--dtype f32 ' if label == 'f32' else ''}--steps {steps} --warmup {warm} --no-cpu-baseline --no-ceiling --no-other-configs\n"
               f"kept form: {kept}   plan: {json.dumps({k: plan.get(k) for k in ('rows_per_tile', 'slide', 'merged_stream', 'nt_store', 'ring_pages', 'tile_steps', 'uniform_row_fraction', 'autotune_us', 'merged_us', 'y_store_us', 'placement_us', 'placement_tries')})}\n"
               f"timed region ({len(timed)} launches): mean {mean(timed):.2f} us  min {min(timed):.2f}  max {max(timed):.2f}"
               f"  -> {B} algorithmic bytes / mean = {B / mean(timed) / 1e3:.1f} GB/s = {B / mean(timed) / 1e3 / 8000:.4f} of 8 TB/s\n"
               f"bench.py's HIP events on the same launches: {d['ms_per_step'] * 1e3:.2f} us = {d['value']:.1f} GFLOP/s, roofline.frac {d['roofline']['frac']}"
               f" (algorithmic bytes), moved_frac {d['roofline'].get('moved_frac')}\n"
               f"(the launches before it: autotune -- both forms, two rounds -- the placement walks, {warm} warm-up)\n")
        for n in sorted({n for n, _ in seq}):
            v = [t for m, t in seq if m == n]
            txt += f"all launches of {n}: {len(v)}, mean {mean(v):.2f} us, min {min(v):.2f}\n"
        open(f"{OUT}/kernel_trace_config3_{label}_timed_region.txt", "w").write(txt)
        lines.append(txt)

    # ---- HBM traffic per launch / per assembly
    def one_kernel(tag, kernel, key, note):
        f, w = pmc(f"fetch{tag}"), pmc(f"write{tag}")
        if kernel not in f:
            return f"## config {tag}: {kernel} not in the counter pass"
        rd = mean(f[kernel]) * 2 * 1024
        wr = mean(w.get(kernel, [0.0])) * 1024
        traffic[key] = {"hbm_bytes_per_launch": int(rd + wr), "read": int(rd), "written": int(wr), "kernel": kernel, "source": src}
        return (f"## config {tag}: {note}\n{kernel:28s} launches {len(f[kernel]):4d}  FETCH_SIZE {mean(f[kernel]):12.1f} KiB x2 = {rd / 1e6:9.2f} MB read   "
                f"WRITE_SIZE {wr / 1e6:9.2f} MB written\n=> {key}: {(rd + wr) / 1e6:.1f} MB per launch")
    t = []
    for tag, key, note in (("3", "config3_banded_f64_n1", "bench.py (config 3)"),
                           ("3f32", "config3_banded_f32_n1", "bench.py --dtype f32 (config 3)"),
                           ("2", "config2_banded_f64_n1", "bench.py --config 2 --copies 1 (188 MB working set: the Infinity Cache serves part of it)"),
                           ("2u", "config2_uniform_f64_n1", "bench.py --config 2 --dist uniform --copies 1"),
                           ("4", "config4_scatter_f64", "bench.py --config 4 --copies 1")):
        d = bench_json(f"fetch{tag}")
        if d:
            t.append(one_kernel(tag, d["roofline"]["kernel"].split(" ")[0], key, note + ", kernel " + d["roofline"]["kernel"].split(" ")[0]))
    # config 5: every kernel of the assembly call (the CSR planning of the result included), per assembly
    f5, w5 = pmc("fetch5"), pmc("write5")
    if f5:
        n_asm = max(1, len(f5.get("coo_group_sort", [])))
        # (round 4: the product kernels' plan of the result is no longer part of the assembly call -- spal_csr_plan / the first product)
        own = [k for k in f5 if k.startswith(("radix_", "digit_scan", "scan_", "rows_", "group_offsets", "groups_", "coo_"))]
        out = [f"## config 5: bench.py --config 5, every kernel of the assembly call, per assembly ({n_asm} assemblies in the counter pass)"]
        tot_r = tot_w = 0.0
        for k in sorted(own):
            rd = sum(f5[k]) * 2 * 1024 / n_asm
            wr = sum(w5.get(k, [])) * 1024 / n_asm
            out.append(f"{k:28s} {len(f5[k]) / n_asm:5.1f} launches per assembly  {rd / 1e6:9.2f} MB read  {wr / 1e6:9.2f} MB written")
            tot_r += rd
            tot_w += wr
        out.append(f"=> config5_assembly_f64: {tot_r / 1e6:.1f} MB read + {tot_w / 1e6:.1f} MB written = {(tot_r + tot_w) / 1e6:.1f} MB per assembly")
        traffic["config5_assembly_f64"] = {"hbm_bytes_per_launch": int(tot_r + tot_w), "read": int(tot_r), "written": int(tot_w), "source": src}
        for k in ("csr_spmv_cblock", "csr_spmv_cblock_rows", "csr_spmv_stream"):
            if k in f5:
                rd, wr = mean(f5[k]) * 2 * 1024, mean(w5.get(k, [0.0])) * 1024
                out.append(f"the product on the result: {k}: {rd / 1e6:.1f} MB read + {wr / 1e6:.1f} MB written per launch")
                traffic["config5_result_spmv_f64"] = {"hbm_bytes_per_launch": int(rd + wr), "read": int(rd), "written": int(wr), "kernel": k, "source": src}
        t.append("\n".join(out))
    open(f"{OUT}/pmc_traffic.txt", "w").write("# HBM bytes per launch from rocprofv3 --pmc passes (tools/prof_r4.sh)\n" + "\n".join(t) + "\n")
    lines += t
    json.dump(traffic, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
