#!/usr/bin/env python3
"""Turns the outputs of tools/prof_r2.sh (gpurun_out/prof/*) into the summaries kept under profiles/r02 and
refreshes profiles/traffic.json (development tool).  FETCH_SIZE is doubled (gfx950 counts 128-byte requests as
64 bytes, MI355X_MICROARCH.md section HBM); units are KiB."""
import collections
import csv
import json
import re
import shutil
import sys

P = "gpurun_out/prof"
OUT = sys.argv[1] if len(sys.argv) > 1 else "profiles/r02"


def short(name):
    m = re.match(r"(?:void )?(?:spal::)?([A-Za-z0-9_]+)", name)
    return m.group(1) if m else name[:40]


def bench_json(name):
    for line in reversed(open(f"{P}/{name}.log").read().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    return None


def pmc_means(name):
    """{kernel: {counter: (mean, launches)}}"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f"{P}/{name}/b_counter_collection.csv")):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    lines = []
    traffic = json.load(open("profiles/traffic.json"))
    # ---- config 3: stats + timed region
    shutil.copy(f"{P}/stats3/b_kernel_stats.csv", f"{OUT}/kernel_stats_config3_banded_f64.csv")
    d = bench_json("stats3")
    rows = list(csv.DictReader(open(f"{P}/stats3/b_kernel_trace.csv")))
    seq = [(short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows
           if "csr_spmv_" in r["Kernel_Name"]]
    steps, warm = d["steps"], d["warmup"]
    kept = "csr_spmv_slide" if d["config"]["plan"]["slide"] else "csr_spmv_stream"
    tail = [t for n, t in seq if n == kept][-(warm + steps + 3 + steps):]
    timed, after = tail[warm:warm + steps], tail[warm + steps:]
    txt = (f"# per-launch durations from rocprofv3 --kernel-trace of: python3 bench.py --steps {steps} --warmup {warm} --no-cpu-baseline --no-ceiling\n"
           f"# launch order: autotune (4 forms x 2 rounds x 33 launches, then placement tries), {warm} warm-up, {steps} timed, 3 + {steps} 'kernel alone'\n"
           f"kept form: {kept}   plan: {json.dumps({k: d['config']['plan'][k] for k in ('rows_per_tile', 'slide', 'ring_pages', 'tile_steps', 'uniform_row_fraction', 'nt_store', 'autotune_us', 'placement_us')})}\n"
           f"timed region ({len(timed)} launches): mean {sum(timed) / len(timed):.2f} us  min {min(timed):.2f}  max {max(timed):.2f}"
           f"   <- bench.py's HIP events on the same launches: {d['ms_per_step'] * 1e3:.2f} us = {d['value']:.1f} GFLOP/s, roofline.frac {d['roofline']['frac']}"
           f" (algorithmic bytes), moved_frac {d['roofline']['moved_frac']}\n"
           f"kernel alone after it ({len(after)} launches): mean {sum(after) / len(after):.2f} us\n")
    for n in sorted({n for n, _ in seq}):
        v = [t for m, t in seq if m == n]
        txt += f"all launches of {n}: {len(v)}, mean {sum(v) / len(v):.2f} us, min {min(v):.2f}\n"
    open(f"{OUT}/kernel_trace_config3_timed_region.txt", "w").write(txt)
    lines.append(txt)
    # ---- HBM traffic per launch
    def traffic_of(tag, kernels, key, note):
        f, w = pmc_means(f"fetch{tag}"), pmc_means(f"write{tag}")
        out = [f"## config {tag}: {note}"]
        tot_r = tot_w = 0.0
        for k in kernels:
            if k not in f:
                continue
            rd = f[k]["FETCH_SIZE"][0] * 2 * 1024
            wr = w.get(k, {}).get("WRITE_SIZE", (0, 0))[0] * 1024
            per = kernels[k]
            out.append(f"{k:28s} launches {f[k]['FETCH_SIZE'][1]:4d}  FETCH_SIZE {f[k]['FETCH_SIZE'][0]:12.1f} KiB x2 = {rd / 1e6:9.2f} MB read   "
                       f"WRITE_SIZE {wr / 1e6:9.2f} MB written   (x {per} per unit of work)")
            tot_r += rd * per
            tot_w += wr * per
        out.append(f"=> {key}: {tot_r / 1e6:.1f} MB read + {tot_w / 1e6:.1f} MB written = {(tot_r + tot_w) / 1e6:.1f} MB per launch / assembly")
        traffic[key] = {"hbm_bytes_per_launch": int(tot_r + tot_w), "read": int(tot_r), "written": int(tot_w),
                        "source": f"profiles/r02/pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE doubled)"}
        return "\n".join(out)
    t = []
    t.append(traffic_of("3", {kept: 1}, "config3_banded_f64_n1", f"bench.py (config 3), kernel {kept}"))
    traffic["config3_banded_f64_n1_slide" if kept == "csr_spmv_slide" else "config3_banded_f64_n1_plain"] = traffic["config3_banded_f64_n1"]
    other = "csr_spmv_stream" if kept == "csr_spmv_slide" else "csr_spmv_slide"
    t.append(traffic_of("3", {other: 1}, "config3_banded_f64_n1_" + ("plain" if other == "csr_spmv_stream" else "slide"),
                        f"the other form timed by the autotune, {other}"))
    d2 = bench_json("fetch2")
    k2 = "csr_spmv_slide" if d2["config"]["plan"]["slide"] else "csr_spmv_stream"
    t.append(traffic_of("2", {k2: 1}, "config2_banded_f64_n1", f"bench.py --config 2 --copies 1, kernel {k2} (188 MB working set: the Infinity Cache serves part of it)"))
    t.append(traffic_of("4", {"csc_spmv_scatter": 1, "fill_zero": 1}, "config4_scatter_f64",
                        "bench.py --config 4 --copies 1: csc_spmv_scatter (neighbour hand-off: y is stored, no zero fill; "
                        "a fill_zero line appears only where the atomics flush runs)"))
    f5 = pmc_means("fetch5")
    # kernels of ONE assembly: both radix scatters, the second pass's histogram and its scan, the group kernel, the
    # planning of the resulting CSR handle.  Not counted: what runs once per handle at upload (coo_group_hist,
    # groups_check*, the first pass's histogram and scans).
    per5 = {k: 1 for k in f5 if k.startswith(("radix_", "coo_group_sort", "scan_", "csr_block", "csr_stream_check", "csr_slide_scan"))}
    if "radix_scatter" in per5:
        per5["radix_scatter"] = 2
    t.append(traffic_of("5", per5, "config5_assembly_f64", "bench.py --config 5: kernels of one assembly (radix_scatter runs twice; the second pass's histogram, its scan and the planning of the result once each; upload-time kernels not counted)"))
    open(f"{OUT}/pmc_traffic.txt", "w").write("# HBM bytes per launch from rocprofv3 --pmc passes (tools/prof_r2.sh)\n" + "\n".join(t) + "\n")
    lines += t
    json.dump(traffic, open("profiles/traffic.json", "w"), indent=1, sort_keys=True)
    for c in ("2", "4", "5"):
        shutil.copy(f"{P}/stats{c}/b_kernel_stats.csv", f"{OUT}/kernel_stats_config{c}.csv")
    # ---- SQ counters
    sq = pmc_means("sq3")
    s = ["# SQ counters per launch (rocprofv3 --pmc, bench.py config 3): waves wait on memory, the LDS is a quarter busy"]
    for k in ("csr_spmv_slide", "csr_spmv_stream"):
        if k in sq:
            c = {n: v[0] for n, v in sq[k].items()}
            s.append(f"{k}: " + "  ".join(f"{n} {v:.4g}" for n, v in sorted(c.items())))
            s.append(f"   waiting {100 * c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']:.1f} % of wave cycles; LDS bank-conflict cycles / LDS active cycles "
                     f"{100 * c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE']:.1f} %; LDS active / busy cycles {100 * c['SQ_LDS_IDX_ACTIVE'] / c['SQ_BUSY_CYCLES'] / 4:.1f} % (per SIMD-quad)")
    s.append("# The LDS bank-conflict question (csr_spmv_stream, one super-tile per workgroup, same handle): as shipped vs the ablation build")
    s.append("# whose lanes read the product strip at conflict-free addresses (wrong sums, same instruction count) -- the conflicts vanish, the time does not move")
    for name in ("sq3_plain", "sq3_noconf"):
        m = pmc_means(name).get("csr_spmv_stream", {})
        c = {n: v[0] for n, v in m.items()}
        tm = [l for l in open(f"{P}/{name}.log").read().splitlines() if "median" in l]
        s.append(f"{name}: " + "  ".join(f"{n} {v:.4g}" for n, v in sorted(c.items())) + (f"   | {tm[-1].split('median')[1][:20].strip()} (under the profiler)" if tm else ""))
    open(f"{OUT}/pmc_sq_counters.txt", "w").write("\n".join(s) + "\n")
    lines += s
    print("\n".join(lines))


if __name__ == "__main__":
    main()
