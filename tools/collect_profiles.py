#!/usr/bin/env python
"""Turns a gpurun_out/<tag>/ bundle made by tools/profile_round.sh into the committed artifacts under profiles/.
Usage (repo root): python tools/collect_profiles.py <tag> <round-prefix>      e.g.  r01c r01"""
import collections, csv, glob, json, shutil, subprocess, sys

tag, pre = sys.argv[1], sys.argv[2]
src = f"gpurun_out/{tag}"
bench = json.load(open(f"{src}/bench.json"))
prof = json.loads(open(f"{src}/bench_profiled.json").read().strip().splitlines()[-1])
shutil.copy(f"{src}/bench.json", f"profiles/{pre}_bench_100x100k.json")
open(f"profiles/{pre}_bench_100x100k_under_rocprof.json", "w").write(json.dumps(prof) + "\n")
subprocess.run([sys.executable, "tools/rocprof_summary.py", f"{src}/stats/stats_results.db", f"profiles/{pre}_kernel_stats.csv"], check=True)


def per_dispatch(pattern, kernel_sub):
    vals = collections.defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_sub in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in vals.items()}, {k: len(v) for k, v in vals.items()}


f, nf = per_dispatch(f"{src}/pmc_fetch/**/*counter_collection.csv", "k_adc_rowmax")
w, _ = per_dispatch(f"{src}/pmc_write/**/*counter_collection.csv", "k_adc_rowmax")
sq, _ = per_dispatch(f"{src}/pmc_sq/**/*counter_collection.csv", "k_adc_rowmax")
fetch_kb, write_kb = f["FETCH_SIZE"], w["WRITE_SIZE"]
out = {
    "kernel": "k_adc_rowmax_cf<1024> (adc_variant 5)",
    "workload": "bench.py default: 100 latents x 100k gallery, 8 latents per launch (13 launches per step)",
    "FETCH_SIZE_KB_per_launch_raw": fetch_kb, "WRITE_SIZE_KB_per_launch_raw": write_kb, "dispatches": nf,
    "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced read; both read "
                  "streams here (uint4 PQ codes, float4 LUT tiles) are 16 B/lane, so fetch bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 taken "
                  "as is (uncalibrated)",
    "fetch_bytes_per_launch": 2 * fetch_kb * 1024, "write_bytes_per_launch": write_kb * 1024,
    "traffic_bytes_per_launch": 2 * fetch_kb * 1024 + write_kb * 1024,
    "note": "reads are dominated by LUT tiles re-read (from Infinity Cache) once per gallery chunk; the PQ codes themselves are 1.3 GB and the "
            "row-max results 4.5 GB per launch; writes include the register spill of the 1024-thread build.  The kernel is bound by VALU/LDS "
            "issue, not by memory.",
    "sq_counters_per_launch": sq,
}
json.dump(out, open("profiles/adc_hbm_traffic.json", "w"), indent=1)
for name, pats in (("pmc_sq_summary", [f"{src}/pmc_sq/**/*counter_collection.csv"]),
                   ("pmc_hbm_summary", [f"{src}/pmc_fetch/**/*counter_collection.csv", f"{src}/pmc_write/**/*counter_collection.csv"])):
    txt = subprocess.run([sys.executable, "tools/pmc_summary.py"] + pats, capture_output=True, text=True).stdout
    open(f"profiles/{pre}_{name}.txt", "w").write(txt)
print("value", bench["value"], "| adc avg launch: live", bench["roofline"]["avg_launch_ms"], "ms, under rocprof (live events)", prof["roofline"]["avg_launch_ms"], "ms")
print({k: round(out[k] / 1e9, 2) for k in ("fetch_bytes_per_launch", "write_bytes_per_launch", "traffic_bytes_per_launch")}, "bank conflicts", sq.get("SQ_LDS_BANK_CONFLICT"))
