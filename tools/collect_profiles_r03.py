#!/usr/bin/env python
"""Turns a gpurun_out/<tag>/ bundle made by tools/profile_round3.sh into the committed artifacts under profiles/.
Usage (repo root): python tools/collect_profiles_r03.py <tag>"""
import csv, json, re, shutil, sys

tag = sys.argv[1]
src = f"gpurun_out/{tag}"
pre = "r03"
last = lambda p: json.loads(open(p).read().strip().splitlines()[-1])
bench = last(f"{src}/bench.json"); prof = last(f"{src}/bench_profiled.json")
shutil.copy(f"{src}/bench.json", f"profiles/{pre}_bench_100x100k.json")
open(f"profiles/{pre}_bench_100x100k_under_rocprof.json", "w").write(json.dumps(prof) + "\n")
shutil.copy(f"{src}/kernel_stats.csv", f"profiles/{pre}_kernel_stats.csv")
shutil.copy(f"{src}/pmc_sq_summary.txt", f"profiles/{pre}_pmc_sq_summary.txt")
shutil.copy(f"{src}/pmc_hbm_summary.txt", f"profiles/{pre}_pmc_hbm_summary.txt")
open(f"profiles/{pre}_bench_100x100k_variant8_lds_table_bound_pass.json", "w").write(json.dumps(last(f"{src}/bench_variant8.json")) + "\n")
open(f"profiles/{pre}_bench_100x100k_refine_stats.json", "w").write(json.dumps(last(f"{src}/bench_refine_stats.json")) + "\n")


def counters(path, kernel):
    txt = open(path).read()
    blk = txt.split(kernel + " dispatches")[1].split(" dispatches")[0]
    return {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)\s+([\d.e+-]+) per dispatch", blk)}


sq = counters(f"{src}/pmc_sq_summary.txt", "afis::k_adc_mfma"); hb = counters(f"{src}/pmc_hbm_summary.txt", "afis::k_adc_mfma")
rsq = counters(f"{src}/pmc_sq_summary.txt", "afis::k_tex_refine"); rhb = counters(f"{src}/pmc_hbm_summary.txt", "afis::k_tex_refine")
ks = {r["kernel"].split("(")[0]: r for r in csv.DictReader(open(f"{src}/kernel_stats.csv"))}
valu = json.load(open("profiles/r03_valu_peak.json"))
vop3_3w = None
for r in valu["results"]:
    if r["instruction"].startswith("v_max3_f32"):
        vop3_3w = r["cycles_per_wave64_instruction_per_simd"]["3_waves"]                                       # the kernel runs three waves per SIMD
xcd_cycles = sq["GRBM_GUI_ACTIVE"] / 8                                      # the counter is summed over the 8 XCDs
simd_cycles = 1024 * xcd_cycles; cu_cycles = 256 * xcd_cycles
n_valu = sq["SQ_INSTS_VALU"] - sq["SQ_INSTS_MFMA"]                          # SQ_INSTS_VALU counts the MFMAs too
fetch_b = 2 * hb["FETCH_SIZE"] * 1024; write_b = hb["WRITE_SIZE"] * 1024
out = {
    "round": "round 3", "kernel": "afis::k_adc_mfma (adc_variant 9, default)",
    "workload": "bench.py default: 100 latents x 100k gallery, launch groups cut by latent texture rows (6 launches per step)",
    "avg_launch_ms_rocprof_stats": float(ks["afis::k_adc_mfma"]["avg_ms"]), "launches_profiled": int(ks["afis::k_adc_mfma"]["calls"]),
    "sq_counters_per_launch": sq, "FETCH_SIZE_KB_per_launch_raw": hb["FETCH_SIZE"], "WRITE_SIZE_KB_per_launch_raw": hb["WRITE_SIZE"],
    "correction": "FETCH_SIZE x 2: the kernel's reads are 16 B / 4 B per lane coalesced streams (pair-aligned codes, point terms, B fragments), for which FETCH_SIZE reports half the bytes "
                  "(profiles/r03_fetch_calibration.json: 0.5000 on an 8 GiB known stream); WRITE_SIZE x 1024 as is (8-byte records, 256-byte segments per half wave)",
    "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b, "traffic_bytes_per_launch": fetch_b + write_b,
    "traffic_reading": "reads = the gallery's codes + point terms (20 B per rolled texture point = 1.6 GB per pass) once per row group of 768 latent rows; writes = the bound pass's records, "
                       "16 B per (latent row, rolled template)",
    "fractions": {
        "effective_clock_ghz_while_profiled": round(xcd_cycles / (float(ks["afis::k_adc_mfma"]["avg_ms"]) * 1e-3) / 1e9, 3),
        "mfma_pipe_busy": round(sq["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4),
        "valu_issue": round(n_valu * vop3_3w / simd_cycles, 4),
        "valu_issue_is": f"(SQ_INSTS_VALU - SQ_INSTS_MFMA) x {vop3_3w:.2f} cycles per wave64 VOP3 at three waves per SIMD (profiles/r03_valu_peak.json) / SIMD-cycles",
        "lds_array_busy": round(sq["SQ_LDS_IDX_ACTIVE"] / cu_cycles, 4),
        "lds_bank_conflict_share_of_lds_cycles": round(sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"], 4),
        "wave_cycles_waiting_waitcnt_or_barrier": round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 4),
        "wave_cycles_stalled_at_issue": round(sq["SQ_WAIT_INST_ANY"] / sq["SQ_WAVE_CYCLES"], 4),
        "valu_instructions_per_mfma": round(n_valu / sq["SQ_INSTS_MFMA"], 2)},
    "refine_kernel": {"kernel": "afis::k_tex_refine", "avg_launch_ms_rocprof_stats": float(ks["afis::k_tex_refine"]["avg_ms"]),
                      "fetch_bytes_per_launch_uncorrected": rhb["FETCH_SIZE"] * 1024, "write_bytes_per_launch": rhb["WRITE_SIZE"] * 1024, "sq_counters_per_launch": rsq},
}
json.dump(out, open(f"profiles/{pre}_adc_counters.json", "w"), indent=1)
# shard projection
proj = {"what": "per-rank workloads of BASELINE.json configs[3] on ONE GPU at the final kernels of this bundle: bench.py --queries 100 --gallery 12500 / 25000 / 50000 (the shard of a rank at N = 8 / 4 / 2). "
                "A PROJECTION: no multi-GPU node was available, no 1 -> 8 curve was measured.", "single_gpu_100k": {"ms_per_step": bench["ms_per_step"], "queries_per_s": bench["value"]}, "shards": []}
for g, n in ((50000, 2), (25000, 4), (12500, 8)):
    j = last(f"{src}/bench_shard_{g}.json")
    proj["shards"].append({"n_gpus": n, "shard_templates": g, "ms_per_step": j["ms_per_step"], "stage_ms_per_step": j["stage_ms_per_step"],
                           "projected_queries_per_s": round(100.0 / (j["ms_per_step"] * 1e-3 + 0.0005), 2),
                           "projected_efficiency_vs_linear": round(100.0 / (j["ms_per_step"] * 1e-3 + 0.0005) / (n * bench["value"]), 3)})
proj["exchange_allowance_s"] = 0.0005
proj["exchange_allowance_is"] = "one all-gather of 29 KB per rank, latency-bound: 0.5 ms allowed (torch.distributed / RCCL all_gather of this size on one node is tens of microseconds; the 1-rank ncclAllGather of csrc/rank_exchange.cpp including its staging copies measures below that)"
proj["imbalance_bound"] = "shards are cut by rolled texture points (host/sharding.py::shard_bounds): the largest shard's point count exceeds the mean by at most one template (< 0.01 % at 12.5k templates)"
json.dump(proj, open(f"profiles/{pre}_shard_projection.json", "w"), indent=1)
print("value", bench["value"], "ms/step", bench["ms_per_step"], "| k_adc_mfma avg: live", bench["roofline"]["avg_launch_ms"], "rocprof", ks["afis::k_adc_mfma"]["avg_ms"], "| live under rocprof", prof["roofline"]["avg_launch_ms"])
print(json.dumps(out["fractions"], indent=1)); print(json.dumps(proj["shards"], indent=1)[:900])
