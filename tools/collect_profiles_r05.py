#!/usr/bin/env python
"""Turns a gpurun_out/<tag>/ bundle made by tools/profile_round5.sh into the committed artifacts under profiles/ AND into profiles/r05_tables.md — the one
generation of numbers DESIGN.md / INTEGRATION.md / include/afis_matcher.h quote (every figure there is a row of that file; unit: ms per LAUNCH GROUP of N latents,
N stated, or ms per step of 100 latents).  Usage (repo root): python tools/collect_profiles_r05.py <tag>"""
import csv, json, os, re, shutil, sys

tag = sys.argv[1]
src = f"gpurun_out/{tag}"
pre = sys.argv[2] if len(sys.argv) > 2 else "r05"                      # python tools/collect_profiles_r05.py <tag> r06: the same bundle layout (tools/profile_round6.sh adds the structured workload)
ROUND = {"r05": "round 5", "r06": "round 6"}.get(pre, pre)
last = lambda p: json.loads(open(p).read().strip().splitlines()[-1])
bench = last(f"{src}/bench.json"); prof = last(f"{src}/bench_profiled.json")
shutil.copy(f"{src}/bench.json", f"profiles/{pre}_bench_100x100k.json")
open(f"profiles/{pre}_bench_100x100k_under_rocprof.json", "w").write(json.dumps(prof) + "\n")
shutil.copy(f"{src}/kernel_stats.csv", f"profiles/{pre}_kernel_stats.csv")
shutil.copy(f"{src}/kernel_stats_back_to_back.csv", f"profiles/{pre}_kernel_stats_back_to_back.csv")
b2b = last(f"{src}/bench_back_to_back.json")
open(f"profiles/{pre}_bench_100x100k_back_to_back.json", "w").write(json.dumps(b2b) + "\n")
shutil.copy(f"{src}/pmc_sq_summary.txt", f"profiles/{pre}_pmc_sq_summary.txt")
shutil.copy(f"{src}/pmc_hbm_summary.txt", f"profiles/{pre}_pmc_hbm_summary.txt")
rstats = last(f"{src}/bench_refine_stats.json")
open(f"profiles/{pre}_bench_100x100k_refine_stats.json", "w").write(json.dumps(rstats) + "\n")
for f in ("latency_1x10k", "latency_1x100k"):
    open(f"profiles/{pre}_{f}.json", "w").write(json.dumps(last(f"{src}/{f}.json")) + "\n")
if os.path.exists(f"{src}/bench_8x1M.json"):
    open(f"profiles/{pre}_bench_8x1M_single_gpu.json", "w").write(json.dumps(last(f"{src}/bench_8x1M.json")) + "\n")


def blocks(path):
    out = {}
    for b in re.split(r"\n(?=\S)", open(path).read()):
        name = b.split(" dispatches")[0].replace("void ", "").strip()
        out[name] = {m.group(1): float(m.group(2)) for m in re.finditer(r"(\w+)\s+([\d.e+-]+) per dispatch", b)}
    return out


def pick(d, key):
    for k, v in d.items():
        if key in k: return v
    raise KeyError(key)


sqb, hbb = blocks(f"{src}/pmc_sq_summary.txt"), blocks(f"{src}/pmc_hbm_summary.txt")
# registers / scratch / LDS as the compiler reports them for the shipped flags (tools/kres.sh; rocprofv3's own columns count allocation granules)
import subprocess
res = {}
for f, fl in (("adc_mfma.hip", ["-fno-slp-vectorize", "-fno-honor-nans"]), ("adc_refine.hip", ["-fno-slp-vectorize"]), ("graph.hip", ["-fno-slp-vectorize"]), ("minu.hip", [])):
    for line in subprocess.run(["bash", "tools/kres.sh", f, *fl], capture_output=True, text=True).stdout.splitlines():
        w = line.split()
        if len(w) >= 11 and w[1] == "vgpr": res[w[0]] = {"vgpr": int(w[2]), "scratch": int(w[6]), "occupancy": int(w[8]), "lds": int(w[10])}
def resources(key):
    tagk = {"k_adc_mfma": "k_adc_mfmaILi2", "k_minu_cands_rt": "k_minu_cands_rt", "k_graph_texture": "k_graph_texture", "k_graph_minutiae": "k_graph_minutiae", "k_tex_refine": "k_tex_refine"}[key]
    for k, v in res.items():
        if tagk in k: return v
    return {"vgpr": -1, "scratch": -1, "occupancy": -1, "lds": -1}
ks = {r["kernel"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(f"{src}/kernel_stats_back_to_back.csv"))}     # the counter passes' schedule: kernels alone
ks_def = {r["kernel"].split("(")[0].replace("void ", ""): r for r in csv.DictReader(open(f"{src}/kernel_stats.csv"))}                # the default schedule: bound pass on half the CUs, minutiae stage beside it
KERNELS = [("k_adc_mfma", "S5-S6 bound pass (fp16 matrix cores)"), ("k_tex_refine", "S5-S7 selection by bounds + exact recomputation"), ("k_graph_texture", "S7-S9 texture lists"),
           ("k_minu_cands_rt", "S1-S3 minutiae candidates"), ("k_graph_minutiae", "S8a + S9 minutiae lists")]
launches = bench["stage_ms_per_step"] and int(round(bench["launch_groups_per_step"]))
q_per_launch = bench["config"]["queries"] / launches
valu = json.load(open("profiles/r03_valu_peak.json"))
vop3 = {}
for r in valu["results"]:
    if r["instruction"].startswith("v_max3_f32"):
        vop3 = r["cycles_per_wave64_instruction_per_simd"]
units = {}
for key, what in KERNELS:
    sq, k = pick(sqb, key), pick(ks, key)
    xcd = sq["GRBM_GUI_ACTIVE"] / 8
    simd, cu = 1024 * xcd, 256 * xcd
    ms = float(k["avg_ms"])
    n_valu = sq["SQ_INSTS_VALU"] - sq.get("SQ_INSTS_MFMA", 0.0)
    rr_ = resources(key)
    n_groups_def = int(pick(ks_def, "k_adc_mfma")["calls"])               # the list kernel runs twice per group in the default schedule (helper + joining instance): total time per group, not per call
    units[key] = {"what": what, "avg_launch_ms": round(ms, 2), "avg_launch_ms_default_schedule": round(float(pick(ks_def, key)["total_ms"]) / n_groups_def, 2), "calls": int(k["calls"]), "vgpr": rr_["vgpr"], "scratch": rr_["scratch"], "lds_bytes": rr_["lds"], "waves_per_simd_by_registers": rr_["occupancy"],
                  "clock_ghz": round(xcd / (ms * 1e-3) / 1e9, 3), "mfma_pipe_busy": round(sq.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / simd, 4),
                  "valu_instructions_per_launch": n_valu, "valu_instructions_per_simd_cycle": round(n_valu / simd, 4),
                  "lds_array_busy": round(sq["SQ_LDS_IDX_ACTIVE"] / cu, 4), "lds_bank_conflict_share_of_lds_cycles": round(sq["SQ_LDS_BANK_CONFLICT"] / max(1.0, sq["SQ_LDS_IDX_ACTIVE"]), 4),
                  "wave_cycles_waiting": round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 4), "fetch_KB_raw": pick(hbb, key).get("FETCH_SIZE"), "write_KB_raw": pick(hbb, key).get("WRITE_SIZE")}
b, r = units["k_adc_mfma"], units["k_tex_refine"]
sq = pick(sqb, "k_adc_mfma")
xcd = sq["GRBM_GUI_ACTIVE"] / 8
fetch_b = 2 * b["fetch_KB_raw"] * 1024; write_b = b["write_KB_raw"] * 1024
r_raw = r["fetch_KB_raw"] * 1024; r_fetch_lo = r_raw + 0.5 * write_b; r_fetch_hi = r_raw + 1.0 * write_b; r_write = r["write_KB_raw"] * 1024     # the records (= the bound pass's writes) are an 8 B/lane coalesced stream, tallied at half (profiles/r05_fetch_calibration.json), read once or twice from the fabric
alg_bytes = bench["roofline"]["hbm_view"]["alg_bytes_per_launch"]
bound_cus = bench["config"].get("bound_cus", 0)
out = {
    "round": ROUND, "kernel": "afis::k_adc_mfma<2> (adc_variant 9, default)",
    "schedule_of_these_counters": "one stream, the kernels of a launch group back to back (bench.py --bound-cus 0): every kernel is characterised ALONE; in the default schedule the bound pass runs on "
                                  f"{bound_cus} of the 256 CUs with the minutiae stage beside it (kernels[...].avg_launch_ms_default_schedule)",
    "workload": f"bench.py default: 100 latents x 100k gallery, launch groups cut by latent texture rows ({launches} launches per step, {q_per_launch:.1f} latents per launch on average)",
    "avg_launch_ms_rocprof_stats": b["avg_launch_ms"], "launches_profiled": b["calls"], "sq_counters_per_launch": sq,
    "FETCH_SIZE_KB_per_launch_raw": b["fetch_KB_raw"], "WRITE_SIZE_KB_per_launch_raw": b["write_KB_raw"],
    "correction": "FETCH_SIZE x 2 for the bound pass: its reads are 16 B / 4 B per lane coalesced streams (tile-aligned codes, point terms, B fragments), for which FETCH_SIZE reports half the bytes "
                  "(MI355X_MICROARCH.md section HBM; profiles/r03_fetch_calibration.json: 0.5000 on a known 8 GiB stream, 64.01 B per 4-byte gather); WRITE_SIZE x 1024 as is",
    "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b, "traffic_bytes_per_launch": fetch_b + write_b,
    "traffic_reading": "reads = the gallery's codes + point terms (20 B per rolled texture point) once per row group of 768 latent rows; writes = ONE 8-byte record per (latent row, rolled template) "
                       "(round 3: two, one per lane half: 17.9 GB per launch)",
    "stage_traffic": {"what": "bound pass + selection / recomputation kernel, HBM-side bytes per launch", "bound_pass": fetch_b + write_b,
                      "refine_fetch_x1": r_fetch_lo, "refine_fetch_x2": r_fetch_hi, "refine_write": r_write,
                      "refine_fetch_note": "FETCH_SIZE as reported + the untallied half of the record stream (8 B per lane, coalesced: tallied at exactly half, profiles/r05_fetch_calibration.json) for one (x1 key) or two (x2 key) reads of the records that reach the fabric; "
                                           "the 16-byte code-word gathers are 64-byte requests tallied in full",
                      "total_low": fetch_b + write_b + r_fetch_lo + r_write, "total_high": fetch_b + write_b + r_fetch_hi + r_write, "algorithmic_bytes_per_launch": alg_bytes,
                      "ratio_low": round((fetch_b + write_b + r_fetch_lo + r_write) / alg_bytes, 3), "ratio_high": round((fetch_b + write_b + r_fetch_hi + r_write) / alg_bytes, 3)},
    "fractions": {
        "effective_clock_ghz_while_profiled": b["clock_ghz"],
        "other_kernels_clock_ghz": {k: units[k]["clock_ghz"] for k in units if k != "k_adc_mfma"},
        "clock_reading": "the bound pass is the one kernel the chip does not hold its 2.4 GHz for: it runs power-limited (every other kernel of the step runs at 2.34-2.38 GHz)",
        "mfma_pipe_busy": b["mfma_pipe_busy"],
        "valu_issue": round(b["valu_instructions_per_launch"] * vop3["3_waves"] / (1024 * xcd), 4),
        "valu_issue_is": f"(SQ_INSTS_VALU - SQ_INSTS_MFMA) x {vop3['3_waves']:.2f} cycles per wave64 VOP3 at three waves per SIMD (profiles/r03_valu_peak.json) / SIMD-cycles",
        "lds_array_busy": b["lds_array_busy"], "lds_bank_conflict_share_of_lds_cycles": b["lds_bank_conflict_share_of_lds_cycles"],
        "wave_cycles_waiting_waitcnt_or_barrier": b["wave_cycles_waiting"],
        "valu_instructions_per_mfma": round(b["valu_instructions_per_launch"] / sq["SQ_INSTS_MFMA"], 2)},
    "kernels": units,
}
json.dump(out, open(f"profiles/{pre}_adc_counters.json", "w"), indent=1)
# shard projection
proj = {"what": "per-rank workloads of BASELINE.json configs[3] on ONE GPU at the kernels of this bundle: bench.py --queries 100 --gallery 12500 / 25000 / 50000 (the shard of a rank at N = 8 / 4 / 2). "
                "A PROJECTION: no multi-GPU node was available, no 1 -> 8 curve was measured.", "single_gpu_100k": {"ms_per_step": bench["ms_per_step"], "queries_per_s": bench["value"]}, "shards": []}
for g, n in ((50000, 2), (25000, 4), (12500, 8)):
    j = last(f"{src}/bench_shard_{g}.json")
    proj["shards"].append({"n_gpus": n, "shard_templates": g, "ms_per_step": j["ms_per_step"], "stage_ms_per_step": j["stage_ms_per_step"],
                           "projected_queries_per_s": round(100.0 / (j["ms_per_step"] * 1e-3 + 0.0005), 2),
                           "projected_efficiency_vs_linear": round(100.0 / (j["ms_per_step"] * 1e-3 + 0.0005) / (n * bench["value"]), 3)})
proj["exchange_allowance_s"] = 0.0005
json.dump(proj, open(f"profiles/{pre}_shard_projection.json", "w"), indent=1)

# ---- the tables the documents quote -------------------------------------------------------------------------------------------------
st = bench["stage_ms_per_step"]
L = []
L.append(f"# {ROUND.capitalize()} numbers (generated by tools/collect_profiles_r05.py from gpurun_out/{tag}; do not edit)\n")
L.append(f"Workload: bench.py default = {bench['config']['queries']} latents x {bench['config']['gallery']} templates, one MI355X; a step = {launches} launch groups of {q_per_launch:.1f} latents on average "
         f"({bench['config']['mean_latent_tex_rows']:.0f} latent texture rows, {bench['config']['mean_rolled_tex_points']:.0f} rolled texture points, {bench['config']['mean_rolled_minutiae']:.0f} rolled minutiae per template).\n")
L.append("## Step\n")
L.append("| quantity | value |\n|---|---|")
L.append(f"| queries/s (default schedule: bound pass on {bound_cus} CUs, minutiae stage beside it) | **{bench['value']:.2f}** |\n| ms per step (100 latents) | {bench['ms_per_step']:.1f} |\n| latency per latent (ms per step / 100) | {bench['ms_per_step'] / 100:.2f} |")
L.append(f"| queries/s, one stream / kernels back to back (`--bound-cus 0`, same box) | {b2b['value']:.2f} ({b2b['ms_per_step']:.1f} ms per step) |")
L.append("| stage times below: default schedule — they OVERLAP (the bound pass's is its own stream's, the minutiae stage's is what it took beside it): their sum exceeds the step | |")
for k_, lab in (("adc_bound_ms", "bound pass (on its CUs)"), ("adc_refine_ms", "selection + recomputation"), ("tex_tail_ms", "texture lists (S7-S9)"), ("cands_ms", "minutiae candidates (S1-S3), beside the bound pass"),
                ("minu_graph_ms", "minutiae lists (S8a, S9), beside the bound pass and after it"), ("lut_ms", "row constants"), ("fuse_ms", "fusion"), ("topk_ms", "rank lists")):
    L.append(f"| {lab}: ms per step, default schedule | {st[k_]:.1f} |")
sb = b2b["stage_ms_per_step"]
for k_, lab in (("adc_bound_ms", "bound pass"), ("adc_refine_ms", "selection + recomputation"), ("tex_tail_ms", "texture lists"), ("cands_ms", "minutiae candidates"), ("minu_graph_ms", "minutiae lists")):
    L.append(f"| {lab}: ms per step / share, kernels back to back | {sb[k_]:.1f} / {100 * sb[k_] / sb['total_ms']:.1f} % |")
rf = bench["roofline"]
L.append(f"| roofline (bound pass, default schedule): algorithmic flops / the WHOLE chip's fp16 matrix peak (2 500 TFLOP/s) | **{rf['frac']:.3f}** ({rf['achieved']:.0f} TFLOP/s) at {rf['avg_launch_ms']:.1f} ms per launch on {rf.get('cus_used', 256)} CUs; {rf.get('frac_of_cus_used', rf['frac']):.3f} of THOSE CUs' peak |")
L.append(f"| roofline (bound pass alone on the whole chip: `roofline.alone_on_the_chip` of the same run; `--bound-cus 0` run) | {(rf.get('alone_on_the_chip') or {}).get('frac', float('nan')):.3f} at {(rf.get('alone_on_the_chip') or {}).get('avg_launch_ms', float('nan')):.1f} ms; {b2b['roofline']['frac']:.3f} ({b2b['roofline']['achieved']:.0f} TFLOP/s) at {b2b['roofline']['avg_launch_ms']:.1f} ms per launch |")
ck = rf.get("measured_clock_ghz", {}); cka = (rf.get("alone_on_the_chip") or {}).get("measured_clock_ghz", {})
L.append(f"| shader clock measured INSIDE the run (s_memtime / s_memrealtime of sampled workgroups): bound pass / candidate kernel | default schedule {ck.get('bound_pass_ghz', 0):.2f} / {ck.get('candidate_kernel_ghz', 0):.2f} GHz; alone on the chip {cka.get('bound_pass_ghz', 0):.2f} / {cka.get('candidate_kernel_ghz', 0):.2f} GHz |")
mt = bench.get("minutiae_candidate_tasks", {})
L.append(f"| candidate tasks per step: small / medium / large class of the matrix-core kernel / any-shape kernel | {mt.get('fast_kernel_small_class')} / {mt.get('fast_kernel_medium_class')} / {mt.get('fast_kernel_large_class')} / {mt.get('any_shape_fallback_kernel')} (fallback share {mt.get('fallback_share')}) |")
L.append(f"| hbm_view.frac (bound + recomputation, 24 algorithmic B per rolled point per query / 8 TB/s), kernels back to back | {b2b['roofline']['hbm_view']['frac']:.4f} ({bench['roofline']['hbm_view']['frac']:.4f} in the default schedule, where the bound pass has half the chip) |")
L.append(f"| CPU baseline (oracle, {bench['cpu_baseline']['threads']} threads, {bench['cpu_baseline']['sample']}) | {bench['cpu_baseline']['pairs_per_s']:.0f} pairs/s = {bench['cpu_baseline']['value']:.4f} queries/s |")
L.append(f"| reference-faithful CPU loop (8 threads, static 16, re-parse per pair) | {bench['cpu_baseline']['reference_faithful_8_threads_static16_reparse_per_pair_queries_per_s']:.4f} queries/s |\n")
if os.path.exists(f"{src}/bench_wide.json"):
    w = last(f"{src}/bench_wide.json"); ws = w["stage_ms_per_step"]; wt = w["minutiae_candidate_tasks"]
    open(f"profiles/{pre}_bench_wide.json", "w").write(json.dumps(w) + "\n")
    if os.path.exists(f"{src}/kernel_stats_wide.csv"): shutil.copy(f"{src}/kernel_stats_wide.csv", f"profiles/{pre}_kernel_stats_wide.csv")
    L.append("## The off-envelope workload (`bench.py --workload wide`: rolled minutiae clip(N(130, 40), 20, 400), latent minutiae U{20..150}; NOT the headline)\n")
    L.append("| quantity | value |\n|---|---|")
    L.append(f"| queries/s / ms per step | **{w['value']:.2f}** / {w['ms_per_step']:.1f} ({w['config']['schedule'][:60]}...) |")
    L.append(f"| mean rolled / selected latent minutiae | {w['config']['mean_rolled_minutiae']:.1f} / {w['config'].get('mean_latent_minutiae_selected', 0):.1f} (headline: {bench['config']['mean_rolled_minutiae']:.1f} / {bench['config'].get('mean_latent_minutiae_selected', 0):.1f}): {w['config']['mean_rolled_minutiae'] * w['config'].get('mean_latent_minutiae_selected', 0) / max(1e-9, bench['config']['mean_rolled_minutiae'] * bench['config'].get('mean_latent_minutiae_selected', 0)):.2f} x the similarity cells per candidate task |")
    L.append(f"| candidates / minutiae lists / bound pass / recomputation / texture lists: ms per step | {ws['cands_ms']:.1f} / {ws['minu_graph_ms']:.1f} / {ws['adc_bound_ms']:.1f} / {ws['adc_refine_ms']:.1f} / {ws['tex_tail_ms']:.1f} |")
    L.append(f"| candidate tasks per step: small / medium / large class / any-shape kernel | {wt['fast_kernel_small_class']} / {wt['fast_kernel_medium_class']} / {wt['fast_kernel_large_class']} / {wt['any_shape_fallback_kernel']} (fallback share {wt['fallback_share']}) |")
    scale = w['config']['mean_rolled_minutiae'] * w['config'].get('mean_latent_minutiae_selected', 0) / max(1e-9, bench['config']['mean_rolled_minutiae'] * bench['config'].get('mean_latent_minutiae_selected', 0))
    scaled = sb['total_ms'] + (scale - 1.0) * sb['cands_ms']
    L.append(f"| the headline's back-to-back step with its candidate stage scaled by the cell ratio ({sb['total_ms']:.0f} + {scale - 1:.2f} x {sb['cands_ms']:.0f} ms) | {scaled:.0f} ms: the wide step is {w['ms_per_step'] / scaled:.2f} x that |\n")
L.append(f"## Kernels (rocprofv3 --kernel-trace --stats and --pmc passes of `bench.py --bound-cus 0`: every kernel alone on the chip; per launch group of {q_per_launch:.1f} latents)\n")
L.append("| kernel | avg launch ms alone (in the default schedule) | clock GHz | VGPR / scratch B / LDS B | MFMA pipe busy | vector instructions per SIMD-cycle x 4 (a PRICE of 4 cycles per wave64 instruction, not a unit fraction: full-rate adds / multiplies issue in 2, so it can exceed 1) | LDS busy (conflict share) | wave-cycles waiting |\n|---|---|---|---|---|---|---|---|")
for key, what in KERNELS:
    u = units[key]
    L.append(f"| `{key}` ({what}) | {u['avg_launch_ms']:.1f} ({u['avg_launch_ms_default_schedule']:.1f}) | {u['clock_ghz']:.2f} | {u['vgpr']} / {u['scratch']} / {u['lds_bytes']} | {u['mfma_pipe_busy']:.2f} | {4 * u['valu_instructions_per_simd_cycle']:.2f} | "
             f"{u['lds_array_busy']:.2f} ({u['lds_bank_conflict_share_of_lds_cycles']:.2f}) | {u['wave_cycles_waiting']:.2f} |")
s_ = out["stage_traffic"]
L.append(f"\n## HBM-side traffic of the ADC stage per launch (PMC passes; bound pass FETCH_SIZE x 2)\n")
L.append("| term | GB |\n|---|---|")
L.append(f"| bound pass reads | {fetch_b / 1e9:.1f} |\n| bound pass writes (one 8-byte record per row and template) | {write_b / 1e9:.1f} |\n| recomputation reads (FETCH_SIZE + the untallied half of the record stream, read once ... twice from the fabric) | {r_fetch_lo / 1e9:.1f} ... {r_fetch_hi / 1e9:.1f} |"
         f"\n| recomputation writes | {r_write / 1e9:.1f} |\n| stage total | {s_['total_low'] / 1e9:.1f} ... {s_['total_high'] / 1e9:.1f} |\n| algorithmic (24 B per rolled point per latent) | {alg_bytes / 1e9:.1f} |"
         f"\n| ratio | {s_['ratio_low']:.2f} ... {s_['ratio_high']:.2f} |\n")
rs = rstats["refine_stats"]
L.append("## Selection / recomputation counters (one step, mf_stats)\n")
L.append(f"rows evaluated {100 * rs['rows_evaluated'] / rs['rows']:.1f} % of {rs['rows']:.3g}; cells per evaluated row {rs['cells_evaluated'] / rs['rows_evaluated']:.3f}; rows evaluated over every point "
         f"{100 * rs['rows_evaluated_in_full'] / rs['rows_evaluated']:.3f} %; rows outside their bounds {rs['bound_violations']}.\n")
L.append("## Single latent (BASELINE.json configs[1]; bench.py --queries 1)\n")
L.append("| gallery | ms per latent | lut / bound / recompute / texture lists / candidates / minutiae lists / rank list (ms) |\n|---|---|---|")
for f, gsz in (("latency_1x10k", "10 000"), ("latency_1x100k", "100 000")):
    j = last(f"{src}/{f}.json"); s2 = j["stage_ms_per_step"]
    L.append(f"| {gsz} | {j['ms_per_step']:.2f} | {s2['lut_ms']:.2f} / {s2['adc_bound_ms']:.2f} / {s2['adc_refine_ms']:.2f} / {s2['tex_tail_ms']:.2f} / {s2['cands_ms']:.2f} / {s2['minu_graph_ms']:.2f} / {s2['topk_ms']:.2f} |")
L.append("\n## Shards (the per-rank workload of an N-GPU job on ONE GPU; projection, not a scaling measurement)\n")
L.append("| N | shard templates | ms per step | projected queries/s | of linear |\n|---|---|---|---|---|")
for s3 in proj["shards"]:
    L.append(f"| {s3['n_gpus']} | {s3['shard_templates']} | {s3['ms_per_step']:.1f} | {s3['projected_queries_per_s']:.1f} | {s3['projected_efficiency_vs_linear']:.3f} |")
# ---- round 6: the structured workload (bench.py --workload structured --dup 0 / 10 / 30; tools/profile_round6.sh) ----
st_runs = {d: last(f"{src}/bench_structured_dup{d}.json") for d in (0, 10, 30) if os.path.exists(f"{src}/bench_structured_dup{d}.json")}
if st_runs:
    doc = {"what": "bench.py --workload structured (host/synth_structured.py: texture points on the extractor's 16-px grid inside a foreground blob, smooth ridge flow, descriptors near a shared manifold PQ-encoded "
                   "afterwards; --dup = the share of a rolled template's texture points whose 16-byte code vector occurs at another point of the template too) at BASELINE.json configs[2]'s sizes, one MI355X, "
                   "same box and bundle as the headline line beside it.  NOT the headline.",
           "headline_same_box": {"queries_per_s": bench["value"], "ms_per_step": bench["ms_per_step"], "stage_ms_per_step": bench["stage_ms_per_step"], "stage_ms_per_step_back_to_back": bench.get("stage_ms_per_step_back_to_back")},
           "runs": {}}
    L.append("\n## The structured workload (`bench.py --workload structured`; NOT the headline) on the same box\n")
    L.append("| repeated code vectors | queries/s | ms per step (x the headline's) | back to back: bound / recompute / texture lists / candidates / minutiae lists (ms per step) | rows evaluated per pair | cells per evaluated row | evaluated rows over every point | non-mates scoring > 0 | candidate tasks to the any-shape kernel |\n|---|---|---|---|---|---|---|---|---|")
    hb = bench.get("stage_ms_per_step_back_to_back") or {}
    L.append(f"| (headline, i.i.d. templates) | {bench['value']:.2f} | {bench['ms_per_step']:.1f} (1.00) | {hb.get('adc_bound_ms', 0):.0f} / {hb.get('adc_refine_ms', 0):.0f} / {hb.get('tex_tail_ms', 0):.0f} / {hb.get('cands_ms', 0):.0f} / {hb.get('minu_graph_ms', 0):.0f} | {rs['rows_evaluated'] / rs['pairs']:.1f} | {rs['cells_evaluated'] / rs['rows_evaluated']:.3f} | {100 * rs['rows_evaluated_in_full'] / rs['rows_evaluated']:.3f} % | | {mt.get('any_shape_fallback_kernel')} |")
    for d, j in sorted(st_runs.items()):
        open(f"profiles/{pre}_bench_structured_dup{d}.json", "w").write(json.dumps(j) + "\n")
        jb = j.get("stage_ms_per_step_back_to_back") or {}; jr = j.get("refine_stats") or {}; js = j.get("score_stats") or {}
        doc["runs"][f"dup{d}"] = {"queries_per_s": j["value"], "ms_per_step": j["ms_per_step"], "step_time_relative_to_headline": round(j["ms_per_step"] / bench["ms_per_step"], 3),
                                   "measured_share_of_repeated_code_vectors": j["config"].get("structured_dup_share_measured"), "stage_ms_per_step": j["stage_ms_per_step"], "stage_ms_per_step_back_to_back": jb,
                                   "refine_stats": jr, "score_stats": js, "minutiae_candidate_tasks": j["minutiae_candidate_tasks"], "rank1_hits": j["rank1_hits"], "power": j.get("power")}
        L.append(f"| {100 * (j['config'].get('structured_dup_share_measured') or 0):.1f} % (--dup {d}) | **{j['value']:.2f}** | {j['ms_per_step']:.1f} ({j['ms_per_step'] / bench['ms_per_step']:.3f}) | {jb.get('adc_bound_ms', 0):.0f} / {jb.get('adc_refine_ms', 0):.0f} / {jb.get('tex_tail_ms', 0):.0f} / {jb.get('cands_ms', 0):.0f} / {jb.get('minu_graph_ms', 0):.0f} | "
                 f"{jr.get('evaluated_rows_per_pair', 0):.1f} | {jr.get('cells_per_evaluated_row', 0):.3f} | {100 * jr.get('share_of_evaluated_rows_in_full', 0):.3f} % | {100 * js.get('non_mates_with_positive_score', 0):.1f} % | {j['minutiae_candidate_tasks']['any_shape_fallback_kernel']} |")
    json.dump(doc, open(f"profiles/{pre}_bench_structured.json", "w"), indent=1)
    if os.path.exists(f"{src}/kernel_stats_structured.csv"): shutil.copy(f"{src}/kernel_stats_structured.csv", f"profiles/{pre}_kernel_stats_structured_back_to_back.csv")
pw = bench.get("power") or {}
if pw.get("timed_schedule") or pw.get("kernels_back_to_back"):
    L.append("\n## Board power (bench.py: a host thread samples the GPU's hwmon power file during the timed steps)\n")
    L.append("| schedule | mean W | max W | J per query |\n|---|---|---|---|")
    for k_, lab in (("timed_schedule", f"default schedule (bound pass on {bound_cus} CUs)"), ("kernels_back_to_back", "kernels back to back")):
        if pw.get(k_): L.append(f"| {lab} | {pw[k_]['watts_mean']:.0f} | {pw[k_]['watts_max']:.0f} | {pw[k_]['joules_per_query']:.2f} |")
if os.path.exists("profiles/r04_cli_scale.json"):       # (round 4's measurement: the host side of `match` did not change in round 5)
    c = json.load(open("profiles/r04_cli_scale.json"))
    L.append(f"\n## `match` end to end, {c['Q']} latents x {c['G']} templates (tools/cli_scale_r04.py, ROUND 4's run; wall seconds, stages from the process's own clock; measured at the end of the round, after the host-side staging / allocation work — the kernels are the bundle's)\n")
    L.append("| run | wall s | scan | load + parse | commit + upload | latents | search | write |\n|---|---|---|---|---|---|---|---|")
    for name, rr in c["runs"].items():
        sm = rr["stages_ms"]
        L.append(f"| {name} | {rr['wall_s']:.2f} | {sm['scan'] / 1e3:.2f} | {sm['load'] / 1e3:.2f} | {sm['commit'] / 1e3:.2f} | {sm['latents'] / 1e3:.3f} | {sm['search'] / 1e3:.2f} | {sm['write'] / 1e3:.2f} |")
    L.append(f"\nReference-faithful CPU loop on the same host (8 threads, every rolled file re-parsed per pair, bounded sample): {c['cpu_reference_faithful_pairs_per_s_8_threads']:.0f} pairs/s "
             f"= {c['cpu_reference_faithful_extrapolated_s_for_this_job']:.0f} s for this job (extrapolated).\n")
if os.path.exists(f"{src}/bench_8x1M.json"):
    j = last(f"{src}/bench_8x1M.json")
    L.append(f"\n## 8 latents x 1 M templates resident on one GPU\n\n{j['ms_per_step']:.0f} ms per step ({j['value']:.2f} queries/s); stages {j['stage_ms_per_step']}.\n")
open(f"profiles/{pre}_tables.md", "w").write("\n".join(L) + "\n")
print("\n".join(L))
