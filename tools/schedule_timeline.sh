#!/bin/bash
# One bench step of the default schedule under rocprofv3 --kernel-trace: the begin / end of every kernel of the step's launch groups, per queue
# (the evidence for DESIGN.md §8's "0.96 of the packing bound").   bash tools/schedule_timeline.sh <out dir under gpurun_out>
O=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/$O
rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alone > $R/$O/bench.json 2> $R/$O/bench.err
python - "$R/$O" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("afis::", "").replace("void ", ""), r["Queue_Id"]) for r in rows]
ks.sort()
# the timed step = the last N launches of the bound pass (N = launch groups per step: half of the trace's launches, warm-up + one timed step) and everything from the first of them on
b = [i for i, k in enumerate(ks) if k[2].startswith("k_adc_mfma")]
N = max(1, len(b) // 2)
first = b[-N]
# start a little earlier: the row-constant kernels of that group
while first > 0 and ks[first - 1][0] > ks[b[-N]][0] - 2_000_000: first -= 1
t0 = ks[first][0]
with open(out + "/timeline.txt", "w") as w:
    w.write("# begin_ms end_ms dur_ms queue kernel   (one bench step, default schedule; t = 0 at the step's first kernel; kernels under 0.05 ms omitted)\n")
    for s, e, n, q in ks[first:]:
        if e - s < 50_000: continue
        w.write(f"{(s - t0) / 1e6:9.2f} {(e - t0) / 1e6:9.2f} {(e - s) / 1e6:8.2f}  q{q:>3}  {n[:60]}\n")
print(open(out + "/timeline.txt").read()[:4000])
PY
