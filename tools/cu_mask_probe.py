"""How do the kernels' times change when the context's stream may use only some of the CUs?  (hipExtStreamCreateWithCUMask through the AFIS_CU_MASK knob.)
python tools/cu_mask_probe.py [G] [Q]  — one process per mask (the mask is read at afis_create), same workload, stage times of the best of 3 steps."""
import subprocess, sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = sys.argv[1] if len(sys.argv) > 1 else "100000"; Q = sys.argv[2] if len(sys.argv) > 2 else "20"
def words(bits):
    w = [0] * 8
    for b in bits: w[b >> 5] |= 1 << (b & 31)
    return ",".join("%x" % x for x in w)
masks = {"all256": words(range(256)), "low192": words(range(192)), "low128": words(range(128)), "even128": words(range(0, 256, 2)), "mod4ne3_192": words([b for b in range(256) if b % 4 != 3]),
         "mod8lt6_192": words([b for b in range(256) if b % 8 < 6])}
for name, m in masks.items():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gallery", G, "--queries", Q, "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], env=dict(os.environ, AFIS_CU_MASK=m), capture_output=True, text=True)
    try:
        j = json.loads(r.stdout.strip().splitlines()[-1]); st = j["stage_ms_per_step"]
        print(name, {k: st[k] for k in ("adc_bound_ms", "adc_refine_ms", "tex_tail_ms", "cands_ms", "minu_graph_ms", "total_ms")}, flush=True)
    except Exception as e:
        print(name, "FAILED", r.returncode, r.stderr[-300:], flush=True)
