#!/usr/bin/env python
"""Throughput of the PQ encoder (afis_pq_encode): points/s end to end (host pointers in, host pointers out, so PCIe-inclusive).
The kernel-only time comes from running this script under `rocprofv3 --kernel-trace --stats` (k_pq_encode)."""
import importlib, json, sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
M = importlib.import_module("msu-latentafis_amd.host.matcher")
cbb = open(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"), "rb").read()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
rng = np.random.default_rng(0)
des = rng.standard_normal((n, 96), dtype=np.float32) * 0.1
m = M.Matcher(cbb)
m.pq_encode(des[:100000])
t0 = time.perf_counter(); reps = 3
for _ in range(reps):
    codes = m.pq_encode(des)
dt = (time.perf_counter() - t0) / reps
flop = n * 16 * 256 * 17.0
print(json.dumps({"metric": "PQ-encoded texture points/s (host to host)", "value": round(n / dt, 1), "points": n, "s_per_call": round(dt, 4),
                  "algorithmic_fp32_ops": flop, "note": "17 fp32 ops per (point, sub-quantizer, codeword); kernel-only time: rocprofv3 stats of k_pq_encode"}))
