#!/usr/bin/env python
"""What would a NARROWER operand type cost the bound pass of adc_variant 9?  (CPU only.)

k_adc_mfma bounds every (latent row, rolled point) similarity with G = a~ . b~ - |b|^2 / 2 in fp16 operands; the price of the rounding is the per-row tolerance
Tg (adc_refine.hip::k_mf_rows): every point whose G lies within Tg of the row's best is a candidate the recomputation kernel must evaluate exactly, and a row with
more than four such cells in a lane half is evaluated over ALL points of the template ("many").  fp8 (e4m3) operands run the matrix pipe at twice the fp16 rate
and halve the LDS operand bytes — but Tg grows with the operand rounding unit.  This tool prices that: for sampled latent rows against sampled rolled templates
it computes G exactly (float64), derives Tg for each operand format FROM THE SAME FORMULA as k_mf_rows (sum over sub-quantizers of the largest rounding residual
against any codeword, both sides), and counts what the refine would have to do:
    cells per evaluated row, share of rows with >= 5 cells within Tg ("many": full evaluation), rows that pass the top-200 bound selection per pair.
Formats: fp16 (shipped), bf16, e4m3 with a per-row / per-codeword power-of-two scale (what v_mfma_scale_f32_32x32x64_f8f6f4's block scales give),
and e4m3 hi + lo (two-term split: 3 MFMAs = 1.5 x the fp16 cost).
usage: python tools/bound_precision_costing.py [--workload headline|structured] [--out profiles/r06_bound_pass_precision.json]
"""
import argparse, importlib, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
T = importlib.import_module("msu-latentafis_amd.host.templates"); S = importlib.import_module("msu-latentafis_amd.host.synth"); SS = importlib.import_module("msu-latentafis_amd.host.synth_structured")


def q_mant(x, mant_bits, emin):
    """round to nearest (ties to even) onto a binary float grid with `mant_bits` explicit mantissa bits and smallest normal exponent emin (subnormals below it)"""
    x = np.asarray(x, np.float64)
    m, e = np.frexp(x)                                   # x = m 2^e, 0.5 <= |m| < 1
    e = np.maximum(e, emin + 1)
    step = np.ldexp(1.0, e - 1 - mant_bits)
    return np.round(x / step) * step                     # np.round: ties to even


FORMATS = {
    "fp16": dict(q=lambda v: q_mant(v, 10, -14), mfma_cost=1.0),
    "bf16": dict(q=lambda v: q_mant(v, 7, -126), mfma_cost=1.0),
    "e4m3_scaled": dict(q=None, mfma_cost=0.5),          # per-vector power-of-two scale to the top of the e4m3 range, 3 mantissa bits
    "e4m3_hi_lo": dict(q=None, mfma_cost=1.5),           # x ~ hi + lo, both e4m3 with their own scales: three MFMAs (hi hi, hi lo, lo hi)
}


def q_e4m3_scaled(v, axis=-1):
    amax = np.maximum(np.abs(v).max(axis=axis, keepdims=True), 1e-30)
    sc = np.exp2(np.floor(np.log2(448.0 / amax)))        # power-of-two block scale (one per vector: an MX block of 32 would be finer; 6-element sub-vectors do not get their own)
    return q_mant(v * sc, 3, -6) / sc


def quant(name, v):
    if name == "e4m3_scaled": return q_e4m3_scaled(v)
    if name == "e4m3_hi_lo":
        hi = q_e4m3_scaled(v); lo = q_e4m3_scaled(v - hi)
        return hi + lo
    return FORMATS[name]["q"](v)


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="headline", choices=["headline", "structured"]); ap.add_argument("--out", default="")
    ap.add_argument("--latents", type=int, default=2); ap.add_argument("--gallery", type=int, default=40)
    a = ap.parse_args()
    cb = T.Codebook.load(os.path.join(ROOT, "tests", "golden", "codebook_EmbeddingSize_96_stride_16_subdim_6.dat"))
    if a.workload == "structured":
        sg = SS.DUP_SIGMA[10]; lats = SS.make_structured_latents(3, a.latents, sigma=sg); gal = SS.make_packed_gallery_structured(3, a.gallery, cb, sigma=sg)
    else:
        lats = S.make_latents(3, a.latents); gal = S.make_packed_gallery(3, a.gallery, cb)
    W = cb.words.astype(np.float64)                                         # [16][256][6]
    out = {"what": __doc__.split("\n\n")[0].strip(), "workload": a.workload, "latents": a.latents, "gallery_templates": a.gallery, "formats": {}}
    for name, f in FORMATS.items():
        Wq = quant(name, W.reshape(16 * 256, 6)).reshape(16, 256, 6) if name.startswith("e4m3") else quant(name, W)
        dW = W - Wq
        st = {"rows": 0, "cells_within_T": 0, "rows_many": 0, "pairs": 0, "rows_active": 0, "Tg": []}
        for L in lats:
            A = L.tex[0].des.astype(np.float64)[:1000]                      # [n][96]
            Aq = quant(name, A) if not name.startswith("e4m3") else quant(name, A)          # e4m3: one scale per row
            dA = (A - Aq).reshape(len(A), 16, 6); Aq6 = Aq.reshape(len(A), 16, 6)
            # k_mf_rows: P = sum_m max_c |da_m . cw_mc|, Q = sum_m max_c |a~_m . db_mc|; Tg = 2 Eg + (small terms), Eg = 1.001 (P + Q) + accumulation
            P = np.abs(np.einsum("nmd,mcd->nmc", dA, W)).max(-1).sum(-1)
            Q = np.abs(np.einsum("nmd,mcd->nmc", Aq6, dW)).max(-1).sum(-1)
            Tg = 2.0 * 1.001 * (P + Q) + 1e-4                               # the fp32 accumulation / index-perturbation terms are ~1e-4 at these magnitudes
            Es = Tg
            c = 6.0 - (A * A).sum(1)
            for g in range(gal.G):
                lo_, hi_ = int(gal.tex_off[g]), int(gal.tex_off[g + 1])
                codes = gal.tex_codes[lo_:hi_][:1000]
                _, first = np.unique(codes, axis=0, return_index=True)      # first occurrence of every code vector (the bound pass masks the repeats)
                codes = codes[np.sort(first)]
                B = W[np.arange(16)[None, :], codes, :].reshape(len(codes), 96)
                G = A @ B.T - 0.5 * (B * B).sum(1)[None, :]
                best = G.max(1)
                within = (G >= (best - Tg)[:, None]).sum(1)
                mid = c + 2 * best
                lo_b, hi_b = mid - Es, mid + Es
                if len(A) > 200:
                    thr = np.sort(lo_b)[-200]
                    act = hi_b >= thr
                else:
                    act = np.ones(len(A), bool)
                st["pairs"] += 1; st["rows"] += len(A); st["rows_active"] += int(act.sum())
                st["cells_within_T"] += int(np.minimum(within[act], 4).sum()); st["rows_many"] += int((within[act] >= 5).sum())
                st["cells_if_many_rows_are_evaluated_in_full"] = st.get("cells_if_many_rows_are_evaluated_in_full", 0) + int(np.where(within[act] >= 5, len(codes), within[act]).sum())
            st["Tg"] += list(Tg)
        tg = np.array(st.pop("Tg"))
        out["formats"][name] = {"mfma_cost_relative_to_fp16": f["mfma_cost"], "Tg_median": float(np.median(tg)), "Tg_p90": float(np.percentile(tg, 90)),
                                "rows_evaluated_per_pair": round(st["rows_active"] / st["pairs"], 1),
                                "share_of_evaluated_rows_with_5_or_more_cells_within_T": round(st["rows_many"] / max(1, st["rows_active"]), 4),
                                "cells_per_evaluated_row_with_full_rows_counted": round(st["cells_if_many_rows_are_evaluated_in_full"] / max(1, st["rows_active"]), 2)}
    base = out["formats"]["fp16"]["cells_per_evaluated_row_with_full_rows_counted"] * out["formats"]["fp16"]["rows_evaluated_per_pair"]
    for name, r in out["formats"].items():
        r["refine_cells_relative_to_fp16"] = round(r["cells_per_evaluated_row_with_full_rows_counted"] * r["rows_evaluated_per_pair"] / base, 2)
    print(json.dumps(out, indent=1))
    if a.out: json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
