#!/usr/bin/env python3
"""The "state at HEAD" table of DESIGN.md section 3, generated from the committed round profile:
   profiles/rNN_bench256.json (bench.py line), rNN_kernel_stats_bench256.csv (rocprofv3 --kernel-trace --stats of one bench step),
   rNN_pmc_traffic.json (FETCH_SIZE / WRITE_SIZE passes).   usage: state_table.py [rNN]   -> markdown on stdout"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
P = lambda n: os.path.join(ROOT, "profiles", "%s_%s" % (tag, n))
b = json.load(open(P("bench256.json")))
rows = list(csv.DictReader(open(P("kernel_stats_bench256.csv"))))
pmc = json.load(open(P("pmc_traffic.json")))
k = dict(b["roofline"]["other_kernels"])
dom = "spmv_St_r" if "cg_update_xp_u" in k else "spmv_St"
k[dom] = {"ms": b["roofline"]["avg_launch_ms"], "algorithmic_bytes": b["roofline"]["algorithmic_bytes_per_launch"], "frac": b["roofline"]["frac"],
          "must_move_bytes": b["roofline"].get("must_move_bytes")}


def stat(sub):
    best = None
    for r in rows:
        if (sub in r["Name"] or (sub == "k_spmv_S_ell<0," and "k_spmv_S_ell2<" in r["Name"]) or (sub == "k_spmv_St_ell<3," and "k_spmv_St_ell2<" in r["Name"])) and (best is None or int(r["Calls"]) > int(best["Calls"])):
            best = r
    return best


table = [("spmv_St_r", "k_spmv_St_ell<3,", "`k_spmv_St_ell2<1>` (dominant; two units in flight per wave): `y = -S^T t - 1/2 uInv p` in registers, alpha, `r -= alpha y`, partials of `r.r`, `r.z`", "k_spmv_St_r"),
         ("spmv_S", "k_spmv_S_ell<0,", "`k_spmv_S_ell2<1>` (two units in flight per wave): `t = dt McInv (S p)`, partial of the active-face share of `p.Ap`", "k_spmv_S"),
         ("cg_update_xp_u", "k_cg_update_xp_u", "`k_cg_update_xp_u`: beta, `x += alpha p`, `p = D^-1 r + beta p`, partials of `x.x`, `sum uInv p^2`", "k_cg_update_xp_u"),
         ("tiles", "k_tile_apply<0,", "`k_tile_apply<0,64>`: `J^T`, 26x26 `BInv`, `J` per tile", "k_tile_apply")]
print("| Kernel (one launch each per PCG iteration) | stored bytes | must-move bytes | rocprof avg in the solve (calls) | bench.py in sequence | stored bytes / rocprof avg, of 8 TB/s | PMC traffic (x must-move) |")
print("|---|---|---|---|---|---|---|")
tot_us = 0.
for key, sub, what, pk in table:
    if key not in k:
        continue
    st = stat(sub)
    avg_us = float(st["AverageNs"]) / 1e3 if st else float("nan")
    tot_us += avg_us
    by = k[key]["algorithmic_bytes"]
    mm = k[key].get("must_move_bytes")
    tr = pmc.get(pk, {}).get("traffic_bytes_per_launch")
    print("| %s | %.3f GB | %s | %.1f us (%s) | %.3f ms | **%.2f** | %s |" % (
        what, by / 1e9, ("%.3f GB" % (mm / 1e9)) if mm else "= stored", avg_us, st["Calls"] if st else "-", k[key]["ms"], by / (avg_us * 1e-6) / 8e12,
        ("%.2f GB (x%.2f)" % (tr / 1e9, tr / (mm or by))) if tr else "-"))
print()
print("Sum of the four rocprof averages: %.3f ms per iteration; bench line: %.1f ms/step, %d iterations, solve %.1f ms, setup %.1f ms (`profiles/%s_bench256.json`)." % (
    tot_us / 1e3, b["value"], b["cg_iterations"], b["stage_ms"]["solve"], sum(v for kk, v in b["stage_ms"].items() if kk not in ("solve", "recover", "writeback")), tag))
