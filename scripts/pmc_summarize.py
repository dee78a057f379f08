"""Summarise the FETCH_SIZE / WRITE_SIZE passes of scripts/profile_round.sh into per-kernel HBM traffic per launch.
FETCH_SIZE is doubled (gfx950 tallies 128-B read requests as 64 B: MI355X_MICROARCH.md, HBM section; re-calibrated here
on k_cg_update_r, which reads exactly three vectors of 8 n bytes and writes one); WRITE_SIZE is exact.  Counter unit: KiB."""
import collections
import csv
import glob
import json
import sys

out_dir = sys.argv[1]
KEYS = {"k_spmv_St_pipe<3": "k_spmv_St_r", "k_cg_update_xp_u(": "k_cg_update_xp_u", "k_tile_apply<0": "k_tile_apply", "k_spmv_St_pipe<0": "k_spmv_St", "k_spmv_S_pipe<0": "k_spmv_S", "k_spmv_S_ell<0": "k_spmv_S", "k_spmv_S_ell2<": "k_spmv_S", "k_spmv_St_ell<3": "k_spmv_St_r", "k_spmv_St_ell2<": "k_spmv_St_r", "k_spmv_St_ell<0": "k_spmv_St", "k_cg_update_r(": "k_cg_update_r", "k_cg_update_xp(": "k_cg_update_xp",
        "k_tile_gather": "k_tile_gather", "k_tile_expand": "k_tile_expand"}


def collect(sub, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for fn in glob.glob(f"{out_dir}/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != counter:
                continue
            for k, name in KEYS.items():
                if k in r["Kernel_Name"]:
                    acc[name][0] += 1
                    acc[name][1] += float(r["Counter_Value"])
    return {k: (v[0], v[1] / v[0]) for k, v in acc.items() if v[0]}


f, w = collect("pmc_fetch", "FETCH_SIZE"), collect("pmc_write", "WRITE_SIZE")
res = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (bench.py --maxit 20, 256^3 cavity). Counter unit KiB. "
               "FETCH_SIZE is doubled (gfx950: 128-B requests tallied as 64 B; k_cg_update_r reads exactly 3 vectors of 8 n bytes: "
               "see its entry). WRITE_SIZE is exact. traffic = 2*FETCH + WRITE, per launch."}
for k in KEYS.values():
    if k in f and k in w:
        res[k] = {"launches": f[k][0], "FETCH_SIZE_KiB_avg": f[k][1], "WRITE_SIZE_KiB_avg": w[k][1],
                  "traffic_bytes_per_launch": (2 * f[k][1] + w[k][1]) * 1024}
print(json.dumps(res, indent=1))
