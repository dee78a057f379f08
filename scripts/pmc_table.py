"""Per-kernel averages of every counter found under <dir>/p*/ (rocprofv3 counter_collection csv)."""
import collections, csv, glob, json, sys
KEYS = {"k_spmv_St_pipe<3": "St_r", "k_cg_update_xp_u(": "upd_xp_u", "k_spmv_St_pipe<0": "St", "k_spmv_S_pipe<0": "S", "k_spmv_S_ell<0": "S_ell", "k_spmv_St_ell<3": "St_r_ell", "k_spmv_St_ell<0": "St_ell", "k_cg_update_r(": "upd_r", "k_cg_update_xp(": "upd_xp",
        "k_tile_gather": "gather", "k_tile_expand": "expand", "k_apply_fused": "fused", "k_tile_apply<0": "tile_apply"}
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for fn in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        for k, name in KEYS.items():
            if k in r["Kernel_Name"]:
                a = acc[name][r["Counter_Name"]]
                a[0] += 1; a[1] += float(r["Counter_Value"])
print(json.dumps({k: {c: round(v[1] / v[0], 1) for c, v in d.items()} | {"launches": max(v[0] for v in d.values())} for k, d in acc.items()}, indent=1))
