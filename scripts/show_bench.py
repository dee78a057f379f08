"""Key figures of a bench.py JSON line (file argument)."""
import json, sys
d = json.load(open(sys.argv[1]))
print("ms/step %.1f  iterations %d  solve %.1f  setup %.1f" % (d["value"], d["cg_iterations"], d["stage_ms"]["solve"], sum(v for k, v in d["stage_ms"].items() if k not in ("solve", "recover", "writeback"))))
r = d["roofline"]
print("dominant:", r["kernel"][:60], "| %.3f ms  %.0f GB/s  frac %.3f  alg %.3f GB  traffic %s" % (r["avg_launch_ms"], r["achieved"], r["frac"], r["algorithmic_bytes_per_launch"] / 1e9, r["traffic"]))
if "stream_runs" in r: print("stream runs:", {k: v for k, v in r["stream_runs"].items() if k != "note"})
for k, v in r["other_kernels"].items():
    print("  %-16s %.3f ms  alg %.3f GB  frac %.3f" % (k, v["ms"], v["algorithmic_bytes"] / 1e9, v["frac"]) + ("  replayed %.3f" % v["replayed_ms"] if "replayed_ms" in v else ""))
if "cpu_baseline" in d: print("cpu_baseline %.0f ms/step on %s cores" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]))
if "pcie_inclusive_ms" in d: print("pcie_inclusive_ms %.1f" % d["pcie_inclusive_ms"])
