#!/usr/bin/env bash
# The first lease of a multi-GPU MI355X node should yield the 1 -> 8 GPU curve AND its explanation in ONE run.
#
#   scripts/first_rccl_run.sh [OUTDIR] [GPU counts ...]         default: gpurun_out/first_rccl  "1 2 4 8"
#
# Per GPU count N it runs, one process per GPU (python -m torch.distributed.run, rendezvous on 127.0.0.1; gloo only carries the RCCL unique
# id, the barriers and the max of the wall time — the data path is the library's own RCCL communicator):
#   (1) the driver's default invocation            bench.py --gpus N --steps 5 --warmup 2
#       -> headline (256^3 per GPU, weak: z-slabs) + its `multi_gpu` block + the `strong_512` block (the 512^3 coil of BASELINE config 4 cut N
#          ways: single domain, 2 slabs, 2x2x1, 2x2x2 bricks) — the north star's curve from the invocation the driver itself uses;
#   (2) the strong series on its own, bricks:      bench.py --gpus N --scaling strong --scene coil --res 512 --bricks DIMS --steps 5 --warmup 2
#   (3) N = 8 also as 8 z-slabs, and the 256^3 spheres scene of BASELINE config 5 as 2x2x2 bricks.
# Kept per run under OUTDIR/<tag>/: line.json (the ONE JSON line of rank 0), stderr.txt of the launcher, ranks/ (stdout + stderr of EVERY rank:
# torchrun --redirects 3 --tee 3), with PS_VERBOSE=1 — the exchange mode ("one round" / "three forwarding rounds"), the transport that ran
# ("rccl", or "tcp (FALLBACK: ...)" with the reason: a failed ps_comm_selftest is named there), list sizes and the chunk lists of the overlap.
# At the end: OUTDIR/summary.txt — per run ms/step, iterations, transport, overlap, exchange / all-reduce ms per iteration, halo bytes, and
# the speed-up over N = 1.
#
# REHEARSAL on a one-GPU box (no real RCCL with N > 1 there: RCCL refuses duplicate devices):
#   PS_FIRST_RUN_REHEARSAL=1 scripts/first_rccl_run.sh gpurun_out/first_rccl_rehearsal 1 2 4
# runs the same sequence at small sizes with every rank on GPU 0 over the ASYNCHRONOUS stand-in transport (tests/stub_rccl: the RCCL branch of the
# library — comm stream, events, grouped send / receive — on stream-ordered device copies); at most 4 ranks (the box admits 6 GPU processes).
set -u
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/first_rccl}
shift || true
COUNTS=${*:-"1 2 4 8"}
REH=${PS_FIRST_RUN_REHEARSAL:-0}
mkdir -p "$OUT"
export PS_VERBOSE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
if [ "$REH" = "1" ]; then
    make -C tests/stub_rccl -s || { echo "tests/stub_rccl did not build"; exit 2; }
    export PS_RCCL_LIB=$PWD/tests/stub_rccl/libps_stub_rccl.so PS_DIST_OVERLAP=1 PS_FUSED_R=1
    RES_WEAK=64; RES_STRONG=128; STRONG_BLOCK=64; STEPS=2; WARM=1
else
    RES_WEAK=256; RES_STRONG=512; STRONG_BLOCK=512; STEPS=5; WARM=2
fi
PORT=29870
bricks_for() { case $1 in 2) echo 1x1x2;; 4) echo 2x2x1;; 8) echo 2x2x2;; *) echo "";; esac; }

run() {   # run <tag> <N> <bench.py arguments ...>
    local tag=$1 n=$2; shift 2
    local d=$OUT/$tag
    mkdir -p "$d/ranks"
    PORT=$((PORT + 3))
    echo "=== $tag: N=$n bench.py $*" | tee -a "$OUT/summary.txt"
    if [ "$n" = "1" ]; then
        python bench.py --gpus 1 "$@" > "$d/line.json" 2> "$d/stderr.txt"
    else
        python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $PORT \
            --redirects 3 --tee 3 --log-dir "$d/ranks" bench.py --gpus "$n" "$@" > "$d/stdout_all.txt" 2> "$d/stderr.txt"
        grep -h '^{"metric"' "$d/stdout_all.txt" | tail -1 > "$d/line.json"      # (--tee prefixes nothing on the launcher's own stdout copy of rank 0's line)
        [ -s "$d/line.json" ] || grep -h -o '{"metric".*' "$d/stdout_all.txt" | tail -1 > "$d/line.json"
    fi
    echo "    launcher rc=$?; exchange mode as the ranks printed it (PS_VERBOSE): $(cat "$d/stderr.txt" "$d"/ranks/*/*/*/stderr.log 2>/dev/null | grep -h 'exchanges of the solve' | sort | uniq -c | tr '\n' ';')" >> "$OUT/summary.txt"
    python - "$d/line.json" <<'PY' | tee -a "$OUT/summary.txt"
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("    NO LINE (%s): see stderr.txt and ranks/" % e); sys.exit(0)
mg = d.get("multi_gpu") or {}
print("    %.1f ms/step  %d iterations  transport=%s  overlap=%s  exchange %s ms/it  all-reduce %s ms/it  halo max %s B/it  parallelism: %s"
      % (d["ms_per_step"], d["cg_iterations"], d.get("transport"), mg.get("overlap"), mg.get("exchange_ms_per_iter"), mg.get("allreduce_ms_per_iter"),
         (mg.get("halo_bytes_per_iter") or {}).get("max_per_rank"), d["config"].get("parallelism")))
s = d.get("strong_512")
if s:
    m2 = s.get("multi_gpu") or {}
    print("    strong block: %s" % (("%.1f ms/step  %d iterations  %s  %s  exchange %s ms/it  all-reduce %s ms/it" % (s["ms_per_step"], s["cg_iterations"], s.get("decomposition"), s.get("transport"),
          m2.get("exchange_ms_per_iter"), m2.get("allreduce_ms_per_iter"))) if "ms_per_step" in s else s))
PY
}

: > "$OUT/summary.txt"
{ echo "# first multi-GPU run: $(date -u)  rehearsal=$REH  counts: $COUNTS"; rocm-smi --showtopo 2>/dev/null | head -40; } >> "$OUT/summary.txt"
for N in $COUNTS; do
    run default_n$N $N --steps $STEPS --warmup $WARM --res $RES_WEAK --strong-res $STRONG_BLOCK $([ "$REH" = "1" ] && echo --cpu-sample-res 32)
    B=$(bricks_for $N)
    if [ "$N" = "1" ]; then
        run strong_coil_n1 1 --scaling strong --scene coil --res $RES_STRONG --steps $STEPS --warmup $WARM --no-strong-512 --no-cpu-baseline --no-other-preconditioners
    else
        run strong_coil_bricks_n$N $N --scaling strong --scene coil --res $RES_STRONG --bricks $B --steps $STEPS --warmup $WARM --no-strong-512
    fi
    if [ "$N" = "8" ]; then
        run strong_coil_slabs_n8 8 --scaling strong --scene coil --res $RES_STRONG --steps $STEPS --warmup $WARM --no-strong-512
        run strong_spheres256_bricks_n8 8 --scaling strong --scene spheres --res 256 --bricks 2x2x2 --steps $STEPS --warmup $WARM --no-strong-512
        PS_DIST_FORWARD=1 run strong_coil_bricks_n8_forwarding 8 --scaling strong --scene coil --res $RES_STRONG --bricks 2x2x2 --steps $STEPS --warmup $WARM --no-strong-512
    fi
done
python - "$OUT" <<'PY' | tee -a "$OUT/summary.txt"
import glob, json, os, sys
out = sys.argv[1]
rows = {}
for f in sorted(glob.glob(os.path.join(out, "strong_coil*", "line.json")) + glob.glob(os.path.join(out, "default_n*", "line.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    tag = os.path.basename(os.path.dirname(f))
    rows[tag] = d
print("# strong series (the coil scene cut N ways, ms/step; speed-up over N = 1):")
base = rows.get("strong_coil_n1", {}).get("ms_per_step")
for tag, d in rows.items():
    if tag.startswith("strong_coil"):
        print("  %-40s N=%d  %9.1f ms/step  x%.2f  %s" % (tag, d["n_gpus"], d["ms_per_step"], (base / d["ms_per_step"]) if base else float("nan"), d.get("transport")))
print("# weak series (the cavity, one block of res^3 cells per GPU, ms/step; efficiency = t(1) / t(N)):")
b1 = rows.get("default_n1", {}).get("ms_per_step")
for tag, d in rows.items():
    if tag.startswith("default_n"):
        print("  %-40s N=%d  %9.1f ms/step  eff %.2f  %s" % (tag, d["n_gpus"], d["ms_per_step"], (b1 / d["ms_per_step"]) if b1 else float("nan"), d.get("transport")))
PY
echo "done: $OUT/summary.txt"
