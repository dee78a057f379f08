#!/usr/bin/env python3
"""Instruction histogram of one kernel of a hipcc -S listing, whole body and hottest loop.
usage: isa_hist.py <file.s> <substring of the mangled kernel name> [--dump]
The loop is taken as the span between the last backward branch target and that branch (the persistent chunk loop)."""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
dump = "--dump" in sys.argv
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if l.endswith(":") and key in l and not l.startswith("\t") and not l.startswith("."):
        start = i
        break
    m = re.match(r"^(_Z\S+):", l)
    if m and key in m.group(1):
        start = i
        break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
ins = []
for l in body:
    s = l.strip()
    m = re.match(r"^(\.LBB\S+):", s)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if not s or s.startswith(";") or s.startswith("."):
        continue
    ins.append(s.split(";")[0].strip())
# backward branches
loops = []
for i, s in enumerate(ins):
    m = re.match(r"^s_c?branch\S*\s+(\.LBB\S+)", s)
    if m and m.group(1) in labels and labels[m.group(1)] <= i:
        loops.append((labels[m.group(1)], i))


def cls(op):
    if op.startswith("buffer_load") or op.startswith("global_load") or op.startswith("flat_load"):
        return "VMEM_RD"
    if op.startswith("buffer_store") or op.startswith("global_store"):
        return "VMEM_WR"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("s_waitcnt"):
        return "WAIT"
    if op.startswith("s_barrier"):
        return "BARRIER"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("v_"):
        return "VALU"
    return "OTHER"


def hist(seq, title):
    c = collections.Counter(cls(s.split()[0]) for s in seq)
    ops = collections.Counter(s.split()[0] for s in seq)
    print(title, "instructions:", len(seq), dict(c))
    print("   top ops:", ", ".join(f"{k}:{v}" for k, v in ops.most_common(28)))


hist(ins, "whole kernel")
if loops:
    a, b = max(loops, key=lambda ab: ab[1] - ab[0])
    hist(ins[a:b + 1], "largest loop")
    if dump:
        print("\n".join(ins[a:b + 1]))
for l in body:
    if "vgpr_count" in l or "sgpr_count" in l or "Occupancy" in l or "lds_size" in l.lower() or "ScratchSize" in l:
        print(l.strip())
