#!/usr/bin/env python3
"""Instruction census of a kernel in hipcc's -save-temps assembly: per basic block, counts by issue class.

usage: isa_blocks.py file.s <substring of the kernel symbol> [min_instructions]
Classes: pk (v_pk_*), trans (v_exp/v_sqrt/v_rcp/v_rsq/v_log), dpp (VALU with a DPP modifier), valu (other v_*),
ds_r / ds_w, vmem (global_/buffer_/flat_), salu (s_* except waitcnt/barrier/nop), wait (s_waitcnt), other.
"""
import re
import sys
from collections import Counter, OrderedDict


def klass(op, rest):
    if op.startswith("v_"):
        if "dpp" in rest or op.endswith("_dpp"):
            return "dpp"
        if op.startswith("v_pk_"):
            return "pk"
        if re.match(r"v_(exp|sqrt|rcp|rsq|log|sin|cos)_", op):
            return "trans"
        if op.endswith("_f64") or "_f64_" in op:
            return "f64"
        return "valu"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "ds_r"
    if op.startswith("ds_"):
        return "ds_w"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_waitcnt":
        return "wait"
    if op in ("s_barrier", "s_nop", "s_setprio", "s_sleep"):
        return op
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, sym = sys.argv[1], sys.argv[2]
    min_n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path).read().splitlines()
    start = None
    for i, ln in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", ln) and sym in ln.split(":")[0]:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    blocks = OrderedDict()
    cur = "entry"
    blocks[cur] = Counter()
    for ln in lines[start + 1:]:
        s = ln.strip()
        if s.startswith(".Lfunc_end"):
            break
        if s.startswith(".LBB") and s.rstrip().endswith(":") or re.match(r"^\.LBB\d+_\d+:", s):
            cur = s.split(":")[0]
            blocks[cur] = Counter()
            continue
        if not s or s.startswith((";", ".", "//")):
            continue
        m = re.match(r"^([a-z_0-9]+)\s*(.*)$", s)
        if not m:
            continue
        op, rest = m.group(1), m.group(2)
        blocks[cur][klass(op, rest)] += 1
        if op in ("s_cbranch_scc0", "s_cbranch_scc1", "s_cbranch_vccz", "s_cbranch_vccnz", "s_cbranch_execz", "s_cbranch_execnz", "s_branch"):
            blocks[cur]["->" + rest.split()[0]] += 0
    keys = ["valu", "pk", "trans", "dpp", "f64", "ds_r", "ds_w", "vmem", "salu", "wait", "s_barrier", "s_setprio", "s_nop"]
    print(f"{'block':<12}" + "".join(f"{k:>9}" for k in keys) + "   branches")
    tot = Counter()
    for b, c in blocks.items():
        n = sum(v for k, v in c.items() if not k.startswith("->"))
        tot.update({k: v for k, v in c.items() if not k.startswith("->")})
        if n < min_n:
            continue
        br = " ".join(k for k in c if k.startswith("->"))
        print(f"{b:<12}" + "".join(f"{c.get(k, 0):>9}" for k in keys) + "   " + br)
    print(f"{'TOTAL':<12}" + "".join(f"{tot.get(k, 0):>9}" for k in keys))


if __name__ == "__main__":
    main()
