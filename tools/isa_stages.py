#!/usr/bin/env python3
"""Instruction census of the fused kernel's loader loop, stage by stage.

Compiles svgf_atrous_fused.hip for gfx950 with the timeline stamps on (-DSVGF_LANE_TIMELINE: one s_memtime per stage
boundary), finds the loop body of the AoS kernel (FUSED = 1) — the last basic-block run that holds all stamp kinds — and
counts the instructions between consecutive s_memtime marks by class.  Offline (no GPU): `python tools/isa_stages.py`.
  --planar    the FUSED = 2 instantiation
  --keep DIR  leave the .s there
"""
import argparse, os, re, subprocess, sys, tempfile, collections

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "cuda-path-tracer-denoising_amd", "csrc")


def classify(op):
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_div_", "v_exp", "v_log")): return "div"
    if op.startswith("v_cmp"): return "cmp"
    if op.startswith("v_cndmask"): return "cndmask"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    if "f64" in op: return "f64"
    if op.startswith("v_"): return "valu"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--planar", action="store_true")
    ap.add_argument("--keep")
    ap.add_argument("--no-timeline", action="store_true", help="whole-loop census of the production build (no stamps)")
    a = ap.parse_args()
    d = a.keep or tempfile.mkdtemp(prefix="isa_")
    os.makedirs(d, exist_ok=True)
    s = os.path.join(d, "fused.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
           "-mllvm", "-simplifycfg-sink-common=false", os.path.join(CSRC, "svgf_atrous_fused.hip"), "-o", s]
    if not a.no_timeline: cmd.insert(1, "-DSVGF_LANE_TIMELINE")
    if not (a.keep and os.environ.get("ISA_REUSE") and os.path.exists(s)): subprocess.check_call(cmd)
    txt = open(s).read()
    # kernels: split at the .globl / function label
    want = "Li1ELi2ELi0EEE" if a.planar else "Li1ELi1ELi0EEE"
    kern = None
    for m in re.finditer(r"^(_Z\w*k_atrous_lane\w*):[^\n]*\n(.*?)\n\.Lfunc_end", txt, re.S | re.M):
        if want in m.group(1): kern = m
    if kern is None:
        sys.exit("kernel not found")
    name, body = kern.group(1), kern.group(2)
    meta = re.search(re.escape(name) + r".*?\.vgpr_count:\s*(\d+).*?", txt[kern.end():], re.S)
    lines = [l.strip() for l in body.splitlines()]
    ins = [(i, l.split()[0]) for i, l in enumerate(lines) if l and not l.startswith((";", ".", "//")) and not l.split(";")[0].strip().endswith(":")]
    print(name[:90], "instructions", len(ins))
    for key in ("vgpr_count", "sgpr_count", "scratch", "vgpr_spill", "sgpr_spill"):
        m = re.search(r"\." + key + r"\w*:\s*(\d+)", txt[kern.end():kern.end() + 200000])
    for m in re.finditer(r"; (NumVgprs|NumSgprs|ScratchSize|Occupancy|NumAgprs): (\d+)", txt[kern.end():kern.end() + 4000]):
        print("  ", m.group(1), m.group(2))
    # the loop: labels and backward branches
    label_at = {l.split(":")[0]: i for i, l in enumerate(lines) if l.startswith(".LBB") and ":" in l}
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r"s_cbranch_\w+\s+(\.LBB\w+)|s_branch\s+(\.LBB\w+)", l)
        if m:
            t = m.group(1) or m.group(2)
            if t in label_at and label_at[t] < i: loops.append((label_at[t], i))
    if a.no_timeline:
        # largest loop with a barrier in it = the loader loop (the compute loop has no global loads of 12-byte pieces)
        cand = [(e - b, b, e) for b, e in loops if any("s_barrier" in lines[k] for k in range(b, e)) and any("global_load_dwordx3" in lines[k] for k in range(b, e))]
        _, b, e = max(cand)
        c = collections.Counter(classify(op) for i, op in ins if b <= i <= e)
        print(f"loader loop: lines {b}..{e}: {sum(c.values())} instructions", dict(c))
        return
    cand = [(e - b, b, e) for b, e in loops if sum("s_memtime" in lines[k] for k in range(b, e)) >= 6 and any("global_load_dwordx3" in lines[k] for k in range(b, e))]
    if not cand: sys.exit("loop with stamps not found")
    _, b, e = max(cand)
    marks = [i for i in range(b, e) if "s_memtime" in lines[i]]
    print(f"loader loop: lines {b}..{e}, {len(marks)} stamps")
    names = ["C2 (blend, commit)", "C1 (consistency, request)", "B (reprojection)", "A (primary loads)", "second pixel (all stages)", "barrier", "loop tail"]
    edges = marks + [e]
    tot = collections.Counter()
    for k in range(len(marks)):
        c = collections.Counter(classify(op) for i, op in ins if edges[k] < i < edges[k + 1])
        tot.update(c)
        print(f"  {names[k] if k < len(names) else 'rest':28s} {sum(c.values()):5d}  " + " ".join(f"{t}={n}" for t, n in sorted(c.items())))
    print(f"  {'loop body':28s} {sum(tot.values()):5d}  " + " ".join(f"{t}={n}" for t, n in sorted(tot.items())))


if __name__ == "__main__":
    main()
