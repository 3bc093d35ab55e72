#!/usr/bin/env python3
"""Compiler-flag variants of the lane a-trous translation unit (svgf_atrous_lane.hip) as prebuilt libraries
libsvgf_hip.so.<TAG> for tools/experiments/exp_ab_multi.sh: same sources, same results, only the backend's scheduling /
register-pressure heuristics differ.  Prints VGPRs and scratch of the five default-path kernels for every variant.
    python tools/experiments/build_lane_flag_variants.py        (CPU container; cross-compiles gfx950)
"""
import concurrent.futures, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cuda-path-tracer-denoising_amd"))
import build  # noqa: E402

VARIANTS = {
    "A": [],
    "B": ["-mllvm", "-amdgpu-schedule-metric-bias=0"],
    "C": ["-mllvm", "-amdgpu-use-amdgpu-trackers=1"],
    "D": ["-mllvm", "-enable-post-misched=0"],
    "E": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"],
    "F": ["-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule=1"],
    "G": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
}


def run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return r.returncode, r.stdout


def main():
    cflags = [f for f in build.HIPCC_FLAGS if f != "-shared"]
    hipcc = build.hipcc_path()
    lane = os.path.join(build.CSRC, "svgf_atrous_lane.hip")
    with tempfile.TemporaryDirectory(prefix="svgf_var_") as d:
        others = [s for s in build.HIP_SOURCES if s != "svgf_atrous_lane.hip"]
        objs = {s: os.path.join(d, s + ".o") for s in others}
        with concurrent.futures.ThreadPoolExecutor(max_workers=16) as pool:
            jobs = [pool.submit(run, [hipcc] + cflags + build.HIPCC_FILE_FLAGS.get(s, []) + ["-c", os.path.join(build.CSRC, s), "-o", objs[s]]) for s in others]

            def variant(tag, flags):
                o = os.path.join(d, f"lane_{tag}.o")
                rc, out = run([hipcc] + cflags + flags + ["-c", lane, "-o", o])
                if rc:
                    return tag, None, out[-400:]
                asm = os.path.join(d, f"lane_{tag}.s")
                run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only"] + flags + [lane, "-o", asm])
                txt = open(asm).read() if os.path.exists(asm) else ""
                stats = []
                for m in re.finditer(r"\.name:\s+(\S*k_atrous_lane\S*).*?\.private_segment_fixed_size:\s*(\d+).*?\.vgpr_count:\s*(\d+)", txt, re.S):
                    t = re.search(r"laneILi(\d)ELb(\d)", m.group(1))
                    stats.append(f"S=2^{t.group(1)}{'v' if t.group(2) == '1' else ' '}:{m.group(3)}v/{m.group(2)}s")
                return tag, o, " ".join(stats)
            vj = [pool.submit(variant, t, f) for t, f in VARIANTS.items()]
            for j in jobs:
                rc, out = j.result()
                if rc:
                    sys.exit(out)
            for j in vj:
                tag, o, info = j.result()
                if o is None:
                    print(f"{tag} {' '.join(VARIANTS[tag])}: does not compile: {info}")
                    continue
                lib = build.LIB + "." + tag
                rc, out = run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + list(objs.values()) + [o, "-o", lib])
                print(f"{tag} {' '.join(VARIANTS[tag]) or '(baseline)'}: {info}" + (" LINK FAILED " + out[-200:] if rc else ""))


if __name__ == "__main__":
    main()
