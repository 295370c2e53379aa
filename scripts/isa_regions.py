#!/usr/bin/env python3
"""Static instruction counts of the restart kernel per source region: compiles a scratch copy of pt_kernels.hip with
assembler-comment markers at the region boundaries and counts VALU / SALU / LDS / scratch / lane-spill instructions between
them in the ISA listing (block placement follows the source closely enough for this to be a useful map; the markers are
scheduling barriers, so the listing differs slightly from the shipped code).   python3 scripts/isa_regions.py [--dump REGION]"""
import os, re, subprocess, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "cuda-pathtracer_amd", "csrc", "pt_kernels.hip")
MARKS = [  # (unique source anchor, marker name) — the marker goes in front of the anchor
    ("    if (!idle) {\n      if (!walking) {\n        best.t = PT_MAX_DIST;", "round_begin"),
    ("      } else if (WIDE) {\n        traverse_round4<STATS, VARIANT == PT_RS_WIDE8", "traverse_begin"),
    ("      if (node == PT_END) {\n        // the iteration's first variate", "traverse_end"),
    ("        if (path_post<STATS>(p, st, r1, n, cnt)) {\n          path_finish_sample(p, st);", "lights_end"),
    ("          path_finish_sample(p, st);\n          idle = true;\n          if (STATS) samples++;", "post_end"),
    ("  if (!p.is_static) {\n    st.acc = found ? inter.diffuse_col : env_lookup(p, st.d);", "resolve_end"),
    ("  const f3 d = st.d;\n  const float cos_theta = dot(inter.normal, d);", "miss_end"),
    ("  const float pmax = __builtin_fmaxf(st.throughput.x, __builtin_fmaxf(st.throughput.y, st.throughput.z));\n  if (r1 > pmax && st.b() > 1u) return true;\n  st.throughput = st.throughput * rcp_hot(pmax);\n  ++st.bk;\n  return", "bsdf_end"),
]
s = open(SRC).read()
for anchor, name in MARKS:
    assert s.count(anchor) == 1, (name, s.count(anchor))
    s = s.replace(anchor, 'asm volatile("; PT_MARK %s" ::: "memory");\n' % name + anchor)
with tempfile.TemporaryDirectory() as td:
    src = os.path.join(td, "pt_kernels.hip"); open(src, "w").write(s)
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                           "-fno-vectorize", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "cuda-pathtracer_amd", "host"),
                           "-I" + os.path.join(ROOT, "cuda-pathtracer_amd", "csrc"), "-x", "hip", "--cuda-device-only", "-S", "-o", out, src],
                          stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_ZN5ptamd21pt_megakernel_restartILb1ELi0EEEvNS_7KParamsE:")][0]
end = [i for i, l in enumerate(lines) if i > start and re.match(r"\.Lfunc_end\d+:", l)][0]
marks = [(i, re.search(r"; PT_MARK (\w+)", lines[i]).group(1)) for i in range(start, end) if "; PT_MARK" in lines[i]]
bounds = [(start, "prologue")] + marks + [(end, "end")]
dump = sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "--dump" else None
for (a, n), (b, _) in zip(bounds, bounds[1:]):
    c = collections.Counter()
    for l in lines[a:b]:
        t = l.strip().split()[0] if l.strip() else ""
        if t.startswith("v_readlane") or t.startswith("v_writelane"): c["lane-spill"] += 1
        elif t.startswith("v_mov"): c["v_mov"] += 1; c["VALU"] += 1
        elif t.startswith("v_"): c["VALU"] += 1
        elif t.startswith("s_") and not t.startswith(("s_waitcnt", "s_nop")): c["SALU"] += 1
        elif t.startswith("ds_"): c["LDS"] += 1
        elif t.startswith("scratch_"): c["scratch"] += 1
        elif t.startswith(("global_", "flat_", "buffer_")): c["VMEM"] += 1
    print("%-16s %s" % (n, dict(c)))
    if dump == n:
        print("\n".join(l for l in lines[a:b] if l.strip() and not l.strip().startswith((";", "."))))
