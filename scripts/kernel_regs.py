#!/usr/bin/env python3
"""Register budget of every kernel of pt_kernels.hip as the Makefile builds it: VGPRs, SGPRs, spills, scratch, LDS, occupancy
(from the compiler's own summary comments in the ISA listing).   python3 scripts/kernel_regs.py [name-filter] [-Dmacro ...]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flt = [a for a in sys.argv[1:] if not a.startswith("-")]
extra = [a for a in sys.argv[1:] if a.startswith("-") and a != "--keep"]
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fno-vectorize",
                           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "cuda-pathtracer_amd", "host"), "-I" + os.environ.get("PT_SRC_DIR", os.path.join(ROOT, "cuda-pathtracer_amd", "csrc")),
                           "-x", "hip", "--cuda-device-only", "-S", "-o", out] + extra + [os.path.join(os.environ.get("PT_SRC_DIR", os.path.join(ROOT, "cuda-pathtracer_amd", "csrc")), "pt_kernels.hip")], stderr=subprocess.DEVNULL)
    txt = open(out).read()
if "--keep" in sys.argv: open("/tmp/pt_kernels.s", "w").write(txt)
for m in re.finditer(r"; Kernel info:.*?\n(.*?); COMPUTE_PGM_RSRC2", txt, re.S):
    pass
blocks = re.split(r"\n\s*\.section\s+\.AMDGPU\.csdata", txt)
names = re.findall(r"^\s*\.amdhsa_kernel (\S+)", txt, re.M)
infos = re.findall(r"; Function info:.*?$|; Kernel info:\n(.*?)(?=\n[^;])", txt, re.S | re.M)
# simpler: walk line by line
cur = None; rows = {}
for line in txt.split("\n"):
    m = re.match(r"^(\S+):\s*; @(\S+)", line)
    if m: cur = m.group(2)
    for key in ("NumVgprs", "NumAgprs", "NumSgprs", "ScratchSize", "Occupancy", "LDSByteSize", "VGPRBlocks"):
        m = re.match(r"^; %s: (\d+)" % key, line)
        if m and cur: rows.setdefault(cur, {})[key] = int(m.group(1))
    m = re.match(r"^\s*\.vgpr_spill_count:\s*(\d+)", line)
import shutil
filt = shutil.which("c++filt") or "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"
for k, v in rows.items():
    if "Occupancy" not in v: continue
    name = subprocess.run([filt, k], capture_output=True, text=True).stdout.strip() if filt else k
    name = re.sub(r"\(ptamd::KParams.*", "", name).replace("void ptamd::", "")
    if flt and not any(f in name for f in flt): continue
    print("%-60s vgpr %3d agpr %3d sgpr %3d scratch %4d occupancy %d" % (name[:60], v.get("NumVgprs", -1), v.get("NumAgprs", 0), v.get("NumSgprs", -1), v.get("ScratchSize", 0), v["Occupancy"]))
