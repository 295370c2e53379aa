"""The C-ABI library loads on a CPU-only host and exports every symbol include/ptamd.h declares;
compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ptamd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptamd_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(P):
    lib = P.native.load()
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ptamd.h but not exported by libptamd.so"
    assert set(names) == set(P.native.SIGNATURES), "python binding out of sync with the header"


def test_struct_layouts_match_reference(P):
    N = P.native
    assert C.sizeof(N.Face) == 112 and N.Face.normals.offset == 36 and N.Face.texcoords.offset == 72
    assert N.Face.tangent.offset == 96 and N.Face.material_id.offset == 108
    assert C.sizeof(N.Material) == 16 and C.sizeof(N.Light) == 32 and N.Light.vec.offset == 12
    assert N.Light.emission.offset == 24 and N.Light.radius.offset == 28
    assert C.sizeof(N.Camera) == 64 and N.Camera.fov_x.offset == 48 and N.Camera.focus_dist.offset == 60


def test_argument_errors_do_not_crash(P):
    lib = P.native.load()
    assert lib.ptamd_host_scene_load(None, 0, None) == P.native.PTAMD_ERR_ARG
    assert b"null" in lib.ptamd_get_last_error()
    assert lib.ptamd_upload_scene(None, None, None) == P.native.PTAMD_ERR_ARG
    assert lib.ptamd_raytrace_ex(None, None) == P.native.PTAMD_ERR_ARG
    assert lib.ptamd_raytrace(None, None, 0, 0, None, 1, 1, None, None, 0, 0) == P.native.PTAMD_ERR_ARG
    assert lib.ptamd_version().startswith(b"ptamd")


def test_create_fails_loudly_without_gpu(P):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(P.PtamdError) as e:
        P.Context(0)
    assert e.value.status == P.native.PTAMD_ERR_HIP
    assert "no CPU fallback" in str(e.value)


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under the package or include/ may reference it."""
    bad = []
    for base in ("cuda-pathtracer_amd", "include"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"pt_oracle|libpt_oracle|oracle/|import\s+oracle|or_render", txt):
                        # comments that merely point the reader at the oracle's documentation are fine
                        hits = [l for l in txt.splitlines() if re.search(r"pt_oracle|libpt_oracle|or_render", l)
                                and not l.strip().startswith(("//", "*", "#", "/*"))]
                        if hits:
                            bad.append((f, hits[:2]))
    assert not bad, bad


def test_cpp_host_example_compiles_and_fails_loudly_without_gpu(P, tmp_path):
    """examples/headless_render.cpp (C++ host over raytrace.hpp + interop.hpp, the mirrors of raytrace.h and
    driver/interop.h) builds against the C-ABI with a plain g++; run here, without a GPU, it must stop at ptamd_create
    with an error message — never render through some other path."""
    import subprocess
    exe = str(tmp_path / "headless_render")
    lib_dir = os.path.join(ROOT, "cuda-pathtracer_amd")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(lib_dir, "host"), os.path.join(ROOT, "examples", "headless_render.cpp"),
                           "-L" + lib_dir, "-lptamd", "-Wl,-rpath," + lib_dir, "-o", exe])
    import torch
    if torch.cuda.is_available():
        return
    r = subprocess.run([exe, os.path.join(ROOT, "assets", "indoor.scene"), "32", "32", "1", str(tmp_path / "o.png")],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "ptamd_create" in r.stderr and not os.path.exists(str(tmp_path / "o.png"))


def _build_multigpu_host(tmp_path):
    import subprocess
    exe = str(tmp_path / "multigpu_render")
    lib_dir = os.path.join(ROOT, "cuda-pathtracer_amd")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-Wall", "-Wextra", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "multigpu_render.cpp"), "-L" + lib_dir, "-lptamd", "-lrccl",
                           "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_cpp_multigpu_host_compiles_and_fails_loudly_without_gpu(P, tmp_path):
    """examples/multigpu_render.cpp — the C++ N-GPU host (row bands + ncclAllGather straight from RCCL, no Python) —
    builds against the C-ABI, HIP and RCCL; without a GPU it stops with an error and writes nothing."""
    import subprocess
    import torch
    exe = _build_multigpu_host(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    if torch.cuda.is_available():
        return
    out = str(tmp_path / "o.png")
    r = subprocess.run([exe, os.path.join(ROOT, "assets", "indoor.scene"), "64", "64", "2", "3", out], capture_output=True, text=True)
    assert r.returncode == 1 and "no HIP device" in r.stderr and not os.path.exists(out)


@pytest.mark.gpu
def test_cpp_multigpu_host_renders_and_gathers(P, tmp_path):
    """On the GPU box (one GPU): the C++ host with --ranks 1 --check renders its band, runs the RCCL all-gather and must
    reproduce the frame a plain full-frame launch gives; the PNG it writes equals the Python host's surface."""
    import json
    import subprocess
    import numpy as np
    import torch
    if os.environ.get("PTAMD_DEFAULT_KERNEL", "6") not in ("3", "5", "6"):
        pytest.skip("PTAMD_DEFAULT_KERNEL selects a kernel that cannot batch frames")
    exe = _build_multigpu_host(tmp_path)
    out = str(tmp_path / "o.png")
    r = subprocess.run([exe, os.path.join(ROOT, "assets", "indoor.scene"), "320", "180", "4", "4", out, "--ranks", "1",
                        "--frames", "3", "--check"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and "equals" in line["check"]
    hs = P.HostScene.load(os.path.join(ROOT, "assets", "indoor.scene"))
    with P.Context(0) as ctx:
        sid, cid = ctx.upload_scene(hs), ctx.upload_cubemap(P.cubemap_for_scene(hs))
        fr = P.FrameRenderer(ctx, sid, cid, hs.camera_struct(), 320, 180)
        fr.render(spp=4, bounces=4)
        torch.cuda.synchronize()
        want = fr.surface.cpu().numpy()[:, :, :3]
    np.testing.assert_array_equal(P.load_image8(out), want)


def test_optional_gl_presenter_builds_and_refuses_to_run_without_a_context(P, tmp_path):
    """include/ptamd_gl.h / libptamd_gl.so — the GL half of driver::Interop (interop.cpp:14-20,36-72,104-116) for hosts
    that own a window: HIP-GL interop through a registered pixel buffer + the reference's flipped blit.  No display
    exists on the build box or on an MI355X node, so: it builds and links against libGL and HIP, exports what its header
    declares, compiles into the C++ Interop mirror, and without a current GL context create() fails with a message."""
    import subprocess
    if not os.path.exists("/usr/include/GL/glext.h"):
        pytest.skip("no GL headers on this host")
    subprocess.check_call(["make", "-s", "gl"], cwd=ROOT)
    lib = C.CDLL(os.path.join(ROOT, "cuda-pathtracer_amd", "libptamd_gl.so"))
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "ptamd_gl.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(ptamd_gl_[a-z0-9_]+)\s*\(", text)))
    assert len(names) == 5
    for n in names:
        assert hasattr(lib, n), n
    lib.ptamd_gl_get_last_error.restype = C.c_char_p
    h = C.c_void_p()
    assert lib.ptamd_gl_presenter_create(64, 64, C.byref(h)) == P.native.PTAMD_ERR_ARG and not h.value
    assert b"no current OpenGL context" in lib.ptamd_gl_get_last_error()
    assert lib.ptamd_gl_presenter_present(None, None, None) == P.native.PTAMD_ERR_ARG
    # the C++ mirror of driver::Interop compiles with the GL overload of blit()
    src = tmp_path / "with_gl.cpp"
    src.write_text('#define PTAMD_WITH_GL 1\n#include "interop.hpp"\n'
                   'int f(ptamd_host::Interop& i, ptamd_gl_presenter* g) { return i.blit(g, nullptr); }\n')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-c", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "cuda-pathtracer_amd", "host"), str(src), "-o", str(tmp_path / "with_gl.o")])


def test_native_load_brings_torch_in_first():
    """One HIP runtime per process: libptamd.so must bind to the libamdhip64 / libhsa-runtime64 PyTorch bundles, so native.load()
    imports torch (when installed) before it dlopens the library — build() followed by smoke() in one interpreter found
    "no ROCm-capable device" on the GPU box when the order was the other way round."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import cuda_pathtracer_amd as P; assert 'torch' not in sys.modules; "
            "P.native.load(); assert 'torch' in sys.modules; print('ok')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_bench_refuses_counters_of_another_build(tmp_path):
    """bench.py prices its roofline with PMC counters only when they were collected on THIS build of the device code
    (ptamd_build_id, stamped into profiles/pmc_latest.json by scripts/summarize_pmc.py)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = {"build_id": "0123456789abcdef", "kernel": "restart", "scene": "indoor.scene", "workload": "1920x1080", "spp": 4, "bounces": 4,
           "frames_per_launch": 4, "valu_insts_per_sample": 85.0}
    path = tmp_path / "pmc.json"
    path.write_text(json.dumps(rec))
    same = bench.load_pmc(str(path), "restart", 1920, 1080, 4, 4, 4, 1, "indoor.scene", "0123456789abcdef")
    other = bench.load_pmc(str(path), "restart", 1920, 1080, 4, 4, 4, 1, "indoor.scene", "fedcba9876543210")
    assert same is not None and same["stale"] is False
    assert other is not None and other["stale"] is True
    assert bench.load_pmc(str(path), "restart", 1280, 720, 4, 4, 4, 1, "indoor.scene", "0123456789abcdef") is None   # another workload
    import cuda_pathtracer_amd as P
    bid = P.native.load().ptamd_build_id().decode()
    assert len(bid) == 16 and all(c in "0123456789abcdef" for c in bid) and bid in P.native.load().ptamd_version().decode()
    committed = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
    assert "build_id" in committed
