"""Host half of the library under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; GPU sanitizers are not
available on the pool).  `make san` compiles tests/san/host_san.cpp with the SAME host sources the product library is built
from (scene / OBJ / MTL parser — the role of scene.cpp:86-170,202-262,304-358 and material_loader.cpp:153-401 —, the JPEG /
PNG / HDR decoders that stand in for stb_image, the PNG writer, the resampler, the BVH builder and its three host walks).

The harness loads every shipped asset through every entry point, checks that the three walks agree on every scene, then feeds
the loaders damaged copies of every input class: a loader may refuse a file, it may not touch memory it does not own."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    subprocess.check_call(["make", "-s", "san"], cwd=ROOT)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([os.path.join(ROOT, "build", "host_san"), os.path.join(ROOT, "assets"), str(tmp_path), "400", "20261004"],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-4000:] + out.stdout[-1000:]
    assert "no sanitizer finding" in out.stdout
    # the whole assets (6 scenes x 4 flag sets; cornell.scene names an OBJ the reference does not ship either: refused 4 times)
    assert "shipped inputs 28 loaded / 4 refused" in out.stdout
