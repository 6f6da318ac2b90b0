"""The library's HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY section 5 "Race detection / sanitizers", 8b
"Threading"): the host-pointer staging layer (device arena, two pinned buffers, per-device state, the mutex that serialises callers), the
plan builders' workspace carving (plan_quad, plan_dquad, plan_dmesh, plan_latlon, build_pass_plan_any) and the plan handle's life cycle,
compiled for the host only (hipcc --cuda-host-only) against a fake HIP runtime whose "device" memory is host memory and whose asynchronous
copies are deferred until the API orders them (tests/sanitize/).  GPU AddressSanitizer is not available on the pool; kernels are covered by
the every-element comparisons of the GPU tests.  Test infrastructure only: nothing under tests/sanitize is part of the product."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def built(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("ogg_sanitize"))
    p = subprocess.run([os.path.join(HERE, "sanitize", "build_and_run.sh"), out], capture_output=True, text=True, timeout=900)
    return out, p


def test_host_code_is_clean_under_asan_and_ubsan(built):
    out, p = built
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "sanitize driver ok" in p.stdout
    for word in ("AddressSanitizer", "runtime error", "LeakSanitizer", "FAIL "):
        assert word not in p.stderr and word not in p.stdout, p.stderr[-4000:]


def test_the_sanitizer_is_live(built):
    """The same binary with a deliberate one-byte overrun in front of the tests: it must die with an AddressSanitizer report (a harness that
    cannot fail proves nothing)."""
    out, _ = built
    p = subprocess.run([os.path.join(out, "driver_asan")], env=dict(os.environ, OGG_SANITIZE_SELFTEST="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode != 0 and "AddressSanitizer" in p.stderr and "heap-buffer-overflow" in p.stderr


def test_thread_sanitizer_variant():
    """Two threads in the mutex-guarded host-pointer layer under ThreadSanitizer: slow (minutes), so only on request."""
    if not os.environ.get("OGG_SANITIZE_TSAN"):
        pytest.skip("set OGG_SANITIZE_TSAN=1 (takes ~3 minutes)")
    out = "/tmp/ogg_sanitize_tsan"
    shutil.rmtree(out, ignore_errors=True)
    p = subprocess.run([os.path.join(HERE, "sanitize", "build_and_run.sh"), out, "tsan"], capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0 and "sanitize driver ok" in p.stdout and "ThreadSanitizer" not in p.stderr, p.stderr[-4000:]


def test_the_product_does_not_reach_into_the_sanitizer_harness():
    pkg = os.path.join(ROOT, "ocean_model_grid_generator_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(d, f), errors="replace").read()
                assert "tests/sanitize" not in text and "fake_hip_runtime" not in text, os.path.join(d, f)
