"""The build refuses a prefill64_kernel that spills registers: the kernel counts its own vmcnt around LDS-DMA issued by inline asm, and a
scratch access (with the waits the compiler adds for it) silently breaks that count (vattention_amd/build.py, NO_SPILL_KERNELS)."""
import os
import shutil

import pytest

from vattention_amd import build

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")

SPILLS = """
#include <hip/hip_runtime.h>
namespace vattn_k {
__global__ void prefill64_kernel(float* p, int n) {
    float a[256];                                   // indexed dynamically: lives in scratch
    for (int i = 0; i < 256; i++) a[i] = p[i] * (float)threadIdx.x;
    float s = 0.f;
    for (int i = 0; i < n; i++) s += a[(i * 7 + threadIdx.x) & 255];
    p[threadIdx.x] = s;
}
}
"""
CLEAN = """
#include <hip/hip_runtime.h>
namespace vattn_k {
__global__ void prefill64_kernel(float* p, int n) { p[threadIdx.x] = p[threadIdx.x] * (float)n; }
}
"""


def _compile(tmp_path, text):
    src = tmp_path / "prefill64_kernels.hip"          # the guard goes by the translation unit's name
    src.write_text(text)
    build._compile(HIPCC, ["--offload-arch=" + build.ARCH, "-O3", "-std=c++17"], str(src), str(tmp_path / "o.o"))


def test_a_spilling_guarded_kernel_fails_the_build(tmp_path):
    with pytest.raises(RuntimeError, match="register spills"):
        _compile(tmp_path, SPILLS)


def test_a_clean_guarded_kernel_builds(tmp_path):
    _compile(tmp_path, CLEAN)
    assert os.path.getsize(tmp_path / "o.o") > 0


def test_an_unguarded_translation_unit_is_compiled_plainly(tmp_path):
    src = tmp_path / "other.hip"
    src.write_text(SPILLS)
    build._compile(HIPCC, ["--offload-arch=" + build.ARCH, "-O3", "-std=c++17"], str(src), str(tmp_path / "o2.o"))
    assert os.path.getsize(tmp_path / "o2.o") > 0
    shutil.rmtree(tmp_path, ignore_errors=True)
