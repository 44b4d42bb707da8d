"""In-tree build of every native artefact (explicit hipcc / g++ commands, no JIT cache).

  vattention_amd/libvattn_amd.so        page manager + HIP VMM backend + gfx950 kernels  (hipcc)
  vattention_amd/_vtensor*.so           torch binding: tensor over a raw VA             (g++)
  tests/native/libvattn_fake_backend.so host-only physical backend for CPU tests        (g++)
  tools/lab/libvattn_lab.so             the kernels + measurement scaffolding (-DVATTN_LAB), for kbench and the variant tests
hipcc cross-compiles gfx950 without a GPU.  Artefacts are git-ignored but travel with gpurun.
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vattention_amd")
CSRC = os.path.join(PKG, "csrc")
ARCH = "gfx950"
LIB_SOURCES = ("page_manager.cpp", "hip_backend.cpp", "capi.cpp", "vmm_selfcheck.hip", "attn_api.hip", "prefill_kernels.hip", "prefill64_kernels.hip", "prefill64p_kernels.hip", "decode_kernels.hip",
               "cache_kernels.hip", "hybrid_kernels.hip")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("[build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


# The tile steps of the prefill kernels are 64 hand-placed groups written as `#pragma unroll` loops over compile-time group numbers.  LLVM stops
# honouring the pragma once the unrolled body passes -pragma-unroll-threshold (16 K cost units): the loop then stays a loop, the register arrays
# are indexed at run time (s_set_gpr_idx) and half the kernel goes to scratch (found in round 6 when 16 MFMAs were added to the step).
UNROLL_FLAGS = ["-mllvm", "-pragma-unroll-threshold=1000000"]


# prefill64_kernel counts its own `vmcnt` around LDS-DMA issued by inline asm: a register spill (scratch access, compiler-inserted
# waits the hand-placed ones do not know about) silently breaks it, and the kernel sits at the SGPR / VGPR limits.  Its translation unit
# is therefore compiled with the resource-usage remarks on, and the build FAILS when any instantiation of a guarded kernel spills.
NO_SPILL_KERNELS = {"prefill64_kernels.hip": "prefill64_kernel", "prefill64p_kernels.hip": "prefill64p_kernel", "decode_kernels.hip": "decode_"}
# ... except: the bf16 build of decode_stream_kernel (d = 128, one head block) that takes the fused-RoPE path at run time — bf16 rotates through fp32,
# 12 registers more than three workgroups per CU leave; calls without rotation get the build without that path (decode_kernels.hip,
# launch_decode_stream).  A scratch segment costs ~9 us per launch (profiles/r06_decode_bf16_scratch.txt): no other kernel may grow one unnoticed.
SPILL_ALLOWED = ("decode_stream_kernelIDF16bLi128ELb1ELi1ELin1E",)


# prefill64p_kernel receives its queue tickets in v255 asynchronously (an atomic issued by inline asm, read a tile step later): no other
# instruction of the kernel may touch that register.  Checked on the generated device assembly.
RESERVED_VGPR = {"prefill64p_kernels.hip": (255, (r"^global_atomic_add v255, ", r"^v_readfirstlane_b32 s\d+, v255$"))}


def _check_reserved_vgpr(hipcc, flags, src, index, allowed):
    import re
    r = subprocess.run([hipcc, *flags, "-S", "--cuda-device-only", src, "-o", "-"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    if r.returncode != 0:
        raise RuntimeError("could not produce the device assembly of %s for the reserved-register check" % src)
    single, tuples = re.compile(r"\bv%d\b" % index), re.compile(r"\bv\[(\d+):(\d+)\]")
    bad, hits = [], 0
    for line in r.stdout.splitlines():
        t = line.split(";")[0].strip()
        if not t or t.startswith("."):
            continue
        if not (single.search(t) or any(int(a) <= index <= int(b) for a, b in tuples.findall(t))):
            continue
        if any(re.match(a, t) for a in allowed):
            hits += 1
        else:
            bad.append(t)
    if bad or not hits:
        raise RuntimeError("%s: v%d is reserved for the asynchronous queue ticket, but it %s" % (
            os.path.basename(src), index, ("is also used by: " + "; ".join(bad[:4])) if bad else "is never used"))


def _compile(hipcc, flags, src, obj):
    guard = NO_SPILL_KERNELS.get(os.path.basename(src))
    if not guard:
        return _run([hipcc, *flags, "-c", src, "-o", obj])
    cmd = [hipcc, *flags, "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", obj]
    print("[build]", " ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr)
        raise subprocess.CalledProcessError(r.returncode, cmd)
    name, bad, seen = None, [], 0
    for line in r.stderr.splitlines():
        if "remark: Function Name:" in line:
            name = line.split("Function Name:")[1].split()[0]
            seen += guard in name
        elif name and guard in name and ("ScratchSize" in line or "VGPRs Spill" in line):      # (SGPRs spilled to vector LANES touch no memory)
            value = int(line.split("]:")[-1].split("[")[0].strip() if "ScratchSize" in line else line.split("Spill:")[1].split()[0])
            if value and not any(a in name for a in SPILL_ALLOWED):
                bad.append("%s: %s" % (name, line.split("remark:")[1].split("[-R")[0].strip()))
    if not seen:
        raise RuntimeError("no resource remarks for %s in %s: the spill guard saw nothing" % (guard, src))
    if os.path.basename(src) in RESERVED_VGPR:
        _check_reserved_vgpr(hipcc, flags, src, *RESERVED_VGPR[os.path.basename(src)])
    if bad:
        raise RuntimeError("register spills / a scratch segment in a kernel that must have none (hand-counted vmcnt; + 9 us per launch):\n  " + "\n  ".join(bad))


def build_lib(force=False):
    out = os.path.join(PKG, "libvattn_amd.so")
    srcs = [os.path.join(CSRC, f) for f in LIB_SOURCES]
    deps = srcs + [os.path.join(CSRC, "page_manager.h"), os.path.join(CSRC, "attn_common.h"), os.path.join(CSRC, "prefill64_common.h"), os.path.join(CSRC, "prefill_body.h"), os.path.join(CSRC, "decode_body.h"),
                   os.path.join(ROOT, "include", "vattn.h"),
                   os.path.join(ROOT, "include", "vattn_kernels.h")]
    if force or _newer(out, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wno-inline-asm", *UNROLL_FLAGS] + os.environ.get("VATTN_CXXFLAGS", "").split()
        objdir = os.path.join(ROOT, "build", "obj")
        os.makedirs(objdir, exist_ok=True)
        objs = [os.path.join(objdir, os.path.basename(f) + ".o") for f in srcs]
        # one hipcc per translation unit, in parallel (the prefill kernels dominate: ~40 s)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
            list(ex.map(lambda so: _compile(hipcc, flags, so[0], so[1]), zip(srcs, objs)))
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-pthread", "-Wl,-Bsymbolic", *objs, "-o", out])
    return out


# The lab library (round 6): the kernels' LAB COPIES under tools/lab/csrc/ — every alternative schedule, operand path, in-launch merge
# protocol, timing ablation, clock stamp and the fused hybrid launch, behind `variant` bits — plus the product sources that have no lab
# side (the C entry points, the persistent prefill kernel, cache_flat).  The product sources carry none of that code any more.
LAB_DIR_SRC = os.path.join(ROOT, "tools", "lab", "csrc")
LAB_SOURCES = ("attn_api.hip", os.path.join(LAB_DIR_SRC, "prefill_kernels_lab.hip"), os.path.join(LAB_DIR_SRC, "prefill64_lab.hip"), os.path.join(LAB_DIR_SRC, "prefill32_lab.hip"), "prefill64p_kernels.hip",
               os.path.join(LAB_DIR_SRC, "decode_kernels_lab.hip"), "cache_kernels.hip", os.path.join(LAB_DIR_SRC, "hybrid_lab.hip"))
LAB_HEADERS = ("decode_body_lab.h", "prefill_body_lab.h")


def build_lab(force=False):
    """tools/lab/libvattn_lab.so: the KERNEL sources compiled with -DVATTN_LAB — the product kernels plus the measurement scaffolding
    (alternative operand paths and schedules, in-launch merge protocols, timing ablations with wrong results) that tools/kbench.py
    and the variant tests select through `variant` bits the product library rejects.  Test / measurement infrastructure: nothing in
    vattention_amd/ loads it unless a caller passes such a variant (kernels.klib_lab)."""
    outdir = os.path.join(ROOT, "tools", "lab")
    os.makedirs(outdir, exist_ok=True)
    out = os.path.join(outdir, "libvattn_lab.so")
    srcs = [f if os.path.isabs(f) else os.path.join(CSRC, f) for f in LAB_SOURCES]
    deps = srcs + [os.path.join(CSRC, "attn_common.h"), os.path.join(CSRC, "prefill64_common.h")] + [os.path.join(LAB_DIR_SRC, h) for h in LAB_HEADERS] + [
        os.path.join(ROOT, "include", "vattn_kernels.h")]
    if force or _newer(out, deps):
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-DVATTN_LAB", "-Wno-inline-asm", "-I" + CSRC, *UNROLL_FLAGS]
        objdir = os.path.join(ROOT, "build", "obj_lab")
        os.makedirs(objdir, exist_ok=True)
        objs = [os.path.join(objdir, os.path.basename(f) + ".o") for f in srcs]
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
            list(ex.map(lambda so: _run([hipcc, *flags, "-c", so[0], "-o", so[1]]), zip(srcs, objs)))
        # -Bsymbolic: this library's calls bind to ITS OWN definitions — it exports the same C ABI (and the same C++ helpers and kernel
        # launch stubs) as the product library that is already loaded with RTLD_GLOBAL, and must not be interposed by it
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-Wl,-Bsymbolic", *objs, "-o", out])
    return out


def build_vtensor(force=False):
    import torch
    ext = sysconfig.get_config_var("EXT_SUFFIX")
    out = os.path.join(PKG, "_vtensor" + ext)
    src = os.path.join(CSRC, "vtensor_ext.cpp")
    if force or _newer(out, [src]):
        tdir = os.path.dirname(torch.__file__)
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-w", "-DTORCH_EXTENSION_NAME=_vtensor",
              "-DTORCH_API_INCLUDE_EXTENSION_H", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
              "-I" + os.path.join(tdir, "include"), "-I" + os.path.join(tdir, "include", "torch", "csrc", "api", "include"),
              "-I" + sysconfig.get_paths()["include"], src,
              "-L" + os.path.join(tdir, "lib"), "-Wl,-rpath," + os.path.join(tdir, "lib"),
              "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-o", out])
    return out


def build_fake_backend(force=False):
    out = os.path.join(ROOT, "tests", "native", "libvattn_fake_backend.so")
    src = os.path.join(ROOT, "tests", "native", "fake_backend.cpp")
    if force or _newer(out, [src, os.path.join(ROOT, "include", "vattn.h")]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", src, "-o", out])
    return out


def build_reference_oracle():
    """oracle/_ref: the real reference allocator against a fake CUDA driver — only where
    /root/reference exists (this container); the GPU box uses the prebuilt file."""
    script = os.path.join(ROOT, "oracle", "build_ref.sh")
    ref = os.environ.get("VATTN_REFERENCE_DIR", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "vattention")):
        print("[build] reference sources absent: oracle/_ref not rebuilt")
        return None
    import glob
    have = glob.glob(os.path.join(ROOT, "oracle", "_ref", "vattention_ref*.so"))
    shim = glob.glob(os.path.join(ROOT, "oracle", "ref_shim", "*"))
    if have and not _newer(have[0], [f for f in shim if os.path.isfile(f)] + [script]):
        return have[0]
    _run(["bash", script])
    return True


def build_reference_pyref():
    """oracle/_ref/pyref: the reference's own wrapper / cache-engine Python files byte-compiled where they lie (oracle/build_pyref.py),
    so the GPU box can execute them (tests/ref_loader.py).  Only where /root/reference exists."""
    ref = os.environ.get("VATTN_REFERENCE_DIR", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "sarathi-lean")):
        print("[build] reference sources absent: oracle/_ref/pyref not rebuilt")
        return None
    _run([sys.executable, os.path.join(ROOT, "oracle", "build_pyref.py")])
    return True


def build_probe(force=False):
    """tools/power_ceiling_probe: whole-chip MFMA streams for bench.py's `roofline.other.power_ceiling` (what the board's power budget leaves of
    the MFMA peak on random operands).  Measurement infrastructure: a stand-alone binary, nothing of the product links it."""
    out = os.path.join(ROOT, "tools", "power_ceiling_probe")
    src = out + ".cpp"
    if force or _newer(out, [src]):
        _run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=" + ARCH, "-O3", "-Wno-unused-value", "-o", out, src])
    return out


def build_all(force=False):
    build_lib(force)
    build_lab(force)
    build_probe(force)
    build_vtensor(force)
    build_fake_backend(force)
    build_reference_oracle()
    build_reference_pyref()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
