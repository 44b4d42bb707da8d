#!/usr/bin/env python3
"""build/base/libvattn_amd.so = the product library of the last COMMIT (git archive HEAD), for same-box A/B timing against the working
tree (tools/p64_ab.sh).  usage: python tools/build_base.py [rev]"""
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"
tmp = "/tmp/vattn_base"
shutil.rmtree(tmp, ignore_errors=True)
os.makedirs(tmp)
tar = subprocess.run(["git", "-C", ROOT, "archive", rev, "vattention_amd/csrc", "include", "vattention_amd/build.py"], check=True, capture_output=True).stdout
subprocess.run(["tar", "-x", "-C", tmp], input=tar, check=True)
src = open(os.path.join(tmp, "vattention_amd/build.py")).read()
files = [x.strip().strip("\"'") for x in re.search(r"LIB_SOURCES\s*=\s*\(([^)]*)\)", src).group(1).split(",") if x.strip()]
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread", "-Wno-unused-value", "-Wno-inline-asm", "-I" + tmp + "/include"]
os.makedirs(tmp + "/obj")


def cc(f):
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", tmp + "/vattention_amd/csrc/" + f, "-o", tmp + "/obj/" + f + ".o"])


with ThreadPoolExecutor(8) as ex:
    list(ex.map(cc, files))
os.makedirs(ROOT + "/build/base", exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-pthread", "-Wl,-Bsymbolic", *[tmp + "/obj/" + f + ".o" for f in files], "-o", ROOT + "/build/base/libvattn_amd.so"])
print("built build/base/libvattn_amd.so from", subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", rev], capture_output=True, text=True).stdout.strip())
