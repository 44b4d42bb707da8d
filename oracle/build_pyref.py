#!/usr/bin/env python3
"""Recipe for oracle/_ref/pyref: byte-compiles the REFERENCE's own Python files of the hot path — read where they lie under
/root/reference, never copied — so that tests can EXECUTE the reference's attention wrapper and cache engine against the MI355X
drop-ins on the GPU box, where /root/reference does not exist (same status as oracle/_ref/*.so: a build product of the
reference, git-ignored, test infrastructure only).

    sarathi-lean/sarathi/model_executor/attention/base_attention_wrapper.py
    sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_wrapper.py
    sarathi-lean/sarathi/worker/cache_engine/base_cache_engine.py
    sarathi-lean/sarathi/worker/cache_engine/vATTN_cache_engine.py
"""
import os
import py_compile
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("VATTN_REFERENCE_DIR", "/root/reference")
FILES = {
    "base_attention_wrapper": "sarathi-lean/sarathi/model_executor/attention/base_attention_wrapper.py",
    "vattention_flashattention_wrapper": "sarathi-lean/sarathi/model_executor/attention/vattention_flashattention_wrapper.py",
    "base_cache_engine": "sarathi-lean/sarathi/worker/cache_engine/base_cache_engine.py",
    "vATTN_cache_engine": "sarathi-lean/sarathi/worker/cache_engine/vATTN_cache_engine.py",
}


def main() -> int:
    if not os.path.isdir(os.path.join(REF, "sarathi-lean")):
        print("reference sources not present at %s: oracle/_ref/pyref not rebuilt" % REF)
        return 0
    out = os.path.join(HERE, "_ref", "pyref")
    os.makedirs(out, exist_ok=True)
    for name, rel in FILES.items():
        src = os.path.join(REF, rel)
        dst = os.path.join(out, name + ".pyc")
        py_compile.compile(src, cfile=dst, dfile="<reference>/" + rel, doraise=True, invalidation_mode=py_compile.PycInvalidationMode.UNCHECKED_HASH)
        print("compiled", rel, "->", os.path.relpath(dst, os.path.dirname(HERE)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
