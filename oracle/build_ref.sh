#!/bin/bash
# Recipe for oracle/_ref: builds the REFERENCE vattention allocator (host code only) from
# /root/reference/vattention/*.{cu,h} — read in place, never copied — against the fake CUDA
# driver in oracle/ref_shim/.  Output: oracle/_ref/vattention_ref*.so (git-ignored).
# Needs: g++, this image's torch headers, pybind11.  No GPU, no CUDA toolkit.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${VATTN_REFERENCE_DIR:-/root/reference}/vattention"
[ -f "$REF/vattention.cu" ] || { echo "reference sources not present at $REF: skipping oracle/_ref build"; exit 0; }
OUT="$HERE/_ref"; mkdir -p "$OUT"
PY=${PYTHON:-python3}
TORCH_DIR=$($PY -c "import torch,os;print(os.path.dirname(torch.__file__))")
PYINC=$($PY -c "import sysconfig;print(sysconfig.get_paths()['include'])")
EXT=$($PY -c "import sysconfig;print(sysconfig.get_config_var('EXT_SUFFIX'))")
g++ -O0 -std=c++17 -fPIC -shared -w -fpermissive \
    -DTORCH_EXTENSION_NAME=vattention_ref -DTORCH_API_INCLUDE_EXTENSION_H -D_GLIBCXX_USE_CXX11_ABI=$($PY -c "import torch;print(int(torch._C._GLIBCXX_USE_CXX11_ABI))") \
    -I"$HERE/ref_shim" -I"$REF" -I"$TORCH_DIR/include" -I"$TORCH_DIR/include/torch/csrc/api/include" -I"$PYINC" \
    "$HERE/ref_shim/ref_module.cpp" "$HERE/ref_shim/fakecuda.cpp" \
    -L"$TORCH_DIR/lib" -Wl,-rpath,"$TORCH_DIR/lib" -ltorch -ltorch_cpu -lc10 -ltorch_python \
    -o "$OUT/vattention_ref$EXT"
echo "built $OUT/vattention_ref$EXT"
