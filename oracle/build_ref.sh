#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- never imported by the product path.
#
# Builds the reference's own native entropy-coder modules (pybind11) from the
# sources where they lie under /root/reference, into oracle/_ref/ (git-ignored).
# Nothing is copied: the compiler reads the reference files in place.
#   compressai.ans   <- src/compress/cpp_exts/rans/rans_interface.cpp (+ third_party/ryg_rans/rans64.h)
#   compressai._CXX  <- src/compress/cpp_exts/ops/ops.cpp
# The vendored rans_interface.hpp has had its #includes stripped (hpp:17-21), so
# the real system/pybind11 headers it needs are force-included on the command line.
set -euo pipefail
REF=${PC_REFERENCE_ROOT:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref/compressai"
if [ ! -d "$REF/src/compress/cpp_exts" ]; then
  echo "[build_ref] $REF not present: keeping prebuilt oracle/_ref (if any)"; exit 0
fi
mkdir -p "$OUT"
PYINC="$(python3 -m pybind11 --includes)"
EXT="$(python3 -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')"
CXXFLAGS="-O3 -std=c++17 -shared -fPIC -fvisibility=hidden"
g++ $CXXFLAGS $PYINC -I"$REF/src/third_party/ryg_rans" -I"$REF/src/compress/cpp_exts/rans" \
    -include pybind11/pybind11.h -include pybind11/stl.h -include vector -include string -include cstdint \
    "$REF/src/compress/cpp_exts/rans/rans_interface.cpp" -o "$OUT/ans$EXT"
g++ $CXXFLAGS $PYINC "$REF/src/compress/cpp_exts/ops/ops.cpp" -o "$OUT/_CXX$EXT"
echo "[build_ref] built $OUT/ans$EXT and _CXX$EXT"
