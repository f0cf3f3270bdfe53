#!/usr/bin/env bash
# Builds the REAL reference host code (from where it lies under /root/reference, nothing is copied)
# into oracle/_ref/ -- test infrastructure that pins the oracle and exercises the drop-in boundary:
#
#   libref{1,2,3}d.so            src/{1,2,3}d/main.cu compiled as C++ with -Dmain=ref_main: exports the
#                                reference's own test_cpu() (the only checker the reference has) and its
#                                whole harness as ref_main(argc, argv).  The gpu_*() symbols it calls
#                                resolve to THIS repo's shim library, i.e. the reference harness runs
#                                unchanged on top of the HIP engine.
#   ref_lorastencil_{1,2,3}d_check   the same sources built as executables with -DCHECK_ERROR: the
#                                reference's self-check (one sweep vs test_cpu, 1e-7) against the HIP engine.
#
# The three main.cu files are plain host C++ (no CUDA API, macros unexpanded); the reference's CUDA
# kernels (gpu*.cu: nvcuda::wmma + PTX cp.async) cannot be built here and are not attempted.
# Only runs where /root/reference exists (this container); the GPU box uses the prebuilt files.
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
REF="${LORA_REFERENCE_DIR:-/root/reference}"
LIB="$HERE/../lorastencil_amd/lib"
OUT="$HERE/_ref"
if [ ! -d "$REF/src" ]; then
    echo "build_ref.sh: $REF not present, keeping prebuilt oracle/_ref" >&2
    exit 0
fi
if [ ! -f "$LIB/liblorastencil_shims.so" ]; then
    echo "build_ref.sh: build lorastencil_amd/csrc first (liblorastencil_shims.so missing)" >&2
    exit 1
fi
mkdir -p "$OUT"
CXXFLAGS="-O2 -std=c++17 -fPIC -x c++ -w"
LINK="-L$LIB -llorastencil_shims -llorastencil_hip -Wl,-rpath,\$ORIGIN/../../lorastencil_amd/lib"
for d in 1 2 3; do
    src="$REF/src/${d}d/main.cu"
    inc="-I$REF/src/${d}d"
    g++ $CXXFLAGS $inc -Dmain=ref_main -shared -o "$OUT/libref${d}d.so" "$src" $LINK
    g++ $CXXFLAGS $inc -DCHECK_ERROR -o "$OUT/ref_lorastencil_${d}d_check" "$src" $LINK
done
echo "built: $(ls "$OUT" | tr '\n' ' ')"
