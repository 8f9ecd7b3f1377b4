#!/bin/bash
# Development: one-translation-unit build of the float32 / 44-row kernels only (-DARB_QUICK, ~1 min; ARB_QUICK=2 in the
# environment adds two column sets / the inspect kernel / the optional inputs) for same-box A/B runs:
#   tools/quick_build.sh <name> [extra flags]  ->  build/ab/<name>.so   (load with ARBSTEP_LIB=build/ab/<name>.so)
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/build/ab
n=$1; shift
hipcc --offload-arch=gfx950 -O3 -ffp-contract=on -std=c++17 -fPIC -I $R/include -DARB_QUICK=${ARB_QUICK:-1} -DARB_DEVELOPMENT "$@" -shared -o $R/build/ab/$n.so $R/arboris_python_amd/csrc/arb_kernels.hip 2>&1 | grep -i -A5 "error" | head -20
ls -la $R/build/ab/$n.so
