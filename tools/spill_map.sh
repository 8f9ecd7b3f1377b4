#!/bin/bash
# Where does a build of the step kernel spill?  Compiles arb_kernels.hip (ARB_QUICK) with line tables and the given extra
# flags, disassembles the plain float32 44-row kernel and prints a histogram of scratch loads/stores per source line.
# usage: tools/spill_map.sh <name> [extra hipcc flags]      e.g. tools/spill_map.sh w3 -DARB_WAVES_PER_EU=3
R=$(cd "$(dirname "$0")/.." && pwd); B=/opt/rocm/lib/llvm/bin
n=$1; shift
T=$R/build/spill_$n; mkdir -p $T
hipcc --offload-arch=gfx950 -O3 -ffp-contract=on -std=c++17 -fPIC -I $R/include -DARB_QUICK=1 -gline-tables-only "$@" -shared -o $T/lib.so $R/arboris_python_amd/csrc/arb_kernels.hip 2>/dev/null
$B/llvm-objcopy -O binary --only-section=.hip_fatbin $T/lib.so $T/fb
$B/clang-offload-bundler --unbundle --type=o --input=$T/fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co 2>/dev/null
$B/llvm-objdump -d -l --no-show-raw-insn $T/co > $T/dis.txt

python3 $R/tools/spill_hist.py $T/dis.txt
