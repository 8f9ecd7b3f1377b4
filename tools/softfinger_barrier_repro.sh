#!/bin/bash
# Reproducer of the second compiler problem without a root cause: in float32 lane-per-world code
# (arb_softfinger_test_kernel<float>: one lane per input tuple, csrc/arb_aux_kernels.h) hipcc 7.2 for gfx950 at -O2 / -O3
# hands softfinger_slide_finish a WRONG incoming force f after the long inlined float64 root finder: f_new is right,
# df = f_new - f comes out ~1e17 -- for one input tuple in about two million solves.  -O1 and the host build are right.
# The shipped code pins f in vector registers before the root finder (arb_math.h: softfinger_try, the `asm volatile("" :
# "+v"(f[0]) ...)` line); -DARB_NO_SOFTFINGER_BARRIER compiles it without the pin.
#
# 1. build a development library without the pin (cross-compiles without a GPU, ~2 min):
#      ARB_QUICK=2 tools/quick_build.sh nobarrier -DARB_NO_SOFTFINGER_BARRIER
# 2. on a GPU box, the device unit test that holds the input (the harvested tuples of tests/golden/g3_contacts.npz and the
#    1e5 solves harvested from an episode; it compares the device's lane-per-tuple solve with the host build, bit for bit):
#      ARBSTEP_LIB=$PWD/build/ab/nobarrier.so python -m pytest tests/test_gpu_device_solve.py -m gpu -x -q
#    expected WITH the bug: test_device_solve_on_1e5_harvested_tuples_f32 fails with one tuple whose dforce is ~1e17.
# 3. what to look at: the ISA of arb_softfinger_test_kernel<float> between the cone test and the call of the 4x4 solve --
#    the four VGPRs holding f[0..3] are dead across the root finder in the miscompiled build (the register allocator reuses
#    them and reloads f from the wrong stack slot); tools/isa_loops.py prints the live ranges around a label.
# State of knowledge (round 6): not re-run since round 3 (it needs a GPU box and the bug is rare); the pin costs nothing
# measurable (one wave-level no-op), so it stays until a compiler release is shown to be clean with step 2.
echo "read the comments in this file; nothing is run" >&2
