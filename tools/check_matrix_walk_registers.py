#!/usr/bin/env python3
"""Checks the compiled kernels of the hand-scheduled matrix-core walk (fsp4ScanMatrixPinnedKernel,
fsp4TileMatrixPinnedKernel; csrc/em2_scan_symmetric.hip and csrc/em2_scan_sharded.hip): the steps keep the wave's rows and accumulators in v64..v255
without the compiler knowing (tools/gen_matrix_step_asm.py), so the compiler's own code in those kernels must never
touch those registers -- nor v28 / v29, which hold the lane's record offsets from step to step --, should not spill between
the steps, and must not use flat_ instructions there (their out-of-order completion would break the counted LDS waits).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -o /tmp/sym.s em2_scan_symmetric.hip
    python3 tools/check_matrix_walk_registers.py /tmp/sym.s
"""
import re
import sys

OWNED_FIRST = 64          # v30..v63 are the steps' temporaries: dead between steps, the compiler may use them there
# ... but v28 / v29 hold the lane's record offsets from step to step, and v48..v63 are the ring of column fragments, which
# the step of a pair's first tile leaves IN FLIGHT for the step of the second (gen_matrix_step_asm.py, CARRY)
PERSISTENT = (28, 29) + tuple(range(48, 64))


def functions(lines):
    name, start = None, 0
    for i, line in enumerate(lines):
        m = re.match(r"^(_Z\w*scanTilesMatrix(?:Pinned|Wide)\w*):", line)
        if m:
            name, start = m.group(1), i
        elif name and (line.startswith(".Lfunc_end") or ".end_amdhsa_kernel" in line):
            yield name, lines[start:i]
            name = None


def registers(line):
    text = line.split(";")[0]
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        yield from range(int(m.group(1)), int(m.group(2)) + 1)
    for m in re.finditer(r"\bv(\d+)\b", text):
        yield int(m.group(1))


def main():
    lines = open(sys.argv[1]).read().split("\n")
    failures = 0
    found = 0
    for name, body in functions(lines):
        found += 1
        in_asm = False
        first_step = last_step = None
        for i, line in enumerate(body):
            if "v_mfma" in line:
                first_step = i if first_step is None else first_step
                last_step = i
        for i, line in enumerate(body):
            if "#ASMSTART" in line:
                in_asm = True
                continue
            if "#ASMEND" in line:
                in_asm = False
                continue
            if in_asm or line.lstrip().startswith((";", ".")):
                continue
            between = first_step is not None and first_step <= i <= last_step
            bad = [r for r in registers(line) if r >= OWNED_FIRST or (between and r in PERSISTENT)]
            if bad:
                print("%s: compiler code touches v%d between the steps: %s" % (name, bad[0], line.strip()))
                failures += 1
            if between and re.search(r"\bs_(buffer_)?load_", line):
                # the steps count their LDS waits (lgkmcnt); scalar loads share the counter and return out of order
                print("%s: scalar load between the steps: %s" % (name, line.strip()))
                failures += 1
            if "flat_" in line and first_step is not None and first_step <= i <= last_step:
                print("%s: flat instruction between the steps: %s" % (name, line.strip()))
                failures += 1
        spills = sum(1 for i, line in enumerate(body) if "scratch_" in line and first_step is not None and first_step <= i <= last_step)
        print("%s: %d lines, %d scratch accesses between the first and the last step" % (name, len(body), spills))
    if not found:
        print("no ...MatrixPinnedKernel found")
        return 1
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
