"""The hand-scheduled matrix-core walk keeps the wave's rows and accumulators in v64..v255 without the compiler knowing
(tools/gen_matrix_step_asm.py): the compiler's own code inside scanTilesMatrixPinned must never touch those registers,
must not spill between the steps, and must not use flat_ instructions there.  Nothing but the compiled code can show
that, so this test compiles the two units that contain the walk -- em2_scan_symmetric.hip (the scan kernels) and
em2_scan_sharded.hip (the tile kernels of the sharded scan) -- for gfx950 (no GPU needed) and checks the assembly; it also
checks that the committed asm header is what the generator writes."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "expressionmatrix2_amd", "csrc")


def test_generated_header_is_current():
    text = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_matrix_step_asm.py")], capture_output=True, text=True,
                          check=True).stdout
    assert text == open(os.path.join(CSRC, "em2_matrix_step_asm.h")).read()


import pytest


@pytest.mark.parametrize("unit", ["em2_scan_symmetric.hip", "em2_scan_sharded.hip"])
def test_compiled_walk_leaves_the_steps_registers_alone(tmp_path, unit):
    out = str(tmp_path / "unit.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-inline-asm",
           "--cuda-device-only", "-S", "-o", out, os.path.join(CSRC, unit)]
    done = subprocess.run(cmd, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr[-3000:]
    check = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_matrix_walk_registers.py"), out], capture_output=True,
                           text=True)
    assert check.returncode == 0, check.stdout[-3000:]
    assert "scanTilesMatrixPinned" in check.stdout


@pytest.mark.parametrize("knob", ["EM2_GEN_TILE_BOUND", "EM2_GEN_CMPX"])
def test_experiment_forms_of_the_step_still_assemble(tmp_path, knob):
    """The generator's experiment knobs (measured, not used) write steps the assembler accepts: the
    microbenchmark tools/ubench_matrix_step.hip compiles against each of them."""
    env = dict(os.environ)
    env[knob] = "1"
    text = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_matrix_step_asm.py")], capture_output=True, text=True,
                          check=True, env=env).stdout
    assert text != open(os.path.join(CSRC, "em2_matrix_step_asm.h")).read()
    (tmp_path / "em2_matrix_step_asm.h").write_text(text)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-w", "--cuda-device-only", "-c", "-I", str(tmp_path), "-o",
           str(tmp_path / "ubench.o"), os.path.join(ROOT, "tools", "ubench_matrix_step.hip")]
    done = subprocess.run(cmd, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr[-3000:]
