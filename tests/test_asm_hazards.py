"""CPU: the hand-written inline-asm memory instructions of kernels.hip (LDS-DMA requests, untracked loads) are invisible to
the compiler's hazard recogniser.  tools/check_asm_hazards.py scans the gfx950 assembly of every kernel instantiation for a
VALU-written SGPR (v_readlane of a spilled SGPR, v_readfirstlane) read by such a statement fewer than 5 wait states later —
the bug class that produced wild addresses in the SGPR-spilling kernels of round 3.  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inline_asm_vmem_statements_have_their_wait_states(tmp_path):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    src = os.path.join(ROOT, "deepcut-cnn_amd", "csrc", "kernels.hip")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-c", src,
                           "-save-temps=obj", "-o", str(tmp_path / "kernels.o")], cwd=str(tmp_path))
    asm = [f for f in os.listdir(str(tmp_path)) if f.endswith(".s") and "amdgcn" in f]
    assert len(asm) == 1, asm
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_hazards.py"), str(tmp_path / asm[0])],
                         stdout=subprocess.PIPE, universal_newlines=True)
    assert out.returncode == 0, out.stdout[-4000:]
    assert "inline-asm VMEM blocks checked, 0 hazards" in out.stdout
