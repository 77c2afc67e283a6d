"""CPU: the hand-written inline-asm memory instructions of kernels.hip (LDS-DMA requests, untracked loads) are invisible to
the compiler's hazard recogniser.  tools/check_asm_hazards.py scans the gfx950 assembly of every kernel instantiation for a
VALU-written SGPR (v_readlane of a spilled SGPR, v_readfirstlane) read by such a statement fewer than 5 wait states later —
the bug class that produced wild addresses in the SGPR-spilling kernels of round 3.  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_inline_asm_vmem_statements_have_their_wait_states():
    # the device assembly of kernels.hip with the library's own flags: build.py leaves it beside the objects (keyed by a hash
    # of the sources, the flags and the compiler) when it compiles the library, so a suite that runs after build() does not
    # compile the translation unit a second time; anything stale or missing is compiled here
    sys.path.insert(0, os.path.join(ROOT, "deepcut-cnn_amd"))
    try:
        import build as _build
    finally:
        sys.path.pop(0)
    asm = _build.device_asm()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_asm_hazards.py"), asm],
                         stdout=subprocess.PIPE, universal_newlines=True)
    assert out.returncode == 0, out.stdout[-4000:]
    assert "inline-asm VMEM blocks checked, 0 hazards" in out.stdout
