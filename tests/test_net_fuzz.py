"""Mutated model definitions through net construction (tools/fuzz_net_definitions.py), in a process of its own: a crash or a
hang of the library is a finding.  CPU only — parsing, InsertSplits, shape inference and lowering are host code."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 2])
def test_mutated_definitions_build_or_are_refused(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_net_definitions.py"), str(seed), "400"], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    built, refused = [int(v) for v in __import__("re").findall(r"(\d+) built, (\d+) refused", r.stdout)[0]]
    assert built + refused == 400 and built > 10 and refused > 10, r.stdout  # both outcomes are exercised
