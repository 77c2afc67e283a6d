#!/usr/bin/env python3
"""Inline-asm VMEM statements are invisible to the compiler's hazard recogniser.  On gfx950 a VALU instruction that writes an
SGPR (v_readlane / v_readfirstlane / v_cmp ... to an SGPR pair) must be followed by 5 wait states before a VMEM instruction
reads that SGPR.  This script scans the device assembly of kernels.hip (hipcc --offload-device-only -S) for inline-asm blocks
(;;#ASMSTART ... ;;#ASMEND) containing buffer_load and reports those whose scalar operands were written by a VALU
instruction fewer than 5 wait states earlier (s_nop N counts N+1, every other instruction 1).

    python deepcut-cnn_amd/build.py            # leaves deepcut-cnn_amd/lib/kernels.gfx950.s (build.device_asm() makes it on demand)
    python tools/check_asm_hazards.py deepcut-cnn_amd/lib/kernels.gfx950.s
"""
import re
import sys


def sregs(tok):
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def main(path):
    lines = open(path).read().split("\n")
    kernel, bad, blocks = None, 0, 0
    i = 0
    while i < len(lines):
        ln = lines[i]
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            kernel = m.group(1)
        if "#ASMSTART" in ln:
            j = i + 1
            body = []
            while "#ASMEND" not in lines[j]:
                body.append(lines[j].strip())
                j += 1
            vm = [b for b in body if b.startswith("buffer_load")]
            if vm:
                blocks += 1
                used = set()
                for b in vm:
                    for tok in re.split(r"[\s,]+", b):
                        used |= sregs(tok)
                # m0 source of the s_mov inside the block counts too
                for b in body:
                    if b.startswith("s_mov_b32 m0"):
                        used |= sregs(b.split(",")[1].strip())
                # wait states provided inside the block before the VMEM instruction
                inside = 0
                for b in body:
                    if b.startswith("buffer_load"):
                        break
                    mm = re.match(r"s_nop (\d+)", b)
                    inside += int(mm.group(1)) + 1 if mm else 1
                # walk back over preceding instructions
                states, k = inside, i - 1
                while k >= 0 and states < 5:
                    t = lines[k].strip()
                    k -= 1
                    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
                        if t.endswith(":"):
                            break  # basic-block boundary: give up (conservative: not reported)
                        continue
                    op = t.split()[0]
                    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or (op.startswith("v_cmp") and "_e64" in op):
                        dst = t.split()[1].rstrip(",")
                        if sregs(dst) & used:
                            print("HAZARD %s: line %d `%s` -> asm at line %d (%d wait states)" % (kernel, k + 2, t, i + 1, states))
                            bad += 1
                            break
                    mm = re.match(r"s_nop (\d+)", t)
                    states += int(mm.group(1)) + 1 if mm else 1
            i = j
        i += 1
    print("%d inline-asm VMEM blocks checked, %d hazards" % (blocks, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
