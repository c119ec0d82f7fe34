#!/usr/bin/env python
"""Scan a gfx950 assembly listing (hipcc --cuda-device-only -S) for an instruction that touches the RESULT registers of an
inline-asm MFMA too soon after it.

The compiler pads its own MFMAs (hazard recogniser) but sees an inline-asm statement as an opaque instruction: a spill
store, a merge copy or an epilogue multiply that the scheduler / register allocator places right behind the last MFMAs of
a k-loop reads registers the matrix pipe has not written yet (CDNA3 ISA 4.5: XDL write VGPR -> VALU / VMEM / LDS read
needs 11 wait states for an 8-pass MFMA, 19 for a 16-pass one; no hardware interlock).  The scan walks each kernel in
program order (a linear approximation of the control flow), ages every MFMA result register by the wait states of the
instructions that follow (1 each, s_nop N = N + 1, an MFMA = 4) and reports every non-MFMA access younger than NEED.

usage: mfma_result_hazard_scan.py file.s [NEED=16]"""
import re
import sys

REG = re.compile(r'\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b')


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
        else:
            out.add((m.group(4), int(m.group(5))))
    return out


def main(path, need=16):
    kern, age, bad, in_asm = '?', {}, 0, False
    for line in open(path):
        t = line.split(';')[0].strip() if not line.strip().startswith(';;#') else line.strip()
        if t.endswith(':') and t.startswith('_Z'):
            kern, age = t, {}
        if t.startswith(';;#ASMSTART'):
            in_asm = True
            continue
        if t.startswith(';;#ASMEND'):
            in_asm = False
            continue
        if not t or t[0] in ';.' or t.endswith(':'):
            continue
        parts = t.split(None, 1)
        op, rest = parts[0], parts[1] if len(parts) > 1 else ''
        if op.startswith('v_mfma'):
            if in_asm:
                dst = regs(rest.split(',')[0])
                for r in dst:
                    age[r] = 0
                cost = 4
                for r in list(age):
                    if r not in dst:
                        age[r] += cost
            else:
                for r in list(age):
                    age[r] += 4
            continue
        if op == 's_nop':
            cost = int(rest) + 1
        else:
            cost = 1
            if op[0] in 'vdgsb' and not op.startswith('s_'):
                hit = [r for r in regs(rest) if r in age and age[r] < need]
                if hit:
                    bad += 1
                    if bad <= 12:
                        print(kern[:60], '|', t[:90], '| age', min(age[r] for r in hit))
        for r in list(age):
            age[r] += cost
            if age[r] > 64:
                del age[r]
    print('accesses to young inline-asm MFMA results:', bad)
    return bad


if __name__ == '__main__':
    sys.exit(1 if main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16) else 0)
