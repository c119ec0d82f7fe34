#!/usr/bin/env python
"""Scan a gfx950 assembly listing (hipcc -save-temps) for a vector instruction that writes an A / B operand register of an
inline-asm MFMA fewer than 3 instructions ahead of it (hipcc pads nothing around asm statements: the VALU -> MFMA operand
hazard needs its wait states by hand).  usage: mfma_hazard_scan.py file.s"""
import re
import sys


def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def main(path):
    kern, hist, bad, in_asm = '?', [], 0, False
    for line in open(path):
        t = line.strip()
        if t.endswith(':') and t.startswith('_Z'):
            kern, hist = t, []
        if t.startswith(';;#ASMSTART'):
            in_asm = True
        elif t.startswith(';;#ASMEND'):
            in_asm = False
        if not t or t[0] in ';.' or t.endswith(':'):
            continue
        op = t.split()[0]
        if op.startswith('v_mfma') and in_asm:          # (the compiler pads its own MFMAs)
            ops = [x.strip(',') for x in t.split()[1:]]
            src = regs(ops[1]) | regs(ops[2])
            for back in hist[-2:]:
                bop = back.split()[0]
                if bop.startswith('v_') and not bop.startswith('v_mfma') and len(back.split()) > 1:
                    if regs(back.split()[1].strip(',')) & src:
                        bad += 1
                        if bad <= 10:
                            print(kern[:70], '|', back, '->', t)
        hist.append(t)
    print('suspicious VALU -> MFMA operand pairs:', bad)
    return bad


if __name__ == '__main__':
    sys.exit(1 if main(sys.argv[1]) else 0)
