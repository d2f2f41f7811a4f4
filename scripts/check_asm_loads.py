#!/usr/bin/env python3
"""Build check (wired into csrc/Makefile and __graft_entry__.build()): no inline-asm statement of the HIP sources may contain a
vector-memory load with a REGISTER destination (an LDS read only with its `s_waitcnt lgkmcnt(0)` in the same statement).

Why: a load issued from one asm statement and waited for (`s_waitcnt vmcnt`) in another leaves a window in which hipcc considers the
destination register defined and may copy or spill it before the data has landed (LESSONS.md lesson 24: wrong images at batch 256 next to
other streams).  Loads into registers are therefore always the compiler's own (`__builtin_amdgcn_raw_buffer_load_*`, plain loads): it
places their s_waitcnt itself.  What asm statements may issue are LDS-DMA copies (`buffer_load_* ... lds`, `global_load_lds_*`), which
have no register destination, and waits / barriers, which name no data register.

Exit status 1 and one line per offending statement otherwise."""
import glob
import os
import re
import sys

LOAD = re.compile(r'\b(?:global|buffer|flat|scratch)_load_\w+[^\n\\]*', re.S)
LDS_READ = re.compile(r'\bds_(?:read\w*|\w+_rtn_\w+)')


def asm_statements(text):
    """(line number, concatenated string literals) of every asm statement."""
    for m in re.finditer(r'\basm\s*(?:volatile)?\s*\(', text):
        i, depth, start = m.end(), 1, m.end()
        while i < len(text) and depth:
            c = text[i]
            if c == '"':                       # skip a string literal
                i += 1
                while text[i] != '"':
                    i += 2 if text[i] == '\\' else 1
            elif c == '(':
                depth += 1
            elif c == ')':
                depth -= 1
            i += 1
        body = text[start:i]
        lits, j = [], 0                        # the instruction text: the string literals in front of the first ':' outside a literal
        while j < len(body) and body[j] != ':':
            if body[j] == '"':
                k = j + 1
                while body[k] != '"':
                    k += 2 if body[k] == '\\' else 1
                lits.append(body[j + 1:k])
                j = k
            j += 1
        yield text.count('\n', 0, m.start()) + 1, ''.join(lits).replace('\\n', '\n').replace('\\t', ' ')


def main():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'pyopenvino_amd', 'csrc')
    bad = []
    for path in sorted(glob.glob(os.path.join(root, '*.hip')) + glob.glob(os.path.join(root, '*.h'))):
        text = open(path).read()
        for line, code in asm_statements(text):
            # an LDS read in asm is allowed only with its wait in the SAME statement (the spin loop of pvhip_wino.hip's counters)
            insts = code.split('\n')
            for k, inst in enumerate(insts):
                if LDS_READ.search(inst) and not any('s_waitcnt lgkmcnt(0)' in later for later in insts[k + 1:]):
                    bad.append('{}:{}: inline asm reads LDS into a register without waiting for it in the same statement: {}'.format(
                        os.path.relpath(path), line, inst.strip()))
            for inst in code.split('\n'):
                m = LOAD.search(inst)
                if m and not (re.search(r'\blds\b', inst) or '_load_lds_' in inst):
                    bad.append('{}:{}: inline asm loads into a register: {}'.format(os.path.relpath(path), line, inst.strip()))
    for b in bad:
        print(b)
    if bad:
        print('check_asm_loads: register loads must be compiler-tracked (builtins / plain loads), see LESSONS.md lesson 24')
        return 1
    print('check_asm_loads: ok (inline asm issues LDS-DMA copies, waits and barriers only)')
    return 0


if __name__ == '__main__':
    sys.exit(main())
