#!/usr/bin/env python
"""ISA lint for the gfx950 build of libcc4.so (every translation unit under cage_challenge_4_amd/csrc).

hipcc (ROCm 7.2, -O3) was seen to lower a wave-uniform `cond ? a : b` whose compare stayed on the VALU
(v_cmp -> vcc) into an `s_cselect` -- which reads SCC, not VCC -- inside one unrolled copy of a loop in k_step; the
result depended on whatever SALU compare ran last.  The parity tests caught it on the GPU; this scan catches the
pattern at build time, without a GPU: every SCC reader (s_cselect / s_cbranch_scc* / s_addc / s_subb / s_cmov) must
have an SCC writer before it in its straight-line block when a VALU compare (v_cmp) is the only compare in it.

usage: isa_scan.py [file.s ...]      (no argument: the ISA files the last build left under cage_challenge_4_amd/csrc/build/)
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCC_WRITER = re.compile(r'^\s*s_(cmp|cmpk|and_|or_|xor_|andn2|orn2|nand|nor|xnor|add_|sub_|addc|subb|lshl|lshr|ashr|bfe|min_|max_|'
                        r'abs|not_|bitcmp|absdiff|wqm|quadmask|bcnt|ff0|ff1|flbit|addk|mulk)')
SCC_READER = re.compile(r'^\s*s_(cselect|cbranch_scc|addc|subb|cmov)')
LABEL = re.compile(r'^\.LBB|^[A-Za-z_][\w$.]*:')


def scan(text):
    """Returns [(line_no, reader, context)] for SCC readers whose nearest preceding compare is a VALU compare."""
    lines = text.split('\n')
    bad = []
    for i, l in enumerate(lines):
        if not SCC_READER.match(l):
            continue
        j = i - 1
        vcmp = None
        while j >= 0:
            t = lines[j]
            if SCC_WRITER.match(t):
                break                                            # a producer in straight-line code: fine
            if LABEL.match(t) or re.match(r'^\s*s_c?branch', t):   # reached the top of the block without one
                if vcmp is not None:
                    bad.append((i + 1, l.strip(), vcmp))
                break
            if vcmp is None and re.match(r'^\s*v_cmp', t):
                vcmp = t.strip()
            j -= 1
    return bad


def main():
    files = sys.argv[1:]
    if not files:       # the ISA of every translation unit as the last build left it (cage_challenge_4_amd/csrc/build/<unit>/*-gfx950.s: -save-temps)
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, 'cage_challenge_4_amd', 'csrc', 'build', '*', '*-gfx950.s')))
        if not files:
            raise RuntimeError('no ISA files: build the library first (python __graft_entry__.py)')
    total = 0
    for f in files:
        bad = scan(open(f).read())
        for b in bad:
            print(os.path.basename(os.path.dirname(f)) + ': SCC read after VALU compare: line %d: %s   <<  %s' % b)
        total += len(bad)
    print('isa_scan: %d suspicious SCC reads in %d file(s)' % (total, len(files)))
    return 1 if total else 0


if __name__ == '__main__':
    try:
        sys.exit(main())            # 0 clean, 1 findings
    except Exception as exc:        # the lint itself could not run (compile step failed, no temp dir, ...)
        print('isa_scan: could not run:', exc)
        sys.exit(2)
