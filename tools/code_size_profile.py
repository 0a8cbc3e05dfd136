#!/usr/bin/env python3
"""Where a kernel's code bytes come from: bytes per source function (innermost frame and inclusive), and how many inlined copies of
each function the kernel holds.

  hipcc --offload-arch=gfx950 <the Makefile's OPT flags> -gline-tables-only --cuda-device-only -c -o u.o csrc/cc4_k_run1.hip
  clang-offload-bundler --unbundle --type=o --input=u.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=u.co
  python tools/code_size_profile.py u.co _Z13k_run_philox18StepArgs7RunArgs8XchgArgs

Line tables do not change the generated code (checked: the kernel's size equals profiles/kernel_resources.txt's code_bytes).
"""
import collections
import re
import subprocess
import sys

LLVM = '/opt/rocm/lib/llvm/bin/'


def main():
    co, sym = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
    under = sys.argv[4] if len(sys.argv) > 4 else None      # only instructions inlined (at any depth) into this function; counts VECTOR instructions
    dis = subprocess.run([LLVM + 'llvm-objdump', '-d', '--disassemble-symbols=' + sym, co], capture_output=True, text=True).stdout
    insn = []
    for line in dis.splitlines():
        m = re.search(r'// ([0-9A-F]{12}): ((?:[0-9A-F]{8} ?)+)', line)
        if m:
            op = line.split()[0]
            insn.append((int(m.group(1), 16), 4 * len(m.group(2).split()), op))
    total = sum(b for _, b, _ in insn)
    print(f'{sym}: {len(insn)} instructions, {total} bytes')
    inp = '\n'.join(f'0x{a:x}' for a, _, _ in insn) + '\n'
    out = subprocess.run([LLVM + 'llvm-symbolizer', '--obj=' + co, '--inlines', '--functions=short', '--output-style=LLVM'],
                         input=inp, capture_output=True, text=True).stdout
    blocks = [b for b in out.split('\n\n') if b.strip()]
    assert len(blocks) == len(insn), (len(blocks), len(insn))
    inner = collections.Counter()
    incl = collections.Counter()
    copies = collections.defaultdict(set)
    for (addr, nb, op), b in zip(insn, blocks):
        ls = b.strip().splitlines()
        frames = [(ls[i], ls[i + 1]) for i in range(0, len(ls) - 1, 2)]      # innermost first: (function, file:line:col)
        if under:
            if under not in [f for f, _ in frames]:
                continue
            frames = frames[:[f for f, _ in frames].index(under) + 1]
            valu = op.startswith('v_') and not op.startswith(('v_readlane', 'v_readfirstlane', 'v_writelane'))
            nb = 1 if valu else 0
        inner[frames[0][0]] += nb
        seen = set()
        for d, (fn, _) in enumerate(frames):
            if fn not in seen:
                incl[fn] += nb
                seen.add(fn)
            # an inlined copy is identified by the chain of call sites above it
            copies[fn].add(tuple(loc for _, loc in frames[d + 1:]))
    if under:
        total = sum(inner.values())
        print(f'static VECTOR instructions inlined into {under}: {total}')
    print(f'\n{"bytes (own)":>12} {"%":>6} {"copies":>6}  function')
    for fn, nb in inner.most_common(top):
        print(f'{nb:12d} {100 * nb / total:6.1f} {len(copies[fn]):6d}  {fn}')
    print(f'\n{"bytes (incl)":>12} {"%":>6} {"copies":>6} {"per copy":>9}  function')
    for fn, nb in incl.most_common(top):
        print(f'{nb:12d} {100 * nb / total:6.1f} {len(copies[fn]):6d} {nb // max(1, len(copies[fn])):9d}  {fn}')


if __name__ == '__main__':
    main()
