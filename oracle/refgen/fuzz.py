"""Differential fuzz: many seeds, random blue, reference vs oracle.
usage: fuzz.py start count [steps] [blue: sleep|random|builtin|decoy_one|decoy|decoy_restore|restore|remove|analyse|block|block_allow|mix] [init] [red: fsm|sleep|discovery|random] [green: enterprise|sleep]"""
import sys
from compare import run
start, count = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 500
blue = sys.argv[4] if len(sys.argv) > 4 else 'random'
init = sys.argv[5] if len(sys.argv) > 5 else 'ctor'
red = sys.argv[6] if len(sys.argv) > 6 else 'fsm'
green = sys.argv[7] if len(sys.argv) > 7 else 'enterprise'
bad = []
for seed in range(start, start + count):
    r = run(seed, steps, blue, init, verbose=True, red=red, green=green)
    if r is not None:
        bad.append((seed, r))
        print('FAIL', seed, r, flush=True)
print('DONE bad =', bad)
