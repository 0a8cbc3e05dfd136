"""CPU: the engine's RNG layer against numpy itself (the third-party arithmetic on the path: numpy
Generator(PCG64(SeedSequence(seed))), gym.utils.seeding.np_random).  SURVEY Appendix A lists the algorithms."""
import ctypes
import numpy as np
import pytest

SEEDS = [0, 1, 7, 123, 2**31, 2**32 + 5, 2**63 + 11, 999999937]


def run_script(lib, seed, ops, args, mode=0):
    n = len(ops)
    ops = np.asarray(ops, np.int32)
    args = np.asarray(args, np.uint32)
    out = np.zeros(n, np.uint64)
    lib.cc4o_rng_script(ctypes.c_uint64(seed), mode, n, ops.ctypes.data_as(ctypes.c_void_p),
                        args.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


@pytest.mark.parametrize('seed', SEEDS)
def test_stream_matches_numpy(seed, oracle_lib):
    g = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
    rs = np.random.default_rng(seed + 1)
    ops, args, want = [], [], []
    for i in range(3000):
        op = int(rs.integers(0, 5))
        if op == 0:
            ops.append(0); args.append(0); want.append(np.float64(g.random()).view(np.uint64))
        elif op == 1:
            n = int(rs.choice([1, 2, 3, 4, 9, 57, 100, 254, 256, 9000, 10848, 2**31 + 3]))
            ops.append(1); args.append(n); want.append(int(g.integers(0, n)))
        elif op == 2:
            n = int(rs.integers(1, 93))
            lst = list(range(n)); g.shuffle(lst)
            ops.append(2); args.append(n); want.append(0)
        elif op == 3:
            n = int(rs.integers(1, 90))
            ops.append(1); args.append(n); want.append(int(g.choice(n)))          # choice(int)
        else:
            n = int(rs.integers(2, 40))
            v = g.choice([f'x{k}' for k in range(n)], replace=False)              # size-1 Floyd draw
            ops.append(1); args.append(n); want.append(int(str(v)[1:]))
    got = run_script(oracle_lib, seed, ops, args)
    assert [int(x) for x in got] == [int(x) for x in want]


def test_choice_with_probabilities_is_searchsorted_right(oracle_lib):
    # FiniteStateRedAgent._choose_host_and_action: choice(options, p=...) == cdf.searchsorted(random(), 'right')
    g1 = np.random.Generator(np.random.PCG64(np.random.SeedSequence(5)))
    g2 = np.random.Generator(np.random.PCG64(np.random.SeedSequence(5)))
    for p in ([.5, .25, .25], [.5, .5], [.25, .5, .25], [.75, .25], [.5, .5, 0.], [1., 0.], [.5, .25, .25, 0.], [.5, .5, 0.]):
        for _ in range(200):
            a = int(g1.choice(len(p), p=p))
            u = g2.random()
            cdf = np.cumsum(p); cdf /= cdf[-1]
            assert a == int(np.searchsorted(cdf, u, side='right'))


def test_philox_known_answer(oracle_lib):
    # Random123 known-answer vectors for philox4x32-10 (kat_vectors): counter/key all zero and all ones
    got = run_script(oracle_lib, 0, [3], [0], mode=1)   # key 0, counter (0,0,0,0)
    lo, hi = int(got[0]) & 0xFFFFFFFF, int(got[0]) >> 32
    assert (lo, hi) == (0x6627e8d5, 0xe169c58d)
