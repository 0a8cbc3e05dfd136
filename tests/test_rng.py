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


def _philox4x32_10(ctr, key):
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = list(ctr); k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k1) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return c


def run_script2(lib, seed, ops, t, mode=1):
    n = len(ops)
    ops = np.asarray(ops, np.int32)
    t = np.asarray(t, np.float64)
    out = np.zeros(n, np.uint64)
    lib.cc4o_rng_script2(ctypes.c_uint64(seed), mode, n, ops.ctypes.data_as(ctypes.c_void_p), t.ctypes.data_as(ctypes.c_void_p),
                         out.ctypes.data_as(ctypes.c_void_p))
    return out


def test_counter_mode_words_are_the_philox_blocks_in_order(oracle_lib):
    """The generator's four-word queue (r06): stream `id` of step 7 of key `seed` hands out the words of block 0, then block 1, ... of the counter
    (block, stream, step, episode 0), c[0] first -- whatever mix of draws and stream switches came before."""
    seed = 0x123456789ABCDEF
    key = (seed & 0xFFFFFFFF, seed >> 32)
    ops, t = [], []
    for stream in (0x500, 0x601, 0x300 + 17):
        ops += [4] + [3] * 11
        t += [float(stream)] + [0.0] * 11
    got = run_script2(oracle_lib, seed, ops, t)
    i = 0
    for stream in (0x500, 0x601, 0x300 + 17):
        i += 1
        want = sum((_philox4x32_10((blk, stream, 7, 0), key) for blk in range(3)), [])[:11]
        assert [int(x) for x in got[i:i + 11]] == want
        i += 11


def test_threshold_draws_equal_the_double_comparison(oracle_lib):
    """rng_random_lt / _le / _quarter (integer compares of the 32-bit word in the counter mode) against the expressions they replace, on the same words:
    k * 2**-32 < t, <= t, int(4 * k * 2**-32) -- including thresholds that are exact multiples of 2**-32, 0, 1, negatives, > 1 and NaN."""
    rs = np.random.default_rng(3)
    n = 4000
    thr = np.concatenate([rs.random(n - 1000), rs.integers(0, 2**32, 400) / 2.0**32, [0.0, 1.0, -0.5, 1.5, float('nan'), 2.0**-40, 1 - 2.0**-33, 0.01, 0.05, 0.75] * 60])
    rs.shuffle(thr)
    seed = 99
    words = run_script2(oracle_lib, seed, [3] * n, [0.0] * n)
    u = words.astype(np.float64) * 2.0**-32
    with np.errstate(invalid='ignore'):
        assert [int(x) for x in run_script2(oracle_lib, seed, [0] * n, thr)] == [int(a < b) for a, b in zip(u, thr)]
        assert [int(x) for x in run_script2(oracle_lib, seed, [1] * n, thr)] == [int(a <= b) for a, b in zip(u, thr)]
    assert [int(x) for x in run_script2(oracle_lib, seed, [2] * n, thr)] == [int(a * 4.0) for a in u]
    # the words themselves fall exactly on a threshold: k * 2**-32 < k * 2**-32 is false, <= is true
    exact = words[:200].astype(np.float64) * 2.0**-32
    assert not any(run_script2(oracle_lib, seed, [0] * 200, exact)) and all(run_script2(oracle_lib, seed, [1] * 200, exact))
