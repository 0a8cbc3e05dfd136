"""CPU: libcc4.so loads and exports every symbol include/cc4.h declares; without a GPU it refuses to create an
environment (no CPU fallback).  No compute calls are made here."""
import ctypes
import os
import re
import pytest
from conftest import ROOT


def header_symbols(names=('cc4.h', 'cc4_debug.h')):
    """Every entry point the headers under include/ declare: the drop-in boundary (cc4.h) and the debug / test hooks (cc4_debug.h)."""
    out = set()
    for name in names:
        txt = open(os.path.join(ROOT, 'include', name)).read()
        txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
        out |= set(re.findall(r'\b(cc4_[a-z_0-9]+)\s*\(', txt))
    return sorted(out)


def test_every_declared_symbol_is_exported():
    from cage_challenge_4_amd import _lib
    syms = header_symbols()
    assert len(syms) >= 25
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), f'{s} declared under include/ but not exported by libcc4.so'
    # (cc4_debug_policy_probe: a concluded experiment, compiled only with -DCC4_POLICY_PROBE -- declared, not exported by the product library)
    assert set(syms) == set(_lib.SIGNATURES), 'python binding and header disagree on the ABI'
    assert not [s for s in header_symbols(('cc4.h',)) if s.startswith('cc4_debug_')], 'debug hooks belong in include/cc4_debug.h'


def test_constants_agree_with_header():
    from cage_challenge_4_amd import _lib
    lib = _lib.load()
    assert lib.cc4_state_bytes() % 16 == 0
    txt = open(os.path.join(ROOT, 'include', 'cc4.h')).read()
    assert f'#define CC4_OBS_PER_ENV {_lib.OBS_PER_ENV}' in txt
    assert f'#define CC4_MASK_PER_ENV {_lib.MASK_PER_ENV}' in txt
    assert lib.cc4_algorithmic_bytes_per_env_step() == 2 * lib.cc4_state_bytes() + 4 * 578 + 20 + 9


def test_no_gpu_means_loud_failure(has_gpu):
    if has_gpu:
        pytest.skip('a GPU is visible')
    from cage_challenge_4_amd import CC4VecEnv
    from cage_challenge_4_amd._lib import CC4Error
    with pytest.raises(CC4Error, match='no HIP device'):
        CC4VecEnv(4)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'cage_challenge_4_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.h', '.hip', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                assert 'liboracle' not in txt and 'oracle_binding' not in txt and '../oracle' not in txt, f
