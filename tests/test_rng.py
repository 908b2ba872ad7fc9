"""Pin the oracle's PCG64 / SeedSequence / Lemire restatement against numpy itself (the generator gymnasium
hands to the reference: Wind_Farm_Env.py:689, WindEnv.py:24-35)."""
import ctypes as C

import numpy as np
import pytest


@pytest.mark.parametrize("seed", [0, 1, 5, 1234, 2**31 - 1, 2**32 + 7, 2**63 + 12345])
def test_uniform_stream_matches_numpy(oracle_lib, seed):
    L = oracle_lib.lib()
    n = 257
    out = np.zeros(n)
    L.wgo_rng_uniform(C.c_uint64(seed), C.c_int(n), C.c_double(6.0), C.c_double(15.0),
                      out.ctypes.data_as(C.c_void_p))
    ref = np.random.default_rng(seed).uniform(6.0, 15.0, size=n)
    ref_scalar = np.array([np.random.default_rng(seed).uniform(6.0, 15.0)])
    assert np.array_equal(out, ref)
    assert out[0] == ref_scalar[0]


@pytest.mark.parametrize("seed", [0, 3, 99, 31337])
def test_mixed_stream_matches_numpy(oracle_lib, seed):
    """3 doubles, integers(0, 100000), N doubles — the draw pattern of reset() with turbtype 'Random'."""
    L = oracle_lib.lib()
    rounds, tail = 40, 5
    ou = np.zeros(rounds * (3 + tail))
    oi = np.zeros(rounds, dtype=np.uint32)
    L.wgo_rng_mixed(C.c_uint64(seed), C.c_int(rounds), C.c_uint32(100000), C.c_int(tail),
                    ou.ctypes.data_as(C.c_void_p), oi.ctypes.data_as(C.c_void_p))
    g = np.random.default_rng(seed)
    ru, ri = [], []
    for _ in range(rounds):
        ru += [g.uniform(0, 1) for _ in range(3)]
        ri.append(int(g.integers(0, 100000)))
        ru += list(g.uniform(0, 1, size=tail))
    assert np.array_equal(ou, np.array(ru))
    assert np.array_equal(oi, np.array(ri, dtype=np.uint32))


def test_philox_known_answer(oracle_lib):
    """Random123 known-answer vectors for Philox4x32-10."""
    L = oracle_lib.lib()

    def run(ctr, key):
        c = (C.c_uint32 * 4)(*ctr)
        k = (C.c_uint32 * 2)(*key)
        o = (C.c_uint32 * 4)()
        L.wgo_rng_philox(c, k, o)
        return [int(x) for x in o]

    assert run([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
