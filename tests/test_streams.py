"""Synthetic-input generators of the measurement contract (deepmimic_amd/streams.py)."""
import numpy as np

from deepmimic_amd import streams


def test_philox4x32_10_known_answers():
    """Random123 kat_vectors for philox4x32-10 (Salmon et al.)."""
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for c, k, want in kat:
        assert [int(x) for x in streams.philox4x32_10(c, k)] == want


def test_noise_is_keyed_by_global_env_id_and_step():
    a = streams.normal_noise(np.arange(64), 5, 28)
    b = streams.normal_noise(np.arange(32, 64), 5, 28)          # second shard of a 2-rank job
    assert np.array_equal(a[32:], b)
    assert not np.array_equal(a, streams.normal_noise(np.arange(64), 6, 28))
    big = streams.normal_noise(np.arange(4096), 0, 28)
    assert abs(big.mean()) < 1e-3 and abs(big.std() - 0.05) < 1e-3 and np.isfinite(big).all()


def test_reset_phase_is_the_knuth_hash():
    ph = streams.reset_phase([0, 1, 2, 4095], 1.25)
    assert ph[0] == 0.0 and abs(ph[1] - 2654435761 / 2 ** 32 * 1.25) < 1e-15
    assert ((ph >= 0) & (ph < 1.25)).all()
