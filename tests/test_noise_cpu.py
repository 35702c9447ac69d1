"""oracle/noise_ref.py (the numpy restatement of ldm_gaussian_noise) against the published Philox4x32-10 known-answer vectors
(Random123 kat_vectors: Salmon et al., SC'11) and the properties the batch-sharded sampler relies on."""
import numpy as np

from oracle import noise_ref as nr


def test_philox4x32_10_known_answers():
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = nr.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]
        assert tuple(int(v) for v in got) == want


def test_rows_do_not_depend_on_the_shard_and_are_standard_normal():
    full = nr.gaussian_rows(12345678901234567, 7, 0, 64, 4 * 64 * 64)
    part = nr.gaussian_rows(12345678901234567, 7, 40, 9, 4 * 64 * 64)
    assert np.array_equal(full[40:49], part)
    assert not np.array_equal(full[0], full[1])
    assert not np.array_equal(part, nr.gaussian_rows(12345678901234567, 8, 40, 9, 4 * 64 * 64))
    assert not np.array_equal(part, nr.gaussian_rows(12345678901234566, 7, 40, 9, 4 * 64 * 64))
    assert abs(float(full.mean())) < 3e-3 and abs(float(full.std()) - 1.0) < 3e-3
    assert abs(float((full ** 4).mean()) - 3.0) < 0.05 and np.isfinite(full).all()
    ragged = nr.gaussian_rows(5, -1, 3, 2, 7)          # row length not a multiple of 4: the first 7 of the row's 8 draws
    assert ragged.shape == (2, 7) and np.array_equal(ragged, nr.gaussian_rows(5, -1, 3, 2, 8)[:, :7])
