"""Known-answer tests for the shared counter-based generator (oracle/philox.py):
Random123's published philox4x32_10 vectors (kat_vectors file of the Random123
distribution), plus distribution sanity."""
import numpy as np

from oracle import philox


def test_philox4x32_10_random123_kat():
  kat = [
      ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
      ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
      ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
       (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
  ]
  for ctr, key, want in kat:
    got = philox.philox4x32_10(np.array([ctr], np.uint32), key)[0]
    assert tuple(int(v) for v in got) == want


def test_normal_moments_and_streams():
  z = philox.normal(1 << 18, seed=0, stream_id=3, subseq=0)
  assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01
  assert np.isfinite(z).all()
  a = philox.normal(1000, 0, 0, 1)
  assert not np.array_equal(a, philox.normal(1000, 0, 0, 2))
  assert not np.array_equal(a, philox.normal(1000, 0, 1, 1))
  assert not np.array_equal(a, philox.normal(1000, 1, 0, 1))
  np.testing.assert_array_equal(a[:999], philox.normal(999, 0, 0, 1))  # prefix-stable
