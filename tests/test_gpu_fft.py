"""The partition FFT kernel families in isolation (what replaces Intel IPP's real FFT,
src/FFTBackend.cpp:123-150 -- forward unscaled, CCS-like layout, 1 / N on the inverse) through the diagnostic entry
cpq_diag_partition_fft: forward transform of every overlap-save frame [previous block | block] against numpy.fft.rfft in
fp64, inverse of the same spectra against the input (the second half of the inverse of a frame's spectrum is the block
itself), and the round trip.  One case per kernel family:

  P = 64 / 128 / 256      k_rfft_*_ols_generic   (radix-2 Stockham through LDS)
  P = 512                 k_rfft_*_ols           (wave-level 512-point transform, 8 points per lane)
  P = 1024 / 2048         k_rfft_*_ols_wg        (one workgroup per transform, mixed radix)
  P = 4096                k_rfft_*_ols_p4        (four-step inside a workgroup; spectra stored permuted)
  P = 8192 ... 131072     k_big_cols_* / k_big_rows_*   (four-step through a scratch buffer; spectra stored permuted;
                          65536 / 131072: 128- / 256-point columns, 32 / 16 columns per workgroup)

Tolerance: 4e-15 of the largest spectral magnitude on the forward transform and of the largest sample against numpy's
inverse, 2e-15 on the round trip
(log2(N) = 7 ... 16 butterfly levels of fp64 rounding; measured values are printed)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: the -m gpu tests need a gfx950 device")
    import convopeq_amd
    return convopeq_amd


def _bins(P):
    """storage element -> bin of the packed spectrum (element 0 = (DC, Nyquist))"""
    e = np.arange(P)
    if P <= 2048:
        return e
    m1 = P // 512
    return (e // 512) + m1 * (e % 512)


@pytest.mark.parametrize("P", [64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072])
def test_partition_fft_forward_inverse_round_trip(amd, P):
    from convopeq_amd import _capi
    lib = _capi.load()
    n_ch, T = 3, 5 if P <= 4096 else 3
    rng = np.random.default_rng(1000 + P)
    x = rng.standard_normal((n_ch, T, P))
    x[1, 1] = 0.0                                           # a silent block
    x[2, 0, :] = 0.0
    x[2, 0, 17] = 1.0                                       # an impulse
    spec = np.empty((n_ch, T, P, 2))
    out = np.empty((n_ch, T, P))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    assert lib.cpq_diag_partition_fft(P, n_ch, T, dp(np.ascontiguousarray(x)), dp(spec), dp(out)) == 0
    bins = _bins(P)
    assert sorted(bins.tolist()) == list(range(P))          # the documented storage order is a permutation
    worst_f, worst_b = 0.0, 0.0
    for c in range(n_ch):
        prev = np.zeros(P)
        for t in range(T):
            ref = np.fft.rfft(np.concatenate([prev, x[c, t]]))          # 2P real points -> P + 1 bins
            got = spec[c, t, :, 0] + 1j * spec[c, t, :, 1]
            scale = max(np.abs(ref).max(), 1e-300)
            # element 0 packs the two real bins
            assert abs(got[0].real - ref[0].real) <= 4e-15 * scale and abs(got[0].imag - ref[P].real) <= 4e-15 * scale, (P, c, t)
            assert abs(ref[0].imag) < 1e-12 * scale and abs(ref[P].imag) < 1e-12 * scale
            err = np.abs(got[1:] - ref[bins[1:]]).max() / scale
            worst_f = max(worst_f, err)
            assert err <= 4e-15, (P, c, t, err)
            # the inverse of the frame's spectrum, second half = the block (numpy's irfft as the independent inverse)
            back = np.fft.irfft(ref, 2 * P)[P:]
            worst_b = max(worst_b, np.abs(out[c, t] - back).max() / max(np.abs(back).max(), 1.0))
            prev = x[c, t]
    rt = np.abs(out - x).max() / np.abs(x).max()
    print(f"P = {P}: forward {worst_f:.2e} of the largest bin, inverse {worst_b:.2e}, round trip {rt:.2e}")
    assert worst_b <= 4e-15 and rt <= 2e-15, (P, worst_b, rt)          # (worst_b is the distance between two roundings of the same inverse)
