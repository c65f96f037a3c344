#!/usr/bin/env python3
"""numpy model of the four-step real FFT used for partitions above 4096 samples (fft_kernels.hip: k_big_*).

Frame of 2P real samples -> z[n] = x[2n] + i x[2n+1], n < M = P -> complex FFT of M = M1 * 512 points as
  step 1: for each column n2 < 512: M1-point FFT over n1 (n = n1 * 512 + n2), times W_M^(n2 k1)       -> A[k1][n2]
  step 2: for each row k1 < M1: 512-point FFT over n2                                                  -> Z[k1 + M1 k2]
stored PERMUTED as Zp[k1 * 512 + k2]; the real-FFT split pairs (k1, k2) with (M1 - k1, 511 - k2) (k1 > 0) or
(0, (512 - k2) % 512).  The packed spectrum keeps that permuted order (the MAC is element-wise); element 0 = (DC, Nyquist).
The inverse runs the mirror image and returns the frame's second half.  Checked against numpy.fft here."""
import numpy as np


def fwd(frame, M1):
    P = frame.size // 2
    M = P
    z = frame[0::2] + 1j * frame[1::2]
    a = z.reshape(M1, 512)
    A = np.fft.fft(a, axis=0)                                   # over n1 -> k1
    n2 = np.arange(512)[None, :]
    k1 = np.arange(M1)[:, None]
    A = A * np.exp(-2j * np.pi * n2 * k1 / M)
    Zp = np.fft.fft(A, axis=1)                                  # over n2 -> k2 ; Zp[k1, k2] = Z[k1 + M1 k2]
    # split in permuted order
    k2 = np.arange(512)[None, :]
    pk1 = (M1 - k1) % M1
    pk2 = np.where(k1 == 0, (512 - k2) % 512, 511 - k2)
    Zm = Zp[pk1, pk2]
    E = 0.5 * (Zp + np.conj(Zm))
    O = -0.5j * (Zp - np.conj(Zm))
    k = k1 + M1 * k2
    X = E + np.exp(-2j * np.pi * k / (2 * M)) * O
    Xp = X.copy()
    Xp[0, 0] = (Zp[0, 0].real + Zp[0, 0].imag) + 1j * (Zp[0, 0].real - Zp[0, 0].imag)    # (DC, Nyquist)
    return Xp.reshape(-1)


def inv(Xp, M1):
    M = Xp.size
    Y = Xp.reshape(M1, 512)
    k1 = np.arange(M1)[:, None]
    k2 = np.arange(512)[None, :]
    pk1 = (M1 - k1) % M1
    pk2 = np.where(k1 == 0, (512 - k2) % 512, 511 - k2)
    Ym = Y[pk1, pk2]
    k = k1 + M1 * k2
    E = 0.5 * (Y + np.conj(Ym))
    D = 0.5 * (Y - np.conj(Ym))
    O = D * np.exp(2j * np.pi * k / (2 * M))
    Z = E + 1j * O
    Z[0, 0] = 0.5 * (Y[0, 0].real + Y[0, 0].imag) + 0.5j * (Y[0, 0].real - Y[0, 0].imag)
    # inverse four-step: rows (over k2 -> n2), twiddle, columns (over k1 -> n1)
    A = np.fft.ifft(Z, axis=1) * 512
    n2 = np.arange(512)[None, :]
    A = A * np.exp(2j * np.pi * n2 * k1 / M)
    z = np.fft.ifft(A, axis=0) * M1 / M
    z = z.reshape(-1)
    out = np.empty(2 * M)
    out[0::2] = z.real
    out[1::2] = z.imag
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for M1 in (16, 32, 64):
        P = 512 * M1
        x = rng.standard_normal(2 * P)
        Xp = fwd(x, M1)
        ref = np.fft.rfft(x)
        k = (np.arange(M1)[:, None] + M1 * np.arange(512)[None, :]).reshape(-1)
        want = ref[k].copy()
        want[0] = ref[0].real + 1j * ref[P].real
        print(M1, "fwd err", np.abs(Xp - want).max(), "inv err", np.abs(inv(Xp, M1) - x).max())
