"""numpy prototype of the wave-level 512-point complex FFT / 1024-point real FFT index math used by
convopeq_amd/csrc/kernels_fft.hip (one wave = one transform, 8 points per lane, three radix-8
passes with two LDS exchanges).  Development aid: validates lane/register/LDS indexing against
numpy.fft before it is written in HIP.  Not product code."""
import numpy as np

N2 = 512
W512 = np.exp(-2j * np.pi * np.arange(512) / 512)
W1024 = np.exp(-2j * np.pi * np.arange(513) / 1024)
W8 = np.exp(-2j * np.pi * np.arange(8) / 8)


def dft8(v, inverse=False):
    """v: [8, lanes] -> DFT over axis 0"""
    w = np.conj(W8) if inverse else W8
    M = w[(np.arange(8)[:, None] * np.arange(8)[None, :]) % 8]
    return M @ v


def cfft512_lanes(v, inverse=False):
    """v[j][l] = z[l + 64 j]  ->  out[r][l2] = Z[l2 + 64 r]"""
    tw = np.conj(W512) if inverse else W512
    lane = np.arange(64)
    A = dft8(v, inverse)                                    # A[p][l]
    for p in range(8):
        A[p] *= tw[(lane * p) % 512]
    E1 = np.zeros(72 * 8, dtype=complex)
    for p in range(8):
        E1[72 * p + lane] = A[p]
    pp, cc = lane >> 3, lane & 7
    u = np.stack([E1[72 * pp + 8 * b + cc] for b in range(8)])   # u[b][l']
    Bq = dft8(u, inverse)
    for q in range(8):
        Bq[q] *= tw[(8 * cc * q) % 512]
    E2 = np.zeros(66 * 8, dtype=complex)
    for q in range(8):
        E2[66 * cc + pp + 8 * q] = Bq[q]
    w = np.stack([E2[66 * c + lane] for c in range(8)])          # w[c][l'']
    return dft8(w, inverse)                                      # out[r][l''] = Z[l'' + 64 r]


def rfft1024_packed(x):
    z = x[0::2] + 1j * x[1::2]
    lane = np.arange(64)
    v = np.stack([z[lane + 64 * j] for j in range(8)])
    Z = cfft512_lanes(v)
    E3 = np.zeros(512, dtype=complex)
    for r in range(8):
        E3[lane + 64 * r] = Z[r]
    out = np.zeros(512, dtype=complex)
    for r in range(8):
        k = lane + 64 * r
        zc = np.conj(E3[(512 - k) & 511])
        e = 0.5 * (Z[r] + zc)
        o = -0.5j * (Z[r] - zc)
        X = e + W1024[k] * o
        X = np.where(k == 0, (Z[r].real + Z[r].imag) + 1j * (Z[r].real - Z[r].imag), X)
        out[k] = X
    return out     # packed: out[0] = DC + i*Nyquist


def irfft1024_packed_second_half(Y):
    lane = np.arange(64)
    v = []
    for j in range(8):
        k = lane + 64 * j
        yk = Y[k]
        yc = np.conj(Y[(512 - k) & 511])
        e = 0.5 * (yk + yc)
        o = 0.5 * (yk - yc) * np.conj(W1024[k])
        zk = e + 1j * o
        z0 = 0.5 * (Y[0].real + Y[0].imag) + 0.5j * (Y[0].real - Y[0].imag)
        v.append(np.where(k == 0, z0, zk))
    z = cfft512_lanes(np.stack(v), inverse=True) / 512.0
    out = np.zeros(512)
    for r in range(4, 8):
        n = lane + 64 * r
        out[2 * n - 512] = z[r].real
        out[2 * n + 1 - 512] = z[r].imag
    return out


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    x = rng.standard_normal(1024)
    P = rfft1024_packed(x)
    R = np.fft.rfft(x)
    ref = R[:512].copy()
    ref[0] = R[0].real + 1j * R[512].real
    print("fwd err", np.abs(P - ref).max())
    y = irfft1024_packed_second_half(P)
    print("inv err", np.abs(y - x[512:]).max())
