"""numpy prototype of the time-parallel SVF band algorithm (zero-state chunk runs + state scan + correction),
checked against the sequential oracle.  Development aid for svf_kernels.hip, not product code."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as O

LD = np.longdouble


def tables(c, lc, nchunk_log2=6):
    a1, a2, a3, m0, m1, m2 = [LD(v) for v in (c.a1, c.a2, c.a3, c.m0, c.m1, c.m2)]
    A = np.array([[2 * a1 - 1, -2 * a2], [2 * a2, 1 - 2 * a3]], dtype=LD)
    C = np.array([m1 * a1 + m2 * a2, -m1 * a2 + m2 * (1 - a3)], dtype=LD)
    G = np.zeros((lc, 2), dtype=LD)
    P = np.eye(2, dtype=LD)
    for i in range(lc):
        G[i] = C @ P
        P = A @ P
    M = [P]                      # A^lc
    for k in range(1, nchunk_log2):
        M.append(M[-1] @ M[-1])
    return G.astype(np.float64), [m.astype(np.float64) for m in M]


def band_tp(x, c, sat, s_in, lc=64, nch=64):
    """one span of nch*lc samples"""
    G, M = tables(c, lc)
    X = x.reshape(nch, lc)
    ic1 = np.zeros(nch); ic2 = np.zeros(nch)
    yz = np.empty_like(X)
    for i in range(lc):
        v0 = X[:, i]
        v3 = v0 - ic2
        v1 = c.a1 * ic1 + c.a2 * v3
        v2 = c.a2 * ic1 + (c.a3 * v3 + ic2)
        ic1 = 2 * v1 - ic1
        ic2 = 2 * v2 - ic2
        yz[:, i] = c.m0 * v0 + (c.m1 * v1 + c.m2 * v2)
    e = np.stack([ic1, ic2], axis=1)                      # [nch, 2]
    e[0] += M[0] @ s_in
    S = e.copy()
    for k in range(6):
        sh = 1 << k
        S[sh:] = S[sh:] + (S[:-sh] @ M[k].T)              # uses pre-update values (numpy evaluates RHS first)
    s0 = np.vstack([s_in[None, :], S[:-1]])
    y = yz + s0[:, :1] * G[None, :, 0] + s0[:, 1:2] * G[None, :, 1]
    if sat > 0:
        xc = np.clip(y, -4.5, 4.5)
        th = xc * (27 + xc * xc) / (27 + 9 * xc * xc)
        y = y * (1 - sat) + th * sat
    y = np.clip(y, -100, 100)
    return y.reshape(-1), S[-1]


def run(params, n=8 * 4096, sat=0.2):
    x = O.gen_pcm(n)
    ref = x.copy()
    tp = x.copy()
    for b in range(20):
        bp = params.bands[b]
        c = O.svf_design(bp.type, bp.frequency, bp.gain, bp.q, 48000.0)
        st = np.zeros(2)
        O.lib().orc_svf_band_stereo_lane(O.dp(ref), n, c, O.dp(st), sat)
        s_in = np.zeros(2)
        out = []
        for sp in range(n // 4096):
            y, s_in = band_tp(tp[sp * 4096:(sp + 1) * 4096], c, sat, s_in)
            out.append(y)
        tp = np.concatenate(out)
        print(b, "max diff", np.abs(tp - ref).max(), "state diff", np.abs(s_in - st).max(), "|state|", np.abs(st).max())
    return np.abs(tp - ref).max()


if __name__ == "__main__":
    print("bench preset:", run(O.eq_params_bench(0.2)))
    p = O.eq_params_bench(0.2)
    for b in range(20):
        p.bands[b].q = 20.0
        p.bands[b].gain = 12.0 if b % 2 == 0 else -12.0
    p.bands[0].frequency = 20.0
    print("extreme Q:", run(p))
