"""Prototype of the layered (time-varying) reference semantics: out = x*h0 + sum_L g_L * gather_L(x*h_L), where the
gather replays the B13 delay-line reader (src/MKLNonUniformConvolver.cpp:1653-1688) on the natural-time tail
convolutions.  Checked against the oracle's stateful emulation.  Development aid, not product code."""
import sys, os
import numpy as np
from scipy.signal import fftconvolve
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as O


def layered(x, h, B):
    p = O.plan(len(h), B)
    n = len(x)
    out = fftconvolve(x, h[:p.len[0]])[:n]
    ncb = n // B
    for l in range(1, p.numLayers):
        hl = h[p.offset[l]:p.offset[l] + p.len[l]]
        ynat = fftconvolve(x, hl)[:n]                      # delay-line content: natural time, no offset
        PL, oL, ppc = p.partSize[l], p.outputDelay[l], p.partsPerCallback[l]
        D = (p.numPartsIR[l] + ppc - 1) // ppc - 1
        bpp = PL // B
        R = 0
        for c in range(ncb):
            done = (c - D + 1) // bpp if c - D + 1 > 0 else 0       # tail blocks written by the end of Add() in callback c
            W = done * PL
            max_read = max(0, W - oL)
            start = max(R, max_read)
            if start + B > W:
                continue
            out[c * B:(c + 1) * B] += p.gain[l] * ynat[start:start + B]
            R = start + B
    return out


if __name__ == "__main__":
    for (L, B, nb) in ((131072, 1024, 400), (131072, 2048, 200), (131072, 512, 600), (524288, 1024, 900), (40000, 1024, 200)):
        h = O.gen_ir(L)
        x = O.gen_pcm(B * nb)
        c = O.Nuc(); c.set_impulse(h, B); y = c.run(x, B)
        z = layered(x, h, B)
        print(L, B, "lti", c.plan().ltiValid, "rms diff", np.sqrt(np.mean((y - z) ** 2)), "signal", np.sqrt(np.mean(y ** 2)))
