"""Contrast-transfer-function filters for particle stacks (reference src/ctf.py), host-side preprocessing.

`parse_ctf` reads the 8-column whitespace table (defocus cs voltage apix bfactor ampcont dfdiff dfang);
`ctf_filter` turns every row into a real-space (n, m) kernel: the CTF is evaluated on the FFT frequency grid
(scaled by 1/apix), inverse transformed, centred with fftshift and negated (reference src/ctf.py:32-55).  The kernels
are applied per image on the GPU by tvae_ctf_corr (reference train_particles.py:298-302).
"""
import numpy as np


def compute_2d_ctf(freqs, dfu, dfv, dfang, volt, cs, w, bfactor=None):
    """CTF at 2-D spatial frequencies `freqs` (N,2) [1/A]; units as in the reference (kV, mm, fraction)."""
    volt = volt * 1000.0
    cs = cs * 10 ** 7
    lam = 12.2639 / np.sqrt(volt + 0.97845e-6 * volt ** 2)        # relativistic electron wavelength [A]
    fx, fy = freqs[:, 0], freqs[:, 1]
    s2 = fx ** 2 + fy ** 2
    df = 0.5 * (dfu + dfv + (dfu - dfv) * np.cos(2 * (np.arctan2(fy, fx) - dfang)))
    gamma = 2 * np.pi * (-0.5 * df * lam * s2 + 0.25 * cs * lam ** 3 * s2 ** 2)
    ctf = np.sqrt(1 - w ** 2) * np.sin(gamma) - w * np.cos(gamma)
    if bfactor is not None:
        ctf = ctf * np.exp(-bfactor / 4 * s2)
    return ctf.astype(freqs.dtype)


def parse_ctf(f):
    import pandas as pd
    t = pd.read_csv(f, sep=r'\s+', header=None)
    t.columns = ['defocus', 'cs', 'voltage', 'apix', 'bfactor', 'ampcont', 'dfdiff', 'dfang']
    return t


def ctf_filter(ctf_params, n, m, scale=1):
    gy, gx = np.meshgrid(np.fft.fftfreq(n), np.fft.fftfreq(m), indexing='ij')
    freqs = np.stack([gy.ravel(), gx.ravel()], 1)
    out = np.zeros((len(ctf_params), n, m), dtype=np.float32)
    for i in range(len(ctf_params)):
        apix = ctf_params.apix[i] * scale
        c = compute_2d_ctf(freqs / apix, ctf_params.defocus[i] * 10000, ctf_params.defocus[i] * 10000,
                           2 * np.pi * ctf_params.dfang[i] / 360, ctf_params.voltage[i], ctf_params.cs[i],
                           ctf_params.ampcont[i] / 100, ctf_params.bfactor[i])
        out[i] = -np.fft.fftshift(np.fft.ifft2(c.reshape(n, m))).real
    return out
