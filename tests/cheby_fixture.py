"""A synthetic TEMPO2 predictor text (ChebyModelSet) for the tests: the Chebyshev series of a known phase law
    phase(t, f) = PHI0 + F0 tau + F1 tau^2 / 2 + KF (f - 1382) tau,   tau = seconds since the start of the segment,
so that the parsers / evaluators can be checked against the closed form.  (tempo2 itself is not in the image.)"""
from decimal import Decimal, getcontext

import numpy as np

PHI0 = Decimal("12345678901.625")
F0, F1, KF, DC = 11.1946499395, -1.5666e-11, 3.0e-9, 2.5e4
FREQ = (1182.0, 1582.0)


def law(tau, f):
    return F0 * tau + 0.5 * F1 * tau * tau + KF * (f - 1382.0) * tau


def cheby_text(day=55299, frac0="0.05", frac1="0.25", nseg=2, nx=12, ny=2):
    getcontext().prec = 40
    out = ["ChebyModelSet %d segments" % nseg]
    d0, d1 = float(frac0), float(frac1)
    span = (d1 - d0) / nseg
    for k in range(nseg):
        a, b = d0 + k * span, d0 + (k + 1) * span
        xs = np.cos(np.pi * (np.arange(nx) + 0.5) / nx)
        ys = np.cos(np.pi * (np.arange(ny) + 0.5) / ny)
        tau = (xs + 1.0) * 0.5 * (b - a) * 86400.0
        f = FREQ[0] + (ys + 1.0) * 0.5 * (FREQ[1] - FREQ[0])
        ci = np.cos(np.pi * np.outer(np.arange(nx), np.arange(nx) + 0.5) / nx)       # [i][k]
        cj = np.cos(np.pi * np.outer(np.arange(ny), np.arange(ny) + 0.5) / ny)       # [j][l]
        # absolute phase at the start of segment k: PHI0 + law(start of segment), folded into the constant term (x 4)
        t_start = k * span * 86400.0
        phi_k = PHI0 + Decimal(repr(F0)) * Decimal(repr(t_start)) + Decimal(repr(0.5 * F1)) * Decimal(repr(t_start)) ** 2
        slope = [F0 + F1 * t_start, F1]          # the law restarts at the segment start: tau -> t_start + tau
        g = (slope[0] * tau[:, None] + 0.5 * slope[1] * tau[:, None] ** 2 + KF * (f[None, :] - 1382.0) * (t_start + tau[:, None]))
        c = 4.0 / (nx * ny) * ci @ g @ cj.T
        c00 = Decimal(repr(float(c[0, 0]))) + 4 * phi_k
        out += ["ChebyModel BEGIN", "PSRNAME J0835-4510", "SITENAME PKS",
                "TIME_RANGE %d.%s %d.%s" % (day, ("%.10f" % a)[2:], day, ("%.10f" % b)[2:]),
                "FREQ_RANGE %r %r" % FREQ, "DISPERSION_CONSTANT %r" % DC, "NCOEFF_TIME %d" % nx, "NCOEFF_FREQ %d" % ny]
        for i in range(nx):
            vals = [("%s" % c00) if (i == 0 and j == 0) else repr(float(c[i, j])) for j in range(ny)]
            out.append("COEFFS " + " ".join(vals))
        out.append("ChebyModel END")
    return "\n".join(out) + "\n"


def closed_form(sec_of_day, f, frac0="0.05"):
    """(absolute phase as Decimal, spin frequency) of the law at `sec_of_day` seconds of the predictor's day."""
    tau = sec_of_day - float(frac0) * 86400.0
    ph = PHI0 + Decimal(repr(F0)) * Decimal(repr(tau)) + Decimal(repr(0.5 * F1)) * Decimal(repr(tau)) ** 2 \
        + Decimal(repr(KF * (f - 1382.0))) * Decimal(repr(tau)) + Decimal(repr(DC / (f * f)))
    return ph, F0 + F1 * tau + KF * (f - 1382.0)
