"""numpy model of the in-workgroup Stockham index math used by dspsr_amd/csrc/wgfft.h
(butterfly u = G*tid + g = col + T*(p + P*s); reads u + i*(F/R)*T; twiddle W_{RQ}^{ks};
writes (s*P*R + k*P + p)*T + col) and of the three-pass filterbank decomposition
(columns FFT + twiddle, rows FFT, Hermitian split of rows s and Rr-1-s)."""
import numpy as np
import pytest


def wgfft_model(x, logF, logT, sign, pts=32):
    F, T = 1 << logF, 1 << logT
    nt = F * T // pts
    buf = x.reshape(-1).copy()
    q, r = divmod(logF, 4)
    radices = [16] * q + ([1 << r] if r else [])
    if not radices:
        radices = [1]
    logP = 0
    for R in radices:
        logR = R.bit_length() - 1
        G = pts // R
        Q = F >> (logP + logR)
        new = np.empty_like(buf)
        stride = (F // R) * T
        seen = set()
        for tid in range(nt):
            for g in range(G):
                u = G * tid + g
                assert u not in seen
                seen.add(u)
                col = u & (T - 1)
                rest = u >> logT
                p = rest & ((1 << logP) - 1)
                s = rest >> logP
                v = np.array([buf[u + i * stride] for i in range(R)])
                k = np.arange(R)
                out = np.array([np.sum(v * np.exp(sign * 2j * np.pi * np.arange(R) * kk / R)) for kk in k])
                out = out * np.exp(sign * 2j * np.pi * k * s / (R * Q))
                for kk in k:
                    pos = (s << (logP + logR)) + (kk << logP) + p
                    new[(pos << logT) | col] = out[kk]
        assert len(seen) == (F // R) * T
        buf = new
        logP += logR
    return buf.reshape(F, T)


@pytest.mark.parametrize("logF,logT", [(4, 3), (5, 2), (6, 1), (7, 2), (8, 1), (9, 1), (3, 4), (1, 5), (2, 4), (0, 5)])
def test_stage_index_math(logF, logT):
    rng = np.random.default_rng(0)
    F, T = 1 << logF, 1 << logT
    x = rng.standard_normal((F, T)) + 1j * rng.standard_normal((F, T))
    for sign in (-1, 1):
        y = wgfft_model(x, logF, logT, sign)
        ref = np.fft.fft(x, axis=0) if sign < 0 else np.fft.ifft(x, axis=0) * F
        assert np.allclose(y, ref)


def test_three_pass_decomposition_and_hermitian_split():
    rng = np.random.default_rng(1)
    C, M = 8, 16
    N = C * M
    L, R = 2 * N, 2 * C
    x0, x1 = rng.standard_normal(L), rng.standard_normal(L)
    w = x0 + 1j * x1
    A = np.fft.fft(w.reshape(M, R), axis=0)
    ka, nb = np.arange(M)[:, None], np.arange(R)[None, :]
    A = A * np.exp(-2j * np.pi * ka * nb / L)
    rows = np.fft.fft(A, axis=1).T                      # rows[s'][m] = W[m + M*s']
    assert np.allclose(rows.reshape(-1), np.fft.fft(w))
    X0ref, X1ref = np.fft.rfft(x0)[:N], np.fft.rfft(x1)[:N]
    for s in range(C):
        a = rows[s]
        b = np.empty(M, complex)
        b[1:] = rows[R - 1 - s][::-1][:M - 1]
        b[0] = rows[(R - s) % R][0]
        assert np.allclose((a + np.conj(b)) / 2, X0ref[s * M:(s + 1) * M])
        assert np.allclose((a - np.conj(b)) / 2j, X1ref[s * M:(s + 1) * M])


def test_persistent_item_order_is_a_permutation():
    """persistent_item(): chunked XCD-aware dealing of work items covers every item exactly once."""
    def item(b, grid, j, run):
        if grid & 7:
            return b + j * grid
        nxl = grid >> 3
        q = j * nxl + (b >> 3)
        return (q // run) * (8 * run) + (b & 7) * run + (q % run)
    for grid, run, total in [(256, 32, 2048), (256, 4, 1000), (8, 3, 100), (5, 7, 33), (248, 16, 4096)]:
        got = []
        for b in range(grid):
            j = 0
            while True:
                it = item(b, grid, j, run)
                if it >= total:
                    break
                got.append(it)
                j += 1
        assert sorted(got) == list(range(total))
