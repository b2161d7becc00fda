"""Index model of the two-pass path (csrc/filterbank.hip FB_HAS(6)) in numpy: the decomposition the kernels implement --
   n = nb + Fb*na, Fa = 2^14 as even/odd 2^13-point transforms + one radix-2 step (k_fwd_col1), A[a][nb][j] with ka = a*M + j,
   twiddle W_L^{nb*ka}, Fb-point transforms over nb -> kb, channel c = a + (Fa/M)*kb, chirp, inverse M-point transforms
   (k_rows_inv) -- equals the direct formulation of Filterbank.C:561-662 (forward FFT of L points, Response::operate, backward
   FFTs of M contiguous bins per channel).  Small sizes (Fa = 2^6): the index algebra does not depend on them."""
import numpy as np
import pytest


@pytest.mark.parametrize("logFa,logM,logFb", [(6, 3, 2), (6, 4, 1), (8, 4, 3)])
def test_two_pass_decomposition_equals_direct(logFa, logM, logFb):
    Fa, M, Fb = 1 << logFa, 1 << logM, 1 << logFb
    L = Fa * Fb
    C = L // M
    rng = np.random.default_rng(1)
    x = rng.standard_normal(L) + 1j * rng.standard_normal(L)
    H = np.exp(1j * rng.uniform(-np.pi, np.pi, L))
    # direct: forward FFT, x response, backward (unnormalised) FFT of each channel's M bins
    X = np.fft.fft(x) * H
    want = np.stack([np.fft.ifft(X[c * M:(c + 1) * M]) * M for c in range(C)])
    # pass 1: column nb = samples nb + Fb*na; even / odd halves + radix-2 combine with W = exp(-2 pi i / Fa)
    A = np.zeros((Fa // M, Fb, M), complex)
    for nb in range(Fb):
        y = x[nb::Fb]
        E, O = np.fft.fft(y[0::2]), np.fft.fft(y[1::2])
        k = np.arange(Fa // 2)
        w = np.exp(-2j * np.pi * k / Fa)
        Y = np.concatenate([E + w * O, E - w * O])
        assert np.allclose(Y, np.fft.fft(y))
        A[:, nb, :] = Y.reshape(Fa // M, M)                  # A[a][nb][j], ka = a*M + j
    # pass 2: tile a: twiddle, Fb-point transform over nb, chirp, inverse over j
    got = np.zeros((C, M), complex)
    for a in range(Fa // M):
        ka = a * M + np.arange(M)
        v = A[a] * np.exp(-2j * np.pi * np.outer(np.arange(Fb), ka) / L)      # [nb][j]
        S = np.fft.fft(v, axis=0)                                                # [kb][j]: bin k = ka + Fa*kb
        for kb in range(Fb):
            c = a + (Fa // M) * kb
            assert np.allclose(S[kb], np.fft.fft(x)[c * M:(c + 1) * M])
            got[c] = np.fft.ifft(S[kb] * H[c * M:(c + 1) * M]) * M
    assert np.allclose(got, want)


@pytest.mark.parametrize("R,logLs,logM", [(3, 7, 3), (5, 6, 2), (3, 9, 4)])
def test_sub_sequence_decomposition_equals_direct(R, logLs, logM):
    """nchan_subband = 3 * 2^k / 5 * 2^k (csrc/filterbank.hip k_sub_split / k_sub_combine): L = R * L' points as R interleaved
    sub-sequences, X[k + q L'] = sum_c W_R^(c q) W_L^(c k) F_c[k]; band q of the combined spectrum = rows q*Rr' .. of the full one."""
    Ls, M = 1 << logLs, 1 << logM
    L = R * Ls
    rng = np.random.default_rng(2)
    w = rng.standard_normal(L) + 1j * rng.standard_normal(L)
    F = [np.fft.fft(w[c::R]) for c in range(R)]
    k = np.arange(Ls)
    X = np.zeros(L, complex)
    for q in range(R):
        X[q * Ls:(q + 1) * Ls] = sum(np.exp(-2j * np.pi * c * q / R) * np.exp(-2j * np.pi * c * k / L) * F[c] for c in range(R))
    assert np.allclose(X, np.fft.fft(w))
    # spectrum rows of M bins: row of bin k + q L' is q * (L'/M) + k // M
    rows = (np.arange(L) // M).reshape(R, Ls // M, M)
    assert np.array_equal(rows[:, :, 0], np.arange(R)[:, None] * (Ls // M) + np.arange(Ls // M)[None, :])


@pytest.mark.parametrize("R,C,logMi,real", [(3, 4, 4, True), (5, 2, 3, True), (3, 8, 3, False), (9, 2, 2, True), (15, 1, 3, False)])
def test_pseudo_channel_decomposition_equals_direct(R, C, logMi, real):
    """freq_res = R * 2^k (csrc/filterbank.hip, k_sub_combine<R, true> + k_inv_chan + k_time_combine<R>): bin R m' + r of channel c
    is bin m' of pseudo-channel row c*R + r; for real input (two polarisations packed as one complex sequence of L = 2N points)
    the mirror bin L - k goes where the inverse pass looks for it -- row Rr-1-s, bin M'-m' (m' >= 1), row Rr-s, bin 0 (m' = 0);
    whole M'-point backward transforms per row, then y[n] = sum_r exp(+2 pi i r n / M) y_r[n mod M'].  Equals the direct
    formulation of Filterbank.C:561-662 with freq_res = M = R * M'."""
    Mi = 1 << logMi
    M = R * Mi
    N = C * M
    rng = np.random.default_rng(3)
    H = np.exp(1j * rng.uniform(-np.pi, np.pi, N))
    if real:
        p0, p1 = rng.standard_normal(2 * N), rng.standard_normal(2 * N)
        X0, X1 = np.fft.rfft(p0)[:N] * H, np.fft.rfft(p1)[:N] * H
        L, Rr = 2 * N, 2 * C * R
        Z = np.fft.fft(p0 + 1j * p1)
    else:
        z0 = rng.standard_normal(N) + 1j * rng.standard_normal(N)
        X0 = np.fft.fft(z0) * H
        X1 = X0
        L, Rr = N, C * R
        Z = np.fft.fft(z0)
    want = [np.stack([np.fft.ifft(X[c * M:(c + 1) * M]) * M for c in range(C)]) for X in (X0, X1)]
    # the permuted spectrum rows[row][bin] and chirp
    rows = np.zeros((Rr, Mi), complex)
    Hp = np.zeros((C * R, Mi), complex)
    for kk in range(L):
        up = real and kk > N
        if real and kk == N:
            continue
        kq = L - kk if up else kk
        c, mm = divmod(kq, M)
        mi, r = divmod(mm, R)
        s = c * R + r
        if not up:
            rows[s, mi] = Z[kk]
            Hp[s, mi] = H[kq]
        elif mi:
            rows[Rr - 1 - s, Mi - mi] = Z[kk]
        else:
            rows[Rr - s, 0] = Z[kk]
    # the inverse pass on pseudo-channel row s: Hermitian split exactly as k_inv_chan reads its mirror, chirp, backward transform
    got0, got1 = np.zeros((C, M), complex), np.zeros((C, M), complex)
    n = np.arange(M)
    for c in range(C):
        for r in range(R):
            s = c * R + r
            a = rows[s]
            if real:
                b = np.empty(Mi, complex)
                b[1:] = rows[Rr - 1 - s, Mi - np.arange(1, Mi)]
                b[0] = rows[Rr - s, 0] if s else rows[0, 0]
                x0, x1 = 0.5 * (a + np.conj(b)), -0.5j * (a - np.conj(b))
            else:
                x0 = x1 = a
            y0, y1 = np.fft.ifft(x0 * Hp[s]) * Mi, np.fft.ifft(x1 * Hp[s]) * Mi
            w = np.exp(2j * np.pi * r * n / M)
            got0[c] += w * y0[n % Mi]
            got1[c] += w * y1[n % Mi]
    assert np.allclose(got0, want[0]) and np.allclose(got1, want[1])
