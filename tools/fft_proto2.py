"""numpy model of the front-end kernel's data flow (r-sequential passes, 16 lanes per frame): checks the index algebra of
frontend.hip against numpy's rfft.  Lane = m0 / n0 (16 per frame), registers = n1 / m1.  Development aid."""
import numpy as np
rng = np.random.default_rng(1)
x = rng.standard_normal(512)
win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(512) / 512)
ref = np.abs(np.fft.rfft(np.concatenate([x * win, np.zeros(1536)]))) ** 2       # |X[k]|^2, k = 0..1024
W = lambda N, e: np.exp(-2j * np.pi * (e % N) / N)
zw = (x * win)[0::2] + 1j * (x * win)[1::2]                                     # packed, windowed: 256 complex
Z = {}
for r in range(4):
    # pass A: lane n0 holds a[n1] = zw[16 n1 + n0] * W64^(n1 r); FFT16 over n1 -> m0
    a = np.array([[zw[16 * n1 + n0] * W(64, n1 * r) for n1 in range(16)] for n0 in range(16)])       # [n0][n1]
    A = np.fft.fft(a, axis=1)                                                                        # [n0][m0]
    # transpose: lane m0 gets B[n0]; twiddle W1024^(n0 (4 m0 + r))
    B = A.T.copy()                                                                                   # [m0][n0]
    for m0 in range(16):
        for n0 in range(16):
            B[m0, n0] *= W(1024, n0 * (4 * m0 + r))
    Zr = np.fft.fft(B, axis=1)                                                                       # [m0][m1] = Z_r[m0 + 16 m1]
    Z[r] = Zr
full = np.fft.fft(np.concatenate([zw, np.zeros(768)]))
for r in range(4):
    for m0 in range(16):
        for m1 in range(16):
            assert abs(Z[r][m0, m1] - full[4 * (m0 + 16 * m1) + r]) < 1e-9
P = np.full(1025, np.nan)
def bfly(zk, zz, k):
    zc = np.conj(zz); a = zk + zc; d = zk - zc
    rw = -1j * W(2048, k) * d
    return abs(a + rw) ** 2 / 4, abs(a - rw) ** 2 / 4           # |X[k]|^2, |X[1024 - k]|^2
# r = 0: every position its own bin; partner lane (16 - m0) % 16, register 15 - m1 (lane 0: (16 - m1) % 16)
for m0 in range(16):
    for m1 in range(12):
        pl, pr = ((16 - m0) % 16, 15 - m1) if m0 else (0, (16 - m1) % 16)
        k = 4 * (m0 + 16 * m1)
        P[k], _ = bfly(Z[0][m0, m1], Z[0][pl, pr], k)
# r = 2: pairs (m, 255 - m): positions m1 < 8 compute own + partner (lane 15 - m0, register 15 - m1)
for m0 in range(16):
    for m1 in range(8):
        k = 4 * (m0 + 16 * m1) + 2
        own, par = bfly(Z[2][m0, m1], Z[2][15 - m0, 15 - m1], k)
        P[k] = own
        if 15 - m1 < 12: P[1024 - k] = par
        assert 1024 - k == 4 * ((15 - m0) + 16 * (15 - m1)) + 2
# r = 1 with 3: position (m0, m1) of Z_1 with (15 - m0, 15 - m1) of Z_3 -> X[4m + 1] and X[4 (255 - m) + 3]
for m0 in range(16):
    for m1 in range(16):
        k = 4 * (m0 + 16 * m1) + 1
        own, par = bfly(Z[1][m0, m1], Z[3][15 - m0, 15 - m1], k)
        if m1 < 12: P[k] = own
        if 15 - m1 < 12: P[1024 - k] = par
        assert 1024 - k == 4 * ((15 - m0) + 16 * (15 - m1)) + 3
assert not np.isnan(P[:768]).any()
print("max rel err bins 0..767:", np.max(np.abs(P[:768] - ref[:768]) / (np.abs(ref[:768]) + 1e-12)))
