"""The algebra behind TP_TUNE_TRI_STATS (tokenpacker_amd/csrc/tp_pack_qr.hip), restated in numpy fp64 and checked against the
plain LayerNorm + Linear composition the reference computes (builder.py:112-113 followed by nn.MultiheadAttention's in-projection):
    LN(W2 h + b2) W'^T + b  ==  rstd (Wcc h + dcc) + b',     rstd = 1 / sqrt(||R h + c~||^2 / E + eps),
with W2c = W2 - column means, [W2c | b2c] = Q [R | c~], Wcc = (W' diag(gamma)) W2c, dcc = (W' diag(gamma)) b2c, b' = W' beta + b."""
import numpy as np


def test_centred_chain_and_triangular_second_moment():
    rng = np.random.default_rng(5)
    E, rows, eps = 96, 40, 1e-6
    w2 = rng.normal(size=(E, E)) * 0.1 + 0.3                 # a common offset: the row mean of H2 is far from zero
    b2 = rng.normal(size=E) * 0.1 + 2.0
    gamma, beta = 1.0 + 0.1 * rng.normal(size=E), 0.1 * rng.normal(size=E)
    w_in, b_in = rng.normal(size=(E, E)) * 0.1, rng.normal(size=E) * 0.1
    h = rng.normal(size=(rows, E))
    # the reference's composition
    h2 = h @ w2.T + b2
    mu, var = h2.mean(axis=1, keepdims=True), h2.var(axis=1, keepdims=True)
    want = ((h2 - mu) / np.sqrt(var + eps) * gamma + beta) @ w_in.T + b_in
    # pack time
    w2c, b2c = w2 - w2.mean(axis=0, keepdims=True), b2 - b2.mean()
    q, r_aug = np.linalg.qr(np.concatenate([w2c, b2c[:, None]], axis=1), mode="complete")
    r, c_til = r_aug[:, :E], r_aug[:, E]
    assert np.abs(np.tril(r, -1)).max() == 0.0               # upper triangular: output n needs the inputs k >= n only
    wf = w_in * gamma                                         # the LayerNorm-folded in-projection weight W'
    wcc, dcc, b_f = wf @ w2c, wf @ b2c, w_in @ beta + b_in
    # forward
    y = h @ r.T + c_til
    second_moment = (y * y).sum(axis=1, keepdims=True) / E
    np.testing.assert_allclose(second_moment, var, rtol=1e-10)
    got = (h @ wcc.T + dcc) / np.sqrt(second_moment + eps) + b_f
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-11)
    # the tiles the statistics GEMM skips: with 4 column tiles, tile t multiplies only the K range [t E/4, E)
    tiles = 4
    y_tiled = np.concatenate([h[:, t * E // tiles:] @ r[t * E // tiles:(t + 1) * E // tiles, t * E // tiles:].T for t in range(tiles)], axis=1)
    np.testing.assert_allclose(y_tiled + c_til, y, rtol=0, atol=1e-12)
