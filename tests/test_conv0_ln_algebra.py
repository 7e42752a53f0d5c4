"""The algebra behind the matrix-core conv0 + LayerNorm + GELU kernels (unispeech_amd/csrc/conv0_bwd_mfma.hip; the block is
WavLM/WavLM.py:403-418 with extractor_mode "layer_norm"), checked in float64 on the CPU against the straightforward formulas:
  * a frame's LayerNorm statistics as a quadratic form of its ten samples (centred second moments of the parameters);
  * the weight / bias gradient without a second pass over the channels: sum_t dz rstd_t x  minus terms that only need weighted
    moments of the waveform.
No GPU, no library: this pins the derivation the kernels implement (their numerics are tested in tests/gpu_checks.py)."""
import numpy as np


def _setup(seed=0, F=37, C=24, K=10):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((F, K))                      # im2col rows of the waveform
    W = 0.4 * rng.standard_normal((C, K))
    cb = 0.3 + 0.2 * rng.standard_normal(C)
    gm = 1 + 0.1 * rng.standard_normal(C)
    bt = 0.1 * rng.standard_normal(C)
    dz = rng.standard_normal((F, C))                     # g * gelu'(z): whatever arrives at the LayerNorm output
    return x, W, cb, gm, bt, dz


def test_frame_statistics_are_a_quadratic_form_of_the_samples():
    x, W, cb, _, _, _ = _setup()
    y = x @ W.T + cb
    mean, var = y.mean(1), y.var(1)
    wbar, cbar = W.mean(0), cb.mean()
    Wc, cc = W - wbar, cb - cbar
    G, u, s = Wc.T @ Wc / len(cb), cc @ Wc / len(cb), (cc ** 2).mean()
    np.testing.assert_allclose(x @ wbar + cbar, mean, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(np.einsum("fj,jk,fk->f", x, G, x) + 2 * x @ u + s, var, rtol=1e-11, atol=1e-12)
    assert np.all(np.linalg.eigvalsh(G) > -1e-12)        # a positive form: no cancellation in the variance


def test_weight_gradient_without_a_second_channel_pass():
    x, W, cb, gm, _, dz = _setup(1)
    F, C = dz.shape
    eps = 1e-5
    y = x @ W.T + cb
    mean, rstd = y.mean(1, keepdims=True), 1.0 / np.sqrt(y.var(1, keepdims=True) + eps)
    xh = (y - mean) * rstd
    h = dz * gm
    s1, s2 = h.mean(1, keepdims=True), (h * xh).mean(1, keepdims=True)
    dconv = rstd * (h - s1 - xh * s2)                    # LayerNorm backward, per frame
    dW_ref, dcb_ref = dconv.T @ x, dconv.sum(0)
    dgamma_ref, dbeta_ref = (dz * xh).sum(0), dz.sum(0)
    # the kernels' form: per-channel sums of dz against the image (rstd x | rstd | 1) ...
    T = dz.T @ (rstd * x)                                # [C, K]
    TR = dz.T @ rstd[:, 0]                               # [C]
    # ... and 16 x 16 weighted moments of X'' = (x | 1 | mean) with a_t = rstd^2 s2, b_t = rstd s1
    a, b = (rstd ** 2 * s2)[:, 0], (rstd * s1)[:, 0]
    X2 = np.concatenate([x, np.ones((F, 1)), mean], axis=1)
    D = X2.T @ (a[:, None] * X2)                         # rows / columns 0..9 taps, 10 ones, 11 mean
    v = b @ X2                                           # row 13 of the kernel's matrix
    K = x.shape[1]
    t3 = W @ D[:K, :K] + np.outer(cb, D[K, :K]) - D[K + 1, :K]
    dW = gm[:, None] * T - v[:K] - t3
    t3b = W @ D[:K, K] + cb * D[K, K] - D[K + 1, K]
    dcb = gm * TR - v[K] - t3b
    np.testing.assert_allclose(dW, dW_ref, rtol=1e-9, atol=1e-10)
    np.testing.assert_allclose(dcb, dcb_ref, rtol=1e-9, atol=1e-10)
    # dgamma / dbeta are plain sums over frames (lane-local / the ones column)
    np.testing.assert_allclose((dz * xh).sum(0), dgamma_ref)
    np.testing.assert_allclose(dz.T @ np.ones(F), dbeta_ref)
