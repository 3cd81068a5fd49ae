"""Pins the fp64 CPU oracle (oracle/glf_oracle.c) against golden vectors
generated from the reference's Python proof of concept (tools/gen_golden.py).

The reference ships no tests; these goldens are outputs of the
reference's own functions run in the build container.
"""
import numpy as np
import pytest

import oracle as orc


def test_sampling_matches_reference_grids(golden):
    g = golden("sampling.npz")
    tags = sorted({k.rsplit("_", 1)[0] for k in g.files})
    assert len(tags) >= 8
    for tag in tags:
        M, N, p_req = (int(x) for x in g[tag + "_shape"])
        idx = orc.sampling(N, M, p_req)  # C signature is (width, height)
        if tag + "_idx" in g.files:
            np.testing.assert_array_equal(idx, g[tag + "_idx"])
        else:
            n, first, second, last, checksum = (int(x) for x in g[tag + "_summary"])
            assert idx.size == n
            assert (int(idx[0]), int(idx[1]), int(idx[-1])) == (first, second, last)
            assert int(idx.astype(np.uint64).sum() % (1 << 62)) == checksum
        assert np.all(np.diff(idx.astype(np.int64)) > 0)  # "Must be sorted ASC"


def test_baseline_config_sample_counts():
    # SURVEY appendix B (C sampler semantics, hpc/sampling.c:8-13)
    assert orc.sampling(450, 300, 50).size == 54
    assert orc.sampling(512, 512, int(512 * 512 * 0.01)).size == 2601
    assert orc.sampling(1024, 1024, int(1024 * 1024 * 0.005)).size == 5329
    assert orc.sampling(2048, 2048, int(2048 * 2048 * 0.005)).size == 21316
    assert orc.sampling(4096, 4096, int(4096 * 4096 * 0.005)).size == 85264


def test_sampling_rejects_degenerate():
    with pytest.raises(ValueError):
        orc.sampling(4, 4, 1000)  # more samples than pixels -> dist 0
    with pytest.raises(ValueError):
        orc.sampling(0, 10, 5)


def test_syn32_every_stage(golden):
    g = golden("syn32.npz")
    img, idx = g["img"], g["idx"]
    np.testing.assert_array_equal(orc.sampling(32, 32, 10), idx)
    KA, KB = orc.affinity(img, idx)
    np.testing.assert_allclose(KA, g["K_A"], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(KB, g["K_B"], rtol=1e-13, atol=1e-300)
    D = orc.degree(img, idx)
    np.testing.assert_allclose(D, g["D_A"], rtol=1e-13)
    LA, alpha = orc.laplacian(KA, D)
    assert alpha == pytest.approx(float(g["alpha"]), rel=1e-14)
    np.testing.assert_allclose(LA, g["L_A"], rtol=1e-12, atol=1e-16)
    # pixel-row shards of the degree add up (multi-GPU sharding)
    Dsh = orc.degree(img, idx, row0=0, row1=13) + orc.degree(img, idx, row0=13, row1=32)
    np.testing.assert_allclose(Dsh, D, rtol=1e-13)


@pytest.mark.parametrize("name,kernel", [("photometric", orc.PHOTOMETRIC), ("spatial", orc.SPATIAL)])
def test_syn32_other_kernels(golden, name, kernel):
    # python/affinity_methods/{photometric,spatial}.py use h = 10 (not 30/40)
    g = golden("syn32.npz")
    prm = orc.default_params(kernel)
    prm.h_loc = 10.0
    prm.h_val = 10.0
    KA, KB = orc.affinity(g["img"], g["idx"], prm)
    np.testing.assert_allclose(KA, g["K_A_" + name], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(KB, g["K_B_" + name], rtol=1e-13, atol=1e-300)


def _tight_eig(LA, m, seed=1):
    X0 = orc.random_vectors(LA.shape[0], m, seed)
    return orc.inverse_power_iteration(LA, m, X0, epsilon=1e-9, inner_rtol=1e-12, max_outer=20000)


def _lapack_pairs(LA, m):
    w, V = np.linalg.eigh(LA)
    return np.ascontiguousarray(V[:, :m].T), w[:m]


def test_syn32_nystroem_permutation_filter_exact(golden):
    """Nystroem + Permutation + C filter pinned with LAPACK eigenpairs fed to the
    oracle's stages: the PoC-derived golden must be reproduced to rounding."""
    g = golden("syn32.npz")
    img, idx, alpha = g["img"], g["idx"], float(g["alpha"])
    for m, key in ((4, "z_c_m4"), (8, "z_c_m8")):
        vecs, vals = _lapack_pairs(g["L_A"], m)
        np.testing.assert_allclose(vals, g["mu"][:m], rtol=1e-12)
        phi = orc.permutation(orc.nystroem(img, idx, alpha, vecs, vals), idx)
        np.testing.assert_allclose(np.abs(phi.T), g["phi_perm_abs"][:, :m], rtol=1e-8, atol=1e-12)
        zf, out = orc.result_from_laplacian(img, phi, vals, gain=3.0)
        np.testing.assert_allclose(zf, g[key], rtol=1e-11, atol=1e-10)
        np.testing.assert_array_equal(out, np.clip(g[key], 0, 255).astype(np.uint8))


def test_syn32_iterative_eigensolver(golden):
    """hpc/inverse_power_it.c semantics: the stopping rule measures the SUBSPACE
    residual; 1/norm eigenvalue estimates (:204) and the normalised pre-GS
    iterates (:230) converge at the rate of the gaps inside the block, so close
    pairs (mu_3/mu_4 = 0.991 here) are only approximately resolved."""
    g = golden("syn32.npz")
    LA, mu = g["L_A"], g["mu"]
    for m in (4, 8):
        vecs, vals, st = _tight_eig(LA, m)
        assert st["residual"] <= 1e-9 and st["outer_its"] > 10
        np.testing.assert_allclose(vals, mu[:m], atol=1e-4)
        np.testing.assert_allclose(vals[:2], mu[:2], rtol=1e-10)
        # invariant subspace agrees with LAPACK's
        V, _ = _lapack_pairs(LA, m)
        Q, _ = orc.orthonormalise(vecs)
        np.testing.assert_allclose(Q.T.dot(Q), V.T.dot(V), atol=1e-6)
        # well separated vectors agree up to sign
        np.testing.assert_allclose(np.abs(vecs[:2]), np.abs(V[:2]), atol=1e-8)
    # loose epsilon (the reference default 0.1): few iterations, residual reported
    X0 = orc.random_vectors(9, 4, 1)
    _, vals, st = orc.inverse_power_iteration(LA, 4, X0, epsilon=0.1)
    assert st["outer_its"] == 7 and st["residual"] <= 0.1
    # opti_gs = 3: GS only every third iteration + trailing GS (:174-186)
    # the norms then span k = 1..3 un-normalised solves, so 1/norm estimates mu^k
    # (a property of the reference, reproduced, not fixed)
    _, vals3, st3 = orc.inverse_power_iteration(LA, 4, X0, opti_gs=3, epsilon=1e-6, inner_rtol=1e-12)
    k = st3["outer_its"] % 3 or 3
    assert vals3[0] == pytest.approx(mu[0] ** k, rel=1e-6)


def test_permutation_literal_equals_fast(golden):
    g = golden("syn32.npz")
    idx = g["idx"]
    rng = np.random.RandomState(0)
    phi_sf = rng.standard_normal((3, 32 * 32))
    a = orc.permutation(phi_sf, idx, literal=True)
    b = orc.permutation(phi_sf, idx, literal=False)
    np.testing.assert_array_equal(a, b)
    # sample rows land on their pixel (hpc/utils.c:149-152)
    np.testing.assert_array_equal(a[:, idx], phi_sf[:, :idx.size])


def test_test_png_stages(golden, png):
    g = golden("test_png.npz")
    img = png("test.png")
    assert img.shape == (100, 100)
    idx = orc.sampling(100, 100, int(100 * 100 * 0.01))
    np.testing.assert_array_equal(idx, g["idx"])
    KA, KB = orc.affinity(img, idx)
    np.testing.assert_allclose(KA, g["K_A"], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(KB[:, ::97], g["K_B_cols97"], rtol=1e-13, atol=1e-300)
    assert np.linalg.norm(KB) == pytest.approx(float(g["K_B_fro"]), rel=1e-13)
    yR = np.delete(img.reshape(-1).astype(np.float64), idx)
    np.testing.assert_allclose(KB.dot(yR), g["K_B_y"], rtol=1e-12)
    D = orc.degree(img, idx)
    np.testing.assert_allclose(D, g["D_A"], rtol=1e-13)
    LA, alpha = orc.laplacian(KA, D)
    assert alpha == pytest.approx(float(g["alpha"]), rel=1e-14)
    np.testing.assert_allclose(np.linalg.eigvalsh(LA), g["mu"], rtol=1e-10)
    # C filter with the 16 / 99 smallest LAPACK eigenpairs (golden stored as f32)
    for m, key in ((16, "z_c_m16"), (99, "z_c_m99")):
        vecs, vals = _lapack_pairs(LA, m)
        phi = orc.permutation(orc.nystroem(img, idx, alpha, vecs, vals), idx)
        zf, _ = orc.result_from_laplacian(img, phi, vals, gain=3.0)
        np.testing.assert_allclose(zf, g[key], rtol=1e-6, atol=1e-4)
    # iterative solver: mu_16 / mu_17 = 0.97 -> slow but convergent
    vecs, vals, st = _tight_eig(LA, 8)
    Q, _ = orc.orthonormalise(vecs)
    np.testing.assert_allclose(np.linalg.eigvalsh(Q.dot(LA).dot(Q.T)), g["mu"][:8], rtol=1e-8)
    np.testing.assert_allclose(np.sort(vals), g["mu"][:8], atol=5e-3)  # in-cluster estimates lag


def test_cat50_config1(golden, png):
    g = golden("cat50.npz")
    img = png("cat_small.png")
    assert img.shape == (300, 450)
    idx = orc.sampling(450, 300, 50)
    np.testing.assert_array_equal(idx, g["idx"])
    KA, _ = orc.affinity(img, idx, want_KB=False)
    np.testing.assert_allclose(KA, g["K_A"], rtol=1e-13, atol=1e-300)
    D = orc.degree(img, idx)
    np.testing.assert_allclose(D, g["D_A"], rtol=1e-13)
    LA, alpha = orc.laplacian(KA, D)
    assert alpha == pytest.approx(float(g["alpha"]), rel=1e-14)
    # m = p - 1 = 53, the reference default (hpc/image_processing.c:96-108)
    for m, key in ((53, "z_c_m53"), (16, "z_c_m16")):
        vecs, vals = _lapack_pairs(LA, m)
        phi = orc.permutation(orc.nystroem(img, idx, alpha, vecs, vals), idx)
        zf, _ = orc.result_from_laplacian(img, phi, vals, gain=3.0)
        np.testing.assert_allclose(zf, g[key], rtol=1e-6, atol=1e-4)


def test_whole_path_matches_stagewise(golden, png):
    img = png("test.png")
    zf, out, info = orc.image_processing(img, 100, 8, epsilon=1e-9, inner_rtol=1e-12, seed=1)
    g = golden("test_png.npz")
    assert info["p"] == 100 and info["m"] == 8
    np.testing.assert_allclose(np.sort(info["eigvals"]), g["mu"][:8], atol=5e-3)
    idx = orc.sampling(100, 100, 100)
    KA, _ = orc.affinity(img, idx, want_KB=False)
    LA, alpha = orc.laplacian(KA, orc.degree(img, idx))
    assert info["alpha"] == alpha
    vecs, vals, _ = _tight_eig(LA, 8)
    phi = orc.permutation(orc.nystroem(img, idx, alpha, vecs, vals), idx)
    zf2, out2 = orc.result_from_laplacian(img, phi, vals, gain=3.0)
    np.testing.assert_array_equal(zf, zf2)
    np.testing.assert_array_equal(out, out2)
    # num_eigvals >= p falls back to p - 1
    _, _, info2 = orc.image_processing(img, 100, 100, epsilon=0.1)
    assert info2["m"] == 99


def test_default_epsilon_runs_and_is_loose(png):
    # epsilon = 0.1 absolute Frobenius (hpc/image_processing.c:151) converges in a
    # handful of outer iterations; eigenvalues are then only approximately right.
    img = png("test.png")
    _, out, info = orc.image_processing(img, 100, 16, epsilon=0.1)
    assert 1 <= info["outer_its"] < 200
    assert info["residual"] <= 0.1
    assert out.dtype == np.uint8 and out.shape == img.shape


def test_gram_schmidt_properties():
    X = orc.random_vectors(200, 12, 7)
    Q, norms = orc.orthonormalise(X)
    np.testing.assert_allclose(Q.dot(Q.T), np.eye(12), atol=1e-12)
    assert norms[0] == pytest.approx(np.linalg.norm(X[0]), rel=1e-14)
    # span preserved: projector equality
    Qr, _ = np.linalg.qr(X.T)
    np.testing.assert_allclose(Q.T.dot(Q), Qr.dot(Qr.T), atol=1e-10)


def test_block_pcg_solves():
    rng = np.random.RandomState(3)
    M = rng.standard_normal((60, 60))
    A = M.dot(M.T) / 60 + 2 * np.eye(60)
    B = rng.standard_normal((5, 60))
    X, its = orc.block_pcg(A, B, rtol=1e-10)
    np.testing.assert_allclose(X.dot(A), B, rtol=0, atol=1e-8)
    assert 1 <= its <= 200


def test_entire_computation_small():
    # -no_approx formula z = clamp(y - L y) (hpc/display.c:128-149) against numpy
    rng = np.random.RandomState(5)
    img = rng.randint(0, 256, (12, 9)).astype(np.uint8)
    zf, out = orc.entire_computation(img)
    r, c = np.divmod(np.arange(img.size), 9)
    y = img.reshape(-1).astype(np.float64)
    K = np.exp(-((r[:, None] - r[None]) ** 2 + (c[:, None] - c[None]) ** 2) / 1600.0) * \
        np.exp(-((y[:, None] - y[None]) ** 2) / 900.0)
    D = K.sum(1)
    L = (np.diag(D) - K) / D.mean()
    z = y - L.dot(y)
    np.testing.assert_allclose(zf.reshape(-1), z, rtol=1e-12)
    np.testing.assert_array_equal(out.reshape(-1), np.clip(z, 0, 255).astype(np.uint8))


def test_barbara_config2(golden, png):
    """BASELINE config 2 (512x512, 1 %): degree, alpha, spectrum and the C filter
    with the 64 smallest LAPACK pairs against the PoC-derived golden."""
    g = golden("barbara.npz")
    img = png("barbara.png")
    assert img.shape == (512, 512)
    idx = orc.sampling(512, 512, int(512 * 512 * 0.01))
    assert idx.size == int(g["p"]) == 2601
    KA, _ = orc.affinity(img, idx, want_KB=False)
    D = orc.degree(img, idx)
    np.testing.assert_allclose(D, g["D_A"], rtol=1e-12)
    LA, alpha = orc.laplacian(KA, D)
    assert alpha == pytest.approx(float(g["alpha"]), rel=1e-13)
    w, V = np.linalg.eigh(LA)
    np.testing.assert_allclose(w[:64], g["mu64"], rtol=1e-9)
    assert w[-1] == pytest.approx(float(g["mu_max"]), rel=1e-9)
    vecs = np.ascontiguousarray(V[:, :64].T)
    phi = orc.permutation(orc.nystroem(img, idx, alpha, vecs, w[:64]), idx)
    zf, out = orc.result_from_laplacian(img, phi, w[:64], gain=3.0)
    np.testing.assert_allclose(zf[::64], g["z_c_m64_rows"], rtol=1e-6, atol=2e-4)
    assert np.mean(out != g["z_c_m64_u8"]) < 1e-3


def test_bounded_sample_helpers_agree_with_full_stages(golden):
    """The slices bench.py times for cpu_baseline are the same arithmetic as the full stages."""
    g = golden("syn32.npz")
    img, idx, alpha = g["img"], g["idx"], float(g["alpha"])
    np.testing.assert_allclose(orc.laplacian_rows(img, idx, g["D_A"], alpha, 2, 7), g["L_A"][2:7], rtol=1e-12, atol=1e-16)
    X = orc.random_vectors(9, 3, 2)
    np.testing.assert_allclose(orc.matvec_rows(g["L_A"][2:7], X), X.dot(g["L_A"][2:7].T), rtol=1e-12)
    vecs, vals = _lapack_pairs(g["L_A"], 4)
    full = orc.permutation(orc.nystroem(img, idx, alpha, vecs, vals), idx).reshape(4, 32, 32)
    part = orc.nystroem_rows(img, idx, alpha, vecs, vals, 10, 14).reshape(4, 4, 32)
    mask = np.ones((32, 32), dtype=bool)
    mask.reshape(-1)[idx] = False
    np.testing.assert_allclose(part[:, mask[10:14]], full[:, 10:14][:, mask[10:14]], rtol=1e-12)


def test_nlm_kernel_against_the_poc(golden):
    """The non-local-means affinity (python/affinity_methods/NLM.py:9-34; f1 of SURVEY 8f): the oracle's restatement against
    the PoC's kernel rows (tools/gen_golden_nlm.py re-indexes the PoC's transposed column layout to raster order) on a square
    and on a non-square image -- symmetric padding, the sum-normalised 7 x 7 Gaussian mask on the patch VALUES, h = 3."""
    g = golden("nlm.npz")
    for tag in ("syn32", "rect"):
        img, idx, K = g[tag + "_img"], g[tag + "_idx"], g[tag + "_K"]
        h, w = img.shape
        prm = orc.default_params(orc.NLM)
        prm.h_val = 3.0
        KA, KB = orc.affinity(img, idx, prm=prm)
        rest = np.ones(h * w, dtype=bool)
        rest[idx] = False
        np.testing.assert_allclose(KA, K[:, idx], rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(KB, K[:, rest], rtol=1e-12, atol=1e-300)
        np.testing.assert_allclose(np.diag(KA), 1.0, rtol=0, atol=0)
        np.testing.assert_allclose(orc.degree(img, idx, prm=prm), K.sum(1), rtol=1e-13)
        D = K.sum(1)
        LA, alpha = orc.laplacian(KA, D)
        np.testing.assert_allclose(orc.laplacian_rows(img, idx, D, alpha, 2, 5, prm=prm), LA[2:5], rtol=1e-13, atol=1e-300)


def test_poc_alternative_filters_and_balancing(golden):
    """the numpy restatements of the PoC's sinkhorn / orthogonalisation / smoothing_matrix / smoothing /
    sharpening (python/image_processing.py:90-241) against the PoC's own outputs on the 32 x 32 image (tests/golden/f4.npz,
    tools/gen_golden_f4.py). Singular vectors are compared up to sign."""
    g = golden("f4.npz")
    phi, Pi = orc.poc_nystroem(g["K_A"], g["K_B"])
    np.testing.assert_allclose(Pi, g["Pi"], rtol=1e-10)
    np.testing.assert_allclose(np.abs(phi), np.abs(g["phi"]), rtol=0, atol=1e-9 * np.abs(g["phi"]).max())
    W_A, W_B = orc.poc_sinkhorn(g["phi"], g["Pi"])
    np.testing.assert_allclose(W_A, g["W_A"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(W_B, g["W_B"], rtol=1e-9, atol=1e-12)
    # doubly stochastic in the PoC's sense: rows of [W_A W_B] sum to 1 after 100 scalings
    np.testing.assert_allclose(np.concatenate((W_A, W_B), axis=1).sum(axis=1), 1.0, rtol=1e-6)
    V, P = orc.poc_orthogonalisation(g["W_A"], g["W_B"])
    np.testing.assert_allclose(P, g["Pi_orth"], rtol=1e-8)
    np.testing.assert_allclose(np.abs(V), np.abs(g["V_orth"]), rtol=0, atol=1e-7 * np.abs(g["V_orth"]).max())
    np.testing.assert_allclose(V.T @ V, np.identity(V.shape[1]), atol=1e-8)      # what the step is for
    Vs, Ls = orc.poc_smoothing_matrix(g["idx"], g["phi"], g["Pi"])
    np.testing.assert_allclose(Ls, g["L_smooth"], rtol=1e-9)
    np.testing.assert_allclose(np.abs(Vs), np.abs(g["V_smooth"]), rtol=0, atol=1e-8 * np.abs(g["V_smooth"]).max())
    y = g["img"].astype(np.float64)
    np.testing.assert_allclose(orc.poc_smoothing_filter(y, g["V_smooth"], g["L_smooth"]), g["z_smooth"], rtol=1e-10, atol=1e-9)
    np.testing.assert_allclose(orc.poc_sharpening_filter(y, g["V_smooth"], g["L_smooth"]), g["z_sharp"], rtol=1e-9, atol=1e-8)
    # the Gram form of the sharpening filter (what the HIP path evaluates) is the same filter
    V_, L_ = g["V_smooth"], g["L_smooth"]
    w = orc.sharpening_weights(V_.T @ V_, L_, V_.T @ y.reshape(-1))
    np.testing.assert_allclose((V_ @ w).reshape(y.shape), g["z_sharp"], rtol=1e-9, atol=1e-8)
