"""Parity of the HIP path (through the C-ABI, libglf.so) against the fp64 CPU
oracle on the same inputs, stage by stage and end to end, plus the committed
golden vectors of the reference's Python PoC. Needs an MI355X: -m gpu.

Tolerances (fp32 device arithmetic vs fp64 oracle; north_star asks for a stated
fp64 -> fp32 PSNR tolerance):
  kernel entries      rel 3e-6 for entries >= 1e-6, rel 5e-5 below (the f32 rounding of the
                      exponent t ~ 90 is amplified by exp: entries ~1e-27)
  degree D_A, alpha   rel 2e-6   (f32 within an image row, f64 across)
  L_A                 abs 3e-6 * max|L_A|
  eigenvalues         abs 2e-4   (same X0, same stopping rule)
  Phi                 abs PHI_TOL = 2e-5 * max|Phi| (an f32 contraction over p terms lands near 1e-6)
  z (float)           abs 2e-2 grey levels; u8 output: PSNR >= 50 dB and
                      |delta| <= 1 grey level on >= 99 % of pixels
  z - y (correction)  rel-L2 <= CORR_TOL against the oracle run with the SAME number of outer iterations
                      (never skipped: when the two stopping rules trip one iteration apart the oracle is re-run
                      pinned to the GPU's count, oracle/parity.py)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import glf  # noqa: E402  (fails loudly when libglf.so is missing)
import oracle as orc  # noqa: E402
import parity  # noqa: E402  (oracle/parity.py: oracle runs pinned to the GPU's outer-iteration count)
from conftest import psnr  # noqa: E402


PHI_TOL = 2e-5
CORR_TOL = 2e-3   # || z_gpu - z_ref || / || z_ref - y ||: end to end through the iterative eigen-solve (inner solves to rtol 1e-5)


def assert_kernel_close(got, ref):
    big = ref >= 1e-6
    np.testing.assert_allclose(got[big], ref[big], rtol=3e-6)
    np.testing.assert_allclose(got, ref, rtol=5e-5, atol=1e-37)


@pytest.fixture(scope="module", params=["f16s", "f32"])
def ctx(request):
    """Every test runs with both evaluations of the Nystroem contraction (include/glf.h):
    split-f16 operands on the f16 matrix pipe (default) and exact f32-input MFMA."""
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    c = glf.Context(0)
    c.set_contraction(glf.CONTRACT_F16_SPLIT if request.param == "f16s" else glf.CONTRACT_F32_MFMA)
    yield c
    c.close()


@pytest.fixture(autouse=True)
def _default_kernel_selection(request):
    """Tests force kernel families through glf_ctx_set_tuning on the module-scoped context; every test starts and ends with
    the default (size-based) selection."""
    yield
    if "ctx" in request.fixturenames:
        request.getfixturevalue("ctx").reset_tuning()


PATH_ID = {"grid": 1, "rank": 3, "band": 4}   # glf_stats.nystroem_path / matvec_path


def _set_paths(ctx, paths):
    """Kernel family of the three K_B / K_A stages: "direct" (entry by entry over ALL samples; stored L_A), "grid" (grid-factored, all
    256 grey levels), "rank" (grid-factored with the photometric table as a rank-R expansion) or "band" (entry by entry over the
    samples within the kernel's radius only: the default at benchmark sizes)."""
    ctx.set_tuning(NYS_PATH=paths, DEG_PATH="direct" if paths == "direct" else "grid",
                   MV_PATH={"direct": "dense", "grid": "grid", "rank": "rank", "band": "band"}[paths])


def _lapack_pairs(LA, m):
    w, V = np.linalg.eigh(LA)
    return np.ascontiguousarray(V[:, :m].T), w[:m]


def _images(golden, png):
    return {
        "syn32": (golden("syn32.npz")["img"], 10),
        "test": (png("test.png"), 100),
        "ragged": (glf.synth_image(53, 37, seed=3), 20),
        "cat50": (png("cat_small.png"), 50),
    }


@pytest.mark.parametrize("name", ["syn32", "test", "ragged", "cat50"])
def test_affinity_degree_laplacian(ctx, golden, png, name):
    img, p_req = _images(golden, png)[name]
    h, w = img.shape
    idx = glf.Sampling(w, h, p_req)
    d_img = ctx.to_device(img)
    K_A, K_B = ctx.ComputeAffinityMatrices(d_img, idx)
    KA_ref, _ = orc.affinity(img, idx, want_KB=False)
    assert_kernel_close(ctx.mat_to_numpy(K_A), KA_ref)
    D_ref = orc.degree(img, idx)
    np.testing.assert_allclose(ctx.degree_of(K_B), D_ref, rtol=2e-6)
    assert (K_B.kind, K_B.rows, K_B.cols, K_B.scale) == (glf.MAT_KERNEL_B, idx.size, h * w - idx.size, 1.0)
    L_A, L_B, alpha = ctx.ComputeLaplacianMatrix(K_A, K_B)
    LA_ref, alpha_ref = orc.laplacian(KA_ref, D_ref)
    assert alpha == pytest.approx(alpha_ref, rel=2e-6)
    np.testing.assert_allclose(ctx.mat_to_numpy(L_A), LA_ref, rtol=0, atol=3e-6 * np.abs(LA_ref).max())
    assert L_B.scale == pytest.approx(-alpha_ref, rel=1e-5)
    # same L_A when K_A is not materialised
    L_A2, _, _ = ctx.ComputeLaplacianMatrix(None, K_B)
    np.testing.assert_array_equal(ctx.mat_to_numpy(L_A2), ctx.mat_to_numpy(L_A))
    ctx.destroy(K_A, L_A, L_A2, K_B)


@pytest.mark.parametrize("paths", ["direct", "grid"])
@pytest.mark.parametrize("kernel,ok", [(glf.KERNEL_PHOTOMETRIC, orc.PHOTOMETRIC), (glf.KERNEL_SPATIAL, orc.SPATIAL)])
def test_other_kernels_against_golden(ctx, golden, kernel, ok, paths, monkeypatch):
    ctx.set_tuning(DEG_PATH=paths)   # the degree in both forms (one factor of the kernel is constant here)
    g = golden("syn32.npz")
    name = "photometric" if kernel == glf.KERNEL_PHOTOMETRIC else "spatial"
    d_img = ctx.to_device(g["img"])
    K_A, K_B = ctx.ComputeAffinityMatrices(d_img, g["idx"], kernel=kernel, h_loc=10.0, h_val=10.0)
    assert_kernel_close(ctx.mat_to_numpy(K_A), g["K_A_" + name])
    D_ref = g["K_A_" + name].sum(1) + g["K_B_" + name].sum(1)
    np.testing.assert_allclose(ctx.degree_of(K_B), D_ref, rtol=3e-6)
    ctx.destroy(K_A, K_B)


def test_stage_goldens_from_python_poc(ctx, golden, png):
    """K_A, D_A, alpha of the reference's PoC (tests/golden) reproduced by the HIP path."""
    for npz, img, p_req in (("syn32.npz", golden("syn32.npz")["img"], 10), ("test_png.npz", png("test.png"), 100),
                            ("cat50.npz", png("cat_small.png"), 50)):
        g = golden(npz)
        h, w = img.shape
        idx = glf.Sampling(w, h, p_req)
        np.testing.assert_array_equal(idx, g["idx"])
        K_A, K_B = ctx.ComputeAffinityMatrices(ctx.to_device(img), idx)
        assert_kernel_close(ctx.mat_to_numpy(K_A), g["K_A"])
        np.testing.assert_allclose(ctx.degree_of(K_B), g["D_A"], rtol=2e-6)
        L_A, L_B, alpha = ctx.ComputeLaplacianMatrix(K_A, K_B)
        assert alpha == pytest.approx(float(g["alpha"]), rel=2e-6)
        ctx.destroy(K_A, L_A, K_B)


@pytest.mark.parametrize("n,m", [(200, 12), (1000, 33), (2601, 64), (77, 5), (5000, 100)])
def test_orthonormalise_and_normalise(ctx, n, m):
    X = orc.random_vectors(n, m, 7)          # (m, n): one row per vector
    Q_ref, norms_ref = orc.orthonormalise(X)
    Xd = ctx.dense_from_numpy(X.T)           # device layout: n x m
    norms = ctx.OrthonormaliseVecs(Xd)
    Q = ctx.mat_to_numpy(Xd).T
    np.testing.assert_allclose(norms, norms_ref, rtol=2e-5)
    np.testing.assert_allclose(Q, Q_ref, rtol=0, atol=2e-4 * np.abs(Q_ref).max() * np.sqrt(m))
    np.testing.assert_allclose(Q.dot(Q.T), np.eye(m), atol=5e-5 * m)
    Y = ctx.dense_from_numpy((X * 3.0).T)
    nn = ctx.NormaliseVecs(Y)
    np.testing.assert_allclose(nn, 3.0 * np.linalg.norm(X, axis=1), rtol=1e-6)
    np.testing.assert_allclose(np.linalg.norm(ctx.mat_to_numpy(Y).astype(np.float64), axis=0), 1.0, rtol=2e-6)
    ctx.destroy(Xd, Y)


@pytest.mark.parametrize("name,m,eps", [("syn32", 4, 1e-3), ("test", 16, 1e-2), ("test", 16, 0.1), ("cat50", 53, 0.1),
                                         ("test", 99, 0.1)])
def test_inverse_power_iteration_matches_oracle(ctx, golden, png, name, m, eps):
    img, p_req = _images(golden, png)[name]
    h, w = img.shape
    idx = glf.Sampling(w, h, p_req)
    KA, _ = orc.affinity(img, idx, want_KB=False)
    LA, alpha = orc.laplacian(KA, orc.degree(img, idx))
    p = idx.size
    X0 = glf.random_vectors(p, m, 1)
    A = ctx.dense_from_numpy(LA, ld=(p + 63) // 64 * 64)
    vecs, vals, st = ctx.InversePowerIteration(A, m, epsilon=eps, inner_rtol=1e-5, X0=X0)
    # the oracle with the same number of outer iterations (its own stopping rule may trip one iteration apart: then it
    # is re-run pinned to the GPU's count) -- the comparison below is never skipped
    vecs_ref, vals_ref, st_ref, free_its = parity.oracle_ipi_matching(LA, m, X0, eps, st["outer_its"], inner_rtol=1e-5)
    assert abs(st["outer_its"] - free_its) <= 1
    assert st["residual"] <= eps
    np.testing.assert_allclose(ctx.mat_to_numpy(vals), vals_ref, atol=2e-4)
    V = ctx.mat_to_numpy(vecs).T
    np.testing.assert_allclose(V, vecs_ref, atol=2e-3)
    np.testing.assert_allclose(np.linalg.norm(V, axis=1), 1.0, rtol=1e-5)
    ctx.destroy(A, vecs, vals)


def test_inverse_power_iteration_edge_cases(ctx, golden):
    LA = golden("syn32.npz")["L_A"]
    A = ctx.dense_from_numpy(LA, ld=64)
    with pytest.raises(glf.GlfError):
        ctx.InversePowerIteration(A, 0)
    with pytest.raises(glf.GlfError):
        ctx.InversePowerIteration(A, 9)  # m must be < p
    # opti_gs > 1: eigenvalue estimates become mu^k (reference behaviour), still finite
    X0 = glf.random_vectors(9, 4, 1)
    _, vals3_ref, st_ref = orc.inverse_power_iteration(LA, 4, X0, opti_gs=3, epsilon=1e-3)
    vecs, vals, st = ctx.InversePowerIteration(A, 4, optiGramSchmidt=3, epsilon=1e-3, X0=X0)
    assert st["outer_its"] == st_ref["outer_its"]
    np.testing.assert_allclose(ctx.mat_to_numpy(vals)[:2], vals3_ref[:2], rtol=5e-3)
    ctx.destroy(vecs, vals)
    # outer-iteration cap reports non-convergence instead of spinning
    with pytest.raises(glf.GlfError) as ei:
        ctx.InversePowerIteration(A, 4, epsilon=1e-30, max_outer=3, X0=X0)
    assert ei.value.status == glf.ERR_NOCONV
    ctx.destroy(A)


@pytest.mark.parametrize("name,m", [("syn32", 4), ("syn32", 8), ("test", 16), ("ragged", 5), ("cat50", 53)])
def test_nystroem_permutation_filter(ctx, golden, png, name, m):
    img, p_req = _images(golden, png)[name]
    h, w = img.shape
    N = h * w
    idx = glf.Sampling(w, h, p_req)
    p = idx.size
    KA, _ = orc.affinity(img, idx, want_KB=False)
    LA, alpha = orc.laplacian(KA, orc.degree(img, idx))
    vecs, vals = _lapack_pairs(LA, m)
    phi_sf_ref = orc.nystroem(img, idx, alpha, vecs, vals)          # (m, N) sample-first
    phi_ref = orc.permutation(phi_sf_ref, idx)
    zf_ref, out_ref = orc.result_from_laplacian(img, phi_ref, vals, gain=3.0)

    d_img = ctx.to_device(img)
    _, K_B = ctx.ComputeAffinityMatrices(d_img, idx, want_KA=False)
    L_A, L_B, alpha_gpu = ctx.ComputeLaplacianMatrix(None, K_B)
    phi_A = ctx.dense_from_numpy(vecs.T)
    Pi = ctx.diag_from_numpy(vals)
    Pi_inv = ctx.InverseDiagMat(Pi)
    np.testing.assert_allclose(ctx.mat_to_numpy(Pi_inv), 1.0 / vals, rtol=1e-6)
    phi_sf = ctx.Nystroem(L_B, phi_A, Pi_inv)
    assert (phi_sf.rows, phi_sf.cols, phi_sf.row_order) == (N, m, glf.ROWS_SAMPLE_FIRST)
    scale = np.abs(phi_sf_ref).max()
    got_sf = ctx.mat_to_numpy(phi_sf)
    np.testing.assert_allclose(got_sf, phi_sf_ref.T, rtol=0, atol=PHI_TOL * scale)
    np.testing.assert_array_equal(got_sf[:p], vecs.T.astype(np.float32))   # upper part = phi_A (hpc/nystroem.c:25-34)
    with pytest.raises(glf.GlfError):
        ctx.ComputeResultFromLaplacian(d_img, phi_sf, Pi)                   # must be permuted first
    phi = ctx.Permutation(phi_sf, idx)
    got = ctx.mat_to_numpy(phi)
    np.testing.assert_array_equal(got, orc.permutation(got_sf.T.astype(np.float64), idx).T.astype(np.float32))
    np.testing.assert_array_equal(got[idx], got_sf[:p])                     # hpc/utils.c:149-152
    out, zf = ctx.ComputeResultFromLaplacian(d_img, phi, Pi, gain=3.0)
    zf, out = zf.cpu().numpy(), out.cpu().numpy()
    np.testing.assert_allclose(zf, zf_ref, rtol=0, atol=2e-2)
    assert np.mean(np.abs(out.astype(int) - out_ref.astype(int)) <= 1) >= 0.999
    assert psnr(out, out_ref) >= 50.0
    ctx.destroy(L_A, phi_A, Pi, Pi_inv, phi_sf, phi, K_B)


@pytest.mark.parametrize("w,h,ns,m", [(128, 96, 150, 8), (192, 64, 300, 40), (64, 200, 90, 64)])
def test_nystroem_paths_agree(ctx, w, h, ns, m, monkeypatch):
    """Three implementations of the same contraction: the grid-factored form (GLF_NYS_PATH=grid; the default from
    1024-pixel-wide images on, for the tensor-grid sample sets hpc/sampling.c produces), the direct kernel with
    table-driven generation (GLF_NYS_PATH=direct, width % 64 == 0) and the direct kernel with v_exp_f32 (GLF_NYS_NO_LUT). Each must match the fp64 oracle, and each other far
    below that tolerance."""
    img = glf.synth_image(w, h, seed=5)
    idx = glf.Sampling(w, h, ns)
    KA, _ = orc.affinity(img, idx, want_KB=False)
    LA, alpha = orc.laplacian(KA, orc.degree(img, idx))
    vecs, vals = _lapack_pairs(LA, m)
    ref = orc.nystroem(img, idx, alpha, vecs, vals).T                     # (N, m) sample-first
    d_img = ctx.to_device(img)
    _, K_B = ctx.ComputeAffinityMatrices(d_img, idx, want_KA=False)
    L_A, L_B, _ = ctx.ComputeLaplacianMatrix(None, K_B)
    phi_A, Pi = ctx.dense_from_numpy(vecs.T), ctx.diag_from_numpy(vals)
    Pi_inv = ctx.InverseDiagMat(Pi)
    got = {}
    for mode, tune in (("grid", {"NYS_PATH": "grid"}), ("grid_v1", {"NYS_PATH": "grid", "ROWPASS": "v1"}),
                       ("rank", {"NYS_PATH": "rank"}), ("band", {"NYS_PATH": "band"}),
                       ("lut", {"NYS_PATH": "direct"}), ("exp", {"NYS_PATH": "direct", "NYS_NO_LUT": "1"})):
        ctx.set_tuning(NYS_PATH=None, NYS_NO_LUT=None, ROWPASS=None)
        ctx.set_tuning(**tune)
        phi = ctx.Nystroem(L_B, phi_A, Pi_inv)
        got[mode] = ctx.mat_to_numpy(phi)
        ctx.destroy(phi)
    ctx.set_tuning(NYS_PATH=None, NYS_NO_LUT=None, ROWPASS=None)
    scale = np.abs(ref).max()
    for mode in got:
        np.testing.assert_allclose(got[mode], ref, rtol=0, atol=PHI_TOL * scale, err_msg=mode)
    np.testing.assert_allclose(got["lut"], got["exp"], rtol=0, atol=2e-5 * scale)
    np.testing.assert_allclose(got["grid"], got["exp"], rtol=0, atol=2e-5 * scale)
    # rank form: the photometric table as its rank-R eigen-expansion (max |F F^T - P| <= 2^-30), T' formed in LDS
    np.testing.assert_allclose(got["rank"], got["exp"], rtol=0, atol=2e-5 * scale)
    # band form: the same entries as the direct kernel generates, those beyond the radius (exact zeros in this arithmetic) left out
    np.testing.assert_allclose(got["band"], got["lut"], rtol=0, atol=2e-6 * scale)
    # the two row-pass kernels (row-tile form: Er as the A operand; v1: one image row per wave) split different operands
    np.testing.assert_allclose(got["grid"], got["grid_v1"], rtol=0, atol=2e-5 * scale)
    ctx.destroy(L_A, phi_A, Pi, Pi_inv, K_B)


@pytest.mark.parametrize("w,h,ns", [(128, 96, 150), (450, 300, 1350), (77, 200, 60)])
def test_degree_paths_agree(ctx, w, h, ns, monkeypatch):
    """D_A from the grid-factored form (GLF_DEG_PATH=grid) and from the direct sweep (=direct) against the oracle."""
    img = glf.synth_image(w, h, seed=9)
    idx = glf.Sampling(w, h, ns)
    ref = orc.degree(img, idx)
    d_img = ctx.to_device(img)
    got = {}
    for mode in ("grid", "direct"):
        ctx.set_tuning(DEG_PATH=mode)
        _, K_B = ctx.ComputeAffinityMatrices(d_img, idx, want_KA=False)
        got[mode] = ctx.degree_of(K_B)
        ctx.destroy(K_B)
    ctx.set_tuning(DEG_PATH=None)
    for mode in got:
        np.testing.assert_allclose(got[mode], ref, rtol=2e-6, err_msg=mode)
    np.testing.assert_allclose(got["grid"], got["direct"], rtol=5e-7)


def test_non_grid_sample_set_takes_the_direct_kernels(ctx):
    """The stage API accepts any ascending sample set (hpc/affinity.h:5); one that is not a tensor grid must fall
    back to the direct kernels and still match the oracle."""
    w, h, m = 128, 64, 8
    img = glf.synth_image(w, h, seed=21)
    rng = np.random.default_rng(3)
    idx = np.sort(rng.choice(w * h, size=90, replace=False)).astype(np.uint32)
    KA, _ = orc.affinity(img, idx, want_KB=False)
    deg_ref = orc.degree(img, idx)
    LA, alpha = orc.laplacian(KA, deg_ref)
    vecs, vals = _lapack_pairs(LA, m)
    ref = orc.nystroem(img, idx, alpha, vecs, vals).T
    d_img = ctx.to_device(img)
    _, K_B = ctx.ComputeAffinityMatrices(d_img, idx, want_KA=False)
    np.testing.assert_allclose(ctx.degree_of(K_B), deg_ref, rtol=2e-6)
    L_A, L_B, _ = ctx.ComputeLaplacianMatrix(None, K_B)
    phi_A, Pi = ctx.dense_from_numpy(vecs.T), ctx.diag_from_numpy(vals)
    Pi_inv = ctx.InverseDiagMat(Pi)
    phi = ctx.Nystroem(L_B, phi_A, Pi_inv)
    np.testing.assert_allclose(ctx.mat_to_numpy(phi), ref, rtol=0, atol=PHI_TOL * np.abs(ref).max())
    ctx.destroy(L_A, phi_A, Pi, Pi_inv, K_B, phi)


def test_filter_golden_from_python_poc(ctx, golden, png):
    """z_c goldens (PoC stages + LAPACK pairs, tools/gen_golden.py): GPU L_A -> LAPACK
    pairs -> GPU Nystroem/Permutation/filter must land on them."""
    for npz, img, p_req, m, key in (("syn32.npz", golden("syn32.npz")["img"], 10, 4, "z_c_m4"),
                                    ("test_png.npz", png("test.png"), 100, 16, "z_c_m16"),
                                    ("cat50.npz", png("cat_small.png"), 50, 16, "z_c_m16")):
        g = golden(npz)
        h, w = img.shape
        idx = glf.Sampling(w, h, p_req)
        d_img = ctx.to_device(img)
        _, K_B = ctx.ComputeAffinityMatrices(d_img, idx, want_KA=False)
        L_A, L_B, alpha = ctx.ComputeLaplacianMatrix(None, K_B)
        vecs, vals = _lapack_pairs(ctx.mat_to_numpy(L_A).astype(np.float64), m)
        phi_A, Pi = ctx.dense_from_numpy(vecs.T), ctx.diag_from_numpy(vals)
        Pi_inv = ctx.InverseDiagMat(Pi)
        phi = ctx.Permutation(ctx.Nystroem(L_B, phi_A, Pi_inv), idx)
        out, zf = ctx.ComputeResultFromLaplacian(d_img, phi, Pi)
        np.testing.assert_allclose(zf.cpu().numpy(), g[key], rtol=0, atol=3e-2)
        ctx.destroy(L_A, phi_A, Pi, Pi_inv, phi, K_B)


def _assert_end_to_end(img, ns, m, eps, out, zf, info, eigvals=False, prm=None, corr_tol=CORR_TOL):
    """Unconditional end-to-end comparison with the oracle run of the SAME outer-iteration count: alpha, (eigenvalues,)
    z as rel-L2 on z and -- the sharper measure -- on the correction z - y, PSNR and the 99 % within-one-level rule."""
    kw = dict(inner_rtol=1e-5, seed=1)
    if prm is not None:
        kw["prm"] = prm
    zf_ref, out_ref, ref, free_its = parity.oracle_run_matching(img, ns, m, eps, info["outer_its"], **kw)
    assert (info["p"], info["m"]) == (ref["p"], ref["m"])
    assert info["alpha"] == pytest.approx(ref["alpha"], rel=2e-6)
    assert abs(info["outer_its"] - free_its) <= 1
    if eigvals:
        np.testing.assert_allclose(info["eigvals"], ref["eigvals"], atol=2e-4)
    out, zf = np.asarray(out), np.asarray(zf, dtype=np.float64)
    assert np.linalg.norm(zf - zf_ref) / np.linalg.norm(zf_ref) <= 1e-4
    corr = np.linalg.norm(zf_ref - img.astype(np.float64))
    assert corr > 0 and np.linalg.norm(zf - zf_ref) / corr <= corr_tol, np.linalg.norm(zf - zf_ref) / corr
    assert psnr(out, out_ref) >= 50.0
    assert np.mean(np.abs(out.astype(int) - out_ref.astype(int)) <= 1) >= 0.99
    return ref


E2E = [
    # name, num_samples, m, epsilon
    ("syn32", 10, 4, 1e-3),
    ("test", 100, 16, 0.1),
    ("test", 100, 99, 0.1),       # m = p - 1, the reference default
    ("ragged", 20, 5, 0.1),
    ("cat50", 50, 53, 0.1),       # BASELINE config 1
]


@pytest.mark.parametrize("name,ns,m,eps", E2E)
def test_image_processing_end_to_end(ctx, golden, png, name, ns, m, eps):
    img, _ = _images(golden, png)[name]
    opt = glf.default_options(num_samples=ns, num_eigvals=m, epsilon=eps)
    out, zf, info = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
    out, zf = out.cpu().numpy(), zf.cpu().numpy()
    _assert_end_to_end(img, ns, m, eps, out, zf, info, eigvals=True)


@pytest.mark.parametrize("pool", ["debug", "reusing"])
@pytest.mark.parametrize("mode", ["f16s", "f32"])
def test_call_sequence_of_the_recorded_fault(golden, png, mode, pool, monkeypatch):
    """Regression for the GPU memory access fault recorded in round 1 (gpurun_out/full_test.log: fault inside
    image_processing on the 53x37 image, p = 24, m = 5, ld = 32, on a context that had just run p = 100, m = 99, ld = 128
    and, before that, stage calls up to n = 5000, m = 100; the same test alone passed). The sequence is replayed once on ONE
    fresh context per contraction mode: under the debug pool (GLF_POOL_DEBUG=1: exact-size work buffers followed by
    canary guard zones, NaN-filled instead of holding an earlier call's data, never reused) no guard may be touched and
    no NaN may reach an output; under the normal reusing pool the outputs must be the same to the bit (a result that
    depended on stale pool contents would differ). DESIGN.md section 10 has the audit."""
    import torch
    if pool == "debug":
        monkeypatch.setenv("GLF_POOL_DEBUG", "1")
    else:
        monkeypatch.delenv("GLF_POOL_DEBUG", raising=False)
    c = glf.Context(0)
    c.set_contraction(glf.CONTRACT_F16_SPLIT if mode == "f16s" else glf.CONTRACT_F32_MFMA)
    try:
        assert c.debug_violations() == (0 if pool == "debug" else -1)
        # large stage calls first (what the module-scoped context of the failing run had seen)
        X = orc.random_vectors(5000, 100, 7)
        Xd = c.dense_from_numpy(X.T)
        norms = c.OrthonormaliseVecs(Xd)
        assert np.all(np.isfinite(norms))
        c.destroy(Xd)
        imgs = _images(golden, png)
        outs = {}
        for name, ns, m, eps in (("syn32", 10, 4, 1e-3), ("test", 100, 16, 0.1), ("test", 100, 99, 0.1), ("ragged", 20, 5, 0.1),
                                 ("cat50", 50, 53, 0.1), ("ragged", 20, 5, 0.1)):
            img = imgs[name][0]
            out, zf, info = c.image_processing(c.to_device(img), glf.default_options(num_samples=ns, num_eigvals=m, epsilon=eps),
                                               want_float=True)
            assert torch.isfinite(zf).all()
            _assert_end_to_end(img, ns, m, eps, out.cpu().numpy(), zf.cpu().numpy(), info, eigvals=True)
            outs.setdefault((name, m), []).append(zf.cpu().numpy().view(np.int32))
        np.testing.assert_array_equal(outs[("ragged", 5)][0], outs[("ragged", 5)][1])    # before and after the cat50 call
        if pool == "debug":
            assert c.debug_violations() == 0
        _FAULT_SEQ[(mode, pool)] = outs
        other = _FAULT_SEQ.get((mode, "reusing" if pool == "debug" else "debug"))
        if other is not None:                                                              # debug pool == reusing pool, bit for bit
            for k in outs:
                np.testing.assert_array_equal(outs[k][0], other[k][0])
    finally:
        c.close()


_FAULT_SEQ = {}


@pytest.mark.parametrize("sweeps,colpass", [("segments", "ws"), ("segments", "v1"), ("samples", "ws")])
@pytest.mark.parametrize("name,ns,m,eps", [("ragged", 20, 5, 0.1), ("cat50", 50, 53, 0.1), ("test", 100, 16, 0.1), ("test", 400, 70, 0.1)])
def test_end_to_end_with_rank_forms_forced(ctx, golden, png, name, ns, m, eps, sweeps, colpass):
    """The rank form of the grid-factored contractions (photometric table as a rank-R expansion, T' formed in LDS: the
    default from 1024-pixel-wide images on) forced on the small reference images, for the Nystroem extension and for the
    L_A sweeps of the eigen-solve, against the fp64 oracle. colpass: the column pass with the waves' roles split (k_rank_colpass_ws,
    the default) or every wave forming and contracting (k_rank_colpass); sweeps: the L_A sweeps through that column pass with
    the samples as pixels (default) or through k_rank_samples (the term index contracted on the target side)."""
    ctx.set_tuning(NYS_PATH="rank", DEG_PATH="grid", MV_PATH="rank", SWEEP_COLPASS=sweeps, COLPASS=colpass)
    img, _ = _images(golden, png)[name]
    out, zf, info = ctx.image_processing(ctx.to_device(img), glf.default_options(num_samples=ns, num_eigvals=m, epsilon=eps),
                                         want_float=True)
    if info["contraction"] == glf.CONTRACT_F16_SPLIT:
        assert info["nystroem_path"] == 3 and info["matvec_path"] == 3
    _assert_end_to_end(img, ns, m, eps, out.cpu().numpy(), zf.cpu().numpy(), info)


@pytest.mark.parametrize("name,ns,m,eps", [("ragged", 20, 5, 0.1), ("cat50", 50, 53, 0.1), ("test", 100, 16, 0.1), ("test", 400, 70, 0.1)])
def test_end_to_end_with_band_form_forced(ctx, golden, png, name, ns, m, eps):
    """The band form (k_band: the kernel entries of the samples within the radius of each target generated and contracted
    directly; the default once the radius is small against the image) forced on the small reference images -- whose every
    sample lies within the radius, so the band is the whole grid -- for the Nystroem extension and the L_A sweeps, against
    the fp64 oracle."""
    ctx.set_tuning(NYS_PATH="band", DEG_PATH="grid", MV_PATH="band")
    img, _ = _images(golden, png)[name]
    out, zf, info = ctx.image_processing(ctx.to_device(img), glf.default_options(num_samples=ns, num_eigvals=m, epsilon=eps),
                                         want_float=True)
    if info["contraction"] == glf.CONTRACT_F16_SPLIT:
        assert info["nystroem_path"] == 4 and info["matvec_path"] == 4
        assert info["filter_fused"] == (1 if m <= 64 else 0)   # the filter in k_band's epilogue, Phi never written (one block of <= 64 columns)
    _assert_end_to_end(img, ns, m, eps, out.cpu().numpy(), zf.cpu().numpy(), info)
    if info["contraction"] == glf.CONTRACT_F16_SPLIT and m <= 64:
        # the same run with Phi written and the filter as its own stage: equal up to the order of the f32 sums
        ctx.set_tuning(NO_FUSED_FILTER="1")
        out2, zf2, info2 = ctx.image_processing(ctx.to_device(img), glf.default_options(num_samples=ns, num_eigvals=m, epsilon=eps),
                                                want_float=True)
        ctx.set_tuning(NO_FUSED_FILTER=None)
        assert info2["filter_fused"] == 0 and info2["nystroem_path"] == 4
        assert float((zf2 - zf).abs().max()) <= 2e-4
        assert int((out2.int() - out.int()).abs().max()) <= 1


@pytest.mark.parametrize("rowpass", ["default", "v1", "rt"])
@pytest.mark.parametrize("name,ns,m,eps", [("ragged", 20, 5, 0.1), ("cat50", 50, 53, 0.1), ("test", 100, 16, 0.1)])
def test_end_to_end_with_grid_forms_forced(ctx, golden, png, name, ns, m, eps, rowpass, monkeypatch):
    """The grid-factored degree and Nystroem contraction are chosen automatically from 1024-pixel-wide images on;
    forced here on the small reference images (odd widths, m up to p - 1) against the fp64 oracle. rowpass: the default
    choice (row-tile kernel for the Nystroem passes, k_grid_rowpass for the L_A sweeps), or one kernel for both uses."""
    ctx.set_tuning(NYS_PATH="grid")
    ctx.set_tuning(DEG_PATH="grid")
    ctx.set_tuning(MV_PATH="grid")
    if rowpass != "default":
        ctx.set_tuning(ROWPASS=rowpass)
        ctx.set_tuning(ROWPASS_OP=rowpass)
    img, _ = _images(golden, png)[name]
    out, zf, info = ctx.image_processing(ctx.to_device(img), glf.default_options(num_samples=ns, num_eigvals=m, epsilon=eps),
                                         want_float=True)
    if info["contraction"] == glf.CONTRACT_F16_SPLIT:
        assert info["nystroem_path"] == 1
    _assert_end_to_end(img, ns, m, eps, out.cpu().numpy(), zf.cpu().numpy(), info)


@pytest.mark.parametrize("paths", ["direct", "grid"])
@pytest.mark.parametrize("kernel,ok", [(glf.KERNEL_PHOTOMETRIC, orc.PHOTOMETRIC), (glf.KERNEL_SPATIAL, orc.SPATIAL)])
def test_end_to_end_other_kernels(ctx, kernel, ok, paths, monkeypatch):
    """Photometric-only and spatial-only kernels (hpc/affinity.c:8-57) through the whole path in both kernel families:
    one of the three factors of the grid forms is identically 1, and nothing underflows to zero for the photometric one."""
    ctx.set_tuning(NYS_PATH=paths, DEG_PATH=paths)
    ctx.set_tuning(MV_PATH="grid" if paths == "grid" else "dense")
    img = glf.synth_image(128, 96, seed=13)
    prm = orc.default_params(ok)
    prm.h_loc, prm.h_val = 25.0, 35.0
    opt = glf.default_options(num_samples=160, num_eigvals=12, epsilon=0.05)
    opt.kernel, opt.h_loc, opt.h_val = kernel, 25.0, 35.0
    for skip in (0, 1):
        opt.skip_exact_zeros = skip
        out, zf, info = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
        _assert_end_to_end(img, 160, 12, 0.05, out.cpu().numpy(), zf.cpu().numpy(), info, prm=prm)


def test_barbara_config2(ctx, golden, png):
    """BASELINE config 2: 512x512 barbara, 1 % samples, fp32, one GPU."""
    img = png("barbara.png")
    g = golden("barbara.npz")
    opt = glf.default_options(num_samples=2621, num_eigvals=64, epsilon=0.1)
    out, zf, info = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
    assert info["p"] == 2601 and info["m"] == 64
    assert info["alpha"] == pytest.approx(float(g["alpha"]), rel=2e-6)
    out, zf = out.cpu().numpy(), zf.cpu().numpy()
    _assert_end_to_end(img, 2621, 64, 0.1, out, zf, info, eigvals=True)
    # stage API on the same input lands on the PoC golden (LAPACK pairs)
    idx = glf.Sampling(512, 512, 2621)
    d_img = ctx.to_device(img)
    _, K_B = ctx.ComputeAffinityMatrices(d_img, idx, want_KA=False)
    np.testing.assert_allclose(ctx.degree_of(K_B), g["D_A"], rtol=2e-6)
    L_A, L_B, alpha = ctx.ComputeLaplacianMatrix(None, K_B)
    w, V = np.linalg.eigh(ctx.mat_to_numpy(L_A).astype(np.float64))
    np.testing.assert_allclose(w[:64], g["mu64"], rtol=0, atol=2e-5)
    phi_A, Pi = ctx.dense_from_numpy(V[:, :64]), ctx.diag_from_numpy(w[:64])
    Pi_inv = ctx.InverseDiagMat(Pi)
    phi = ctx.Permutation(ctx.Nystroem(L_B, phi_A, Pi_inv), idx)
    out2, zf2 = ctx.ComputeResultFromLaplacian(d_img, phi, Pi)
    np.testing.assert_allclose(zf2.cpu().numpy()[::64], g["z_c_m64_rows"], rtol=0, atol=5e-2)
    assert psnr(out2.cpu().numpy(), g["z_c_m64_u8"]) >= 50.0
    ctx.destroy(L_A, phi_A, Pi, Pi_inv, phi, K_B)


def test_size_independent_properties_1024(ctx):
    """BASELINE config 5 tile size (1024x1024, 0.5 %): properties that need no oracle run."""
    import torch
    img = glf.synth_image(1024, 1024, seed=5)
    d_img = ctx.to_device(img)
    idx = glf.Sampling(1024, 1024, int(1024 * 1024 * 0.005))
    assert idx.size == 5329
    _, K_B = ctx.ComputeAffinityMatrices(d_img, idx, want_KA=False)
    D = ctx.degree_of(K_B)
    # row sums bounded by the spatial kernel's mass and by 1 (self affinity)
    assert D.min() >= 1.0 and D.max() <= np.pi * 1600 * 1.0001
    # oracle on a sub-sample of the samples (seconds): full-size degree parity
    sub = idx[::211]
    np.testing.assert_allclose(D[::211], orc.degree(img, sub), rtol=2e-6)
    L_A, L_B, alpha = ctx.ComputeLaplacianMatrix(None, K_B)
    LA = ctx.mat_to_numpy(L_A)
    np.testing.assert_array_equal(LA, LA.T)                          # symmetric
    np.testing.assert_allclose(np.diag(LA), alpha * (D - 1.0), rtol=1e-6)
    m = 32
    vecs, vals, st = ctx.InversePowerIteration(L_A, m, epsilon=0.1)
    assert st["residual"] <= 0.1 and st["outer_its"] >= 1
    lam = ctx.mat_to_numpy(vals)
    assert np.all(lam > 0) and np.all(np.isfinite(lam))
    Pi_inv = ctx.InverseDiagMat(vals)
    phi_sf = ctx.Nystroem(L_B, vecs, Pi_inv)
    phi = ctx.Permutation(phi_sf, idx)
    a, b = ctx.mat_to_numpy(phi_sf), ctx.mat_to_numpy(phi)
    np.testing.assert_array_equal(b[idx], a[:idx.size])
    rest = np.ones(1024 * 1024, dtype=bool)
    rest[idx] = False
    np.testing.assert_array_equal(b[rest], a[idx.size:])             # raster order of the remaining pixels
    # a few non-sample pixel rows of Phi against the oracle's kernel (fp64)
    V = ctx.mat_to_numpy(vecs).astype(np.float64)
    prm = orc.default_params()
    for px in (0, 123457, 1024 * 1024 - 1):
        r, c, v = px // 1024, px % 1024, float(img.reshape(-1)[px])
        k = np.array([orc.kernel_entry(prm, (float(i // 1024), float(i % 1024), float(img.reshape(-1)[i])), (r, c, v))
                      for i in idx])
        expect = (-alpha * k) @ (V / lam.astype(np.float64))
        np.testing.assert_allclose(b[px], expect, rtol=0, atol=PHI_TOL * np.abs(b).max())
    # fused path == stage path on the same eigen-solve settings
    out_s, zf_s = ctx.ComputeResultFromLaplacian(d_img, phi, vals)
    opt = glf.default_options(num_samples=int(1024 * 1024 * 0.005), num_eigvals=m, epsilon=0.1)
    out_f, zf_f, info = ctx.image_processing(d_img, opt, want_float=True)
    assert info["p"] == 5329 and info["outer_its"] == st["outer_its"]
    assert torch.max(torch.abs(zf_f - zf_s)).item() <= 2e-2
    assert psnr(out_f.cpu().numpy(), out_s.cpu().numpy()) >= 55.0
    ctx.destroy(L_A, vecs, vals, Pi_inv, phi_sf, phi, K_B)


def test_errors_are_loud(ctx):
    import torch
    img = glf.synth_image(64, 48, seed=1)
    d_img = ctx.to_device(img)
    with pytest.raises(glf.GlfError):       # descending indices
        ctx.ComputeAffinityMatrices(d_img, np.array([50, 40, 30], dtype=np.uint32))
    with pytest.raises(glf.GlfError):       # index out of range
        ctx.ComputeAffinityMatrices(d_img, np.array([5, 64 * 48], dtype=np.uint32))
    with pytest.raises(glf.GlfError):       # more samples than pixels
        ctx.image_processing(d_img, glf.default_options(num_samples=10 ** 6))
    bad = glf.default_options()
    bad.struct_size = 8
    with pytest.raises(glf.GlfError):
        ctx.image_processing(d_img, bad)
    out, _, info = ctx.image_processing(d_img, glf.default_options(num_samples=12, num_eigvals=1000))
    assert info["m"] == info["p"] - 1       # num_eigvals >= p -> p - 1 (hpc/image_processing.c:96-108)
    assert out.dtype == torch.uint8


@pytest.mark.parametrize("paths", ["grid", "direct", "rank"])
def test_exact_zero_skipping_is_bit_identical(ctx, paths, monkeypatch):
    """glf_options.skip_exact_zeros drops whole 64-sample chunks whose kernel entries are exactly zero
    in the arithmetic in use; the result must not change by a single bit, only the executed work."""
    import torch
    _set_paths(ctx, paths)   # the grid-factored forms (all grey levels / rank form) or the entry-by-entry kernels
    img = glf.synth_image(1280, 1024, seed=11)
    d_img = ctx.to_device(img)
    ns = int(1280 * 1024 * 0.005)
    res = {}
    for skip in (0, 1):
        opt = glf.default_options(num_samples=ns, num_eigvals=32, epsilon=0.1, skip_exact_zeros=skip)
        out, zf, info = ctx.image_processing(d_img, opt, want_float=True)
        res[skip] = (out.clone(), zf.clone(), info)
    (o0, z0, i0), (o1, z1, i1) = res[0], res[1]
    assert i0["skip_exact_zeros"] == 0 and i1["skip_exact_zeros"] == 1
    assert torch.equal(o0, o1)
    assert torch.equal(z0.view(torch.int32), z1.view(torch.int32))      # bit for bit
    np.testing.assert_array_equal(i0["eigvals"], i1["eigvals"])
    dense = float(i0["p"]) * 1280 * 1024
    assert i0["nystroem_evaluated"] == pytest.approx(dense, rel=0.02)    # chunk padding only
    assert i1["nystroem_evaluated"] < 0.8 * i0["nystroem_evaluated"]
    assert i0["degree_evaluated"] == dense
    assert i1["degree_evaluated"] < 0.9 * dense
    assert i0["alpha"] == i1["alpha"]                                    # D_A identical to the last bit


@pytest.mark.parametrize("shape", [(32, 32), (53, 37), (100, 100), (128, 96)])
def test_entire_computation_no_approx(ctx, png, shape):
    """-no_approx (hpc/image_processing.c:155-181): z = clamp(y - L y) with the full N x N Laplacian,
    against the fp64 oracle (which is itself checked against numpy in tests/test_oracle_golden.py)."""
    w, h = shape
    img = png("test.png") if shape == (100, 100) else glf.synth_image(w, h, seed=2)
    zf_ref, out_ref = orc.entire_computation(img)
    out, zf, alpha = ctx.EntireComputation(ctx.to_device(img))
    out, zf = out.cpu().numpy(), zf.cpu().numpy()
    np.testing.assert_allclose(zf, zf_ref, rtol=0, atol=2e-3)          # D_i y_i ~ 1e5: f32 output rounding
    assert np.mean(out != out_ref) < 2e-3
    assert np.abs(out.astype(int) - out_ref.astype(int)).max() <= 1
    # the approximate path tends to it as samples and eigenpairs grow (sanity, loose)
    assert 0 < alpha < 1


@pytest.mark.parametrize("paths", ["direct", "grid", "rank", "band"])
@pytest.mark.parametrize("w,h,ns,m", [(16, 12, 6, 2), (24, 31, 9, 3), (200, 160, 500, 128), (256, 256, 655, 256)])
def test_extreme_shapes_end_to_end(ctx, w, h, ns, m, paths, monkeypatch):
    """Tiny images (p < one 64-sample chunk) and the widest supported blocks (ld = 128, 256: the MB = 4 / 8
    instantiations of the direct kernel, two / four 64-column blocks of the grid form) against the oracle.
    m = 256 is the stated upper limit of this build."""
    _set_paths(ctx, paths)
    img = glf.synth_image(w, h, seed=21)
    eps = 0.2
    opt = glf.default_options(num_samples=ns, num_eigvals=m, epsilon=eps)
    out, zf, info = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
    out, zf = out.cpu().numpy(), zf.cpu().numpy()
    if paths != "direct" and info["contraction"] == glf.CONTRACT_F16_SPLIT and info["p"] >= 4:
        assert info["nystroem_path"] == PATH_ID[paths]
    _assert_end_to_end(img, ns, m, eps, out, zf, info)


def test_nlm_kernel_stages_against_golden_and_oracle(ctx, golden):
    """Non-local means (python/affinity_methods/NLM.py:9-34): K_A against the PoC's kernel rows, D_A, L_A, the Nystroem
    contraction and the filter against the oracle, through the stage API (hpc/affinity.h:5 with kernel = GLF_KERNEL_NLM)."""
    g = golden("nlm.npz")
    prm = orc.default_params(orc.NLM)
    prm.h_val = 3.0
    for tag, m in (("syn32", 4), ("rect", 3)):
        img, idx, K = g[tag + "_img"], g[tag + "_idx"], g[tag + "_K"]
        h, w = img.shape
        p = idx.size
        d_img = ctx.to_device(img)
        K_A, K_B = ctx.ComputeAffinityMatrices(d_img, idx, kernel=glf.KERNEL_NLM, h_val=3.0)
        got = ctx.mat_to_numpy(K_A).astype(np.float64)
        ref = K[:, idx]
        big = ref >= 1e-6
        np.testing.assert_allclose(got[big], ref[big], rtol=5e-6)            # f32 patch distances, one v_exp_f32
        np.testing.assert_allclose(got, ref, rtol=3e-4, atol=1e-37)
        D_ref = K.sum(1)
        np.testing.assert_allclose(ctx.degree_of(K_B), D_ref, rtol=3e-6)
        L_A, L_B, alpha = ctx.ComputeLaplacianMatrix(K_A, K_B)
        LA_ref, alpha_ref = orc.laplacian(ref, D_ref)
        assert alpha == pytest.approx(alpha_ref, rel=3e-6)
        np.testing.assert_allclose(ctx.mat_to_numpy(L_A), LA_ref, rtol=0, atol=5e-6 * np.abs(LA_ref).max())
        L_A2, _, _ = ctx.ComputeLaplacianMatrix(None, K_B)                   # regenerated entries (k_nlm_matrix, Laplacian form)
        np.testing.assert_allclose(ctx.mat_to_numpy(L_A2), LA_ref, rtol=0, atol=5e-6 * np.abs(LA_ref).max())
        vecs, vals = _lapack_pairs(LA_ref, m)
        phi_sf_ref = orc.nystroem(img, idx, alpha_ref, vecs, vals, prm=prm)
        phi_A, Pi = ctx.dense_from_numpy(vecs.T), ctx.diag_from_numpy(vals)
        Pi_inv = ctx.InverseDiagMat(Pi)
        phi_sf = ctx.Nystroem(L_B, phi_A, Pi_inv)
        np.testing.assert_allclose(ctx.mat_to_numpy(phi_sf), phi_sf_ref.T, rtol=0, atol=PHI_TOL * np.abs(phi_sf_ref).max())
        phi = ctx.Permutation(phi_sf, idx)
        out, zf = ctx.ComputeResultFromLaplacian(d_img, phi, Pi, gain=3.0)
        zf_ref, out_ref = orc.result_from_laplacian(img, orc.permutation(phi_sf_ref, idx), vals, gain=3.0)
        np.testing.assert_allclose(zf.cpu().numpy(), zf_ref, rtol=0, atol=2e-2)
        assert psnr(out.cpu().numpy(), out_ref) >= 50.0
        ctx.destroy(K_A, L_A, L_A2, phi_A, Pi, Pi_inv, phi_sf, phi, K_B)


@pytest.mark.parametrize("w,h,ns,m", [(96, 80, 60, 8), (53, 37, 20, 5), (160, 128, 300, 40)])
def test_nlm_kernel_end_to_end(ctx, w, h, ns, m):
    """The whole path with the non-local-means affinity (glf_options.kernel = GLF_KERNEL_NLM, h_val = 3 as in the PoC) against
    the oracle: degree sweep, stored L_A, eigen-solve, f32-MFMA Nystroem contraction with patch distances generated in
    registers, filter. 160 x 128 with 300 requested samples: several 64-sample chunks and a ragged last one."""
    img = glf.synth_image(w, h, seed=23)
    prm = orc.default_params(orc.NLM)
    prm.h_val = 3.0
    eps = 0.1
    opt = glf.default_options(num_samples=ns, num_eigvals=m, epsilon=eps)
    opt.kernel, opt.h_val = glf.KERNEL_NLM, 3.0
    out, zf, info = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
    assert info["nystroem_path"] == 0 and info["matvec_path"] == 0         # no factored form exists for patch distances
    _assert_end_to_end(img, ns, m, eps, out.cpu().numpy(), zf.cpu().numpy(), info, eigvals=True, prm=prm)


def test_random_sampler_end_to_end(ctx):
    """glf_options.sampling = GLF_SAMPLING_RANDOM (the PoC's other sampler, python/sampling/random.py; `-sampling random` in the
    host program): the whole path on a non-grid sample set -- entry-by-entry kernels, stored L_A -- against the oracle's
    stages fed the same indices."""
    w, h, ns, m, eps = 96, 80, 70, 8, 1e-3
    img = glf.synth_image(w, h, seed=29)
    opt = glf.default_options(num_samples=ns, num_eigvals=m, epsilon=eps)
    opt.sampling, opt.sampling_seed = glf.SAMPLING_RANDOM, 5
    out, zf, info = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
    idx = glf.RandomSampling(w, h, ns, seed=5)
    assert info["p"] == ns and info["nystroem_path"] == 0 and info["matvec_path"] == 0
    KA, _ = orc.affinity(img, idx, want_KB=False)
    D = orc.degree(img, idx)
    LA, alpha = orc.laplacian(KA, D)
    assert info["alpha"] == pytest.approx(alpha, rel=2e-6)
    X0 = glf.random_vectors(ns, m, 1)
    vecs, vals, st = orc.inverse_power_iteration(LA, m, X0, epsilon=eps, inner_rtol=1e-5)
    np.testing.assert_allclose(info["eigvals"], vals, atol=2e-4)
    phi = orc.permutation(orc.nystroem(img, idx, alpha, vecs, vals), idx)
    zref, out_ref = orc.result_from_laplacian(img, phi, vals, gain=3.0)
    np.testing.assert_allclose(zf.cpu().numpy(), zref.reshape(h, w), rtol=0, atol=2e-2)
    assert np.mean(np.abs(out.cpu().numpy().astype(int) - out_ref.reshape(h, w).astype(int)) <= 1) >= 0.99


def test_nlm_kernel_refuses_images_below_3x3(ctx):
    """The 7 x 7 patches are padded symmetrically (python/affinity_methods/NLM.py:17); nlm.hip reflects an index once, which
    is exact from 3 pixels on -- a smaller image is refused instead of reading past the image."""
    opt = glf.default_options(num_samples=2, num_eigvals=1, epsilon=0.1)
    opt.kernel, opt.h_val = glf.KERNEL_NLM, 3.0
    for w, h in ((2, 9), (9, 2)):
        img = glf.synth_image(w, h, seed=1)
        with pytest.raises(glf.GlfError) as e:
            ctx.image_processing(ctx.to_device(img), opt)
        assert e.value.status == glf.ERR_UNSUPPORTED


@pytest.mark.parametrize("paths", ["direct", "grid", "rank", "band"])
@pytest.mark.parametrize("ns,m", [(600, 0), (600, 300), (300, 257)])
def test_more_than_256_eigenpairs_end_to_end(ctx, ns, m, paths, monkeypatch):
    """The reference's default is m = p - 1 eigenpairs (hpc/image_processing.c:96-108; num_eigvals = 0 here): beyond 256 the
    vectors are processed as panels of 256 columns (cross-panel terms of the classical Gram-Schmidt and of the residual as small
    f64 GEMMs). 256 x 192 image: p = 588 -> m = 587 = panels of 256 + 256 + 75; m = 300 and m = 257 (a one-column last panel)."""
    _set_paths(ctx, paths)
    img = glf.synth_image(256, 192, seed=17)
    eps = 0.2
    p = glf.Sampling(256, 192, ns).size
    m_eff = m if m else p - 1
    assert m_eff > 256
    opt = glf.default_options(num_samples=ns, num_eigvals=m, epsilon=eps)
    out, zf, info = ctx.image_processing(ctx.to_device(img), opt, want_float=True)
    assert (info["p"], info["m"]) == (p, m_eff) and info["eigvals"].shape == (m_eff,)
    _assert_end_to_end(img, ns, m_eff, eps, out.cpu().numpy(), zf.cpu().numpy(), info, eigvals=True)


def test_more_than_256_vectors_stage_api(ctx):
    """The stage API with more than 256 vectors (row-major, ld a multiple of 256): OrthonormaliseVecs / NormaliseVecs,
    InversePowerIteration, Nystroem, Permutation, ComputeResultFromLaplacian against the oracle."""
    n, m = 1500, 300
    X = orc.random_vectors(n, m, 11)
    Q_ref, norms_ref = orc.orthonormalise(X)
    Xd = ctx.dense_from_numpy(X.T)
    assert Xd.ld == 512
    norms = ctx.OrthonormaliseVecs(Xd)
    Q = ctx.mat_to_numpy(Xd).T
    np.testing.assert_allclose(norms, norms_ref, rtol=2e-5)
    np.testing.assert_allclose(Q, Q_ref, rtol=0, atol=2e-4 * np.abs(Q_ref).max() * np.sqrt(m))
    np.testing.assert_allclose(Q.dot(Q.T), np.eye(m), atol=5e-5 * m)
    Y = ctx.dense_from_numpy((X * 3.0).T)
    np.testing.assert_allclose(ctx.NormaliseVecs(Y), 3.0 * np.linalg.norm(X, axis=1), rtol=1e-6)
    ctx.destroy(Xd, Y)
    # eigen-solve + extension + filter, stage by stage, m = 260 of p = 588
    img = glf.synth_image(256, 192, seed=17)
    h, w = img.shape
    idx = glf.Sampling(w, h, 600)
    p, m, eps = idx.size, 260, 0.2
    KA, _ = orc.affinity(img, idx, want_KB=False)
    LA, alpha = orc.laplacian(KA, orc.degree(img, idx))
    X0 = glf.random_vectors(p, m, 1)
    A = ctx.dense_from_numpy(LA, ld=(p + 63) // 64 * 64)
    vecs, vals, st = ctx.InversePowerIteration(A, m, epsilon=eps, inner_rtol=1e-5, X0=X0)
    vecs_ref, vals_ref, st_ref, free_its = parity.oracle_ipi_matching(LA, m, X0, eps, st["outer_its"], inner_rtol=1e-5)
    assert abs(st["outer_its"] - free_its) <= 1 and st["residual"] <= eps
    np.testing.assert_allclose(ctx.mat_to_numpy(vals), vals_ref, atol=2e-4)
    np.testing.assert_allclose(ctx.mat_to_numpy(vecs).T, vecs_ref, atol=2e-3)
    d_img = ctx.to_device(img)
    _, K_B = ctx.ComputeAffinityMatrices(d_img, idx, want_KA=False)
    L_A, L_B, _ = ctx.ComputeLaplacianMatrix(None, K_B)
    Pi_inv = ctx.InverseDiagMat(vals)
    phi_sf = ctx.Nystroem(L_B, vecs, Pi_inv)
    V = ctx.mat_to_numpy(vecs).T.astype(np.float64)
    lam = ctx.mat_to_numpy(vals).astype(np.float64)
    phi_sf_ref = orc.nystroem(img, idx, alpha, V, lam)
    np.testing.assert_allclose(ctx.mat_to_numpy(phi_sf), phi_sf_ref.T, rtol=0, atol=PHI_TOL * np.abs(phi_sf_ref).max())
    phi = ctx.Permutation(phi_sf, idx)
    np.testing.assert_array_equal(ctx.mat_to_numpy(phi), orc.permutation(ctx.mat_to_numpy(phi_sf).T.astype(np.float64), idx).T.astype(np.float32))
    out, zf = ctx.ComputeResultFromLaplacian(d_img, phi, vals, gain=3.0)
    zf_ref, out_ref = orc.result_from_laplacian(img, orc.permutation(phi_sf_ref, idx), lam, gain=3.0)
    np.testing.assert_allclose(zf.cpu().numpy(), zf_ref, rtol=0, atol=2e-2)
    assert psnr(out.cpu().numpy(), out_ref) >= 50.0
    ctx.destroy(A, vecs, vals, L_A, Pi_inv, phi_sf, phi, K_B)


@pytest.mark.parametrize("w,h,ns", [(320, 1536, 30720), (1408, 200, 17600)])
def test_grid_forms_with_many_grid_rows_or_columns(ctx, w, h, ns, monkeypatch):
    """More than 304 grid rows (the Psi slice of the grid row pass no longer fits the LDS beside the A slice and comes
    from global memory) and more than 320 grid columns: grid forms against the direct kernels, dense and skipping."""
    import torch
    img = glf.synth_image(w, h, seed=2)
    d_img = ctx.to_device(img)
    res = {}
    for paths in ("grid", "direct"):
        ctx.set_tuning(NYS_PATH=paths)
        ctx.set_tuning(DEG_PATH=paths)
        ctx.set_tuning(MV_PATH="grid" if paths == "grid" else "dense")
        for skip in (0, 1):
            opt = glf.default_options(num_samples=ns, num_eigvals=24, epsilon=0.2, skip_exact_zeros=skip)
            out, zf, info = ctx.image_processing(d_img, opt, want_float=True)
            res[paths, skip] = (zf.clone(), info)
    g = glf.Sampling(w, h, ns)
    assert res["grid", 0][1]["p"] == g.size
    if res["grid", 0][1]["contraction"] == glf.CONTRACT_F16_SPLIT:
        assert res["grid", 0][1]["nystroem_path"] == 1
    for paths in ("grid", "direct"):   # skipping is bit-identical within a family
        assert torch.equal(res[paths, 0][0].view(torch.int32), res[paths, 1][0].view(torch.int32))
    a, b = res["grid", 0], res["direct", 0]
    assert a[1]["alpha"] == pytest.approx(b[1]["alpha"], rel=1e-6)
    assert a[1]["outer_its"] == b[1]["outer_its"]
    rel = float(torch.linalg.norm(a[0].double() - b[0].double()) / torch.linalg.norm(b[0].double()))
    assert rel < 1e-6


def test_run_to_run_bitwise_reproducible(ctx):
    """Every reduction runs in a fixed order (no atomics): the same input gives the same bits, run after run
    (1280 x 1024: wide enough for the grid-factored forms to be chosen automatically)."""
    import torch
    d_img = ctx.to_device(glf.synth_image(1280, 1024, seed=3))
    opt = glf.default_options(num_samples=6553, num_eigvals=32, epsilon=0.1)
    runs = []
    for _ in range(3):
        out, zf, info = ctx.image_processing(d_img, opt, want_float=True)
        runs.append((out.clone(), zf.clone(), info["alpha"], info["eigvals"].copy()))
    for out, zf, alpha, lam in runs[1:]:
        assert torch.equal(out, runs[0][0]) and torch.equal(zf.view(torch.int32), runs[0][1].view(torch.int32))
        assert alpha == runs[0][2]
        np.testing.assert_array_equal(lam, runs[0][3])


def test_batch_throughput_mode_is_bit_identical_to_single_calls(ctx):
    """BASELINE config 5 (a batch of equally sized tiles sharing one sample set): glf_image_processing_batch deals the
    tiles to several contexts working concurrently; every tile's output must equal the single-image call's, bit for bit
    (here 7 tiles of 256 x 192 on 3 contexts, and the 1-context / more-contexts-than-tiles / empty-batch edges)."""
    import torch
    tiles = np.stack([glf.synth_image(256, 192, seed=100 + t) for t in range(7)])
    d_tiles = torch.from_numpy(tiles).to(ctx.device)
    opt = glf.default_options(num_samples=500, num_eigvals=16, epsilon=0.1)
    single = []
    for t in range(7):
        out, _, info = ctx.image_processing(d_tiles[t], opt)
        single.append((out.cpu().numpy(), info))
    extra = [glf.Context(0) for _ in range(2)]
    try:
        for group in ([ctx] + extra, [ctx], [ctx] + extra):
            outs, infos = glf.image_processing_batch(group, d_tiles if group is not None else d_tiles, opt)
            outs = outs.cpu().numpy()
            for t in range(7):
                assert np.array_equal(outs[t], single[t][0]), "tile %d" % t
                assert infos[t]["p"] == single[t][1]["p"] and infos[t]["outer_its"] == single[t][1]["outer_its"]
                assert infos[t]["alpha"] == single[t][1]["alpha"]
        outs, infos = glf.image_processing_batch([ctx] + extra, d_tiles[:2].contiguous(), opt)   # more contexts than tiles
        assert np.array_equal(outs.cpu().numpy()[1], single[1][0])
        outs, infos = glf.image_processing_batch([ctx] + extra, d_tiles[:0].contiguous(), opt)   # empty batch
        assert outs.shape[0] == 0
        with pytest.raises(glf.GlfError):                                                        # a context listed twice
            glf.image_processing_batch([ctx, ctx], d_tiles, opt)
    finally:
        for c in extra:
            c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["smooth", "sharpen"])
def test_smoothing_and_sharpening_filters(ctx, png, mode):
    """f4: the PoC's `smoothing` (z = W y) and `sharpening` (z = (1 + beta) W^2 y - beta W^3 y, beta = 1.5) filters
    (python/image_processing.py:213, :231-235) with W = Phi L Phi^T on the eigenpairs of this path (L = 1 - mu: W = I - L_A's
    extension). The PoC applies W factor by factor -- its vectors are Nystroem-extended, not orthonormal -- and so does the
    HIP path (the Gram matrix Phi^T Phi between the factors, filter.hip). Checked against the numpy restatements of the PoC's
    two filter lines (oracle.poc_*_filter, themselves pinned to the PoC's outputs in tests/test_oracle_golden.py) fed the
    GPU's own Phi and eigenvalues; a result that assumed Phi^T Phi = I would fail the sharpening case."""
    img = np.ascontiguousarray(png("test.png"))
    if img.ndim == 3:
        img = np.ascontiguousarray(img[:, :, 0])
    h, w = img.shape
    beta = 1.5
    opt = glf.default_options(num_samples=100, num_eigvals=16, epsilon=0.1,
                              filter_mode=glf.FILTER_SMOOTH if mode == "smooth" else glf.FILTER_SHARPEN, filter_beta=beta)
    d_img = ctx.to_device(img)
    out, zf, info = ctx.image_processing(d_img, opt, want_float=True, capture=True)
    out, zf = out.cpu().numpy(), zf.cpu().numpy()
    lam = np.asarray(info["eigvals"], dtype=np.float64)
    m = lam.size
    phi = info["capture"]["phi"].cpu().numpy().astype(np.float64)[:, :m]   # [N, m] raster rows as the filter kernel read them
    y = img.astype(np.float64)
    zref = orc.poc_smoothing_filter(y, phi, 1.0 - lam) if mode == "smooth" else orc.poc_sharpening_filter(y, phi, 1.0 - lam, beta)
    if mode == "sharpen":   # the orthonormal shortcut f(s) = (1 + beta) s^2 - beta s^3 is a different filter on these vectors
        s1 = 1.0 - lam
        z_short = (phi @ (((1.0 + beta) * s1 ** 2 - beta * s1 ** 3) * (phi.T @ y.reshape(-1)))).reshape(h, w)
        assert np.max(np.abs(z_short - zref)) > 20 * 2e-3 * max(1.0, np.max(np.abs(zref)))
    ref8 = np.clip(zref, 0.0, 255.0).astype(np.uint8)
    assert np.max(np.abs(zf.astype(np.float64) - zref)) <= 2e-3 * max(1.0, np.max(np.abs(zref)))
    assert np.mean(out == ref8) >= 0.999 and np.max(np.abs(out.astype(int) - ref8.astype(int))) <= 1
    # the filters do what their names say: smoothing lowers the total variation of the image, sharpening raises it above that
    tv = lambda a: float(np.abs(np.diff(a.astype(np.float64), axis=1)).sum())
    if mode == "smooth":
        assert tv(zf) < tv(img)


@pytest.mark.gpu
def test_sinkhorn_and_orthogonalisation_against_the_poc(ctx, golden):
    """f4: the PoC's balancing steps on the device (csrc/balance.hip). sinkhorn (python/image_processing.py:90-107): the 100
    alternating scalings as 200 products Phi (Pi o (Phi^T x)) on the resident Phi / Pi, then the rows of W_AB; orthogonalisation
    (:110-127) on the PoC's own W_A, W_B. Against the PoC's outputs (tests/golden/f4.npz) and the numpy restatements."""
    g = golden("f4.npz")
    phi64, Pi64 = g["phi"], g["Pi"]                      # [1024, 9] sample-first rows, 9 eigenvalues of K_A
    n = Pi64.size
    phi, Pi = ctx.dense_from_numpy(phi64.astype(np.float32)), ctx.diag_from_numpy(Pi64.astype(np.float32))
    r, c, W_AB = ctx.Sinkhorn(phi, Pi, iterations=100, rows=n)
    # (Phi and Pi travel as f32, the PoC holds them in f64: compare with the restatement on the SAME rounded inputs tightly,
    # with the PoC's own outputs at f32 input accuracy)
    r_ref, c_ref = orc.poc_sinkhorn_scalings(phi64.astype(np.float32).astype(np.float64), Pi64.astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(r, r_ref, rtol=1e-9)
    np.testing.assert_allclose(c, c_ref, rtol=1e-9)
    np.testing.assert_allclose(W_AB[:, :n], g["W_A"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(W_AB[:, n:], g["W_B"], rtol=2e-4, atol=1e-7)
    np.testing.assert_allclose(W_AB.sum(axis=1), 1.0, rtol=1e-5)          # balanced
    ctx.destroy(phi, Pi)
    V, P = ctx.Orthogonalisation(g["W_A"], g["W_B"])
    np.testing.assert_allclose(P, g["Pi_orth"], rtol=1e-9)
    np.testing.assert_allclose(np.abs(V), np.abs(g["V_orth"]), rtol=0, atol=1e-8 * np.abs(g["V_orth"]).max())
    np.testing.assert_allclose(V.T @ V, np.identity(n), atol=1e-9)
    # a larger, odd-sized case against the restatement (the tournament ordering of the Jacobi sweeps has a bye then)
    rng = np.random.RandomState(5)
    X = rng.rand(37, 37)
    A = X @ X.T + 37 * np.identity(37)
    B = rng.rand(37, 211)
    V2, P2 = ctx.Orthogonalisation(A, B)
    V2_ref, P2_ref = orc.poc_orthogonalisation(A, B)
    np.testing.assert_allclose(P2, P2_ref, rtol=1e-10)
    np.testing.assert_allclose(np.abs(V2), np.abs(V2_ref), rtol=0, atol=1e-9 * np.abs(V2_ref).max())


@pytest.mark.gpu
def test_cached_work_buffers_age_out():
    """The workspace pool keeps a call's buffers for the next image of the same size, but not for ever: buffers no call has
    taken for 64 public calls are returned at the next allocation (a service that walks through many image sizes must not
    keep every size's buffers)."""
    c = glf.Context(0)
    try:
        big = glf.synth_image(512, 512, seed=2)
        c.image_processing(c.to_device(big), glf.default_options(num_samples=2600, num_eigvals=32, epsilon=0.1))
        after_big = c.cached_bytes()
        assert after_big > 50 << 20                     # tens of MB of K_A / L_A / vector blocks are cached
        c.image_processing(c.to_device(big), glf.default_options(num_samples=2600, num_eigvals=32, epsilon=0.1))
        assert c.cached_bytes() == after_big            # the same size again allocates nothing
        for i in range(80):                             # 80 calls on small images of changing size: misses trigger the sweep
            s = 32 + 8 * (i % 10)
            c.image_processing(c.to_device(glf.synth_image(s, s, seed=i)), glf.default_options(num_samples=16, num_eigvals=4, epsilon=0.1))
        assert c.cached_bytes() < after_big // 4
    finally:
        c.close()
