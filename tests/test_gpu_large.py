"""The benchmark-sized configurations under test (-m gpu): BASELINE cfg3 (2048^2, 0.5 %, m = 64) and cfg4 (4096^2, the
headline), through the kernels bench.py times (grid-factored degree, L_A operator, Nystroem passes incl. the
2048-row pass boundary), against the fp64 oracle on sampled slices -- the oracle cannot run those sizes whole
(4096^2: ~2 h on 128 threads).

  degree      D_A on ~48 samples spread over the grid vs orc.degree (hpc/laplacian.c:18-20)          rel 2e-6
  eigenpairs  2048^2: the oracle's whole eigen-solve (hpc/inverse_power_it.c:86-252) on L_A built from the oracle's
              K_A and the GPU's degree vector, same X0 / stopping rule, iteration count pinned when it differs;
              both sizes: the reported residual || (I - X X^T) A X ||_F recomputed in fp64 from a dense L_A
  Phi rows    orc.nystroem_rows (hpc/nystroem.c:41-57) fed the GPU's Phi_A, eigenvalues, alpha: rows 0, 1, the pass
              boundary H/2 - 1 | H/2, H - 33, H - 1, ...                                   abs 2e-5 * max|Phi|
  z rows      hpc/display.c:58-83 on those rows with c = Phi^T y recomputed in fp64: error measured on the
              correction z - y as the filter kernel computes it (glf_capture.d_corr: rel-L2 <= 2e-4) and on the float z
              (RMS <= 6e-6 grey levels = ulp(z): at 4096^2 the filter moves a pixel by ~1e-3 grey levels RMS, so the
              float z cannot carry the correction to better than ~4e-3 relative), u8: >= 99.9 % within one grey
              level, PSNR >= 50 dB

Arithmetic (SURVEY 8d "fp32"): every contraction of the default path multiplies operands split into f16 (hi, lo) pairs
(22 significant bits, f32 accumulation). test_contraction_arithmetic_bounds compares that against the exact-f32-operand
MFMA kernels and the fp64 oracle on Phi and on z - y, incl. kernels with a large dynamic range.
"""
import json
import os
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import glf  # noqa: E402
import oracle as orc  # noqa: E402
import parity  # noqa: E402  (oracle/parity.py)
from conftest import ROOT  # noqa: E402
from test_abi import SYNTH_CRC32  # noqa: E402

PHI_TOL = 2e-5          # max |Phi_gpu - Phi_ref| / max |Phi_ref| over the checked rows
CORR_TOL = 2e-4         # || corr_gpu - (z_ref - y) || / || z_ref - y || over the checked rows (the correction term itself)
Z_RMS_TOL = 6e-6        # RMS error of the float z in grey levels: ulp(z) / sqrt(12) for z ~ 128 is 4.4e-6


@pytest.fixture(scope="module")
def ctx():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a visible MI355X"
    c = glf.Context(0)
    yield c
    c.close()


def _record(name, payload):
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, name), "w") as f:
            json.dump(payload, f, indent=1, default=float)


def _dense_LA_tensor(ctx, d_img, idx):
    """L_A through the stage API (k_sample_matrix: entries checked against the oracle by test_gpu_parity.py and below)."""
    _, K_B = ctx.ComputeAffinityMatrices(d_img, idx, want_KA=False)
    L_A, _, alpha = ctx.ComputeLaplacianMatrix(None, K_B)
    t = glf.device_tensor_from_ptr(L_A.data, int(L_A.rows) * int(L_A.ld), ctx.torch.float32, ctx.device).view(int(L_A.rows), int(L_A.ld))
    return t, L_A, K_B, alpha


def _residual_fp64(torch, LA, p, X):
    """|| A Q - Q (Q^T A Q) ||_F in fp64 for Q = orth(X) (the norm depends on span(X) only); also the Ritz values."""
    Q, _ = torch.linalg.qr(X.double())
    AQ = torch.empty_like(Q)
    for i0 in range(0, p, 4096):
        i1 = min(p, i0 + 4096)
        AQ[i0:i1] = LA[i0:i1, :p].double() @ Q
    G = Q.T @ AQ
    R = AQ - Q @ G
    return float(torch.linalg.norm(R)), torch.linalg.eigvalsh(0.5 * (G + G.T)).cpu().numpy()


@pytest.mark.parametrize("size,form,gain", [(1024, "band", 3.0), (2048, "band", 3.0), (2048, "rank", 3.0), (2048, "grid", 3.0),
                                            (2048, "band", 2000.0), (4096, "band", 3.0), (4096, "rank", 3.0), (4096, "grid", 3.0)])
def test_headline_config_sampled_parity(ctx, size, form, gain):
    """form: "band" = the default kernels (k_band: the kernel entries of the samples within the radius of each target generated
    and contracted directly: what bench.py's headline times), "rank" / "grid" = the grid-factored contractions, which carry every
    sample (rank: the photometric table as a rank-R expansion, T' formed in LDS; grid: all 256 grey levels through HBM;
    GLF_NYS_PATH / GLF_MV_PATH: bench.py's other legs). 1024: BASELINE cfg5's tile (m = 64, whole path). gain:
    hpc/display.c:73 has 3.0, with which the filter moves a pixel by ~1e-3 grey levels at these sizes -- the 8-bit output
    is y or y - 1 and its comparison nearly vacuous; the gain = 2000 case moves pixels by several grey levels, so that the
    u8 / PSNR comparison with the oracle can fail."""
    torch = ctx.torch
    ctx.reset_tuning()
    if form != "band":
        ctx.set_tuning(NYS_PATH=form, MV_PATH=form)
    img = glf.synth_image(size, size, seed=0)
    assert zlib.crc32(img.tobytes()) == SYNTH_CRC32[size]
    N, m, eps = size * size, 64, 0.1
    ns = int(N * 0.005)
    idx = glf.Sampling(size, size, ns)
    p = idx.size
    assert p == {1024: 5329, 2048: 21316, 4096: 85264}[size]
    d_img = ctx.to_device(img)
    opt = glf.default_options(num_samples=ns, num_eigvals=m, epsilon=eps)
    opt.gain = gain
    if size == 1024:
        ctx.set_tuning(MV_PATH="band")        # (auto keeps a stored L_A below 16384 samples; the tile goes through the band form here)
    out, zf, info = ctx.image_processing(d_img, opt, want_float=True, capture=True)
    if form == "band":
        # what bench.py times is the same call WITHOUT the capture: then Phi is not written at all -- the filter runs in the
        # epilogue of k_band and c = Phi^T y comes from the degree stage's value-weighted sums. Its output
        # against the captured run's (Phi written, k_apply_filter), whose Phi / c / z the checks below pin to the oracle:
        out_f, zf_f, info_f = ctx.image_processing(d_img, opt, want_float=True)
        assert info_f["filter_fused"] == 1 and info["filter_fused"] == 0 and info_f["nystroem_path"] == 4
        np.testing.assert_array_equal(info_f["eigvals"], info["eigvals"])
        dz = (zf_f - zf).abs()
        # 4 ulp(z) + the rounding of the 64-term f32 dot product Phi[px] . w, which the gain multiplies (gain = 2000: outlier
        # pixels are corrected by thousands of grey levels before the clamp)
        assert float(dz.max()) <= 6e-5 + 1e-5 * gain, float(dz.max())
        assert float((zf_f - zf).double().norm() / (zf - d_img.float()).double().norm()) <= (2e-6 if gain > 3.0 else 2e-2)
        d8 = (out_f.int() - out.int()).abs()
        assert int(d8.max()) <= 1 and float((d8 != 0).float().mean()) <= 1e-4
        del out_f, zf_f, dz, d8
    ctx.reset_tuning()
    cap = info["capture"]
    # the kernel families bench.py times
    want_path = {"band": 4, "rank": 3, "grid": 1}[form]
    assert info["nystroem_path"] == want_path and info["matvec_path"] == want_path and info["contraction"] == glf.CONTRACT_F16_SPLIT
    assert (info["p"], info["m"], cap["ld"]) == (p, m, 64)
    assert torch.isfinite(zf).all() and info["residual"] <= eps
    report = {"size": size, "form": form, "p": p, "m": m, "outer_its": info["outer_its"], "inner_its_total": info["inner_its_total"],
              "residual": info["residual"], "ms_total": info["ms_total"]}

    # ---- degree on a sub-sample of the samples, alpha -----------------------------------------------------------------
    D = cap["degree"]
    sub = np.unique(np.concatenate([np.arange(0, p, max(1, p // 40)), [p - 1, p // 2]])).astype(np.int64)
    D_ref = orc.degree(img, idx[sub])
    np.testing.assert_allclose(D[sub], D_ref, rtol=2e-6)
    assert info["alpha"] == pytest.approx(p / D.sum(), rel=1e-12)
    report["degree_max_rel"] = float(np.max(np.abs(D[sub] / D_ref - 1.0)))

    # ---- eigenpairs ------------------------------------------------------------------------------------------------------
    LA_t, L_A, K_B, alpha_stage = _dense_LA_tensor(ctx, d_img, idx)
    assert alpha_stage == pytest.approx(info["alpha"], rel=1e-12)
    # rows of the dense L_A against the oracle (first, middle, last sample)
    for i in (0, p // 2, p - 1):
        row_ref = orc.laplacian_rows(img, idx, D, info["alpha"], i, i + 1)[0]
        np.testing.assert_allclose(LA_t[i, :p].cpu().numpy(), row_ref, rtol=0, atol=3e-6 * np.abs(row_ref).max())
    phi_A = cap["phi_A"][:, :m]
    np.testing.assert_allclose(torch.linalg.norm(phi_A.double(), dim=0).cpu().numpy(), 1.0, rtol=2e-6)   # NormaliseVecs, :230
    res64, ritz = _residual_fp64(torch, LA_t, p, phi_A)
    report["residual_fp64_from_dense_LA"] = res64
    assert res64 == pytest.approx(info["residual"], rel=2e-2)      # the stopping rule's quantity, recomputed independently
    lam = info["eigvals"]
    assert np.all(np.isfinite(lam)) and np.all(lam > 0)
    report["eigval_vs_ritz_max_rel"] = float(np.max(np.abs(np.sort(lam) / ritz - 1.0)))
    assert report["eigval_vs_ritz_max_rel"] <= 0.5                    # sanity only: the reference's 1 / |u_j| estimates (:204) are loose at eps = 0.1
    if size <= 2048 and gain == 3.0:
        KA, _ = orc.affinity(img, idx, want_KB=False)
        LA_ref, alpha_ref = orc.laplacian(KA, D)
        del KA
        assert alpha_ref == pytest.approx(info["alpha"], rel=1e-12)
        X0 = glf.random_vectors(p, m, 1)
        vecs_ref, vals_ref, st_ref, free_its = parity.oracle_ipi_matching(LA_ref, m, X0, eps, info["outer_its"], inner_rtol=1e-5)
        assert abs(free_its - info["outer_its"]) <= 1
        del LA_ref
        np.testing.assert_allclose(lam, vals_ref, atol=2e-4)
        V = phi_A.cpu().numpy().T.astype(np.float64)
        report["eigvec_max_abs_err"] = float(np.abs(V - vecs_ref).max())
        np.testing.assert_allclose(V, vecs_ref, atol=2e-3)
        report["oracle_outer_its_free"] = free_its
    del LA_t
    ctx.destroy(L_A, K_B)

    # ---- Phi rows and z rows ---------------------------------------------------------------------------------------------
    phi = cap["phi"]                                                   # [N, 64] raster order
    y64 = d_img.reshape(-1).double()
    c64 = (phi.double().T @ y64).cpu().numpy()[:m]
    np.testing.assert_allclose(cap["c"], c64, rtol=0, atol=1e-6 * np.abs(c64).max())    # the library's Phi^T y reduction
    report["c_max_rel"] = float(np.abs(cap["c"] - c64).max() / np.abs(c64).max())
    half = size // 2                                                   # 4096: the boundary between the two 2048-row passes
    rows = sorted({0, 1, 33, half - 1, half, half + 1, size - 33, size - 1, 1000})
    phi_v = phi.view(size, size, 64)
    res = parity.check_rows(img, idx, info["alpha"], phi_A.cpu().numpy(), lam, c64, rows,
                            phi_gpu=lambda r: phi_v[r, :, :m].cpu().numpy(), zf_gpu=lambda r: zf[r].cpu().numpy(),
                            out_gpu=lambda r: out[r].cpu().numpy(), gain=gain, corr_gpu=lambda r: cap["corr"].view(size, size)[r].cpu().numpy())
    report["rows_check"] = res
    if gain > 100.0:     # the filter must be visible in this case: the reference output differs from the input by >= 1 grey level RMS
        assert res["rms_correction_grey_levels"] >= 1.0 and res["psnr_ref_vs_input_db"] < 48.0
    report["eigvals_min_max"] = [float(lam.min()), float(lam.max())]
    _record("large_parity_%d%s%s.json" % (size, "" if form == "band" else "_" + form, "" if gain == 3.0 else "_gain%g" % gain), report)
    print(json.dumps(report, default=float))
    assert res["phi_max_abs_err_over_max"] <= PHI_TOL
    assert res["rel_l2_correction"] <= CORR_TOL                       # the correction term z - y itself
    z_tol = Z_RMS_TOL + CORR_TOL * res["rms_correction_grey_levels"]  # (ulp(z)-bound; a large gain scales the correction's own error)
    assert res["rms_err_z_grey_levels"] <= z_tol                      # and the float z
    assert res["u8_within1_frac"] >= 0.999 and res["u8_max_diff"] <= 1
    assert res["psnr_db"] >= 50.0
    # the whole image: every pixel's correction follows from its Phi row -- z recomputed in fp64 from the captured Phi
    wv = torch.from_numpy(gain * lam * c64).to(ctx.device)
    corr = torch.empty(N, dtype=torch.float64, device=ctx.device)
    for i0 in range(0, N, 1 << 20):                                    # (one 16.7M-row gemv is more than hipBLAS launches)
        corr[i0:i0 + (1 << 20)] = (phi[i0:i0 + (1 << 20), :m].double() * wv).sum(1)
    z_all = y64 + corr
    rel_all = float(torch.linalg.norm(cap["corr"].double() - corr) / torch.linalg.norm(corr))
    report["rel_l2_correction_all_pixels_vs_fp64_from_gpu_phi"] = rel_all
    assert rel_all <= 1e-5, rel_all                                   # the filter kernel itself (f32 dot of 64 terms) on every pixel
    assert float(torch.sqrt(torch.mean((zf.reshape(-1).double() - z_all) ** 2))) <= z_tol
    out_all = torch.clamp(torch.floor(z_all), 0.0, 255.0).to(torch.uint8)        # trunc(z) for z >= 0, 0 below (Q4); clamp at 255 (hpc/display.c:75-78)
    mism = float((out_all.reshape(size, size) != out).double().mean())
    report["u8_mismatch_frac_all_pixels"] = mism
    _record("large_parity_%d%s%s.json" % (size, "" if form == "band" else "_" + form, "" if gain == 3.0 else "_gain%g" % gain), report)
    assert mism <= 1e-4, mism               # only where the f32 / f64 corrections straddle an integer


HDR_KERNELS = [
    # (h_loc, h_val): the reference's constants; a narrow spatial kernel (the radius where 2^15 Er rounds to a zero f16,
    # 5.3 h_loc, falls inside the sample neighbourhood the f32 kernels still see out to 10.2 h_loc) with a flat
    # photometric factor; and a sharp photometric factor (entries down to the f32 denormals)
    (40.0, 30.0), (6.0, 400.0), (40.0, 5.0),
]


@pytest.mark.parametrize("size", [1024, 2048])
def test_contraction_arithmetic_bounds(ctx, size, monkeypatch):
    """Phi and z - y from (a) the default contraction (band form), (a') the grid-factored forms (rank form; all 256
    grey levels), (b) the entry-by-entry split-f16 kernel and (c) the exact-f32-operand MFMA kernel, each against the fp64 oracle on sampled rows; same Phi_A / eigenvalues for all
    three (one eigen-solve)."""
    torch = ctx.torch
    img = glf.synth_image(size, size, seed=0)
    assert zlib.crc32(img.tobytes()) == SYNTH_CRC32[size]
    N, m = size * size, 64
    ns = int(N * 0.005)
    idx = glf.Sampling(size, size, ns)
    p = idx.size
    d_img = ctx.to_device(img)
    y64 = d_img.reshape(-1).double()
    rows = [0, size // 2 - 1, size // 2, size - 1]
    report = {}
    for h_loc, h_val in (HDR_KERNELS if size == 1024 else HDR_KERNELS[:1]):
        prm = orc.default_params()
        prm.h_loc, prm.h_val = h_loc, h_val
        ctx.set_contraction(glf.CONTRACT_F16_SPLIT)
        _, K_B = ctx.ComputeAffinityMatrices(d_img, idx, want_KA=False, h_loc=h_loc, h_val=h_val)
        L_A, L_B, alpha = ctx.ComputeLaplacianMatrix(None, K_B)
        # eigenpairs of this L_A (LAPACK on the host at 1024^2 would take minutes: the library's own solve, once)
        vecs, vals, st = ctx.InversePowerIteration(L_A, m, epsilon=0.1, allow_noconv=True, max_outer=8)
        lam = ctx.mat_to_numpy(vals).astype(np.float64)
        phi_A = ctx.mat_to_numpy(vecs)
        Pi_inv = ctx.InverseDiagMat(vals)
        got = {}
        for mode in ("band_f16s", "rank_f16s", "grid_f16s", "direct_f16s", "direct_f32"):
            ctx.set_contraction(glf.CONTRACT_F32_MFMA if mode == "direct_f32" else glf.CONTRACT_F16_SPLIT)
            # (rank: h_val = 5 needs more than 64 terms for 2^-30 -- that kernel falls back to the exact grid form, by design)
            ctx.set_tuning(NYS_PATH={"band_f16s": "band", "rank_f16s": "rank", "grid_f16s": "grid"}.get(mode, "direct"))
            phi_sf = ctx.Nystroem(L_B, vecs, Pi_inv)
            phi = ctx.Permutation(phi_sf, idx)
            ctx.destroy(phi_sf)
            out, zf = ctx.ComputeResultFromLaplacian(d_img, phi, vals, gain=3.0)
            pt = glf.device_tensor_from_ptr(phi.data, N * int(phi.ld), torch.float32, ctx.device).view(size, size, int(phi.ld))
            c64 = (pt.reshape(N, -1).double().T @ y64).cpu().numpy()[:m]
            res = parity.check_rows(img, idx, alpha, phi_A, lam, c64, rows, phi_gpu=lambda r: pt[r, :, :m].cpu().numpy(),
                                    zf_gpu=lambda r: zf[r].cpu().numpy(), out_gpu=lambda r: out[r].cpu().numpy(), gain=3.0, prm=prm)
            got[mode] = (pt[rows].clone(), res)
            del pt
            ctx.destroy(phi)
        ctx.set_tuning(NYS_PATH=None)
        ctx.set_contraction(glf.CONTRACT_F16_SPLIT)
        key = "%dx%d h_loc=%g h_val=%g" % (size, size, h_loc, h_val)
        report[key] = {mode: got[mode][1] for mode in got}
        report[key]["eigen_solve"] = dict(st, lam_min=float(lam.min()), lam_max=float(lam.max()))
        print(json.dumps({key: report[key]}, default=float))
        _record("arithmetic_%d.json" % size, report)
        for mode, (_, res) in got.items():
            assert res["phi_max_abs_err_over_max"] <= PHI_TOL, (key, mode, res)
            # (the stage API returns the float z only: its error is bounded by ulp(z), the correction by that on top)
            assert res["rms_err_z_grey_levels"] <= Z_RMS_TOL + CORR_TOL * res["rms_correction_grey_levels"], (key, mode, res)
            assert res["u8_within1_frac"] >= 0.999, (key, mode, res)
        # the split-f16 results against the exact-f32-operand kernel directly (same rows): the 22-bit operands cost less
        # than the fp32 accumulation already does
        ref32 = got["direct_f32"][0].double()
        for mode in ("band_f16s", "rank_f16s", "grid_f16s", "direct_f16s"):
            d = float((got[mode][0].double() - ref32).abs().max() / ref32.abs().max())
            report[key][mode + "_vs_f32_max_abs_over_max"] = d
            assert d <= PHI_TOL, (key, mode, d)
        ctx.destroy(L_A, vecs, vals, Pi_inv, K_B)
    _record("arithmetic_%d.json" % size, report)
    print(json.dumps(report, default=float))


@pytest.mark.gpu
@pytest.mark.parametrize("mv", ["band", "rank", "grid"])
@pytest.mark.parametrize("size,m", [(1024, 64), (1024, 100)])
def test_narrow_sweeps_return_the_same_numbers(size, m, mv):
    """Block PCG applies the operator to the still-iterating columns only once few are left (packed into a block of 32
    columns, or of a multiple of 64 below ld): the operator's columns are independent, so eigenvalues and the filtered image
    must come out bit for bit as with full-width sweeps (GLF_NO_NARROW), and the narrow sweeps must actually occur."""
    img = glf.synth_image(size, size, seed=5)
    res = {}
    for narrow in (True, False):
        c = glf.Context(0)
        try:
            c.set_tuning(NO_NARROW="0" if narrow else "1", MV_PATH=mv)
            opt = glf.default_options(num_samples=int(size * size * 0.005), num_eigvals=m, epsilon=0.05)
            out, zf, info = c.image_processing(c.to_device(img), opt, want_float=True)
            res[narrow] = (out.cpu().numpy(), zf.cpu().numpy(), info)
        finally:
            c.close()
    a, b = res[True], res[False]
    assert a[2]["narrow_sweeps"] > 0 and b[2]["narrow_sweeps"] == 0
    assert a[2]["outer_its"] == b[2]["outer_its"] and a[2]["matvecs"] == b[2]["matvecs"]
    np.testing.assert_array_equal(a[2]["eigvals"].view(np.int64), b[2]["eigvals"].view(np.int64))
    np.testing.assert_array_equal(a[1].view(np.int32), b[1].view(np.int32))
    np.testing.assert_array_equal(a[0], b[0])
