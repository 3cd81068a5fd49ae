"""Parity helpers shared by tests/ and bench.py's parity leg.

TEST INFRASTRUCTURE ONLY (like oracle.py): the product package never imports this.

* oracle_run_matching: the fp64 oracle's whole path with the SAME number of outer iterations as a GPU run, so that
  end-to-end comparisons never have to be skipped when the two stopping rules (absolute Frobenius residual <= eps,
  hpc/inverse_power_it.c:161) trip one iteration apart.
* check_rows: sampled image rows of Phi and z of a GPU run at sizes where the oracle cannot run whole (2048^2,
  4096^2): the oracle's Nystroem rows (hpc/nystroem.c:41-57) fed the GPU's Phi_A / eigenvalues, then the filter
  (hpc/display.c:58-83) with c = Phi^T y; errors are reported on Phi and on the CORRECTION z - y (z itself is ~ y, an
  error of per cent in the correction would hide behind it).
"""
import numpy as np

import oracle as orc


def oracle_run_matching(img, ns, m, epsilon, gpu_outer_its, **kw):
    """(zf_ref, out_ref, info_ref, free_its): the oracle's run; when its outer-iteration count (free_its) differs from the
    GPU's it is run again with epsilon = 0 and max_outer = the GPU's count (exactly that many iterations)."""
    zf, out, info = orc.image_processing(img, ns, m, epsilon=epsilon, **kw)
    free_its = info["outer_its"]
    if free_its != gpu_outer_its:
        zf, out, info = orc.image_processing(img, ns, m, epsilon=0.0, max_outer=int(gpu_outer_its), **kw)
        assert info["outer_its"] == gpu_outer_its
    return zf, out, info, free_its


def oracle_ipi_matching(LA, m, X0, epsilon, gpu_outer_its, **kw):
    """(vecs, vals, stats, free_its): the oracle's eigen-solve, pinned to the GPU's outer-iteration count when its own
    stopping rule trips elsewhere; free_its = the count of the unpinned run."""
    vecs, vals, st = orc.inverse_power_iteration(LA, m, X0, epsilon=epsilon, **kw)
    free_its = st["outer_its"]
    if free_its != gpu_outer_its:
        vecs, vals, st = orc.inverse_power_iteration(LA, m, X0, epsilon=0.0, max_outer=int(gpu_outer_its), **kw)
        assert st["outer_its"] == gpu_outer_its
    return vecs, vals, st, free_its


def filter_rows(y_rows, phi_rows, lam, c, gain=3.0, filter_pow=1):
    """hpc/display.c:58-83 on some pixels: z = y + gain Phi (f(Pi) c); clamp (Q4), truncating cast (hpc/utils.c:525)."""
    w = (np.asarray(lam, dtype=np.float64) ** filter_pow) * np.asarray(c, dtype=np.float64)
    z = y_rows.astype(np.float64) + gain * (phi_rows @ w)
    out = np.clip(z, 0.0, 255.0).astype(np.uint8)
    return z, out


def check_rows(img, idx, alpha, phi_A, lam, c, rows, phi_gpu, zf_gpu, out_gpu, gain=3.0, prm=None, corr_gpu=None):
    """img: u8 [H, W]; idx: sample indices; phi_A: [p, m] (GPU eigenvectors, any float dtype); lam, c: [m];
    rows: image rows to check; phi_gpu(r) -> [W, m] array of the GPU's Phi for image row r (raster order, the sample
    pixels holding their Phi_A rows, hpc/utils.c:149-152); zf_gpu(r) -> [W] float z before clamp; out_gpu(r) -> [W] u8;
    corr_gpu(r) -> [W] the correction z - y as the filter kernel computed it (glf_capture.d_corr), optional.
    Returns a dict of error measures over the checked rows (nothing is asserted here). The correction is measured twice:
    from the float z (bounded below by ulp(z) ~ 4e-6 grey levels, which is 1e-3 of the correction at 4096^2 where the
    filter moves a pixel by ~1e-3 grey levels RMS) and, when captured, from the correction term itself."""
    h, w = img.shape
    p, m = phi_A.shape
    phi_A64 = np.ascontiguousarray(phi_A.T, dtype=np.float64)        # oracle layout: m vectors of length p
    lam = np.asarray(lam, dtype=np.float64)
    pos = {int(px): i for i, px in enumerate(idx)}
    phi_err_max = phi_ref_max = 0.0
    rel_big = 0.0
    num_c = den_c = num_z = den_z = num_k = 0.0
    u8_equal = u8_within1 = npx = 0
    u8_maxdiff = 0
    mse = mse_in = 0.0
    for r in rows:
        ref = orc.nystroem_rows(img, idx, alpha, phi_A64, lam, r, r + 1, prm=prm).T      # [W, m], extension formula
        for c_px in range(w):                                                          # sample pixels keep phi_A (hpc/nystroem.c:25-34)
            i = pos.get(r * w + c_px)
            if i is not None:
                ref[c_px] = phi_A[i].astype(np.float64)
        got = np.asarray(phi_gpu(r), dtype=np.float64)
        err = np.abs(got - ref)
        phi_err_max = max(phi_err_max, float(err.max()))
        phi_ref_max = max(phi_ref_max, float(np.abs(ref).max()))
        big = np.abs(ref) > 1e-3 * np.abs(ref).max()
        rel_big = max(rel_big, float((err[big] / np.abs(ref[big])).max()))
        y = img[r].astype(np.float64)
        z_ref, out_ref = filter_rows(y, ref, lam, c, gain)
        z_got = np.asarray(zf_gpu(r), dtype=np.float64)
        num_c += float(np.sum((z_got - z_ref) ** 2))
        den_c += float(np.sum((z_ref - y) ** 2))
        num_z += float(np.sum((z_got - z_ref) ** 2))
        den_z += float(np.sum(z_ref ** 2))
        if corr_gpu is not None:
            num_k += float(np.sum((np.asarray(corr_gpu(r), dtype=np.float64) - (z_ref - y)) ** 2))
        o = np.asarray(out_gpu(r)).astype(np.int64)
        d = np.abs(o - out_ref.astype(np.int64))
        u8_equal += int(np.sum(d == 0))
        u8_within1 += int(np.sum(d <= 1))
        u8_maxdiff = max(u8_maxdiff, int(d.max()))
        mse += float(np.sum(d.astype(np.float64) ** 2))
        mse_in += float(np.sum((out_ref.astype(np.float64) - img[r].astype(np.float64)) ** 2))   # how much the filter did
        npx += w
    mse /= max(1, npx)
    mse_in /= max(1, npx)
    return {
        # PSNR of the reference output against the INPUT rows: how visible the filter is (inf: the 8-bit output equals the input)
        "psnr_ref_vs_input_db": float("inf") if mse_in == 0 else float(10.0 * np.log10(255.0 ** 2 / mse_in)),
        "rows": [int(r) for r in rows], "pixels": npx,
        "phi_max_abs_err_over_max": phi_err_max / phi_ref_max if phi_ref_max > 0 else 0.0,
        "phi_max_rel_err_big_entries": rel_big,          # entries > 1e-3 max|Phi row|
        "rel_l2_correction_from_float_z": (num_c / den_c) ** 0.5 if den_c > 0 else 0.0,   # || z_gpu - z_ref || / || z_ref - y ||
        "rel_l2_correction": ((num_k / den_c) ** 0.5 if den_c > 0 else 0.0) if corr_gpu is not None else None,
        "rms_err_z_grey_levels": (num_c / max(1, npx)) ** 0.5,
        "rel_l2_z": (num_z / den_z) ** 0.5 if den_z > 0 else 0.0,
        "rms_correction_grey_levels": (den_c / max(1, npx)) ** 0.5,
        "u8_equal_frac": u8_equal / max(1, npx), "u8_within1_frac": u8_within1 / max(1, npx), "u8_max_diff": u8_maxdiff,
        "psnr_db": float("inf") if mse == 0 else float(10.0 * np.log10(255.0 ** 2 / mse)),
    }
