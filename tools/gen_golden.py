#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's Python proof of concept.

Runs ONLY in the build container (needs /root/reference); the fixtures it
writes are committed, this script is committed, the reference never travels.
It imports the reference's own modules (python/sampling, python/affinity_methods,
python/image_processing) and calls their functions; the few lines of
python/image_processing.py:274-305 that sit inside the monolithic
image_processing() (behind a dead dense N x N block, :263-272) are re-evaluated
here with numpy on the reference's K_A/K_B so the goldens stay small.

Usage: python tools/gen_golden.py [--skip-barbara]
"""
import argparse
import os
import shutil
import sys
import tempfile

import numpy as np
from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def load_ref():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "python"))
    scratch = tempfile.mkdtemp(prefix="glf_gold_")
    os.makedirs(os.path.join(scratch, "results"))
    os.chdir(scratch)
    import sampling
    import affinity_methods
    import image_processing as ip
    return sampling, affinity_methods, ip


def synth32():
    """32x32 deterministic test image (ours, not the reference's)."""
    r, c = np.mgrid[0:32, 0:32]
    base = 110 + 60 * np.sin(r / 5.0) * np.cos(c / 7.0) + 40 * ((r // 8 + c // 8) % 2)
    rng = np.random.RandomState(1234)
    return np.clip(np.rint(base + rng.normal(0, 12, base.shape)), 0, 255).astype(np.uint8)


def stage_goldens(y, p_req, sampling, affinity_methods, ip, affinity_name="bilateral"):
    """Everything the PoC pins for one image (python/image_processing.py:249-305)."""
    M, N = y.shape
    idx = sampling.methods["spatially_uniform"](M, N, p_req)            # :254
    K_A, K_B = ip.affinity(y, idx, affinity_methods.methods[affinity_name])  # :261
    D_A = np.sum(K_A, axis=1) + np.sum(K_B, axis=1)                      # :274
    alpha = 1.0 / np.mean(D_A)                                           # :275
    L_A = alpha * (np.diag(D_A) - K_A)                                   # :277
    L_B = -alpha * K_B                                                   # :278
    mu, phi_A = np.linalg.eigh(L_A)                                      # :287
    phi = np.concatenate((phi_A, np.dot(L_B.T, phi_A * (1.0 / mu))))     # :288-292
    phi_perm = ip.permutation(phi, idx)                                  # :302
    yv = y.reshape(M * N).astype(np.float64)
    z_py = yv - np.dot(phi_perm * (mu + 5), phi_perm.T.dot(yv))          # :304-305
    return dict(idx=idx.astype(np.uint32), K_A=K_A, K_B=K_B, D_A=D_A, alpha=alpha, L_A=L_A,
                mu=mu, phi_A=phi_A, phi_perm=phi_perm, z_py=z_py.reshape(M, N), yv=yv)


def c_filter_subset(g, m, gain=3.0):
    """The C filter (hpc/display.c:58-83 semantics) evaluated on the PoC's
    LAPACK eigenpairs restricted to the m smallest: sign/rotation-free inside
    well-separated eigenvalues, used to pin Nystroem+permutation+filter."""
    phi = g["phi_perm"][:, :m]
    mu = g["mu"][:m]
    return g["yv"] + gain * np.dot(phi * mu, phi.T.dot(g["yv"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-barbara", action="store_true")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    sampling, affinity_methods, ip = load_ref()

    # input images the reference ships (data files, not source)
    for name in ("test.png", "cat_small.png", "barbara.png", "pixel_mountains.png"):
        shutil.copyfile(os.path.join(REF, "input", name), os.path.join(OUT, name))

    # --- sampling grids (python/sampling/spatially_uniform.py:9-24) -------------
    grids = {}
    for tag, (M, N, p_req) in {
        "test_1pct": (100, 100, 100), "cat_50": (300, 450, 50), "cat_1pct": (300, 450, 1350),
        "barbara_1pct": (512, 512, 2621), "t1024_05pct": (1024, 1024, 5242),
        "s2048_05pct": (2048, 2048, 20971), "s4096_05pct": (4096, 4096, 83886),
        "odd_37x53_20": (37, 53, 20), "syn32_10": (32, 32, 10),
    }.items():
        idx = sampling.methods["spatially_uniform"](M, N, p_req)
        grids[tag + "_shape"] = np.array([M, N, p_req], dtype=np.int64)
        if idx.size <= 6000:
            grids[tag + "_idx"] = idx.astype(np.uint32)
        else:  # large grids: count, ends and a checksum
            grids[tag + "_summary"] = np.array(
                [idx.size, idx[0], idx[1], idx[-1], int(idx.astype(np.uint64).sum() % (1 << 62))],
                dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "sampling.npz"), **grids)

    # --- 32x32 synthetic: every stage in full -----------------------------------
    y32 = synth32()
    g = stage_goldens(y32, 10, sampling, affinity_methods, ip)
    extra = {}
    for name in ("photometric", "spatial"):
        ge = stage_goldens(y32, 10, sampling, affinity_methods, ip, affinity_name=name)
        extra["K_A_" + name] = ge["K_A"]
        extra["K_B_" + name] = ge["K_B"]
    np.savez_compressed(
        os.path.join(OUT, "syn32.npz"), img=y32, idx=g["idx"], K_A=g["K_A"], K_B=g["K_B"], D_A=g["D_A"],
        alpha=g["alpha"], L_A=g["L_A"], mu=g["mu"], phi_perm_abs=np.abs(g["phi_perm"]),
        z_py=g["z_py"], z_c_m4=c_filter_subset(g, 4).reshape(32, 32),
        z_c_m8=c_filter_subset(g, 8).reshape(32, 32), **extra)

    # --- test.png 100x100, 1 % ---------------------------------------------------
    yt = np.array(Image.open(os.path.join(REF, "input", "test.png")))
    assert yt.ndim == 2 and yt.dtype == np.uint8
    g = stage_goldens(yt, int(100 * 100 * 0.01), sampling, affinity_methods, ip)
    np.savez_compressed(
        os.path.join(OUT, "test_png.npz"), idx=g["idx"], K_A=g["K_A"], K_B_cols97=g["K_B"][:, ::97].copy(),
        K_B_fro=np.linalg.norm(g["K_B"]), K_B_y=g["K_B"].dot(np.delete(g["yv"], g["idx"])),
        D_A=g["D_A"], alpha=g["alpha"], mu=g["mu"], z_py=g["z_py"].astype(np.float32),
        z_c_m16=c_filter_subset(g, 16).reshape(100, 100).astype(np.float32),
        z_c_m99=c_filter_subset(g, 99).reshape(100, 100).astype(np.float32))

    # --- cat_small.png, 50 requested samples (BASELINE config 1) -----------------
    yc = np.array(Image.open(os.path.join(REF, "input", "cat_small.png")))
    assert yc.ndim == 2 and yc.dtype == np.uint8
    g = stage_goldens(yc, 50, sampling, affinity_methods, ip)
    np.savez_compressed(
        os.path.join(OUT, "cat50.npz"), idx=g["idx"], K_A=g["K_A"], D_A=g["D_A"], alpha=g["alpha"], mu=g["mu"],
        K_B_y=g["K_B"].dot(np.delete(g["yv"], g["idx"])), z_py=g["z_py"].astype(np.float32),
        z_c_m53=c_filter_subset(g, 53).reshape(yc.shape).astype(np.float32),
        z_c_m16=c_filter_subset(g, 16).reshape(yc.shape).astype(np.float32))

    # --- barbara.png 1 % (BASELINE config 2): summaries only ---------------------
    if not args.skip_barbara:
        yb = np.array(Image.open(os.path.join(REF, "input", "barbara.png")))
        assert yb.ndim == 2 and yb.dtype == np.uint8
        g = stage_goldens(yb, int(512 * 512 * 0.01), sampling, affinity_methods, ip)
        np.savez_compressed(
            os.path.join(OUT, "barbara.npz"), p=g["idx"].size, D_A=g["D_A"], alpha=g["alpha"], mu64=g["mu"][:64],
            mu_max=g["mu"][-1], z_c_m64_u8=np.clip(c_filter_subset(g, 64), 0, 255).astype(np.uint8).reshape(512, 512),
            z_c_m64_rows=c_filter_subset(g, 64).reshape(512, 512)[::64].astype(np.float32))
    print("goldens written to", OUT)


if __name__ == "__main__":
    main()
