#!/usr/bin/env python3
"""Generate tests/golden/f4.npz: the PoC's alternative filters and balancing steps on the 32 x 32 test image.

Runs ONLY in the build container (needs /root/reference). Calls the reference's own functions
(python/image_processing.py): nystroem :69-86, sinkhorn :90-107, orthogonalisation :110-127, smoothing_matrix :151-194,
smoothing :197-219, sharpening :222-241 -- on the inputs the PoC itself would give them (:259-262, :341, :357: the affinity's
Nystroem pairs in sample-first order). Stored: the inputs and every output, so that the fp64 restatement (oracle/oracle.py,
poc_*) is pinned without the reference travelling.

Usage: python tools/gen_golden_f4.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, load_ref, synth32  # noqa: E402


def main():
    sampling, affinity_methods, ip = load_ref()
    ip.save_image = True                       # display_or_save writes under ./results of the scratch directory, no window
    y = synth32()
    M, N = y.shape
    idx = sampling.methods["spatially_uniform"](M, N, 10)                       # :254 (p = 9 on 32 x 32)
    K_A, K_B = ip.affinity(y, idx, affinity_methods.methods["bilateral"])        # :261
    phi, Pi = ip.nystroem(K_A, K_B)                                              # :262 -- sample-first rows, descending Pi
    W_A, W_B = ip.sinkhorn(phi, Pi)                                              # :90-107
    V_o, Pi_o = ip.orthogonalisation(W_A.copy(), W_B.copy())                     # :110-127
    V_s, L_s = ip.smoothing_matrix(idx, phi.copy(), Pi.copy())                   # :151-194
    z_smooth = ip.smoothing(y.astype(np.float64), idx, phi.copy(), Pi.copy())    # :197-219
    z_sharp = ip.sharpening(y.astype(np.float64), idx, phi.copy(), Pi.copy())    # :222-241
    np.savez_compressed(os.path.join(OUT, "f4.npz"), img=y, idx=np.asarray(idx), K_A=K_A, K_B=K_B, phi=phi, Pi=Pi, W_A=W_A, W_B=W_B,
                        V_orth=V_o, Pi_orth=Pi_o, V_smooth=V_s, L_smooth=L_s, z_smooth=z_smooth, z_sharp=z_sharp)
    print("f4.npz:", {k: v.shape for k, v in dict(phi=phi, W_A=W_A, W_B=W_B, V_orth=V_o, V_smooth=V_s, z_smooth=z_smooth).items()})


if __name__ == "__main__":
    main()
