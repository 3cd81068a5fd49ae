#!/usr/bin/env python3
"""tests/golden/nlm.npz from the reference PoC's non-local-means affinity (python/affinity_methods/NLM.py:9-34).

Runs ONLY in the build container (imports /root/reference/python; the fixture and this script are committed, the reference
never travels). The PoC's kernel row holds the pixel (row j, col i) at column i*M + j (im2col of the transposed padded image,
NLM.py:21) although everything downstream indexes it by raster index (python/image_processing.py:59-64); the fixture stores the
rows re-indexed to raster order: K[s][row * N + col] = K_AB[s][col * M + row]. Arithmetic, padding and weights are the PoC's.
"""
import os
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import synth32  # noqa: E402


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(REF, "python"))
    scratch = tempfile.mkdtemp(prefix="glf_gold_")
    os.makedirs(os.path.join(scratch, "results"))
    os.chdir(scratch)
    import sampling
    import affinity_methods
    out = {}
    y32 = synth32()
    rect = np.ascontiguousarray(synth32()[3:27, :20])      # 24 x 20: rows != columns, where the PoC's layout shows
    for tag, y, p_req in (("syn32", y32, 10), ("rect", rect, 8)):
        M, N = y.shape
        idx = sampling.methods["spatially_uniform"](M, N, p_req)
        K_AB = affinity_methods.methods["NLM"](y, idx)       # (p, M*N), column i*M + j <-> pixel (row j, col i)
        rows, cols = np.mgrid[0:M, 0:N]
        K = K_AB[:, (cols * M + rows).reshape(-1)]            # raster order
        out[tag + "_img"] = y
        out[tag + "_idx"] = idx.astype(np.uint32)
        out[tag + "_K"] = K
    np.savez_compressed(os.path.join(OUT, "nlm.npz"), **out)
    print("written", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
