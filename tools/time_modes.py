#!/usr/bin/env python3
"""Stage timings of the whole path at a given size with and without exact-zero skipping (GPU box)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "image-processing-graph-laplacian_amd"))
import glf  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
img = glf.synth_image(size, size, seed=0)
with glf.Context(0) as ctx:
    d = ctx.to_device(img)
    for skip in (0, 1):
        opt = glf.default_options(num_samples=int(size * size * 0.005), num_eigvals=m, epsilon=0.1, skip_exact_zeros=skip)
        for rep in range(2):
            out, zf, info = ctx.image_processing(d, opt)
        dense = info["p"] * float(size) * size
        print("skip=%d" % skip, {k[3:]: round(info[k], 2) for k in ("ms_affinity", "ms_laplacian", "ms_eigen", "ms_nystroem", "ms_filter", "ms_total")},
              "outer", info["outer_its"], "nys_frac %.4f deg_frac %.4f" % (info["nystroem_evaluated"] / dense, info["degree_evaluated"] / dense),
              "Mpx/s %.2f" % (size * size / info["ms_total"] / 1e3), "nys_kernel_ms %.2f" % info["nystroem_kernel_ms"])
