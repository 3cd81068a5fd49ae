"""Profiling build -DRK_STAMP (tools/build_variant.sh stamp "-DRK_STAMP"): one 4096^2 run, then the s_memtime stamps of
workgroup 0 of the LAST k_rank_colpass launch: GLF_LIBRARY=tools/dbg/libglf_stamp.so python tools/rank_stamps.py"""
import sys, os, ctypes as C, collections
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "image-processing-graph-laplacian_amd"))
import numpy as np, torch, glf
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = glf.Context(0)
d_img = ctx.to_device(glf.synth_image(W, W, seed=7))
opt = glf.default_options(num_samples=int(W * W * 0.005), num_eigvals=64)
if len(sys.argv) > 2: ctx.set_tuning(MV_PATH=sys.argv[2])
out, zf, info = ctx.image_processing(d_img, opt)
buf = (C.c_ulonglong * 2048)()
assert glf._lib.glf_debug_rank_stamps(buf) == 0
a = np.array(buf, dtype=np.uint64).reshape(8, 256)
for w in (0, 3, 7):
    tags = (a[w] >> np.uint64(56)).astype(int); t = (a[w] & np.uint64((1 << 56) - 1)).astype(np.int64)
    n = int((tags > 0).sum())
    print("wave", w, "stamps", n)
    d = collections.defaultdict(list)
    for i in range(1, n):
        d[(tags[i - 1], tags[i])].append(int(t[i] - t[i - 1]))
    for k in sorted(d): print("   %d->%d: n=%3d mean %8.0f  min %7d max %7d" % (k[0], k[1], len(d[k]), np.mean(d[k]), min(d[k]), max(d[k])))
    seq = " ".join("%d:%d" % (tags[i], t[i] - t[i - 1]) for i in range(1, min(n, 70)))
    print("   ", seq)
