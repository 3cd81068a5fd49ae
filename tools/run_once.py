"""One whole-path run for profiling: python tools/run_once.py [W] [sample_frac] [m] [skip] [reps] [epsilon]."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "image-processing-graph-laplacian_amd"))
import torch, glf
W = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.005
m = int(sys.argv[3]) if len(sys.argv) > 3 else 64
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
eps = float(sys.argv[6]) if len(sys.argv) > 6 else None
ctx = glf.Context(0)
d_img = ctx.to_device(glf.synth_image(W, W, seed=7))
opt = glf.default_options(num_samples=int(W * W * frac), num_eigvals=m) if eps is None else glf.default_options(num_samples=int(W * W * frac), num_eigvals=m, epsilon=eps)
opt.skip_exact_zeros = skip
for _ in range(reps):
    out, zf, info = ctx.image_processing(d_img, opt)
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in info.items() if k.startswith("ms_") or k == "nystroem_kernel_ms"})
