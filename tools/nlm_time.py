"""NLM affinity at 1024^2, m = 64: whole-path stage times (python tools/nlm_time.py [size])."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "image-processing-graph-laplacian_amd"))
import torch, glf
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ctx = glf.Context(0)
d = ctx.to_device(glf.synth_image(W, W, seed=7))
opt = glf.default_options(num_samples=int(W * W * 0.005), num_eigvals=64, epsilon=0.1, kernel=glf.KERNEL_NLM, h_val=3.0)
for _ in range(2):
    out, zf, info = ctx.image_processing(d, opt, want_float=True)
print({k: round(v, 2) for k, v in info.items() if k.startswith("ms_")}, info["outer_its"], info["p"])
