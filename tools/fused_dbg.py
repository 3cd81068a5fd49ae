import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "image-processing-graph-laplacian_amd"))
import numpy as np, torch, glf
ctx = glf.Context(0)
W = 2048
img = glf.synth_image(W, W, seed=0)
d = ctx.to_device(img)
opt = glf.default_options(num_samples=int(W * W * 0.005), num_eigvals=64, epsilon=0.1)
opt.gain = 2000.0
o1, z1, i1 = ctx.image_processing(d, opt, want_float=True)
ctx.set_tuning(NO_FUSED_FILTER="1")
o2, z2, i2 = ctx.image_processing(d, opt, want_float=True)
print(i1["filter_fused"], i2["filter_fused"])
y = d.float()
c1 = (z1 - y).double(); c2 = (z2 - y).double()
dz = (c1 - c2).abs()
k = int(dz.argmax()); r, c = divmod(k, W)
print("max|dz|", float(dz.max()), "at", r, c, "corr there", float(c1.view(-1)[k]), float(c2.view(-1)[k]), "pixel", int(img[r, c]))
print("rel l2 of corrections", float((c1 - c2).norm() / c2.norm()), "rms corr", float(c2.pow(2).mean().sqrt()), "max corr", float(c2.abs().max()))
idx = glf.Sampling(W, W, int(W * W * 0.005))
mask = np.zeros(W * W, bool); mask[idx] = True
m = torch.from_numpy(mask).to(dz.device).view(W, W)
print("max at samples", float(dz[m].max()), "at non-samples", float(dz[~m].max()))
print("quantiles", [float(q) for q in torch.quantile(dz.view(-1)[::7].float(), torch.tensor([0.5, 0.9, 0.99, 0.999, 0.9999], device=dz.device))])
