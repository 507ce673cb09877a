import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from bench_configs import make_priors
import vortex_amd as V
from vortex_amd import _lib as L, ops
dev = torch.device("cuda:0")
B, Lm, NL, sthr, P = 128, 5, 2, 0.5, 50400
priors = make_priors(640).to(dev)
g = torch.Generator(device=dev).manual_seed(5)
x = torch.randn(B, P, 4 + 2 * Lm + NL, generator=g, device=dev)
x[..., 4 + 2 * Lm + 1:] = x[..., 4 + 2 * Lm + 1:] * 2 - 3
img = torch.randint(0, 256, (B, 640, 640, 3), generator=g, dtype=torch.uint8, device=dev)
norm = V.Normalize([0.5] * 3, [0.5] * 3).to(dev)
pp = V.PriorBoxPostProcess(priors, n_landmarks=Lm).to(dev)
st, it = torch.tensor([sthr]), torch.tensor([0.4])
out = torch.empty((B, 3, 640, 640), device=dev)
def run(name, fn, n=12):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    L.profile_enable(64)
    for _ in range(n): fn()
    torch.cuda.synchronize()
    r = L.profile_read()
    print(name, {k: round(v["avg_ms"], 4) for k, v in r.items()}, flush=True)
    L.profile_disable()
run("norm only", lambda: norm(img))
run("norm + decode", lambda: (norm(img), ops.prior_decode(x, priors, Lm, sthr)))
run("norm + full post", lambda: (norm(img), pp(x, st, it)))
keep = []
def hold():
    keep.append(norm(img))
    if len(keep) > 1: keep.pop(0)
    return pp(x, st, it)
run("norm (one output kept alive) + full post", hold)
imgs = [torch.randint(0, 256, (B, 640, 640, 3), generator=g, dtype=torch.uint8, device=dev) for _ in range(4)]
k = [0]
def rot():
    k[0] = (k[0] + 1) % 4
    return norm(imgs[k[0]])
run("norm only, 4 rotating inputs (628 MB)", rot)
y = torch.empty(100_000_000, device=dev)
run("norm + 400 MB fill", lambda: (norm(img), y.fill_(1.0)))
run("norm + 413 MB read (sum)", lambda: (norm(img), x.sum()))
img416 = torch.randint(0, 256, (256, 416, 416, 3), generator=g, dtype=torch.uint8, device=dev)
run("norm 416 b256 only", lambda: norm(img416))
run("norm 416 b256 + 413 MB read", lambda: (norm(img416), x.sum()))
