"""Diagnostic (GPU box): per-tensor gradient error of libcbgx's backward at the configs[4] shape (B real-size graphs in one
batch) against autograd on the CPU oracle run on 4-graph sub-batches; both kernel generations, run-to-run spread, and a
column-block breakdown of the first Linears (type | rbf | h_dst | h_src columns of the [128, 340] weight).
usage: python scripts/grad_err_config_sized.py [B=32] [seed=404]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cbgbench_amd as C
from cbgbench_amd import synthetic, _native
from oracle import training as TR, weights as W
from test_gpu_config_sized import sub_batch

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 404
torch.set_num_threads(min(16, os.cpu_count() or 1))
dev = "cuda:0"
batch = synthetic.denovo_batch(B, seed=seed)
n_lig = batch["ligand_pos"].shape[0]
g = torch.Generator().manual_seed(7)
t = torch.randint(0, 1000, (B,), generator=g)
t[min(3, B - 1)] = 0
eps = torch.randn(n_lig, 3, generator=g)
u = torch.rand(n_lig, 13, generator=g)
sd = W.synthetic_state_dict(13, 9, seed=0)


def gpu_grads(impl):
    import contextlib
    with (_native.first_generation_kernels() if impl else contextlib.nullcontext()):
        m = C.get_model(C.default_targetdiff_config(13))
        m.load_state_dict(sd, strict=True)
        m = m.to(dev).train()
        ld, _ = m(synthetic.batch_to(batch, dev), t=t.to(dev), noise=(eps.to(dev), u.to(dev)))
        (1.0 * ld["pos"] + 100.0 * ld["atom"]).backward()
        torch.cuda.synchronize()
        return {k: p.grad.detach().cpu().double() for k, p in m.named_parameters() if p.requires_grad}


ref = None
for g0 in range(0, B, 4):
    sb, ml = sub_batch(batch, g0, min(B, g0 + 4))
    nb = min(B, g0 + 4) - g0
    _, grads = TR.loss_and_grads(sd, sb, t[g0:g0 + nb], eps[ml], u[ml], 13)
    w = nb / B
    if ref is None:
        ref = {k: w * v.double() for k, v in grads.items()}
    else:
        for k, v in grads.items():
            ref[k] += w * v.double()

runs = {"mfma": gpu_grads(0), "mfma_again": gpu_grads(0), "valu_v1": gpu_grads(1)}


def rel(a, b):
    n = float(b.norm())
    return float((a - b).norm()) / n if n > 1e-12 else 0.0


for name, gr in runs.items():
    rows = sorted(((rel(gr[k], ref[k]), k) for k in ref if float(ref[k].norm()) > 1e-9), reverse=True)
    over = sum(1 for r in rows if r[0] > 2e-4)
    print(f"== {name}: {over} of {len(rows)} tensors over 2e-4; median {rows[len(rows) // 2][0]:.2e}")
    for r in rows[:10]:
        print("   %.3e %s" % r)
rows = sorted(((rel(runs["mfma"][k], runs["mfma_again"][k]), k) for k in ref), reverse=True)
print("== run-to-run (atomics) spread, worst:", ["%.2e %s" % r for r in rows[:3]])
rows = sorted(((rel(runs["mfma"][k], runs["valu_v1"][k]), k) for k in ref), reverse=True)
print("== mfma vs valu_v1, worst:", ["%.2e %s" % r for r in rows[:3]])
for blk in (0, 4, 8):
    for lay, fn in (("h2x", "xk_func"), ("h2x", "xv_func"), ("x2h", "hk_func"), ("x2h", "hv_func")):
        k = f"denoiser.blocks.{blk}.{lay}_layers.0.{fn}.net.0.weight"
        a, r = runs["mfma"][k], ref[k]
        parts = {"type": slice(0, 4), "rbf": slice(4, 84), "h_dst": slice(84, 212), "h_src": slice(212, 340)}
        print(k, " ".join(f"{n}: {rel(a[:, s], r[:, s]):.2e} (|ref| {float(r[:, s].norm()):.2e})" for n, s in parts.items()))
