"""Diagnostic (GPU box): wall time of the phases of one training step (configs[4] shape), synchronised at the phase boundaries, and
the same step without the synchronisations -- where the ~3 ms between the kernels' 23 ms and the step's 26 ms go."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import cbgbench_amd as C
from cbgbench_amd import synthetic, train as TRN
from torch.nn.utils import clip_grad_norm_

dev = "cuda:0"
model = bench.make_model(dev)
batch = synthetic.batch_to(synthetic.denovo_batch(32, seed=3000), dev); batch["num_graphs"] = 32
import types
opt = TRN.get_optimizer(types.SimpleNamespace(type="adam", lr=1e-4, weight_decay=0.0, beta1=0.95, beta2=0.999), model)
fg = TRN.FlatGradients(model)
model.train()
w = {"pos": 1.0, "atom": 100.0}
def sync(): torch.cuda.synchronize()
for _ in range(3): TRN.train_step(model, batch, opt, fg, w, 8.0)
sync()
acc = {}
for it in range(8):
    t = [time.perf_counter()]
    def mark(name):
        sync(); t.append(time.perf_counter()); acc[name] = acc.get(name, 0.0) + t[-1] - t[-2]
    fg.zero(); mark("zero")
    ld, _ = model(batch); mark("forward"); loss = TRN.sum_weighted_losses(ld, w); mark("loss sum")
    loss.backward(); mark("backward")
    fg.all_reduce_mean(); gn = fg.clip_norm_(8.0); mark("clip")
    opt.step(); mark("adam")
print({k: round(1e3 * v / 8, 3) for k, v in acc.items()}, "ms per phase, sum", round(1e3 * sum(acc.values()) / 8, 3))
sync(); t0 = time.perf_counter()
for _ in range(8): TRN.train_step(model, batch, opt, fg, w, 8.0)
sync(); print("unsynchronised step ms", round(1e3 * (time.perf_counter() - t0) / 8, 3))
