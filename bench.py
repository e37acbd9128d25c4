#!/usr/bin/env python
"""bench.py -- denoising graph-steps/s of the TargetDiff reverse-diffusion hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--pockets P] [--samples S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload = the whole BASELINE.json configs[1] job ("configs/denovo, 100 pockets x 10 samples each, 1000 steps, fp32,
1 MI355X"): 100 distinct pockets x 10 samples of each (sample.py:177 replicates one pocket num_samples times) = 1000
pocket+ligand graphs (~5.3e5 nodes, 1.7e7 edges) resident in HBM as --graphs-per-batch sized batches (default 340
graphs = 34 pockets x 10 samples: three batches, all in flight on three streams; independent pockets are batched together because
one pocket's 10 graphs cannot fill 256 CUs; round 4 sweep: 200 / 250 / 340 / 500 graphs per batch = 28.26 / 28.24 / 28.49 / 27.80 k
graph-steps/s, profiles/sweep_r04e.log).  A reverse-diffusion step of a batch = ligand embedding + pocket/ligand composition + the 9-layer equivariant
denoiser in libcbgx + position/type posterior sampling + trajectory store.  One bench "step" advances the WHOLE job
(all batches) by one reverse-diffusion step at each of the five time blocks t = 999-i, 749-i, 499-i, 249-i, 24-(i mod 25)
(the network has no time input in the shipped configs; t selects the posterior coefficients and the noise scale, down
to the noise-free t = 0 step), i.e. 5 denoising steps x 1000 graphs = 5000 graph-steps per bench step; `value` =
graph-steps / wall time.  Synthetic pockets (cbgbench_amd/synthetic.py), deterministic synthetic weights
(cbgbench_amd/synthetic_weights.py).

`--gpus N` with N > 1 and no torchrun environment re-executes itself under `python -m torch.distributed.run` with N
ranks on 127.0.0.1; under torchrun it checks WORLD_SIZE == N.  Every rank owns its own 1000-graph job (weak scaling,
no data-path collective; SURVEY.md 8e = BASELINE configs[3]: 1000 pockets per GPU); the timed region is bracketed by
barrier + synchronize, the MAX over ranks is reported and `ranks_seen` comes from an RCCL all-reduce.

Prints ONE JSON line on rank 0 (contract in the task statement) incl. `roofline` (dominant kernel =
the fused x2h edge kernel, timed live with HIP events on its own stream via cbgx_profile_*) and
`cpu_baseline` (the CPU oracle = port of the reference's PyTorch-CPU path, timed on this host: one process at its best thread
count, and `cores / threads` such processes side by side).

At N = 1 the default run also measures a `secondary` block in the same process, a few seconds each, so that every BASELINE
config is under the driver's eyes and not only in builder-side files (--no-secondary skips it): the linker batch of 256
(configs[2]), the training step at 32 graphs per GPU with its own roofline (configs[4] shape), the diffbp and diffsbdd samplers,
the reference's own 10-graph batch (sample.py:177-183) and a single graph, and ONE END-TO-END run of
`cbgbench_amd.sample_cli` at T = 1000 on 20 pockets x 10 samples (wall time including prior construction, the static-context
cache, the trajectory download and the per-pocket result files), with its ratio to the step-sampled headline number.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import cbgbench_amd as C  # noqa: E402
from cbgbench_amd import _native, sharding, synthetic  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32-in MFMA = f32 vector peak
# Algorithmic bytes of the message-passing stage at the reference's tensor boundary:
# X2H: 1032 B per edge (k 512 + v 512 + e_w 4 + nbr 4) + 1536 B per node (q, h residual, out)
# H2X:  596 B per edge (k 512 + v 64 + x_j 12 + e_w 4 + nbr 4) + 536 B per node
X2H_BYTES_PER_EDGE, X2H_BYTES_PER_NODE = 1032, 1536
H2X_BYTES_PER_EDGE, H2X_BYTES_PER_NODE = 596, 536
# Factored algorithmic FLOPs: 122 880 per edge-layer + 8*32 768 + 131 072 per node-layer
FLOPS_PER_EDGE_LAYER, FLOPS_PER_NODE_LAYER = 122880, 8 * 32768 + 131072
# what the fused x2h edge kernel actually issues on the matrix cores: per edge the scores against the folded
# query (2*128*16) and the value aggregation (2*16*128) as exact-fp32 MFMAs; the rbf columns of both first Linears
# (2*20*256 useful flop per edge) as split-f16 MFMAs: 128 v_mfma_f32_16x16x16_f16 of 8192 flop per 32-edge node; per node the
# value's second Linear applied after aggregation (2*128*128, packed fp32 VALU).  Both second Linears have left the edge.
X2H_FP32_MFMA_FLOPS_PER_EDGE = 2 * 128 * 16 + 2 * 16 * 128
X2H_F16_MFMA_FLOPS_PER_NODE = 128 * 2 * 16 * 16 * 16
X2H_MFMA_ISSUE_CYCLES_PER_NODE = 128 * 17 + 128 * 32     # issue cycles of a node's MFMAs on its SIMD (roofline.issue_frac)
N_SIMDS, SIMD_CLOCK_HZ = 256 * 4, 2.4e9
N_CUS = 256
X2H_BWD_ATOMICS_PER_NODE = 136      # 64-lane atomic instructions of the x2h edge backward per node (train_bwd_x2h.hip, pass 4 + coordinates)
ATOMIC_INSTR_S = 60e-9              # one of them on a CU's atomic path (scripts/ubench/vmem.hip shapes 5, 8 - 11: 56 - 66 ns)
X2H_EXEC_FLOPS_PER_EDGE, X2H_EXEC_FLOPS_PER_NODE = 2 * 20 * 256 + 2 * 128 * 16 + 2 * 16 * 128, 2 * 128 * 128
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense f16 / bf16 matrix peak


def measured_traffic(n_nodes):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, profiles/traffic_x2h.json), scaled per node; None if absent."""
    path = os.path.join(ROOT, "profiles", "traffic_x2h.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        t = json.load(f)
    return int(round((t["fetch_bytes_per_node"] + t["write_bytes_per_node"]) * n_nodes))


TRAFFIC_SOURCE = "committed constant: profiles/traffic_x2h.json (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; separate passes) x nodes"


def build_batch(pockets, samples, seed, num_classes=13):
    """P distinct pockets, each replicated S times with fresh ligand priors (sample.py:177-183)."""
    rng = np.random.default_rng(seed)
    pk = [synthetic.make_pocket(rng, int(rng.integers(350, 651))) for _ in range(pockets)]
    plist, nlig = [], []
    for p in pk:
        for _ in range(samples):
            plist.append(p)
            nlig.append(int(rng.integers(10, 46)))
    return synthetic.make_batch(plist, nlig, rng, num_classes)


def make_model(device, T=1000, name="targetdiff"):
    """the three diffusion model classes of the reference share the denoiser (repo/models/diffusion/{targetdiff,diffbp,diffsbdd}.py)"""
    from cbgbench_amd import synthetic_weights
    cfg = {"targetdiff": lambda: C.default_targetdiff_config(13, 9, T),
           "diffbp": lambda: C.default_diffbp_config(13, num_diffusion_timesteps=T),
           "diffsbdd": lambda: C.default_diffsbdd_config(8, num_diffusion_timesteps=T)}[name]()
    model = C.get_model(cfg).eval()
    synthetic_weights.fill_(model, seed=0)      # deterministic random-init weights (no checkpoints ship with the reference)
    return model.to(device)


def oracle_state_dict(T=1000):
    """cpu_baseline leg only: the same synthetic weights in the oracle's state-dict form"""
    from oracle import weights
    return weights.synthetic_state_dict(13, 9, seed=0, num_timesteps=T)


_CPU_THREADS = {}


def _pick_cpu_threads(sd, seed):
    """The CPU port is PyTorch-CPU; its best thread count is well below the core count on big hosts
    (256 threads on this path is >30x slower than 16).  Calibrate on one single-graph denoiser call (once per process)."""
    if "best" in _CPU_THREADS:
        return _CPU_THREADS["best"]
    from oracle import unitransformer as OU
    batch = build_batch(1, 1, seed)
    n = batch["protein_pos"].shape[0] + batch["ligand_pos"].shape[0]
    x = torch.cat([batch["protein_pos"], batch["ligand_pos"]])
    h = torch.zeros(n, 128)
    bi = torch.zeros(n, dtype=torch.long)
    lig = torch.cat([batch["protein_lig_flag"], batch["ligand_lig_flag"]])
    best, best_t = None, 1
    ncpu = os.cpu_count() or 1
    for th in [t for t in (4, 8, 16, 32, 64) if t <= ncpu] or [ncpu]:
        torch.set_num_threads(th)
        with torch.no_grad():
            OU.unitransformer_forward(sd, x, h, bi, lig, lig)            # warm
            t0 = time.perf_counter()
            OU.unitransformer_forward(sd, x, h, bi, lig, lig)
            dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, best_t = dt, th
    _CPU_THREADS["best"] = best_t
    return best_t


def cpu_baseline(sd, seed, max_seconds=20.0, model="targetdiff", threads=None):
    """The CPU oracle (oracle/{targetdiff,diffbp,diffsbdd}.py: ports of the reference's PyTorch-CPU step, reference
    formulation with materialised [E,340] edge inputs) on this host's cores, on a bounded sample of the same workload:
    whole steps of one 10-graph batch (1 pocket x 10 samples) until ~max_seconds."""
    from oracle import targetdiff as OT
    if threads is None:
        threads = _pick_cpu_threads(sd if model == "targetdiff" else oracle_state_dict(), seed)
    torch.set_num_threads(threads)
    C_ = 8 if model == "diffsbdd" else 13
    batch = build_batch(1, 10, seed, num_classes=C_)
    x = batch["ligand_pos"]
    c = torch.nn.functional.one_hot(batch["ligand_atom_type"], C_).float()
    g = torch.Generator().manual_seed(seed)
    n_lig = x.shape[0]
    if model == "diffbp":
        from oracle import diffbp as OB
        step = lambda x, c, k: OB.denoise_step(sd, batch, x, c, 999 - k, torch.randn(n_lig, 3, generator=g),
                                               torch.rand(n_lig, generator=g), 13, 1000)
    elif model == "diffsbdd":
        from oracle import diffsbdd as OS
        gamma_tab = sd["pos_scheduler.gamma.gamma"]
        bl, br = batch["ligand_element_batch"], batch["protein_element_batch"]
        B = int(bl.max()) + 1
        v_rec = batch["protein_atom_feature"] / 4.0
        pocket = {"x_rec": batch["protein_pos"]}

        def step(x, c, k):     # one iteration of the loop at diffsbdd.py:296-304 (oracle/diffsbdd.py::sample)
            s_, t_ = torch.full((B,), 999 - k) / 1000, (torch.full((B,), 999 - k) + 1) / 1000
            x_pred, c_out = OS.denoise(sd, batch, x, c, pocket["x_rec"], v_rec)
            x, pocket["x_rec"] = OS.sample_p_zs_given_zt(gamma_tab, 1000, s_, t_, x, pocket["x_rec"], bl, br, B, x_pred,
                                                         torch.randn(n_lig, 3, generator=g), True)
            c, _ = OS.sample_p_zs_given_zt(gamma_tab, 1000, s_, t_, c, v_rec, bl, br, B, c_out,
                                           torch.randn(n_lig, C_, generator=g), False)
            return x, c
    else:
        step = lambda x, c, k: OT.denoise_step(sd, batch, x, c, 999 - k, torch.randn(n_lig, 3, generator=g),
                                               torch.rand(n_lig, 13, generator=g), 13)
    steps, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while True:
            x, c = step(x, c, steps)
            steps += 1
            el = time.perf_counter() - t0
            if el > max_seconds or steps >= 8:
                break
    return {"value": round(10 * steps / el, 4), "unit": "graph-steps/s", "cores": threads,
            "kind": "port", "sample": f"{steps} denoising steps of one 10-graph batch (N={batch['protein_pos'].shape[0] + n_lig} nodes, "
            f"sample.py:177-183), oracle/{model}.py on PyTorch-CPU fp32, {threads} threads (best of 4..64 of {os.cpu_count()} "
            f"cores), {el:.1f} s"}


def cpu_worker(argv):
    """`bench.py --cpu-worker THREADS SECONDS MODEL`: one process of the concurrent CPU leg; prints its graph-steps/s."""
    threads, seconds, model = int(argv[0]), float(argv[1]), argv[2]
    from oracle import weights as OW
    osd = {"targetdiff": lambda: oracle_state_dict(), "diffbp": lambda: OW.synthetic_state_dict_diffbp(13, 9, seed=0, num_timesteps=1000),
           "diffsbdd": lambda: OW.synthetic_state_dict_diffsbdd(8, 9, seed=0, num_timesteps=1000)}[model]()
    r = cpu_baseline(osd, seed=1000, max_seconds=seconds, model=model, threads=threads)
    print(json.dumps({"value": r["value"]}), flush=True)


def cpu_baseline_concurrent(threads, model="targetdiff", seconds=12.0):
    """The same CPU step as cpu_baseline in `cores // threads` processes side by side (each on its own copy of the 10-graph
    batch, `threads` threads): what this host's cores deliver when they are all used, the fair same-host figure next to one
    GPU.  A single PyTorch-CPU process cannot use more than ~16 threads on this path (256 threads are > 30x slower)."""
    import subprocess
    ncpu = os.cpu_count() or 1
    procs = max(1, ncpu // threads)
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
    t0 = time.perf_counter()
    ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(threads), str(seconds), model],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env, cwd=ROOT) for _ in range(procs)]
    vals = []
    for p in ps:
        try:
            out, _ = p.communicate(timeout=seconds * 6 + 180)
            vals.append(json.loads([l for l in out.splitlines() if l.startswith("{")][-1])["value"])
        except Exception:      # a worker that failed or timed out contributes nothing (and is said so)
            p.kill()
    return {"value": round(float(sum(vals)), 4), "unit": "graph-steps/s", "processes": procs, "processes_reported": len(vals),
            "threads_each": threads, "cores": procs * threads,
            "sample": f"{procs} processes x {threads} threads side by side, ~{seconds:.0f} s each, rates summed"}


# backward of the message-passing stage at the reference's tensor boundary (autograd of x2h_attention.py:80-97):
# per edge read k, v (512 B each), e_w, index and write dk, dv (512 B each); per node read q, dL/dh_out and write dq, dL/dh
X2H_BWD_BYTES_PER_EDGE = 4 * 512 + 4 + 4
X2H_BWD_BYTES_PER_NODE = 4 * 512


def cpu_train_baseline(sd, seed, max_seconds=25.0):
    """The CPU oracle's training step (oracle/training.py: reference formulation, torch.autograd backward) on a
    bounded sample: 4-graph batches, forward + backward (no optimiser), at its best thread count."""
    from oracle import training as TR
    threads = _pick_cpu_threads(sd, seed)
    torch.set_num_threads(threads)
    batch = build_batch(4, 1, seed)
    g = torch.Generator().manual_seed(seed)
    n_lig = batch["ligand_pos"].shape[0]
    steps, t0 = 0, time.perf_counter()
    while True:
        t = torch.randint(0, 1000, (4,), generator=g)
        TR.loss_and_grads(sd, batch, t, torch.randn(n_lig, 3, generator=g), torch.rand(n_lig, 13, generator=g), 13)
        steps += 1
        el = time.perf_counter() - t0
        if el > max_seconds or steps >= 4:
            break
    return {"value": round(4 * steps / el, 4), "unit": "graph-steps/s", "cores": threads, "kind": "port",
            "sample": f"{steps} forward+backward passes of one 4-graph batch, oracle/training.py (torch.autograd, PyTorch-CPU fp32), "
                      f"{threads} threads, {el:.1f} s"}


def bench_train(args, rank, world, dev):
    """BASELINE configs[4] shape on the GPUs at hand: train.py semantics (forward, backward, gradient all-reduce, clip,
    Adam) with `--pockets` graphs per GPU per step (default 32).  `--model diffbp|diffsbdd` trains those classes."""
    from cbgbench_amd import train as TRN
    model = make_model(dev, name=args.model)
    model.train()
    TRN.broadcast_parameters(model)
    fg = TRN.FlatGradients(model)
    # configs/denovo/train/targetdiff.yml:42-47 through the product's own factory (train.get_optimizer: FlatAdam on device parameters)
    import types
    opt = TRN.get_optimizer(types.SimpleNamespace(type="adam", lr=5e-4, weight_decay=0.0, beta1=0.95, beta2=0.999), model)
    # loss weights of configs/denovo/train/{targetdiff,diffbp,diffsbdd}.yml
    weights = {"targetdiff": {"pos": 1.0, "atom": 100.0}, "diffbp": None, "diffsbdd": None}[args.model]
    n_graphs = args.pockets
    batch = synthetic.batch_to(build_batch(n_graphs, 1, seed=3000 + rank, num_classes=model.num_classes), dev)
    batch["num_graphs"] = n_graphs      # what train_cli's collate records: the model then needs no device round trip for it
    batch["max_ligand_atoms"] = int(torch.bincount(batch["ligand_element_batch"]).max())     # (host-side knowledge of the collate too)
    N = batch["protein_pos"].shape[0] + batch["ligand_pos"].shape[0]
    torch.manual_seed(2022 + rank)
    t_ar = 0.0
    for _ in range(args.warmup):
        TRN.train_step(model, batch, opt, fg, weights, 8.0)
    sharding.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        t_ar += TRN.train_step(model, batch, opt, fg, weights, 8.0)[3]
    torch.cuda.synchronize(); sharding.barrier()
    elapsed = time.perf_counter() - t0
    el_max, units = sharding.reduce_max_sum(elapsed, n_graphs * args.steps, device=dev)
    per_rank = sharding.gather_values(round(n_graphs * args.steps / elapsed, 1), device=dev)
    seen = ranks_seen(dev)
    if seen != world:
        raise SystemExit(f"bench.py: all-reduce saw {seen} ranks, expected {world}")
    out = {
        "metric": "training graph-steps/s (pocket+ligand graphs x optimiser steps per second: forward + backward + "
                  "gradient all-reduce + clip + Adam)",
        "value": round(units / el_max, 2), "unit": "graph-steps/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * el_max / args.steps, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"configs/denovo {args.model} training (BASELINE configs[4] shape): {n_graphs} graphs per GPU per step, "
                               f"Adam lr 5e-4, clip 8.0, one flat gradient all-reduce ({fg.flat.numel()} fp32) per step",
                   "graphs_per_batch_per_gpu": n_graphs, "nodes_per_batch": N, "sharding": f"data-parallel x{world} ranks", "ranks_seen": seen,
                   "collective_backend": collective_backend(), "per_rank_graph_steps_per_s": per_rank,
                   "allreduce_ms_per_step": round(1e3 * t_ar / max(args.steps, 1), 4)},
    }
    if rank == 0 and not args.no_roofline:
        lib = _native.lib()
        names = _native.PROFILE_CLASSES
        prof_steps = min(args.steps, 3)
        _native.check(lib.cbgx_profile_begin(400 * prof_steps + 64), "cbgx_profile_begin")
        for _ in range(prof_steps):
            TRN.train_step(model, batch, opt, fg, weights, 8.0)
        NCLS = len(names)
        ms = (ctypes.c_double * NCLS)(); cnt = (ctypes.c_int * NCLS)()
        _native.check(lib.cbgx_profile_end(ms, cnt, NCLS), "cbgx_profile_end")
        per = {n: [round(1e3 * ms[i] / max(cnt[i], 1), 1), cnt[i]] for i, n in enumerate(names) if cnt[i]}
        k = names.index("edge_x2h_bwd")
        bwd_bytes = X2H_BWD_BYTES_PER_EDGE * 32 * N + X2H_BWD_BYTES_PER_NODE * N
        bwd_s = 1e-3 * ms[k] / max(cnt[k], 1)
        achieved = bwd_bytes / bwd_s / 1e9 if bwd_s > 0 else 0.0
        out["roofline"] = {
            # algorithmic bytes = backward of the message-passing stage at the reference tensor boundary (2056 B/edge + 2048 B/node);
            # the kernel recomputes the per-edge forward instead of reading it
            "bound": "hbm", "kernel": "cbgx::edge_backward_x2h_kernel (backward of the x2h block)",
            "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": None, "algorithmic_bytes_per_launch": bwd_bytes, "avg_launch_us": round(1e6 * bwd_s, 3),
            # what bounds this kernel (round 6, profiles/abl_bwd_r06c.log + profiles/ubench_vmem_r06d.log): the compute unit's fp32
            # atomic path -- a 64-lane global_atomic_add_f32 occupies it for ~60 ns whatever its shape (a dword store 5.5 ns), a node issues
            # 136 of them (neighbour rows of d P: 32 edges x 256 columns) -- its busy time per launch / the launch time
            "atomic_path_frac": round(N * X2H_BWD_ATOMICS_PER_NODE * ATOMIC_INSTR_S / N_CUS / bwd_s, 4) if bwd_s > 0 else None,
            "per_kernel_us_avg_and_launches": per,
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.model == "targetdiff":
        out["cpu_baseline"] = cpu_train_baseline(oracle_state_dict(), seed=3000)
    return out


T_BLOCKS = (999, 749, 499, 249, 24)     # first t of each time block; bench step i runs t = block - i (last block: i mod 25)


def block_times(i, T=1000):
    """the reverse-diffusion times bench step i visits, one per time block (scaled for models with T != 1000)"""
    out = []
    for b in T_BLOCKS:
        b = min(b * T // 1000, T - 1)
        out.append(max(b - (i % 25 if b < 25 else i % 225), 0))
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` outside torchrun: re-execute as N ranks of this node through cbgbench_amd.launch -- a FileStore
    rendezvous, so no port is picked here for somebody else to take before the ranks bind it; a failing rank's stderr is printed"""
    from cbgbench_amd import launch
    return launch.launch(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])


def ranks_seen(dev):
    """number of ranks that took part, from a real all-reduce on the job's backend (RCCL when one GPU per rank)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return 1
    one = torch.ones(1, dtype=torch.float32, device="cpu" if dist.get_backend() == "gloo" else dev)
    dist.all_reduce(one)
    return int(one.item())


def collective_backend():
    import torch.distributed as dist
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def device_identity(dev):
    """A string that is equal for two ranks exactly when they drive the same physical GPU of this job: host name + the device
    visibility environment + the device index under it (one node, torchrun: indices are node-global; per-rank *_VISIBLE_DEVICES
    settings differ in the environment part).  The hardware uuid / PCI address is appended for the record when torch exposes it, but
    is not what decides -- on some ROCm builds those fields are empty or identical."""
    import socket
    vis = "|".join(os.environ.get(k, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
    return f"{socket.gethostname()}|{vis}|{dev.index}"


def device_hardware_id(dev):
    pr = torch.cuda.get_device_properties(dev)
    return {a: str(getattr(pr, a)) for a in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id") if hasattr(pr, a)}


def usable_uuid(hw):
    """the hardware uuid when torch exposes a real one (some ROCm builds leave it empty or all zeros)"""
    u = hw.get("uuid", "")
    return u if u and u.strip("0-") else None


def distinct_devices(dev, world):
    """(n_env, n_uuid): the number of distinct physical GPUs over all ranks by device_identity (environment + index: what decides)
    and by hardware uuid (None unless every rank reports a usable one).  The uuid count is a CROSS-CHECK that is reported, not
    enforced: some ROCm builds return the same uuid for every GPU, which must not stop a correct 8-GPU launch -- RCCL itself refuses
    two ranks on one device."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        u = usable_uuid(device_hardware_id(dev))
        return 1, (1 if u else None)
    ids = [None] * world
    dist.all_gather_object(ids, (device_identity(dev), device_hardware_id(dev)))
    n_env = len(set(i for i, _ in ids))
    uuids = [usable_uuid(hw) for _, hw in ids]
    n_uuid = len(set(uuids)) if all(u is not None for u in uuids) else None
    return n_env, n_uuid


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pockets", type=int, default=None,
                    help="distinct pockets of the job per GPU (default 100 -> 1000 graphs; linker: 256 graphs; train: 32 graphs)")
    ap.add_argument("--samples", type=int, default=10, help="samples (graphs) per pocket")
    ap.add_argument("--graphs-per-batch", type=int, default=340,
                    help="graphs per resident batch (whole pockets; default 340 = 34 pockets x 10 samples, ~170 k nodes: the "
                         "1000-graph job as three batches)")
    ap.add_argument("--workload", choices=["denovo", "linker", "train"], default="denovo",
                    help="denovo = BASELINE configs[1] (default); linker = configs[2]: --pockets distinct pockets, one "
                         "graph each, fixed context atoms + a few generated linker atoms (partial gen_flag); train = "
                         "configs[4] shape: forward + backward + all-reduce + Adam on --pockets graphs per GPU")
    ap.add_argument("--model", choices=["targetdiff", "diffbp", "diffsbdd"], default="targetdiff",
                    help="model class timed (default: targetdiff, the driver line). Sampling: diffbp adds the CoMPredictor H2X "
                         "stack + score / mask-type step per step, diffsbdd the zero-COM variational step (its pocket moves "
                         "every step: no static-context cache); --workload train trains the class")
    ap.add_argument("--graph", choices=["on", "off"], default="off",
                    help="replay one captured hipGraph per batch and denoising step instead of stream launches.  No gain for one "
                         "batch (a small batch is bound by its dependent-kernel chain on the device); with many small batches "
                         "in flight (--streams 8) it removes the host as the limit")
    ap.add_argument("--streams", type=int, default=3,
                    help="resident batches in flight at once, round-robin over this many HIP streams (sampling workloads; default 3, "
                         "capped by the number of resident batches): while one batch sits in its matrix-bound edge kernel the node "
                         "kernels and small kernels of the others fill what it leaves (+6 % at three 200-graph batches)")
    ap.add_argument("--edge-workgroups", type=int, default=0,
                    help="cbgx_set_edge_workgroups: CUs the persistent x2h edge kernel may take (default 0 = all; leaving CUs free "
                         "for the other streams measured slower at every setting, profiles/README.md)")
    ap.add_argument("--split-job", action="store_true",
                    help="strong scaling: ONE --pockets x --samples job sharded over the ranks (rank r takes pockets r, r + W, ...: "
                         "what the pocket loop of sample.py:159 gives a user who adds GPUs) instead of one such job per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary block (other configs measured in the same run; N = 1, default workload only)")
    return ap.parse_args(argv)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-worker":
        return cpu_worker(sys.argv[2:])
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))
    if sharding.env_rank_world()[1] != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {sharding.env_rank_world()[1]} rank(s) "
                         f"(WORLD_SIZE); refusing to report a mislabelled n_gpus")
    rank, world, local = sharding.init_process_group()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the hot path)")
    gloo_shared = os.environ.get("CBGX_DIST_BACKEND") == "gloo"
    # one process per GPU; LOCAL_RANK -> device.  Only with CBGX_DIST_BACKEND=gloo may ranks share a device (the
    # multi-rank path exercised on a 1-GPU box); RCCL needs one GPU per rank.
    if local >= torch.cuda.device_count() and not gloo_shared:
        raise SystemExit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local}, {torch.cuda.device_count()} visible)")
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:      # the host-side job construction of N ranks shares the host's cores
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    # n_gpus must mean physical GPUs: every rank names its device, and a launch whose ranks share GPUs is refused -- unless it is
    # the declared gloo dry run of the multi-rank path on one GPU, which then reports the devices it really used next to `ranks`
    n_dev, n_uuid = distinct_devices(dev, world)
    if n_dev != world and not gloo_shared:
        raise SystemExit(f"bench.py: {world} ranks on {n_dev} distinct GPU(s); one GPU per rank is required "
                         f"(CBGX_DIST_BACKEND=gloo declares a shared-GPU dry run)")
    if args.pockets is None:
        args.pockets = {"train": 32, "linker": 256}.get(args.workload, 100)
    primary_default = (args.workload == "denovo" and args.model == "targetdiff" and args.pockets == 100 and args.samples == 10
                       and args.graphs_per_batch == 340 and args.graph == "off")
    out = bench_train(args, rank, world, dev) if args.workload == "train" else bench_sampling(args, rank, world, dev)
    out["n_gpus"] = n_dev
    if n_uuid is not None:
        out["config"]["devices_by_uuid"] = n_uuid      # hardware cross-check of n_gpus (informative: see distinct_devices)
        if n_uuid != n_dev:
            out["config"]["note_uuid"] = f"device uuids name {n_uuid} distinct GPU(s), the launch environment {n_dev}"
    if n_dev != world:
        out["ranks"] = world
        out["config"]["note_shared_gpus"] = f"{world} ranks shared {n_dev} GPU(s) over gloo: a dry run of the multi-rank path, not a scaling point"
    if rank == 0 and world == 1 and primary_default and not args.no_secondary:
        out["secondary"] = secondary_block(args, dev, out)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if collective_backend() is not None:
        sharding.barrier()      # rank 0 may still be in its (untimed) roofline pass: leave together
        torch.distributed.destroy_process_group()


def bench_sampling(args, rank, world, dev):
    """the sampling line: `--pockets` x `--samples` graphs per GPU resident as whole-pocket batches, one bench step = one
    reverse-diffusion step of the whole job at each of the five time blocks.  Returns the JSON line as a dict."""
    model = make_model(dev, name=args.model)
    T = model.num_diffusion_timesteps
    num_classes = model.num_classes

    # the job of this rank, resident in HBM as whole-pocket batches
    if args.workload == "linker":
        args.samples = 1
    ppb = max(1, args.graphs_per_batch // args.samples)          # pockets per batch
    split = bool(getattr(args, "split_job", False)) and world > 1
    # --split-job: the job's pockets 0 .. P-1 are dealt round-robin (cbgbench_amd/sharding.py, the rule sample_cli uses); this rank
    # builds and runs only its own
    my_pockets = len(sharding.shard_indices(args.pockets, rank, world)) if split else args.pockets
    chunks = [min(ppb, my_pockets - s) for s in range(0, my_pockets, ppb)]
    states = []
    for b, npk in enumerate(chunks):
        seed = 1000 + 97 * rank + 7919 * b
        batch = (synthetic.linker_batch(npk, seed=seed) if args.workload == "linker"
                 else build_batch(npk, args.samples, seed=seed, num_classes=num_classes))
        states.append(model.begin_sampling(synthetic.batch_to(batch, dev), keep_trajectory=True))
    n_graphs = my_pockets * args.samples
    N = sum(st["N"] for st in states)
    torch.manual_seed(2024 + rank)   # sample.py:106 seed (+rank: independent streams per shard)
    n_blocks = len(T_BLOCKS)

    n_streams = max(1, min(args.streams, len(states)))
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)] if n_streams > 1 else None
    edge_wgs = args.edge_workgroups
    _native.lib().cbgx_set_edge_workgroups(edge_wgs)
    try:      # the workgroup limit is process-wide: reset it on every way out
        if streams:
            for sx in streams:
                sx.wait_stream(torch.cuda.current_stream(dev))       # the states were built on the current stream

        def bench_step(i):
            for t in block_times(i, T):
                for b, st in enumerate(states):
                    if streams:      # batch b lives on stream b mod S: its steps stay ordered, different batches overlap
                        with torch.cuda.stream(streams[b % n_streams]):
                            model.denoise_step(st, t)
                    else:
                        model.denoise_step(st, t)

        use_graph = args.graph == "on" and args.warmup + args.steps + 2 < T and args.model == "targetdiff"
        if use_graph:
            # every resident batch's step captured once as a hipGraph on its stream, replayed once per bench step: the steps of a
            # batch follow each other (t = T-3, T-4, ...) instead of visiting the five time blocks
            n_blocks = 1
            made = [model.make_step_graph(st, warmup=2, stream=streams[b % n_streams] if streams else None)
                    for b, st in enumerate(states)]

            def replay_all():
                for b, (replay, _) in enumerate(made):
                    if streams:
                        with torch.cuda.stream(streams[b % n_streams]):
                            replay()
                    else:
                        replay()

            for _ in range(args.warmup):
                replay_all()
            sharding.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                replay_all()
            torch.cuda.synchronize(); sharding.barrier()
            elapsed = time.perf_counter() - t0
            t_idx = T - 1 - made[0][1] - args.warmup - args.steps
            st = states[0]
            st["x_lig"], st["c_lig"] = st["traj_x"][t_idx + 1].clone(), st["traj_c"][t_idx + 1].clone()
        else:
            for i in range(args.warmup):
                bench_step(i)
            sharding.barrier(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.warmup, args.warmup + args.steps):
                bench_step(i)
            torch.cuda.synchronize(); sharding.barrier()
            elapsed = time.perf_counter() - t0
        el_max, graph_steps = sharding.reduce_max_sum(elapsed, n_graphs * n_blocks * args.steps, device=dev)
        per_rank = sharding.gather_values(round(n_graphs * n_blocks * args.steps / elapsed, 1), device=dev)   # a straggler shows here
        seen = ranks_seen(dev)
        if seen != world:
            raise SystemExit(f"bench.py: all-reduce saw {seen} ranks, expected {world}")

        shape = (f"{my_pockets} pockets x {args.samples} samples = {n_graphs} graphs per GPU, {len(states)} resident batch(es) of <= "
                 f"{ppb * args.samples}")
        # the line must fit the driver's 8 KB tail: every description is one short string, the long-form text is DESIGN.md section 6
        out = {
            "metric": "denoising graph-steps/s (pocket+ligand graphs x reverse-diffusion steps per second)",
            "value": round(graph_steps / el_max, 2), "unit": "graph-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * el_max / args.steps, 4), "higher_is_better": True,
            "scaling": "strong" if split else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"configs/denovo {args.model} sampling, BASELINE configs[1] job: {shape}; N_rec~U[350,650], "
                                    f"N_lig~U[10,45], k=32, 9 layers; bench step = 1 reverse-diffusion step of the job at each of "
                                    f"{n_blocks} time blocks") if args.workload == "denovo" else
                                   (f"configs/linker {args.model} sampling, BASELINE configs[2]: {shape} fragment-pair pockets, "
                                    f"10-35 context + 3-14 generated atoms (partial gen_flag); bench step = 1 step at each of "
                                    f"{n_blocks} time blocks"),
                       "graphs_per_gpu": n_graphs, "nodes_per_gpu": N, "denoising_steps_per_bench_step": n_blocks,
                       "graph_steps_per_bench_step_per_gpu": n_graphs * n_blocks,
                       "ms_per_denoising_step_of_the_job": round(1e3 * el_max / args.steps / n_blocks, 4),
                       "sharding": (f"ONE job of {args.pockets} pockets split over {world} ranks" if split else
                                    f"independent pockets x{world} ranks") + ", no data-path collective",
                       "ranks_seen": seen, "collective_backend": collective_backend(),
                       "per_rank_graph_steps_per_s": per_rank,
                       "launch": "hipGraph replay per batch and step" if use_graph else "stream launches",
                       "streams": n_streams, "edge_workgroups": edge_wgs or 256},
        }

        if rank == 0 and not args.no_roofline:
            # live per-kernel timing with HIP events on the launch stream (same inputs, separate pass so the
            # event records do not perturb `value`); the dominant kernel's launches all cover one whole batch
            lib = _native.lib()
            st = states[0]
            Nb = st["N"]
            prof_steps = 2
            names = _native.PROFILE_CLASSES

            _native.check(lib.cbgx_profile_begin(80 * 5 * prof_steps + 64), "cbgx_profile_begin")
            for i in range(prof_steps):
                for t in block_times(args.warmup + args.steps + i, T):
                    model.denoise_step(st, t)
            NCLS = len(names)
            ms = (ctypes.c_double * NCLS)(); cnt = (ctypes.c_int * NCLS)()
            _native.check(lib.cbgx_profile_end(ms, cnt, NCLS), "cbgx_profile_end")
            # class "edge_x2h" = the launches that process all N nodes of the batch (the samplers let the library prune
            # the last two layers and cache the first two, reported separately as "edge_x2h_listed")
            per = {n: [round(1e3 * ms[i] / max(cnt[i], 1), 1), cnt[i]] for i, n in enumerate(names) if cnt[i]}
            deg_edges = 32 * Nb  # every node of a >=33-node graph has exactly 32 incoming edges
            x2h_bytes = X2H_BYTES_PER_EDGE * deg_edges + X2H_BYTES_PER_NODE * Nb
            x2h_s = 1e-3 * ms[4] / max(cnt[4], 1)
            achieved = x2h_bytes / x2h_s / 1e9 if x2h_s > 0 else 0.0
            layer_flops = FLOPS_PER_EDGE_LAYER * deg_edges + FLOPS_PER_NODE_LAYER * Nb
            dev_s_layer = 1e-3 * (ms[2] + ms[3] + ms[4] + ms[5] + ms[6]) / max(cnt[4] + cnt[6], 1)
            tf = lambda flops: round(flops / x2h_s / 1e12, 2) if x2h_s > 0 else 0
            out["roofline"] = {
                # dominant kernel = the x2h edge stage of a layer that covers all N nodes of a batch: ONE launch of
                # edge_x2h_dual_kernel (protein-only role with the query folded in registers + general role); achieved =
                # algorithmic bytes at the reference tensor boundary (1032 B/edge + 1536 B/node) / launch time,
                # HIP events on the launch stream, first resident batch.  Long-form reading: DESIGN.md section 6.
                "bound": "hbm", "kernel": "cbgx::edge_x2h_dual_kernel (fused x2h edge stage, all nodes of a layer)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": measured_traffic(Nb), "traffic_source": TRAFFIC_SOURCE,
                # the hardware view next to the nominal `frac`: measured HBM bytes of a launch / its time / the HBM peak (the kernel is
                # fused -- per-edge k / v never exist in memory -- so this is far below `frac` by design)
                "hbm_real": (round(measured_traffic(Nb) / x2h_s / 1e9 / HBM_PEAK_GBS, 4)
                             if (measured_traffic(Nb) and x2h_s > 0) else None),
                # the kernel's own bound: the matrix-pipe issue floor of a launch -- per node 128 f16 MFMAs (17
                # cycles measured, scripts/ubench/pipes.hip) + 128 fp32 16x16x4 MFMAs (32 cycles) = 6 272 SIMD cycles, on 1024 SIMDs at
                # 2.4 GHz -- divided by the measured launch time.  (fp32 MFMAs and VALU share a SIMD's issue port: ~1 400 VALU per node
                # on top)
                "issue_frac": round(Nb * X2H_MFMA_ISSUE_CYCLES_PER_NODE / (N_SIMDS * SIMD_CLOCK_HZ) / x2h_s, 4) if x2h_s > 0 else None,
                "nodes_per_launch": Nb, "algorithmic_bytes_per_launch": x2h_bytes, "avg_launch_us": round(1e6 * x2h_s, 3),
                "mfma_view": {"fp32_mfma_tflops": tf(X2H_FP32_MFMA_FLOPS_PER_EDGE * deg_edges), "fp32_peak_tflops": FP32_MFMA_PEAK_TFLOPS,
                              "f16_mfma_tflops_issued": tf(X2H_F16_MFMA_FLOPS_PER_NODE * Nb), "f16_peak_tflops": F16_MFMA_PEAK_TFLOPS,
                              "useful_tflops": tf(X2H_EXEC_FLOPS_PER_EDGE * deg_edges + X2H_EXEC_FLOPS_PER_NODE * Nb),
                              "reference_factored_equiv_tflops": round(layer_flops / dev_s_layer / 1e12, 2) if dev_s_layer else 0},
                "per_kernel_us_avg_and_launches": per,
                "profiled_sections_per_denoising_step": round(sum(cnt) / (5.0 * prof_steps), 1),
            }
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            from oracle import weights as OW
            osd = {"targetdiff": oracle_state_dict, "diffbp": lambda: OW.synthetic_state_dict_diffbp(13, 9, seed=0, num_timesteps=T),
                   "diffsbdd": lambda: OW.synthetic_state_dict_diffsbdd(8, 9, seed=0, num_timesteps=T)}[args.model]()
            out["cpu_baseline"] = cpu_baseline(osd, seed=1000, model=args.model)
            out["cpu_baseline"]["all_cores"] = cpu_baseline_concurrent(out["cpu_baseline"]["cores"], model=args.model)
    finally:
        _native.lib().cbgx_set_edge_workgroups(0)
    del states, model
    torch.cuda.empty_cache()
    return out


def _row(out, keep=()):
    """the fields of a full line that a secondary row keeps (compact: the whole line has to fit the driver's 8 KB tail)"""
    r = {"value": out["value"], "ms_per_step": out["ms_per_step"], "steps": out["steps"]}
    if "roofline" in out:
        rf = out["roofline"]
        r["frac"] = round(rf["frac"], 4)
        r["dominant_us"] = round(rf["avg_launch_us"], 1)
        r["us"] = {k: round(v[0]) for k, v in rf["per_kernel_us_avg_and_launches"].items()}
    for k in keep:
        r[k] = out["config"][k]
    return r


def sample_cli_end_to_end(dev, pockets=20, samples=10, config="targetdiff_test.yml", pockets_per_batch=None, streams=3):
    """ONE run of the sampling driver (cbgbench_amd/sample_cli.py = the role of the reference's sample.py:159-230) at the full
    T = 1000: synthetic pockets -> priors -> one 200-graph batch -> model.sample (static-context cache, 1000 steps, the 1001-entry
    trajectory kept on the device and downloaded once) -> one result file per pocket.  Wall time of everything after the model
    is built."""
    import shutil
    import tempfile
    from cbgbench_amd import sample_cli
    cfg = os.path.join(ROOT, "tests", "fixtures", config)      # the reference's config schema, T = 1000
    name = config[:config.rfind(".")]
    tmp = tempfile.mkdtemp(prefix="cbgx_bench_")
    try:
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        stats = {}
        rc = sample_cli.main(["--config", cfg, "--out_root", tmp, "--synthetic", str(pockets), "--num_samples", str(samples),
                              "--pockets_per_batch", str(pockets_per_batch or pockets), "--streams", str(streams), "--random_init"], stats=stats)
        torch.cuda.synchronize(dev)
        wall = time.perf_counter() - t0
        files = sorted(os.listdir(os.path.join(tmp, name)))
        nbytes = sum(os.path.getsize(os.path.join(tmp, name, f)) for f in files)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    assert rc == 0 and len(files) == pockets, (rc, files)
    gs = pockets * samples * 1000
    return {"value": round(gs / wall, 2), "unit": "graph-steps/s", "wall_s": round(wall, 3), "graphs": pockets * samples,
            "denoising_steps": 1000, "result_files": len(files), "result_bytes": nbytes,
            "phases_s": {k: round(v, 2) for k, v in stats.items() if not k.endswith("_s")},
            # the `sample` phase itemised: what an end-to-end user pays around the T steps
            "begin_sampling_ms": round(1e3 * stats.get("begin_sampling_s", 0.0), 1),
            "steps_ms": round(1e3 * stats.get("steps_s", 0.0), 1),
            "traj_download_ms": round(1e3 * stats.get("traj_download_s", 0.0), 1),
            "what": "python -m cbgbench_amd.sample_cli, T = 1000, wall time config load -> last result file"}


def secondary_block(args, dev, primary):
    """Other BASELINE configs measured in the same process as the headline (rank 0, N = 1): a few seconds each."""
    ns = lambda **kw: argparse.Namespace(**{**vars(args), "no_cpu_baseline": True, "no_secondary": True, **kw})
    sec = {}
    t_all = time.perf_counter()

    def guarded(name, fn):
        t0 = time.perf_counter()
        try:
            sec[name] = fn()
        except Exception as e:      # a failing secondary row must not take the headline line with it
            sec[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        sec[name]["measured_in_s"] = round(time.perf_counter() - t0, 2)
        torch.cuda.empty_cache()

    guarded("linker_256_graphs", lambda: _row(bench_sampling(ns(workload="linker", pockets=256, samples=1, graphs_per_batch=256,
                                                                  steps=4, warmup=2), 0, 1, dev), keep=("nodes_per_gpu",)))
    # BASELINE configs[3] per-GPU shape: 1000 DISTINCT pockets, one sample each (the driver's --gpus N line shards configs[1] jobs)
    guarded("denovo_1000_pockets_1_sample", lambda: _row(bench_sampling(ns(pockets=1000, samples=1, graphs_per_batch=200, steps=3,
                                                                            warmup=1, no_roofline=True), 0, 1, dev),
                                                         keep=("nodes_per_gpu",)))
    def train_row():
        r = _row(bench_train(ns(workload="train", pockets=32, steps=10, warmup=3), 0, 1, dev), keep=("nodes_per_batch",))
        if not args.no_cpu_baseline:        # the CPU port's training step beside it: bounded, ~15 s
            r["cpu_baseline"] = cpu_train_baseline(oracle_state_dict(), seed=3000, max_seconds=12.0)
            r["cpu_baseline"]["sample"] = r["cpu_baseline"]["sample"][:110]
        return r
    guarded("train_32_graphs", train_row)
    for m in ("diffbp", "diffsbdd"):
        guarded(f"{m}_200_graphs", lambda m=m: _row(bench_sampling(ns(model=m, pockets=20, samples=10, graphs_per_batch=200, steps=3,
                                                                      warmup=1), 0, 1, dev), keep=("nodes_per_gpu",)))
        guarded(f"train_{m}_32_graphs", lambda m=m: _row(bench_train(ns(workload="train", model=m, pockets=32, steps=6, warmup=2,
                                                                        no_roofline=True), 0, 1, dev), keep=("nodes_per_batch",)))
    guarded("denovo_10_graphs", lambda: _row(bench_sampling(ns(pockets=1, samples=10, graphs_per_batch=10, steps=20, warmup=5), 0, 1, dev),
                                             keep=("nodes_per_gpu",)))
    guarded("denovo_1_graph", lambda: _row(bench_sampling(ns(pockets=1, samples=1, graphs_per_batch=1, steps=20, warmup=5), 0, 1, dev),
                                           keep=("nodes_per_gpu",)))
    # the same small batches as a JOB (the reference's loop over batches, sample.py:159-230, here with eight batches in flight on
    # eight streams -- TargetDiff.sample_many / sample_cli --streams 8): throughput, where the two rows above are one batch's latency
    guarded("denovo_10_graphs_per_batch_8_in_flight", lambda: _row(bench_sampling(
        ns(pockets=16, samples=10, graphs_per_batch=10, steps=20, warmup=5, streams=8), 0, 1, dev), keep=("nodes_per_gpu", "streams")))
    guarded("denovo_1_graph_per_batch_8_in_flight", lambda: _row(bench_sampling(
        ns(pockets=32, samples=1, graphs_per_batch=1, steps=20, warmup=5, streams=8), 0, 1, dev), keep=("nodes_per_gpu", "streams")))

    def e2e():
        # the WHOLE configs[1] job through the driver, not a sample of its steps: 100 pockets x 10 samples, T = 1000, as the headline
        # holds it (three batches of <= 340 graphs in flight on three streams), config -> priors -> 1000 steps -> trajectory download
        # -> one result file per pocket (sample.py:159-230); ~35 s
        r = sample_cli_end_to_end(dev, pockets=100, samples=10, pockets_per_batch=34, streams=3)
        r["ratio_to_step_sampled_headline"] = round(r["value"] / primary["value"], 4)
        r["what"] = "python -m cbgbench_amd.sample_cli, the full configs[1] job (100 pockets x 10 samples, T = 1000, 3 batches in flight), wall"
        return r
    guarded("sample_cli_T1000_full_job_1000_graphs", e2e)

    def e2e_linker():
        # BASELINE configs[2] through the driver: a linker config -- context atoms per pocket, frame centred on
        # them, generated atoms appended (assign_gensize) -- 256 pockets x 1 sample in one batch, T = 1000, per-pocket files
        r = sample_cli_end_to_end(dev, pockets=256, samples=1, config="linker_targetdiff_test.yml")
        r["what"] = "python -m cbgbench_amd.sample_cli, configs/linker schema (context atoms, ctx centring), T = 1000, 256 graphs"
        return r
    guarded("sample_cli_linker_T1000_256_graphs", e2e_linker)
    sec["total_s"] = round(time.perf_counter() - t_all, 2)
    return sec


if __name__ == "__main__":
    main()
