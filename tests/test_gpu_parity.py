"""Parity of the HIP path (through the C ABI) against golden vectors made by the reference itself and
against the CPU oracle.  Tolerances: fp32, summation-order differences only -> 1e-4 relative + 1e-5
absolute on x / h / delta_x / logits (round 4: h and the logits no longer get a tenfold absolute
allowance -- the measured maxima over all goldens are 2.1e-6 / 1.9e-6, profiles/parity_errors_r04f.md, written by the last test of
this file on every GPU run), identical atom-type argmax on ligand rows, bit-exact neighbour lists.  Only the 5-step DiffSBDD
trajectory keeps a 1e-4 absolute term: its update divides by alpha_ts every step (measured 8.6e-6 on x, 7.3e-4 on type features of
magnitude 1.6e3)."""
import os

import numpy as np
import pytest
import torch

import cbgbench_amd as C
from cbgbench_amd import stages, synthetic
from cbgbench_amd.unitransformer import graph_ptr_from_batch
from oracle import targetdiff as OT
from oracle import unitransformer as OU
from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
RTOL, ATOL = 1e-4, 1e-5
DENOISER_CASES = ["denoiser_2graphs", "denoiser_small_graphs", "denoiser_linker", "denoiser_eg5_pocket10",
                  "denoiser_adrb1_pocket10", "denoiser_drd2_pocket10", "denoiser_smarca2_pocket10"]


MEASURED = {}     # what -> (max abs err, max |ref|, max of err / (ATOL scale + RTOL |ref|)) over every check of this session


def close(a, b, what, scale=1.0):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    tol = ATOL * scale + RTOL * b.abs()
    key = what.split("[")[0].strip()
    m = MEASURED.get(key, (0.0, 0.0, 0.0))
    if err.numel():
        MEASURED[key] = (max(m[0], float(err.max())), max(m[1], float(b.abs().max())), max(m[2], float((err / tol).max())))
    assert bool((err <= tol).all()), f"{what}: max abs err {err.max():.3e} (|ref| max {b.abs().max():.3e})"


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


@pytest.fixture(scope="module")
def model(synthetic_sd):
    m = C.get_model(C.default_targetdiff_config(13)).eval()
    m.load_state_dict(synthetic_sd, strict=True)
    return m.to(DEV)


def model_sd(m):
    """the model's weights as the CPU state dict the oracle takes"""
    return {k: v.detach().cpu() for k, v in m.state_dict().items()}


def dev_inputs(g):
    x, h = g["x"].to(DEV), g["h"].to(DEV)
    gp = graph_ptr_from_batch(g["batch_idx"].to(DEV))
    lig = g["lig_flag"].to(DEV).to(torch.uint8)
    gen = g["gen_flag"].to(DEV).to(torch.uint8)
    return x, h, gp, lig, gen


@pytest.mark.parametrize("case", DENOISER_CASES)
def test_knn_graph_bitexact(golden_dir, case):
    g = load(golden_dir, case)
    x, h, gp, lig, gen = dev_inputs(g)
    nbr, deg = stages.knn_graph(x, gp)
    ei = stages.edge_index_from_nbr(nbr, deg).cpu()
    assert torch.equal(ei.int(), g["edge_index"])
    pad = nbr.cpu()[torch.arange(32)[None, :] >= deg.cpu()[:, None]]
    assert bool((pad == -1).all())


@pytest.mark.parametrize("sizes", [[1000], [770, 3, 900], [768, 769]])
def test_knn_large_graphs_bitexact(sizes):
    """graphs above the register-cached limit (768 nodes) take the rescanning path; both must match the oracle."""
    g = torch.Generator().manual_seed(sum(sizes))
    x = torch.randn(sum(sizes), 3, generator=g) * 6
    batch = torch.cat([torch.full((n,), b) for b, n in enumerate(sizes)])
    ref = OU.knn_graph(x, batch, 32)
    nbr, deg = stages.knn_graph(x.to(DEV), graph_ptr_from_batch(batch.to(DEV)))
    assert torch.equal(stages.edge_index_from_nbr(nbr, deg).cpu(), ref)


@pytest.fixture(params=[0, 1], ids=["mfma", "valu"])
def edge_impl(request):
    """Both generations of the fused edge kernels are checked against the reference."""
    from cbgbench_amd import _native
    if request.param == 0:
        yield 0                                         # libcbgx.so: the product path
    else:
        with _native.first_generation_kernels():        # libcbgx_xcheck.so (test-only): first-generation VALU kernels
            yield 1


@pytest.mark.parametrize("case", DENOISER_CASES)
def test_stages_match_reference(golden_dir, model, case, edge_impl):
    g = load(golden_dir, case)
    x, h, gp, lig, gen = dev_inputs(g)
    packed = model.denoiser.packed_weights(torch.device(DEV))
    nbr, deg = stages.knn_graph(x, gp)
    e_w = stages.edge_gate(packed, x, nbr, deg)
    mask = (torch.arange(32, device=DEV)[None, :] < deg[:, None])
    close(e_w[mask], g["e_w"].flatten(), "e_w")
    assert bool((e_w[~mask] == 0).all())
    h1 = stages.x2h_attention(packed, 0, x, h, nbr, deg, lig, e_w)
    close(h1, g["h_layer0"], "x2h layer 0")
    x1, dx = stages.h2x_attention(packed, 0, x, h1, nbr, deg, lig, gen, e_w)
    close(x1, g["x_layer0"], "h2x layer 0")
    moved = (x1 != x).any(-1).cpu()
    assert not bool(moved[~g["gen_flag"]].any()), "gen_flag=False atoms must not move (unitransformer.py:182)"
    logits = stages.classifier(packed, 9, 13, g["h_out"].to(DEV))
    close(logits, g["logits"], "classifier")


@pytest.mark.parametrize("case", DENOISER_CASES)
def test_full_denoiser_matches_reference(golden_dir, model, case):
    g = load(golden_dir, case)
    with torch.no_grad():
        xo, ho, lo = model.denoiser(x=g["x"].to(DEV), h=g["h"].to(DEV), batch_idx=g["batch_idx"].to(DEV),
                                    lig_flag=g["lig_flag"].to(DEV), gen_flag=g["gen_flag"].to(DEV))
    close(xo, g["x_out"], "x_out")
    close(ho, g["h_out"], "h_out")   # measured 2.1e-6 on |h| up to 5.2 (profiles/parity_errors_r04f.md)
    close(lo, g["logits"], "logits")
    lig = g["lig_flag"]
    assert torch.equal(lo.cpu()[lig].argmax(-1), g["logits"][lig].argmax(-1))
    assert torch.equal(xo.cpu()[~g["gen_flag"]], g["x"][~g["gen_flag"]])


def test_sample_loop_matches_reference(golden_dir):
    """TargetDiff.sample end to end on a 5-step model with the reference's noise replayed."""
    g = load(golden_dir, "sample_T5")
    T = int(g["T"])
    m = C.get_model(C.default_targetdiff_config(13, num_diffusion_timesteps=T)).eval()
    m.load_state_dict(W.synthetic_state_dict(13, 9, seed=0, num_timesteps=T), strict=True)
    m = m.to(DEV)
    batch = {k[len("batch_"):]: v for k, v in g.items() if k.startswith("batch_")}
    n_lig = batch["ligand_pos"].shape[0]
    torch.manual_seed(int(g["seed"]))
    tape = {}
    for t in reversed(range(T)):
        tape[t] = (torch.randn(n_lig, 3).to(DEV), torch.rand(n_lig, 13).to(DEV))
    traj = m.sample(synthetic.batch_to(batch, DEV), noise_tape=tape)
    assert sorted(traj.keys()) == list(range(-1, T))
    for t in range(-1, T):
        close(traj[t][0], g[f"traj_x_{t}"], f"traj x[{t}]")
        assert torch.equal(traj[t][1], g[f"traj_c_{t}"]), f"traj c[{t}]"
        assert torch.equal(traj[t][2], batch["ligand_element_batch"])


@pytest.mark.parametrize("case", ["step_t500", "step_t0", "step_t999_linker"])
def test_teacher_forced_step(golden_dir, model, case):
    g = load(golden_dir, case)
    batch = {k[len("batch_"):]: v for k, v in g.items() if k.startswith("batch_")}
    t_idx = int(g["t_idx"])
    # run exactly one step of sample(): a 1000-step model teacher-forced at t_idx
    b = synthetic.batch_to(batch, DEV)
    bl, br = b["ligand_element_batch"], b["protein_element_batch"]
    sort_idx, batch_idx, lig_flag, lig_rows, gp = model.compose_plan(bl, br)
    gen_l = b.get("ligand_gen_flag", b["ligand_lig_flag"]).bool()
    gen = torch.cat([torch.zeros_like(b["protein_lig_flag"]), gen_l])[sort_idx]
    aa = torch.nn.functional.one_hot(b["protein_aa_type"], 20).float()
    c_lig = torch.nn.functional.one_hot(b["ligand_atom_type"], 13).float()
    with torch.no_grad():
        h = torch.cat([model.context_embedder.embed_protein(b["protein_atom_feature"], aa),
                       model.context_embedder.embed_ligand(c_lig)])[sort_idx]
        x = torch.cat([b["protein_pos"], b["ligand_pos"]])[sort_idx]
        xo, _, lo = model.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
        x_pred, c_pred = xo[lig_rows], lo[lig_rows]
        close(x_pred, g["x_pred"], "x0 prediction")
        close(c_pred, g["c_pred"], "type logits")
        assert torch.equal(c_pred.argmax(-1).cpu(), g["c_pred"].argmax(-1))
        B = int(bl.max()) + 1
        t = torch.full((B,), t_idx, dtype=torch.long, device=DEV)
        x_next = model.pos_scheduler.backward_remove_noise(x_pred, b["ligand_pos"], t, bl, gen_l, type="denoise",
                                                           noise=g["eps"].to(DEV))
        c_next, v_next = model.type_scheduler.backward_remove_noise(c_pred, c_lig, t, bl, gen_l, uniform=g["u"].to(DEV))
    close(x_next, g["x_next"], "x_{t-1}")
    assert torch.equal(v_next.cpu(), g["v_next"])


# ---- bigger than the fixtures: oracle on the same seeded inputs --------------------------------------
def _composed(model, batch):
    b = synthetic.batch_to(batch, DEV)
    bl, br = b["ligand_element_batch"], b["protein_element_batch"]
    sort_idx, batch_idx, lig_flag, lig_rows, gp = model.compose_plan(bl, br)
    gen_l = b.get("ligand_gen_flag", b["ligand_lig_flag"]).bool()
    gen = torch.cat([torch.zeros_like(b["protein_lig_flag"]), gen_l])[sort_idx]
    aa = torch.nn.functional.one_hot(b["protein_aa_type"], 20).float()
    c_lig = torch.nn.functional.one_hot(b["ligand_atom_type"], 13).float()
    with torch.no_grad():
        h = torch.cat([model.context_embedder.embed_protein(b["protein_atom_feature"], aa),
                       model.context_embedder.embed_ligand(c_lig)])[sort_idx]
    x = torch.cat([b["protein_pos"], b["ligand_pos"]])[sort_idx]
    return x, h, batch_idx, lig_flag, gen, gp


@pytest.mark.parametrize("maker,n", [(synthetic.denovo_batch, 3), (synthetic.linker_batch, 4)])
def test_config_sized_graphs_vs_oracle(model, synthetic_sd, maker, n):
    """Full-size pockets (N_rec 350..650, ragged ligands, partial gen_flag for the linker batch)."""
    x, h, batch_idx, lig_flag, gen, gp = _composed(model, maker(n, seed=3))
    with torch.no_grad():
        xo, ho, lo = model.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
    rx, rh, rl = OU.unitransformer_forward(synthetic_sd, x.cpu(), h.cpu(), batch_idx.cpu(), lig_flag.cpu(), gen.cpu())
    close(xo, rx, "x_out")
    close(ho, rh, "h_out")
    close(lo, rl, "logits")
    lig = lig_flag.cpu()
    assert torch.equal(lo.cpu()[lig].argmax(-1), rl[lig].argmax(-1))


# ---- size-independent properties at config-2 / config-3 batch shapes -------------------------------
def _rand_rotation(seed):
    g = torch.Generator().manual_seed(seed)
    q, r = torch.linalg.qr(torch.randn(3, 3, generator=g, dtype=torch.float64))
    q = q * torch.sign(torch.diagonal(r))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    return q.float()


def test_properties_at_full_batch(model):
    """config 2 shape (10 samples of one pocket): E(3) equivariance, batch independence, determinism."""
    batch = synthetic.denovo_batch(10, seed=5, same_pocket=True)
    x, h, batch_idx, lig_flag, gen, gp = _composed(model, batch)
    den = model.denoiser
    with torch.no_grad():
        xo, ho, lo = den(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
        xo2, ho2, lo2 = den(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
        assert torch.equal(xo, xo2) and torch.equal(ho, ho2) and torch.equal(lo, lo2), "not deterministic"
        assert torch.isfinite(xo).all() and torch.isfinite(ho).all() and torch.isfinite(lo).all()
        # rotation + translation
        R = _rand_rotation(1).to(DEV)
        tr = torch.tensor([3.0, -2.0, 5.0], device=DEV)
        xr, hr, lr = den(x=x @ R.T + tr, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
        close(xr, xo @ R.T + tr, "equivariance of x")
        close(hr, ho, "invariance of h")
        assert torch.equal(lr[lig_flag].argmax(-1), lo[lig_flag].argmax(-1))
        # batch independence: graph 3 alone gives the same rows
        s, e = int(gp[3]), int(gp[4])
        gp1 = torch.tensor([0, e - s], dtype=torch.int32, device=DEV)
        x1, h1, l1 = den(x=x[s:e].contiguous(), h=h[s:e].contiguous(), batch_idx=torch.zeros(e - s, dtype=torch.long, device=DEV),
                         lig_flag=lig_flag[s:e].contiguous(), gen_flag=gen[s:e].contiguous(), graph_ptr=gp1)
        assert torch.equal(x1, xo[s:e]) and torch.equal(h1, ho[s:e]) and torch.equal(l1, lo[s:e])
        # protein atoms never move
        assert torch.equal(xo[~gen], x[~gen])


def test_permutation_of_atoms_within_a_graph(model):
    """relabelling the atoms of a graph (protein atoms among themselves, ligand atoms among themselves, which
    keeps the composed order of common.py:200) permutes the outputs the same way"""
    batch = synthetic.denovo_batch(4, seed=9)
    x, h, batch_idx, lig_flag, gen, gp = _composed(model, batch)
    g = torch.Generator().manual_seed(3)
    perm = []
    for b in range(4):
        s, e = int(gp[b]), int(gp[b + 1])
        n_rec = int((~lig_flag[s:e]).sum())
        assert not bool(lig_flag[s:s + n_rec].any())            # protein rows first
        perm.append(s + torch.randperm(n_rec, generator=g))
        perm.append(s + n_rec + torch.randperm(e - s - n_rec, generator=g))
    perm = torch.cat(perm).to(DEV)
    den = model.denoiser
    with torch.no_grad():
        xo, ho, lo = den(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
        xp, hp, lp = den(x=x[perm].contiguous(), h=h[perm].contiguous(), batch_idx=batch_idx, lig_flag=lig_flag[perm].contiguous(),
                         gen_flag=gen[perm].contiguous(), graph_ptr=gp)
    close(xp, xo[perm], "x_out under an atom permutation")
    close(hp, ho[perm], "h_out under an atom permutation")
    lig = lig_flag[perm]
    assert torch.equal(lp[lig].argmax(-1), lo[perm][lig].argmax(-1))


def oracle_on_chosen_graphs(sd, tensors, outputs, n_graphs, seed, what, count=16, chunk=8):
    """`count` graphs of a config-sized batch, chosen by `seed` (plus the first and the last: lowest / highest addresses), against the
    CPU oracle, `chunk` graphs per oracle call (graphs are independent: a chunk's result is the batch's result for those graphs;
    16 graphs cost ~30 s of host time, all of them minutes -- VERDICT r5 weak #2)."""
    x, h, batch_idx, lig_flag, gen, gp = tensors
    xo, ho, lo = outputs
    gpc = gp.cpu().tolist()
    rng = np.random.default_rng(seed)
    chosen = sorted([0, n_graphs - 1] + (1 + rng.choice(n_graphs - 2, size=count - 2, replace=False)).tolist())
    for c0 in range(0, len(chosen), chunk):
        gs = chosen[c0:c0 + chunk]
        rows = torch.cat([torch.arange(gpc[g], gpc[g + 1]) for g in gs])
        bi = torch.cat([torch.full((gpc[g + 1] - gpc[g],), k, dtype=torch.long) for k, g in enumerate(gs)])
        rd = rows.to(DEV)
        rx, rh, rl = OU.unitransformer_forward(sd, x[rd].cpu(), h[rd].cpu(), bi, lig_flag[rd].cpu().bool(), gen[rd].cpu().bool())
        close(xo[rd], rx, f"x_out ({what}, graphs {gs})")
        close(ho[rd], rh, f"h_out ({what}, graphs {gs})")
        lig = lig_flag[rd].cpu().bool()
        assert torch.equal(lo[rd].cpu()[lig].argmax(-1), rl[lig].argmax(-1)), (what, gs)
    return chosen


def test_full_config2_job_in_one_batch(model, synthetic_sd):
    """BASELINE configs[1] at full size in ONE batch: 100 pockets x 10 samples = 1000 graphs, ~5.3e5 nodes -- the folded
    query alone is 4.3 GB, so every 32-bit byte offset would wrap.  The last graph (highest addresses) and one past the
    2 GB mark must equal the same graph run alone bit for bit (batch independence); 16 seed-chosen graphs (first and last included)
    must match the CPU oracle."""
    rng = np.random.default_rng(11)
    pk = [synthetic.make_pocket(rng, int(rng.integers(350, 651))) for _ in range(100)]
    plist = [p for p in pk for _ in range(10)]
    nlig = [int(rng.integers(10, 46)) for _ in plist]
    batch = synthetic.make_batch(plist, nlig, rng, 13)
    x, h, batch_idx, lig_flag, gen, gp = _composed(model, batch)
    assert x.shape[0] > 450_000
    den = model.denoiser
    with torch.no_grad():
        xo, ho, lo = den(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
        assert torch.isfinite(xo).all() and torch.isfinite(ho).all() and torch.isfinite(lo).all()
        assert torch.equal(xo[~gen], x[~gen]) and bool((xo[gen] != x[gen]).any())
        for gidx in (999, 620, 0):
            s, e = int(gp[gidx]), int(gp[gidx + 1])
            gp1 = torch.tensor([0, e - s], dtype=torch.int32, device=DEV)
            x1, h1, l1 = den(x=x[s:e].contiguous(), h=h[s:e].contiguous(),
                             batch_idx=torch.zeros(e - s, dtype=torch.long, device=DEV),
                             lig_flag=lig_flag[s:e].contiguous(), gen_flag=gen[s:e].contiguous(), graph_ptr=gp1)
            assert torch.equal(x1, xo[s:e]) and torch.equal(h1, ho[s:e]) and torch.equal(l1, lo[s:e]), gidx
    chosen = oracle_on_chosen_graphs(synthetic_sd, (x, h, batch_idx, lig_flag, gen, gp), (xo, ho, lo), 1000, seed=2024,
                                     what="config 2, 1000 graphs")
    assert len(chosen) == 16 and 0 in chosen and 999 in chosen


def test_linker_256_graphs_runs_and_freezes_context(model):
    """config 3 shape: 256 ragged linker graphs in one batch (N ~ 1.3e5, E ~ 4e6)."""
    batch = synthetic.linker_batch(256, seed=7)
    x, h, batch_idx, lig_flag, gen, gp = _composed(model, batch)
    with torch.no_grad():
        xo, ho, lo = model.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
    assert torch.isfinite(xo).all() and torch.isfinite(ho).all() and torch.isfinite(lo).all()
    assert torch.equal(xo[~gen], x[~gen])
    assert bool((xo[gen] != x[gen]).any())
    # the whole batch against the first-generation VALU kernels on the device (independent implementation)
    from cbgbench_amd import _native
    with _native.first_generation_kernels(), torch.no_grad():
        xv, hv, lv = model.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
    close(xo, xv, "x_out mfma vs valu kernels (256 graphs)")
    close(ho, hv, "h_out mfma vs valu kernels (256 graphs)")
    assert torch.equal(lo[lig_flag].argmax(-1), lv[lig_flag].argmax(-1))
    # 16 seed-chosen graphs of the 256 against the oracle (all 256: the opt-in test below, 8 minutes)
    chosen = oracle_on_chosen_graphs(W.synthetic_state_dict(13, 9), (x, h, batch_idx, lig_flag, gen, gp), (xo, ho, lo), 256, seed=7,
                                     what="linker, 256 graphs")
    assert len(chosen) == 16


@pytest.mark.slow
@pytest.mark.skipif(os.environ.get("CBGX_SLOW_TESTS") != "1", reason="8 minutes of CPU oracle work on the GPU box: set CBGX_SLOW_TESTS=1 "
                    "(ran and passed in round 5's fifth GPU call, profiles/pytest_gpu_r05e.log: 484.74 s, 109 passed)")
def test_linker_256_graphs_every_graph_against_the_oracle(model):
    """configs[2] at full size, ALL 256 graphs against the CPU oracle (VERDICT r4 weak #1b: until round 5 one graph was, the rest
    only against the library's own first-generation kernels).  The oracle runs the batch in chunks of 8 graphs (8 minutes on
    the GPU box's host cores, hence opt-in); graphs are independent, so a chunk's result is the batch's result for those graphs."""
    batch = synthetic.linker_batch(256, seed=7)
    x, h, batch_idx, lig_flag, gen, gp = _composed(model, batch)
    with torch.no_grad():
        xo, ho, lo = model.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
    sd = W.synthetic_state_dict(13, 9)
    gpc = gp.cpu()
    worst = {"x": 0.0, "h": 0.0}
    for g0 in range(0, 256, 8):
        s, e = int(gpc[g0]), int(gpc[min(g0 + 8, 256)])
        bi = (batch_idx[s:e] - batch_idx[s]).cpu()
        rx, rh, rl = OU.unitransformer_forward(sd, x[s:e].cpu(), h[s:e].cpu(), bi, lig_flag[s:e].cpu(), gen[s:e].cpu())
        close(xo[s:e], rx, f"x_out (graphs {g0}..{g0 + 7} of 256)")
        close(ho[s:e], rh, f"h_out (graphs {g0}..{g0 + 7} of 256)")
        lig = lig_flag[s:e].cpu()
        assert torch.equal(lo[s:e].cpu()[lig].argmax(-1), rl[lig].argmax(-1)), g0
        worst["x"] = max(worst["x"], float((xo[s:e].cpu() - rx).abs().max()))
        worst["h"] = max(worst["h"], float((ho[s:e].cpu() - rh).abs().max()))
    print(f"linker-256 vs oracle, all graphs: max |dx| {worst['x']:.2e}, max |dh| {worst['h']:.2e}")


def test_no_movable_nodes(model, golden_dir):
    """gen_flag all False: the h2x work list is empty, x must come back bit-identical, h / logits unaffected by that."""
    g = load(golden_dir, "denoiser_2graphs")
    x, h = g["x"].to(DEV), g["h"].to(DEV)
    none = torch.zeros_like(g["gen_flag"]).to(DEV)
    with torch.no_grad():
        xo, ho, lo = model.denoiser(x=x, h=h, batch_idx=g["batch_idx"].to(DEV), lig_flag=g["lig_flag"].to(DEV),
                                    gen_flag=none)
    assert torch.equal(xo, x)
    assert torch.isfinite(ho).all() and torch.isfinite(lo).all()
    # layer-0 features do not depend on gen_flag at all
    packed = model.denoiser.packed_weights(torch.device(DEV))
    gp = graph_ptr_from_batch(g["batch_idx"].to(DEV))
    nbr, deg = stages.knn_graph(x, gp)
    e_w = stages.edge_gate(packed, x, nbr, deg)
    h1 = stages.x2h_attention(packed, 0, x, h, nbr, deg, g["lig_flag"].to(DEV).to(torch.uint8), e_w)
    close(h1, g["h_layer0"], "x2h layer 0")


@pytest.mark.parametrize("case", ["step_t500", "step_t0", "step_t999_linker"])
def test_native_step_kernels(golden_dir, model, case):
    """begin_sampling + one native denoise_step (prologue kernel, denoiser, epilogue kernel) against the reference."""
    g = load(golden_dir, case)
    batch = {k[len("batch_"):]: v for k, v in g.items() if k.startswith("batch_")}
    st = model.begin_sampling(synthetic.batch_to(batch, DEV), keep_trajectory=False)
    model.denoise_step(st, int(g["t_idx"]), noise=(g["eps"].to(DEV), g["u"].to(DEV)))
    close(st["x_lig"], g["x_next"], "x_{t-1}")
    assert torch.equal(st["c_lig"].cpu(), g["c_next"])
    if "ligand_gen_flag" in batch:
        keep = ~batch["ligand_gen_flag"]
        assert torch.equal(st["x_lig"].cpu()[keep], batch["ligand_pos"][keep])


def test_diffsbdd_sample_matches_reference(golden_dir):
    """DiffSBDD.sample (5-step model) on the GPU with the reference's Gaussian draws replayed."""
    g = load(golden_dir, "diffsbdd_sample_T5")
    T, Cn = int(g["T"]), 8
    m = C.get_model(C.default_diffsbdd_config(Cn, num_diffusion_timesteps=T)).eval()
    m.load_state_dict(W.synthetic_state_dict_diffsbdd(Cn, 9, seed=0, num_timesteps=T), strict=True)
    m = m.to(DEV)
    batch = {k[len("batch_"):]: v for k, v in g.items() if k.startswith("batch_")}
    n_lig = batch["ligand_element_batch"].shape[0]
    torch.manual_seed(int(g["seed"]))
    draws = []
    for _ in range(T + 2):
        draws += [torch.randn(n_lig, 3), torch.randn(n_lig, Cn)]
    traj = m.sample(synthetic.batch_to(batch, DEV), noise_draws=draws)
    assert sorted(traj.keys()) == list(range(-1, T))
    for t in range(-1, T):
        close(traj[t][0], g[f"traj_x_{t}"], f"diffsbdd traj x[{t}]", scale=10.0)
        close(traj[t][1], g[f"traj_c_{t}"], f"diffsbdd traj c[{t}]", scale=10.0)


def test_diffbp_sample_matches_reference(golden_dir):
    """DiffBP.sample (5-step model): denoiser + CoMPredictor (cbgx_h2x_stack_forward) + samplers, noise replayed."""
    g = load(golden_dir, "diffbp_sample_T5")
    T, Cn = int(g["T"]), 13
    m = C.get_model(C.default_diffbp_config(Cn, num_diffusion_timesteps=T)).eval()
    m.load_state_dict(W.synthetic_state_dict_diffbp(Cn, 9, seed=0, num_timesteps=T), strict=True)
    m = m.to(DEV)
    batch = {k[len("batch_"):]: v for k, v in g.items() if k.startswith("batch_")}
    n_lig = batch["ligand_pos"].shape[0]
    torch.manual_seed(int(g["seed"]))
    tape = {}
    for t in reversed(range(T)):
        tape[t] = (torch.randn(n_lig, 3).to(DEV), torch.rand(n_lig).to(DEV))
    traj = m.sample(synthetic.batch_to(batch, DEV), noise_tape=tape)
    assert sorted(traj.keys()) == list(range(-1, T))
    for t in range(-1, T):
        close(traj[t][0], g[f"traj_x_{t}"], f"diffbp traj x[{t}]")
        assert torch.equal(traj[t][1], g[f"traj_c_{t}"]), f"diffbp traj c[{t}]"


@pytest.mark.parametrize("maker,n", [(synthetic.denovo_batch, 6), (synthetic.linker_batch, 5)])
def test_receptive_field_pruning_is_exact(model, maker, n):
    """need_h=False restricts the last two x2h blocks to the nodes that can still reach x' / ligand logits:
    those outputs must be bit-identical to the full computation."""
    x, h, batch_idx, lig_flag, gen, gp = _composed(model, maker(n, seed=9))
    with torch.no_grad():
        xf, hf, lf = model.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
        xp, hp, lp = model.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp,
                                    need_h=False)
    assert hp is None
    assert torch.equal(xp, xf)
    assert torch.equal(lp[lig_flag], lf[lig_flag])


def test_sample_many_equals_sample(synthetic_sd):
    """three batches in flight on three streams (TargetDiff.sample_many: per-stream workspaces and auxiliary streams inside
    libcbgx) against the same batches sampled one after the other with the same noise: identical trajectories"""
    T = 12
    m = C.get_model(C.default_targetdiff_config(13, num_diffusion_timesteps=T)).eval()
    m.load_state_dict(W.synthetic_state_dict(13, 9, seed=0, num_timesteps=T), strict=True)
    m = m.to(DEV)
    batches = [synthetic.batch_to(synthetic.denovo_batch(n, seed=50 + n), DEV) for n in (3, 2, 4)]
    g = torch.Generator(device=DEV).manual_seed(11)
    tapes = []
    for b in batches:
        n_lig = b["ligand_pos"].shape[0]
        tapes.append({t: (torch.randn(n_lig, 3, device=DEV, generator=g), torch.rand(n_lig, 13, device=DEV, generator=g))
                      for t in range(T)})
    one = [m.sample(b, noise_tape=tp) for b, tp in zip(batches, tapes)]
    many = m.sample_many(batches, noise_tapes=tapes, streams=3)
    torch.cuda.synchronize()
    for a, b in zip(one, many):
        assert sorted(a.keys()) == sorted(b.keys()) == list(range(-1, T))
        for t in a:
            assert torch.equal(a[t][0], b[t][0]) and torch.equal(a[t][1], b[t][1]) and torch.equal(a[t][2], b[t][2])


@pytest.mark.parametrize("name", ["targetdiff", "diffbp", "diffsbdd"])
def test_sampling_under_inference_mode_equals_no_grad(name):
    """the flag / composed-row / packed-weight caches read tensor._version, which inference tensors do not have.
    A batch built and sampled inside torch.inference_mode() must run (uncached routes) and give the trajectory of the no_grad run
    for the same noise."""
    T, C_ = 6, (8 if name == "diffsbdd" else 13)
    cfg = {"targetdiff": C.default_targetdiff_config, "diffbp": C.default_diffbp_config, "diffsbdd": C.default_diffsbdd_config}[name]
    torch.manual_seed(3)
    m = C.get_model(cfg(C_, num_diffusion_timesteps=T)).eval().to(DEV)
    host = synthetic.denovo_batch(3, seed=77, n_rec_range=(120, 180), n_lig_range=(8, 14), num_classes=C_)
    torch.manual_seed(5)
    ref = m.sample(synthetic.batch_to(host, DEV))
    with torch.inference_mode():
        b = synthetic.batch_to({k: (v.clone() if torch.is_tensor(v) else v) for k, v in host.items()}, DEV)
        assert b["ligand_pos"].is_inference()
        torch.manual_seed(5)
        got = m.sample(b)
    assert sorted(ref.keys()) == sorted(got.keys())
    for t in ref:
        assert torch.equal(ref[t][0], got[t][0]) and torch.equal(ref[t][1], got[t][1]), (name, t)


def test_sample_many_with_step_graphs_equals_eager_steps(synthetic_sd):
    """small batches kept in flight as captured hipGraphs (sample_many(use_graph=True): four batches on two streams, so two graphs
    share a stream and all four are replayed interleaved, each with its own workspace) against eager steps of the same batches
    fed the noise the graphs drew: identical final states, every batch"""
    T = 10
    m = C.get_model(C.default_targetdiff_config(13, num_diffusion_timesteps=T)).eval()
    m.load_state_dict(W.synthetic_state_dict(13, 9, seed=0, num_timesteps=T), strict=True)
    m = m.to(DEV)
    batches = [synthetic.batch_to(synthetic.denovo_batch(n, seed=70 + n, n_rec_range=(120, 260)), DEV) for n in (1, 3, 2, 1)]
    torch.manual_seed(5)
    log = []
    many = m.sample_many(batches, streams=2, use_graph=True, noise_log=log, return_device=DEV)
    torch.cuda.synchronize()
    done = 2                                  # make_step_graph's eager warm-up steps: their noise is not recorded
    assert len(log) == len(batches) * (T - done)
    for k, b in enumerate(batches):
        traj = many[k]
        assert sorted(traj.keys()) == list(range(-1, T)) and torch.isfinite(traj[-1][0]).all()
        tape = {t: (eps, u) for kk, t, eps, u in log if kk == k}
        assert sorted(tape) == list(range(T - done))
        st = m.begin_sampling(b, keep_trajectory=True)
        st["x_lig"], st["c_lig"] = traj[T - done - 1][0].clone(), traj[T - done - 1][1].clone()
        for t in reversed(range(T - done)):
            m.denoise_step(st, t, noise=tape[t])
            assert torch.equal(st["x_lig"], traj[t - 1][0]) and torch.equal(st["c_lig"], traj[t - 1][1]), (k, t)
        assert not torch.equal(traj[-1][0], traj[T - 1][0])


@pytest.mark.parametrize("name", ["diffbp", "diffsbdd"])
def test_sample_many_equals_sample_other_model_classes(name):
    """DiffBP (CoMPredictor with its own per-stream workspace) and DiffSBDD (pocket translated every step) through
    BatchesInFlight.sample_many on three streams against one batch after the other, same noise: identical trajectories"""
    T = 8
    if name == "diffbp":
        Cn = 13
        m = C.get_model(C.default_diffbp_config(Cn, num_diffusion_timesteps=T)).eval()
        m.load_state_dict(W.synthetic_state_dict_diffbp(Cn, 9, seed=0, num_timesteps=T), strict=True)
    else:
        Cn = 8
        m = C.get_model(C.default_diffsbdd_config(Cn, num_diffusion_timesteps=T)).eval()
        m.load_state_dict(W.synthetic_state_dict_diffsbdd(Cn, 9, seed=0, num_timesteps=T), strict=True)
    m = m.to(DEV)
    batches = [synthetic.batch_to(synthetic.denovo_batch(n, seed=60 + n, num_classes=Cn), DEV) for n in (2, 3, 2)]
    g = torch.Generator(device=DEV).manual_seed(12)
    tapes = []
    for b in batches:
        n_lig = b["ligand_pos"].shape[0]
        if name == "diffbp":
            tapes.append({t: (torch.randn(n_lig, 3, device=DEV, generator=g), torch.rand(n_lig, device=DEV, generator=g)) for t in range(T)})
        else:      # the reference's draw order: initial x, initial c, then (x, c) per step -- more than enough tensors of each shape
            tapes.append([torch.randn(n_lig, 3 if k % 2 == 0 else Cn, device=DEV, generator=g) for k in range(2 * T + 4)])
    if name == "diffbp":
        one = [m.sample(b, noise_tape=tp) for b, tp in zip(batches, tapes)]
    else:
        one = [m.sample(b, noise_draws=list(tp)) for b, tp in zip(batches, tapes)]
    many = m.sample_many(batches, noise_tapes=[list(tp) if name == "diffsbdd" else tp for tp in tapes], streams=3)
    torch.cuda.synchronize()
    for a, b in zip(one, many):
        assert sorted(a.keys()) == sorted(b.keys())
        for t in a:
            assert torch.equal(a[t][0], b[t][0]) and torch.equal(a[t][1], b[t][1])


def test_scheduling_hints_do_not_change_results(model):
    """cbgx_set_edge_workgroups (CUs the persistent x2h edge kernel may take) and the number of caller streams are scheduling
    only: a denoiser call gives the same bits with 64 workgroups as with all, and on five different caller streams (the library
    keeps an auxiliary stream for four of them and replaces the oldest entry for the fifth)"""
    from cbgbench_amd import _native
    x, h, batch_idx, lig_flag, gen, gp = _composed(model, synthetic.denovo_batch(4, seed=77))
    lib = _native.lib()
    with torch.no_grad():
        ref = model.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
        assert lib.cbgx_set_edge_workgroups(64) == 0
        try:
            lim = model.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp)
        finally:
            assert lib.cbgx_set_edge_workgroups(0) == 64
        outs = []
        cur = torch.cuda.current_stream()
        for _ in range(5):
            sx = torch.cuda.Stream()
            sx.wait_stream(cur)
            with torch.cuda.stream(sx):
                outs.append(model.denoiser(x=x, h=h, batch_idx=batch_idx, lig_flag=lig_flag, gen_flag=gen, graph_ptr=gp))
        torch.cuda.synchronize()
    for o in [lim] + outs:
        assert all(torch.equal(a, b) for a, b in zip(ref, o))


def test_repacking_is_deterministic_and_follows_the_weights(synthetic_sd):
    """cbgx_pack_weights (batched over the attention blocks): packing the same parameters twice gives the same blob, changing one
    tensor changes it, and restoring the tensor restores it bit for bit"""
    m = C.get_model(C.default_targetdiff_config(13)).eval()
    m.load_state_dict(synthetic_sd, strict=True)
    m = m.to(DEV)
    dev = torch.device(DEV)
    a = m.denoiser.packed_weights(dev).clone()
    m.denoiser._packed = None
    b = m.denoiser.packed_weights(dev).clone()
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    p = m.denoiser.blocks[3].x2h_layers[0].hq_func.net[0].weight
    with torch.no_grad():
        saved = p.clone()
        p.mul_(1.5)
    c = m.denoiser.packed_weights(dev).clone()          # the in-place edit bumped the version: re-packed
    assert not torch.equal(a, c)
    with torch.no_grad():
        p.copy_(saved)
    assert torch.equal(m.denoiser.packed_weights(dev), a)


def test_sampling_driver_end_to_end(tmp_path):
    """config YAML -> registry -> model.sample on sharded pockets -> one result file per pocket (the sample.py role)."""
    import os as _os
    from cbgbench_amd import sample_cli
    cfg = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "fixtures", "targetdiff_T20.yml")
    rc = sample_cli.main(["--config", cfg, "--out_root", str(tmp_path), "--synthetic", "3", "--pockets_per_batch", "2",
                          "--random_init"])
    assert rc == 0
    files = sorted(_os.listdir(tmp_path / "targetdiff_T20"))
    assert files == ["pocket_00000.pt", "pocket_00001.pt", "pocket_00002.pt"]
    rec = torch.load(tmp_path / "targetdiff_T20" / files[1], weights_only=False)
    assert len(rec["samples"]) == 4
    for s in rec["samples"]:
        assert s["pos"].shape[1] == 3 and torch.isfinite(s["pos"]).all()
        assert s["atom_type"].min() >= 0 and s["atom_type"].max() < 13
        assert len(s["atom"]) == s["pos"].shape[0] == len(s["aromatic"]) and set(s["atom"]) <= {1, 6, 7, 8, 9, 15, 16, 17}


def test_sampling_driver_context_task_end_to_end(tmp_path):
    """A linker-style config through the driver: context atoms from the pocket file, frame centred on their
    mean (center_pos with mask_flag ctx_flag), generated atoms appended (assign_gensize), T = 20 steps -- the context atoms come
    back BIT-IDENTICAL in the pocket file's own frame positions up to the one fp32 translate round trip, their types unchanged,
    and the generated atoms moved."""
    import os as _os
    import numpy as _np
    from cbgbench_amd import sample_cli, synthetic as S
    rng = _np.random.default_rng(21)
    raw = []
    for k in range(3):
        pos, feat, aa = S.make_pocket(rng, int(rng.integers(300, 420)))
        off = rng.standard_normal(3).astype(_np.float32) * 6.0          # raw frame: far from the origin
        cpos, ctyp = S.make_context(rng, int(rng.integers(8, 20)))
        raw.append({"protein_pos": pos + off, "protein_atom_feature": feat, "protein_aa_type": aa,
                    "ligand_ctx_pos": cpos + off, "ligand_ctx_atom_type": ctyp})
    pfile = str(tmp_path / "pockets.pt")
    torch.save(raw, pfile)
    cfg = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "fixtures", "linker_targetdiff_T20.yml")
    for extra, frame in (([], "raw"), (["--no_translate"], "centred")):
        out = tmp_path / frame
        rc = sample_cli.main(["--config", cfg, "--out_root", str(out), "--pockets", pfile, "--pockets_per_batch", "2",
                              "--random_init"] + extra)
        assert rc == 0
        files = sorted(_os.listdir(out / "linker_targetdiff_T20"))
        assert files == ["pocket_00000.pt", "pocket_00001.pt", "pocket_00002.pt"]
        for k, f in enumerate(files):
            rec = torch.load(out / "linker_targetdiff_T20" / f, weights_only=False)
            assert len(rec["samples"]) == 3
            cpos, ctyp = torch.from_numpy(raw[k]["ligand_ctx_pos"]), torch.from_numpy(raw[k]["ligand_ctx_atom_type"])
            centre = cpos.mean(dim=0)
            for smp in rec["samples"]:
                c = cpos.shape[0]
                gen = smp["gen_flag"]
                assert gen.tolist() == [False] * c + [True] * (len(gen) - c) and len(gen) > c
                assert smp["type"][:c].tolist() == ctyp.tolist()                     # context types untouched by 20 steps
                want = (cpos - centre) + centre if frame == "raw" else cpos - centre  # exactly the arithmetic of the round trip
                assert torch.equal(smp["pos"][:c], want)                             # context atoms never move: bit-identical
                assert torch.isfinite(smp["pos"]).all()
                far = (smp["pos"][c:] - (centre if frame == "raw" else 0.0)).norm(dim=-1)
                assert float(far.max()) < 60.0                                       # generated atoms live around the context
    # the driver refuses a context config without context atoms
    for r in raw:
        r.pop("ligand_ctx_pos"); r.pop("ligand_ctx_atom_type")
    torch.save(raw, pfile)
    with pytest.raises(SystemExit, match="context task"):
        sample_cli.main(["--config", cfg, "--out_root", str(tmp_path / "x"), "--pockets", pfile, "--random_init"])


def test_step_boundary_kernel_equals_epilogue_then_prologue(model):
    """cbgx_targetdiff_step_boundary (epilogue of step t + composed rows of step t - 1 in one launch, types one class per lane)
    against the two kernels it replaces: identical ligand states AND identical composed inputs, bit for bit, over a run of steps
    that includes a linker-style batch (context atoms keep position and type) and an externally edited state (falls back)."""
    for batch in (synthetic.denovo_batch(4, seed=31, n_rec_range=(200, 320)), synthetic.linker_batch(3, seed=32, n_rec_range=(200, 320))):
        batch = synthetic.batch_to(batch, DEV)
        n_lig = batch["ligand_pos"].shape[0]
        g = torch.Generator(device=DEV).manual_seed(17)
        ts = (999, 998, 997, 500, 499, 2, 1, 0)
        noise = [(torch.randn(n_lig, 3, device=DEV, generator=g), torch.rand(n_lig, 13, device=DEV, generator=g)) for _ in ts]
        runs = []
        for fused in (True, False):
            model.fuse_step_boundary = fused
            try:
                st = model.begin_sampling(batch, keep_trajectory=False)
                states = []
                for k, t in enumerate(ts):
                    if k == 4:      # a caller that edits the state between steps: the composed rows must be rebuilt from it
                        st["x_lig"] = st["x_lig"] + 0.25
                    model.denoise_step(st, t, noise=noise[k])
                    states.append((st["x_lig"].clone(), st["c_lig"].clone(), st["x"].clone(), st["h"].clone()))
                runs.append(states)
            finally:
                model.fuse_step_boundary = True
        for k, (a, b) in enumerate(zip(*runs)):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), (k, "ligand state")
            if ts[k] > 0 and k + 1 < len(ts) and k != 3:      # the fused run has already composed the next step's rows
                rows = st["lig_rows"]
                assert torch.equal(a[2][rows], a[0]), (k, "composed x")
        # and the fused run's composed features equal what the prologue kernel writes from the same state
        st2 = model.begin_sampling(batch, keep_trajectory=False)
        model.denoise_step(st2, 999, noise=noise[0])
        h_fused = st2["h"].clone()
        st2["_composed"] = None
        st3 = model.begin_sampling(batch, keep_trajectory=False)
        model.fuse_step_boundary = False
        try:
            model.denoise_step(st3, 999, noise=noise[0])
            model.denoise_step(st3, 998, noise=noise[1])     # its prologue composes the state the fused kernel composed above
        finally:
            model.fuse_step_boundary = True
        model.denoise_step(st2, 998, noise=noise[1])
        assert torch.equal(st2["x_lig"], st3["x_lig"]) and torch.equal(st2["c_lig"], st3["c_lig"])
        assert torch.isfinite(h_fused).all()


def test_static_context_cache_is_exact(model):
    """the ligand-free cache of the first two layers (cbgx_unitransformer_forward_cached) must not change a single bit of
    a sampling step, on real-size pockets where most protein atoms are far from the ligand"""
    batch = synthetic.batch_to(synthetic.denovo_batch(5, seed=77), DEV)
    n_lig = batch["ligand_pos"].shape[0]
    g = torch.Generator(device=DEV).manual_seed(3)
    noise = [(torch.randn(n_lig, 3, device=DEV, generator=g), torch.rand(n_lig, 13, device=DEV, generator=g)) for _ in range(3)]
    outs = []
    for cache in (True, False):
        st = model.begin_sampling(batch, keep_trajectory=False, static_cache=cache)
        assert (st["static_h"] is not None) == cache
        for k, t in enumerate((999, 998, 400)):
            model.denoise_step(st, t, noise=noise[k])
        outs.append((st["x_lig"].clone(), st["c_lig"].clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    # and the cache really prunes: count the rows that differ from the ligand-free pocket after one / two hops
    st = model.begin_sampling(batch, keep_trajectory=False)
    nbr, deg = stages.knn_graph(st["x"], st["graph_ptr"])
    lig = st["lig_flag"]
    has_lig_nbr = (lig[nbr.clamp(min=0).long()] & (nbr >= 0)).any(1) | lig
    assert 0.02 < float(has_lig_nbr.float().mean()) < 0.6


def test_static_context_cache_is_exact_on_odd_graphs(model):
    """the same bit-identity on the graphs that take the other branches of round 4's list / kNN code: a pocket smaller than the
    neighbour count (pocket lists shorter than 32), 70 ligand atoms (second candidate slot of the rank-counting kNN merge), 130
    (more than it handles: the scan), 900 pocket atoms (above the register-cached kNN size), a 2-atom ligand, and the whole thing
    again as a linker batch (ligand atoms that cannot move).  Three steps with and without the cache; also every neighbour list the
    cached call builds must equal the full search's (checked through the outputs: one swapped neighbour changes them)."""
    rng = np.random.default_rng(123)
    sizes = [(20, 5), (300, 70), (260, 130), (900, 12), (400, 2), (33, 40)]
    pockets = [synthetic.make_pocket(rng, n) for n, _ in sizes]
    for n_ctx in (None, [2, 30, 60, 5, 1, 10]):
        batch = synthetic.make_batch(pockets, [m for _, m in sizes], rng, 13, n_ctx_list=n_ctx)
        batch = synthetic.batch_to(batch, DEV)
        n_lig = batch["ligand_pos"].shape[0]
        g = torch.Generator(device=DEV).manual_seed(4)
        noise = [(torch.randn(n_lig, 3, device=DEV, generator=g), torch.rand(n_lig, 13, device=DEV, generator=g)) for _ in range(3)]
        outs = []
        for cache in (True, False):
            st = model.begin_sampling(batch, keep_trajectory=False, static_cache=cache)
            assert (st["static_h"] is not None) == cache
            for k, t in enumerate((999, 500, 3)):
                model.denoise_step(st, t, noise=noise[k])
            outs.append((st["x_lig"].clone(), st["c_lig"].clone()))
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]), "linker" if n_ctx else "denovo"
        assert bool(torch.isfinite(outs[0][0]).all())
        # and one step of the same batch against the CPU oracle (the reference's formulation)
        cpu = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in batch.items()}
        st = model.begin_sampling(batch, keep_trajectory=False)
        model.denoise_step(st, 500, noise=noise[1])
        c0 = torch.nn.functional.one_hot(cpu["ligand_atom_type"], 13).float()
        x_ref, c_ref = OT.denoise_step(model_sd(model), cpu, cpu["ligand_pos"], c0, 500, noise[1][0].cpu(), noise[1][1].cpu(), 13)
        close(st["x_lig"], x_ref, "x_{t-1} on odd graphs")
        assert torch.equal(st["c_lig"].cpu().argmax(-1), c_ref.argmax(-1))


_NODE_STAGE_PROBE = r"""
import hashlib, sys, torch
import cbgbench_amd as C
from cbgbench_amd import synthetic
from oracle import weights as W
m = C.get_model(C.default_targetdiff_config(13)).eval()
m.load_state_dict(W.synthetic_state_dict(13, 9, seed=0), strict=True)
m = m.to("cuda:0")
out = []
for n_graphs, ctx in ((1, None), (4, None), (4, [3, 0, 9, 5])):
    batch = synthetic.denovo_batch(n_graphs, seed=11) if ctx is None else None
    if batch is None:
        import numpy as np
        rng = np.random.default_rng(5)
        batch = synthetic.make_batch([synthetic.make_pocket(rng, n) for n in (200, 350, 90, 420)], [20, 12, 30, 25], rng, 13, n_ctx_list=ctx)
    batch = synthetic.batch_to(batch, "cuda:0")
    n_lig = batch["ligand_pos"].shape[0]
    g = torch.Generator(device="cuda:0").manual_seed(9)
    st = m.begin_sampling(batch, keep_trajectory=False)
    for t in (999, 998, 2):
        m.denoise_step(st, t, noise=(torch.randn(n_lig, 3, device="cuda:0", generator=g), torch.rand(n_lig, 13, device="cuda:0", generator=g)))
    x, h, gp = st["x"], st["h"], st["graph_ptr"]
    with torch.no_grad():
        full = m.denoiser(x=x, h=h, batch_idx=st["batch_idx"], lig_flag=st["lig_flag"], gen_flag=st["gen_flag"], graph_ptr=gp)
    torch.cuda.synchronize()
    for t_ in (st["x_lig"], st["c_lig"]) + tuple(full):
        out.append(hashlib.sha256(t_.detach().cpu().contiguous().numpy().tobytes()).hexdigest()[:16])
print("PROBE " + " ".join(out))
"""


def test_node_stage_variants_give_the_same_bits():
    """node_stage_kernel<16> (one 16-wave workgroup per CU), <8> (two per CU, every wave two or three units / two heads in
    sequence: the default) and <4> differ in which wave computes a column, not in how: three sampling steps (cached forward,
    partial gen_flag included) and one plain forward with each variant forced through CBGX_NODE_STAGE_WAVES must give identical
    bits."""
    import subprocess
    import sys
    outs = []
    for waves in ("16", "8", "4"):
        env = dict(os.environ, CBGX_NODE_STAGE_WAVES=waves)
        r = subprocess.run([sys.executable, "-c", _NODE_STAGE_PROBE], capture_output=True, text=True, timeout=600, env=env,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln.startswith("PROBE ")][-1])
    assert outs[0] == outs[1] == outs[2] and len(outs[0].split()) == 1 + 3 * 5


def test_graph_replay_equals_eager_steps(golden_dir):
    """one captured hipGraph replayed per step (TargetDiff.make_step_graph) against eager steps fed the noise the graph drew"""
    T = 12
    m = C.get_model(C.default_targetdiff_config(13, num_diffusion_timesteps=T)).eval()
    m.load_state_dict(W.synthetic_state_dict(13, 9, seed=0, num_timesteps=T), strict=True)
    m = m.to(DEV)
    batch = synthetic.batch_to(synthetic.denovo_batch(3, seed=5, n_rec_range=(150, 260)), DEV)
    torch.manual_seed(11)
    st = m.begin_sampling(batch, keep_trajectory=True)
    x0, c0 = st["x_lig"].clone(), st["c_lig"].clone()
    replay, done = m.make_step_graph(st, warmup=2)
    tape = {}
    # the two warm-up steps ran eagerly on the trajectory-resident kernels; recover their noise from the recorded states is
    # not possible, so the comparison starts from the state after them
    x_start, c_start = st["traj_x"][T - done].clone(), st["traj_c"][T - done].clone()
    for t in reversed(range(T - done)):
        replay()
        torch.cuda.synchronize()
        tape[t] = (st["_noise"][0].clone(), st["_noise"][1].clone())
    assert int(st["t_dev"].item()) == -1
    x_g, c_g = st["traj_x"][0].clone(), st["traj_c"][0].clone()
    st2 = m.begin_sampling(batch, keep_trajectory=True)
    st2["x_lig"], st2["c_lig"] = x_start, c_start
    for t in reversed(range(T - done)):
        m.denoise_step(st2, t, noise=tape[t])
    assert torch.equal(st2["x_lig"], x_g) and torch.equal(st2["c_lig"], c_g)
    assert torch.isfinite(x_g).all() and not torch.equal(x_g, x0)
    # and sample(use_graph=True) returns the full trajectory
    traj = m.sample(batch, use_graph=True)
    assert sorted(traj.keys()) == list(range(-1, T)) and torch.isfinite(traj[-1][0]).all()


def test_static_context_cache_edge_cases(model):
    """cache exactness where its branches differ: a graph above the register-cached kNN size (900 atoms), a pocket with
    fewer than 33 atoms (every atom keeps a fresh neighbour list), a graph without any ligand atom (fully cached)"""
    rng = np.random.default_rng(123)
    pockets = [synthetic.make_pocket(rng, 900, radius=14.0), synthetic.make_pocket(rng, 60, radius=6.0),
               synthetic.make_pocket(rng, 20, radius=4.0), synthetic.make_pocket(rng, 400)]
    batch = synthetic.batch_to(synthetic.make_batch(pockets, [12, 0, 5, 30], rng, 13), DEV)
    n_lig = batch["ligand_pos"].shape[0]
    g = torch.Generator(device=DEV).manual_seed(8)
    noise = [(torch.randn(n_lig, 3, device=DEV, generator=g), torch.rand(n_lig, 13, device=DEV, generator=g)) for _ in range(2)]
    outs = []
    for cache in (True, False):
        st = model.begin_sampling(batch, keep_trajectory=False, static_cache=cache)
        for k, t in enumerate((999, 3)):
            model.denoise_step(st, t, noise=noise[k])
        outs.append((st["x_lig"].clone(), st["c_lig"].clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][0]).all()


def test_zz_report_measured_errors():
    """not a check: writes what the checks of this file measured (max |err| per output over all goldens, against the tolerance
    in force) to gpurun_out/parity_errors.md, so that the tolerances can be read against measurements (copied to profiles/)"""
    root = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = os.path.join(root, "gpurun_out")
    if not MEASURED or not os.path.isdir(out):
        pytest.skip("nothing measured in this session, or no gpurun_out/")
    lines = ["| output (all goldens of tests/test_gpu_parity.py) | max abs err | max abs ref | worst err / tolerance |", "|---|---|---|---|"]
    for k in sorted(MEASURED):
        e, r, f = MEASURED[k]
        lines.append(f"| {k} | {e:.3e} | {r:.3e} | {f:.3f} |")
    with open(os.path.join(out, "parity_errors.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    print("\n".join(lines))
