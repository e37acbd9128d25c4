"""The CPU oracle against golden vectors produced by the reference's own code
(oracle/make_golden.py).  fp32 CPU both sides, same op order -> bit-exact."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import targetdiff as T
from oracle import unitransformer as U
from oracle import weights as W

DENOISER_CASES = ["denoiser_2graphs", "denoiser_small_graphs", "denoiser_linker", "denoiser_eg5_pocket10",
                  "denoiser_adrb1_pocket10", "denoiser_drd2_pocket10", "denoiser_smarca2_pocket10"]


def load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


@pytest.mark.parametrize("case", DENOISER_CASES)
def test_denoiser_matches_reference(golden_dir, synthetic_sd, case):
    g = load(golden_dir, case)
    x, h, logits, inter = U.unitransformer_forward(
        synthetic_sd, g["x"], g["h"], g["batch_idx"], g["lig_flag"], g["gen_flag"], return_intermediates=True)
    assert torch.equal(inter["edge_index"].int(), g["edge_index"])
    assert torch.equal(inter["edge_type"].to(torch.int8), g["edge_type"])
    assert torch.equal(inter["e_w"], g["e_w"])
    assert torch.equal(inter["layers"][0][0], g["x_layer0"])
    assert torch.equal(inter["layers"][0][1], g["h_layer0"])
    assert torch.equal(x, g["x_out"]) and torch.equal(h, g["h_out"]) and torch.equal(logits, g["logits"])


def test_small_graph_degrees(golden_dir):
    g = load(golden_dir, "denoiser_small_graphs")
    deg = torch.bincount(g["edge_index"][1].long(), minlength=g["x"].shape[0])
    sizes = torch.bincount(g["batch_idx"])
    assert sizes.tolist() == [25, 33, 2, 34, 1]
    expect = torch.cat([torch.full((int(n),), min(32, int(n) - 1)) for n in sizes])
    assert torch.equal(deg, expect)


def test_schedule_tables_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "schedule_tables.npz"))
    sd = W.schedule_state(1000)
    assert set(z.files) == set(sd)
    for k in z.files:
        assert np.array_equal(z[k], sd[k].numpy()), k


@pytest.mark.parametrize("tag,C", [("add_aromatic", 13), ("basic", 8)])
def test_state_dict_keys_match_reference(golden_dir, tag, C):
    with open(os.path.join(golden_dir, f"state_dict_keys_{tag}.json")) as f:
        ref = json.load(f)
    sd = W.synthetic_state_dict(C, 9, seed=0)
    assert list(ref.keys()) == list(sd.keys()) or set(ref) == set(sd)
    for k, shp in ref.items():
        assert list(sd[k].shape) == shp, k


@pytest.mark.parametrize("case", ["step_t500", "step_t0", "step_t999_linker"])
def test_step_matches_reference(golden_dir, synthetic_sd, case):
    g = load(golden_dir, case)
    batch = {k[len("batch_"):]: v for k, v in g.items() if k.startswith("batch_")}
    c_lig = torch.nn.functional.one_hot(batch["ligand_atom_type"], 13).float()
    x_next, c_next, x_pred, c_pred = T.denoise_step(
        synthetic_sd, batch, batch["ligand_pos"], c_lig, int(g["t_idx"]), g["eps"], g["u"], 13, return_net_out=True)
    assert torch.equal(x_pred, g["x_pred"]) and torch.equal(c_pred, g["c_pred"])
    assert torch.equal(x_next, g["x_next"])
    assert torch.equal(c_next, g["c_next"])
    if "ligand_gen_flag" in batch:
        keep = ~batch["ligand_gen_flag"]
        assert torch.equal(x_next[keep], batch["ligand_pos"][keep])


def test_sample_loop_matches_reference(golden_dir):
    """Full TargetDiff.sample of a 5-step model: RNG call order (randn then rand per step),
    trajectory keys (traj[-1] is the final state) -- targetdiff.py:150-182."""
    g = load(golden_dir, "sample_T5")
    Tn = int(g["T"])
    sd = W.synthetic_state_dict(13, 9, seed=0, num_timesteps=Tn)
    batch = {k[len("batch_"):]: v for k, v in g.items() if k.startswith("batch_")}
    x = batch["ligand_pos"]
    c = torch.nn.functional.one_hot(batch["ligand_atom_type"], 13).float()
    torch.manual_seed(int(g["seed"]))
    for t in reversed(range(Tn)):
        assert torch.equal(x, g[f"traj_x_{t}"]) and torch.equal(c, g[f"traj_c_{t}"])
        eps = torch.randn_like(x)
        u = torch.rand(x.shape[0], 13)
        x, c = T.denoise_step(sd, batch, x, c, t, eps, u, 13)
    assert torch.equal(x, g["traj_x_-1"]) and torch.equal(c, g["traj_c_-1"])


def test_diffsbdd_sample_matches_reference(golden_dir):
    """oracle/diffsbdd.py replays the reference's full 5-step DiffSBDD.sample (draw order, COM projection, the
    translated pocket, the final p(x,h | z0) call) bit-exactly."""
    from oracle import diffsbdd as D
    g = load(golden_dir, "diffsbdd_sample_T5")
    Tn, C = int(g["T"]), 8
    sd = W.synthetic_state_dict_diffsbdd(C, 9, seed=0, num_timesteps=Tn)
    assert torch.equal(sd["pos_scheduler.gamma.gamma"], g["gamma"])
    batch = {k[len("batch_"):]: v for k, v in g.items() if k.startswith("batch_")}
    n_lig = batch["ligand_element_batch"].shape[0]
    torch.manual_seed(int(g["seed"]))
    draws = []
    for _ in range(Tn + 2):
        draws += [torch.randn(n_lig, 3), torch.randn(n_lig, C)]
    traj = D.sample(sd, batch, C, Tn, draws)
    for t in range(-1, Tn):
        assert torch.equal(traj[t][0], g[f"traj_x_{t}"]), t
        assert torch.equal(traj[t][1], g[f"traj_c_{t}"]), t


def test_diffsbdd_gamma_table_matches_reference(golden_dir):
    from oracle import diffsbdd as D
    z = np.load(os.path.join(golden_dir, "diffsbdd_gamma_T1000.npz"))
    assert np.array_equal(z["gamma"], D.polynomial_gamma(1000).numpy())


def test_diffbp_sample_matches_reference(golden_dir):
    """oracle/diffbp.py replays the reference's full 5-step DiffBP.sample (denoiser + CoMPredictor + score-type
    position step + absorbing-state type step) bit-exactly."""
    from oracle import diffbp as D
    g = load(golden_dir, "diffbp_sample_T5")
    Tn, C = int(g["T"]), 13
    sd = W.synthetic_state_dict_diffbp(C, 9, seed=0, num_timesteps=Tn)
    batch = {k[len("batch_"):]: v for k, v in g.items() if k.startswith("batch_")}
    x = batch["ligand_pos"]
    c = torch.nn.functional.one_hot(batch["ligand_atom_type"], C).float()
    torch.manual_seed(int(g["seed"]))
    for t in reversed(range(Tn)):
        assert torch.equal(x, g[f"traj_x_{t}"]) and torch.equal(c, g[f"traj_c_{t}"]), t
        eps = torch.randn_like(x)
        u = torch.rand(x.shape[0])
        x, c = D.denoise_step(sd, batch, x, c, t, eps, u, C, Tn)
    assert torch.equal(x, g["traj_x_-1"]) and torch.equal(c, g["traj_c_-1"])
    assert bool((c.argmax(-1) != 0).any()), "some atoms must have left the absorbing state"


def golden_batch(g):
    return {k[len("batch_"):]: v for k, v in g.items() if k.startswith("batch_")}


@pytest.mark.parametrize("case", ["train_loss_denovo", "train_loss_t0_linker"])
def test_training_loss_and_gradients_match_reference(golden_dir, synthetic_sd, case):
    """loss.backward() of the unmodified reference (train.py:185-189) vs autograd on the restatement."""
    from oracle import training as TR
    g = load(golden_dir, case)
    batch = golden_batch(g)
    if g["draws"].numel():
        B = int(batch["ligand_element_batch"].max()) + 1
        assert torch.equal(TR.sample_time_symmetric(B, 1000, g["draws"]), g["t"])
    losses, grads = TR.loss_and_grads(synthetic_sd, batch, g["t"], g["eps"], g["u"], 13)
    assert float(losses["pos"]) == g["loss_pos"] and float(losses["atom"]) == g["loss_atom"]
    n = 0
    for k, gr in grads.items():
        ref_norm = float(g["gnorm/" + k])
        assert abs(gr.double().norm().item() - ref_norm) <= 1e-5 * ref_norm + 1e-8, k
        flat = gr.reshape(-1)
        sample = flat if flat.numel() <= 2048 else flat[::61]
        torch.testing.assert_close(sample, g["g/" + k], rtol=1e-4, atol=1e-8 + 1e-5 * ref_norm / max(flat.numel(), 1) ** 0.5)
        n += 1
    assert n == 8 + 6 + 9 * 36 + 4     # embedder, gate MLP, 9 layers x 6 MLPs x 6 tensors, classifier
    # the key bias of every attention cannot influence a softmax over the edges of one node
    assert float(g["gnorm/denoiser.blocks.3.x2h_layers.0.hk_func.net.3.bias"]) < 1e-6


def test_diffbp_training_loss_and_gradients_match_reference(golden_dir):
    """DiffBP training objective (diffbp.py:154-231 incl. CoMPredictor and interior_loss): the reference's four losses and
    the gradients of all 400+ tensors vs autograd on the restatement"""
    from oracle import diffbp as OD
    g = load(golden_dir, "train_loss_diffbp")
    batch = golden_batch(g)
    sd = W.synthetic_state_dict_diffbp(13, 9, seed=0, num_timesteps=1000)
    losses, grads = OD.loss_and_grads(sd, batch, g["t"], g["eps"], g["u"], 13, 1000)
    for k in ("pos", "atom", "com", "inter"):
        assert abs(float(losses[k]) - g["loss_" + k]) <= 1e-6 * abs(g["loss_" + k]) + 1e-7, k
    n = 0
    for k, gr in grads.items():
        ref_norm = float(g["gnorm/" + k])
        assert abs(gr.double().norm().item() - ref_norm) <= 1e-5 * ref_norm + 1e-8, k
        flat = gr.reshape(-1)
        sample = flat if flat.numel() <= 2048 else flat[::61]
        torch.testing.assert_close(sample, g["g/" + k], rtol=1e-4, atol=1e-8 + 1e-5 * ref_norm / max(flat.numel(), 1) ** 0.5)
        n += 1
    assert n == 8 + 6 + 9 * 36 + 4 + (6 + 3 * 18)


@pytest.mark.parametrize("case", ["train_loss_diffsbdd", "train_loss_diffsbdd_t0"])
def test_diffsbdd_training_loss_and_gradients_match_reference(golden_dir, case):
    """DiffSBDD's variational training loss (diffsbdd.py:91-195): loss_t, the t = 0 reconstruction terms and the KL
    prior, against the unmodified reference's losses and gradients"""
    from oracle import diffsbdd as OS
    g = load(golden_dir, case)
    batch = golden_batch(g)
    sd = W.synthetic_state_dict_diffsbdd(8, 9, seed=0, num_timesteps=1000)
    losses, grads = OS.loss_and_grads(sd, batch, g["t"], g["eps_x"], g["eps_c"], 8, 1000)
    for k in ("pos", "atom"):
        assert abs(float(losses[k]) - g["loss_" + k]) <= 2e-6 * abs(g["loss_" + k]) + 1e-7, (k, float(losses[k]), g["loss_" + k])
    n = 0
    for k, gr in grads.items():
        ref_norm = float(g["gnorm/" + k])
        assert abs(gr.double().norm().item() - ref_norm) <= 1e-5 * ref_norm + 1e-8, k
        flat = gr.reshape(-1)
        sample = flat if flat.numel() <= 2048 else flat[::61]
        torch.testing.assert_close(sample, g["g/" + k], rtol=1e-4, atol=1e-8 + 1e-5 * ref_norm / max(flat.numel(), 1) ** 0.5)
        n += 1
    assert n == 8 + 6 + 9 * 36 + 4


def test_diffsbdd_eval_loss_matches_reference(golden_dir):
    """DiffSBDD.forward in eval mode (diffsbdd.py:72-86): the variational bound with the SNR-weighted loss_t, the KL prior
    and the t = 0 reconstruction term from a second denoiser call, averaged over the evaluation times"""
    from oracle import diffsbdd as OS
    g = load(golden_dir, "eval_loss_diffsbdd")
    batch = golden_batch(g)
    sd = W.synthetic_state_dict_diffsbdd(8, 9, seed=0, num_timesteps=1000)
    K = int(g["eval_interval"])
    draws = [tuple(g[f"{tag}_{k}"] for tag in ("eps_x", "eps_c", "eps_x0", "eps_c0")) for k in range(K)]
    assert OS.eval_times(1000, K) == [1, 500, 1000]
    with torch.no_grad():
        losses = OS.forward_eval(sd, batch, draws, 8, 1000, eval_interval=K)
    for k in ("pos", "atom"):
        assert abs(float(losses[k]) - g["loss_" + k]) <= 5e-6 * abs(g["loss_" + k]) + 1e-6, (k, float(losses[k]), g["loss_" + k])
