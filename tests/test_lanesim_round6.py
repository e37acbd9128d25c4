"""Index-level CPU models of round 6's host-side / node-level kernels: what every thread reads and writes, restated in numpy and checked
against the tensor operations the kernels replace.  They pin the index math where no GPU is needed; tests/test_gpu_training.py compares
the kernels themselves (test_native_compose_plan_equals_the_tensor_path, test_embed_compose_matches_the_tensor_path, the gradient tests).
  compose_count / compose_scan / compose_place   cbgbench_amd/csrc/train_embed.hip  (compose_context, repo/modules/common.py:189-214)
  embed_compose_backward's slices                cbgbench_amd/csrc/train_embed.hip  (PLContextEmbedder, repo/modules/context_emb.py:137-230)
  fold_grad_kernel's thread mapping              cbgbench_amd/csrc/train_reduce.hip (x2h_attention.py:56-91, the second v Linear)"""
import numpy as np
import pytest
import torch


def compose_plan_model(br, bl, B):
    """the three launches of launch_compose_plan, thread by thread"""
    n_rec, n_lig = len(br), len(bl)
    cnt = np.zeros(2 * B, np.int64)
    flag = 0
    for i in range(n_rec + n_lig):                       # compose_count_kernel (the wave-level aggregation only batches these adds)
        lig = i >= n_rec
        a, k = (bl, i - n_rec) if lig else (br, i)
        g = a[k]
        assert 0 <= g < B
        cnt[(B if lig else 0) + g] += 1
        if k > 0 and a[k - 1] > g:
            flag |= 2 if lig else 1
    pre = np.zeros(2 * B, np.int64)                      # compose_scan_kernel: exclusive prefix sums per array
    pre[:B] = np.cumsum(cnt[:B]) - cnt[:B]
    pre[B:] = np.cumsum(cnt[B:]) - cnt[B:]
    graph_ptr = np.concatenate([pre[:B] + pre[B:], [cnt.sum()]])
    N = n_rec + n_lig
    sort_idx, batch_idx = np.full(N, -1, np.int64), np.full(N, -1, np.int64)
    lig_flag, lig_rows = np.zeros(N, bool), np.full(n_lig, -1, np.int64)
    for i in range(N):                                   # compose_place_kernel
        lig = i >= n_rec
        a, k = (bl, i - n_rec) if lig else (br, i)
        g = a[k]
        if flag & (2 if lig else 1):
            rank = int((a[:k] == g).sum())               # the slow, literal rank
        else:
            rank = k - pre[(B if lig else 0) + g]
        pos = pre[g] + pre[B + g] + (cnt[g] if lig else 0) + rank
        assert sort_idx[pos] == -1                       # every composed row has exactly one writer
        sort_idx[pos], batch_idx[pos], lig_flag[pos] = i, g, lig
        if lig:
            lig_rows[k] = pos
    return sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr


@pytest.mark.parametrize("kind", ["collated", "empty_graphs", "unsorted_ligand", "unsorted_both", "no_protein"])
def test_compose_plan_counting_sort_is_the_stable_argsort(kind):
    rng = np.random.default_rng(3)
    B, n_rec, n_lig = 7, (0 if kind == "no_protein" else 90), 31
    br, bl = rng.integers(0, B, n_rec), rng.integers(0, B, n_lig)
    if kind == "empty_graphs":
        br[br == 2] = 3
        bl[bl == 2] = 1
        bl[bl == 5] = 6
    if kind in ("collated", "empty_graphs", "no_protein", "unsorted_ligand"):
        br = np.sort(br)
    if kind in ("collated", "empty_graphs", "no_protein"):
        bl = np.sort(bl)
    sort_idx, batch_idx, lig_flag, lig_rows, graph_ptr = compose_plan_model(br, bl, B)
    ctx = torch.from_numpy(np.concatenate([br, bl]))
    ref = torch.sort(ctx, stable=True).indices.numpy()   # compose_context's own sort
    assert np.array_equal(sort_idx, ref)
    assert np.array_equal(batch_idx, ctx.numpy()[ref]) and np.array_equal(lig_flag, ref >= n_rec)
    inv = np.empty_like(ref)
    inv[ref] = np.arange(len(ref))
    assert np.array_equal(lig_rows, inv[n_rec:])
    assert np.array_equal(graph_ptr, np.concatenate([[0], np.cumsum(np.bincount(ctx.numpy(), minlength=B))]))


def test_embedder_gradients_are_column_slices_of_one_weight_gradient():
    """cbgx_embed_compose_backward: dWext[c][j] = sum_r dh[r][c] ext[r][j] over the extended rows [feat | onehot aa | 1 | c | 1] holds
    all eight parameter gradients -- dW_pa, dW_res, u, dW_la, v with db_pa = db_res = u, db_la = dW_ind[:, 0] = v, db_ind = u + v --
    against autograd on the three Linears + indicator of the embedder."""
    torch.manual_seed(5)
    F_, A, C, E = 7, 20, 13, 128
    n_rec, n_lig = 40, 9
    lin = lambda i: torch.nn.Linear(i, E)
    pa, res, la, ind = lin(F_), lin(A), lin(C), lin(1)
    feat, aa, c = torch.rand(n_rec, F_), torch.randint(0, A, (n_rec,)), torch.rand(n_lig, C)
    perm = torch.randperm(n_rec + n_lig)                                   # any composition order
    h_rec = pa(feat) + res(torch.nn.functional.one_hot(aa, A).float()) + ind.bias
    h_lig = la(c) + (ind.weight[:, 0] + ind.bias)
    h = torch.cat([h_rec, h_lig])[perm]
    gh = torch.randn_like(h)
    (h * gh).sum().backward()
    ext = torch.zeros(n_rec + n_lig, 128)
    ext[:n_rec, :F_], ext[:n_rec, F_ + A] = feat, 1.0
    ext[torch.arange(n_rec), F_ + aa] = 1.0
    ext[n_rec:, F_ + A + 1:F_ + A + 1 + C], ext[n_rec:, F_ + A + 1 + C] = c, 1.0
    dwext = gh.T @ ext[perm]                                               # [128 c][128 j]: wgrad_mfma_kernel + the reduce
    dw_pa, dw_res, u = dwext[:, :F_], dwext[:, F_:F_ + A], dwext[:, F_ + A]
    dw_la, v = dwext[:, F_ + A + 1:F_ + A + 1 + C], dwext[:, F_ + A + 1 + C]
    close = lambda a, b: torch.allclose(a, b, atol=2e-5, rtol=1e-5)
    assert close(dw_pa, pa.weight.grad) and close(u, pa.bias.grad) and close(dw_res, res.weight.grad) and close(u, res.bias.grad)
    assert close(dw_la, la.weight.grad) and close(v, la.bias.grad) and close(v, ind.weight.grad[:, 0]) and close(u + v, ind.bias.grad)
    assert float(dwext[:, F_ + A + C + 2:].abs().max()) == 0.0           # the padding columns stay zero


def test_fold_grad_thread_mapping_covers_every_output_once():
    """fold_grad_kernel (round 6): thread t = (mq = t & 31, ag = t >> 5) of a 256-thread workgroup owns the columns 4 mq .. 4 mq + 3 of
    the heads 2 ag, 2 ag + 1 of every row of the tile; gb comes from thread (r = t >> 4, a = t & 15).  Every element of Gt [16 a][128 m]
    and gb [16 a] of a row has exactly one writer, and the value is the fold through the second v Linear."""
    rng = np.random.default_rng(0)
    H, HEADS, DH = 128, 16, 8
    G = rng.standard_normal((16, H)).astype(np.float32)                   # one tile of output gradients
    Wbv = rng.standard_normal((H, H)).astype(np.float32)                  # x2h layout [m][n]
    bbv = rng.standard_normal(H).astype(np.float32)
    Gt, writers = np.zeros((16, HEADS, H), np.float32), np.zeros((16, HEADS, H), np.int32)
    gb, gb_writers = np.zeros((16, HEADS), np.float32), np.zeros((16, HEADS), np.int32)
    for t in range(256):
        mq, ag = t & 31, t >> 5
        for r in range(16):
            for hh in range(2):
                a = 2 * ag + hh
                for j in range(4):
                    m = 4 * mq + j
                    s = np.float32(0)
                    for cc in range(DH):
                        s = np.float32(G[r, a * DH + cc] * Wbv[m, a * DH + cc] + s)
                    Gt[r, a, m] = s
                    writers[r, a, m] += 1
        r, a = t >> 4, t & 15
        gb[r, a] = np.float32(sum(G[r, a * DH + cc] * bbv[a * DH + cc] for cc in range(DH)))
        gb_writers[r, a] += 1
    assert (writers == 1).all() and (gb_writers == 1).all()
    ref = np.einsum("rac,mac->ram", G.reshape(16, HEADS, DH), Wbv.reshape(H, HEADS, DH))
    assert np.allclose(Gt, ref, atol=1e-5)
    assert np.allclose(gb, (G * bbv).reshape(16, HEADS, DH).sum(-1), atol=1e-5)
