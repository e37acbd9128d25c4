"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the equivariant message-passing path.

A plain-torch (CPU, fp32 or fp64) restatement of the reference's algorithm, in
the reference's own formulation (materialised ``[E,340]`` edge inputs, separate
``scatter_softmax`` / ``scatter_sum``), driven by a reference-format
``state_dict``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this; the product path
(``cbgbench_amd``) never does.

Parity status: the reference has NO tests / golden vectors for this path
(SURVEY.md section 4 and 8c), so the oracle is pinned the only way available:
against the reference's own modules run in the build container through
``oracle/ref_shim.py`` (``tests/golden/*.npz`` made by ``oracle/make_golden.py``,
re-checked by ``tests/test_oracle_golden.py``).  The two third-party natives the
reference calls (torch_cluster.knn_graph, torch_scatter.scatter_softmax/sum;
unpinned, README.MD:57-58) are restated from their published semantics.

Every function cites the reference lines it follows (paths relative to
/root/reference).
"""
import math

import torch
import torch.nn.functional as F

# repo/modules/common.py:122 -- the 20 hard-coded Gaussian centres (fixed_offset=True)
RBF_OFFSETS = (0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10)
# repo/modules/common.py:125 -- coeff = -0.5 / (offset[1]-offset[0])**2 = -0.5
RBF_COEFF = -0.5


def knn_graph(x, batch, k):
    """torch_cluster.knn_graph(x, k, batch, loop=False, flow='source_to_target')
    as called at repo/modules/e3nn/unitransformer.py:80.

    Returns edge_index [2,E] int64 with row 0 = neighbour j (src) and row 1 =
    centre i (dst), grouped by centre in ascending centre order; neighbours in
    ascending (squared distance, index) order.  A graph with n <= k nodes yields
    n-1 neighbours per node.  Squared distances are ((dx*dx)+(dy*dy))+(dz*dz) in
    the dtype of x with no fused multiply-add, so the HIP kernel can reproduce
    the ordering bit-for-bit.
    """
    N = x.shape[0]
    src_all, dst_all = [], []
    if N == 0:
        return torch.zeros(2, 0, dtype=torch.long)
    batch = batch.to(torch.long)
    starts = torch.nonzero(torch.cat([torch.ones(1, dtype=torch.bool), batch[1:] != batch[:-1]])).flatten().tolist()
    starts.append(N)
    for s, e in zip(starts[:-1], starts[1:]):
        n = e - s
        if n <= 1:
            continue
        p = x[s:e]
        d = p[:, None, :] - p[None, :, :]
        d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        d2 = d2.clone()
        d2.fill_diagonal_(float("inf"))
        kk = min(k, n - 1)
        # stable sort => ties broken by ascending index
        order = torch.sort(d2, dim=1, stable=True).indices[:, :kk]
        src_all.append((order + s).reshape(-1))
        dst_all.append(torch.arange(s, e).repeat_interleave(kk))
    if not src_all:
        return torch.zeros(2, 0, dtype=torch.long)
    return torch.stack([torch.cat(src_all), torch.cat(dst_all)])


def build_edge_type(edge_index, lig_flag):
    """repo/modules/e3nn/unitransformer.py:88-99 -> one-hot [E,4] (int64 there;
    float here -- the later torch.cat promotes it to float anyway)."""
    src, dst = edge_index
    n_src = lig_flag[src].bool()
    n_dst = lig_flag[dst].bool()
    t = torch.zeros(src.shape[0], dtype=torch.long)
    t[n_src & n_dst] = 0
    t[n_src & ~n_dst] = 1
    t[~n_src & n_dst] = 2
    t[~n_src & ~n_dst] = 3
    return t


def gaussian_smearing(dist, dtype):
    """repo/modules/common.py:114-133 (fixed offsets, coeff -0.5)."""
    off = torch.tensor(RBF_OFFSETS, dtype=dtype)
    d = dist - off.view(*[1] * (dist.dim() - 1), -1)
    return torch.exp(RBF_COEFF * torch.pow(d, 2))


def mlp(sd, prefix, z):
    """repo/modules/common.py:151-171: Linear -> LayerNorm -> ReLU -> Linear."""
    w0, b0 = sd[prefix + ".net.0.weight"], sd[prefix + ".net.0.bias"]
    g, be = sd[prefix + ".net.1.weight"], sd[prefix + ".net.1.bias"]
    w1, b1 = sd[prefix + ".net.3.weight"], sd[prefix + ".net.3.bias"]
    y = F.linear(z, w0, b0)
    y = F.layer_norm(y, (y.shape[-1],), g, be, 1e-5)
    y = F.relu(y)
    return F.linear(y, w1, b1)


def scatter_softmax(src, index, n):
    """torch_scatter.scatter_softmax(src, index, dim=0, dim_size=n): per-segment
    max-subtracted softmax, no epsilon (x2h_attention.py:86, h2x_attention.py:67)."""
    idx = index.view(-1, *[1] * (src.dim() - 1)).expand_as(src)
    mx = torch.full((n,) + src.shape[1:], float("-inf"), dtype=src.dtype)
    mx = mx.scatter_reduce(0, idx, src, reduce="amax", include_self=True)
    ex = (src - mx.gather(0, idx)).exp()
    den = torch.zeros((n,) + src.shape[1:], dtype=src.dtype).scatter_add_(0, idx, ex)
    return ex / den.gather(0, idx)


def scatter_sum(src, index, n):
    """torch_scatter.scatter_sum(src, index, dim=0, dim_size=n)."""
    idx = index.view(-1, *[1] * (src.dim() - 1)).expand_as(src)
    return torch.zeros((n,) + src.shape[1:], dtype=src.dtype).scatter_add_(0, idx, src)


def _kv_input(x, h, edge_type, edge_index):
    """x2h_attention.py:46-65 / h2x_attention.py:37-49: [onehot(4) | type (x) rbf (80) | h_dst | h_src]."""
    src, dst = edge_index
    rel_x = x[dst] - x[src]
    dist = torch.norm(rel_x, p=2, dim=-1, keepdim=True)
    r = gaussian_smearing(dist, x.dtype)                       # [E,20]
    onehot = F.one_hot(edge_type, 4).to(x.dtype)               # [E,4]
    r_feat = (onehot.unsqueeze(-1) * r.unsqueeze(-2)).reshape(r.shape[0], -1)  # common.py:61-68
    z = torch.cat([onehot, r_feat, h[dst], h[src]], -1)        # [E,340]
    return z, rel_x


def x2h_attention(sd, prefix, x, h, edge_type, edge_index, e_w, n_heads=16):
    """repo/modules/attention/x2h_attention.py:43-97 (ew_net_type='global', out_fc=False)."""
    N = h.shape[0]
    src, dst = edge_index
    z, _ = _kv_input(x, h, edge_type, edge_index)
    dh = h.shape[-1] // n_heads
    k = mlp(sd, prefix + ".hk_func", z).view(-1, n_heads, dh)
    v = mlp(sd, prefix + ".hv_func", z) * e_w
    v = v.view(-1, n_heads, dh)
    q = mlp(sd, prefix + ".hq_func", h).view(-1, n_heads, dh)
    alpha = scatter_softmax((q[dst] * k / math.sqrt(dh)).sum(-1), dst, N)
    m = alpha.unsqueeze(-1) * v
    out = scatter_sum(m, dst, N).view(N, -1)
    return out + h


def h2x_attention(sd, prefix, x, h, edge_type, edge_index, e_w, n_heads=16):
    """repo/modules/attention/h2x_attention.py:34-73 (ew_net_type='global')."""
    N = h.shape[0]
    src, dst = edge_index
    z, rel_x = _kv_input(x, h, edge_type, edge_index)
    dh = h.shape[-1] // n_heads
    k = mlp(sd, prefix + ".xk_func", z).view(-1, n_heads, dh)
    v = mlp(sd, prefix + ".xv_func", z) * e_w.view(-1, 1)
    v = v.unsqueeze(-1) * rel_x.unsqueeze(1)
    q = mlp(sd, prefix + ".xq_func", h).view(-1, n_heads, dh)
    alpha = scatter_softmax((q[dst] * k / math.sqrt(dh)).sum(-1), dst, N)
    m = alpha.unsqueeze(-1) * v
    out = scatter_sum(m, dst, N)
    return out.mean(1)


def edge_gate(sd, prefix, x, edge_index):
    """unitransformer.py:109-112 + embs/dist_emb.py:6-9: sigmoid(MLP(20->160->1)(rbf(|x_dst-x_src|)))."""
    src, dst = edge_index
    dist = torch.norm(x[dst] - x[src], p=2, dim=-1, keepdim=True)
    return torch.sigmoid(mlp(sd, prefix + ".dist_emb.1", gaussian_smearing(dist, x.dtype)))


def e3_layer(sd, prefix, x, h, edge_type, edge_index, e_w, gen_flag, n_heads=16):
    """E3DualAttentionLayer.forward, unitransformer.py:167-186 (num_x2h = num_h2x = 1)."""
    h_out = x2h_attention(sd, prefix + ".x2h_layers.0", x, h, edge_type, edge_index, e_w, n_heads)
    delta_x = h2x_attention(sd, prefix + ".h2x_layers.0", x, h_out, edge_type, edge_index, e_w, n_heads)
    x_out = x + delta_x * gen_flag.unsqueeze(-1).to(x.dtype)
    return x_out, h_out


def classifier(sd, prefix, h):
    """unitransformer.py:46-51,119-120: Linear -> ShiftedSoftplus (common.py:174-180) -> Linear."""
    y = F.linear(h, sd[prefix + ".classifier.0.weight"], sd[prefix + ".classifier.0.bias"])
    y = F.softplus(y) - math.log(2.0)
    return F.linear(y, sd[prefix + ".classifier.2.weight"], sd[prefix + ".classifier.2.bias"])


def count_layers(sd, prefix):
    n = 0
    while f"{prefix}.blocks.{n}.x2h_layers.0.hk_func.net.0.weight" in sd:
        n += 1
    return n


def unitransformer_forward(sd, x, h, batch_idx, lig_flag, gen_flag, prefix="denoiser", k=32,
                           n_heads=16, return_intermediates=False):
    """UniTransformer.forward, repo/modules/e3nn/unitransformer.py:102-123
    (cutoff_mode='knn', ew_type='global', num_blocks=1)."""
    dtype = x.dtype
    sd = {kk: (vv.to(dtype) if vv.is_floating_point() else vv) for kk, vv in sd.items() if kk.startswith(prefix)}
    edge_index = knn_graph(x, batch_idx, k)
    edge_type = build_edge_type(edge_index, lig_flag)
    e_w = edge_gate(sd, prefix, x, edge_index)
    inter = {"edge_index": edge_index, "edge_type": edge_type, "e_w": e_w, "layers": []}
    for l in range(count_layers(sd, prefix)):
        x, h = e3_layer(sd, f"{prefix}.blocks.{l}", x, h, edge_type, edge_index, e_w, gen_flag, n_heads)
        if return_intermediates:
            inter["layers"].append((x.clone(), h.clone()))
    logits = classifier(sd, prefix, h)
    if return_intermediates:
        return x, h, logits, inter
    return x, h, logits
