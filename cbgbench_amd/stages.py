"""Thin Python wrappers over the per-stage C-ABI entry points (include/cbgx.h).  The denoiser itself
uses the single ``cbgx_unitransformer_forward`` call; these exist so parity tests (and anyone porting
a different caller) can exercise each stage against the reference function it replaces."""
import torch

from . import _native


def _stream(t):
    return _native.current_stream(t.device)


def knn_graph(x, graph_ptr, k=32):
    """-> (nbr [N,32] int32, deg [N] int32); torch_cluster.knn_graph at unitransformer.py:80."""
    N = x.shape[0]
    nbr = torch.empty(N, 32, dtype=torch.int32, device=x.device)
    deg = torch.empty(N, dtype=torch.int32, device=x.device)
    rc = _native.lib().cbgx_knn_graph(_native.ptr(x), _native.ptr(graph_ptr), graph_ptr.numel() - 1, N, k,
                                      _native.ptr(nbr), _native.ptr(deg), _stream(x))
    _native.check(rc, "cbgx_knn_graph")
    return nbr, deg


def edge_index_from_nbr(nbr, deg):
    """[2,E] (src=neighbour, dst=centre) in the reference's grouped-by-centre order."""
    N = nbr.shape[0]
    slot = torch.arange(32, device=nbr.device)[None, :].expand(N, 32)
    mask = slot < deg[:, None]
    dst = torch.arange(N, device=nbr.device)[:, None].expand(N, 32)[mask]
    return torch.stack([nbr[mask].long(), dst.long()])


def edge_gate(packed, x, nbr, deg):
    e_w = torch.empty(x.shape[0], 32, dtype=torch.float32, device=x.device)
    rc = _native.lib().cbgx_edge_gate(_native.ptr(packed), _native.ptr(x), _native.ptr(nbr), _native.ptr(deg),
                                      x.shape[0], _native.ptr(e_w), _stream(x))
    _native.check(rc, "cbgx_edge_gate")
    return e_w


def _ws(n, device):
    return torch.empty(_native.lib().cbgx_workspace_bytes(n, 1), dtype=torch.uint8, device=device)


def x2h_attention(packed, layer, x, h, nbr, deg, lig_flag, e_w):
    N = x.shape[0]
    out = torch.empty_like(h)
    ws = _ws(N, x.device)
    rc = _native.lib().cbgx_x2h_attention(_native.ptr(packed), layer, _native.ptr(x), _native.ptr(h),
                                          _native.ptr(nbr), _native.ptr(deg), _native.ptr(lig_flag),
                                          _native.ptr(e_w), N, _native.ptr(out), _native.ptr(ws), ws.numel(),
                                          _stream(x))
    _native.check(rc, "cbgx_x2h_attention")
    return out


def h2x_attention(packed, layer, x, h, nbr, deg, lig_flag, gen_flag, e_w):
    N = x.shape[0]
    x_out = torch.empty_like(x)
    dx = torch.empty_like(x)
    ws = _ws(N, x.device)
    rc = _native.lib().cbgx_h2x_attention(_native.ptr(packed), layer, _native.ptr(x), _native.ptr(h),
                                          _native.ptr(nbr), _native.ptr(deg), _native.ptr(lig_flag),
                                          _native.ptr(gen_flag), _native.ptr(e_w), N, _native.ptr(x_out),
                                          _native.ptr(dx), _native.ptr(ws), ws.numel(), _stream(x))
    _native.check(rc, "cbgx_h2x_attention")
    return x_out, dx


def classifier(packed, num_layers, num_classes, h):
    N = h.shape[0]
    logits = torch.empty(N, num_classes, dtype=torch.float32, device=h.device)
    ws = _ws(N, h.device)
    rc = _native.lib().cbgx_classifier(_native.ptr(packed), num_layers, num_classes, _native.ptr(h), N,
                                       _native.ptr(logits), _native.ptr(ws), ws.numel(), _stream(h))
    _native.check(rc, "cbgx_classifier")
    return logits


# ---- backward of single attention blocks (training; include/cbgx.h "training" section) ---------------------
_MLP_SHAPES_X2H = [(128, 340), (128,), (128,), (128,), (128, 128), (128,)] * 2 + \
                  [(128, 128), (128,), (128,), (128,), (128, 128), (128,)]
_MLP_SHAPES_H2X = [(128, 340), (128,), (128,), (128,), (128, 128), (128,)] + \
                  [(128, 340), (128,), (128,), (128,), (16, 128), (16,)] + \
                  [(128, 128), (128,), (128,), (128,), (128, 128), (128,)]


def _train_ws(n, device):
    return torch.empty(_native.lib().cbgx_train_workspace_bytes(n), dtype=torch.uint8, device=device)


def _grad_tensors(shapes, device):
    import ctypes
    gs = [torch.empty(s, dtype=torch.float32, device=device) for s in shapes]
    arr = (ctypes.c_void_p * len(gs))(*[g.data_ptr() for g in gs])
    return gs, arr


def x2h_attention_backward(packed, layer, x, h, nbr, deg, lig_flag, e_w, grad_h_out):
    """-> (grad_h, grad_x, grad_e_w, [18 parameter gradients: hk_func(6), hv_func(6), hq_func(6)])."""
    N = x.shape[0]
    gh, gx = torch.empty_like(h), torch.empty_like(x)
    gew = torch.empty_like(e_w)
    grads, arr = _grad_tensors(_MLP_SHAPES_X2H, x.device)
    ws = _train_ws(N, x.device)
    rc = _native.lib().cbgx_x2h_attention_backward(
        _native.ptr(packed), layer, _native.ptr(x), _native.ptr(h), _native.ptr(nbr), _native.ptr(deg),
        _native.ptr(lig_flag), _native.ptr(e_w), N, _native.ptr(grad_h_out.contiguous()), _native.ptr(gh),
        _native.ptr(gx), _native.ptr(gew), arr, _native.ptr(ws), ws.numel(), _stream(x))
    _native.check(rc, "cbgx_x2h_attention_backward")
    return gh, gx, gew, grads


def h2x_attention_backward(packed, layer, x, h, nbr, deg, lig_flag, gen_flag, e_w, grad_x_out):
    """-> (grad_h, grad_x, grad_e_w, [18 parameter gradients: xk_func(6), xv_func(6), xq_func(6)])."""
    N = x.shape[0]
    gh, gx = torch.empty_like(h), torch.empty_like(x)
    gew = torch.empty_like(e_w)
    grads, arr = _grad_tensors(_MLP_SHAPES_H2X, x.device)
    ws = _train_ws(N, x.device)
    rc = _native.lib().cbgx_h2x_attention_backward(
        _native.ptr(packed), layer, _native.ptr(x), _native.ptr(h), _native.ptr(nbr), _native.ptr(deg),
        _native.ptr(lig_flag), _native.ptr(gen_flag), _native.ptr(e_w), N, _native.ptr(grad_x_out.contiguous()),
        _native.ptr(gh), _native.ptr(gx), _native.ptr(gew), arr, _native.ptr(ws), ws.numel(), _stream(x))
    _native.check(rc, "cbgx_h2x_attention_backward")
    return gh, gx, gew, grads
