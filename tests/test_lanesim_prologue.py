"""Lane-level models (numpy, CPU) of the index logic round 5 put into the small-input kernels: what the 64 lanes of a wave compute
must equal the plain definition it replaced.  The kernels themselves are compared with goldens / the oracle by the GPU tests; these
models pin the bit tricks (ds_bpermute source lanes, ballot masks, 64-bit key order, the four-at-a-time ligand walk) so that a
change of the lane mapping cannot silently pass on inputs the GPU fixtures do not happen to contain (partial tiles, rows past a
list's end, equal distances)."""
import numpy as np
import pytest

LANES = np.arange(64)
C, Q = LANES & 15, LANES >> 4


def bpermute(byte_addr, data):
    """ds_bpermute_b32: lane l reads data[(byte_addr[l] >> 2) & 63]"""
    return data[(byte_addr >> 2) & 63]


def ballot(pred):
    return int(sum(1 << int(l) for l in LANES[pred]))


@pytest.mark.parametrize("n_rows,tile", [(16, 0), (17, 1), (5, 0), (100, 6), (64, 3), (33, 2)])
@pytest.mark.parametrize("listed", [True, False])
def test_node_stage_tile_prologue(n_rows, tile, listed):
    """node_stage_kernel / node_qfold_kernel / node_qmlp_kernel (node_mfma.hip): lane (c, q) loads the list entry of row c of the
    tile (clamped past the end); the rows it WRITES, 4q + r, are lane 4q + r's entry; the per-row predicates are ballots over
    lanes 0..15, shifted by 4q."""
    rng = np.random.default_rng(n_rows * 31 + tile)
    n_nodes = 500
    rows = rng.permutation(n_nodes)[:n_rows].astype(np.int32) if listed else None
    lig = rng.integers(0, 2, n_nodes).astype(np.uint8)
    fold_flag = rng.integers(0, 2, n_nodes).astype(np.uint8)
    row0 = tile * 16
    assert row0 < n_rows
    ak = np.minimum(row0 + C, n_rows - 1)
    arow = rows[ak] if listed else ak
    for with_flag in (True, False):
        lig_b = lig[arow]
        fold_b = (fold_flag if with_flag else lig)[arow]
        valid16 = 0xFFFF if n_rows - row0 >= 16 else (1 << (n_rows - row0)) - 1
        lig16 = ballot(lig_b != 0) & 0xFFFF
        fold16 = ((ballot(fold_b != 0) & 0xFFFF) if with_flag else 0xFFFF) & valid16
        lgm = (lig16 >> (4 * Q)) & 15
        fm = (fold16 >> (4 * Q)) & 15
        for r in range(4):
            o = bpermute((4 * Q + r) << 2, arow)
            k = row0 + 4 * Q + r
            orow = np.where(k < n_rows, o, -1)
            # the definition: entry k of the list (or k itself), -1 past the end
            want = np.where(k < n_rows, (rows[np.minimum(k, n_rows - 1)] if listed else k), -1)
            assert np.array_equal(orow, want)
            real = orow >= 0
            assert np.array_equal(((lgm >> r) & 1)[real], (lig[orow[real]] != 0).astype(int))
            want_fold = (fold_flag[orow[real]] != 0) if with_flag else np.ones(real.sum(), bool)
            assert np.array_equal(((fm >> r) & 1)[real].astype(bool), want_fold)
            assert not ((fm >> r) & 1)[~real].any()          # rows past the list's end never ask for a fold


def test_key_order_is_one_64_bit_compare():
    """graph_mfma.hip key_less / take_min: the lexicographic order of (bits(d2), index) is the order of hi << 32 | lo"""
    rng = np.random.default_rng(0)
    hi = rng.integers(0, 2**32, 4000, dtype=np.uint64)
    lo = rng.integers(0, 2**32, 4000, dtype=np.uint64)
    hi[::7] = hi[1::7][: len(hi[::7])]            # plenty of equal distances: the index decides
    a_h, a_l, b_h, b_l = hi[:2000], lo[:2000], hi[2000:], lo[2000:]
    a_h[:300] = b_h[:300]
    lex = (a_h < b_h) | ((a_h == b_h) & (a_l < b_l))
    packed = ((a_h << np.uint64(32)) | a_l) < ((b_h << np.uint64(32)) | b_l)
    assert np.array_equal(lex, packed)


@pytest.mark.parametrize("n_prot,n_lig", [(30, 0), (30, 1), (31, 4), (40, 5), (50, 8), (33, 13), (3, 3)])
def test_ligand_walk_four_at_a_time(n_prot, n_lig):
    """graph_cache_begin_kernel (graph_mfma.hip): the proximity flag of a protein atom = any ligand atom of its graph (the run of
    flagged rows that closes the graph) closer than the atom's cached 32nd-neighbour distance.  The kernel walks the run from the
    end four rows per step with clamped, unconditional loads and an `alive` bit; the definition stops at the first unflagged row."""
    rng = np.random.default_rng(n_prot * 17 + n_lig)
    gs = 7
    n = n_prot + n_lig
    ge = gs + n
    lig = np.zeros(gs + n + 5, np.uint8)
    lig[gs + n_prot:ge] = 1
    if n_prot > 2:
        lig[gs + 1] = 1                               # a flagged row that is NOT part of the closing run must not count
    d2 = rng.random(gs + n + 5).astype(np.float32)
    for lim in (0.0, 0.05, 0.3, 2.0):
        want = False
        j = ge - 1
        while j >= gs and lig[j]:
            want |= bool(d2[j] < lim)
            j -= 1
        got, alive = False, True
        j = ge - 1
        while alive and j >= gs:
            for u in range(4):
                jj = j - u if j - u >= gs else gs
                alive = alive and (j - u >= gs) and lig[jj] != 0
                got |= alive and bool(d2[jj] < lim)
            j -= 4
        assert got == want


def test_first_graph_of_a_workgroup_then_steps():
    """graph_cache_begin_kernel: the graph of node i = a (scalar) search for the workgroup's first node, then steps forward per
    thread -- the same graph as a search for i itself, empty graphs included."""
    graph_ptr = np.array([0, 0, 300, 300, 1500, 1501, 2600, 2600], np.int64)     # graphs 0, 2 and 6 are empty
    n_graphs = len(graph_ptr) - 1

    def search(i):
        lo, hi = 0, n_graphs
        while hi - lo > 1:
            mid = (lo + hi) >> 1
            if graph_ptr[mid] <= i:
                lo = mid
            else:
                hi = mid
        return lo

    for block in range(3):
        g0 = search(block * 1024)
        for i in range(block * 1024, min((block + 1) * 1024, int(graph_ptr[-1]))):
            g = g0
            while g + 1 < n_graphs and graph_ptr[g + 1] <= i:
                g += 1
            assert g == search(i)
            assert graph_ptr[g] <= i < graph_ptr[g + 1]
