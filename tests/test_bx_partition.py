"""Index arithmetic of edge_backward_x2h_kernel's schedule (cbgbench_amd/csrc/train_bwd_x2h.hip: static rounds per XCD, the last
BX_DYN_ROUNDS - 1 full rounds and the partial one handed out through a per-XCD counter), restated in Python: for every node count and
grid the kernel can be launched with, every node is processed exactly once, no index leaves [0, count), and the dynamic part is the
tail of each XCD's range.  The GPU tests run the kernel itself at fixture sizes (fewer nodes than waves: everything dynamic) and at the
configs[4] shape (eight static rounds)."""
import itertools

import pytest

BX_WAVES, DYN = 8, 2


def schedule(count, grid):
    """-> (list of nodes per (workgroup, wave) in processing order for ONE interleaving of the dynamic grabs, set of dynamic nodes)"""
    done, dyn = [], set()
    ctr = [0] * 8
    state = {}
    for wg, wave in itertools.product(range(grid), range(BX_WAVES)):
        if grid % 8 == 0:
            per_xcd = (((count + 7) >> 3) + BX_WAVES - 1) // BX_WAVES * BX_WAVES
            xcd, slot = wg & 7, wg >> 3
            base = min(count, xcd * per_xcd)
            it_end = min(count, (xcd + 1) * per_xcd)
            stride = (grid >> 3) * BX_WAVES
            it = base + slot * BX_WAVES + wave
            full = max((it_end - base) // stride - (DYN - 1), 0)
            tail_base = base + full * stride
            c = xcd
        else:
            it, it_end, stride = wg * BX_WAVES + wave, count, grid * BX_WAVES
            full = max(count // stride - (DYN - 1), 0)
            tail_base = full * stride
            c = 0
        state[(wg, wave)] = dict(it=it, it_end=it_end, stride=stride, full=full, tail=tail_base, c=c, round=0, started=False)
    # run all waves round-robin (any interleaving hands out the same SET of dynamic nodes)
    active = list(state)
    while active:
        nxt = []
        for key in active:
            w = state[key]
            if not w["started"]:
                w["started"] = True
                if w["full"] == 0:
                    w["it"] = w["tail"] + ctr[w["c"]]; ctr[w["c"]] += 1
                    w["dynamic"] = True
                else:
                    w["dynamic"] = False
            if w["it"] >= w["it_end"]:
                continue
            assert 0 <= w["it"] < count
            done.append(w["it"])
            if w["dynamic"]:
                dyn.add(w["it"])
            if w["round"] + 1 < w["full"]:
                w["it"] += w["stride"]; w["round"] += 1
            else:
                w["round"] = w["full"]
                w["it"] = w["tail"] + ctr[w["c"]]; ctr[w["c"]] += 1
                w["dynamic"] = True
            nxt.append(key)
        active = nxt
    return done, dyn


@pytest.mark.parametrize("grid", [1, 2, 7, 8, 16, 64, 200, 256])
def test_every_node_exactly_once(grid):
    for count in [0, 1, 7, 8, 9, 63, 64, 65, 500, 2047, 2048, 2049, 4404, 16506, 16384, 16385, 20000]:
        done, dyn = schedule(count, grid)
        assert sorted(done) == list(range(count)), (grid, count)
        # the dynamic nodes are the tail of each XCD's range: at most DYN rounds of the XCD's waves plus a partial round
        if grid % 8 == 0 and count:
            per_xcd = (((count + 7) >> 3) + BX_WAVES - 1) // BX_WAVES * BX_WAVES
            stride = (grid >> 3) * BX_WAVES
            for xcd in range(8):
                lo, hi = min(count, xcd * per_xcd), min(count, (xcd + 1) * per_xcd)
                d = sorted(n for n in dyn if lo <= n < hi)
                assert d == list(range(hi - len(d), hi)) and len(d) < (DYN + 1) * stride, (grid, count, xcd)


def test_config5_shape_runs_seven_static_rounds_and_hands_out_the_rest():
    done, dyn = schedule(16506, 256)          # 2048 waves, 8.06 nodes per wave
    assert len(done) == 16506
    per_wave_static = (16506 - len(dyn)) / 2048
    assert per_wave_static == 7.0 and len(dyn) == 16506 - 7 * 2048


# ---- the forward edge kernels' variant schedule (edge_mfma.hip, -DCBGX_EDGE_DYN=d: untested on the GPU, off in every build) --------
def forward_schedule(n_items, n_wg, D, waves=8, with_counters=True):
    """edge_body's loop with CBGX_EDGE_DYN = D restated: the first max(full - (D - 1), 1) rounds of an XCD's range static, the rest
    claimed one node ahead through the XCD's counter.  -> processed items in order (one interleaving)"""
    ctr = [0] * 8
    waves_state = []
    for wg in range(n_wg):
        if n_wg % 8 == 0:
            per_xcd = (((n_items + 7) >> 3) + waves - 1) // waves * waves
            first = (wg & 7) * per_xcd + (wg >> 3) * waves
            if first >= min(n_items, ((wg & 7) + 1) * per_xcd):
                continue
        elif wg * waves >= n_items:
            continue
        for wave in range(waves):
            if n_wg % 8 == 0:
                xcd, slot = wg & 7, wg >> 3
                i_begin = xcd * per_xcd + slot * waves + wave
                i_end = min(n_items, (xcd + 1) * per_xcd)
                i_step = (n_wg >> 3) * waves
                base, c = min(n_items, xcd * per_xcd), xcd
            else:
                i_begin, i_end, i_step, base, c = wg * waves + wave, n_items, n_wg * waves, 0, 0
            if i_begin >= i_end:
                continue
            full = (i_end - base) // i_step
            static = max(full - (D - 1), 1)
            tail = base + static * i_step
            on = with_counters and full >= 1 and tail < i_end
            waves_state.append(dict(k=i_begin, end=i_end, step=i_step, static=static, tail=tail, on=on, c=c, round=0))
    done = []
    active = waves_state
    while active:
        nxt = []
        for w in active:
            assert 0 <= w["k"] < n_items
            done.append(w["k"])
            k_next = w["k"] + w["step"]
            if not w["on"]:
                more = k_next < w["end"]
            elif w["round"] + 1 < w["static"]:
                more = True
            else:
                k_next = w["tail"] + ctr[w["c"]]; ctr[w["c"]] += 1
                more = k_next < w["end"]
            if more:
                w["k"] = k_next; w["round"] += 1
                nxt.append(w)
        active = nxt
    return done


@pytest.mark.parametrize("D", [1, 2, 3])
@pytest.mark.parametrize("n_wg", [1, 2, 9, 56, 57, 64, 176, 256])
def test_forward_variant_schedule_covers_every_item_once(n_wg, D):
    for n in [1, 5, 8, 9, 445, 1230, 3170, 4404, 16506, 99543]:
        assert sorted(forward_schedule(n, n_wg, D)) == list(range(n)), (n, n_wg, D)
        assert sorted(forward_schedule(n, n_wg, D, with_counters=False)) == list(range(n))      # no counters: the static schedule
